"""Left Gram recursion of the weighted first pass: product with the absorbed tensor (2 GEMMs, 69 GFLOP at the bulk shape) against
the structured form through the MPS (x) MPO factors (mps._gram_step_structured, 21 GFLOP): agreement and time per site."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tnac4o_amd import ops, mps

torch.manual_seed(3)
for (Dl, p, Dr, b) in [(64, 16, 64, 16), (32, 16, 64, 16), (64, 8, 48, 8)]:
    for hconj in (True, False):
        A = torch.randn(Dl, p, Dr, dtype=torch.float64, device='cuda')
        W = torch.randn(b, p, b, p, dtype=torch.float64, device='cuda')
        T = ops.absorb(A, W, hconj)
        na = T.shape[0]
        X0 = torch.randn(na, na, dtype=torch.float64, device='cuda')
        G = X0 @ X0.t() / na

        def old():
            d0, pp, d1 = T.shape
            X = ops.mm(G, T.view(d0, pp * d1))
            return ops.mm(T.view(d0 * pp, d1).t(), X.view(d0 * pp, d1))

        def new():
            return mps._gram_step_structured(G, A, W, hconj)

        g0, g1 = old(), new()
        err = ((g0 - g1).abs().max() / g0.abs().max()).item()
        ts = []
        for f in (old, new):
            for _ in range(2):
                f()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                f()
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0) / 5)
        print('Dl=%d p=%d Dr=%d b=%d hconj=%d: rel diff %.1e   absorbed-tensor form %.3f ms   structured %.3f ms' % (Dl, p, Dr, b, hconj, err, ts[0], ts[1]), flush=True)
        assert err < 1e-12
