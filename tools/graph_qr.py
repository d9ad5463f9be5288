import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from tnac4o_amd import ops
for (m, n) in [(16384, 1024), (4096, 1024), (1024, 64)]:
    T0 = torch.randn(m, n, dtype=torch.float64, device='cuda')
    T = T0.clone(); Q = torch.empty(m, min(m, n), dtype=torch.float64, device='cuda'); Rm = torch.empty(min(m, n), n, dtype=torch.float64, device='cuda')
    def direct():
        T.copy_(T0); ops.qr_into(T, Q, Rm, overwrite=True)
    direct(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): direct()
    torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 5
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        direct(); torch.cuda.synchronize()          # workspace for this stream
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            direct()
        torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): g.replay()
        torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 5
    err = float((Q @ Rm - T0).abs().max())
    print('QR %6d x %5d: direct %.3f ms   graph replay %.3f ms   residual after replay %.1e' % (m, n, td * 1e3, tg * 1e3, err))
