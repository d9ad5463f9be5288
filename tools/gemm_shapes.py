"""GEMM rates on the operand layouts of the two-level QR (Y, A, Q are slices of row-major 16384 x 1024 arrays)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tnac4o_amd import ops

m, n = 16384, 1024
Y = torch.randn((m, n), dtype=torch.float64, device='cuda'); A = torch.randn((m, n), dtype=torch.float64, device='cuda')


def rate(name, fn, flops, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print('%-64s %8.1f us  %6.1f TFLOP/s' % (name, 1e6 * dt, flops / dt / 1e12), flush=True)


for bw in (32, 128, 256):
    nr = n - bw
    Yb, Ar = Y[:, :bw], A[:, bw:]
    Z = torch.empty((bw, nr), dtype=torch.float64, device='cuda')
    rate('TN  Z = Yb^T Ar      (%d x %d, K=%d)' % (bw, nr, m), lambda: ops.mm(Yb.t(), Ar, out=Z), 2.0 * bw * nr * m)
    rate('NN  Ar -= Yb Z       (%d x %d, K=%d)' % (m, nr, bw), lambda: ops.mm(Yb, Z, out=Ar, alpha=-1.0, beta=1.0), 2.0 * bw * nr * m)
    Zc = torch.randn((bw, nr), dtype=torch.float64, device='cuda'); T = torch.randn((bw, bw), dtype=torch.float64, device='cuda')
    Z2 = torch.empty_like(Zc)
    rate('TN  Z2 = T^T Z       (%d x %d, K=%d)' % (bw, nr, bw), lambda: ops.mm(T.t(), Zc, out=Z2), 2.0 * bw * nr * bw)
G = torch.empty((256, 256), dtype=torch.float64, device='cuda')
rate('TN  G = Yb^T Yb      (256 x 256, K=%d)' % m, lambda: ops.mm(Y[:, :256].t(), Y[:, :256], out=G), 2.0 * 256 * 256 * m)
# contiguous operands for comparison
Yc = Y[:, :256].contiguous(); Ac = A[:, 256:].contiguous(); Zc = torch.empty((256, 768), dtype=torch.float64, device='cuda')
rate('TN  contiguous operands (256 x 768, K=%d)' % m, lambda: ops.mm(Yc.t(), Ac, out=Zc), 2.0 * 256 * 768 * m)
rate('NN  contiguous operands (%d x 768, K=256)' % m, lambda: ops.mm(Yc, Zc, out=Ac, alpha=-1.0, beta=1.0), 2.0 * 256 * 768 * m)
