import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from tnac4o_amd import ops
A = torch.randn(64, 16, 64, dtype=torch.float64, device='cuda'); W = torch.randn(16, 16, 16, 16, dtype=torch.float64, device='cuda')
for h in (True, False):
    for _ in range(3): T = ops.absorb(A, W, h)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): T = ops.absorb(A, W, h)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print('absorb hconj=%s: %.1f us  %.2f TB/s (algorithmic 135.3 MB)' % (h, dt * 1e6, 135.3e6 / dt / 1e12))
