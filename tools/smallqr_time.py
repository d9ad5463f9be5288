"""Kernel-time probe of the one-launch factorisation: run under rocprofv3 --kernel-trace --stats."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tnac4o_amd import ops
g = torch.Generator(device='cpu').manual_seed(1)
shapes = [tuple(int(x) for x in a.split('x')) for a in sys.argv[1:]] or [(1024, 64)]
for m, n in shapes:
    T = torch.randn(m, n, dtype=torch.float64, generator=g).cuda()
    Q = torch.empty(m, n, dtype=torch.float64, device='cuda')
    R = torch.empty(n, n, dtype=torch.float64, device='cuda')
    for _ in range(100):
        ops.qr_into(T, Q, R, overwrite=True)
    torch.cuda.synchronize()
