import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from tnac4o_amd import ops
from test_gpu_fullsize import rnd
A = rnd((4096, 150), 7) * (10.0 ** (-torch.arange(150, dtype=torch.float64, device='cuda') / 10.0))
A = A @ rnd((150, 1024), 8)
_, R = ops.qr(A)
U, S, Vt, keep, disc, info = ops.svd_trunc(R, 256, 1e-17)
I = torch.eye(keep, dtype=torch.float64, device='cuda')
eu = (ops.mm(U.t(), U) - I).abs(); ev = (ops.mm(Vt, Vt.t()) - I).abs()
print('keep', keep, info, 'orthU', float(eu.max()), 'orthV', float(ev.max()))
iu = int(eu.argmax()); print('worst U pair', iu // keep, iu % keep, 'S there', float(S[iu // keep]), float(S[iu % keep]), 'S0', float(S[0]))
iv = int(ev.argmax()); print('worst V pair', iv // keep, iv % keep)
