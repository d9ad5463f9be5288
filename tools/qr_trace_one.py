import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tnac4o_amd import ops
m, n = 16384, 1024
T = torch.randn((m, n), dtype=torch.float64, device='cuda')
Q = torch.empty((m, n), dtype=torch.float64, device='cuda'); R = torch.empty((n, n), dtype=torch.float64, device='cuda')
ops.qr_into(T.clone(), Q, R, overwrite=True)
torch.cuda.synchronize()
ops.qr_into(T.clone(), Q, R, overwrite=True)
torch.cuda.synchronize()
