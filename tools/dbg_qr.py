import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from tnac4o_amd import ops
T = np.ones((90, 70))
Q, Rm = ops.qr(torch.as_tensor(T).cuda())
print(np.isnan(Q.cpu().numpy()).sum(), np.isnan(Rm.cpu().numpy()).sum())
