import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import golden_inputs as gi
from tnac4o_amd import ops
T = gi.g1_matrix((256, 64), 'rankdef')
Ur, Sr, Vr = np.linalg.svd(T, full_matrices=False)
U, S, Vt, keep, disc, info = ops.svd_trunc(torch.as_tensor(T).cuda(), 64, 1e-3)
U, S, Vt = U.cpu().numpy(), S.cpu().numpy(), Vt.cpu().numpy()
print('keep', keep, 'info', info)
print('recon err', np.abs((U * S) @ Vt - (Ur[:, :keep] * Sr[:keep]) @ Vr[:keep]).max())
print('S err', np.abs(S - Sr[:keep]).max(), 'orthU', np.abs(U.T @ U - np.eye(keep)).max(), 'orthV', np.abs(Vt @ Vt.T - np.eye(keep)).max())
