"""The one-launch truncated SVD of centre matrices up to 64 x 64 (svd_trunc_small_kernel) on graded test matrices: executed Hestenes
sweeps, error against LAPACK, orthogonality (TN_SVD_SMALL=0: the block path beside it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from tnac4o_amd import ops
from test_gpu_kernels import _svd_case
for (k, n, seed) in ((64, 64, 1), (64, 64, 2), (60, 60, 1), (48, 64, 2), (64, 40, 1)):
    T = torch.as_tensor(_svd_case(k, n, seed)).cuda()
    Sr = np.linalg.svd(T.cpu().numpy(), compute_uv=False)
    for mode in ('1', '0'):
        os.environ['TN_SVD_SMALL'] = mode
        U, S, V, kk, d, i = ops.svd_trunc(T, 64, 1e-8)
        err = np.abs(S.cpu().numpy() - Sr[:kk]).max() / Sr[0]
        Uh, Vh = U.cpu().numpy(), V.cpu().numpy()
        print(k, n, seed, 'small' if mode == '1' else 'block', 'keep', kk, 'err/S0 %.2e' % err, 'sweeps', i['sweeps'],
              'orthU %.1e orthV %.1e' % (np.abs(Uh.T @ Uh - np.eye(kk)).max(), np.abs(Vh @ Vh.T - np.eye(kk)).max()))
