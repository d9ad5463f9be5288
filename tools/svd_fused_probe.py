"""The Jacobi rounds of tn_svd_trunc in one launch (svdl_kernel: vectors resident in LDS) against the three-launches-per-round form (TN_SVD_FUSED=0):
results must be identical bit for bit; time per call on the shapes of the headline workload (triangular factors with a decaying
spectrum, k = 128 / 192 rows x 200-900 columns)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tnac4o_amd import ops


def case(k, n, seed, decay=0.25):
    g = np.random.default_rng(seed)
    U, _ = np.linalg.qr(g.standard_normal((k, k)))
    V, _ = np.linalg.qr(g.standard_normal((n, k)))
    s = np.exp(-decay * np.arange(k))
    A = (U * s) @ V.T
    return torch.as_tensor(np.triu(A) if seed % 2 else A).cuda()


def run(T, fused):
    os.environ['TN_SVD_FUSED'] = str(int(fused))          # 0: three launches per round, 1: one launch where the vectors fit in LDS
    out = ops.svd_trunc(T, 64, 1e-8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        out = ops.svd_trunc(T, 64, 1e-8)
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / reps * 1e3


bad = 0
for (k, n) in ((128, 200), (128, 450), (192, 300), (192, 600), (192, 900), (100, 100), (64, 700), (256, 1000), (70, 3000)):
    for seed in (1, 2):
        T = case(k, n, seed)
        (U0, S0, V0, k0, d0, i0), t0 = run(T, 0)
        (U1, S1, V1, k1, d1, i1), t1 = run(T, 1)
        same = k0 == k1 and torch.equal(U0, U1) and torch.equal(S0, S1) and torch.equal(V0, V1) and d0 == d1 and i0['sweeps'] == i1['sweeps']
        bad += 0 if same else 1
        print('%4d x %4d seed %d: keep %3d sweeps %2d   separate %.3f ms   one launch %.3f ms   %s' % (k, n, seed, k1, i1['sweeps'], t0, t1,
              'identical' if same else 'DIFFERENT (max |dS| %.2e)' % (float((S0 - S1).abs().max()) if k0 == k1 else -1)), flush=True)
print('FAILED' if bad else 'ok')
sys.exit(1 if bad else 0)
