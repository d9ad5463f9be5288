"""tn_svd_trunc on centre matrices like those of the truncating passes (triangular factors of numerically low rank): time per call,
executed sweeps, quality (U, V orthonormal, reconstruction).  TN_EIG_PIPELINED=0/1 selects the eig_small generation (read once per
process: run the script once per setting)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tnac4o_amd import ops


def case(n, rank, decay, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    rn = lambda *s: torch.randn(*s, dtype=torch.float64, generator=g).cuda()
    U, _ = torch.linalg.qr(rn(n, n))
    V, _ = torch.linalg.qr(rn(n, n))
    S = torch.logspace(0, -decay, n, dtype=torch.float64).cuda()
    S[rank:] *= 1e-3
    A = (U * S[None, :]) @ V.t()
    _, R = torch.linalg.qr(A)              # what the canonisation hands to truncateC: a triangular factor
    return R.contiguous(), S


def main():
    print('TN_EIG_PIPELINED =', os.environ.get('TN_EIG_PIPELINED', '(default 1)'))
    for (n, rank, decay, dmax) in [(1024, 300, 18, 256), (512, 200, 18, 256), (512, 500, 6, 256), (256, 100, 17, 128), (128, 60, 16, 64),
                                   (64, 40, 14, 64)]:
        R, S = case(n, rank, decay, 5)
        out = ops.svd_trunc(R, dmax, 1e-16)
        U, Sg, Vt, keep, disc, info = out
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            ops.svd_trunc(R, dmax, 1e-16)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / reps
        ou = (U.t() @ U - torch.eye(keep, dtype=torch.float64, device='cuda')).abs().max().item()
        ov = (Vt @ Vt.t() - torch.eye(keep, dtype=torch.float64, device='cuda')).abs().max().item()
        ref = torch.linalg.svdvals(R)[:keep]
        ds = ((Sg - ref).abs().max() / ref[0]).item()
        rec = ((U * Sg[None, :]) @ Vt - R).norm().item() / R.norm().item()
        print('svd_trunc %4d x %4d  keep %3d  sweeps %2d  %.3f ms   |U^T U - 1| %.1e  |V V^T - 1| %.1e  dS/S0 %.1e  residual %.1e (discarded %.1e)'
              % (n, n, keep, info['sweeps'], ms, ou, ov, ds, rec, disc), flush=True)


if __name__ == '__main__':
    main()
