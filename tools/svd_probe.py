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
    # numerically low rank, as on the contraction path: the spectrum reaches 10^-decay at index `rank` and keeps falling
    S = torch.pow(10.0, -decay * torch.arange(n, dtype=torch.float64) / rank).clamp_min(1e-300).cuda()
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
        ref = torch.as_tensor(__import__('numpy').linalg.svd(R.cpu().numpy(), compute_uv=False)[:keep]).cuda()
        ds = ((Sg - ref).abs().max() / ref[0]).item()
        rec = ((U * Sg[None, :]) @ Vt - R).norm().item() / R.norm().item()
        print('svd_trunc %4d x %4d  keep %3d  sweeps %2d  %.3f ms   |U^T U - 1| %.1e  |V V^T - 1| %.1e  dS/S0 %.1e  residual %.1e (discarded %.1e)'
              % (n, n, keep, info['sweeps'], ms, ou, ov, ds, rec, disc), flush=True)
        if os.environ.get('TN_PROBE_FAMILIES'):
            import ctypes as C
            from tnac4o_amd._lib import lib
            L = lib()
            L.tn_profile_reset()
            L.tn_profile_enable((1 << 15) - 1)
            ops.svd_trunc(R, dmax, 1e-16)
            torch.cuda.synchronize()
            L.tn_profile_enable(0)
            names = ['gemm128x128', 'gemm128x32', 'gemm32x128', 'gemm64x64', 'splitk_reduce', 'absorb', 'gram', 'eig_small', 'rows_small',
                     'vecs_small', 'panel', 'lu', 'qr_aux', 'svd_aux', 'misc']
            for f, nm in enumerate(names):
                calls, msf, fl, by = C.c_uint64(0), C.c_double(0), C.c_double(0), C.c_double(0)
                L.tn_profile_get(f, C.byref(calls), C.byref(msf), C.byref(fl), C.byref(by))
                if calls.value:
                    print('      %-14s %6d launches  %8.3f ms  (%.1f us each)' % (nm, calls.value, msf.value, 1e3 * msf.value / calls.value), flush=True)


if __name__ == '__main__':
    main()
