#!/usr/bin/env python3
"""Headline benchmark: boundary-MPS PEPS contraction, ms per sweep, chimera L=2048, chi=64 (BASELINE.json).

One step = one sweep = one `_setup_rhoT` (reference tnac4o.py:1674-1695): for each of the 16 rows, build the row
MPO, absorb it into the boundary MPS and compress back to chi=64 with the defaults of search_ground_state
(graduate_truncation, tolS=1e-16, tolV=1e-10, max_sweeps=20).  Synthetic chimera couplings (seed 20260004,
SURVEY.md §8d).  With N GPUs each rank sweeps its own lattice rotation (rank mod 4) of the same couplings — the
reference's 4-rotation loop (examples/e06:97-109) sharded with no data-path collective ("weak" scaling);
`value` = wall ms divided by the number of sweeps all ranks completed.  On each GPU the G = 4 lattice rotations of one
instance run interleaved on 4 HIP streams (the chains are latency-bound), so one step = G sweeps
per rank; the latency of a single chain is reported as config.single_chain_sweep_latency_ms.

Prints ONE JSON line on rank 0.  `roofline` describes the kernel family with the largest summed duration, timed
with HIP events on its launch stream inside the timed region; `cpu_baseline` times the CPU oracle on a bounded
sample (the bottom rows of the same sweep) on this box's host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np   # noqa: E402
import torch         # noqa: E402

FAMILIES = ['gemm_kernel<128,128>', 'gemm_kernel<128,32>', 'gemm_kernel<32,128>', 'gemm_kernel<64,64>',
            'splitk_reduce_kernel', 'absorb_kernel', 'gram_partial_kernel', 'eig_small_kernel',
            'rows_times_small_kernel', 'small_t_times_vecs_kernel', 'tsqr_factor/apply_kernel',
            'lu_reconstruct_kernel', 'qr_aux (diag_qr, assemble_R, init_Q, norms, copies)',
            'svd_aux (norms, init, gather)', 'misc (nfactor, scaling, builders)']
COUNTERS = {'qr_nominal': 15, 'svd_nominal': 16, 'svd_stream': 17, 'svdvals_nominal': 18}   # counter-only families
PHASES = ['gemm_var (attach / projector / environment GEMMs, scaling)', 'absorb', 'qr', 'svd_trunc', 'svdvals', 'mpo_build']
MFMA_FAM = {0, 1, 2, 3}
PEAK_F64_MFMA_TFLOPS = 78.6      # MI355X fp64 matrix peak (vendor figure quoted in SURVEY.md §7; not in the microarch guide)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def _get(lib, phase, f):
    calls, ms, fl, by = C.c_uint64(0), C.c_double(0), C.c_double(0), C.c_double(0)
    lib.tn_profile_get_phase(phase, f, C.byref(calls), C.byref(ms), C.byref(fl), C.byref(by))
    return dict(calls=int(calls.value), ms=ms.value, flops=fl.value, bytes=by.value)


def profile_totals(lib, phase=-1):
    return [dict(kernel=FAMILIES[f], **_get(lib, phase, f)) for f in range(len(FAMILIES))]


def phase_report(lib):
    """Sub-timers of SURVEY.md §8(d): summed kernel durations of one single-chain sweep split by the entry point that
    issued the launch, plus the three SVD-step roofline figures and the nominal QR rate."""
    rep = {}
    for ph, name in enumerate(PHASES):
        t = profile_totals(lib, ph)
        rep[name] = {'kernel_ms': round(sum(x['ms'] for x in t), 3), 'launches': sum(x['calls'] for x in t)}
    qr, svd, stream, sv = (_get(lib, -1, COUNTERS[k]) for k in ('qr_nominal', 'svd_nominal', 'svd_stream', 'svdvals_nominal'))
    t_qr = rep['qr']['kernel_ms'] * 1e-3
    t_svd = rep['svd_trunc']['kernel_ms'] * 1e-3
    if t_qr > 0:
        rep['qr'].update({'calls': qr['calls'], 'nominal_tflops': qr['flops'] / t_qr / 1e12,
                          'frac_of_f64_mfma_peak': qr['flops'] / t_qr / 1e12 / PEAK_F64_MFMA_TFLOPS,
                          'compulsory_GBps': qr['bytes'] / t_qr / 1e9})
    if t_svd > 0:
        rep['svd_trunc'].update({
            'calls': svd['calls'], 'executed_sweeps': stream['calls'],
            'nominal_tflops': svd['flops'] / t_svd / 1e12,
            'hbm_primary_compulsory_GBps': svd['bytes'] / t_svd / 1e9,
            'hbm_primary_frac': svd['bytes'] / t_svd / 1e9 / PEAK_HBM_GBS,
            'hbm_secondary_jacobi_streaming_model_GBps': stream['bytes'] / t_svd / 1e9,
            'hbm_secondary_frac': stream['bytes'] / t_svd / 1e9 / PEAK_HBM_GBS,
            'hbm_tertiary_pmc': None,
            'note': 'block Jacobi (32-wide blocks): the pair Gram/apply are MFMA GEMMs on L2/Infinity-Cache-resident data, '
                    'so the streaming model of a column-pair Jacobi overstates the bytes actually moved; the step is '
                    'latency-bound (eig_small), not HBM-bound'})
    if sv['calls']:
        rep['svdvals'].update({'calls': sv['calls']})
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--L', type=int, default=2048, choices=[128, 512, 2048])
    ap.add_argument('--chi', type=int, default=64)
    ap.add_argument('--cpu-rows', type=int, default=2, help='bottom rows timed on the CPU oracle (0 disables)')
    ap.add_argument('--concurrent', type=int, default=4,
                    help='independent sweeps (lattice rotations of one instance) interleaved per GPU, one stream each')
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--sample', type=int, default=8,
                    help='in the timed region bracket every n-th launch of the dominant kernel family with events')
    ap.add_argument('--no-search', action='store_true', help='skip the (untimed) full search_ground_state figure')
    ap.add_argument('--force-dist', action='store_true',
                    help='initialise the RCCL process group even with one rank (rehearses the multi-GPU code path on one GPU)')
    args = ap.parse_args()

    # stdout carries exactly one JSON line (rank 0): everything the libraries print while we work (RCCL's version banner
    # goes to stdout) is routed to stderr, and the real stdout is restored just before the line is written
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        for k, v in (('RANK', '0'), ('WORLD_SIZE', '1'), ('MASTER_PORT', '29533')):
            os.environ.setdefault(k, v)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))

    import tnac4o_amd
    from tnac4o_amd import _lib
    from tnac4o_amd.auxx import synthetic_chimera
    lib = _lib.lib()
    n = {128: 4, 512: 8, 2048: 16}[args.L]
    seed = {128: 20260002, 512: 20260003, 2048: 20260004}[args.L]
    from tnac4o_amd.parallel import run_concurrent
    G = max(1, args.concurrent)
    J = synthetic_chimera(n, n, seed + rank)                 # every rank sweeps its own instance (weak scaling)
    kw = dict(graduate_truncation=True, Dmax=args.chi, tolS=1e-16, tolV=1e-10, max_sweeps=20)

    def make(rot):
        s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=J, beta=3.0)
        if rot % 4:
            s.rotate_graph(rot % 4)
        return s

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    solvers = [make(g) for g in range(G)]
    solver = solvers[0]

    def step():                  # one step = G sweeps (the rotations of this rank's instance), interleaved on G streams
        run_concurrent([(lambda s=s: s._setup_rhoT(**kw)) for s in solvers])

    # Untimed phase.  (1) one single-chain sweep with events on every kernel family: sweep latency and the per-family
    # table; (2) the W warm-up steps.  The timed steps then bracket only the dominant family's launches with events, so
    # the roofline duration is measured inside the timed region at small overhead.
    warm_prof, single_ms = None, None
    mask_all = (1 << len(FAMILIES)) - 1
    phases = None
    if not args.no_profile and args.warmup > 0:
        lib.tn_profile_reset()
        lib.tn_profile_enable(mask_all)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver._setup_rhoT(**kw)
        torch.cuda.synchronize()
        single_ms = 1e3 * (time.perf_counter() - t0)
        warm_prof = profile_totals(lib)
        phases = phase_report(lib)
        lib.tn_profile_enable(0)
    for _ in range(args.warmup):
        step()
    if not args.no_profile:
        dom_mask = mask_all if warm_prof is None else 1 << max(range(len(warm_prof)), key=lambda i: warm_prof[i]['ms'])
        lib.tn_profile_reset()
        lib.tn_profile_sample(max(1, args.sample))
        lib.tn_profile_enable(dom_mask)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    prof = None
    if not args.no_profile:
        prof = profile_totals(lib)
        lib.tn_profile_enable(0)
        lib.tn_profile_sample(1)
    # second headline, outside the timed region: one full search_ground_state (sweep + 256-site beam,
    # M = 1024) of rotation 0 on a single chain
    search_ms = None
    if rank == 0 and not args.no_search:
        sv = make(0)
        torch.cuda.synchronize()
        t0s = time.perf_counter()
        sv.search_ground_state(M=1024, relative_P_cutoff=1e-8, Dmax=args.chi)
        torch.cuda.synchronize()
        search_ms = 1e3 * (time.perf_counter() - t0s)
        search_info = {'ms': search_ms, 'energy': float(sv.energy[0]), 'degeneracy': int(sv.degeneracy),
                       'log2_probability': float(sv.probability[0]), 'negative_probability': float(sv.negative_probability)}
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = 1e3 * dt / args.steps

    if rank == 0:
        out = {
            'metric': 'PEPS-contraction ms/sweep, chimera L=%d chi=%d (boundary-MPS sweep _setup_rhoT)' % (args.L, args.chi),
            'value': ms_per_step / (world * G), 'unit': 'ms/sweep', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': False, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'chimera L=%d (Nx=Ny=%d, Nc=8) synthetic couplings seed %d, beta=3, chi=%d, one sweep = %d '
                                   'rows of MPO absorb + compress_mps' % (args.L, n, seed, args.chi, n),
                       'sweeps_per_step_all_ranks': world * G,
                       'parallelism': '%d rank(s) x %d lattice rotations interleaved per GPU (one HIP stream each), one instance per '
                                      'rank, no data-path collective' % (world, G),
                       'single_chain_sweep_latency_ms': single_ms,
                       'rhoT_discarded_max': float(max(solver.rhoT_discarded)),
                       'rhoT_overlap_min': float(min(solver.rhoT_overlap)),
                       'bond_dims_mid_row': [int(d) for d in solver.rhoT[n // 2].D]},
        }
        if prof is not None:
            dom = max(range(len(prof)), key=lambda i: prof[i]['ms'])
            d = prof[dom]
            avg_ms = d['ms'] / max(1, d['calls'])
            if dom in MFMA_FAM:
                ach = d['flops'] / (d['ms'] * 1e-3) / 1e12 if d['ms'] > 0 else 0.0
                roof = {'bound': 'mfma', 'achieved': ach, 'peak': PEAK_F64_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': ach / PEAK_F64_MFMA_TFLOPS}
            else:
                ach = d['bytes'] / (d['ms'] * 1e-3) / 1e9 if d['ms'] > 0 else 0.0
                roof = {'bound': 'hbm', 'achieved': ach, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': ach / PEAK_HBM_GBS}
            roof.update({'traffic': None, 'kernel': d['kernel'], 'launches_timed': d['calls'],
                         'launch_sampling': 'every %d-th launch of this family bracketed by HIP events on its stream, inside '
                                            'the timed region' % max(1, args.sample),
                         'avg_launch_ms': avg_ms,
                         'algorithmic_flops_per_launch': d['flops'] / max(1, d['calls']),
                         'algorithmic_bytes_per_launch': d['bytes'] / max(1, d['calls']),
                         'note': 'single-workgroup LDS-resident Jacobi step: latency-bound, far from either roofline'
                                 if d['kernel'].startswith('eig_small') or d['kernel'].startswith('tsqr') else ''})
            # HBM traffic from the PMC counters: rocprofv3's counter mode crashes on this multi-threaded bench, so the
            # FETCH_SIZE / WRITE_SIZE passes are taken on tools/pmc_probe.py (the same kernels at the bulk shapes of this
            # workload) and committed under profiles/; the figure is per launch of the probe, next to the probe's own
            # algorithmic bytes per launch.
            try:
                pm = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')))['families'].get(d['kernel'])
            except (OSError, ValueError, KeyError):
                pm = None
            if pm:
                roof['traffic'] = pm['traffic_bytes_per_launch']
                roof['traffic_detail'] = {
                    'unit': 'bytes per launch', 'source': 'profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, '
                    'separate passes, FETCH_SIZE x2 per the gfx950 correction) on tools/pmc_probe.py: tn_qr 16384 x 1024',
                    'fetch': pm['fetch_bytes_per_launch'], 'write': pm['write_bytes_per_launch'],
                    'algorithmic_bytes_per_launch_same_probe': pm.get('algorithmic_bytes_per_launch'),
                    'traffic_over_algorithmic': pm.get('traffic_over_algorithmic')}
            out['roofline'] = roof
            table = warm_prof if warm_prof is not None else prof
            nsw = 1 if warm_prof is not None else args.steps * G
            out['kernel_table_source'] = ('one single-chain sweep before the timed region (events on all families)'
                                          if warm_prof is not None else 'timed sweeps')
            out['kernel_time_ms_per_sweep'] = {p['kernel']: round(p['ms'] / nsw, 3) for p in table}
            out['kernel_launches_per_sweep'] = {p['kernel']: p['calls'] // nsw for p in table}
            if phases is not None:
                out['phase_kernel_time_single_chain_sweep'] = phases
            gm = [p for i, p in enumerate(table) if i in MFMA_FAM and p['ms'] > 0]
            if gm:
                fl, ms = sum(p['flops'] for p in gm), sum(p['ms'] for p in gm)
                out['gemm_mfma'] = {'achieved': fl / (ms * 1e-3) / 1e12, 'peak': PEAK_F64_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                    'frac': fl / (ms * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS, 'ms_per_sweep': ms / nsw}
            ab = table[5]
            if ab['ms'] > 0:
                out['absorb_hbm'] = {'achieved': ab['bytes'] / (ab['ms'] * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                     'frac': ab['bytes'] / (ab['ms'] * 1e-3) / 1e9 / PEAK_HBM_GBS}
        if search_ms is not None:
            out['full_search_single_chain'] = search_info
        if args.cpu_rows > 0:
            out['cpu_baseline'] = cpu_baseline(J, n, args, solver, kw)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(J, n, args, solver, kw):
    """CPU oracle ("port" of the reference algorithm, numpy/scipy on OpenBLAS) on the bottom `cpu_rows` rows of the
    same sweep, and the GPU time for the same rows; the full-sweep CPU figure is the measured GPU ms/sweep scaled by
    that ratio (a full CPU sweep at L=2048 chi=64 takes tens of minutes)."""
    from oracle import solver_ref as sr
    from oracle import mps_ref as mr
    rows = min(args.cpu_rows, n)
    b = sr.RefSolver(mode='Ising', Nx=n, Ny=n, Nc=8, J=J, beta=3.0)
    t0 = time.perf_counter()
    psi = mr.RefMPS(d=1, L=n, Dmax=1)
    for ny in range(n - 1, n - 1 - rows, -1):
        psi = psi.copy()
        psi.apply_mpo(b._row_mpo(ny), Hconj=True)
        psi.compress_mps(Dmax=kw['Dmax'], tolS=kw['tolS'], tolV=kw['tolV'], max_sweeps=kw['max_sweeps'],
                         graduate_truncation=True)
    cpu_ms = 1e3 * (time.perf_counter() - t0)
    # the same rows on the GPU
    from tnac4o_amd import mps
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    phi = mps.MPS(d=1, L=n, Dmax=1, initial='X')
    for ny in range(n - 1, n - 1 - rows, -1):
        phi = phi.copy()
        phi.apply_mpo(solver._row_mpo(ny), Hconj=True)
        phi.compress_mps(Dmax=kw['Dmax'], tolS=kw['tolS'], tolV=kw['tolV'], max_sweeps=kw['max_sweeps'],
                         graduate_truncation=True)
    torch.cuda.synchronize()
    gpu_ms = 1e3 * (time.perf_counter() - t0)
    try:
        import threadpoolctl
        threads = max([p['num_threads'] for p in threadpoolctl.threadpool_info()] or [os.cpu_count()])
    except Exception:
        threads = os.cpu_count()
    bulk = bulk_site_sample(solver, n, kw, mr, mps)
    return {'value': cpu_ms, 'unit': 'ms for the sample', 'cores': int(threads), 'kind': 'port', 'bulk_site': bulk,
            'sample': 'bottom %d of %d rows of the same sweep (rows %d..%d): MPO absorb + compress_mps, oracle/ numpy+scipy; '
                      'GPU time for the same rows: %.1f ms' % (rows, n, n - 1, n - rows, gpu_ms),
            'gpu_same_sample_ms': gpu_ms, 'speedup_on_sample': cpu_ms / gpu_ms if gpu_ms > 0 else None}


def bulk_site_sample(solver, n, kw, mr, mps):
    """Second bounded CPU sample, representative of the bulk of the sweep (the edge rows above are cheap): the
    right-canonicalisation of ONE absorbed bulk site of the middle row — QR of the (p Dr b) x (Dl b) matrix with the
    oracle's scipy/LAPACK call (mps.py:787-800) vs tn_qr on the same tensor."""
    from tnac4o_amd import ops
    ny, nx = n // 2, n // 2
    psi = solver.rhoT[ny + 1].copy()
    psi.apply_mpo(solver._row_mpo(ny), Hconj=True)
    T = psi.A[nx]
    Dl, p, Dr = T.shape
    host = T.cpu().numpy()
    t0 = time.perf_counter()
    Q, C = mr.qr_pos(host.reshape(Dl, p * Dr).T)
    C = C / mr.pow2_floor_max(C)
    cpu_ms = 1e3 * (time.perf_counter() - t0)
    k = min(p * Dr, Dl)
    Qt = torch.empty((k, p * Dr), dtype=torch.float64, device='cuda')
    Ct = torch.empty((Dl, k), dtype=torch.float64, device='cuda')
    reps = 3
    work = [T.clone() for _ in range(reps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for w in work:
        ops.qr_into(w.view(Dl, p * Dr).t(), Qt.t(), Ct.t(), overwrite=True)
        ops.normalize_pow2_(Ct)
    torch.cuda.synchronize()
    gpu_ms = 1e3 * (time.perf_counter() - t0) / reps
    return {'what': 'QR (+ nfactor) of one absorbed bulk site, %d x %d, row %d site %d' % (p * Dr, Dl, ny, nx),
            'cpu_ms': cpu_ms, 'gpu_ms': gpu_ms, 'speedup': cpu_ms / gpu_ms if gpu_ms > 0 else None,
            'gpu_tflops_nominal': (4.0 * p * Dr * Dl * Dl - 4.0 / 3.0 * Dl ** 3) / (gpu_ms * 1e-3) / 1e12}


if __name__ == '__main__':
    main()
