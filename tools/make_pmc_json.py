#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc summaries of tools/collect_profiles.sh (gpurun_out/prof_r02/) into profiles/r02_pmc_traffic.json:
HBM-side bytes per launch of our kernels (FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md, KB -> bytes),
the whole-call traffic of tn_qr 16384 x 1024 and of tn_svd_trunc 1024 x 1024 against their compulsory bytes, and the raw SQ
MFMA counters.  Usage: make_pmc_json.py [PROBE_TIMES json string]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get('ROUND', 'r05')
BASE = os.path.join(ROOT, 'gpurun_out', 'prof_' + ROUND)


SEG = {}                # tag -> {segment: {kernel: {counter: total}}}


def load(tag):
    """Per kernel over all segments: counter -> (dispatches, total, mean); the per-segment totals go to SEG[tag]."""
    out, seg = {}, {}
    with open(os.path.join(BASE, 'pmc_%s_summary.csv' % tag)) as f:
        for r in csv.DictReader(f):
            d = out.setdefault(r['Kernel'], {})
            n0, t0, _ = d.get(r['Counter'], (0, 0.0, 0.0))
            n, t = n0 + int(r['Dispatches']), t0 + float(r['Total'])
            d[r['Counter']] = (n, t, t / n)
            seg.setdefault(int(r.get('Segment', 0) or 0), {}).setdefault(r['Kernel'], {})[r['Counter']] = float(r['Total'])
    SEG[tag] = seg
    return out


def segment_bytes(segment):
    """HBM bytes of every launch of a segment of tools/pmc_probe.py (FETCH_SIZE x 2 + WRITE_SIZE, KB -> bytes), ours or not."""
    f = sum(v.get('FETCH_SIZE', 0.0) for v in SEG['FETCH_SIZE'].get(segment, {}).values())
    w = sum(v.get('WRITE_SIZE', 0.0) for v in SEG['WRITE_SIZE'].get(segment, {}).values())
    return 2.0 * 1024 * f + 1024 * w


F, W = load('FETCH_SIZE'), load('WRITE_SIZE')
M1 = load('SQ_INSTS_VALU_MFMA_F64+SQ_VALU_MFMA_BUSY_CYCLES+SQ_BUSY_CYCLES')
M2 = load('SQ_INSTS_VALU_MFMA_MOPS_F64+SQ_WAVE_CYCLES+SQ_WAIT_INST_LDS')
times = json.loads(sys.argv[1]) if len(sys.argv) > 1 else {}


def bytes_of(k):
    f = F.get(k, {}).get('FETCH_SIZE', (0, 0, 0))
    w = W.get(k, {}).get('WRITE_SIZE', (0, 0, 0))
    return f[0], 2.0 * 1024 * f[1], 1024 * w[1]        # dispatches, fetch bytes (x2), write bytes (totals over the probe)


kern = {}
for k in F:
    if 'tn::' not in k:
        continue
    n, fb, wb = bytes_of(k)
    kern[k] = {'dispatches': n, 'fetch_bytes_per_launch': fb / n, 'write_bytes_per_launch': wb / n,
               'traffic_bytes_per_launch': (fb + wb) / n}

fam = {}
ts = [k for k in kern if 'cq_gram_kernel' in k or 'cq_pass_kernel' in k or 'cq_post_kernel' in k or 'cq_fused_kernel' in k]
n = sum(kern[k]['dispatches'] for k in ts)
fb = sum(kern[k]['fetch_bytes_per_launch'] * kern[k]['dispatches'] for k in ts)
wb = sum(kern[k]['write_bytes_per_launch'] * kern[k]['dispatches'] for k in ts)
# algorithmic bytes of the panel chain over one 16384 x 1024 QR: per panel the Gram launch reads the panel (8 B per element), a
# substitution pass reads and writes it (16 B), the post launch reads it and writes Y and W (24 B); two passes per panel assumed
# (the six-launch chain of the 16384 x 1024 QR); the single-launch form of the 4096 x 512 QR moves the panel in and Y, W out (24 B)
alg = 0.0
six_launch = any('cq_gram_kernel' in k for k in kern)          # (a lone stream is always admitted to the single-launch form, cq_big_admit)
for p in range(32):
    rows = 16384 - 32 * p
    alg += ((8.0 + 2 * 16.0 + 24.0) if six_launch else 24.0) * rows * 32
for p in range(16):
    rows = 4096 - 32 * p
    alg += 24.0 * rows * 32
fam['panel step (cq_fused_kernel, sq_kernel)'] = {'probe_shape': 'tn_qr 16384 x 1024 (32 panels, %s) + tn_qr 4096 x 512 (16 panels, single-launch form)' % ('six-launch chain' if six_launch else 'single-launch form: 64 workgroups'), 'dispatches': n, 'fetch_bytes_per_launch': fb / n, 'write_bytes_per_launch': wb / n,
                                   'traffic_bytes_per_launch': (fb + wb) / n, 'algorithmic_bytes_per_launch': alg / n,
                                   'traffic_over_algorithmic': (fb + wb) / alg}
for name, pat in (('Jacobi rounds (svdl_kernel)', 'svdl_kernel'), ('eig_small3_kernel (rounds as separate launches)', 'eig_small'), ('absorb_kernel', 'absorb_mfma_kernel'),
                  ('sq_kernel (one-launch factorisation)', 'sq_kernel')):
    ks = [k for k in kern if pat in k]
    if ks:
        k = ks[0]
        fam[name] = dict(kern[k])
        if pat == 'svdl_kernel':
            fam[name]['probe_shape'] = 'tn_svd_trunc 192 x 900 (tools/pmc_probe.py): all Jacobi rounds of the call in one launch of svdl_kernel'
            fam[name]['algorithmic_bytes_per_launch'] = 16.0 * 192 * (900 + 192)
            fam[name]['traffic_over_algorithmic'] = fam[name]['traffic_bytes_per_launch'] / fam[name]['algorithmic_bytes_per_launch']

# whole-call traffic: everything launched between the probe's markers (segment 1 = tn_qr 16384 x 1024, 4 = tn_svd_trunc)
qr_bytes = segment_bytes(1)
m, nn = 16384, 1024
qr_comp = 8.0 * (2 * m * nn + nn * nn)
whole = {'tn_qr_4096x512': {'hbm_bytes': segment_bytes(2), 'compulsory_bytes': 8.0 * (2 * 4096 * 512 + 512 * 512),
                            'traffic_over_compulsory': segment_bytes(2) / (8.0 * (2 * 4096 * 512 + 512 * 512))},
         'tn_qr_16384x1024': {'hbm_bytes': qr_bytes, 'compulsory_bytes': qr_comp, 'traffic_over_compulsory': qr_bytes / qr_comp,
                              'ms_unprofiled': times.get('qr_16384x1024_ms')}}
svd_bytes = segment_bytes(4)
svd_comp = 8.0 * (2 * 320 * 1024 + 320 * 320 + 320)
svd = {'hbm_bytes': svd_bytes, 'compulsory_bytes': svd_comp, 'traffic_over_compulsory': svd_bytes / svd_comp,
       'ms_unprofiled': times.get('svd_trunc_320x1024_ms'), 'sweeps': times.get('svd_sweeps')}
if times.get('svd_trunc_320x1024_ms'):
    svd['GBps'] = svd_bytes / (times['svd_trunc_320x1024_ms'] * 1e-3) / 1e9
    svd['frac_of_hbm_peak'] = svd['GBps'] / 8000.0
whole['tn_svd_trunc_320x1024'] = svd
# sixteen one-launch factorisations 1024 x 64 (segment 6 of the probe): in 8 B, Q out 8 B per element + R
sq_bytes = segment_bytes(6) / 16.0
sq_alg = 8.0 * (2 * 1024 * 64 + 64 * 64)
whole['tn_qr_1024x64_one_launch'] = {'hbm_bytes': sq_bytes, 'compulsory_bytes': sq_alg, 'traffic_over_compulsory': sq_bytes / sq_alg if sq_alg else None,
                                     'us_unprofiled': times.get('qr_1024x64_one_launch_us')}

# the 192 x 900 truncated SVD whose rounds run in one launch (segment 8): in 8 B per element, U / S / V^T out
s2_bytes = segment_bytes(8)
s2_comp = 8.0 * (192 * 900 + 192 * 64 + 64 + 64 * 900)
whole['tn_svd_trunc_192x900_one_launch'] = {'hbm_bytes': s2_bytes, 'compulsory_bytes': s2_comp, 'traffic_over_compulsory': s2_bytes / s2_comp,
                                            'ms_unprofiled': times.get('svd_trunc_192x900_ms'), 'sweeps': times.get('svd_192x900_sweeps'),
                                            'keep': times.get('svd_192x900_keep')}

mfma = {}
for k in kern:
    a, b = M1.get(k, {}), M2.get(k, {})
    if a.get('SQ_INSTS_VALU_MFMA_F64', (0, 0, 0))[1] > 0:
        mfma[k] = {'SQ_INSTS_VALU_MFMA_F64_per_launch': a['SQ_INSTS_VALU_MFMA_F64'][2],
                   'SQ_VALU_MFMA_BUSY_CYCLES_per_launch': a['SQ_VALU_MFMA_BUSY_CYCLES'][2],
                   'SQ_BUSY_CYCLES_per_launch': a['SQ_BUSY_CYCLES'][2],
                   'SQ_INSTS_VALU_MFMA_MOPS_F64_per_launch': b.get('SQ_INSTS_VALU_MFMA_MOPS_F64', (0, 0, 0))[2],
                   'SQ_WAVE_CYCLES_per_launch': b.get('SQ_WAVE_CYCLES', (0, 0, 0))[2],
                   'SQ_WAIT_INST_LDS_per_launch': b.get('SQ_WAIT_INST_LDS', (0, 0, 0))[2],
                   'mfma_busy_over_wave_cycles': a['SQ_VALU_MFMA_BUSY_CYCLES'][2] / max(1.0, b.get('SQ_WAVE_CYCLES', (0, 0, 1))[2])}
out = {'source': 'rocprofv3 --pmc (separate passes: FETCH_SIZE; WRITE_SIZE; SQ MFMA counters) over tools/pmc_probe.py on MI355X, '
                 '%s; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md); KB -> bytes' % ROUND,
       'shapes': 'tn_qr 16384 x 1024 (nb=32, 32 panels), tn_qr 4096 x 512 (16 panels), both in the single-launch form of the Cholesky-QR panel step when admitted, tn_svd_trunc 320 x 1024 (leading rows of the triangular factor of a graded rank-300 matrix), tn_absorb bulk '
                 'site, tn_gemm 16384 x 1024 x 1024',
       'families': fam, 'kernels': kern, 'whole_call': whole, 'svd_step': svd,
       'mfma_counters': {'file': 'profiles/%s_pmc_traffic.json (mfma_counters.kernels)' % ROUND, 'kernels': mfma,
                         'note': 'raw SQ counters per launch; the derived MfmaUtil of rocprofv3 falls back to gfx94x formulas on '
                                 'gfx950, so only ratios of raw counters are quoted'}}
json.dump(out, open(os.path.join(ROOT, 'profiles', '%s_pmc_traffic.json' % ROUND), 'w'), indent=1)
print(json.dumps({'families': fam, 'whole_call': whole}, indent=1))
