"""Hard panels of a real sweep as regression inputs: runs the boundary-MPS sweep of one rotation of the bench instance (chimera
L = 2048, chi = 64, seed 20260004) with TN_PANEL_CAPTURE set, keeps the panels of tn_qr that needed >= 4 substitution passes
(<= 4096 rows) and writes the ten smallest to tests/golden/g12_hard_panels.npz together with the pass count each one took.
Usage (GPU box): capture_panels.py [out.npz]"""
import glob
import os
import re
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cap = tempfile.mkdtemp(prefix='tn_panels_')
os.environ['TN_PANEL_CAPTURE'] = cap
os.environ.setdefault('TN_PANEL_CAPTURE_MIN', '4')
import numpy as np
import torch
import tnac4o_amd
from tnac4o_amd import ops
from tnac4o_amd.auxx import synthetic_chimera

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'g12_hard_panels.npz')
n = 16
s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=synthetic_chimera(n, n, 20260004), beta=3.0)
ops.panel_stats(reset=True)
s._setup_rhoT(graduate_truncation=True, Dmax=64, tolS=1e-16, tolV=1e-10, max_sweeps=20)
torch.cuda.synchronize()
st = ops.panel_stats()
print('panels %d, with >= 3 passes %d, with >= 4 passes %d, fallbacks %d' % (st['panels'], st['panels_with_3_or_more_passes'],
                                                                          st['panels_with_4_or_more_passes'], st['householder_fallbacks']))
files = sorted(glob.glob(os.path.join(cap, 'panel_*.f64')))
items = []
for f in files:
    m = re.search(r'panel_(\d+)_(\d+)x(\d+)_p(\d+)\.f64$', f)
    seq, rows, b, passes = (int(x) for x in m.groups())
    items.append((rows * b, seq, rows, b, passes, f))
items.sort()
print('%d panels captured (<= %s rows)' % (len(items), os.environ.get('TN_PANEL_CAPTURE_MAXROWS', '4096')))
keep = {}
for k, (_, seq, rows, b, passes, f) in enumerate(items[:10]):
    X = np.fromfile(f, dtype=np.float64).reshape(rows, b)
    keep['panel%d' % k] = X
    keep['passes%d' % k] = np.array([passes])
    print('  panel%d: %d x %d, %d passes, cond ~ %.2e' % (k, rows, b, passes, np.linalg.cond(X)))
np.savez_compressed(out, **keep)
print('wrote', out, os.path.getsize(out), 'bytes')
