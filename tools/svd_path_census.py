"""Which truncated SVDs of a sweep leave the one-launch form (svdl_kernel), and why?  Runs the top-down sweep of the bench instance for one
lattice rotation with TN_SVD_TRACE=1 (the separate-launch loop prints a line per outer sweep with nv, L and the live vectors) and counts
the calls by shape.  Usage: python tools/svd_path_census.py [rotation]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, %(root)r)
import torch, tnac4o_amd
from tnac4o_amd.auxx import synthetic_chimera
s = tnac4o_amd.tnac4o(mode='Ising', Nx=16, Ny=16, Nc=8, J=synthetic_chimera(16, 16, 20260004), beta=3.0)
if %(rot)d:
    s.rotate_graph(%(rot)d)
s._setup_rhoT(graduate_truncation=True, Dmax=64, tolS=1e-16, tolV=1e-10, max_sweeps=20)
torch.cuda.synchronize()
'''
rot = int(sys.argv[1]) if len(sys.argv) > 1 else 2
env = dict(os.environ, TN_SVD_TRACE='1')
out = subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT, rot=rot)], env=env, stderr=subprocess.PIPE, text=True).stderr
calls = collections.Counter()
sweeps = collections.Counter()
for line in out.splitlines():
    m = re.match(r'\[tn_svd\] nv=(\d+) L=(\d+) live=(\d+) sweep=(\d+)', line)
    if m:
        key = (int(m.group(1)), int(m.group(2)), int(m.group(3)))
        sweeps[key] += 1
        if int(m.group(4)) == 1:
            calls[key] += 1
print('rotation %d: truncated SVDs on the separate-launch loop (nv, L, live vectors): calls, outer sweeps' % rot)
for k in sorted(calls, key=lambda k: -sweeps[k])[:40]:
    print('  nv=%4d L=%5d live=%4d : %3d calls %4d sweeps' % (k + (calls[k], sweeps[k])))
print('total', sum(calls.values()), 'calls,', sum(sweeps.values()), 'outer sweeps')
