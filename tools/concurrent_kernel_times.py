"""Average event-timed kernel duration per family: one chain alone vs 4 interleaved chains (what inflates under interleaving?)."""
import os, sys, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import tnac4o_amd
from tnac4o_amd import _lib
from tnac4o_amd.auxx import synthetic_chimera
from tnac4o_amd.parallel import run_concurrent
sys.path.insert(0, R)
import bench
lib = _lib.lib()
n = 16
J = synthetic_chimera(n, n, 20260004)
kw = dict(graduate_truncation=True, Dmax=64, tolS=1e-16, tolV=1e-10, max_sweeps=20)
def make(rot):
    s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=J, beta=3.0)
    if rot: s.rotate_graph(rot)
    return s
solvers = [make(g) for g in range(4)]
run_concurrent([(lambda s=s: s._setup_rhoT(**kw)) for s in solvers])      # warm
res = {}
for label, group in (('single', solvers[:1]), ('four', solvers)):
    lib.tn_profile_reset(); lib.tn_profile_sample(4); lib.tn_profile_enable((1 << len(bench.FAMILIES)) - 1)
    run_concurrent([(lambda s=s: s._setup_rhoT(**kw)) for s in group])
    torch.cuda.synchronize()
    res[label] = bench.profile_totals(lib)
    lib.tn_profile_enable(0)
print('%-52s %10s %10s %8s' % ('family', 'single us', 'four us', 'ratio'))
for a, b in zip(res['single'], res['four']):
    if a['calls'] and b['calls']:
        x, y = 1e3 * a['ms'] / a['calls'], 1e3 * b['ms'] / b['calls']
        print('%-52s %10.1f %10.1f %8.2f' % (a['kernel'][:52], x, y, y / x))
