"""Bottom-up sweep (_setup_rhoB, Hconj=False absorption) of the bench instance, single chain: ms per sweep under the environment given
on the command line (A/B of TN_ATTACH_FUSED and friends).  Usage: python tools/rhob_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tnac4o_amd
from tnac4o_amd.auxx import synthetic_chimera
n = 16
s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=synthetic_chimera(n, n, 20260004), beta=3.0)
kw = dict(graduate_truncation=True, Dmax=64, tolS=1e-16, tolV=1e-10, max_sweeps=20)
for which in ('_setup_rhoB', '_setup_rhoT'):
    f = getattr(s, which)
    f(**kw); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); f(**kw); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    disc = s.rhoB_discarded if which == '_setup_rhoB' else s.rhoT_discarded
    print('%s TN_ATTACH_FUSED=%s ms/sweep %s  max discarded %.6e' % (which, os.environ.get('TN_ATTACH_FUSED', '(default)'), ['%.1f' % t for t in ts], max(float(x) for x in disc)), flush=True)
