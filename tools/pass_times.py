"""Wall time of one single-chain sweep split by the passes of compress_mps (synchronised after each pass; mps.py:175-200):
absorb, pass 1 canonise_right, copy, pass 2 canonise_left(4 chi), variational(4 chi), pass 3 canonise_right(2 chi),
pass 4 canonise_left(chi), final variational."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tnac4o_amd
from tnac4o_amd import mps
from tnac4o_amd.auxx import synthetic_chimera

acc = collections.OrderedDict()
state = {'pass': 0}


def timed(name, fn):
    def inner(self, *a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(self, *a, **k)
        torch.cuda.synchronize()
        key = name
        if name in ('canonise_left', 'canonise_right', 'canonise_right_weighted', 'variational_compress'):
            state['pass'] += 1
            key = '%d %s%s' % (state['pass'], name, ' compress Dmax=%s' % k.get('Dmax') if k.get('compress') else '')
        acc[key] = acc.get(key, 0.0) + time.perf_counter() - t0
        return r
    return inner


orig_compress = mps.MPS.compress_mps


def compress(self, *a, **k):
    state['pass'] = 0
    return orig_compress(self, *a, **k)


mps.MPS.compress_mps = compress
for nm in ('canonise_left', 'canonise_right', 'canonise_right_weighted', 'variational_compress', 'apply_mpo', 'copy'):
    setattr(mps.MPS, nm, timed(nm, getattr(mps.MPS, nm)))
n = 16
s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=synthetic_chimera(n, n, 20260004), beta=3.0)
kw = dict(graduate_truncation=True, Dmax=64, tolS=1e-16, tolV=1e-10, max_sweeps=20)
s._setup_rhoT(**kw)
acc.clear()
torch.cuda.synchronize(); t0 = time.perf_counter()
s._setup_rhoT(**kw)
torch.cuda.synchronize(); tot = time.perf_counter() - t0
print('sweep %.3f s' % tot)
for k, v in acc.items():
    print('  %-48s %7.3f s  %5.1f %%' % (k, v, 100 * v / tot))
