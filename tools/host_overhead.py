"""Where does the host time of a chain go?  Every libtnpeps call is wrapped with a timer (per thread); one sweep is run as a
single chain and then 4 chains are interleaved.  Prints per chain: wall time, time inside the C library (GIL released),
number of library calls, and the remainder (Python between calls + waiting for the GIL)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tnac4o_amd
from tnac4o_amd import _lib, parallel
from tnac4o_amd.auxx import synthetic_chimera

L = _lib.lib()
acc = {}


def wrap(fn):
    def inner(*a):
        t0 = time.perf_counter()
        r = fn(*a)
        dt = time.perf_counter() - t0
        rec = acc.setdefault(threading.get_ident(), [0.0, 0])
        rec[0] += dt
        rec[1] += 1
        return r
    return inner


for name in _lib.SIGNATURES:
    if name.startswith('tn_profile') or name in ('tn_last_error', 'tn_version', 'tn_build_id'):
        continue
    setattr(L, name, wrap(getattr(L, name)))

# second bucket: torch calls made between library calls (allocation, stream queries, small tensor ops)
tacc = {}


def twrap(fn, key):
    def inner(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        rec = tacc.setdefault((threading.get_ident(), key), [0.0, 0])
        rec[0] += time.perf_counter() - t0
        rec[1] += 1
        return r
    return inner


for mod, name in ((torch, 'empty'), (torch, 'ones'), (torch, 'ones_like'), (torch, 'diag'), (torch, 'zeros'), (torch.cuda, 'current_stream'),
                  (torch.cuda, 'current_device')):
    setattr(mod, name, twrap(getattr(mod, name), name))
for name in ('contiguous', 'clone', 'view', 't', 'reshape', 'item', 'cpu'):
    setattr(torch.Tensor, name, twrap(getattr(torch.Tensor, name), 'Tensor.' + name))

n = 16
J = synthetic_chimera(n, n, 20260004)
kw = dict(graduate_truncation=True, Dmax=64, tolS=1e-16, tolV=1e-10, max_sweeps=20)


def make(rot):
    s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=J, beta=3.0)
    if rot:
        s.rotate_graph(rot)
    return s


def run(s):
    t0 = time.perf_counter()
    s._setup_rhoT(**kw)
    torch.cuda.current_stream().synchronize()
    return (threading.get_ident(), time.perf_counter() - t0)


solvers = [make(r) for r in range(4)]
run(solvers[0])                      # warm-up
acc.clear()
tid, wall = run(solvers[0])
c, k = acc[tid]
print('single chain : wall %.3f s  in library %.3f s  calls %d  python+gil %.3f s (%.1f us per call)' % (wall, c, k, wall - c, 1e6 * (wall - c) / k))
print('      torch: ' + '  '.join('%s %.0f ms/%d' % (key[1], 1e3 * v[0], v[1]) for key, v in sorted(tacc.items(), key=lambda kv: -kv[1][0]) if key[0] == tid)[:400])
tacc.clear()
for trial in range(2):
    acc.clear()
    tacc.clear()
    t0 = time.perf_counter()
    res = parallel.run_concurrent([(lambda s=s: run(s)) for s in solvers])
    tot = time.perf_counter() - t0
    print('4 chains     : step wall %.3f s' % tot)
    for tid, wall in res:
        c, k = acc[tid]
        print('   chain wall %.3f s  in library %.3f s  calls %d  python+gil %.3f s (%.1f us per call)' % (wall, c, k, wall - c, 1e6 * (wall - c) / k))
        print('      torch: ' + '  '.join('%s %.0f ms/%d' % (key[1], 1e3 * v[0], v[1]) for key, v in sorted(tacc.items(), key=lambda kv: -kv[1][0]) if key[0] == tid)[:400])
