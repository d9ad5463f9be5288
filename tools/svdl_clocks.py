"""Where the one-launch Jacobi rounds (svdl_kernel) spend their time over one single-chain sweep of the headline workload
(build with TN_EXTRA_HIPCC_FLAGS=-DTN_CLOCKS): phase clocks of chunk workgroup 0, summed over all launches."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tnac4o_amd
from tnac4o_amd import _lib
from tnac4o_amd.auxx import synthetic_chimera
L = _lib.lib()
buf = (C.c_longlong * 8)()
J = synthetic_chimera(16, 16, 20260004)
s = tnac4o_amd.tnac4o(mode='Ising', Nx=16, Ny=16, Nc=8, J=J, beta=3)
kw = dict(graduate_truncation=True, Dmax=64, tolS=1e-16, tolV=1e-10, max_sweeps=20)
s._setup_rhoT(**kw)                       # warm-up
torch.cuda.synchronize()
L.tn_debug_svdl_clocks(buf, 1)
ebuf = (C.c_longlong * 8)()
L.tn_debug_eig_clocks(ebuf, 1)
t0 = time.perf_counter()
s._setup_rhoT(**kw)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3
L.tn_debug_svdl_clocks(buf, 1)
names = ['load', 'gram', 'barrier', 'eig wait', 'rotate', 'store+norms']
print('sweep %.1f ms; svdl launches %d, rounds %d' % (ms, buf[7], buf[6]))
print('  total ms: ' + '  '.join('%s %.1f' % (names[i], buf[i] / 1e5) for i in range(6)))
print('  per round us: ' + '  '.join('%s %.1f' % (names[i], buf[i] / 100.0 / max(1, buf[6])) for i in (1, 2, 3, 4)))
L.tn_debug_eig_clocks(ebuf, 1)
en = ['partial sums', 'measure + fast path', 'cyclic sweeps', 'Newton-Schulz', 'store']
print('eigenproblems of pair 0: %d calls, %d with cyclic sweeps' % (ebuf[5], ebuf[6]))
print('  total ms: ' + '  '.join('%s %.1f' % (en[i], ebuf[i] / 1e5) for i in range(5)))
print('  per call us: ' + '  '.join('%s %.1f' % (en[i], ebuf[i] / 100.0 / max(1, ebuf[5])) for i in range(5)))
