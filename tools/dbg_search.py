import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import golden_inputs as gi
from oracle import solver_ref as sr, mps_ref as mr
import tnac4o_amd
ins, rot, chi = 2, 0, 32
J = gi.droplet_J(128, ins)
a = tnac4o_amd.tnac4o(mode='Ising', Nx=4, Ny=4, Nc=8, J=J, beta=3.0)
b = sr.RefSolver(mode='Ising', Nx=4, Ny=4, Nc=8, J=J, beta=3.0)
kw = dict(graduate_truncation=True, Dmax=chi, tolS=1e-16, tolV=1e-10, max_sweeps=20)
a._setup_rhoT(**kw); b._setup_rhoT(**kw)
print('D gpu', [m.D for m in a.rhoT]); print('D ref', [m.D for m in b.rhoT])
print('disc gpu', a.rhoT_discarded); print('disc ref', b.rhoT_discarded)
print('ovl gpu', a.rhoT_overlap); print('ovl ref', b.rhoT_overlap)
def chain(psi):
    As = [x.detach().cpu().numpy() for x in psi.A]
    o = mr.RefMPS(d=[x.shape[1] for x in As], L=len(As), Dmax=1, canonise=None); o.A = As; return o
for ny in range(5):
    x, y = chain(a.rhoT[ny]), b.rhoT[ny]
    f = abs(mr.mps_dot(x, y))/np.sqrt(mr.mps_dot(x,x)*mr.mps_dot(y,y))
    print('row', ny, '1-fidelity %.3e' % (1-f))
# exact (untruncated) reference for row comparisons: chi=256
c = sr.RefSolver(mode='Ising', Nx=4, Ny=4, Nc=8, J=J, beta=3.0)
c._setup_rhoT(graduate_truncation=True, Dmax=4096, tolS=1e-16, tolV=1e-10, max_sweeps=20)
for ny in range(5):
    x, y, z = chain(a.rhoT[ny]), b.rhoT[ny], c.rhoT[ny]
    fx = abs(mr.mps_dot(x, z))/np.sqrt(mr.mps_dot(x,x)*mr.mps_dot(z,z)); fy = abs(mr.mps_dot(y, z))/np.sqrt(mr.mps_dot(y,y)*mr.mps_dot(z,z))
    print('row', ny, 'vs exact: gpu 1-f %.3e  ref 1-f %.3e' % (1-fx, 1-fy), 'exact D', z.D)
