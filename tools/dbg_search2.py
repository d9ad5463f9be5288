import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import golden_inputs as gi
from oracle import solver_ref as sr, mps_ref as mr
import tnac4o_amd
for (ins, rot, chi) in [(2, 0, 32), (1, 0, 32), (2, 0, 8)]:
    J = gi.droplet_J(128, ins)
    a = tnac4o_amd.tnac4o(mode='Ising', Nx=4, Ny=4, Nc=8, J=J, beta=3.0)
    b = sr.RefSolver(mode='Ising', Nx=4, Ny=4, Nc=8, J=J, beta=3.0)
    ta, tb = [], []
    a.search_ground_state(M=1024, relative_P_cutoff=1e-8, Dmax=chi, trace=ta)
    b.search_ground_state(M=1024, relative_P_cutoff=1e-8, Dmax=chi, trace=tb)
    print('case', ins, rot, chi, 'logP', a.probability[0], b.probability[0], a.probability[0]-b.probability[0])
    for k, (x, y) in enumerate(zip(ta, tb)):
        same = x[2].shape == y[2].shape and np.array_equal(x[4], y[4])
        if not same:
            print(' step', k, 'branch sets differ', x[2].shape, y[2].shape); break
        big = y[2] > 1e-12
        rel = np.abs(x[2][big]/y[2][big]-1).max()
        print(' step %2d (%d,%d) nb %4d  max rel dP (P>1e-12) %.2e   max abs dP %.2e  min %.2e %.2e' % (k, x[0], x[1], x[2].shape[0], rel, np.abs(x[2]-y[2]).max(), x[3].min(), y[3].min()))
