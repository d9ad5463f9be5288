"""Ground-state search of a bundled droplet instance on the GPU, checked against the reference's golden file."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
import golden_inputs as gi
import tnac4o_amd
L = int(sys.argv[1]); chi = int(sys.argv[2]); pre = len(sys.argv) > 3 and sys.argv[3] == 'pre'
n = {128: 4, 512: 8, 2048: 16}[L]
J = gi.droplet_J(L, 1)
s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=J, beta=3.0)
t0 = time.perf_counter()
if pre:
    s.precondition(mode='balancing')
t1 = time.perf_counter()
s.search_ground_state(M=1024, relative_P_cutoff=1e-8, Dmax=chi)
torch.cuda.synchronize(); t2 = time.perf_counter()
E, bits = gi.golden_groundstate(L, 1)
print('L=%d chi=%d pre=%s: E=%.9f golden %.6f  dE=%.2e  bits equal=%s  log2P=%.6f  deg=%d  neg=%.2e  disc=%.3f' % (
    L, chi, pre, s.energy[0], E, s.energy[0] - E, np.array_equal(s.binary_states()[0], bits), s.probability[0], s.degeneracy,
    s.negative_probability, s.discarded_probability))
print('energy_Jij check %.9f' % tnac4o_amd.energy_Jij(J, s.binary_states()[:1])[0], ' times: precondition %.1fs search %.1fs' % (t1 - t0, t2 - t1))
print('rhoT_discarded max %.2e  overlap min %.15f' % (max(s.rhoT_discarded), min(s.rhoT_overlap)))
