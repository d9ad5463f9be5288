"""The beam search of one rotation of the bench instance (sweep done beforehand): the library walk (tn_beam_search) against the
torch driver of tnac4o_amd/beam.py, and a host-side profile of the former (what is left is the per-cell table building)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tnac4o_amd
from tnac4o_amd import beam
from tnac4o_amd.auxx import synthetic_chimera

n = 16
s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=synthetic_chimera(n, n, 20260004), beta=3.0)
s._setup_rhoT(graduate_truncation=True, Dmax=64, tolS=1e-16, tolV=1e-10, max_sweeps=20)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    beam.search_native(s, 1024, 1e-8, 1e-12)
    torch.cuda.synchronize()
    print('search_native (tn_beam_search): %.1f ms (E = %.6f)' % (1e3 * (time.perf_counter() - t0), s.energy[0]), flush=True)
for rep in range(2):
    t0 = time.perf_counter()
    beam.search_device(s, 1024, 1e-8, 1e-12)
    torch.cuda.synchronize()
    print('search_device: %.1f ms (E = %.6f)' % (1e3 * (time.perf_counter() - t0), s.energy[0]), flush=True)
pr = cProfile.Profile()
pr.enable()
beam.search_native(s, 1024, 1e-8, 1e-12)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
