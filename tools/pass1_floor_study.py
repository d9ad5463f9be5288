"""Experiment: effect of the weight floor (PASS1_FLOOR) of the weighted first pass on the bonds it produces, and on the state
(overlap of the right-canonical result with the plain pass's, via the product's own dot)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tnac4o_amd
from tnac4o_amd import ops, mps
from tnac4o_amd.auxx import synthetic_chimera

n = 16
s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=synthetic_chimera(n, n, 20260004), beta=3.0)
kw = dict(graduate_truncation=True, Dmax=64, tolS=1e-16, tolV=1e-10, max_sweeps=20)
s._setup_rhoT(**kw)
for ny in (8, 3):
    psi = s.rhoT[ny + 1].copy()
    psi.apply_mpo(s._row_mpo(ny), Hconj=True)
    ref = psi.copy(); ref.D = list(psi.D)
    ref.canonise_right()
    nr = mps.dot(ref, ref)
    for floor in (1e-14, 1e-18, 1e-22, 1e-26, 0.0):
        mps.PASS1_FLOOR = floor
        phi = psi.copy(); phi.D = list(psi.D)
        ok = phi.canonise_right_weighted()
        ov = mps.dot(ref, phi) / np.sqrt(nr * mps.dot(phi, phi))
        print('row %d floor %.0e: ok %s bound %.2e bonds sum %d max %d  1-|overlap| %.2e' % (ny, floor, ok, phi.reveal_error_bound, sum(phi.D), max(phi.D), 1 - abs(ov)), flush=True)
