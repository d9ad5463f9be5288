import os, sys, cProfile, pstats, io
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, tnac4o_amd
from tnac4o_amd.auxx import synthetic_chimera
n = 16
s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=synthetic_chimera(n, n, 20260004), beta=3.0)
kw = dict(graduate_truncation=True, Dmax=64, tolS=1e-16, tolV=1e-10, max_sweeps=20)
s._setup_rhoT(**dict(kw, Dmax=8)); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
s._setup_rhoT(**kw); torch.cuda.synchronize()
pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats('tottime').print_stats(22); print(st.getvalue()[:5000])
