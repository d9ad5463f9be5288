import os, sys, time
sys.path.insert(0, '/root/repo')
os.environ['TN_GEMM_TRACE'] = '1'
import torch
import tnac4o_amd
from tnac4o_amd.auxx import synthetic_chimera
J = synthetic_chimera(16, 16, 20260004)
s = tnac4o_amd.tnac4o(mode='Ising', Nx=16, Ny=16, Nc=8, J=J, beta=3)
kw = dict(graduate_truncation=True, Dmax=64, tolS=1e-16, tolV=1e-10, max_sweeps=20)
s._setup_rhoT(**kw)
torch.cuda.synchronize()
