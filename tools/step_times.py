"""Wall time of consecutive benchmark steps (4 interleaved sweeps each): is the rate steady?"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import tnac4o_amd
from tnac4o_amd.auxx import synthetic_chimera
from tnac4o_amd.parallel import run_concurrent
n = 16
J = synthetic_chimera(n, n, 20260004)
kw = dict(graduate_truncation=True, Dmax=64, tolS=1e-16, tolV=1e-10, max_sweeps=20)
def make(rot):
    s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=J, beta=3.0)
    if rot: s.rotate_graph(rot)
    return s
solvers = [make(g) for g in range(4)]
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run_concurrent([(lambda s=s: s._setup_rhoT(**kw)) for s in solvers])
    torch.cuda.synchronize(); print('step', i, round(1e3 * (time.perf_counter() - t0) / 4, 1), 'ms/sweep', flush=True)
