"""Single chain on the default (legacy null) stream vs on a stream of its own."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import tnac4o_amd
from tnac4o_amd.auxx import synthetic_chimera
n = 16
J = synthetic_chimera(n, n, 20260004)
kw = dict(graduate_truncation=True, Dmax=64, tolS=1e-16, tolV=1e-10, max_sweeps=20)
s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=J, beta=3.0)
def timed(label):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s._setup_rhoT(**kw)
    torch.cuda.synchronize(); print(label, round(1e3 * (time.perf_counter() - t0), 1), 'ms', flush=True)
timed('default stream (warm-up)'); timed('default stream'); timed('default stream')
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    timed('own stream (warm-up)'); timed('own stream'); timed('own stream')
