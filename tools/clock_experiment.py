"""Does keeping the GPU busy on another stream speed up the latency-bound chain (clock governor effect)?"""
import os, sys, time, threading
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, tnac4o_amd
from tnac4o_amd.auxx import synthetic_chimera
n = 16
s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=synthetic_chimera(n, n, 20260004), beta=3.0)
kw = dict(graduate_truncation=True, Dmax=64, tolS=1e-16, tolV=1e-10, max_sweeps=20)
s._setup_rhoT(**kw); torch.cuda.synchronize()
def sweep():
    torch.cuda.synchronize(); t0 = time.perf_counter(); s._setup_rhoT(**kw); torch.cuda.synchronize(); return time.perf_counter() - t0
print('alone: %.2f s' % sweep())
stop = False
def busy(size):
    st = torch.cuda.Stream(priority=0)
    a = torch.randn(size, size, device='cuda'); b = torch.randn(size, size, device='cuda')
    with torch.cuda.stream(st):
        while not stop:
            for _ in range(20): torch.matmul(a, b)
            st.synchronize()
for size in (512, 2048):
    stop = False
    th = threading.Thread(target=busy, args=(size,)); th.start(); time.sleep(1.0)
    print('with background fp32 matmul %d: %.2f s' % (size, sweep()))
    stop = True; th.join()
