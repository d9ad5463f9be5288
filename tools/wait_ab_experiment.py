"""Same process, alternating steps: chains started with / without streams[i].wait_stream(default stream)."""
import os, sys, time, threading
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import tnac4o_amd
from tnac4o_amd.auxx import synthetic_chimera
n = 16
J = synthetic_chimera(n, n, 20260004)
kw = dict(graduate_truncation=True, Dmax=64, tolS=1e-16, tolV=1e-10, max_sweeps=20)
def make(rot):
    s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=J, beta=3.0)
    if rot: s.rotate_graph(rot)
    return s
solvers = [make(g) for g in range(4)]
acc = {True: [], False: []}
for step in range(9):
    wait = (step % 2 == 1)
    streams = [torch.cuda.Stream() for _ in range(4)]
    cur = torch.cuda.current_stream()
    def work(i):
        with torch.cuda.stream(streams[i]):
            if wait: streams[i].wait_stream(cur)
            solvers[i]._setup_rhoT(**kw)
            streams[i].synchronize()
    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = 1e3 * (time.perf_counter() - t0) / 4
    if step >= 1: acc[wait].append(dt)
    print('step', step, 'wait' if wait else 'nowait', round(dt, 1), flush=True)
print('mean wait', round(sum(acc[True]) / len(acc[True]), 1), 'mean nowait', round(sum(acc[False]) / len(acc[False]), 1))
