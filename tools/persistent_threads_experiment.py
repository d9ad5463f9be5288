"""Cached streams driven by persistent worker threads (one per chain, alive across steps) vs fresh threads per step."""
import os, sys, time, threading, queue
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
streams = [torch.cuda.Stream() for _ in range(4)]
qs = [queue.Queue() for _ in range(4)]
done = queue.Queue()
VAR = sys.argv[1] if len(sys.argv) > 1 else ''
def worker(i):
    if 'setdev' in VAR: torch.cuda.set_device(0)
    with torch.cuda.stream(streams[i]):
        while True:
            job = qs[i].get()
            if job is None: return
            if 'wait' in VAR: streams[i].wait_event(job)
            if 'try' in VAR:
                try:
                    solvers[i]._setup_rhoT(**kw)
                except BaseException as e:
                    print(e)
            else:
                solvers[i]._setup_rhoT(**kw)
            streams[i].synchronize()
            done.put(i)
th = [threading.Thread(target=worker, args=(i,), daemon=True) for i in range(4)]
for t in th: t.start()
for step in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ev = 1
    if 'event' in VAR or 'wait' in VAR:
        ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream())
    for q in qs: q.put(ev)
    for _ in range(4): done.get()
    torch.cuda.synchronize(); print(VAR, 'persistent threads step', step, round(1e3 * (time.perf_counter() - t0) / 4, 1), 'ms/sweep', flush=True)
for q in qs: q.put(None)
