"""What switches the 4-chain run between the fast (1.65 s/sweep) and slow (2.45) regime?  Flags (argv[1], comma list):
cached = same Stream objects every step; clearws = drop ops workspaces before each step; emptycache = torch.cuda.empty_cache()
before each step; nowait = no wait_stream on the default stream; keep = keep previous rhoT alive (no frees during the step)."""
import os, sys, time, threading
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import tnac4o_amd
import tnac4o_amd.ops as ops
from tnac4o_amd.auxx import synthetic_chimera
flags = set((sys.argv[1] if len(sys.argv) > 1 else '').split(','))
n = 16
J = synthetic_chimera(n, n, 20260004)
kw = dict(graduate_truncation=True, Dmax=64, tolS=1e-16, tolV=1e-10, max_sweeps=20)
def make(rot):
    s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=J, beta=3.0)
    if rot: s.rotate_graph(rot)
    return s
solvers = [make(g) for g in range(4)]
cached = [torch.cuda.Stream() for _ in range(4)] if 'cached' in flags else None
graveyard = []
for step in range(5):
    if 'clearws' in flags: ops._ws.clear()
    if 'emptycache' in flags: torch.cuda.empty_cache()
    streams = cached or [torch.cuda.Stream() for _ in range(4)]
    cur = torch.cuda.current_stream()
    def work(i):
        with torch.cuda.stream(streams[i]):
            if 'nowait' not in flags: streams[i].wait_stream(cur)
            if 'keep' in flags and hasattr(solvers[i], 'rhoT'): graveyard.append(solvers[i].rhoT)
            solvers[i]._setup_rhoT(**kw)
            streams[i].synchronize()
    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    if step >= 2: print(sorted(flags), 'step', step, round(1e3 * (time.perf_counter() - t0) / 4, 1), 'ms/sweep', flush=True)
    if 'keep' in flags and len(graveyard) > 8: del graveyard[:4]
