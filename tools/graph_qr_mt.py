import os, sys, time, threading
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from tnac4o_amd import ops
m, n, G, REP = 4096, 1024, 4, 10
T0 = torch.randn(m, n, dtype=torch.float64, device='cuda')
ctx = []
for i in range(G):
    s = torch.cuda.Stream()
    T = T0.clone(); Q = torch.empty(m, n, dtype=torch.float64, device='cuda'); Rm = torch.empty(n, n, dtype=torch.float64, device='cuda')
    def direct(T=T, Q=Q, Rm=Rm):
        T.copy_(T0); ops.qr_into(T, Q, Rm, overwrite=True)
    with torch.cuda.stream(s):
        direct(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            direct()
        torch.cuda.synchronize()
    ctx.append((s, direct, g))
def run(mode, G):
    def work(i):
        s, direct, g = ctx[i]
        with torch.cuda.stream(s):
            for _ in range(REP):
                g.replay() if mode == 'graph' else direct()
            s.synchronize()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(G)]
    [t.start() for t in th]; [t.join() for t in th]
    return (time.perf_counter() - t0) / (REP * G)
for G in (1, 4):
    print('G=%d  direct %.3f ms/QR   graph %.3f ms/QR' % (G, run('direct', G) * 1e3, run('graph', G) * 1e3))
