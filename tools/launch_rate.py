import os, sys, time, threading
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, ctypes as C
from tnac4o_amd import _lib
L = _lib.lib()
N = 60000
for G in (1, 2, 4):
    xs = [torch.ones(8, dtype=torch.float64, device='cuda') for _ in range(G)]
    streams = [torch.cuda.Stream() for _ in range(G)]
    def work(i):
        x = xs[i]; st = C.c_void_p(streams[i].cuda_stream); p = x.data_ptr()
        for _ in range(N):
            L.tn_scale_by(p, 1, p + 8, st)
        streams[i].synchronize()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(G)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    print('G=%d: %d launches in %.2f s -> %.2f us per launch (aggregate %.0f k launches/s)' % (G, G * N, dt, dt / (G * N) * 1e6, G * N / dt / 1e3))
