"""Launch-rate probe: N host threads, each with its own stream, each enqueueing K dependent tiny kernels (tn_gemm 1 x 1 x 1) -- time
per launch for 1 / 2 / 4 chains, and the same with a kernel that touches 32 MB (cache write-back / invalidate at its boundaries)."""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tnac4o_amd  # noqa: F401  (sets GPU_MAX_HW_QUEUES)
from tnac4o_amd import ops

K = 20000


def chain(i, streams, bufs, kind, out):
    with torch.cuda.stream(streams[i]):
        a, b, c, big = bufs[i]
        torch.cuda.current_stream().synchronize()
        t0 = time.perf_counter()
        if kind == 'tiny':
            for _ in range(K):
                ops.mm(a, b, out=c)
        else:
            for _ in range(K // 20):
                ops.scale_(big, a.view(-1))
        torch.cuda.current_stream().synchronize()
        out[i] = time.perf_counter() - t0


for kind in ('tiny', 'big'):
    for n in (1, 2, 4):
        streams = [torch.cuda.Stream() for _ in range(n)]
        bufs = [(torch.ones(1, 1, dtype=torch.float64, device='cuda'), torch.ones(1, 1, dtype=torch.float64, device='cuda'),
                 torch.ones(1, 1, dtype=torch.float64, device='cuda'), torch.ones(4 << 20, dtype=torch.float64, device='cuda')) for _ in range(n)]
        out = [0.0] * n
        th = [threading.Thread(target=chain, args=(i, streams, bufs, kind, out)) for i in range(n)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        cnt = K if kind == 'tiny' else K // 20
        print('%s kernels, %d chain(s): %.2f us per launch per chain' % (kind, n, 1e6 * max(out) / cnt))
