"""Do a latency-bound kernel chain and device-filling GEMMs overlap when ONE host thread feeds them to two streams?
A: 20 x tn_qr 4096 x 256 (panel chains) on stream a;  B: 60 x GEMM 16384 x 1024 x 1024 on stream b;  A and B together."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tnac4o_amd import ops

X = torch.randn(4096, 256, dtype=torch.float64, device='cuda')
Q = torch.empty(4096, 256, dtype=torch.float64, device='cuda'); R = torch.empty(256, 256, dtype=torch.float64, device='cuda')
G1 = torch.randn(16384, 1024, dtype=torch.float64, device='cuda'); G2 = torch.randn(1024, 1024, dtype=torch.float64, device='cuda')
Go = torch.empty(16384, 1024, dtype=torch.float64, device='cuda')
work = [X.clone() for _ in range(20)]


def chain(st):
    with torch.cuda.stream(st):
        for w in work:
            ops.qr_into(w.clone(), Q, R, overwrite=True)


def gemms(st, n=60):
    with torch.cuda.stream(st):
        for _ in range(n):
            ops.mm(G1, G2, out=Go)


for mode in ('pool+pool', 'default+pool'):
    a = torch.cuda.Stream() if mode == 'pool+pool' else torch.cuda.default_stream()
    b = torch.cuda.Stream()
    chain(a); gemms(b, 3); torch.cuda.synchronize()
    t0 = time.perf_counter(); chain(a); torch.cuda.synchronize(); ta = time.perf_counter() - t0
    t0 = time.perf_counter(); gemms(b); torch.cuda.synchronize(); tb = time.perf_counter() - t0
    t0 = time.perf_counter(); gemms(b); chain(a); torch.cuda.synchronize(); tab = time.perf_counter() - t0
    t0 = time.perf_counter(); chain(a); gemms(b); torch.cuda.synchronize(); tba = time.perf_counter() - t0
    print('%s: chain %.1f ms  gemms %.1f ms  together (gemms issued first) %.1f ms  (chain issued first) %.1f ms  sum %.1f' %
          (mode, 1e3 * ta, 1e3 * tb, 1e3 * tab, 1e3 * tba, 1e3 * (ta + tb)), flush=True)
