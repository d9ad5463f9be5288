"""Phase clocks of the TSQR kernels (diagnostics build: TN_EXTRA_HIPCC_FLAGS=-DTN_CLOCKS python -c 'from tnac4o_amd import _lib; _lib.build()')."""
import os, sys, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from tnac4o_amd import ops, _lib
L = _lib.lib()
L.tn_debug_clocks.argtypes = [C.POINTER(C.c_longlong), C.c_int]
X = torch.randn(16384, 32, dtype=torch.float64, device='cuda')
for rep in range(3):
    Q, R_ = ops.qr(X.clone())
    torch.cuda.synchronize()
    buf = (C.c_longlong * 64)()
    L.tn_debug_clocks(buf, 64)
    c = list(buf)
    print('factor: load %d  amax %d  prep %d  columns %d  store %d   (total %d)' % (c[1]-c[0], 0, c[2]-c[1], c[3]-c[2], c[4]-c[3], c[4]-c[0]))
    print('factor column 5: barrier1 %d  dots %d  barrier2 %d  scalar+update %d  (column %d)' % (c[17]-c[16], c[18]-c[17], c[19]-c[18], c[20]-c[19], c[20]-c[16]))
    print('apply : load %d  gram %d  inverse %d  small %d  mfma %d  store %d   (total %d)' % (c[9]-c[8], c[10]-c[9], c[11]-c[10], c[12]-c[11], c[13]-c[12], c[14]-c[13], c[14]-c[8]))
