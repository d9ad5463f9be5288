"""Phase timestamps of sq_kernel (build with TN_EXTRA_HIPCC_FLAGS=-DSQ_CLOCKS): workgroup 0, thread 0, 100 MHz clock."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tnac4o_amd import ops, _lib
L = _lib.lib()
g = torch.Generator(device='cpu').manual_seed(1)
names = ['load', 'gram0+gather', 'factor0', '(sync)', 'substitute', 'racc', 'gram1+gather', 'factor1', '(loop exit)', 'store']
for m, n in ((128, 64), (1024, 64), (4096, 64), (256, 32), (2048, 32)):
    T = torch.randn(m, n, dtype=torch.float64, generator=g).cuda()
    for _ in range(3):
        Q, R = ops.qr(T)
    torch.cuda.synchronize()
    buf = (C.c_longlong * 32)()
    L.tn_debug_sq_clocks(buf, 32)
    t = [buf[i] for i in range(10)]
    print('%5d x %3d ' % (m, n) + '  '.join('%s %.1f' % (names[i + 1], (t[i + 1] - t[i]) / 100.0) for i in range(9)) + '   total %.1f us' % ((t[9] - t[0]) / 100.0))
