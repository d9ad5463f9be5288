"""Phase timestamps of cq_fused_kernel (build with TN_EXTRA_HIPCC_FLAGS=-DTN_CLOCKS): workgroup 0, thread 0, 100 MHz clock.
One tn_qr of m x 64 (two panels; the stamps are those of the LAST panel, of m - 32 rows)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tnac4o_amd import ops, _lib
os.environ['TN_QR_SMALL'] = '0'
L = _lib.lib()
g = torch.Generator(device='cpu').manual_seed(1)
names = ['load', 'gram0', 'barrier', 'reduce+chol', 'substitute', 'gram1', 'barrier', '(to exit: tail1)', 'lu', 'post']
for m in (288, 1056, 2080, 4128, 8224, 16416):
    T = torch.randn(m, 64, dtype=torch.float64, generator=g).cuda()
    for _ in range(3):
        Q, R = ops.qr(T)
    torch.cuda.synchronize()
    buf = (C.c_longlong * 32)()
    L.tn_debug_clocks2(buf, 32)
    t = [buf[i] for i in range(11)]
    print('%6d rows ' % (m - 32) + '  '.join('%s %.1f' % (names[i], (t[i + 1] - t[i]) / 100.0) for i in range(10)) + '   total %.1f us' % ((t[10] - t[0]) / 100.0)
          + '   [lu: load %.1f  eliminate 1 %.1f  inverses 1 %.1f  Schur %.1f  eliminate 2 %.1f  inverses 2 %.1f  off-diagonal %.1f  T %.1f  products+stores %.1f]' % (
              (buf[14] - t[8]) / 100.0, (buf[15] - buf[14]) / 100.0, (buf[16] - buf[15]) / 100.0, (buf[17] - buf[16]) / 100.0, (buf[18] - buf[17]) / 100.0,
              (buf[19] - buf[18]) / 100.0, (buf[12] - buf[19]) / 100.0, (buf[13] - buf[12]) / 100.0, (t[9] - buf[13]) / 100.0))
