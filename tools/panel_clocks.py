"""TN_CLOCKS build: phase timestamps (100 MHz) of the last workgroup of pass launches 1 and 2 of one panel."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panel_probe import panel_orth
from tnac4o_amd._lib import lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
X = torch.randn(n, 32, dtype=torch.float64, device='cuda')
if len(sys.argv) > 2:
    kappa = float(sys.argv[2])
    U, _ = torch.linalg.qr(X)
    V, _ = torch.linalg.qr(torch.randn(32, 32, dtype=torch.float64, device='cuda'))
    X = (U * torch.logspace(0, -torch.log10(torch.tensor(kappa)).item(), 32, dtype=torch.float64, device='cuda')[None, :]) @ V.t()
Y = torch.empty_like(X)
names = ['start', 'loads+state', 'subst', 'store+gram', 'fence', 'ticket', 'fence2', 'tail']
for it in range(4):
    panel_orth(X, 0, out=Y)
    torch.cuda.synchronize()
    buf = (C.c_longlong * 32)()
    C.CDLL(lib()._name).tn_debug_clocks2(buf, 32)
    for l in range(2):
        t = [buf[12 * l + i] for i in range(8)]
        if t[0] == 0:
            continue
        print('n=%d iter %d launch %d: ' % (n, it, l + 1) + '  '.join('%s %.2f' % (names[i], (t[i] - t[i - 1]) / 100.0) for i in range(1, 8) if t[i]) + '  total %.2f us' % ((max(t) - t[0]) / 100.0))
