"""Per-dispatch durations of the panel kernels (run under rocprofv3 --kernel-trace): 50 panels of one shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panel_probe import panel_orth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
method = int(sys.argv[2]) if len(sys.argv) > 2 else 0
X = torch.randn(n, 32, dtype=torch.float64, device='cuda')
if len(sys.argv) > 3:      # ill-conditioned variant: kappa
    kappa = float(sys.argv[3])
    U, _ = torch.linalg.qr(X)
    V, _ = torch.linalg.qr(torch.randn(32, 32, dtype=torch.float64, device='cuda'))
    X = (U * torch.logspace(0, -torch.log10(torch.tensor(kappa)).item(), 32, dtype=torch.float64, device='cuda')[None, :]) @ V.t()
Y = torch.empty_like(X)
for _ in range(50):
    panel_orth(X, method, out=Y)
torch.cuda.synchronize()
