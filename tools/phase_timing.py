"""Wall time per phase of the sweep (synchronised), for finding where a sweep spends its time."""
import os, sys, time, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import tnac4o_amd
from tnac4o_amd import mps, ops
from tnac4o_amd.auxx import synthetic_chimera
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = {128: 4, 512: 8, 2048: 16}[L]
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
def timed(obj, name, label=None):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc[label or name] += time.perf_counter() - t; cnt[label or name] += 1
        return r
    setattr(obj, name, g)
for nm in ['mm', 'absorb']:
    timed(ops, nm, 'ops.' + nm)
shape_acc = collections.defaultdict(float); shape_cnt = collections.defaultdict(int)
_qr = ops.qr_into
def qr_timed(T, *a, **k):
    torch.cuda.synchronize(); t = time.perf_counter()
    r = _qr(T, *a, **k)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    acc['ops.qr_into'] += dt; cnt['ops.qr_into'] += 1
    m, n_ = T.shape
    key = (1 << (max(m, 1) - 1).bit_length(), 1 << (max(n_, 1) - 1).bit_length())    # rounded up to powers of two
    shape_acc[key] += dt; shape_cnt[key] += 1
    return r
ops.qr_into = qr_timed
svd_acc = collections.defaultdict(float); svd_cnt = collections.defaultdict(int)
def shaped(name):
    f = getattr(ops, name)
    def g(Cm, *a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = f(Cm, *a, **k)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        acc['ops.' + name] += dt; cnt['ops.' + name] += 1
        m, n_ = Cm.shape
        key = (name, 1 << (max(min(m, n_), 1) - 1).bit_length(), 1 << (max(m, n_, 1) - 1).bit_length())
        svd_acc[key] += dt; svd_cnt[key] += 1
        return r
    setattr(ops, name, g)
shaped('svd_trunc'); shaped('svdvals')
s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=synthetic_chimera(n, n, 20260004), beta=3.0)
orig = mps.MPS.compress_mps
def compress(self, Dmax, tolS, tolV, max_sweeps, graduate_truncation=True, verbose=False):
    def ph(label, fn):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); acc[label] += time.perf_counter() - t; return r
    ph('p1 canonise_right', self.canonise_right)
    phi = ph('copy', self.copy)
    self.discarded = [0] * (self.L + 1)
    ph('p2 canonise_left 4chi', lambda: self.canonise_left(compress=True, Dmax=Dmax * 4, tol=tolS / 10))
    ph('v1 variational', lambda: self.variational_compress(phi, tol=tolV, max_sweeps=1))
    ph('p3 canonise_right 2chi', lambda: self.canonise_right(compress=True, Dmax=Dmax * 2, tol=tolS / 2))
    ph('p4 canonise_left chi', lambda: self.canonise_left(compress=True, Dmax=Dmax, tol=tolS))
    return ph('v2 variational', lambda: self.variational_compress(phi, tol=tolV, max_sweeps=max_sweeps))
mps.MPS.compress_mps = compress
t0 = time.perf_counter()
s._setup_rhoT(graduate_truncation=True, Dmax=chi, tolS=1e-16, tolV=1e-10, max_sweeps=20)
torch.cuda.synchronize(); tot = time.perf_counter() - t0
print('total %.2f s (with per-call syncs)' % tot)
for k in sorted(acc, key=lambda k: -acc[k]):
    print('%-26s %8.3f s  %6d calls' % (k, acc[k], cnt[k]))
print('QR time by shape (rows, cols rounded up to powers of two):')
for k in sorted(shape_acc, key=lambda k: -shape_acc[k])[:14]:
    print('  %6d x %5d  %8.3f s  %5d calls  %7.2f ms each' % (k[0], k[1], shape_acc[k], shape_cnt[k], 1e3 * shape_acc[k] / shape_cnt[k]))
print('SVD time by shape (min, max dimension rounded up to powers of two):')
for k in sorted(svd_acc, key=lambda k: -svd_acc[k])[:14]:
    print('  %-10s %5d x %5d  %8.3f s  %5d calls  %7.2f ms each' % (k[0], k[1], k[2], svd_acc[k], svd_cnt[k], 1e3 * svd_acc[k] / svd_cnt[k]))
