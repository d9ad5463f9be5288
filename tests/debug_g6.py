"""Diagnostic (not collected by pytest): where does the HIP path's deviation in the conditional-probability tables (G6)
come from — the boundary sweep or the beam kernels?  (a) GPU search vs golden; (b) the CPU oracle's beam run on the GPU's
boundary MPS vs golden (isolates the sweep); (c) GPU beam vs oracle beam on the same boundary MPS (isolates K8/K9)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np                     # noqa: E402
import golden_inputs as gi             # noqa: E402
from oracle import mps_ref as mr, solver_ref as sr   # noqa: E402
import tnac4o_amd                      # noqa: E402
from tnac4o_amd import mps as gmps     # noqa: E402

g = np.load(os.path.join(gi.GOLDEN_DIR, 'g6_pn.npz'))
g5 = np.load(os.path.join(gi.GOLDEN_DIR, 'g5_sweep.npz'))


def cmp(tr, tag, other=None):
    out = []
    for k in g[tag + '_steps']:
        st = int(g[tag + '_stride%d' % k][0])
        a = tr[k][2][::st]
        b = g[tag + '_P%d' % k] if other is None else other[k][2][::st]
        d = np.abs(a - b)
        m = np.abs(b) > 1e-12
        out.append((int(k), float(d.max()), float((d[m] / np.abs(b[m])).max())))
    return out


nsweeps = []
orig_vc = gmps.MPS.variational_compress


def counting_vc(self, phi, tol=None, max_sweeps=1, verbose=False):
    # count executed sweeps by wrapping update_RL_mix calls at the last site
    before = len(nsweeps)
    cnt = [0]
    orig = self.update_RL_mix

    def upd(phi_, n):
        if n == self.L - 1:
            cnt[0] += 1
        return orig(phi_, n)
    self.update_RL_mix = upd
    try:
        r = orig_vc(self, phi, tol=tol, max_sweeps=max_sweeps, verbose=verbose)
    finally:
        del self.update_RL_mix
    nsweeps.append((max_sweeps, cnt[0] - 1))     # setup_RL_mix calls it once per site too
    return r


gmps.MPS.variational_compress = counting_vc

for rot, chi in [(0, 8), (3, 8), (0, 32)]:
    tag = 'L128_r%d_chi%d' % (rot, chi)
    nsweeps.clear()
    s = tnac4o_amd.tnac4o(mode='Ising', Nx=4, Ny=4, Nc=8, J=gi.droplet_J(128, 1), beta=3.0)
    if rot:
        s.rotate_graph(rot)
    tr = []
    s.search_ground_state(M=1024, relative_P_cutoff=1e-8, Dmax=chi, trace=tr)
    print(tag, 'variational sweeps executed (max_sweeps, done):', nsweeps)
    print(tag, 'D gpu   ', [m.D for m in s.rhoT])
    print(tag, 'D golden', g5[tag + '_D'].tolist() if tag + '_D' in g5.files else None)
    print(tag, '(a) gpu vs golden   ', cmp(tr, tag))
    o = sr.RefSolver(mode='Ising', Nx=4, Ny=4, Nc=8, J=gi.droplet_J(128, 1), beta=3.0)
    if rot:
        o.rotate_graph(rot)

    def hook(solver, run):
        solver.rhoT = []
        for m in s.rhoT:
            r = mr.RefMPS(d=[int(a.shape[1]) for a in m.A], L=len(m.A), Dmax=1, canonise=None)
            r.A = [a.detach().cpu().numpy() for a in m.A]
            r.D = list(m.D)
            solver.rhoT.append(r)
        solver.rhoT_overlap, solver.rhoT_discarded = list(s.rhoT_overlap), list(s.rhoT_discarded)
    tro = []
    o.search_ground_state(M=1024, relative_P_cutoff=1e-8, Dmax=chi, trace=tro, sweep_hook=hook)
    print(tag, '(b) oracle beam on gpu rhoT vs golden', cmp(tro, tag))
    print(tag, '(c) gpu beam vs oracle beam, same rhoT', cmp(tr, tag, tro))
    print(tag, 'log2P gpu %.15f oracle-on-gpu-rhoT %.15f' % (s.probability[0], o.probability[0]))
