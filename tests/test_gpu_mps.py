"""GPU parity tests at the level of the reference's call surface: MPS.compress_mps (G3), the boundary sweep
(G5), conditional probabilities (G6) and search_ground_state end results (G7), through tnac4o_amd (HIP path)
against the golden vectors captured from the reference and against the CPU oracle.  Tolerances:
energies 1e-10, log2-probabilities 1e-9, Pn relative 1e-10, overlap 1e-12, discarded rel 1e-6 / abs 1e-14."""
import json
import os

import numpy as np
import pytest

import golden_inputs as gi
from oracle import mps_ref as mr
from oracle import solver_ref as sr

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def load(name):
    return np.load(os.path.join(gi.GOLDEN_DIR, name))


def host_chain(psi):
    """Oracle-side MPS holding the device tensors of a tnac4o_amd.mps.MPS (for gauge-invariant overlaps)."""
    As = [a.detach().cpu().numpy() for a in psi.A]
    o = mr.RefMPS(d=[a.shape[1] for a in As], L=len(As), Dmax=1, canonise=None)
    o.A = As
    return o


def ref_chain(As):
    o = mr.RefMPS(d=[a.shape[1] for a in As], L=len(As), Dmax=1, canonise=None)
    o.A = [np.array(a) for a in As]
    return o


def fidelity(a, b):
    return abs(mr.mps_dot(a, b)) / np.sqrt(mr.mps_dot(a, a) * mr.mps_dot(b, b))


def gpu_solver(L=128, ins=1, rot=0, beta=3.0, pre=False, J=None):
    import tnac4o_amd
    n = {128: 4, 512: 8, 2048: 16}[L]
    s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=J if J is not None else gi.droplet_J(L, ins), beta=beta)
    if rot:
        s.rotate_graph(rot)
    if pre:
        s.precondition(mode='balancing')
    return s


# ------------------------------------------------------------------------------------------------ G3
@pytest.mark.parametrize('case', [0, 1])
def test_compress_golden(case):
    from tnac4o_amd import mps
    g = load('g3_compress.npz')
    L, D, p, b, chi = [(6, 6, 4, 4, 8), (8, 8, 16, 16, 16)][case]
    As = gi.rand_chain(31 + case, [1] + [D] * (L - 1) + [1], [p] * L)
    Ws = gi.rand_mpo(41 + case, L, b, p, p)
    for hconj in (True, False):
        for grad in (True, False):
            psi = mps.MPS(d=[p] * L, L=L, Dmax=1, canonise=None)
            psi.A = [mps._t(a) for a in As]
            psi.D = [1] + [a.shape[2] for a in As]
            mpo = mps.MPO(L=L)
            for n in range(L):
                mpo.set_direct(Ws[n], n)
            psi.apply_mpo(mpo, Hconj=hconj)
            ov = psi.compress_mps(Dmax=chi, tolS=1e-16, tolV=1e-10, max_sweeps=20, graduate_truncation=grad)
            tag = 'rand%d_h%d_g%d' % (case, int(hconj), int(grad))
            assert ov == pytest.approx(g[tag + '_overlap'][0], abs=1e-12)
            assert psi.D == list(g[tag + '_D'])
            np.testing.assert_allclose(np.array(psi.discarded, dtype=float), g[tag + '_discarded'], rtol=1e-6, atol=1e-14)
            for n in range(L + 1):
                np.testing.assert_allclose(psi.S[n], g[tag + '_S%d' % n], rtol=0, atol=1e-12)
            # left-canonical result: every site is an isometry
            for n in range(L):
                A = psi.A[n].detach().cpu().numpy()
                M = A.reshape(-1, A.shape[2])
                assert np.abs(M.T @ M - np.eye(M.shape[1])).max() < 1e-12
            if case == 0:
                assert fidelity(ref_chain([g[tag + '_A%d' % n] for n in range(L)]), host_chain(psi)) > 1 - 1e-12


# ------------------------------------------------------------------------------------------------ G5
@pytest.mark.parametrize('rot,chi', [(0, 8), (1, 8), (2, 8), (3, 8), (0, 32), (1, 32), (2, 32), (3, 32)])
def test_sweep_L128(rot, chi):
    g = load('g5_sweep.npz')
    s = gpu_solver(rot=rot)
    s._setup_rhoT(graduate_truncation=True, Dmax=chi, tolS=1e-16, tolV=1e-10, max_sweeps=20)
    tag = 'L128_r%d_chi%d' % (rot, chi)
    np.testing.assert_allclose(np.array(s.rhoT_overlap, dtype=float), g[tag + '_overlap'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.array(s.rhoT_discarded, dtype=float), g[tag + '_discarded'], rtol=1e-6, atol=1e-14)
    want_D = g[tag + '_D']
    got_D = np.array([m.D for m in s.rhoT])
    # bond dimensions are pinned where they are not decided at the eps noise floor (SURVEY.md §7 "hard parts")
    assert got_D.shape == want_D.shape and np.abs(got_D - want_D).max() <= (0 if chi == 8 else 2)
    if chi == 8:
        for ny in range(s.Ny + 1):
            phi = ref_chain([g[tag + '_A_%d_%d' % (ny, nx)] for nx in range(s.Nx)])
            assert fidelity(phi, host_chain(s.rhoT[ny])) > 1 - 1e-12


@pytest.mark.parametrize('hconj', [True, False])
def test_sweep_vs_oracle_rhoB(hconj):
    """rhoB uses the other absorption orientation (Hconj=False); compare one top-down sweep with the oracle."""
    a = gpu_solver(ins=2)
    b = sr.RefSolver(mode='Ising', Nx=4, Ny=4, Nc=8, J=gi.droplet_J(128, 2), beta=3.0)
    kw = dict(graduate_truncation=False, Dmax=8, tolS=1e-16, tolV=1e-10, max_sweeps=4)
    if hconj:
        a._setup_rhoT(**kw), b._setup_rhoT(**kw)
        ga, gb, oa, ob = a.rhoT, b.rhoT, a.rhoT_overlap, b.rhoT_overlap
    else:
        a._setup_rhoB(**kw), b._setup_rhoB(**kw)
        ga, gb, oa, ob = a.rhoB, b.rhoB, a.rhoB_overlap, b.rhoB_overlap
    np.testing.assert_allclose(np.array(oa, dtype=float), np.array(ob, dtype=float), atol=1e-12)
    for x, y in zip(ga, gb):
        assert fidelity(y, host_chain(x)) > 1 - 1e-12


# ------------------------------------------------------------------------------------------------ G6 / G7
def g7():
    with open(os.path.join(gi.GOLDEN_DIR, 'g7_search.json')) as f:
        return json.load(f)


_SENS = {}


def oracle_sensitivity(L, ins, rot, chi, pre):
    """|d log2P| of the CPU oracle itself when every QR input is perturbed by 1e-16 relative: the conditioning of the
    reference algorithm on this instance (the survey's gesdd->gesvd probe gives the same order).  Instances whose
    conditional tables contain negative entries (droplet #2) amplify rounding by ~1e7."""
    key = (L, ins, rot, chi, pre)
    if key not in _SENS:
        n = {128: 4, 512: 8}[L]
        out = []
        for perturb in (False, True):
            orig = mr.qr_pos
            if perturb:
                rng = np.random.default_rng(0)
                mr.qr_pos = lambda T: orig(T * (1 + 1e-16 * rng.standard_normal(T.shape)))
            try:
                b = sr.RefSolver(mode='Ising', Nx=n, Ny=n, Nc=8, J=gi.droplet_J(L, ins), beta=3.0)
                if rot:
                    b.rotate_graph(rot)
                if pre:
                    b.precondition()
                b.search_ground_state(M=1024, relative_P_cutoff=1e-8, Dmax=chi)
                out.append(b.probability[0])
            finally:
                mr.qr_pos = orig
        _SENS[key] = abs(out[0] - out[1])
    return _SENS[key]


def check_result(s, want, states=True, sens=0.0):
    assert s.energy[0] == pytest.approx(want['energy'], abs=1e-10)
    assert int(s.degeneracy) == want['degeneracy']
    # log2 P: 1e-9, or 10x the movement of the reference algorithm's own result under a 1e-16 perturbation
    # of its QR inputs where that is larger (only droplet #2, whose tables contain negative entries: 4e-10 per 1e-16).
    # Measured on the MI355X (profiles/r02_ranktol_study.json): <= 5.3e-10 on every G7 case, identical with the
    # rank-revealing early exit of the truncating QR passes switched off (rank_tol = 0); a different but equally converged
    # Jacobi run (noise rows left unorthogonalised) moves droplet #2 / chi=32 to 1.9e-9 and nothing else: the difference
    # is rounding amplified by the conditioning of the truncation, not a change of algorithm.
    assert s.probability[0] == pytest.approx(want['probability'], abs=max(1e-9, 10 * sens))
    assert s.discarded_probability == pytest.approx(want['discarded_probability'], abs=max(1e-8, 10 * sens))
    assert s.negative_probability == pytest.approx(want['negative_probability'], rel=1e-6, abs=1e-12)
    if states:
        assert len(s.energy) == want['n_states']
        assert [int(x) for x in s.states[0]] == want['state0']
        assert [int(x) for x in s.binary_states()[0]] == want['bits0']


G7_CASES = [(128, 1, 0, 8, False), (128, 1, 3, 8, False), (128, 1, 0, 32, False), (128, 2, 1, 8, False),
            (128, 3, 2, 8, False), (128, 2, 0, 32, False), (128, 1, 0, 8, True), (128, 3, 2, 32, True)]


@pytest.mark.parametrize('L,ins,rot,chi,pre', G7_CASES)
def test_search_golden(L, ins, rot, chi, pre):
    want = g7()['L%d_i%d_r%d_chi%d_pre%d' % (L, ins, rot, chi, int(pre))]
    s = gpu_solver(L=L, ins=ins, rot=rot, pre=pre)
    s.search_ground_state(M=1024, relative_P_cutoff=1e-8, Dmax=chi)
    check_result(s, want, sens=oracle_sensitivity(L, ins, rot, chi, pre) if ins == 2 else 0.0)
    E, bits = gi.golden_groundstate(L, ins)                    # the reference's own golden file
    assert s.energy[0] == pytest.approx(E, abs=1e-5)
    assert np.array_equal(s.binary_states()[0], bits)
    from tnac4o_amd import energy_Jij
    assert energy_Jij(gi.droplet_J(L, ins), s.binary_states()[:1])[0] == pytest.approx(s.energy[0], abs=1e-9)


def test_search_L512_golden():
    want = g7()['L512_i1_r0_chi32_pre0']
    s = gpu_solver(L=512)
    s.search_ground_state(M=1024, relative_P_cutoff=1e-8, Dmax=32)
    check_result(s, want)
    E, bits = gi.golden_groundstate(512, 1)
    assert s.energy[0] == pytest.approx(E, abs=1e-5) and np.array_equal(s.binary_states()[0], bits)


def test_search_L2048_droplet_golden_energy():
    """The reference's largest bundled instance (chimera2048 droplet #1): full search at chi=32 on the GPU.  The golden
    file gives the energy to 6 digits and one ground state; this instance is two-fold degenerate in the search
    (deg = 2), so the energy, its independent recomputation and the degeneracy are pinned, not the bit string."""
    from tnac4o_amd import energy_Jij
    s = gpu_solver(L=2048)
    s.search_ground_state(M=1024, relative_P_cutoff=1e-8, Dmax=32)
    E, bits = gi.golden_groundstate(2048, 1)
    assert s.energy[0] == pytest.approx(E, abs=1e-5)
    J = gi.droplet_J(2048, 1)
    assert energy_Jij(J, s.binary_states()[:1])[0] == pytest.approx(s.energy[0], abs=1e-8)
    assert energy_Jij(J, bits[None, :])[0] == pytest.approx(s.energy[0], abs=1e-8)      # the golden state has the same energy
    assert int(s.degeneracy) >= 1 and min(s.rhoT_overlap) > 1 - 1e-10


def test_search_J124_degeneracy_golden():
    """J124 C8 #1 (reference test_e06, examples/test_examples.py:139-147): energy -2309 and ground-state degeneracy
    1152 from the reference's results file, with preconditioning, chi=8, M=4096, beta=0.75 — the degeneracy-counting
    merge path of search_ground_state."""
    import tnac4o_amd
    want = g7()['J124_C8_i1_r0_chi8_pre1']
    s = tnac4o_amd.tnac4o(mode='Ising', Nx=8, Ny=8, Nc=8, J=gi.j124_J(1), beta=0.75)
    s.precondition(mode='balancing')
    s.search_ground_state(M=2 ** 12, relative_P_cutoff=1e-8, Dmax=8)
    assert s.energy[0] == pytest.approx(-2309.0, abs=1e-9) and s.energy[0] == pytest.approx(want['energy'], abs=1e-9)
    assert int(s.degeneracy) == 1152 == want['degeneracy']
    with open(os.path.join(gi.INST_DIR, 'C8_J124_results_1-5.txt')) as f:
        line = f.readlines()[1].split()
    assert float(line[1]) == -2309 and int(line[2]) == 1152
    from tnac4o_amd import energy_Jij
    assert energy_Jij(gi.j124_J(1), s.binary_states()[:1])[0] == pytest.approx(s.energy[0], abs=1e-9)


def test_search_rmf_dense_path_vs_oracle():
    """A larger Random Markov Field (8 x 8, d = 4, chi = 16): the dense non-chimera PEPS path of BASELINE config 5 at a
    size the oracle finishes in seconds; GPU search vs the CPU oracle."""
    import tnac4o_amd
    from tnac4o_amd.auxx import synthetic_rmf, energy_RMF
    J = synthetic_rmf(8, 8, 4, 20260005)
    a = tnac4o_amd.tnac4o(mode='RMF', Nx=8, Ny=8, J=J, beta=1.0)
    b = sr.RefSolver(mode='RMF', Nx=8, Ny=8, J=J, beta=1.0)
    kw = dict(M=256, relative_P_cutoff=1e-8, Dmax=16)
    a.search_ground_state(**kw)
    b.search_ground_state(**kw)
    assert a.energy[0] == pytest.approx(b.energy[0], abs=1e-10)
    assert np.array_equal(a.states[0], b.states[0]) and int(a.degeneracy) == int(b.degeneracy)
    assert a.probability[0] == pytest.approx(b.probability[0], abs=1e-8)
    assert energy_RMF(J, a.states[:1])[0] == pytest.approx(a.energy[0], abs=1e-10)
    np.testing.assert_allclose(np.array(a.rhoT_overlap, dtype=float), np.array(b.rhoT_overlap, dtype=float), atol=1e-10)


def test_interleaved_chains_match_sequential():
    """Four lattice rotations swept concurrently (4 host threads, 4 HIP streams, per-stream workspaces) give exactly the
    results of sweeping them one after the other — the mode bench.py measures."""
    from tnac4o_amd.parallel import run_concurrent
    kw = dict(graduate_truncation=True, Dmax=16, tolS=1e-16, tolV=1e-10, max_sweeps=20)
    seq = [gpu_solver(L=512, rot=r) for r in range(4)]
    for s in seq:
        s._setup_rhoT(**kw)
    par = [gpu_solver(L=512, rot=r) for r in range(4)]
    run_concurrent([(lambda s=s: s._setup_rhoT(**kw)) for s in par])
    for a, b in zip(seq, par):
        assert [m.D for m in a.rhoT] == [m.D for m in b.rhoT]
        assert a.rhoT_discarded == b.rhoT_discarded and a.rhoT_overlap == b.rhoT_overlap      # bit-identical
        for x, y in zip(a.rhoT, b.rhoT):
            assert all(torch.equal(p, q) for p, q in zip(x.A, y.A))


def test_search_rmf():
    import tnac4o_amd
    J = gi.minimal_rmf()
    for rot in (0, 1):
        s = tnac4o_amd.tnac4o(mode='RMF', Nx=J['Nx'], Ny=J['Ny'], J=J, beta=2.0)
        if rot:
            s.rotate_graph(rot)
        s.search_ground_state(M=64, relative_P_cutoff=1e-8, Dmax=8)
        check_result(s, g7()['RMF_r%d' % rot])
        assert tnac4o_amd.energy_RMF(J, s.states[:1])[0] == pytest.approx(s.energy[0], abs=1e-10)


def test_calc_pn_negative_rule():
    """The negative-probability rule of tnac4o.py:1795-1807 on synthetic inputs, against the oracle's function."""
    from tnac4o_amd import ops
    rng = np.random.default_rng(3)
    q, nl, nu, p, Dr, br, npref, nsuf, nb = 256, 4, 3, 16, 5, 16, 3, 4, 9
    F = rng.uniform(0, 1, (q, nl, nu))
    T1 = rng.standard_normal((npref, p, Dr))
    RR = rng.standard_normal((nsuf, Dr, br))
    RR[0] = 0.0                                                   # an all-zero branch -> uniform, flag -1
    T1[2] = np.abs(T1[2])
    RR[3] = np.abs(RR[3])                                         # an all-positive branch
    dmap, rmap = rng.integers(0, p, q), rng.integers(0, br, q)
    pref, suf = rng.integers(0, npref, nb), rng.integers(0, nsuf, nb)
    suf[0], pref[1], suf[1] = 0, 2, 3
    lidx, uidx = rng.integers(0, nl, nb), rng.integers(0, nu, nb)
    dv = lambda x: torch.as_tensor(np.ascontiguousarray(x)).cuda()      # noqa: E731
    P, mP = ops.calc_pn(dv(T1), dv(RR), dv(F), dv(dmap.astype(np.int32)), dv(rmap.astype(np.int32)),
                        dv(pref.astype(np.int32)), dv(suf.astype(np.int32)), dv(lidx.astype(np.int32)),
                        dv(uidx.astype(np.int32)))
    P, mP = P.cpu().numpy(), mP.cpu().numpy()
    for k in range(nb):
        T2 = T1[pref[k]] @ RR[suf[k]]
        Pn = F[:, lidx[k], uidx[k]] * T2[dmap, rmap]
        m = Pn.min()
        if m < 0:
            low = Pn < abs(m)
            Pn[low] = abs(m)
            m *= low.sum()
        no = Pn.sum()
        if no > 0:
            Pn, m = Pn / no, m / no
        else:
            Pn, m = Pn + 1.0 / q, -1
        np.testing.assert_allclose(P[k], Pn, rtol=1e-12, atol=1e-300)
        assert mP[k] == pytest.approx(m, rel=1e-10, abs=1e-300)
    assert mP[0] == -1 and mP[1] >= 0


def test_product_path_uses_hip_library():
    """The loaded shared object is the in-tree libtnpeps.so and the product modules never import the oracle."""
    import sys
    import tnac4o_amd
    from tnac4o_amd import _lib
    assert os.path.samefile(_lib.lib()._name, os.path.join(os.path.dirname(tnac4o_amd.__file__), 'libtnpeps.so'))
    for name in ('tnac4o_amd.tnac4o', 'tnac4o_amd.mps', 'tnac4o_amd.ops', 'tnac4o_amd._lib', 'tnac4o_amd.auxx'):
        src = open(sys.modules[name].__file__).read()
        assert 'oracle' not in src.replace("'oracle' not in", '')


BEAM_GPU_WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import numpy as np, torch, torch.distributed as dist
import golden_inputs as gi
import tnac4o_amd
from tnac4o_amd.parallel import solve_rotations
torch.cuda.set_device(0)
dist.init_process_group('gloo')          # two ranks share the one GPU of the test box, so the exchange runs over gloo
J = gi.droplet_J(128, 1)
make = lambda: tnac4o_amd.tnac4o(mode='Ising', Nx=4, Ny=4, Nc=8, J=J, beta=3.0)
from tnac4o_amd import beam
calls = {'native_team': 0, 'torch': 0}
orig_native, orig_device = beam.search_native, beam.search_device
def spy_native(*a, **k):
    out = orig_native(*a, **k)
    if out is not None and k.get('beam_group') is not None:
        calls['native_team'] += 1
    return out
def spy_device(*a, **k):
    calls['torch'] += 1
    return orig_device(*a, **k)
beam.search_native, beam.search_device = spy_native, spy_device
res = solve_rotations(make, rotations=(0,), beam_shards=2, M=256, relative_P_cutoff=1e-8, Dmax=16)
res['state'] = [int(x) for x in res['state']]
res['calls'] = calls
print('RESULT ' + json.dumps(res), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.gpu
@pytest.mark.parametrize('team_walk', ['native', 'torch'])
def test_beam_sharded_product_path_two_ranks(team_walk):
    """SURVEY.md 8e-ii on the product path: 2 processes (one GPU, gloo) split every site-step's branches; both must
    reproduce the single-process result and the reference's golden ground state.  'native': the team walks the search in the
    library (tn_beam_search_team: the conditional tables split over the ranks, completed by the exchange hook -- here broadcasts
    over gloo) and must agree with the single-rank library walk BIT FOR BIT; 'torch': the torch driver with its pruned
    candidate exchange (TN_BEAM_TEAM=torch)."""
    import subprocess
    import sys as _sys
    import json as _json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = BEAM_GPU_WORKER % dict(root=root)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29577' if team_walk == 'native' else '29578', TN_BEAM_TEAM=team_walk)
    procs = [subprocess.Popen([_sys.executable, '-c', code], env=dict(env, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK='0'),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-3000:]
        outs.append(_json.loads([l for l in out.splitlines() if l.startswith('RESULT ')][-1][7:]))
    assert outs[0] == outs[1]
    assert (outs[0]['calls']['native_team'] == 1 and outs[0]['calls']['torch'] == 0) if team_walk == 'native' else outs[0]['calls']['torch'] == 1
    import tnac4o_amd
    J = gi.droplet_J(128, 1)
    s = tnac4o_amd.tnac4o(mode='Ising', Nx=4, Ny=4, Nc=8, J=J, beta=3.0)
    s.search_ground_state(M=256, relative_P_cutoff=1e-8, Dmax=16)
    if team_walk == 'native':
        assert outs[0]['probability'] == float(s.probability[0]) and outs[0]['degeneracy'] == int(s.degeneracy)
    assert outs[0]['energy'] == float(s.energy[0])
    assert outs[0]['state'] == [int(x) for x in s.states[0]]
    assert outs[0]['probability'] == pytest.approx(float(s.probability[0]), abs=1e-12)
    E, _ = gi.golden_groundstate(128, 1)
    assert outs[0]['energy'] == pytest.approx(E, abs=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('rot,chi,M,seed', [(0, 16, 64, 1234), (1, 8, 32, 99)])
def test_gibbs_sampling_golden(rot, chi, M, seed):
    """Gibbs sampling (tnac4o.py:553-650) on the HIP path draws the reference's configurations for a seeded run
    (the conditional tables agree to ~1e-12, so a uniform draw would have to fall that close to a boundary to differ)."""
    g = load('g8_gibbs.npz')
    tag = 'r%d_chi%d_M%d_seed%d' % (rot, chi, M, seed)
    s = gpu_solver(rot=rot)
    np.random.seed(seed)
    E = s.gibbs_sampling(M=M, Dmax=chi)
    assert len(s.states) == M
    assert np.array_equal(np.asarray(s.states).astype(np.int64), g[tag + '_states'].astype(np.int64))
    np.testing.assert_allclose(E, g[tag + '_energy'], rtol=0, atol=1e-10)
    assert np.array_equal(s.binary_states(), g[tag + '_bits'])
    from tnac4o_amd import auxx
    assert np.abs(auxx.energy_Jij(gi.droplet_J(128, 1), s.binary_states()) - E).max() < 1e-6


def _spectrum_same(s, g, tag, bits=True):
    assert len(s.energy) == len(g[tag + '_energy'])
    np.testing.assert_allclose(s.energy, g[tag + '_energy'], rtol=0, atol=1e-9)
    got = s.binary_states() if bits else np.asarray(s.states)
    want = g[tag + ('_bits' if bits else '_states')]
    assert sorted(map(bytes, np.asarray(got, dtype=np.int16))) == sorted(map(bytes, np.asarray(want, dtype=np.int16)))


@pytest.mark.gpu
@pytest.mark.parametrize('rot,chi', [(0, 16), (1, 16), (3, 8)])
def test_low_energy_spectrum_golden(rot, chi, tmp_path):
    """search_low_energy_spectrum (encoding 1) + decode on the HIP path: the reference's 31 states within dE < 1 of
    droplet instance 1 (examples/test_examples.py test_e03), same energies and bit strings; result file round trip."""
    import tnac4o_amd
    g = load('g10_spectrum.npz')
    s = gpu_solver(rot=rot)
    s.search_low_energy_spectrum(excitations_encoding=1, M=1024, relative_P_cutoff=1e-8, Dmax=chi, max_dEng=1.0, lim_hd=0)
    tag = 'L128_i1_r%d_chi%d' % (rot, chi)
    assert [len(s.d), len(s.el)] == list(g[tag + '_n_shapes'])
    f = str(tmp_path / 'spectrum.npy')
    s.save(f)
    s.decode_low_energy_states(max_dEng=1.0)
    assert len(s.energy) == 31
    _spectrum_same(s, g, tag)
    t = tnac4o_amd.load(f)                                    # decode again from the saved forest
    t.decode_low_energy_states(max_dEng=1.0)
    assert np.array_equal(t.energy, s.energy) and np.array_equal(t.states, s.states)


@pytest.mark.gpu
def test_low_energy_spectrum_rmf_golden():
    """examples/e05 (test_e05): 26 states within dE < 3.1 of the minimal RMF model, from two lattice rotations."""
    import tnac4o_amd
    g = load('g10_spectrum.npz')
    J = gi.e05_rmf()
    for rot in (0, 1):
        s = tnac4o_amd.tnac4o(mode='RMF', Nx=J['Nx'], Ny=J['Ny'], J=J, beta=4)
        if rot:
            s.rotate_graph(rot)
        s.search_low_energy_spectrum(excitations_encoding=1, M=1024, relative_P_cutoff=1e-12, Dmax=32, max_dEng=3.1, lim_hd=0)
        s.decode_low_energy_states(max_dEng=3.1, max_states=100)
        assert len(s.energy) == 26
        _spectrum_same(s, g, 'RMF_r%d' % rot, bits=False)
    with pytest.raises(ValueError):
        s.search_low_energy_spectrum(excitations_encoding=4)


@pytest.mark.gpu
@pytest.mark.parametrize('enc,rot,hd,n', [(2, 2, 0, 31), (3, 3, 0, 31), (2, 0, 0, 31), (3, 1, 2, 6)])
def test_low_energy_spectrum_adjacency_encodings_golden(enc, rot, hd, n, tmp_path):
    """Encodings 2 / 3 on the HIP path with the reference's seeded add_noise (test_examples.py test_e03)."""
    import tnac4o_amd
    g = load('g11_spectrum_adjacency.npz')
    s = gpu_solver(rot=rot)
    np.random.seed(100 + enc)
    s.add_noise(amplitude=1e-7)
    s.search_low_energy_spectrum(excitations_encoding=enc, M=1024, relative_P_cutoff=1e-8, Dmax=16, max_dEng=1.0, lim_hd=hd)
    tag = 'L128_i1_e%d_r%d_hd%d' % (enc, rot, hd)
    assert [len(s.d), len(s.el)] == list(g[tag + '_n_shapes'])
    f = str(tmp_path / 'spectrum.npy')
    s.save(f)
    s.decode_low_energy_states(max_dEng=1.0)
    assert len(s.energy) == n
    _spectrum_same(s, g, tag)
    t = tnac4o_amd.load(f)
    t.decode_low_energy_states(max_dEng=1.0)
    assert np.array_equal(t.energy, s.energy) and np.array_equal(t.states, s.states)


@pytest.mark.gpu
def test_low_energy_spectrum_adjacency_rmf_golden():
    import tnac4o_amd
    g = load('g11_spectrum_adjacency.npz')
    J = gi.e05_rmf()
    for enc, rot in ((2, 2), (3, 3)):
        s = tnac4o_amd.tnac4o(mode='RMF', Nx=J['Nx'], Ny=J['Ny'], J=J, beta=4)
        s.rotate_graph(rot)
        np.random.seed(200 + enc)
        s.add_noise(amplitude=1e-7)
        s.search_low_energy_spectrum(excitations_encoding=enc, M=1024, relative_P_cutoff=1e-12, Dmax=32, max_dEng=3.1, lim_hd=0)
        s.decode_low_energy_states(max_dEng=3.1, max_states=100)
        assert len(s.energy) == 26
        _spectrum_same(s, g, 'RMF_e%d_r%d' % (enc, rot), bits=False)


@pytest.mark.gpu
def test_fused_site_steps_bit_identical():
    """The single-call site steps (tn_site_qr, tn_rar, tn_env_mix, tn_apply_truncation: csrc/site.hip) run the same kernels in
    the same order as the separate calls they replace: a whole sweep (all four canonisation passes, both variational sweeps,
    rank-revealing early exits included) must come out bit-identical with and without them."""
    from tnac4o_amd import ops
    kw = dict(graduate_truncation=True, Dmax=32, tolS=1e-16, tolV=1e-10, max_sweeps=20)
    res = []
    saved = ops.FUSED_SITE, ops.PASS1_WEIGHTED
    try:
        ops.PASS1_WEIGHTED = False               # (the weighted first pass exists in the fused form only)
        for fused in (False, True):
            ops.FUSED_SITE = fused
            s = gpu_solver(L=512, rot=1)
            s._setup_rhoT(**kw)
            res.append(s)
    finally:
        ops.FUSED_SITE, ops.PASS1_WEIGHTED = saved
    a, b = res
    assert [m.D for m in a.rhoT] == [m.D for m in b.rhoT]
    assert a.rhoT_discarded == b.rhoT_discarded and a.rhoT_overlap == b.rhoT_overlap
    for x, y in zip(a.rhoT, b.rhoT):
        assert all(torch.equal(p, q) for p, q in zip(x.A, y.A))
        assert all(np.array_equal(p, q) for p, q in zip(x.S, y.S))


@pytest.mark.gpu
def test_bench_sharded_step_rehearsal_four_ranks():
    """bench.py's N > 1 decomposition on the one GPU of the test box (4 processes on cuda:0, exchange over gloo): 2 rotation
    teams x 2 ranks, the owners sweep and broadcast the boundary MPS to their beam partners inside the timed step, then the
    sharded 2-rotation ground-state search with its all-gather; the JSON line must be well formed and the search must find the
    golden ground-state energy of the synthetic-free droplet-size instance it runs on."""
    import json as _json
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [_sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '4', '--master-addr', '127.0.0.1',
           '--master-port', '29641', os.path.join(root, 'bench.py'), '--gpus', '4', '--rehearse-one-gpu', '--nrot', '2', '--L', '128',
           '--chi', '16', '--steps', '1', '--warmup', '1', '--cpu-rows', '0', '--beam-shards', '2']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    d = _json.loads(line)
    assert d['n_gpus'] == 4 and d['scaling'] == 'strong' and d['unit'] == 'ms/sweep' and d['value'] > 0
    assert '2 rotation teams x 2 rank' in d['config']['parallelism']
    assert d['full_solve']['rotations'] == 2 and np.isfinite(d['full_solve']['energy'])
    # the same instance solved in-process: identical energy
    import tnac4o_amd
    from tnac4o_amd.auxx import synthetic_chimera
    s = tnac4o_amd.tnac4o(mode='Ising', Nx=4, Ny=4, Nc=8, J=synthetic_chimera(4, 4, 20260002), beta=3.0)
    s.search_ground_state(M=1024, relative_P_cutoff=1e-8, Dmax=16)
    assert d['full_solve']['energy'] == pytest.approx(float(s.energy[0]), abs=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize('L,chi', [(512, 32), (2048, 64)])
def test_weighted_first_pass_matches_plain_pass(L, chi):
    """The weighted rank-revealing first canonisation pass (MPS.canonise_right_weighted) against the plain one (the
    reference's `canonise_right()`, mps.py:187) on real sweeps: the a-posteriori bound on the relative change of the state
    holds without fallbacks (<= 2^-56), the compressed boundary MPS are the same states (fidelity 1 - 1e-13), overlaps and
    discarded weights agree, and the pass really shrinks the bonds."""
    from tnac4o_amd import ops, mps
    kw = dict(graduate_truncation=True, Dmax=chi, tolS=1e-16, tolV=1e-10, max_sweeps=20)
    bounds, bonds = [], []
    orig = mps.MPS._compress_native

    def spy(self, *a, **k):                 # (the production path: the C++ chain driver reports the pass's bound and bond sums)
        out = orig(self, *a, **k)
        if self.native_info['weighted_used']:
            bounds.append((self.native_info['reveal_fallbacks'] == 0, self.native_info['reveal_error_bound']))
            bonds.append((self.native_info['bonds_before'], self.native_info['bonds_after']))
        return out
    res = []
    saved = ops.PASS1_WEIGHTED, os.environ.get('TN_QR_NBO')
    mps.MPS._compress_native = spy
    # (the comparison is about the FIRST pass: the later stages run the same way in all three runs -- the shortcuts that depend on
    #  what the 4 chi pass met, see test_pass_shortcuts_match_full_passes, would take different branches behind a plain first pass)
    saved_sw = {k: os.environ.get(k) for k in ('TN_VAR_TARGET', 'TN_VAR1_SKIP')}
    os.environ['TN_VAR_TARGET'], os.environ['TN_VAR1_SKIP'] = 'phi', '0'
    try:
        # plain pass twice (two-level and single-level QR blocking: a pure rounding-level change), then the weighted pass
        for weighted, nbo in ((False, '256'), (False, '0'), (True, '256')):
            ops.PASS1_WEIGHTED = weighted
            os.environ['TN_QR_NBO'] = nbo
            # L=512: droplet #1;  L=2048: the synthetic instance bench.py times
            from tnac4o_amd.auxx import synthetic_chimera
            s = gpu_solver(L=L, rot=0) if L == 512 else gpu_solver(L=L, J=synthetic_chimera(16, 16, 20260004))
            s._setup_rhoT(**kw)
            res.append(s)
    finally:
        ops.PASS1_WEIGHTED = saved[0]
        if saved[1] is None:
            os.environ.pop('TN_QR_NBO', None)
        else:
            os.environ['TN_QR_NBO'] = saved[1]
        mps.MPS._compress_native = orig
        for k, v in saved_sw.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    a, a2, b = res
    assert bounds and all(ok and 0.0 <= e <= 2.0 ** -57 for ok, e in bounds)
    print('bond sums before / after the weighted pass:', bonds)
    if L == 2048:
        assert sum(got for _, got in bonds) < 0.75 * sum(full for full, _ in bonds)   # the pass really shrinks the bonds
    np.testing.assert_allclose(np.array(a.rhoT_overlap, dtype=float), np.array(b.rhoT_overlap, dtype=float), rtol=0, atol=1e-12)
    # discarded weights: 1e-6 relative, or 3x what the plain pass itself moves under the rounding-level change of its QR blocking
    # (deep rows of a chi=64 sweep amplify rounding: single rows move by tens of per cent between the two plain runs)
    da, da2, db = (np.array(x.rhoT_discarded, dtype=float) for x in (a, a2, b))
    allow = 1e-14 + 1e-6 * da + 3.0 * np.abs(da2 - da)
    print('discarded weights, weighted vs plain: largest |difference| / allowance = %.3f' % float((np.abs(db - da) / allow).max()))
    assert np.all(np.abs(db - da) <= allow), (da, da2, db)
    worst = 0.0
    for x, x2, y in zip(a.rhoT, a2.rhoT, b.rhoT):
        spread = 1.0 - fidelity(host_chain(x), host_chain(x2))
        got = 1.0 - fidelity(host_chain(x), host_chain(y))
        worst = max(worst, got / (1e-13 + 3.0 * max(spread, 0.0)))
        assert got < 1e-13 + 3.0 * max(spread, 0.0)
    print('fidelity, weighted vs plain: largest (1 - F) / allowance = %.3f' % worst)


@pytest.mark.gpu
@pytest.mark.parametrize('L,chi', [(128, 32), (512, 32), (2048, 64)])
def test_pass_shortcuts_match_full_passes(L, chi, monkeypatch):
    """The three shortcuts of the intermediate stages of compress_mps (csrc/chain.hip) against the reference's full sequence
    (mps.py:187-199) run with the same kernels: (i) no decomposition at a bond of the 4 chi / 2 chi pass where the truncation rule cannot
    truncate (min(C.shape) <= Dmax, tol <= eps: only singular values below eps S0 would go) -- the bond indices that carry nothing are
    dropped by their norms instead (tn_bond_deflate); (ii) a 4 chi pass that truncated nothing leaves phi itself with smaller bonds: that
    copy is the target of the variational sweeps; (iii) the one sweep of the 4 chi stage towards the state itself is not run.  All three
    change the state by O(L eps): the compressed boundary MPS of every row are the same states (fidelity >= 1 - 1e-13), the overlaps agree
    to 1e-12, the bonds up to the decisions that fall on the eps floor, and the discarded weights to 1e-6 relative or the first-order
    bound of what an O(L eps) change of the state does to the tail of its Schmidt spectrum (sqrt(tail) x 1e-14 S0 ~ 2e-13)."""
    from tnac4o_amd.auxx import synthetic_chimera
    kw = dict(graduate_truncation=True, Dmax=chi, tolS=1e-16, tolV=1e-10, max_sweeps=20)
    res = []
    for full in (True, False):
        if full:
            monkeypatch.setenv('TN_GAUGE_SVD', '1'); monkeypatch.setenv('TN_VAR_TARGET', 'phi'); monkeypatch.setenv('TN_VAR1_SKIP', '0')
        else:
            for k in ('TN_GAUGE_SVD', 'TN_VAR_TARGET', 'TN_VAR1_SKIP'):
                monkeypatch.delenv(k, raising=False)
        s = gpu_solver(L=L, rot=0) if L < 2048 else gpu_solver(L=L, J=synthetic_chimera(16, 16, 20260004))
        s._setup_rhoT(**kw)
        res.append(s)
    a, b = res
    info = [m.native_info for m in b.rhoT if m is not None and hasattr(m, 'native_info')]
    if L == 2048:
        assert sum(i['gauge_skipped'] for i in info) > 100 and sum(i['var1_skipped'] for i in info) >= 10        # the shortcuts are what ran
    np.testing.assert_allclose(np.array(a.rhoT_overlap, dtype=float), np.array(b.rhoT_overlap, dtype=float), rtol=0, atol=1e-12)
    da, db = (np.array(x.rhoT_discarded, dtype=float) for x in (a, b))
    allow = 1e-6 * da + 2e-13
    print('discarded weights, shortcuts vs full passes: largest |difference| / allowance = %.3f' % float((np.abs(db - da) / allow).max()))
    assert np.all(np.abs(db - da) <= allow), (da, db)
    worst = 0.0
    for x, y in zip(a.rhoT, b.rhoT):
        if x is None:
            continue
        worst = max(worst, 1.0 - fidelity(host_chain(x), host_chain(y)))
        assert max(abs(p - q) for p, q in zip(x.D, y.D)) <= 2, (x.D, y.D)
    print('1 - fidelity, shortcuts vs full passes: largest %.2e' % worst)
    assert worst < 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize('L,chi,rot,bottom', [(128, 8, 3, False), (128, 32, 0, True), (512, 32, 1, False), (2048, 64, 0, False), (2048, 64, 2, True)])
def test_native_chain_driver_bit_identical_to_python_driver(L, chi, rot, bottom):
    """tn_compress_mps (apply_mpo + compress_mps of a row in ONE library call, walked in C++: csrc/chain.hip) against the Python
    driver that issues the same steps one library call at a time (MPS._compress_python): a whole sweep -- absorption, the weighted
    rank-revealing first pass with its host decisions, all truncating passes, the variational sweeps with lazy / batched Schmidt
    values -- must come out bit-identical: site tensors, bond dimensions, overlaps, discarded weights, Schmidt values."""
    from tnac4o_amd import ops
    from tnac4o_amd.auxx import synthetic_chimera
    kw = dict(graduate_truncation=True, Dmax=chi, tolS=1e-16, tolV=1e-10, max_sweeps=20)
    res = []
    saved = ops.NATIVE_CHAIN
    try:
        for native in (False, True):
            ops.NATIVE_CHAIN = native
            s = gpu_solver(L=L, rot=rot) if L != 2048 else gpu_solver(L=L, J=synthetic_chimera(16, 16, 20260004))
            (s._setup_rhoB if bottom else s._setup_rhoT)(**kw)
            res.append(s)
    finally:
        ops.NATIVE_CHAIN = saved
    a, b = res
    ra, rb = (a.rhoB, b.rhoB) if bottom else (a.rhoT, b.rhoT)
    assert [m.D for m in ra] == [m.D for m in rb]
    if bottom:
        assert a.rhoB_discarded == b.rhoB_discarded and a.rhoB_overlap == b.rhoB_overlap
    else:
        assert a.rhoT_discarded == b.rhoT_discarded and a.rhoT_overlap == b.rhoT_overlap
    for x, y in zip(ra, rb):
        assert all(torch.equal(p, q) for p, q in zip(x.A, y.A))
        assert all(np.array_equal(p, q) for p, q in zip(x.S, y.S))
        assert getattr(x, 'reveal_error_bound', None) == getattr(y, 'reveal_error_bound', None)
    if L == 2048:
        info = rb[8].native_info
        assert info['weighted_used'] and info['reveal_fallbacks'] == 0 and info['arena_peak'] <= info['arena_bytes']
        print('arena: peak %.2f GB of %.2f GB; bonds %d -> %d after the weighted pass' % (info['arena_peak'] / 1e9, info['arena_bytes'] / 1e9,
                                                                                       info['bonds_before'], info['bonds_after']))
