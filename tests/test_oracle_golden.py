"""Pin the CPU oracle against vectors captured from the reference (tools/make_golden.py) and
against the reference's own golden data (tests/golden/instances/).  CPU only."""
import json
import os

import numpy as np
import pytest

import golden_inputs as gi
from oracle import mps_ref as mr
from oracle import solver_ref as sr


def load(name):
    return np.load(os.path.join(gi.GOLDEN_DIR, name))


def solver(L=128, ins=1, rot=0, beta=3.0, pre=False):
    n = {128: 4, 512: 8, 2048: 16}[L]
    s = sr.RefSolver(mode='Ising', Nx=n, Ny=n, Nc=8, J=gi.droplet_J(L, ins), beta=beta)
    if rot:
        s.rotate_graph(rot)
    if pre:
        s.precondition()
    return s


# ---------------------------------------------------------------- G1: gauge / truncation rules
@pytest.mark.parametrize('shape', gi.G1_SHAPES)
@pytest.mark.parametrize('kind', ['plain', 'rankdef', 'graded'])
def test_g1_linalg(shape, kind):
    g = load('g1_linalg.npz')
    T = gi.g1_matrix(shape, kind)
    tag = '%dx%d_%s' % (shape[0], shape[1], kind)
    U, S, V = mr.svd_gauged(T.copy())
    np.testing.assert_allclose(S, g[tag + '_S'], rtol=0, atol=1e-13 * S[0])
    assert np.abs((U * S) @ V - T).max() < 1e-12
    flip = (np.abs(U.min(0)) > U.max(0)) & (np.abs(V.min(1)) > V.max(1))
    assert not flip.any()                                   # gauge is idempotent
    Q, R = mr.qr_pos(T.copy())
    assert (np.diag(R) >= 0).all()
    np.testing.assert_allclose(np.abs(np.diag(R)), g[tag + '_absdiagR'], rtol=0, atol=1e-12 * np.abs(R).max())
    np.testing.assert_allclose(mr.svdvals(T), g[tag + '_svdS'], rtol=0, atol=1e-13 * S[0])
    assert mr.pow2_floor_max(T) == g[tag + '_nfactor'][0]
    for Dmax, tol in ((8, 1e-16), (10 ** 6, 1e-16), (10 ** 6, 1e-3)):
        _, C, _, keep, disc = mr.truncate_center(T.copy(), Dmax, tol)
        want = g[tag + '_trunc_%d_%g' % (Dmax, tol)]
        if kind != 'rankdef' or tol > 1e-10 or Dmax == 8:   # eps-level rank decisions are noise
            assert keep == int(want[0])
            assert disc == pytest.approx(want[1], rel=1e-6, abs=1e-14)


def test_g1_nfactor_probe():
    g = load('g1_linalg.npz')
    for x, y in zip(g['nfactor_probe_in'], g['nfactor_probe_out']):
        assert mr.pow2_floor_max(np.array([x])) == y


# ---------------------------------------------------------------- G2: absorption index order
def test_g2_absorb():
    g = load('g2_absorb.npz')
    tags = sorted({k[:-2] for k in g.files if k.endswith('_A')})
    assert len(tags) == 4
    for tag in tags:
        hconj = bool(int(tag.split('_')[-1]))
        T = mr.absorb_site(g[tag + '_A'], g[tag + '_W'], hconj)
        assert tuple(T.shape) == tuple(g[tag + '_shape'])
        if tag + '_T' in g.files:
            np.testing.assert_allclose(T, g[tag + '_T'], rtol=0, atol=1e-14)
        else:
            np.testing.assert_allclose(T[::7, ::3, ::5], g[tag + '_Tsub'], rtol=0, atol=1e-13)
            np.testing.assert_allclose([T.sum(), np.abs(T).sum()], g[tag + '_Tsum'], rtol=1e-12)


# ---------------------------------------------------------------- G3: compress_mps
def chain_from(As):
    psi = mr.RefMPS(d=[a.shape[1] for a in As], L=len(As), Dmax=1, canonise=None)
    psi.A = [a.copy() for a in As]
    psi.D = [As[0].shape[0]] + [a.shape[2] for a in As]
    return psi


@pytest.mark.parametrize('case', [0, 1])
def test_g3_compress(case):
    g = load('g3_compress.npz')
    L, D, p, b, chi = [(6, 6, 4, 4, 8), (8, 8, 16, 16, 16)][case]
    As = gi.rand_chain(31 + case, [1] + [D] * (L - 1) + [1], [p] * L)
    Ws = gi.rand_mpo(41 + case, L, b, p, p)
    for hconj in (True, False):
        for grad in (True, False):
            psi = chain_from(As)
            mpo = mr.RefMPO(L)
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
            if case == 0:      # gauge-invariant: overlap with the reference's stored tensors
                phi = chain_from([g[tag + '_A%d' % n] for n in range(L)])
                o = mr.mps_dot(phi, psi) / np.sqrt(mr.mps_dot(phi, phi) * mr.mps_dot(psi, psi))
                assert abs(o) > 1 - 1e-12


# ---------------------------------------------------------------- G4: PEPS tensors / MPO sites
def test_g4_peps():
    g = load('g4_peps.npz')
    for pre in (False, True):
        s = solver(pre=pre)
        for (ny, nx) in [(0, 0), (1, 1), (3, 3), (0, 3), (2, 0)]:
            tag = 'L128_pre%d_%d_%d' % (int(pre), ny, nx)
            T = s.peps_dense(ny, nx)
            assert tuple(T.shape) == tuple(g[tag + '_shape'])
            W = s.mpo_site(ny, nx)
            np.testing.assert_allclose(W, T.sum(0), rtol=1e-15)
            if tag + '_W' in g.files:
                np.testing.assert_allclose(W, g[tag + '_W'], rtol=1e-13)
            np.testing.assert_allclose([W.sum(), (W ** 2).sum(), W.max(), W[W > 0].min()], g[tag + '_Wsum'], rtol=1e-12)
            np.testing.assert_allclose(W.reshape(-1)[::997], g[tag + '_probe'], rtol=1e-13)
        if pre:
            np.testing.assert_allclose(s.Xu, g['L128_pre1_Xu'], rtol=1e-12)
            np.testing.assert_allclose(s.Xd, g['L128_pre1_Xd'], rtol=1e-12)
            np.testing.assert_allclose(s.overlaps_ud, g['L128_pre1_overlaps_ud'], rtol=1e-10)


def test_g4_rmf():
    g = load('g4_peps.npz')
    J = gi.minimal_rmf()
    s = sr.RefSolver(mode='RMF', Nx=J['Nx'], Ny=J['Ny'], J=J, beta=2.0)
    for (ny, nx) in [(0, 0), (1, 2), (2, 4)]:
        np.testing.assert_allclose(s.peps_dense(ny, nx), g['rmf_%d_%d_T' % (ny, nx)], rtol=1e-14)


# ---------------------------------------------------------------- G5: sweeps
@pytest.mark.parametrize('rot', [0, 1, 2, 3])
@pytest.mark.parametrize('chi', [8, 32])
def test_g5_sweep_L128(rot, chi):
    g = load('g5_sweep.npz')
    s = solver(rot=rot)
    s._setup_rhoT(graduate_truncation=True, Dmax=chi, tolS=1e-16, tolV=1e-10, max_sweeps=20)
    tag = 'L128_r%d_chi%d' % (rot, chi)
    np.testing.assert_allclose(np.array(s.rhoT_overlap, dtype=float), g[tag + '_overlap'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.array(s.rhoT_discarded, dtype=float), g[tag + '_discarded'], rtol=1e-6, atol=1e-14)
    assert np.array_equal(np.array([m.D for m in s.rhoT]), g[tag + '_D'])
    if chi == 8:
        for ny in range(s.Ny + 1):
            phi = chain_from([g[tag + '_A_%d_%d' % (ny, nx)] for nx in range(s.Nx)])
            psi = s.rhoT[ny]
            o = mr.mps_dot(phi, psi) / np.sqrt(mr.mps_dot(phi, phi) * mr.mps_dot(psi, psi))
            assert abs(o) > 1 - 1e-12


# ---------------------------------------------------------------- G6: conditional probabilities
@pytest.mark.parametrize('rot,chi', [(0, 8), (3, 8), (0, 32)])
def test_g6_marginals(rot, chi):
    g = load('g6_pn.npz')
    tag = 'L128_r%d_chi%d' % (rot, chi)
    trace = []
    s = solver(rot=rot)
    s.search_ground_state(M=1024, relative_P_cutoff=1e-8, Dmax=chi, trace=trace)
    assert [t[2].shape[0] for t in trace] == list(g[tag + '_nbranch'])
    for k in g[tag + '_steps']:
        st = int(g[tag + '_stride%d' % k][0])
        np.testing.assert_allclose(trace[k][2][::st], g[tag + '_P%d' % k], rtol=1e-10, atol=1e-300)
        np.testing.assert_allclose(trace[k][3][::st], g[tag + '_min%d' % k], rtol=1e-6, atol=1e-18)
    assert s.negative_probability == pytest.approx(g[tag + '_neg'][0], rel=1e-6, abs=1e-18)


# ---------------------------------------------------------------- G7: end results
def g7():
    with open(os.path.join(gi.GOLDEN_DIR, 'g7_search.json')) as f:
        return json.load(f)


def check_result(s, want):
    assert s.energy[0] == pytest.approx(want['energy'], abs=1e-10)
    assert int(s.degeneracy) == want['degeneracy']
    assert s.probability[0] == pytest.approx(want['probability'], abs=1e-9)
    assert s.discarded_probability == pytest.approx(want['discarded_probability'], abs=1e-9)
    assert s.negative_probability == pytest.approx(want['negative_probability'], rel=1e-6, abs=1e-16)
    assert len(s.energy) == want['n_states']
    assert [int(x) for x in s.states[0]] == want['state0']
    assert [int(x) for x in s.binary_states()[0]] == want['bits0']


G7_FAST = [(128, 1, 0, 8, True), (128, 1, 3, 8, False), (128, 2, 0, 32, False), (128, 3, 2, 32, True),
           (128, 2, 1, 8, False), (128, 3, 2, 8, False)]


@pytest.mark.parametrize('L,ins,rot,chi,pre', G7_FAST)
def test_g7_search(L, ins, rot, chi, pre):
    want = g7()['L%d_i%d_r%d_chi%d_pre%d' % (L, ins, rot, chi, int(pre))]
    s = solver(L=L, ins=ins, rot=rot, pre=pre)
    s.search_ground_state(M=1024, relative_P_cutoff=1e-8, Dmax=chi)
    check_result(s, want)
    # the reference's own golden file: energy and full bit string
    E, bits = gi.golden_groundstate(L, ins)
    assert s.energy[0] == pytest.approx(E, abs=1e-5)        # file has 6 digits (test_examples.py:27-33)
    assert np.array_equal(s.binary_states()[0], bits)
    # independent energy recomputation (auxx.energy_Jij)
    assert sr.energy_Jij(gi.droplet_J(L, ins), s.binary_states()[:1])[0] == pytest.approx(s.energy[0], abs=1e-9)


def test_g7_rmf():
    J = gi.minimal_rmf()
    for rot in (0, 1):
        s = sr.RefSolver(mode='RMF', Nx=J['Nx'], Ny=J['Ny'], J=J, beta=2.0)
        if rot:
            s.rotate_graph(rot)
        s.search_ground_state(M=64, relative_P_cutoff=1e-8, Dmax=8)
        check_result(s, g7()['RMF_r%d' % rot])


@pytest.mark.slow
def test_g7_L512():
    want = g7()['L512_i1_r0_chi32_pre0']
    s = solver(L=512)
    s.search_ground_state(M=1024, relative_P_cutoff=1e-8, Dmax=32)
    check_result(s, want)
    E, bits = gi.golden_groundstate(512, 1)
    assert s.energy[0] == pytest.approx(E, abs=1e-5)
    assert np.array_equal(s.binary_states()[0], bits)


# ---------------------------------------------------------------- G8: Gibbs sampling
@pytest.mark.parametrize('rot,chi,M,seed', [(0, 16, 64, 1234), (1, 8, 32, 99)])
def test_g8_gibbs_sampling(rot, chi, M, seed):
    g = load('g8_gibbs.npz')
    tag = 'r%d_chi%d_M%d_seed%d' % (rot, chi, M, seed)
    s = solver(rot=rot)
    np.random.seed(seed)
    E = s.gibbs_sampling(M=M, Dmax=chi)
    assert np.array_equal(np.asarray(s.states), g[tag + '_states'])          # same draws, same configurations
    np.testing.assert_allclose(E, g[tag + '_energy'], rtol=0, atol=1e-10)
    assert np.array_equal(s.binary_states(), g[tag + '_bits'])
    assert s.negative_probability == pytest.approx(float(g[tag + '_neg'][0]), abs=1e-12)
    # the reference's own consistency check (examples/test_examples.py:36-56): energies recomputed from the bit strings
    assert np.abs(sr.energy_Jij(gi.droplet_J(128, 1), s.binary_states()) - E).max() < 1e-6


# ---------------------------------------------------------------- dgebal restatement (preconditioner, tnac4o.py:1845)
def test_gebal_restatement_matches_scipy():
    """oracle/gebal_ref.py (LAPACK dgebal, job 'S') against scipy.linalg.matrix_balance(permute=False, separate=True), the
    call the reference makes: identical power-of-two scale factors on random, wide-range, graded and sparse matrices."""
    import scipy.linalg
    from oracle.gebal_ref import gebal_scale
    rng = np.random.default_rng(1)
    for t in range(400):
        n = int(rng.integers(1, 33))
        kind = t % 4
        if kind == 0:
            A = rng.standard_normal((n, n))
        elif kind == 1:
            A = np.exp(rng.uniform(-40, 40, (n, n)))
        elif kind == 2:
            d = np.exp(rng.uniform(-30, 30, n))
            A = np.abs(rng.standard_normal((n, n))) * d[:, None] / d[None, :]
        else:
            A = np.exp(rng.uniform(-20, 0, (n, n)))
            A[rng.random((n, n)) < 0.3] = 0
        _, sc = scipy.linalg.matrix_balance(A, permute=False, separate=True)
        assert np.array_equal(sc[0], gebal_scale(A)[0])
