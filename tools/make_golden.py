#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the reference read-only.

Runs ONLY in the authoring container (needs /root/reference); neither the tests nor the GPU
box ever import the reference.  Inputs come from tests/golden_inputs.py (seeded) or from the
reference's own instance files copied under tests/golden/instances/.  Outputs are data only
(G8 = Gibbs sampling, §8f-2).  Usage:  python tools/make_golden.py [g1 g2 ... g8]
"""
import json
import os
import sys
import time

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, os.path.join(REPO, 'tests'))

import numpy as np                      # noqa: E402
import tnac4o as ref                    # noqa: E402
from tnac4o import mps as rmps          # noqa: E402
import golden_inputs as gi              # noqa: E402

OUT = gi.GOLDEN_DIR


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name), **arrs)
    print('wrote', name, sum(np.asarray(v).nbytes for v in arrs.values()) // 1024, 'KiB raw')


def chain_from(As):
    """A reference MPS object holding the given site tensors (no canonisation)."""
    L = len(As)
    psi = rmps.MPS(d=[a.shape[1] for a in As], L=L, Dmax=1, initial='X', canonise=None)
    psi.A = [a.copy() for a in As]
    psi.D = [As[0].shape[0]] + [a.shape[2] for a in As]
    return psi


def g1():
    out = {}
    dummy = rmps.MPS(d=2, L=2)
    for shape in gi.G1_SHAPES:
        for kind in ('plain', 'rankdef', 'graded'):
            T = gi.g1_matrix(shape, kind)
            tag = '%dx%d_%s' % (shape[0], shape[1], kind)
            U, S, V = rmps.svd(T.copy())
            out[tag + '_S'] = S
            out[tag + '_svd_res'] = np.array([np.abs((U * S) @ V - T).max()])
            Q, R = rmps.qr(T.copy())
            out[tag + '_absdiagR'] = np.abs(np.diag(R))
            out[tag + '_svdS'] = rmps.svd_S(T.copy())
            out[tag + '_nfactor'] = np.array([rmps.nfactor(T)])
            for Dmax, tol in ((8, 1e-16), (10 ** 6, 1e-16), (10 ** 6, 1e-3)):
                pL, C, pR, keep, disc = dummy._mps_truncateC(T.copy(), Dmax, tol)
                out[tag + '_trunc_%d_%g' % (Dmax, tol)] = np.array([keep, disc])
    out['nfactor_probe_in'] = np.array([1.0, 1.5, 2.0, 3.999, 1e-61, 7.3e17, 0.75, 2.0 ** -1022])
    out['nfactor_probe_out'] = np.array([rmps.nfactor(np.array([x])) for x in out['nfactor_probe_in']])
    save('g1_linalg.npz', **out)


def g2():
    out = {}
    dummy = rmps.MPS(d=2, L=2)
    for (Dl, p, Dr, a, b) in [(3, 4, 5, 2, 3), (8, 16, 8, 16, 16)]:
        rng = np.random.default_rng(Dl * 100 + a)
        A = rng.standard_normal((Dl, p, Dr))
        for hconj in (True, False):
            pin, pout = (6, p) if hconj else (p, 6)      # distinct in/out dims catch a swapped leg
            W = rng.standard_normal((a, pout, b, pin))
            T = dummy._mps_HA(A, W, hconj)[0]
            tag = '%d_%d_%d_%d_%d_%d' % (Dl, p, Dr, a, b, int(hconj))
            out[tag + '_A'], out[tag + '_W'] = A, W
            if T.size <= 20000:
                out[tag + '_T'] = T
            else:
                out[tag + '_Tsub'] = T[::7, ::3, ::5].copy()
                out[tag + '_Tsum'] = np.array([T.sum(), np.abs(T).sum()])
            out[tag + '_shape'] = np.array(T.shape)
    save('g2_absorb.npz', **out)


def compress_record(psi, out, tag, keepA):
    out[tag + '_D'] = np.array(psi.D)
    out[tag + '_discarded'] = np.array(psi.discarded, dtype=float)
    for n, S in enumerate(psi.S):
        out[tag + '_S%d' % n] = np.asarray(S)
    out[tag + '_normC'] = np.array([psi.normC])
    if keepA:
        for n, A in enumerate(psi.A):
            out[tag + '_A%d' % n] = A


def g3():
    out = {}
    # (a) seeded random chain absorbed by a seeded wide-range MPO
    for case, (L, D, p, b, chi) in enumerate([(6, 6, 4, 4, 8), (8, 8, 16, 16, 16)]):
        dims = [1] + [D] * (L - 1) + [1]
        As = gi.rand_chain(31 + case, dims, [p] * L)
        Ws = gi.rand_mpo(41 + case, L, b, p, p)
        for hconj in (True, False):
            for grad in (True, False):
                psi = chain_from(As)
                mpo = rmps.MPO(L=L)
                for n in range(L):
                    mpo.set_direct(Ws[n], n)
                psi.apply_mpo(mpo, Hconj=hconj)
                ov = psi.compress_mps(Dmax=chi, tolS=1e-16, tolV=1e-10, max_sweeps=20, graduate_truncation=grad)
                tag = 'rand%d_h%d_g%d' % (case, int(hconj), int(grad))
                out[tag + '_overlap'] = np.array([ov])
                compress_record(psi, out, tag, keepA=(case == 0))
    save('g3_compress.npz', **out)


def solver(L=128, ins=1, rot=0, beta=3.0, pre=False):
    n = {128: 4, 512: 8, 2048: 16}[L]
    s = ref.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=gi.droplet_J(L, ins), beta=beta)
    if rot:
        s.rotate_graph(rot)
    if pre:
        s.precondition(mode='balancing')
    return s


def g4():
    out = {}
    for pre in (False, True):
        s = solver(pre=pre)
        for (ny, nx) in [(0, 0), (1, 1), (3, 3), (0, 3), (2, 0)]:
            T = s._peps_tensor(ny, nx)
            W = T.sum(0)
            tag = 'L128_pre%d_%d_%d' % (int(pre), ny, nx)
            out[tag + '_shape'] = np.array(T.shape)
            if (ny, nx) in [(1, 1), (3, 3)] or not pre and (ny, nx) == (0, 0):
                out[tag + '_W'] = W
            out[tag + '_Wsum'] = np.array([W.sum(), (W ** 2).sum(), W.max(), W[W > 0].min()])
            out[tag + '_probe'] = W.reshape(-1)[::997].copy()
        if pre:
            out['L128_pre1_Xu'], out['L128_pre1_Xd'] = s.Xu, s.Xd
            out['L128_pre1_overlaps_ud'] = s.overlaps_ud
    # RMF example
    J = gi.minimal_rmf()
    s = ref.tnac4o(mode='RMF', Nx=J['Nx'], Ny=J['Ny'], J=J, beta=2.0)
    for (ny, nx) in [(0, 0), (1, 2), (2, 4)]:
        out['rmf_%d_%d_T' % (ny, nx)] = s._peps_tensor(ny, nx)
    save('g4_peps.npz', **out)


def g5():
    out = {}
    cases = [(128, rot, chi) for rot in range(4) for chi in (8, 32)] + [(512, 0, 32)]
    for (L, rot, chi) in cases:
        t = time.time()
        s = solver(L=L, rot=rot)
        s._setup_rhoT(graduate_truncation=True, Dmax=chi, tolS=1e-16, tolV=1e-10, max_sweeps=20)
        tag = 'L%d_r%d_chi%d' % (L, rot, chi)
        out[tag + '_overlap'] = np.array(s.rhoT_overlap, dtype=float)
        out[tag + '_discarded'] = np.array(s.rhoT_discarded, dtype=float)
        out[tag + '_D'] = np.array([m.D for m in s.rhoT])
        if L == 128 and chi == 8:
            for ny in range(s.Ny + 1):
                for nx in range(s.Nx):
                    out[tag + '_A_%d_%d' % (ny, nx)] = s.rhoT[ny].A[nx]
        print(tag, '%.1fs' % (time.time() - t))
    save('g5_sweep.npz', **out)


def run_search(s, M, P, chi, capture=None):
    if capture is not None:
        rec, cur = capture, []
        orig_pn, orig_eng = s._calculate_Pn, s._update_Eng

        def pn(*a):
            r = orig_pn(*a)
            cur.append((r[0].copy(), r[1]))
            return r

        def eng(*a):
            rec.append((np.array([c[0] for c in cur]), np.array([c[1] for c in cur], dtype=float)))
            cur.clear()
            return orig_eng(*a)
        s._calculate_Pn, s._update_Eng = pn, eng
    s.search_ground_state(M=M, relative_P_cutoff=P, Dmax=chi)
    return s


def g6():
    out = {}
    for (rot, chi) in [(0, 8), (3, 8), (0, 32)]:
        rec = []
        s = run_search(solver(rot=rot), 1024, 1e-8, chi, capture=rec)
        tag = 'L128_r%d_chi%d' % (rot, chi)
        steps = [0, 1, 2, 3, 4, 5, 7, 10, 13]
        worst = int(np.argmin([r[1].min() for r in rec]))
        if worst not in steps:
            steps.append(worst)
        out[tag + '_steps'] = np.array(steps)
        for k in steps:                                  # at most ~64 branches per step (small fixture)
            stride = max(1, rec[k][0].shape[0] // 64)
            out[tag + '_stride%d' % k] = np.array([stride])
            out[tag + '_P%d' % k] = rec[k][0][::stride]
            out[tag + '_min%d' % k] = rec[k][1][::stride]
        out[tag + '_nbranch'] = np.array([r[0].shape[0] for r in rec])
        out[tag + '_neg'] = np.array([s.negative_probability])
    save('g6_pn.npz', **out)


def result_record(s):
    return dict(energy=float(s.energy[0]), degeneracy=int(s.degeneracy), probability=float(s.probability[0]),
                discarded_probability=float(s.discarded_probability),
                negative_probability=float(s.negative_probability), rotation=int(s.rotation),
                n_states=int(len(s.energy)), state0=[int(x) for x in s.states[0]],
                bits0=[int(x) for x in s.binary_states()[0]],
                rhoT_discarded=[float(x) for x in s.rhoT_discarded],
                rhoT_overlap=[float(x) for x in s.rhoT_overlap])


def g7():
    res = {}
    cases = []
    for ins in (1, 2, 3):
        for rot in range(4):
            cases.append((128, ins, rot, 8, False))
        cases.append((128, ins, 0, 8, True))
        cases.append((128, ins, 0, 32, False))
        cases.append((128, ins, 2, 32, True))
    cases.append((512, 1, 0, 32, False))
    for (L, ins, rot, chi, pre) in cases:
        t = time.time()
        s = run_search(solver(L=L, ins=ins, rot=rot, pre=pre), 1024, 1e-8, chi)
        key = 'L%d_i%d_r%d_chi%d_pre%d' % (L, ins, rot, chi, int(pre))
        res[key] = result_record(s)
        print(key, res[key]['energy'], '%.1fs' % (time.time() - t))
        with open(os.path.join(OUT, 'g7_search.json'), 'w') as f:
            json.dump(res, f, indent=0)
    # J124 C8 #1 (examples/e06, test_examples.py:139-147)
    t = time.time()
    s = ref.tnac4o(mode='Ising', Nx=8, Ny=8, Nc=8, J=gi.j124_J(1), beta=0.75)
    s.precondition(mode='balancing')
    s.search_ground_state(M=2 ** 12, relative_P_cutoff=1e-8, Dmax=8)
    res['J124_C8_i1_r0_chi8_pre1'] = result_record(s)
    print('J124', s.energy[0], s.degeneracy, '%.1fs' % (time.time() - t))
    # RMF minimal
    J = gi.minimal_rmf()
    for rot in (0, 1):
        s = ref.tnac4o(mode='RMF', Nx=J['Nx'], Ny=J['Ny'], J=J, beta=2.0)
        if rot:
            s.rotate_graph(rot)
        s.search_ground_state(M=64, relative_P_cutoff=1e-8, Dmax=8)
        res['RMF_r%d' % rot] = result_record(s)
    with open(os.path.join(OUT, 'g7_search.json'), 'w') as f:
        json.dump(res, f, indent=0)


def g8():
    """Gibbs sampling (tnac4o.py:553-650, examples/e02): seeded numpy global generator, droplet L=128 #1."""
    out = {}
    for rot, chi, M, seed in ((0, 16, 64, 1234), (1, 8, 32, 99)):
        s = solver(128, 1, rot)
        np.random.seed(seed)
        E = s.gibbs_sampling(M=M, Dmax=chi)
        tag = 'r%d_chi%d_M%d_seed%d' % (rot, chi, M, seed)
        out[tag + '_energy'] = np.asarray(E)
        out[tag + '_states'] = np.asarray(s.states).astype(np.int16)
        out[tag + '_bits'] = np.asarray(s.binary_states()).astype(np.int8)
        out[tag + '_neg'] = np.array([s.negative_probability])
    save('g8_gibbs.npz', **out)


def g9():
    """A result file written by the reference's own `save` (tnac4o.py:200-231) -- data for the I/O compatibility test."""
    s = solver(128, 1, 1)
    s.search_ground_state(M=1024, relative_P_cutoff=1e-8, Dmax=8)
    s.save(os.path.join(OUT, 'g9_saved_by_reference.npy'))
    print('saved', s.energy[0], len(s.energy))


def g10():
    """Low-energy spectrum, encoding 1 (tnac4o.py:652-915 + decode :1360-1389; examples e03 / e05, test_examples.py:59-138)."""
    import pickle
    out = {}
    for rot, chi in ((0, 16), (1, 16), (3, 8)):
        s = solver(128, 1, rot)
        s.search_low_energy_spectrum(excitations_encoding=1, M=1024, relative_P_cutoff=1e-8, Dmax=chi, max_dEng=1.0, lim_hd=0)
        tag = 'L128_i1_r%d_chi%d' % (rot, chi)
        out[tag + '_n_shapes'] = np.array([len(s.d), len(s.el)])
        s.decode_low_energy_states(max_dEng=1.0)
        out[tag + '_energy'] = np.asarray(s.energy)
        out[tag + '_states'] = np.asarray(s.states).astype(np.int16)
        out[tag + '_bits'] = np.asarray(s.binary_states()).astype(np.int8)
        print(tag, len(s.energy))
    # a second instance with a Hamming-distance limit
    s = solver(128, 2, 0)
    s.search_low_energy_spectrum(excitations_encoding=1, M=1024, relative_P_cutoff=1e-8, Dmax=16, max_dEng=0.8, lim_hd=3)
    s.decode_low_energy_states(max_dEng=0.8, max_states=20)
    out['L128_i2_r0_chi16_hd3_energy'] = np.asarray(s.energy)
    out['L128_i2_r0_chi16_hd3_bits'] = np.asarray(s.binary_states()).astype(np.int8)
    print('hd3', len(s.energy))
    # RMF minimal example (examples/e05_minimal_RMF.py)
    J = gi.e05_rmf()
    for rot in (0, 1):
        s = ref.tnac4o(mode='RMF', Nx=J['Nx'], Ny=J['Ny'], J=J, beta=4)
        if rot:
            s.rotate_graph(rot)
        s.search_low_energy_spectrum(excitations_encoding=1, M=1024, relative_P_cutoff=1e-12, Dmax=32, max_dEng=3.1, lim_hd=0)
        s.decode_low_energy_states(max_dEng=3.1, max_states=100)
        out['RMF_r%d_energy' % rot] = np.asarray(s.energy)
        out['RMF_r%d_states' % rot] = np.asarray(s.states).astype(np.int16)
        print('RMF', rot, len(s.energy))
    save('g10_spectrum.npz', **out)


def g11():
    """Low-energy spectrum with the adjacency-based encodings 2 and 3 (tnac4o.py:943-1358), as in test_examples.py test_e03 /
    test_e05: seeded add_noise(1e-7), droplet L=128 #1 (31 states) and the minimal RMF model (26 states)."""
    np.int = int                     # the reference's _exc_merge still spells the removed alias (tnac4o.py:2214-2215)
    out = {}
    for enc, rot, hd in ((2, 2, 0), (3, 3, 0), (2, 0, 0), (3, 1, 2)):
        s = solver(128, 1, rot)
        np.random.seed(100 + enc)
        s.add_noise(amplitude=1e-7)
        s.search_low_energy_spectrum(excitations_encoding=enc, M=1024, relative_P_cutoff=1e-8, Dmax=16, max_dEng=1.0, lim_hd=hd)
        tag = 'L128_i1_e%d_r%d_hd%d' % (enc, rot, hd)
        out[tag + '_n_shapes'] = np.array([len(s.d), len(s.el)])
        if enc == 2 and rot == 2:
            s.save(os.path.join(OUT, 'g11_saved_by_reference_e2.npy'))
        s.decode_low_energy_states(max_dEng=1.0)
        out[tag + '_energy'] = np.asarray(s.energy)
        out[tag + '_bits'] = np.asarray(s.binary_states()).astype(np.int8)
        print(tag, len(s.energy))
    J = gi.e05_rmf()
    for enc, rot in ((2, 2), (3, 3)):
        s = ref.tnac4o(mode='RMF', Nx=J['Nx'], Ny=J['Ny'], J=J, beta=4)
        s.rotate_graph(rot)
        np.random.seed(200 + enc)
        s.add_noise(amplitude=1e-7)
        s.search_low_energy_spectrum(excitations_encoding=enc, M=1024, relative_P_cutoff=1e-12, Dmax=32, max_dEng=3.1, lim_hd=0)
        s.decode_low_energy_states(max_dEng=3.1, max_states=100)
        out['RMF_e%d_r%d_energy' % (enc, rot)] = np.asarray(s.energy)
        out['RMF_e%d_r%d_states' % (enc, rot)] = np.asarray(s.states).astype(np.int16)
        print('RMF', enc, rot, len(s.energy))
    save('g11_spectrum_adjacency.npz', **out)


if __name__ == '__main__':
    todo = sys.argv[1:] or ['g1', 'g2', 'g3', 'g4', 'g5', 'g6', 'g7', 'g8', 'g9', 'g10', 'g11']
    for name in todo:
        t = time.time()
        globals()[name]()
        print(name, 'done in %.1fs' % (time.time() - t))
