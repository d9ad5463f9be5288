"""Seeded input builders shared by tools/make_golden.py (which feeds them to the reference)
and by the tests (which feed them to the oracle and to the HIP path).  Pure numpy."""
import os
import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
INST_DIR = os.path.join(GOLDEN_DIR, 'instances')

# G1 shapes: plain, rank-deficient and graded-spectrum variants are derived.
G1_SHAPES = [(64, 16), (256, 64), (1024, 64), (37, 91)]


def g1_matrix(shape, kind, seed=7):
    rng = np.random.default_rng(seed + 1000 * shape[0] + shape[1] + {'plain': 0, 'rankdef': 1, 'graded': 2}[kind])
    m, n = shape
    k = min(m, n)
    if kind == 'plain':
        return rng.standard_normal((m, n))
    U, _ = np.linalg.qr(rng.standard_normal((m, k)))
    V, _ = np.linalg.qr(rng.standard_normal((n, k)))
    if kind == 'rankdef':
        s = np.zeros(k)
        s[:max(1, k // 3)] = np.linspace(1.0, 0.1, max(1, k // 3))
    else:
        s = 10.0 ** (-np.arange(k) / 2.0)
    return (U * s) @ V.T


def rand_chain(seed, dims, phys):
    """Site tensors A[n] (D[n], phys[n], D[n+1]) with entries in [-1,1)."""
    rng = np.random.default_rng(seed)
    return [rng.uniform(-1, 1, (dims[n], phys[n], dims[n + 1])) for n in range(len(phys))]


def rand_mpo(seed, L, b, p_out, p_in, span=30.0):
    """Positive MPO tensors (b_l, p_out, b_r, p_in) with a wide dynamic range, open ends."""
    rng = np.random.default_rng(seed)
    out = []
    for n in range(L):
        bl = 1 if n == 0 else b
        br = 1 if n == L - 1 else b
        out.append(np.exp(-span * rng.uniform(0, 1, (bl, p_out, br, p_in))))
    return out


def droplet_J(L=128, instance=1):
    """Couplings of a bundled droplet instance, prepared as examples/e01:57-65 does."""
    path = os.path.join(INST_DIR, 'chimera%d_%03d.txt' % (L, instance))
    rows = np.loadtxt(path)
    J = [[int(r[0]) - 1, int(r[1]) - 1, float(r[2])] for r in rows]
    dJ = 1 / 75
    return [[i, j, round(v / dJ) * dJ] for i, j, v in J]


def j124_J(instance=1):
    path = os.path.join(INST_DIR, 'C8_J124_%03d.txt' % instance)
    rows = np.loadtxt(path)
    return [[int(r[0]) - 1, int(r[1]) - 1, float(r[2])] for r in rows]


def golden_groundstate(L, instance):
    """(energy, bits) from the reference's groundstates_otn2d.txt (copied lines 1-5)."""
    with open(os.path.join(INST_DIR, 'chimera%d_groundstates_1-5.txt' % L)) as f:
        line = f.readlines()[instance - 1].split()
    assert line[0] == '%03d.txt' % instance
    return float(line[2]), np.array([int(x) for x in line[3:]], dtype=np.int8)


def e05_rmf():
    """The model of the reference's minimal RMF example (examples/e05_minimal_RMF.py:33-53, pinned by test_examples.py
    test_e05: 26 states within dE < 3.1): a 3 x 5 grid of 3-state variables, a unit penalty whenever neighbours differ,
    and linear fields -1.5, 0, 1.5 on the outer rows / 1.25, 0, -1.25 on the middle row."""
    Ny, Nx, d = 3, 5, 3
    fun = {1: 1.0 - np.eye(d), 2: np.array([-1.5, 0.0, 1.5]), 3: np.array([1.25, 0.0, -1.25])}
    fac = {}
    for ny in range(Ny):
        for nx in range(Nx):
            fac[(ny, nx)] = 3 if ny == 1 else 2
            if nx + 1 < Nx:
                fac[(ny, nx, ny, nx + 1)] = 1
            if ny + 1 < Ny:
                fac[(ny, nx, ny + 1, nx)] = 1
    return {'fun': fun, 'fac': fac, 'N': np.full((Ny, Nx), d, dtype=int), 'Nx': Nx, 'Ny': Ny}


def minimal_rmf():
    """3x5 RMF with d=3 mirroring the structure of examples/e05 (seeded tables)."""
    rng = np.random.default_rng(5)
    Ny, Nx, d = 3, 5, 3
    fun, fac, k = {}, {}, 0
    for ny in range(Ny):
        for nx in range(Nx):
            fun[k] = rng.uniform(0, 1, d)
            fac[(ny, nx)] = k
            k += 1
    for ny in range(Ny):
        for nx in range(Nx - 1):
            fun[k] = rng.uniform(0, 1, (d, d))
            fac[(ny, nx, ny, nx + 1)] = k
            k += 1
    for ny in range(Ny - 1):
        for nx in range(Nx):
            fun[k] = rng.uniform(0, 1, (d, d))
            fac[(ny, nx, ny + 1, nx)] = k
            k += 1
    return {'fun': fun, 'fac': fac, 'N': np.full((Ny, Nx), d, dtype=int), 'Nx': Nx, 'Ny': Ny}
