"""Coupling I/O, energy checkers (call surface of the reference's tnac4o/auxx.py) and the seeded synthetic
workload generators used by bench.py.  Host-side, numpy only."""
import numpy as np


def load_Jij(file_name):
    """Read `i j Jij` lines (auxx.py:26-38)."""
    return [[int(r[0]), int(r[1]), float(r[2])] for r in np.loadtxt(file_name)]


def round_Jij(J, dJ):
    """Round couplings to multiples of dJ (auxx.py:41-52)."""
    dJ = float(dJ)
    return [[r[0], r[1], round(r[2] / dJ) * dJ] for r in J]


def minus_Jij(J):
    """auxx.py:55-65."""
    return [[r[0], r[1], -r[2]] for r in J]


def Jij_f2p(J):
    """1-based -> 0-based spin indices (auxx.py:68-81)."""
    return [[r[0] - 1, r[1] - 1, r[2]] for r in J]


def _dense_upper(J, L):
    Jd = np.zeros((L, L))
    for i, j, v in J:
        a, b = (i, j) if i <= j else (j, i)
        Jd[a, b] += v
    return Jd


def energy_Jij(J, states):
    """Ising energies of 0/1 bit strings (auxx.py:84-109)."""
    st = 2.0 * np.asarray(states, dtype=float) - 1.0
    Jd = _dense_upper(J, st.shape[1])
    return np.sum((st @ np.triu(Jd, 1)) * st, 1) + st @ Jd.diagonal()


def energy_RMF(J, states):
    """Cost function of RMF configurations (auxx.py:112-135)."""
    states = np.asarray(states)
    E = np.zeros(len(states))
    for key, val in J['fac'].items():
        if len(key) == 2:
            E += J['fun'][val][states[:, key[0] * J['Nx'] + key[1]]]
        else:
            E += J['fun'][val][states[:, key[0] * J['Nx'] + key[1]], states[:, key[2] * J['Nx'] + key[3]]]
    return E


def save_states_txt(file_name, energies, bit_strings):
    """States as text, one per line: energy then the bit string (1 = spin up, 0 = spin down) -- the format of the
    reference's sampling / search scripts (examples/e02_sample_droplet_instances.py:125-134)."""
    bit_strings = np.asarray(bit_strings)
    L = bit_strings.shape[1]
    with open(file_name, 'w') as f:
        print("# One line per state; First column is the energy, the rest is a state; \
                1 = spin up = si=+1; 0 = spin down = si=-1", file=f)
        for E, st in zip(energies, bit_strings):
            row = np.zeros(L + 1)
            row[0], row[1:] = E, st
            np.savetxt(f, row.reshape(1, L + 1), fmt=' '.join(['%4.6f'] + ['%i'] * L), delimiter=' ')


def load_states_txt(file_name):
    """Inverse of save_states_txt: (energies, bit strings)."""
    a = np.atleast_2d(np.loadtxt(file_name, comments='#'))
    return a[:, 0], a[:, 1:].astype(np.int8)


def synthetic_chimera(Nx, Ny, seed):
    """Seeded chimera couplings with the droplet instances' topology: per cell 8 fields,
    K4,4 between spins {0..3} and {4..7}, spins 0-3 couple downwards, 4-7 to the right; values are
    multiples of 1/75 in [-1,1], zero re-drawn.  Spin index i = (ny*Nx + nx)*8 + m."""
    rng = np.random.default_rng(seed)

    def draw():
        while True:
            v = round(rng.uniform(-1, 1) * 75) / 75
            if v != 0:
                return v
    J = []
    L = Nx * Ny * 8
    for i in range(L):
        J.append([i, i, draw()])
    for c in range(Nx * Ny):
        for mv in range(4):
            for mh in range(4, 8):
                J.append([c * 8 + mv, c * 8 + mh, draw()])
    for ny in range(Ny - 1):
        for nx in range(Nx):
            for m in range(4):
                J.append([(ny * Nx + nx) * 8 + m, ((ny + 1) * Nx + nx) * 8 + m, draw()])
    for ny in range(Ny):
        for nx in range(Nx - 1):
            for m in range(4, 8):
                J.append([(ny * Nx + nx) * 8 + m, (ny * Nx + nx + 1) * 8 + m, draw()])
    return J


def synthetic_rmf(Nx, Ny, d, seed):
    """Seeded nearest-neighbour Random Markov Field with local dimension d."""
    rng = np.random.default_rng(seed)
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
