"""Oracle: the solver-side contraction and beam search on the CPU.

TEST INFRASTRUCTURE — see oracle/__init__.py.  Restates, in plain numpy, the parts of the
reference's ``tnac4o/tnac4o.py`` that sit on the hot path:
coupling split, PEPS/MPO construction, boundary-MPS sweeps, right/left environments,
conditional probabilities, the branch-and-bound row-major search, 90-degree rotations and
the 'balancing' preconditioner.  Control flow (argpartition / unique / merge order) follows
the reference so that tie-breaking agrees.

Differences in *representation* only: couplings are kept dense; the 5-leg PEPS tensor
(q,l,d,r,u), which is 1/(2^L2 2^L3) dense in the reference (tnac4o.py:1599-1607), is kept as
its non-zero factor F[s,l,u] plus the maps s->d(s), s->r(s).
"""
import itertools
import numpy as np
import scipy.linalg

from . import mps_ref as mr


def bit_table(n):
    """b[s,i] = i-th bit of s.  Reference `_cluster_configurations` (tnac4o.py:1461-1467)
    returns 1-b (first spin fastest); the Ising spin is 1-2b."""
    s = np.arange(2 ** n)[:, None]
    return ((s >> np.arange(n)[None, :]) & 1).astype(np.int64)


def spins(n):
    return 1 - 2 * bit_table(n)


class RefSolver:
    """CPU restatement of ``tnac4o.tnac4o`` (tnac4o.py:78-198) for the ground-state path."""

    def __init__(self, mode='Ising', Nx=4, Ny=4, Nc=8, beta=1, J=None):
        self.mode, self.beta = mode, beta
        self.Nx_model, self.Ny_model = Nx, Ny
        self.Nx, self.Ny = Nx, Ny
        self.Nc = Nc if mode == 'Ising' else 1
        self.indtype = np.int8 if self.Nc <= 8 else np.int16
        self.L = Nx * Ny * self.Nc
        self.order = np.arange(Nx * Ny)
        self.order_i = np.arange(Nx * Ny)
        self.rotation = 0
        self.energy = np.zeros(0)
        self.probability = np.zeros(0)
        self.degeneracy = 0
        self.states = np.zeros((0, Nx * Ny), dtype=self.indtype)
        if mode == 'Ising':
            # tnac4o.py:176-181: accumulate into an upper-triangular matrix
            Jd = np.zeros((self.L, self.L))
            for i, j, v in J:
                a, b = (i, j) if i <= j else (j, i)
                Jd[a, b] += v
            self.J = Jd
            self.ind0 = [[self._active(ny, nx) for nx in range(Nx)] for ny in range(Ny)]   # :185-191
        else:
            self.J = {'fun': J['fun'], 'fac': dict(J['fac']), 'N': J['N']}
            self.N = np.array(J['N']).copy()
        self._divide_couplings()

    # ------------------------------------------------------------------ coupling split
    def _active(self, ny, nx):
        ind = self.Nc * (self.Nx * ny + nx) + np.arange(self.Nc)
        w = np.abs(self.J[ind, :]).sum(1) + np.abs(self.J[:, ind]).sum(0)
        return ind[w > 1e-12]

    def _divide_couplings(self):
        """tnac4o.py:1391-1457."""
        Ny, Nx = self.Ny, self.Nx
        self.lu = np.ones((Ny, Nx), dtype=int)
        self.lr = np.ones((Ny, Nx), dtype=int)
        self.ll = np.ones((Ny, Nx), dtype=int)
        self.ld = np.ones((Ny, Nx), dtype=int)
        if self.mode == 'Ising':
            self.ind = [[self._active(ny, nx) for nx in range(Nx)] for ny in range(Ny)]
            self.sN = np.array([[len(self.ind[ny][nx]) for nx in range(Nx)] for ny in range(Ny)])
            self.N = 2 ** self.sN
            self.Jin = [[None] * Nx for _ in range(Ny)]
            self.Jl = [[np.zeros((self.sN[ny][nx], 0)) for nx in range(Nx)] for ny in range(Ny)]
            self.Ju = [[np.zeros((self.sN[ny][nx], 0)) for nx in range(Nx)] for ny in range(Ny)]
            self.id = [[np.zeros(0, dtype=int) for _ in range(Nx)] for _ in range(Ny)]
            self.ir = [[np.zeros(0, dtype=int) for _ in range(Nx)] for _ in range(Ny)]
            self.sl = np.zeros((Ny, Nx), dtype=int)
            self.sd = np.zeros((Ny, Nx), dtype=int)
            self.sr = np.zeros((Ny, Nx), dtype=int)
            self.su = np.zeros((Ny, Nx), dtype=int)
            for ny in range(Ny):
                for nx in range(Nx):
                    ind = self.ind[ny][nx]
                    self.Jin[ny][nx] = self.J[np.ix_(ind, ind)]
                    if nx > 0:
                        JJ = self.J[np.ix_(self.ind[ny][nx - 1], ind)]
                        rows = np.nonzero(np.abs(JJ).sum(1))[0]
                        self.Jl[ny][nx] = JJ[rows].T
                        self.ir[ny][nx - 1] = rows
                        self.sr[ny][nx - 1] = self.sl[ny][nx] = len(rows)
                        self.lr[ny][nx - 1] = 2 ** len(rows)
                    if ny > 0:
                        JJ = self.J[np.ix_(self.ind[ny - 1][nx], ind)]
                        rows = np.nonzero(np.abs(JJ).sum(1))[0]
                        self.Ju[ny][nx] = JJ[rows].T
                        self.id[ny - 1][nx] = rows
                        self.sd[ny - 1][nx] = self.su[ny][nx] = len(rows)
                        self.ld[ny - 1][nx] = 2 ** len(rows)
        else:
            fac = self.J['fac']
            for ny in range(Ny):
                for nx in range(Nx):
                    if (ny, nx - 1, ny, nx) in fac or (ny, nx, ny, nx - 1) in fac:
                        self.ll[ny, nx] = self.N[ny][nx - 1]
                    if (ny, nx, ny, nx + 1) in fac or (ny, nx + 1, ny, nx) in fac:
                        self.lr[ny, nx] = self.N[ny][nx + 1]
                    if (ny - 1, nx, ny, nx) in fac or (ny, nx, ny - 1, nx) in fac:
                        self.lu[ny, nx] = self.N[ny - 1][nx]
                    if (ny, nx, ny + 1, nx) in fac or (ny + 1, nx, ny, nx) in fac:
                        self.ld[ny, nx] = self.N[ny + 1][nx]
        self._reset_X()

    def _reset_X(self):
        """tnac4o.py:1811-1822."""
        Ny, Nx = self.Ny, self.Nx
        self.Xu = np.ones((Ny, Nx, np.max(self.ld)))
        self.Xd = np.ones((Ny, Nx, np.max(self.ld)))
        self.Xl = np.ones((Ny, Nx, np.max(self.lr)))
        self.Xr = np.ones((Ny, Nx, np.max(self.lr)))
        self.overlaps_ud = np.empty((0, Ny - 1))

    # ------------------------------------------------------------------ rotations
    def rotate_graph(self, rot=1):
        """tnac4o.py:290-340: cell (ny,nx) -> (Nx-1-nx, ny)."""
        for _ in range(rot):
            Nx, Ny, Nc = self.Nx, self.Ny, self.Nc
            order_i = np.arange(Nx * Ny)
            if self.mode == 'Ising':
                self.rotation += 1
                perm = np.arange(self.L)
                for nx in range(Nx):
                    for ny in range(Ny):
                        src = (ny * Nx + nx) * Nc + np.arange(Nc)
                        dst = ((Nx - nx - 1) * Ny + ny) * Nc + np.arange(Nc)
                        perm[src] = dst
                        order_i[(Nx - nx - 1) * Ny + ny] = ny * Nx + nx
                Jp = self.J[np.ix_(perm, perm)]
                self.J = np.triu(Jp) + np.tril(Jp, -1).T
            else:
                fac, new = self.J['fac'], {}
                Nn = np.zeros((Nx, Ny), dtype=int)
                for key, val in fac.items():
                    if len(key) == 2:
                        new[(Nx - key[1] - 1, key[0])] = val
                    else:
                        new[(Nx - key[1] - 1, key[0], Nx - key[3] - 1, key[2])] = val
                for nx in range(Nx):
                    for ny in range(Ny):
                        Nn[Nx - nx - 1, ny] = self.N[ny, nx]
                        order_i[ny * Nx + nx] = (Nx - nx - 1) * Ny + ny
                self.J['fac'], self.N = new, Nn
            self.Nx, self.Ny = Ny, Nx
            self.order = order_i[self.order]
        self.order_i[self.order] = np.arange(self.Nx * self.Ny)
        self.rotation = self.rotation % 4
        self._divide_couplings()

    # ------------------------------------------------------------------ bond indices
    def _ind_bond_down(self, st, ny, nx):
        """tnac4o.py:1469-1478."""
        if self.mode == 'Ising':
            b = bit_table(self.sN[ny][nx])
            return b[st][:, self.id[ny][nx]] @ (2 ** np.arange(self.sd[ny][nx]))
        return np.mod(st, self.ld[ny, nx])

    def _ind_bond_right(self, st, ny, nx):
        """tnac4o.py:1480-1489."""
        if self.mode == 'Ising':
            b = bit_table(self.sN[ny][nx])
            return b[st][:, self.ir[ny][nx]] @ (2 ** np.arange(self.sr[ny][nx]))
        return np.mod(st, self.lr[ny, nx])

    # ------------------------------------------------------------------ energies
    def _rmf_tables(self, ny, nx):
        """Unary, left-pair and up-pair tables of an RMF site (tnac4o.py:1613-1638)."""
        fac, fun, N = self.J['fac'], self.J['fun'], self.N[ny][nx]
        Es = np.reshape(fun[fac[(ny, nx)]], N) if (ny, nx) in fac else np.zeros(N)
        if (ny, nx - 1, ny, nx) in fac:
            E1 = fun[fac[(ny, nx - 1, ny, nx)]].T
        elif (ny, nx, ny, nx - 1) in fac:
            E1 = fun[fac[(ny, nx, ny, nx - 1)]]
        else:
            E1 = np.zeros((N, self.ll[ny, nx]))
        if (ny - 1, nx, ny, nx) in fac:
            E4 = fun[fac[(ny - 1, nx, ny, nx)]].T
        elif (ny, nx, ny - 1, nx) in fac:
            E4 = fun[fac[(ny, nx, ny - 1, nx)]]
        else:
            E4 = np.zeros((N, self.lu[ny, nx]))
        return Es, E1, E4

    def _cell_energies(self, ny, nx):
        """(Es[s], Ese1[s,l], Ese4[s,u]) — the three energy tables of a cell
        (Ising: tnac4o.py:1570-1581; RMF: 1613-1635)."""
        if self.mode == 'Ising':
            st = spins(self.sN[ny][nx])
            Jin = self.Jin[ny][nx]
            Es = np.sum((st @ np.triu(Jin, 1)) * st, 1) + st @ Jin.diagonal()
            E1 = (st @ self.Jl[ny][nx]) @ spins(self.sl[ny][nx]).T
            E4 = (st @ self.Ju[ny][nx]) @ spins(self.su[ny][nx]).T
            return Es, E1, E4
        return self._rmf_tables(ny, nx)

    def _update_Eng(self, states, ny, nx):
        """Energy increment of adding cell (ny,nx) to partial configurations (tnac4o.py:1506-1558)."""
        Es, E1, E4 = self._cell_energies(ny, nx)
        pos = ny * self.Nx + nx
        dE = 1.0 * Es[states[:, pos]]
        if nx > 0:
            left = states[:, pos - 1]
            il = self._ind_bond_right(left, ny, nx - 1) if self.mode == 'Ising' else left
            dE += E1[states[:, pos], il]
        if ny > 0:
            up = states[:, pos - self.Nx]
            iu = self._ind_bond_down(up, ny - 1, nx) if self.mode == 'Ising' else up
            dE += E4[states[:, pos], iu]
        return dE

    # ------------------------------------------------------------------ PEPS tensors
    def peps_factor(self, ny, nx):
        """Non-zero part of the PEPS tensor: F[s,l,u], d(s), r(s), and leg sizes (pd, br).

        T[s,l,d,r,u] = F[s,l,u] [d=d(s)] [r=r(s)]   (tnac4o.py:1562-1672).
        Floating-point evaluation order follows the reference: exp((Es+E1)+E4), then
        *Xu, *Xl, *Xr, *Xd.
        """
        b = self.beta
        Es, E1, E4 = self._cell_energies(ny, nx)
        Es = b * (np.min(Es) - Es)
        E1 = b * (np.min(E1) - E1)
        E4 = b * (np.min(E4) - E4)
        F = np.exp((Es[:, None, None] + E1[:, :, None]) + E4[:, None, :])
        nl, nu = F.shape[1], F.shape[2]
        F = F * self.Xu[ny][nx][:nu][None, None, :]
        F = F * self.Xl[ny][nx][:nl][None, :, None]
        q = F.shape[0]
        s = np.arange(q)
        if self.mode == 'Ising':
            bt = bit_table(self.sN[ny][nx])
            rmap = bt[:, self.ir[ny][nx]] @ (2 ** np.arange(self.sr[ny][nx]))
            dmap = bt[:, self.id[ny][nx]] @ (2 ** np.arange(self.sd[ny][nx]))
            br, pd = 2 ** self.sr[ny][nx], 2 ** self.sd[ny][nx]
        else:
            br, pd = self.lr[ny, nx], self.ld[ny, nx]
            rmap = s % br if br > 1 else np.zeros(q, dtype=int)
            dmap = s % pd if pd > 1 else np.zeros(q, dtype=int)
        F = F * self.Xr[ny][nx][rmap][:, None, None]
        F = F * self.Xd[ny][nx][dmap][:, None, None]
        return F, dmap, rmap, pd, br

    def peps_dense(self, ny, nx):
        """The reference's dense (q,l,d,r,u) tensor — for small-case checks only."""
        F, dmap, rmap, pd, br = self.peps_factor(ny, nx)
        q, nl, nu = F.shape
        T = np.zeros((q, nl, pd, br, nu))
        T[np.arange(q), :, dmap, rmap, :] = F
        return T

    def mpo_site(self, ny, nx):
        """W[l,d,r,u] = sum_s T[s,l,d,r,u] (tnac4o.py:1686), summed in increasing s."""
        F, dmap, rmap, pd, br = self.peps_factor(ny, nx)
        q, nl, nu = F.shape
        W = np.zeros((nl, pd, br, nu))
        for s in range(q):
            W[:, dmap[s], rmap[s], :] += F[s]
        return W

    # ------------------------------------------------------------------ sweeps
    def _row_mpo(self, ny):
        mpo = mr.RefMPO(self.Nx)
        for nx in range(self.Nx):
            mpo.set_direct(self.mpo_site(ny, nx), nx)
        return mpo

    def _setup_rhoT(self, graduate_truncation=True, Dmax=32, tolS=1e-16, tolV=1e-10, max_sweeps=20):
        """Top boundary MPS for every row, built bottom-up (tnac4o.py:1674-1695)."""
        Ny = self.Ny
        self.rhoT = [None] * (Ny + 1)
        self.rhoT_overlap = [1] * (Ny + 1)
        self.rhoT_discarded = [0] * (Ny + 1)
        self.rhoT[Ny] = mr.RefMPS(d=1, L=self.Nx, Dmax=1)
        for ny in range(Ny - 1, -1, -1):
            psi = self.rhoT[ny + 1].copy()
            psi.apply_mpo(self._row_mpo(ny), Hconj=True)
            self.rhoT_overlap[ny] = psi.compress_mps(Dmax=Dmax, tolS=tolS, tolV=tolV, max_sweeps=max_sweeps,
                                                     graduate_truncation=graduate_truncation)
            self.rhoT_discarded[ny] = max(psi.discarded)
            self.rhoT[ny] = psi

    def _setup_rhoB(self, graduate_truncation=True, Dmax=32, tolS=1e-16, tolV=1e-10, max_sweeps=20):
        """Bottom boundary MPS, built top-down (tnac4o.py:1697-1718)."""
        Ny = self.Ny
        self.rhoB = [None] * (Ny + 1)
        self.rhoB_overlap = [1] * (Ny + 1)
        self.rhoB_discarded = [0] * (Ny + 1)
        self.rhoB[0] = mr.RefMPS(d=1, L=self.Nx, Dmax=1)
        for ny in range(Ny):
            psi = self.rhoB[ny].copy()
            psi.apply_mpo(self._row_mpo(ny), Hconj=False)
            self.rhoB_overlap[ny + 1] = psi.compress_mps(Dmax=Dmax, tolS=tolS, tolV=tolV, max_sweeps=max_sweeps,
                                                         graduate_truncation=graduate_truncation)
            self.rhoB_discarded[ny + 1] = max(psi.discarded)
            self.rhoB[ny + 1] = psi

    # ------------------------------------------------------------------ preconditioning
    def precondition(self, mode='balancing', steps=2, beta_cond=(), Dmax_cond=(), max_scale=1024,
                     graduate_truncation=False, tolS=1e-16, tolV=1e-10, max_sweeps=20):
        """tnac4o.py:342-379."""
        beta_cond = list(beta_cond) or [self.beta * 2.0 ** (n - steps) for n in range(steps)]
        Dmax_cond = list(Dmax_cond) or [8] * len(beta_cond)
        main_beta = self.beta
        for b, D in zip(beta_cond, Dmax_cond):
            self.beta = b
            self._update_conditioning(Dmax=D, graduate_truncation=graduate_truncation, tolS=tolS, tolV=tolV,
                                      max_sweeps=max_sweeps, max_scale=max_scale)
        self.beta = main_beta

    def _balance_site(self, B, T, ny, nx, max_scale, overlaps):
        """One balancing step on the vertical bond above cell (ny,nx) (tnac4o.py:1844-1867)."""
        env = B.bond_env_mix(T, nx)
        _, sc = scipy.linalg.matrix_balance(env, permute=False, separate=True)
        sc = np.minimum(np.maximum(sc[0], 1 / max_scale), max_scale)
        o1 = B.expectation_mix(T, nx) * (1 / (np.linalg.norm(B.A[nx]) * np.linalg.norm(T.A[nx])))
        B.apply_diagonalO(sc, nx)
        T.apply_diagonalO(1 / sc, nx)
        o2B, o2T = np.linalg.norm(B.A[nx]), np.linalg.norm(T.A[nx])
        o2 = B.expectation_mix(T, nx) * (1 / (o2B * o2T))
        if o1 < overlaps[0, ny - 1]:
            overlaps[0, ny - 1] = o1
            overlaps[1, ny - 1] = max(o1, o2)
        k = self.ld[ny - 1, nx]
        self.Xd[ny - 1, nx, :k] *= sc
        self.Xu[ny, nx, :k] *= 1 / sc

    def _update_conditioning(self, graduate_truncation=False, Dmax=8, tolS=1e-16, tolV=1e-10, max_sweeps=4,
                             max_scale=1024):
        """'ud' balancing sweep (tnac4o.py:1824-1918)."""
        max_scale = mr.pow2_floor_max(np.sqrt(max_scale))
        kw = dict(graduate_truncation=graduate_truncation, Dmax=Dmax, tolS=tolS, tolV=tolV, max_sweeps=max_sweeps)
        self._setup_rhoT(**kw)
        self._setup_rhoB(**kw)
        overlaps = np.ones((2, self.Ny - 1))
        Nx = self.Nx
        for ny in range(1, self.Ny):
            B, T = self.rhoB[ny], self.rhoT[ny]
            for nx in range(Nx):
                B.update_RL_mix(T, nx)
                B.R[nx + 1] *= 1 / np.linalg.norm(B.R[nx + 1])
            for nx in range(Nx - 1, -1, -1):
                self._balance_site(B, T, ny, nx, max_scale, overlaps)
                if nx > 0:
                    B.orth_right(nx)
                    B.attach_AC()
                    T.orth_right(nx)
                    T.attach_AC()
                    B.update_RR_mix(T, nx)
                    B.R[nx] *= 1 / np.linalg.norm(B.R[nx])
            for nx in range(Nx):
                self._balance_site(B, T, ny, nx, max_scale, overlaps)
                if nx < Nx - 1:
                    B.orth_left(nx)
                    B.attach_CA()
                    T.orth_left(nx)
                    T.attach_CA()
                    B.update_RL_mix(T, nx)
                    B.R[nx + 1] *= 1 / np.linalg.norm(B.R[nx + 1])
        self.overlaps_ud = np.vstack([self.overlaps_ud, overlaps])
        self.rhoB = []

    # ------------------------------------------------------------------ search pieces
    def _setup_RR(self, vind, ny):
        """Right environments per distinct boundary-index suffix (tnac4o.py:1768-1784)."""
        top = self.rhoT[ny + 1]
        RRl = [{(): np.ones((1, 1))}]
        for nx in range(self.Nx - 1, 0, -1):
            W = self.mpo_site(ny, nx)
            new = {}
            for row in vind:
                key = tuple(row[nx + 1:])
                if key not in new:
                    T = np.tensordot(top.A[nx], RRl[-1][key[1:]], axes=(2, 0))
                    R = np.tensordot(T, W[:, :, :, key[0]], axes=([1, 2], [1, 2]))
                    R *= 1 / mr.pow2_floor_max(R)
                    new[key] = R
            RRl.append(new)
        return RRl

    @staticmethod
    def conditional_probabilities(Fslice, dmap, rmap, RL, AT, RR):
        """Conditional distribution of one cell for one branch (tnac4o.py:1786-1807).

        Fslice[s] = F[s, l, u] for the branch's (l,u); Pn[s] = Fslice[s] * T2[d(s), r(s)].
        Negative handling: entries below |min| are raised to |min|; all-zero -> uniform, flag -1.
        """
        T2 = np.tensordot(np.tensordot(RL, AT, axes=(0, 0)), RR, axes=(1, 0))
        Pn = Fslice * T2[dmap, rmap]
        mPn = Pn.min()
        if mPn < 0.0:
            low = Pn < np.abs(mPn)
            Pn[low] = np.abs(mPn)
            mPn *= np.sum(low)
        no = np.sum(Pn)
        if no > 0.0:
            Pn *= 1.0 / no
            mPn *= 1.0 / no
        else:
            Pn += 1.0 / len(Pn)
            mPn = -1
        return Pn, mPn

    def search_ground_state(self, M=2 ** 10, relative_P_cutoff=1e-6, min_dEng=1e-12, graduate_truncation=True,
                            Dmax=32, tolS=1e-16, tolV=1e-10, max_sweeps=20, trace=None, sweep_hook=None,
                            pn_gather=None, merge_hook=None, row_hook=None):
        """Row-major branch-and-bound (tnac4o.py:381-551).  ``trace`` (a list) receives the
        (ny, nx, newprob, minprob) tables of every site-step when given (for golden checks).
        ``sweep_hook(solver, run_sweep)`` / ``pn_gather(compute, nb, q)``: optional injection points used by the CPU
        multi-rank tests to wrap the sweep and the per-branch table in the product's sharding helpers; with both None
        this is the plain reference algorithm.  ``merge_hook(site, parents, order, starts, Eng, prob, states, rep, probn,
        selected)`` is told about every merge (the droplet bookkeeping of tnac4o.py:843-873 plugs in here in the tests),
        ``row_hook()`` is called after each row."""
        M_ = M
        kw_sweep = dict(graduate_truncation=graduate_truncation, Dmax=Dmax, tolS=tolS, tolV=tolV, max_sweeps=max_sweeps)
        if sweep_hook is None:
            self._setup_rhoT(**kw_sweep)
        else:
            sweep_hook(self, lambda: self._setup_rhoT(**kw_sweep))
        gather = pn_gather
        Nx, Ny = self.Nx, self.Ny
        vind = np.zeros((1, Nx + 1), dtype=self.indtype)
        states = np.zeros((1, Nx * Ny), dtype=self.indtype)
        Eng, prob, deg = np.zeros(1), np.zeros(1), np.ones(1, dtype=int)
        pd_max, globalmin = -np.inf, 0.0

        for ny in range(Ny):
            RRl = self._setup_RR(vind, ny)
            RLl = {(): np.ones(1)}
            top = self.rhoT[ny + 1]
            for nx in range(Nx):
                q, nb = self.N[ny][nx], prob.size
                F, dmap, rmap, _, _ = self.peps_factor(ny, nx)
                def pn_slice(lo, hi):
                    P, mP = np.zeros((hi - lo, q)), np.zeros(hi - lo)
                    for k in range(lo, hi):
                        t = tuple(vind[k])
                        P[k - lo], mP[k - lo] = self.conditional_probabilities(
                            F[:, t[nx], t[nx + 1]], dmap, rmap, RLl[t[:nx]], top.A[nx], RRl[Nx - nx - 1][t[nx + 2:]])
                    return P, mP
                if gather is None:
                    newprob, minprob = pn_slice(0, nb)
                else:
                    newprob, minprob = gather(pn_slice, nb, q)
                if trace is not None:
                    trace.append((ny, nx, newprob.copy(), minprob.copy(), vind.copy()))
                with np.errstate(divide='ignore'):
                    newprob = np.log2(newprob)
                newprob += prob[:, None]
                prob = newprob.reshape(nb * q)
                minprob = np.min(minprob)

                order = np.arange(prob.size)
                if relative_P_cutoff > 0:                                   # :458-465
                    cutoff = np.max(prob) + np.log2(relative_P_cutoff)
                    keep = max(int((prob > cutoff).sum()), 1)
                    if keep < prob.size:
                        order = prob.argpartition(-keep - 1)
                        pd_max = max(pd_max, prob[order[-keep - 1]])
                        order = order[-keep:]
                        prob = prob[order]

                inds, indc = order // q, np.mod(order, q)                    # :469-478
                states = states[inds]
                states[:, ny * Nx + nx] = indc
                vind = vind[inds]
                deg = deg[inds]
                vind[:, nx] = self._ind_bond_down(indc, ny, nx)
                vind[:, nx + 1] = self._ind_bond_right(indc, ny, nx)
                Eng = Eng[inds]
                Eng += self._update_Eng(states, ny, nx)

                vindn, inv = np.unique(vind, return_inverse=True, axis=0)    # :481-515
                inv = inv.reshape(-1)
                order = inv.argsort()
                inv = inv[order]
                sizes = [len(list(g)) for _, g in itertools.groupby(inv)]
                n_grp = len(sizes)
                indn = np.zeros(n_grp, dtype=int)
                degn = np.zeros(n_grp, dtype=int)
                probn = np.zeros(n_grp)
                lo = 0
                for k, sz in enumerate(sizes):
                    ind = order[lo:lo + sz]
                    Ek = Eng[ind]
                    imin = np.argmin(Ek)
                    indn[k] = ind[imin]
                    same = ind[(Ek - Ek[imin]) <= min_dEng]
                    if len(same) > 1:
                        degn[k] = sum(deg[same])
                        probn[k] = np.mean(prob[same])
                    else:
                        degn[k] = deg[same][0]
                        probn[k] = prob[same][0]
                    lo += sz
                sel = None
                if probn.size > M_:                                         # :518-526
                    sel = probn.argpartition(-M_ - 1)
                    pd_max = max(pd_max, probn[sel[-M_ - 1]])
                    sel = sel[-M_:]
                if merge_hook is not None:
                    starts = np.cumsum([0] + sizes[:-1])
                    merge_hook(ny * Nx + nx, inds, order, starts, Eng, prob, states, indn, probn,
                               np.arange(n_grp) if sel is None else sel)
                vind, prob, deg = vindn, probn, degn
                states, Eng = states[indn], Eng[indn]
                if sel is not None:
                    vind, states, prob, Eng, deg = vind[sel], states[sel], prob[sel], Eng[sel], deg[sel]

                RLnew = {}                                                  # :528-535
                for row in vind:
                    t = tuple(row[:nx + 1])
                    if t not in RLnew:
                        R = np.dot(RLl[t[:-1]], top.A[nx][:, t[-1], :])
                        R *= 1 / mr.pow2_floor_max(R)
                        RLnew[t] = R
                RLl = RLnew
                globalmin = min(globalmin, minprob)

            if row_hook is not None:
                row_hook()
            vind[:, 1:] = vind[:, :-1]                                      # :540-542
            vind[:, 0] = 0

        self.energy = Eng
        self.degeneracy = deg[0]
        self.states = states[:, self.order]
        self.probability = prob
        self.discarded_probability = pd_max
        self.negative_probability = min(globalmin, 0)
        return Eng

    # ------------------------------------------------------------------ output
    def gibbs_sampling(self, M=2 ** 10, graduate_truncation=True, Dmax=32, tolS=1e-15, tolV=1e-10, max_sweeps=20):
        """Gibbs sampling cell by cell (tnac4o.py:553-650); numpy's global generator, one rand(M) per cell."""
        self._setup_rhoT(graduate_truncation=graduate_truncation, Dmax=Dmax, tolS=tolS, tolV=tolV,
                         max_sweeps=max_sweeps)
        Nx, Ny = self.Nx, self.Ny
        vind = np.zeros((M, Nx + 1), dtype=int)
        states = np.zeros((M, Nx * Ny), dtype=int)
        Eng = np.zeros(M)
        globalmin = 1.0
        for ny in range(Ny):
            RRl = self._setup_RR(vind, ny)
            RLl = {(): np.ones(1)}
            top = self.rhoT[ny + 1]
            for nx in range(Nx):
                q = self.N[ny][nx]
                F, dmap, rmap, _, _ = self.peps_factor(ny, nx)
                newprob = np.zeros((M, q))
                minprob = np.zeros(M)
                seen = {}
                for kk in range(M):                                          # :601-612
                    t = tuple(vind[kk])
                    if t in seen:
                        newprob[kk] = newprob[seen[t]]
                        minprob[kk] = minprob[seen[t]]
                    else:
                        seen[t] = kk
                        newprob[kk], minprob[kk] = self.conditional_probabilities(
                            F[:, t[nx], t[nx + 1]], dmap, rmap, RLl[t[:nx]], top.A[nx], RRl[Nx - nx - 1][t[nx + 2:]])
                minprob = np.min(minprob)
                newprob = newprob.cumsum(axis=1)                             # :616-622
                rr = np.random.rand(M)
                indc = np.zeros(M, dtype=int)
                for kk in range(M):
                    indc[kk] = np.searchsorted(newprob[kk], rr[kk])
                states[:, ny * Nx + nx] = indc
                vind[:, nx] = self._ind_bond_down(indc, ny, nx)
                vind[:, nx + 1] = self._ind_bond_right(indc, ny, nx)
                Eng += self._update_Eng(states, ny, nx)
                RLnew = {}                                                   # :628-636
                for row in vind:
                    t = tuple(row[:nx + 1])
                    if t not in RLnew:
                        r = np.dot(RLl[t[:-1]], top.A[nx][:, t[-1], :])
                        r *= 1 / mr.pow2_floor_max(r)
                        RLnew[t] = r
                RLl = RLnew
                globalmin = min(globalmin, minprob)
            vind[:, 1:] = vind[:, :-1]
            vind[:, 0] = 0
        self.energy = Eng
        self.degeneracy = 0
        self.states = states[:, self.order]
        self.probability = np.zeros(1)
        self.discarded_probability = 0
        self.negative_probability = min(globalmin, 0)
        return Eng

    def binary_states(self, number=-1):
        """tnac4o.py:261-288: 1 = spin up, 0 = spin down, 2 = inactive."""
        ns = self.states.shape[0]
        ns = ns + number + 1 if number < 0 else min(number, ns)
        if self.mode != 'Ising':
            return self.states[:ns]
        out = np.zeros((ns, self.L), dtype=np.int8) + 2
        k = -1
        for ny in range(self.Ny_model):
            for nx in range(self.Nx_model):
                k += 1
                act = self.ind0[ny][nx]
                out[:, act] = (1 - bit_table(len(act)))[self.states[:ns, k]]
        return out


# ---------------------------------------------------------------------- coupling helpers
def load_Jij(path):
    """auxx.py:26-38."""
    return [[int(r[0]), int(r[1]), float(r[2])] for r in np.loadtxt(path)]


def Jij_f2p(J):
    """auxx.py:68-81."""
    return [[r[0] - 1, r[1] - 1, r[2]] for r in J]


def round_Jij(J, dJ):
    """auxx.py:41-52."""
    dJ = float(dJ)
    return [[r[0], r[1], round(r[2] / dJ) * dJ] for r in J]


def energy_Jij(J, states):
    """Independent energy check (auxx.py:84-109): states are 0/1 bit strings."""
    st = 2.0 * np.asarray(states, dtype=float) - 1.0
    L = st.shape[1]
    Jd = np.zeros((L, L))
    for i, j, v in J:
        a, b = (i, j) if i <= j else (j, i)
        Jd[a, b] += v
    return np.sum((st @ np.triu(Jd, 1)) * st, 1) + st @ Jd.diagonal()
