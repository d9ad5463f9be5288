"""The ``tnac4o`` solver class on MI355X: same constructor, methods and result attributes as the reference's
``tnac4o.tnac4o`` (tnac4o/tnac4o.py:78-551) for the ground-state path, with the PEPS contraction on the GPU.

Split of work
  host (numpy, O(nnz) / O(M) integer work): coupling split, rotations, energy tables, the branch bookkeeping of
      search_ground_state (cut-off, merge of equal boundary indices, top-M) — restated from the reference so that
      tie-breaking agrees;
  GPU (libtnpeps): boundary-MPS sweeps (absorb, QR, Jacobi SVD, GEMMs) through ``tnac4o_amd.mps``; right
      environments for every distinct boundary suffix (batched GEMMs); left environments for every distinct prefix;
      conditional probabilities of all branches of a site-step in one launch (tn_calc_pn).

The 5-leg PEPS tensor (q,l,d,r,u) of the reference (tnac4o.py:1562-1672; 134 MB and 1/256 dense for chimera) is
never formed: T[s,l,d,r,u] = F[s,l,u] [d = dmap[s]] [r = rmap[s]].
"""
import itertools
import logging
import os

import numpy as np
import functools

import torch

from . import mps, ops


@functools.lru_cache(maxsize=64)
def _bits(n):
    # (memoised, read-only: a sweep asks for the same handful of tables four times per site)
    s = np.arange(2 ** n)[:, None]
    out = ((s >> np.arange(n)[None, :]) & 1).astype(np.int64)
    out.setflags(write=False)
    return out


@functools.lru_cache(maxsize=64)
def _spins(n):
    out = 1 - 2 * _bits(n)
    out.setflags(write=False)
    return out


def _dev_f64(x):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float64).cuda()


def _dev_i32(x):
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.int32)).cuda()


def _unique_rows(a):
    """Sorted unique rows and the inverse map — the same result as np.unique(a, axis=0, return_inverse=True)
    (lexicographic row order) for non-negative integer rows, computed on packed 64-bit keys instead of a sort of
    multi-byte records (30x faster for the 262144 x 17 candidate tables of a site-step).  Handles zero-width keys."""
    n, w = a.shape
    if w == 0:
        return np.zeros((1, 0), dtype=a.dtype), np.zeros(n, dtype=np.int64)
    if n == 0 or a.min() < 0:
        u, inv = np.unique(a, axis=0, return_inverse=True)
        return u, inv.reshape(-1)
    bits = max(1, int(a.max()).bit_length())
    inv, c0 = None, 0
    while c0 < w:
        # the rank of the prefix processed so far (order preserving) goes in the high bits, as many new columns as fit below
        used = 0 if inv is None else max(1, int(inv.max()).bit_length())
        ncol = max(1, min(w - c0, (63 - used) // bits))
        key = np.zeros(n, dtype=np.uint64) if inv is None else inv.astype(np.uint64)
        for j in range(c0, c0 + ncol):
            key = (key << np.uint64(bits)) | a[:, j].astype(np.uint64)
        _, inv = np.unique(key, return_inverse=True)
        inv = inv.reshape(-1)
        c0 += ncol
    first = np.zeros(int(inv.max()) + 1, dtype=np.int64)
    first[inv[::-1]] = np.arange(n - 1, -1, -1)      # any representative row of each group
    return a[first], inv


def _merge_groups(inv, Eng, prob, deg, min_dEng, canonical=True):
    """The merge of branches with identical boundary indices (tnac4o.py:481-509), vectorised: per group the
    representative is the first minimal-energy member, the degeneracy is summed over members within min_dEng of the minimum
    and their log-probabilities are averaged.  canonical (default; see tnac4o_amd/beam.py): members in candidate order (stable
    sort), the mean added up in member order -- what tn_merge_groups does on the device.  canonical=False: numpy's own order
    (unstable argsort, np.mean), the order the reference happens to get."""
    order = inv.argsort(kind='stable') if canonical else inv.argsort()
    ginv = inv[order]
    n_grp = int(ginv[-1]) + 1
    starts = np.flatnonzero(np.r_[True, ginv[1:] != ginv[:-1]])
    E = Eng[order]
    Emin = np.minimum.reduceat(E, starts)
    pos = np.arange(E.size)
    is_min = E == Emin[ginv]
    first_min = np.minimum.reduceat(np.where(is_min, pos, E.size), starts)
    indn = order[first_min]
    near = (E - Emin[ginv]) <= min_dEng
    cnt = np.add.reduceat(near.astype(np.int64), starts)
    degn = np.add.reduceat(np.where(near, deg[order], 0), starts)
    probn = prob[indn].copy()
    # single-member case: deg/prob of that member (== the representative); several: mean in the reference's summation order
    for k in np.flatnonzero(cnt > 1):
        lo = starts[k]
        hi = starts[k + 1] if k + 1 < n_grp else E.size
        same = order[lo:hi][near[lo:hi]]
        if canonical:
            acc = 0.0
            for v in prob[same]:
                acc += float(v)
            probn[k] = acc / len(same)
        else:
            probn[k] = np.mean(prob[same])
    return indn, degn, probn, order, starts


def load(file_name):
    """Load a solution written by `tnac4o.save` -- by this package or by the reference (same .npy pickle of a dict,
    tnac4o.py:31-75).  Couplings are not stored, so the returned instance only carries the results (energy, states, ...)
    and what `binary_states` needs."""
    d = np.load(file_name, allow_pickle=True).item()
    ins = tnac4o(mode=d.get('mode'), Nx=d.get('Nx'), Ny=d.get('Ny'), Nc=d.get('Nc'), beta=d.get('beta'))
    for k in ('energy', 'probability', 'degeneracy', 'states', 'discarded_probability', 'negative_probability'):
        setattr(ins, k, d.get(k))
    if d.get('rotation') is not None:
        ins.rotation = d.get('rotation')
    if ins.mode == 'Ising':
        ins.ind0 = d.get('ind')
        ins.adj = np.zeros((0, 0))
    else:
        ins.ind0, ins.adj = [], []
    if d.get('excitations_encoding') is not None:          # droplet bookkeeping of the reference (:64-74): carried over
        for k in ('excitations_encoding', 'd', 'invd', 'el', 'free_d'):
            setattr(ins, k, d.get(k))
        if ins.excitations_encoding > 1:
            from . import droplets
            if ins.mode == 'Ising':
                adj = d.get('adj')
                ins.adj = np.asarray(adj.toarray() if hasattr(adj, 'toarray') else adj) != 0
                ins._conn = droplets.Connectivity('Ising', ins.Nx, ind=ins.ind0, adj=ins.adj | ins.adj.T)
            else:
                ins._conn = droplets.Connectivity('RMF', ins.Nx)
    return ins


class tnac4o:
    """Ising ('Ising') or Random-Markov-Field ('RMF') problem on an Nx x Ny lattice of cells (tnac4o.py:145-198)."""

    def __init__(self, mode='Ising', Nx=4, Ny=4, Nc=8, beta=1, J=None):
        self.mode, self.beta = mode, beta
        self.Nx_model, self.Ny_model = Nx, Ny
        self.Nx, self.Ny = Nx, Ny
        if mode == 'Ising':
            if Nc > 9:
                raise ValueError('Single cluster is too large.')
            self.Nc = Nc
            self.indtype = np.int8 if Nc <= 8 else np.int16
        elif mode == 'RMF':
            self.Nc = 1
            self.indtype = np.int8
        else:
            raise ValueError("mode must be 'Ising' or 'RMF'")
        self.L = Nx * Ny * self.Nc
        self.order = np.arange(Nx * Ny)
        self.order_i = np.arange(Nx * Ny)
        self.logger = logging.getLogger('tnac4o')
        self.energy = np.zeros(0)
        self.probability = np.zeros(0)
        self.rotation = 0
        self.degeneracy = 0
        self.states = np.zeros((0, Nx * Ny), dtype=self.indtype)
        self.discarded_probability = -np.inf
        self.negative_probability = 0.0
        self.ind0, self.J0 = [], []
        if J is not None:
            if mode == 'Ising':
                Jd = np.zeros((self.L, self.L))              # upper triangular accumulation (tnac4o.py:176-181)
                for i, j, v in J:
                    a, b = (i, j) if i <= j else (j, i)
                    Jd[a, b] += v
                self.J = Jd
                self.J0 = Jd.copy()
                self.ind0 = [[self._active(ny, nx) for nx in range(Nx)] for ny in range(Ny)]
                self.active = sum(len(self.ind0[ny][nx]) for ny in range(Ny) for nx in range(Nx))
            else:
                self.J = {'fun': J['fun'], 'fac': dict(J['fac']), 'N': J['N']}
                self.N = np.array(J['N']).copy()
            self._divide_couplings()

    # ------------------------------------------------------------------------------------ problem setup (host)
    def _active(self, ny, nx):
        ind = self.Nc * (self.Nx * ny + nx) + np.arange(self.Nc)
        w = np.abs(self.J[ind, :]).sum(1) + np.abs(self.J[:, ind]).sum(0)
        return ind[w > 1e-12]

    def _divide_couplings(self):
        """Per-cell coupling blocks and bond index sets (tnac4o.py:1391-1457)."""
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
            self.sl, self.sd, self.sr, self.su = (np.zeros((Ny, Nx), dtype=int) for _ in range(4))
            for ny in range(Ny):
                for nx in range(Nx):
                    ind = self.ind[ny][nx]
                    self.Jin[ny][nx] = self.J[np.ix_(ind, ind)]
                    for (oy, ox, Jn, idx, s_here, s_there, ldim) in (
                            (ny, nx - 1, self.Jl, self.ir, self.sl, self.sr, self.lr),
                            (ny - 1, nx, self.Ju, self.id, self.su, self.sd, self.ld)):
                        if oy < 0 or ox < 0:
                            continue
                        JJ = self.J[np.ix_(self.ind[oy][ox], ind)]
                        rows = np.nonzero(np.abs(JJ).sum(1))[0]
                        Jn[ny][nx] = JJ[rows].T
                        idx[oy][ox] = rows
                        s_here[ny][nx] = s_there[oy][ox] = len(rows)
                        ldim[oy][ox] = 2 ** len(rows)
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
        """Gauge diagonals on the PEPS bonds (tnac4o.py:1811-1822)."""
        Ny, Nx = self.Ny, self.Nx
        self.Xu = np.ones((Ny, Nx, np.max(self.ld)))
        self.Xd = np.ones((Ny, Nx, np.max(self.ld)))
        self.Xl = np.ones((Ny, Nx, np.max(self.lr)))
        self.Xr = np.ones((Ny, Nx, np.max(self.lr)))
        self.overlaps_ud = np.empty((0, Ny - 1))

    def rotate_graph(self, rot=1):
        """Rotate the lattice by 90 degrees `rot` times: cell (ny,nx) -> (Nx-1-nx, ny) (tnac4o.py:290-340)."""
        for _ in range(rot):
            Nx, Ny, Nc = self.Nx, self.Ny, self.Nc
            order_i = np.arange(Nx * Ny)
            if self.mode == 'Ising':
                self.rotation += 1
                cells = np.arange(Nx * Ny).reshape(Ny, Nx)
                dst = ((Nx - 1 - np.arange(Nx))[None, :] * Ny + np.arange(Ny)[:, None])      # [ny, nx] -> new cell
                perm = np.empty(self.L, dtype=int)
                perm[(cells[:, :, None] * Nc + np.arange(Nc)).reshape(-1)] = (dst[:, :, None] * Nc + np.arange(Nc)).reshape(-1)
                order_i[dst.reshape(-1)] = cells.reshape(-1)
                Jp = self.J[np.ix_(perm, perm)]
                self.J = np.triu(Jp) + np.tril(Jp, -1).T
            else:
                new = {}
                for key, val in self.J['fac'].items():
                    if len(key) == 2:
                        new[(Nx - key[1] - 1, key[0])] = val
                    else:
                        new[(Nx - key[1] - 1, key[0], Nx - key[3] - 1, key[2])] = val
                Nn = np.zeros((Nx, Ny), dtype=int)
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

    # ------------------------------------------------------------------------------------ local tables (host)
    def _ind_bond_down(self, st, ny, nx):
        """tnac4o.py:1469-1478."""
        if self.mode == 'Ising':
            return _bits(self.sN[ny][nx])[st][:, self.id[ny][nx]] @ (2 ** np.arange(self.sd[ny][nx]))
        return np.mod(st, self.ld[ny, nx])

    def _ind_bond_right(self, st, ny, nx):
        """tnac4o.py:1480-1489."""
        if self.mode == 'Ising':
            return _bits(self.sN[ny][nx])[st][:, self.ir[ny][nx]] @ (2 ** np.arange(self.sr[ny][nx]))
        return np.mod(st, self.lr[ny, nx])

    def _cell_energies(self, ny, nx):
        """Es[s], Ese1[s,l], Ese4[s,u] (Ising tnac4o.py:1570-1581, RMF 1613-1635)."""
        if self.mode == 'Ising':
            st = _spins(self.sN[ny][nx])
            Jin = self.Jin[ny][nx]
            Es = np.sum((st @ np.triu(Jin, 1)) * st, 1) + st @ Jin.diagonal()
            return Es, (st @ self.Jl[ny][nx]) @ _spins(self.sl[ny][nx]).T, (st @ self.Ju[ny][nx]) @ _spins(self.su[ny][nx]).T
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

    def _update_Eng(self, states, ny, nx):
        """Energy added by cell (ny,nx) to partial configurations (tnac4o.py:1506-1558)."""
        Es, E1, E4 = self._cell_energies(ny, nx)
        pos = ny * self.Nx + nx
        dE = 1.0 * Es[states[:, pos]]
        if nx > 0:
            left = states[:, pos - 1]
            dE += E1[states[:, pos], self._ind_bond_right(left, ny, nx - 1) if self.mode == 'Ising' else left]
        if ny > 0:
            up = states[:, pos - self.Nx]
            dE += E4[states[:, pos], self._ind_bond_down(up, ny - 1, nx) if self.mode == 'Ising' else up]
        return dE

    def _peps_factor(self, ny, nx):
        """F[s,l,u], dmap[s], rmap[s], pd, br with T[s,l,d,r,u] = F[s,l,u][d=dmap[s]][r=rmap[s]]
        (tnac4o.py:1562-1672; same floating-point evaluation order as the reference)."""
        b = self.beta
        Es, E1, E4 = self._cell_energies(ny, nx)
        Es, E1, E4 = b * (np.min(Es) - Es), b * (np.min(E1) - E1), b * (np.min(E4) - E4)
        F = np.exp((Es[:, None, None] + E1[:, :, None]) + E4[:, None, :])
        nl, nu = F.shape[1], F.shape[2]
        F = F * self.Xu[ny][nx][:nu][None, None, :]
        F = F * self.Xl[ny][nx][:nl][None, :, None]
        q = F.shape[0]
        if self.mode == 'Ising':
            bt = _bits(self.sN[ny][nx])
            rmap = bt[:, self.ir[ny][nx]] @ (2 ** np.arange(self.sr[ny][nx]))
            dmap = bt[:, self.id[ny][nx]] @ (2 ** np.arange(self.sd[ny][nx]))
            br, pd = 2 ** self.sr[ny][nx], 2 ** self.sd[ny][nx]
        else:
            s = np.arange(q)
            br, pd = int(self.lr[ny, nx]), int(self.ld[ny, nx])
            rmap = s % br if br > 1 else np.zeros(q, dtype=int)
            dmap = s % pd if pd > 1 else np.zeros(q, dtype=int)
        F = F * self.Xr[ny][nx][rmap][:, None, None]
        F = F * self.Xd[ny][nx][dmap][:, None, None]
        return F, np.asarray(dmap, dtype=np.int64), np.asarray(rmap, dtype=np.int64), int(pd), int(br)

    def _site_tables(self, ny, nx):
        """Host side of K7: the three beta-scaled, min-shifted energy tables and the index maps of a cell (O(q) work;
        tnac4o.py:1570-1583, 1598-1607).  The exponentials, gauge products and the sum over s run on the GPU."""
        b = self.beta
        Es, E1, E4 = self._cell_energies(ny, nx)
        Es, E1, E4 = b * (np.min(Es) - Es), b * (np.min(E1) - E1), b * (np.min(E4) - E4)
        q = Es.shape[0]
        if self.mode == 'Ising':
            bt = _bits(self.sN[ny][nx])
            rmap = bt[:, self.ir[ny][nx]] @ (2 ** np.arange(self.sr[ny][nx]))
            dmap = bt[:, self.id[ny][nx]] @ (2 ** np.arange(self.sd[ny][nx]))
            br, pd = 2 ** self.sr[ny][nx], 2 ** self.sd[ny][nx]
        else:
            s = np.arange(q)
            br, pd = int(self.lr[ny, nx]), int(self.ld[ny, nx])
            rmap = s % br if br > 1 else np.zeros(q, dtype=int)
            dmap = s % pd if pd > 1 else np.zeros(q, dtype=int)
        return Es, np.ascontiguousarray(E1), np.ascontiguousarray(E4), dmap, rmap, int(pd), int(br)

    def _peps_factors_dev(self, cells):
        """[(F, dmap, rmap, pd, br)] as device tensors for a list of cells (K7, tn_peps_factor).  The seven small float tables of ALL the
        cells travel in ONE host-to-device copy and their index maps in another (nine separate copies per cell cost the sweep 45 ms of
        host time per chain in round 3; two per cell still 11 ms); the kernels run per cell on views of the two buffers."""
        keep = getattr(self, '_factor_keep', None)                 # set for the duration of one search_ground_state call
        out = [keep.get(c) if keep is not None else None for c in cells]
        todo = [i for i, o in enumerate(out) if o is None]
        if not todo:
            return out
        fparts, iparts, meta = [], [], []
        for i in todo:
            ny, nx = cells[i]
            Es, E1, E4, dmap, rmap, pd, br = self._site_tables(ny, nx)
            nl, nu = E1.shape[1], E4.shape[1]
            parts = [np.ravel(Es), np.ravel(E1), np.ravel(E4), np.ravel(self.Xu[ny][nx][:nu]), np.ravel(self.Xl[ny][nx][:nl]),
                     np.ravel(self.Xr[ny][nx]), np.ravel(self.Xd[ny][nx])]
            sizes = [int(x.size) for x in parts]
            for x, n in zip(parts, sizes):                      # every table stays 16-byte aligned inside the packed buffer
                fparts.append(np.asarray(x, dtype=np.float64))
                if n % 2:
                    fparts.append(np.zeros(1))
            q = int(np.size(Es))
            iparts.append(np.asarray(dmap, dtype=np.int32))
            iparts.append(np.asarray(rmap, dtype=np.int32))
            meta.append((i, sizes, q, nl, nu, pd, br))
        dev = torch.as_tensor(np.concatenate(fparts)).cuda()
        maps = torch.as_tensor(np.concatenate(iparts)).cuda()
        off = ioff = 0
        for i, sizes, q, nl, nu, pd, br in meta:
            views = []
            for n in sizes:
                views.append(dev[off:off + n])
                off += n + (n % 2)
            dm, rm = maps[ioff:ioff + q], maps[ioff + q:ioff + 2 * q]
            ioff += 2 * q
            F = ops.peps_factor(views[0], views[1].view(q, nl), views[2].view(q, nu), views[3], views[4], views[5], views[6], dm, rm)
            out[i] = (F, dm, rm, pd, br)
            if keep is not None:
                keep[cells[i]] = out[i]
        return out

    def _peps_factor_dev(self, ny, nx):
        """(F, dmap, rmap, pd, br) of one cell as device tensors."""
        return self._peps_factors_dev([(ny, nx)])[0]

    def _mpo_site_dev(self, ny, nx):
        """Row-MPO site W[l,d,r,u] built on the device (K7, tn_mpo_from_factor)."""
        F, dm, rm, pd, br = self._peps_factor_dev(ny, nx)
        return ops.mpo_from_factor(F, dm, rm, pd, br)

    def _mpo_site(self, ny, nx):
        """W[l,d,r,u] = sum_s T[s,l,d,r,u] (tnac4o.py:1686) as a host array (host twin of _mpo_site_dev, used by tests)."""
        F, dmap, rmap, pd, br = self._peps_factor(ny, nx)
        q, nl, nu = F.shape
        W = np.zeros((pd, br, nl, nu))
        np.add.at(W, (dmap, rmap), F)                 # unbuffered, increasing s: the reference's summation order
        return np.ascontiguousarray(W.transpose(2, 0, 1, 3))

    def _row_mpo(self, ny):
        At = mps.MPO(L=self.Nx)
        for nx, (F, dm, rm, pd, br) in enumerate(self._peps_factors_dev([(ny, nx) for nx in range(self.Nx)])):
            At.set_direct(ops.mpo_from_factor(F, dm, rm, pd, br), nx)
        return At

    # ------------------------------------------------------------------------------------ sweeps (GPU)
    def _setup_rhoT(self, graduate_truncation=True, Dmax=32, tolS=1e-16, tolV=1e-10, max_sweeps=20):
        """Top boundary MPS of every row, built bottom-up (tnac4o.py:1674-1695)."""
        Ny = self.Ny
        self.rhoT = [None] * (Ny + 1)
        self.rhoT_overlap = [1] * (Ny + 1)
        self.rhoT_discarded = [0] * (Ny + 1)
        self.rhoT[Ny] = mps.MPS(d=1, L=self.Nx, Dmax=1, initial='X')
        for ny in range(Ny - 1, -1, -1):
            psi = self.rhoT[ny + 1].copy()
            self.rhoT_overlap[ny] = psi.apply_mpo_compress(self._row_mpo(ny), Hconj=True, Dmax=Dmax, tolS=tolS, tolV=tolV,
                                                           max_sweeps=max_sweeps, graduate_truncation=graduate_truncation)
            self.rhoT_discarded[ny] = max(psi.discarded)
            self.rhoT[ny] = psi

    def _setup_rhoB(self, graduate_truncation=True, Dmax=32, tolS=1e-16, tolV=1e-10, max_sweeps=20):
        """Bottom boundary MPS, built top-down (tnac4o.py:1697-1718)."""
        Ny = self.Ny
        self.rhoB = [None] * (Ny + 1)
        self.rhoB_overlap = [1] * (Ny + 1)
        self.rhoB_discarded = [0] * (Ny + 1)
        self.rhoB[0] = mps.MPS(d=1, L=self.Nx, Dmax=1, initial='X')
        for ny in range(Ny):
            psi = self.rhoB[ny].copy()
            self.rhoB_overlap[ny + 1] = psi.apply_mpo_compress(self._row_mpo(ny), Hconj=False, Dmax=Dmax, tolS=tolS, tolV=tolV,
                                                               max_sweeps=max_sweeps, graduate_truncation=graduate_truncation)
            self.rhoB_discarded[ny + 1] = max(psi.discarded)
            self.rhoB[ny + 1] = psi

    # ------------------------------------------------------------------------------------ preconditioning
    def precondition(self, mode='balancing', steps=2, beta_cond=(), Dmax_cond=(), max_scale=1024,
                     graduate_truncation=False, tolS=1e-16, tolV=1e-10, max_sweeps=20):
        """'balancing' gauge fix of the vertical bonds at reduced beta (tnac4o.py:342-379)."""
        if mode != 'balancing':
            return
        beta_cond = list(beta_cond) or [self.beta * 2.0 ** (n - steps) for n in range(steps)]
        Dmax_cond = list(Dmax_cond) or [8] * len(beta_cond)
        main_beta = self.beta
        for b, D in zip(beta_cond, Dmax_cond):
            self.beta = b
            self.logger.info('Preconditioning with beta = %.2f', b)
            self._update_conditioning(Dmax=D, graduate_truncation=graduate_truncation, tolS=tolS, tolV=tolV,
                                      max_sweeps=max_sweeps, max_scale=max_scale)
        self.beta = main_beta

    def _balance_site(self, B, T, ny, nx, max_scale, pending):
        """One balancing step for the vertical bond above cell (ny,nx) (tnac4o.py:1844-1867), entirely on the GPU: the
        p x p bond environment, its dgebal scaling clamped to [1/max_scale, max_scale] (tn_balance), the two overlaps.
        Nothing is read back here: the scale vector and the overlaps are queued in `pending` and folded into the host-side
        gauge tables Xd / Xu and the diagnostics once per conditioning pass (_flush_balance)."""
        sc = ops.balance(B.bond_env_mix(T, nx), max_scale)
        nrm = torch.linalg.vector_norm
        o1 = B.expectation_mix_dev(T, nx) * torch.reciprocal(nrm(B.A[nx]) * nrm(T.A[nx]))
        B.scale_site_(nx, sc)
        T.scale_site_(nx, sc, inv=True)                     # powers of two: dividing is exactly multiplying by 1/sc
        o2 = B.expectation_mix_dev(T, nx) * torch.reciprocal(nrm(B.A[nx]) * nrm(T.A[nx]))
        pending.append((ny, nx, sc, o1.reshape(1), o2.reshape(1)))

    def _flush_balance(self, pending, overlaps):
        """One device-to-host copy for a whole conditioning pass, then the reference's bookkeeping in its order
        (tnac4o.py:1857-1865)."""
        if not pending:
            return
        flat = torch.cat([torch.cat([sc, o1, o2]) for (_, _, sc, o1, o2) in pending]).cpu().numpy()
        off = 0
        for (ny, nx, sc, _, _) in pending:
            k = sc.numel()
            scale, o1, o2 = flat[off:off + k], float(flat[off + k]), float(flat[off + k + 1])
            off += k + 2
            if o1 < overlaps[0, ny - 1]:
                overlaps[0, ny - 1] = o1
                overlaps[1, ny - 1] = max(o1, o2)
            kk = self.ld[ny - 1, nx]
            self.Xd[ny - 1, nx, :kk] *= scale
            self.Xu[ny, nx, :kk] *= 1 / scale

    def _update_conditioning(self, graduate_truncation=False, Dmax=8, tolS=1e-16, tolV=1e-10, max_sweeps=4,
                             max_scale=1024):
        """tnac4o.py:1824-1918 ('ud' direction; the 'lr' branch is dead code in the reference)."""
        max_scale = 2.0 ** np.floor(np.log2(np.sqrt(max_scale)))
        kw = dict(graduate_truncation=graduate_truncation, Dmax=Dmax, tolS=tolS, tolV=tolV, max_sweeps=max_sweeps)
        self._setup_rhoT(**kw)
        self._setup_rhoB(**kw)
        overlaps = np.ones((2, self.Ny - 1))
        Nx = self.Nx
        pending = []

        def renorm(B, k):                      # R *= 1 / ||R||  with the norm kept on the device
            B.R[k] = B.R[k] * torch.reciprocal(torch.linalg.vector_norm(B.R[k]))
        for ny in range(1, self.Ny):
            B, T = self.rhoB[ny], self.rhoT[ny]
            for nx in range(Nx):
                B.update_RL_mix(T, nx, keep_on_device=True)
                renorm(B, nx + 1)            # for nx = Nx-1 this touches the unused 1x1 slot R[Nx], as in the reference
            for nx in range(Nx - 1, -1, -1):
                self._balance_site(B, T, ny, nx, max_scale, pending)
                if nx > 0:
                    B.orth_right(nx)
                    B.attach_AC()
                    T.orth_right(nx)
                    T.attach_AC()
                    B.update_RR_mix(T, nx)
                    renorm(B, nx)
            for nx in range(Nx):
                self._balance_site(B, T, ny, nx, max_scale, pending)
                if nx < Nx - 1:
                    B.orth_left(nx)
                    B.attach_CA()
                    T.orth_left(nx)
                    T.attach_CA()
                    B.update_RL_mix(T, nx)
                    renorm(B, nx + 1)
        self._flush_balance(pending, overlaps)
        self.overlaps_ud = np.vstack([self.overlaps_ud, overlaps])
        self.rhoB = []

    # ------------------------------------------------------------------------------------ search (GPU + host)
    def _setup_RR(self, vind, ny):
        """Right environments for every distinct boundary-index suffix of the beam (tnac4o.py:1768-1784).

        Returns a list over levels j = 0 .. Nx-1 (level j belongs to site nx = Nx - j): (keys, RR) with keys the
        sorted unique suffixes vind[:, nx+1:] and RR a device tensor (nkeys, Dl(nx), bl(nx))."""
        top = self.rhoT[ny + 1]
        dev = top.A[0].device
        levels = [(np.zeros((1, 0), dtype=vind.dtype), torch.ones((1, 1, 1), dtype=torch.float64, device=dev))]
        for nx in range(self.Nx - 1, 0, -1):
            keys, _ = _unique_rows(vind[:, nx + 1:])
            pkeys, prr = levels[-1]
            _, pinv = _unique_rows(np.vstack([pkeys, keys[:, 1:]]))          # parent rows: match suffix[1:] to pkeys
            parent = pinv[len(pkeys):]
            # pkeys are sorted-unique, so their own inverse is the identity: parent indexes rows of prr directly
            W = self._mpo_site_dev(ny, nx)                                   # (bl, p, br, pu)
            bl, p, br, pu = W.shape
            A = top.A[nx]
            Dl, _, Dr = A.shape
            if Dl * bl <= 2048:              # K9: gather + both contractions + nfactor in one launch
                RR = ops.env_rr(A.contiguous(), prr, W, _dev_i32(parent), _dev_i32(keys[:, 0]))
            else:                            # very wide bonds: the same contraction as two batched GEMMs
                RRg = prr[torch.as_tensor(parent, device=dev)]                   # (nk, Dr, br)
                T = ops.bmm(A.view(1, Dl * p, Dr), RRg)                          # (nk, Dl p, br)
                Wt = W.permute(3, 1, 2, 0).reshape(pu, p * br, bl).contiguous()
                Wsel = Wt[torch.as_tensor(keys[:, 0].astype(np.int64), device=dev)]
                RR = ops.bmm(T.view(-1, Dl, p * br), Wsel)                       # (nk, Dl, bl)
                ops.nfactor_batched_(RR)
            levels.append((keys, RR))
        return levels

    def _setup_rhoT_shared(self, group, **kw):
        """The sweep on the first rank of `group`, then rhoT (and its diagnostics) broadcast to the partners
        (the sweep is a sequential chain; only the beam is split)."""
        import torch.distributed as dist
        from . import parallel
        owner = dist.get_rank(group) == 0
        if owner:
            self._setup_rhoT(**kw)
        rows = parallel.broadcast_site_tensors([m.A for m in self.rhoT] if owner else None, group)
        diag = parallel.broadcast_object((list(self.rhoT_overlap), list(self.rhoT_discarded)) if owner else None, group)
        if not owner:
            self.rhoT = []
            dev = torch.device('cuda', torch.cuda.current_device())
            for A in rows:
                m = mps.MPS(d=1, L=self.Nx, Dmax=1, initial='X')
                A = [torch.as_tensor(a, dtype=torch.float64).to(dev) for a in A]      # gloo hands back host arrays
                m.A = A
                m.D = [int(A[0].shape[0])] + [int(a.shape[2]) for a in A]
                self.rhoT.append(m)
            self.rhoT_overlap, self.rhoT_discarded = diag

    def search_low_energy_spectrum(self, excitations_encoding=1, M=2 ** 10, relative_P_cutoff=1e-6, max_dEng=0., lim_hd=0,
                                   min_dEng=1e-12, graduate_truncation=True, Dmax=32, tolS=1e-16, tolV=1e-10,
                                   max_sweeps=20):
        """Branch-and-bound search that also records the droplets of the branches it merges away, from which the
        low-energy spectrum up to max_dEng is rebuilt by `decode_low_energy_states` (tnac4o.py:652-915, encoding 1:
        droplet independence from the row-major order of the cells).  Returns the lowest energies found; stores the
        same result attributes as search_ground_state plus the excitation forest `el` and the shape table `d`."""
        from . import droplets
        if excitations_encoding not in (1, 2, 3):
            raise ValueError('Available droplets handling strategies are excitations_encoding = 1,2,3.')
        self.excitations_encoding = excitations_encoding
        if excitations_encoding == 1:
            rec = droplets.ExcitationRecorder(max_dEng, lim_hd, self.mode)
        else:                                       # independence from the interaction graph of the (rotated) lattice
            conn = droplets.Connectivity(self.mode, self.Nx, J=self.J if self.mode == 'Ising' else None,
                                         ind=self.ind if self.mode == 'Ising' else None)
            cls = droplets.AdjacencyRecorder if excitations_encoding == 2 else droplets.FlatRecorder
            rec = cls(max_dEng, lim_hd, self.mode, conn)
        Eng = self.search_ground_state(M=M, relative_P_cutoff=relative_P_cutoff, min_dEng=min_dEng,
                                       graduate_truncation=graduate_truncation, Dmax=Dmax, tolS=tolS, tolV=tolV,
                                       max_sweeps=max_sweeps, recorder=rec)
        self.el, self.d = rec.finish(self.order_i)
        self.invd = rec.shapes.semi_hash_index()
        self.free_d = rec.shapes.next_id
        if excitations_encoding > 1:                # decoding works in the unrotated cell order (tnac4o.py:1130, 1357)
            self._conn = droplets.Connectivity(self.mode, self.Nx_model, J=self.J0 if self.mode == 'Ising' else None,
                                               ind=self.ind0 if self.mode == 'Ising' else None)
            self.adj = self._conn.adj if self.mode == 'Ising' else []
        return Eng

    def add_noise(self, amplitude=1e-7):
        """Small random perturbation of the couplings (numpy's global generator, like the reference) that removes
        accidental degeneracies before a search with excitations_encoding 2 or 3 (tnac4o.py:917-941)."""
        self.logger.info('Adding noise to the coupling with ampliture %.2e', amplitude)
        if self.mode == 'Ising':
            rows, cols = self.J.nonzero()
            self.J[rows, cols] += (np.random.rand(len(rows)) * 2 - 1) * amplitude
            self._divide_couplings()
        else:
            fun = {}
            for key, val in self.J['fun'].items():
                fun[key] = np.array(val, dtype=float, copy=True)
                if fun[key].ndim == 1:
                    fun[key] += (np.random.rand(fun[key].shape[0]) * 2 - 1) * amplitude
            self.J['fun'] = fun
            self._divide_couplings()

    def decode_low_energy_states(self, max_dEng=0., max_states=1024):
        """Turn the recorded excitation forest into explicit states, lowest energies first (tnac4o.py:1360-1389).
        Replaces energy / states by the decoded spectrum; returns the lowest excitation energy (0)."""
        from . import droplets
        enc = getattr(self, 'excitations_encoding', 1)
        if enc == 1:
            E, st = droplets.decode_states(self.states[0], self.el, self.d, self.Nx_model * self.Ny_model, max_dEng,
                                           max_states, self.indtype)
        else:
            E, st = droplets.decode_states_adjacent(self.states[0], self.el, self.d, self._conn, max_dEng, max_states,
                                                    self.indtype, one_layer=(enc == 3))
        self.energy = E + self.energy[0]
        self.states = st
        return E[0]

    def search_ground_state(self, M=2 ** 10, relative_P_cutoff=1e-6, min_dEng=1e-12, graduate_truncation=True,
                            Dmax=32, tolS=1e-16, tolV=1e-10, max_sweeps=20, trace=None, beam_group=None, recorder=None):
        """Row-major branch-and-bound for the most probable configuration (tnac4o.py:381-551).  Results are stored in
        energy, degeneracy, states, probability (log2), discarded_probability, negative_probability.
        ``recorder`` (droplets.ExcitationRecorder): told about every merge (search_low_energy_spectrum).
        ``trace`` (a list) receives (ny, nx, Pn table, minPn, vind) of every site-step when given (parity tests).
        ``beam_group`` (a torch.distributed group): the ranks of the group work on this one solve together -- the first
        computes the sweep, every site-step's branches are split between them (parallel.gather_branch_tables) and each
        rank ends with the complete, identical result."""
        from . import parallel
        self.logger.info('Searching ground state with beta = %.2f', self.beta)
        kw_sweep = dict(graduate_truncation=graduate_truncation, Dmax=Dmax, tolS=tolS, tolV=tolV, max_sweeps=max_sweeps)
        # the PEPS factor of a cell (K7) serves the row MPO of the sweep and, unchanged, the conditional tables of the search: kept
        # between the two for the duration of this call (134 MB at L = 2048) instead of being rebuilt
        self._factor_keep = {}
        try:
            return self._search_ground_state(M, relative_P_cutoff, min_dEng, kw_sweep, trace, beam_group, recorder)
        finally:
            self._factor_keep = None

    def _search_ground_state(self, M, relative_P_cutoff, min_dEng, kw_sweep, trace, beam_group, recorder):
        from . import parallel
        if beam_group is None:
            self._setup_rhoT(**kw_sweep)
        else:
            self._setup_rhoT_shared(beam_group, **kw_sweep)
        # TN_BEAM: 'device' (default) = the beam step resident on the GPU (tn_beam_search / tnac4o_amd/beam.py); 'host' = the same canonical order
        # with the bookkeeping in numpy (used whenever a droplet recorder or a trace wants the intermediate tables on the host);
        # 'numpy' = the host path in numpy's own (unspecified) argpartition / argsort order, as the reference happens to run
        beam_mode = os.environ.get('TN_BEAM', 'device')
        if beam_mode == 'device' and recorder is None and trace is None:
            from . import beam
            # the whole loop in the library: tn_beam_search for one rank on the rotation, tn_beam_search_team for a beam group (the
            # conditional tables of a site-step split over its ranks, completed through torch.distributed).  A site-step at which NO candidate passes the cut-off (every log2 p is -inf or
            # NaN: a degenerate contraction) is not handled on the device: the search is redone on the host path, which keeps the single
            # best candidate there like the reference's keep = max(count, 1) (tnac4o.py:460-462).
            try:
                if beam.NATIVE_BEAM and not (beam_group is not None and os.environ.get('TN_BEAM_TEAM', 'native') == 'torch'):
                    # (a beam group walks the search in the library too: tn_beam_search_team with the site-steps' conditional tables
                    #  split over the ranks; TN_BEAM_TEAM=torch keeps the torch driver with its pruned candidate exchange)
                    E = beam.search_native(self, M, relative_P_cutoff, min_dEng, beam_group=beam_group)
                    if E is not None:
                        return E
                return beam.search_device(self, M, relative_P_cutoff, min_dEng, beam_group=beam_group)
            except beam.NoCandidate:
                if beam_group is not None:
                    raise
                self.logger.warning('beam search: no candidate passed the cut-off at some site; redoing the search on the host path')
        canonical = beam_mode != 'numpy'
        Nx, Ny = self.Nx, self.Ny
        vind = np.zeros((1, Nx + 1), dtype=self.indtype)
        states = np.zeros((1, Nx * Ny), dtype=self.indtype)
        Eng, prob, deg = np.zeros(1), np.zeros(1), np.ones(1, dtype=int)
        pd_max, globalmin = -np.inf, 0.0
        dev = self.rhoT[0].A[0].device

        for ny in range(Ny):
            self.logger.info('Row %d / %d', ny + 1, Ny)
            levels = self._setup_RR(vind, ny)
            top = self.rhoT[ny + 1]
            pkeys = np.zeros((1, 0), dtype=vind.dtype)                       # distinct prefixes, sorted
            RL = torch.ones((1, 1), dtype=torch.float64, device=dev)         # (nprefix, Dl)
            for nx in range(Nx):
                q, nb = int(self.N[ny][nx]), prob.size
                F, dmap, rmap, _, _ = self._peps_factor_dev(ny, nx)
                AT = top.A[nx]
                Dl, p, Dr = AT.shape
                # every prefix's left environment through the top site in one GEMM: T1[prefix, d, chi']
                T1 = ops.mm(RL, AT.view(Dl, p * Dr)).view(-1, p, Dr)
                _, pref = _unique_rows(np.vstack([pkeys, vind[:, :nx]]))
                pref = pref[len(pkeys):]
                skeys, RR = levels[Nx - nx - 1]
                _, suf = _unique_rows(np.vstack([skeys, vind[:, nx + 2:]]))
                suf = suf[len(skeys):]
                def pn_slice(lo, hi):                                        # K8 on the branches lo..hi-1
                    if not canonical:
                        return ops.calc_pn(T1, RR, F, dmap, rmap, _dev_i32(pref[lo:hi]), _dev_i32(suf[lo:hi]),
                                           _dev_i32(vind[lo:hi, nx]), _dev_i32(vind[lo:hi, nx + 1]))
                    # canonical order: the expanded log-probabilities come from the same launch (the device's log2, so that
                    # this path and tnac4o_amd.beam see the same bits); they ride behind the table
                    P, mP_, LP = ops.calc_pn(T1, RR, F, dmap, rmap, _dev_i32(pref[lo:hi]), _dev_i32(suf[lo:hi]), _dev_i32(vind[lo:hi, nx]),
                                             _dev_i32(vind[lo:hi, nx + 1]), parent_log2p=_dev_f64(prob[lo:hi]))
                    return torch.cat([P, LP], dim=1), mP_
                newprob, mP = parallel.gather_branch_tables(pn_slice, nb, q if not canonical else 2 * q, beam_group)
                minprob = float(mP.min())
                if canonical:
                    newprob, logp = np.ascontiguousarray(newprob[:, :q]), np.ascontiguousarray(newprob[:, q:])
                if trace is not None:
                    trace.append((ny, nx, newprob.copy(), mP.copy(), vind.copy()))

                if canonical:
                    prob = logp.reshape(nb * q)
                else:
                    with np.errstate(divide='ignore'):
                        newprob = np.log2(newprob)
                    newprob += prob[:, None]
                    prob = newprob.reshape(nb * q)

                order = np.arange(prob.size)
                if relative_P_cutoff > 0:                                    # tnac4o.py:458-465
                    cutoff = np.max(prob) + np.log2(relative_P_cutoff)
                    keep = max(int((prob > cutoff).sum()), 1)
                    if keep < prob.size:
                        if canonical:                                        # ascending flat index; the largest value cut
                            kept = prob > cutoff
                            if not kept.any():                               # all -inf / NaN: the single best, as keep = 1 does in the reference
                                kept[int(np.argmax(prob))] = True
                            order = np.flatnonzero(kept)
                            pd_max = max(pd_max, float(np.max(prob[~kept])))
                        else:
                            order = prob.argpartition(-keep - 1)
                            pd_max = max(pd_max, prob[order[-keep - 1]])
                            order = order[-keep:]
                        prob = prob[order]

                inds, indc = order // q, np.mod(order, q)                    # tnac4o.py:469-478
                states = states[inds]
                states[:, ny * Nx + nx] = indc
                vind = vind[inds]
                deg = deg[inds]
                vind[:, nx] = self._ind_bond_down(indc, ny, nx)
                vind[:, nx + 1] = self._ind_bond_right(indc, ny, nx)
                Eng = Eng[inds]
                Eng += self._update_Eng(states, ny, nx)

                vindn, inv = _unique_rows(vind)                              # merge equal boundaries (:481-515)
                indn, degn, probn, gorder, gstarts = _merge_groups(inv, Eng, prob, deg, min_dEng, canonical=canonical)
                sel = None
                if probn.size > M:                                           # keep the M most probable (:518-526)
                    if canonical:                                            # ties to the smaller group index; survivors in group order
                        srt = np.argsort(-probn, kind='stable')
                        pd_max = max(pd_max, probn[srt[M]])
                        sel = np.sort(srt[:M])
                    else:
                        sel = probn.argpartition(-M - 1)
                        pd_max = max(pd_max, probn[sel[-M - 1]])
                        sel = sel[-M:]
                if recorder is not None:                                     # droplets of the merged-away branches
                    recorder.merge_step(ny * Nx + nx, inds, gorder, gstarts, Eng, prob, states, indn, probn,
                                        np.arange(probn.size) if sel is None else sel)
                vind, prob, deg = vindn, probn, degn
                states, Eng = states[indn], Eng[indn]
                if sel is not None:
                    vind, states, prob, Eng, deg = vind[sel], states[sel], prob[sel], Eng[sel], deg[sel]

                # left environments of the new distinct prefixes: rows of T1 (tnac4o.py:528-535)
                nkeys, _ = _unique_rows(vind[:, :nx + 1])
                _, par = _unique_rows(np.vstack([pkeys, nkeys[:, :nx]]))
                par = par[len(pkeys):]
                RL = ops.env_rl(T1, _dev_i32(par), _dev_i32(nkeys[:, nx]))
                pkeys = nkeys
                globalmin = min(globalmin, minprob)

            if recorder is not None and hasattr(recorder, 'end_row'):
                recorder.end_row()
            vind[:, 1:] = vind[:, :-1]                                       # tnac4o.py:540-542
            vind[:, 0] = 0

        self.energy = Eng
        self.degeneracy = deg[0]
        self.states = states[:, self.order]
        self.probability = prob
        self.discarded_probability = pd_max
        self.negative_probability = min(globalmin, 0)
        return Eng

    def gibbs_sampling(self, M=2 ** 10, graduate_truncation=True, Dmax=32, tolS=1e-15, tolV=1e-10, max_sweeps=20):
        """Draw M configurations from the Boltzmann distribution, cell by cell from the conditional probabilities of the
        boundary-MPS contraction (tnac4o.py:553-650).  Uses numpy's global generator like the reference (np.random.rand,
        one vector of M numbers per cell), so a seeded run draws the same configurations.  Stores energy (M,), states
        (M, Nx*Ny), negative_probability; returns the sampled energies."""
        self.logger.info('Sampling with beta = %.2f', self.beta)
        self._setup_rhoT(graduate_truncation=graduate_truncation, Dmax=Dmax, tolS=tolS, tolV=tolV, max_sweeps=max_sweeps)
        Nx, Ny = self.Nx, self.Ny
        vind = np.zeros((M, Nx + 1), dtype=np.int64)           # plain ints here, as in the reference (:584-585)
        states = np.zeros((M, Nx * Ny), dtype=np.int64)
        Eng = np.zeros(M)
        globalmin = 1.0
        dev = self.rhoT[0].A[0].device
        for ny in range(Ny):
            self.logger.info('Row %d / %d', ny + 1, Ny)
            levels = self._setup_RR(vind, ny)
            top = self.rhoT[ny + 1]
            pkeys = np.zeros((1, 0), dtype=vind.dtype)
            RL = torch.ones((1, 1), dtype=torch.float64, device=dev)
            for nx in range(Nx):
                q = int(self.N[ny][nx])
                F, dmap, rmap, _, _ = self._peps_factor_dev(ny, nx)
                AT = top.A[nx]
                Dl, p, Dr = AT.shape
                T1 = ops.mm(RL, AT.view(Dl, p * Dr)).view(-1, p, Dr)
                # distinct boundary configurations only (the reference's `seen` dictionary, :601-612)
                uvind, uinv = _unique_rows(vind)
                _, pref = _unique_rows(np.vstack([pkeys, uvind[:, :nx]]))
                pref = pref[len(pkeys):]
                skeys, RR = levels[Nx - nx - 1]
                _, suf = _unique_rows(np.vstack([skeys, uvind[:, nx + 2:]]))
                suf = suf[len(skeys):]
                P, mP = ops.calc_pn(T1, RR, F, dmap, rmap, _dev_i32(pref), _dev_i32(suf), _dev_i32(uvind[:, nx]),
                                    _dev_i32(uvind[:, nx + 1]))
                newprob = P.cpu().numpy()[uinv]
                minprob = float(mP.min().item())
                newprob = newprob.cumsum(axis=1)                             # :616-622
                rr = np.random.rand(M)
                indc = np.array([np.searchsorted(newprob[kk], rr[kk]) for kk in range(M)], dtype=np.int64)
                states[:, ny * Nx + nx] = indc
                vind[:, nx] = self._ind_bond_down(indc, ny, nx)
                vind[:, nx + 1] = self._ind_bond_right(indc, ny, nx)
                Eng += self._update_Eng(states, ny, nx)
                nkeys, _ = _unique_rows(vind[:, :nx + 1])                    # left environments (:628-636)
                _, par = _unique_rows(np.vstack([pkeys, nkeys[:, :nx]]))
                par = par[len(pkeys):]
                RL = ops.env_rl(T1, _dev_i32(par), _dev_i32(nkeys[:, nx]))
                pkeys = nkeys
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

    # ------------------------------------------------------------------------------------ output
    def binary_states(self, number=-1):
        """Bit strings: 1 spin up, 0 spin down, 2 inactive (tnac4o.py:261-288)."""
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
                out[:, act] = (1 - _bits(len(act)))[self.states[:ns, k]]
        return out

    def save(self, file_name):
        """Save the solution to a .npy file readable by `load` here and by the reference's `tnac4o.load`
        (tnac4o.py:200-231: a pickled dict with these keys)."""
        d = {'mode': self.mode, 'rotation': self.rotation, 'energy': self.energy, 'probability': self.probability,
             'degeneracy': self.degeneracy, 'states': self.states, 'discarded_probability': self.discarded_probability,
             'negative_probability': self.negative_probability, 'Nx': self.Nx_model, 'Ny': self.Ny_model, 'Nc': self.Nc,
             'beta': self.beta}
        if self.mode == 'Ising':
            d['ind'] = self.ind0
        if hasattr(self, 'excitations_encoding'):
            for k in ('excitations_encoding', 'd', 'invd', 'el', 'free_d'):
                d[k] = getattr(self, k)
            if self.excitations_encoding > 1 and self.mode == 'Ising':
                import scipy.sparse
                d['adj'] = scipy.sparse.csr_matrix(self.adj)
        np.save(file_name, d)

    def show_properties(self):
        """Print the lattice size and inverse temperature (what tnac4o.py:233-241 reports)."""
        for label, value in (('L', self.L), ('Ny', self.Ny), ('Nx', self.Nx), ('Beta', self.beta)):
            print('%-7s %s' % (label + ':', value))

    def show_solution(self, state=False):
        """Print a summary of the stored result; with state=True also the best configuration (tnac4o.py:244-259)."""
        if len(self.energy) == 0:
            print('No solution to show.')
            return
        rows = [('Energy', '%4.6f' % self.energy[0]), ('Degeneracy', '%2d' % self.degeneracy),
                ('log2(Probability)', '%0.2e' % self.probability[0]), ('Discarded log2(P)', '%0.2e' % self.discarded_probability),
                ('Min P (err)', '%0.2e' % self.negative_probability), ('# of states', '%1d' % len(self.energy)),
                ('Rotation/direction', '%1d' % self.rotation)]
        width = max(len(k) for k, _ in rows)
        for k, v in rows:
            print('%s : %s' % (k.ljust(width), v))
        if state:
            print(self.states[0])
