"""The beam step of search_ground_state (reference tnac4o.py:437-537) with every table resident on the GPU.

The reference expands <= M branches x q cell states, cuts at a relative probability, writes the new boundary indices and energies,
merges branches with identical boundary indices (minimum energy wins, degeneracies add up) and keeps the M most probable -- with
numpy's argpartition / unstable argsort, i.e. in an order that is deterministic but unspecified.  Here the order is CANONICAL:

  * candidates of a site-step are taken in ascending flat index f = branch * q + state (the kept ones: log2 p > max + log2(cutoff));
  * merge groups (equal boundary indices) are ordered lexicographically by their index row, members in candidate order; the
    representative is the FIRST member of minimal energy; the group's log2 p is the representative's if it is alone within min_dEng
    of the minimum, else the mean over those members added up in member order;
  * the M survivors are the M largest group log2 p, ties to the smaller group index, and stay in group order.

Any such order is a valid reading of the reference (results differ from numpy's order only when energies tie exactly inside a merge
group or probabilities tie exactly at a cut); both the device path below and the host path of tnac4o.search_ground_state implement
it, and agree bit for bit (tests/test_gpu_configs.py).

Device path: conditional tables, their log2 + parent log-probability (tn_calc_pn), maximum, cut-off and compaction (torch), gathers of
the branch records, energies (table look-ups), packed-key unique of the boundary rows (torch.unique), the per-group merge
(tn_merge_groups: fixed summation order), top-M (stable sort) and the environment bookkeeping all stay on the GPU; the host learns two
integers per site-step (kept candidates, groups).  With a beam group (several ranks on one rotation) every rank evaluates the tables
of its slice of the branches, prunes them against the global maximum and only the surviving (index, log2 p) pairs travel.
"""
import numpy as np
import torch

from . import ops


def _bits_needed(maxval):
    return max(1, int(maxval).bit_length())


class KeyPacker:
    """Rows of small non-negative integers -> rows of int64 words, order preserving (lexicographic on the columns)."""

    def __init__(self, maxval):
        self.bits = _bits_needed(maxval)
        self.per_word = max(1, 62 // self.bits)

    def pack(self, rows):
        """rows: (n, w) int64 device tensor -> (n, W) int64 (W = 0 for w = 0)."""
        n, w = rows.shape
        if w == 0:
            return rows.new_zeros((n, 0))
        words = []
        for c0 in range(0, w, self.per_word):
            blk = rows[:, c0:c0 + self.per_word]
            k = blk.shape[1]
            sh = torch.arange(k - 1, -1, -1, device=rows.device, dtype=torch.int64) * self.bits
            words.append((blk << sh[None, :]).sum(dim=1))
        return torch.stack(words, dim=1)


def unique_rows(keys):
    """Sorted unique rows of an (n, W) int64 tensor: (number of groups, inverse (n,), first member of every group (ng,)).
    Lexicographic order over the words = lexicographic order of the original index rows (np.unique(axis=0) order)."""
    n, W = keys.shape
    dev = keys.device
    if W == 0 or n == 0:
        return (1 if n else 0), torch.zeros(n, dtype=torch.int64, device=dev), torch.zeros(min(n, 1), dtype=torch.int64, device=dev)
    if W == 1:
        u, inv = torch.unique(keys[:, 0], sorted=True, return_inverse=True)
    else:
        u, inv = torch.unique(keys, dim=0, sorted=True, return_inverse=True)
    ng = int(u.shape[0])
    first = torch.full((ng,), n, dtype=torch.int64, device=dev)
    first.scatter_reduce_(0, inv, torch.arange(n, dtype=torch.int64, device=dev), reduce='amin', include_self=True)
    return ng, inv, first


class SiteTables:
    """Per-cell look-up tables on the device: bond index of a cell state towards the row below / the cell to the right, and the three
    energy tables of tnac4o._cell_energies (reference tnac4o.py:1469-1489, 1506-1558)."""

    def __init__(self, solver, ny, nx, dev):
        q = int(solver.N[ny][nx])
        st = np.arange(q)
        self.q = q
        self.down = torch.as_tensor(np.asarray(solver._ind_bond_down(st, ny, nx), dtype=np.int64)).to(dev)
        self.right = torch.as_tensor(np.asarray(solver._ind_bond_right(st, ny, nx), dtype=np.int64)).to(dev)
        Es, E1, E4 = solver._cell_energies(ny, nx)
        self.Es = torch.as_tensor(np.ascontiguousarray(Es, dtype=np.float64)).to(dev)
        self.E1 = torch.as_tensor(np.ascontiguousarray(E1, dtype=np.float64)).to(dev)
        self.E4 = torch.as_tensor(np.ascontiguousarray(E4, dtype=np.float64)).to(dev)


def _i32(t):
    return t.to(torch.int32).contiguous()


def search_device(solver, M, relative_P_cutoff, min_dEng, beam_group=None):
    """search_ground_state's loop over rows and sites (tnac4o.py:429-542) on the device, canonical order.  solver.rhoT must be set up.
    Stores the result attributes on the solver exactly as the host path does; returns the energies."""
    from . import parallel
    Nx, Ny = solver.Nx, solver.Ny
    dev = solver.rhoT[0].A[0].device
    i64, f64 = torch.int64, torch.float64
    maxidx = int(max(np.max(solver.ld), np.max(solver.lr), 2)) - 1
    packer = KeyPacker(maxidx)
    ninf = float('-inf')
    rank, world = parallel._group_info(beam_group)

    vind = torch.zeros((1, Nx + 1), dtype=i64, device=dev)
    states = torch.zeros((1, Nx * Ny), dtype=torch.int16, device=dev)
    Eng = torch.zeros(1, dtype=f64, device=dev)
    prob = torch.zeros(1, dtype=f64, device=dev)
    deg = torch.ones(1, dtype=i64, device=dev)
    pd_max = torch.full((1,), ninf, dtype=f64, device=dev)
    globalmin = torch.zeros(1, dtype=f64, device=dev)
    log_cut = float(np.log2(relative_P_cutoff)) if relative_P_cutoff > 0 else None
    tables = {}

    def tab(ny, nx):
        if (ny, nx) not in tables:
            tables[(ny, nx)] = SiteTables(solver, ny, nx, dev)
        return tables[(ny, nx)]

    for ny in range(Ny):
        solver.logger.info('Row %d / %d', ny + 1, Ny)
        top = solver.rhoT[ny + 1]
        # ---- right environments of every distinct suffix (tnac4o._setup_RR, tnac4o.py:1768-1784); sufidx[:, j] = index of the branch's
        # suffix vind[:, Nx-j+1:] among the keys of level j (level j serves site nx = Nx-1-j)
        nb = prob.numel()
        RRs = [torch.ones((1, 1, 1), dtype=f64, device=dev)]
        sufidx = [torch.zeros(nb, dtype=i64, device=dev)]
        for nx in range(Nx - 1, 0, -1):
            _, inv, first = unique_rows(packer.pack(vind[:, nx + 1:]))
            parent = sufidx[-1][first]                                  # the key's own suffix [1:] in the previous level
            uidx = vind[first, nx + 1]
            W = solver._mpo_site_dev(ny, nx)
            bl, p, br, pu = W.shape
            A = top.A[nx]
            Dl, _, Dr = A.shape
            if Dl * bl <= 2048:
                RR = ops.env_rr(A.contiguous(), RRs[-1], W, _i32(parent), _i32(uidx))
            else:
                RRg = RRs[-1][parent]
                T = ops.bmm(A.view(1, Dl * p, Dr), RRg)
                Wt = W.permute(3, 1, 2, 0).reshape(pu, p * br, bl).contiguous()
                RR = ops.bmm(T.view(-1, Dl, p * br), Wt[uidx])
                ops.nfactor_batched_(RR)
            RRs.append(RR)
            sufidx.append(inv)
        sufmat = torch.stack(sufidx, dim=1)                             # (nb, Nx)
        pref = torch.zeros(nb, dtype=i64, device=dev)
        RL = torch.ones((1, 1), dtype=f64, device=dev)
        for nx in range(Nx):
            tb = tab(ny, nx)
            q, nb = tb.q, prob.numel()
            pos = ny * Nx + nx
            F, dmap, rmap, _, _ = solver._peps_factor_dev(ny, nx)
            AT = top.A[nx]
            Dl, p, Dr = AT.shape
            T1 = ops.mm(RL, AT.view(Dl, p * Dr)).view(-1, p, Dr)
            RR = RRs[Nx - nx - 1]
            suf = sufmat[:, Nx - nx - 1]
            lo, hi = parallel.shard_range(nb, rank, world)
            if hi > lo:
                _, mP, LP = ops.calc_pn(T1, RR, F, dmap, rmap, _i32(pref[lo:hi]), _i32(suf[lo:hi]), _i32(vind[lo:hi, nx]),
                                        _i32(vind[lo:hi, nx + 1]), parent_log2p=prob[lo:hi].contiguous())
                flat = LP.view(-1)
                local_min = mP.min().reshape(1)
                local_max = flat.max().reshape(1)
            else:
                flat = torch.empty(0, dtype=f64, device=dev)
                local_min = torch.full((1,), float('inf'), dtype=f64, device=dev)
                local_max = torch.full((1,), ninf, dtype=f64, device=dev)
            if world > 1:
                local_min, local_max = parallel.allreduce_minmax(local_min, local_max, beam_group)
            globalmin = torch.minimum(globalmin, local_min)
            total = nb * q
            if log_cut is not None:
                cutoff = local_max + log_cut
                mask = flat > cutoff
                idx = mask.nonzero().squeeze(1)
                vals = flat[idx]
                rest = torch.where(mask, torch.full_like(flat, ninf), flat)
                rest_max = rest.max().reshape(1) if flat.numel() else torch.full((1,), ninf, dtype=f64, device=dev)
            else:
                idx = torch.arange(flat.numel(), dtype=i64, device=dev)
                vals = flat
                rest_max = torch.full((1,), ninf, dtype=f64, device=dev)
            idx = idx + lo * q                                           # global flat index
            if world > 1:                                                 # only the surviving (index, log2 p) pairs travel
                idx, vals, rest_max = parallel.allgather_candidates(idx, vals, rest_max, beam_group)
            keep = idx.numel()
            if keep < total:
                pd_max = torch.maximum(pd_max, rest_max)
            parent = torch.div(idx, q, rounding_mode='floor')
            child = idx - parent * q
            st_c = states[parent]
            st_c[:, pos] = child.to(torch.int16)
            vi_c = vind[parent]
            vi_c[:, nx] = tb.down[child]
            vi_c[:, nx + 1] = tb.right[child]
            dE = 1.0 * tb.Es[child]
            if nx > 0:
                left = st_c[:, pos - 1].to(i64)
                dE = dE + tb.E1[child, tab(ny, nx - 1).right[left] if solver.mode == 'Ising' else left]
            if ny > 0:
                up = st_c[:, pos - Nx].to(i64)
                dE = dE + tb.E4[child, tab(ny - 1, nx).down[up] if solver.mode == 'Ising' else up]
            E_c = Eng[parent] + dE
            # ---- merge of equal boundary rows (tnac4o.py:481-515)
            ng, inv, _ = unique_rows(packer.pack(vi_c))
            sinv, perm = torch.sort(inv, stable=True)
            counts = torch.bincount(inv, minlength=ng)
            starts = torch.cat([torch.zeros(1, dtype=i64, device=dev), torch.cumsum(counts, 0)])
            rep, degn, lpn = ops.merge_groups(E_c[perm].contiguous(), vals[perm].contiguous(), deg[parent][perm].contiguous(), perm.contiguous(),
                                              starts.contiguous(), min_dEng)
            if ng > M:                                                    # keep the M most probable (tnac4o.py:518-526)
                sv, six = torch.sort(lpn, descending=True, stable=True)
                pd_max = torch.maximum(pd_max, sv[M].reshape(1))
                sel = torch.sort(six[:M]).values
                rep, degn, lpn = rep[sel], degn[sel], lpn[sel]
            vind, states, Eng, prob, deg = vi_c[rep], st_c[rep], E_c[rep], lpn, degn
            sufmat = sufmat[parent[rep]]
            # ---- left environments of the new distinct prefixes: rows of T1 (tnac4o.py:528-535)
            _, ninv, nfirst = unique_rows(packer.pack(vind[:, :nx + 1]))
            par = pref[parent[rep]][nfirst]
            didx = vind[nfirst, nx]
            RL = ops.env_rl(T1, _i32(par), _i32(didx))
            pref = ninv
        vind = torch.cat([torch.zeros((vind.shape[0], 1), dtype=i64, device=dev), vind[:, :-1]], dim=1)      # tnac4o.py:540-542

    solver.energy = Eng.cpu().numpy()
    solver.degeneracy = int(deg[0].item())
    solver.states = states.cpu().numpy().astype(solver.indtype)[:, solver.order]
    solver.probability = prob.cpu().numpy()
    solver.discarded_probability = float(pd_max.item())
    solver.negative_probability = min(float(globalmin.item()), 0)
    return solver.energy
