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
the branch records, energies (table look-ups), unique of the boundary rows through one int64 rank key (torch.unique), the per-group merge
(tn_merge_groups: fixed summation order), top-M (stable sort) and the environment bookkeeping all stay on the GPU; the host learns two
integers per site-step (kept candidates, groups).  With a beam group (several ranks on one rotation) every rank evaluates the tables
of its slice of the branches, prunes them against the global maximum and only the surviving (index, log2 p) pairs travel.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import ops


class NoCandidate(RuntimeError):
    """No candidate of a site-step passes the cut-off (every log2 p is -inf or NaN): the caller redoes the search on the host path."""


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

    B = maxidx + 1                                                      # radix of a boundary index

    def unique_keys(key):
        """Sorted unique values of an int64 key vector: (number of groups, inverse, first member of every group)."""
        u, inv = torch.unique(key, sorted=True, return_inverse=True)
        ng = int(u.shape[0])
        first = torch.full((ng,), key.numel(), dtype=i64, device=dev)
        first.scatter_reduce_(0, inv, torch.arange(key.numel(), dtype=i64, device=dev), reduce='amin', include_self=True)
        return ng, inv, first

    for ny in range(Ny):
        solver.logger.info('Row %d / %d', ny + 1, Ny)
        top = solver.rhoT[ny + 1]
        # Every index row the search has to compare -- suffixes for the right environments, prefixes for the left ones, whole rows for
        # the merge -- is compared through ONE int64 key built from order-preserving ranks: a suffix vind[:, c:] is (vind[:, c], rank of
        # vind[:, c+1:]), a prefix vind[:, :c+1] is (rank of vind[:, :c], vind[:, c]), a whole row is (prefix rank, down index, right index,
        # suffix rank).  Sorting the keys sorts the rows lexicographically (what np.unique(axis=0) does on the host path), without packing
        # or multi-word sorts.
        # ---- right environments of every distinct suffix (tnac4o._setup_RR, tnac4o.py:1768-1784); sufidx[j] = rank of the branch's
        # suffix vind[:, Nx-j+1:] among the keys of level j (level j serves site nx = Nx-1-j)
        nb = prob.numel()
        site = {}

        def cell(nx):                                                   # F, dmap, rmap and the MPO site of a cell, built once per row
            if nx not in site:
                F, dm, rm, pd, br = solver._peps_factor_dev(ny, nx)
                site[nx] = (F, dm, rm, pd, br)
            return site[nx]
        RRs = [torch.ones((1, 1, 1), dtype=f64, device=dev)]
        sufidx = [torch.zeros(nb, dtype=i64, device=dev)]
        nkeys_prev = 1
        for nx in range(Nx - 1, 0, -1):
            nk, inv, first = unique_keys(vind[:, nx + 1] * nkeys_prev + sufidx[-1])
            parent = sufidx[-1][first]                                  # the key's own suffix [1:] in the previous level
            uidx = vind[first, nx + 1]
            F, dm, rm, pd, br_ = cell(nx)
            W = ops.mpo_from_factor(F, dm, rm, pd, br_)
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
            nkeys_prev = nk
        nsuf = [int(r.shape[0]) for r in RRs]
        sufmat = torch.stack(sufidx, dim=1)                             # (nb, Nx)
        pref = torch.zeros(nb, dtype=i64, device=dev)
        RL = torch.ones((1, 1), dtype=f64, device=dev)
        for nx in range(Nx):
            tb = tab(ny, nx)
            q, nb = tb.q, prob.numel()
            pos = ny * Nx + nx
            F, dmap, rmap, _, _ = cell(nx)
            AT = top.A[nx]
            Dl, p, Dr = AT.shape
            T1 = ops.mm(RL, AT.view(Dl, p * Dr)).view(-1, p, Dr)
            lvl = Nx - nx - 1
            RR = RRs[lvl]
            suf = sufmat[:, lvl]
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
            if keep == 0:
                raise NoCandidate('no candidate passes the cut-off at site (%d, %d)' % (ny, nx))
            if keep < total:
                pd_max = torch.maximum(pd_max, rest_max)
            parent = torch.div(idx, q, rounding_mode='floor')
            child = idx - parent * q
            st_c = states[parent]
            st_c[:, pos] = child.to(torch.int16)
            down_c, right_c = tb.down[child], tb.right[child]
            dE = 1.0 * tb.Es[child]
            if nx > 0:
                left = st_c[:, pos - 1].to(i64)
                dE = dE + tb.E1[child, tab(ny, nx - 1).right[left] if solver.mode == 'Ising' else left]
            if ny > 0:
                up = st_c[:, pos - Nx].to(i64)
                dE = dE + tb.E4[child, tab(ny - 1, nx).down[up] if solver.mode == 'Ising' else up]
            E_c = Eng[parent] + dE
            # ---- merge of equal boundary rows (tnac4o.py:481-515): row = (prefix, down, right, suffix)
            pref_c = pref[parent]
            pkey = pref_c * B + down_c                                   # the new prefix vind[:, :nx+1] (ranks of the old one x radix)
            key = (pkey * B + right_c) * nsuf[lvl] + suf[parent]
            ng, inv, _ = unique_keys(key)
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
            prep = parent[rep]
            vind = vind[prep]
            vind[:, nx] = down_c[rep]
            vind[:, nx + 1] = right_c[rep]
            states, Eng, prob, deg = st_c[rep], E_c[rep], lpn, degn
            sufmat = sufmat[prep]
            # ---- left environments of the new distinct prefixes: rows of T1 (tnac4o.py:528-535)
            _, ninv, nfirst = unique_keys(pkey[rep])
            par = pref_c[rep][nfirst]
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


# ---------------------------------------------------------------------------------------------------------------- native driver
class _Cell(C.Structure):
    """tn_beam_cell of include/tnpeps.h."""
    _fields_ = ([(n, C.c_void_p) for n in ('F', 'dmap', 'rmap', 'down', 'right', 'Es', 'E1', 'E4', 'left_map', 'up_map', 'A')]
                + [(n, C.c_int64) for n in ('q', 'nl', 'nu', 'pd', 'br', 'e1cols', 'e4cols', 'Dl', 'p', 'Dr')])


NATIVE_BEAM = os.environ.get('TN_NATIVE_BEAM', '1') != '0'      # TN_NATIVE_BEAM=0: the torch driver above (search_device)


_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int)


def _team_exchange(ws, group):
    """The exchange function tn_beam_search_team calls once per site-step (include/tnpeps.h): every rank of the team has filled its
    contiguous slice of the log2 p table and of the minima; one broadcast per rank and array over the team's communicator (RCCL on
    GPUs, gloo in the rehearsals) completes them everywhere.  The arrays live in the search's workspace `ws` (a torch byte buffer):
    the pointers are turned back into views of it, so the collectives see ordinary tensors and order themselves behind the
    launches the library has enqueued on the current stream."""
    import torch.distributed as dist
    base = ws.data_ptr()
    ranks = dist.get_process_group_ranks(group)
    err = []

    def exchange(ctx, lp, mp, nb, q, rank, team):
        try:
            lpt = ws[lp - base:lp - base + nb * q * 8].view(torch.float64)
            mpt = ws[mp - base:mp - base + nb * 8].view(torch.float64)
            for r in range(team):
                lo, hi = nb * r // team, nb * (r + 1) // team
                if hi > lo:
                    dist.broadcast(lpt[lo * q:hi * q], src=ranks[r], group=group)
                    dist.broadcast(mpt[lo:hi], src=ranks[r], group=group)
            return 0
        except Exception as e:                               # noqa: BLE001 -- must not propagate through the C frames
            err.append(e)
            return 1
    return _EXCHANGE_FN(exchange), err


def search_native(solver, M, relative_P_cutoff, min_dEng, beam_group=None):
    """search_ground_state's loop over rows and sites in ONE library call (tn_beam_search, csrc/beamsearch.hip): this function only
    builds the per-cell tables (the PEPS factors on the device, the index and energy tables of tnac4o._update_Eng) and hands over
    pointers.  Same canonical order and the same arithmetic as search_device, hence the same results bit for bit.  Returns None when
    a site does not fit the library's walk (tn_env_rr_batched holds Dl x (left PEPS bond) <= 2048 accumulators): the caller then takes
    search_device.  beam_group (a torch.distributed group whose ranks all hold rhoT): the team form, tn_beam_search_team -- every
    rank walks the whole search, the conditional tables of a site-step are split over the ranks and completed by _team_exchange;
    every rank returns the same result."""
    from ._lib import lib
    Nx, Ny = solver.Nx, solver.Ny
    dev = solver.rhoT[0].A[0].device
    maxidx = int(max(np.max(solver.ld), np.max(solver.lr), 2)) - 1
    ising = solver.mode == 'Ising'
    tabs = {(ny, nx): SiteTables(solver, ny, nx, dev) for ny in range(Ny) for nx in range(Nx)}
    cells = (_Cell * (Nx * Ny))()
    keep = []                                                            # every tensor a pointer was taken from
    qmax = max_env = max_t1 = max_w = 1
    for ny in range(Ny):
        top = solver.rhoT[ny + 1]
        for nx in range(Nx):
            tb = tabs[(ny, nx)]
            F, dm, rm, pd, br = solver._peps_factor_dev(ny, nx)
            A = top.A[nx].contiguous()
            keep += [F, dm, rm, A]
            q, nl, nu = F.shape
            Dl, p, Dr = A.shape
            c = cells[ny * Nx + nx]
            c.F, c.dmap, c.rmap = F.data_ptr(), dm.data_ptr(), rm.data_ptr()
            c.down, c.right, c.Es = tb.down.data_ptr(), tb.right.data_ptr(), tb.Es.data_ptr()
            c.E1 = tb.E1.data_ptr() if nx > 0 else None
            c.E4 = tb.E4.data_ptr() if ny > 0 else None
            c.left_map = tabs[(ny, nx - 1)].right.data_ptr() if (ising and nx > 0) else None
            c.up_map = tabs[(ny - 1, nx)].down.data_ptr() if (ising and ny > 0) else None
            c.A = A.data_ptr()
            c.q, c.nl, c.nu, c.pd, c.br = q, nl, nu, pd, br
            c.e1cols = tb.E1.shape[1] if tb.E1.dim() == 2 else 1
            c.e4cols = tb.E4.shape[1] if tb.E4.dim() == 2 else 1
            c.Dl, c.p, c.Dr = Dl, p, Dr
            if Dl * nl > 2048 or q > 32767:
                return None
            qmax, max_env = max(qmax, q), max(max_env, Dl * nl, Dr * br)
            max_t1, max_w = max(max_t1, p * Dr), max(max_w, nl * pd * br * nu)
    L = lib()
    wsb = int(L.tn_beam_search_ws_bytes(Nx, Ny, M, qmax, max_env, max_t1, max_w))
    ws = ops.workspace(wsb, 3)
    states = torch.empty((M, Nx * Ny), dtype=torch.int16, device=dev)
    Eng = torch.empty(M, dtype=torch.float64, device=dev)
    prob = torch.empty(M, dtype=torch.float64, device=dev)
    deg = torch.empty(M, dtype=torch.int64, device=dev)
    nb, pdm, gmin = C.c_int64(0), C.c_double(0.0), C.c_double(0.0)
    has_cut = relative_P_cutoff > 0
    team, rank, fn, ferr = 1, 0, None, []
    if beam_group is not None:
        import torch.distributed as dist
        team, rank = dist.get_world_size(beam_group), dist.get_rank(beam_group)
        if team > 1:
            fn, ferr = _team_exchange(ws, beam_group)
    rc = L.tn_beam_search_team(Nx, Ny, C.cast(cells, C.c_void_p), M, 1 if has_cut else 0, float(np.log2(relative_P_cutoff)) if has_cut else 0.0,
                               float(min_dEng), maxidx + 1, states.data_ptr(), Eng.data_ptr(), prob.data_ptr(), deg.data_ptr(), C.byref(nb),
                               C.byref(pdm), C.byref(gmin), ws.data_ptr(), wsb, ops._stream(), rank, team,
                               C.cast(fn, C.c_void_p) if fn is not None else None, None)
    if ferr:
        raise ferr[0]
    if rc == -6:                                            # no candidate passed the cut-off at some site (include/tnpeps.h)
        del keep
        raise NoCandidate('tn_beam_search: no candidate survives the cut-off')
    ops.check(rc)
    n = int(nb.value)
    solver.energy = Eng[:n].cpu().numpy()
    solver.degeneracy = int(deg[0].item())
    solver.states = states[:n].cpu().numpy().astype(solver.indtype)[:, solver.order]
    solver.probability = prob[:n].cpu().numpy()
    solver.discarded_probability = float(pdm.value)
    solver.negative_probability = min(float(gmin.value), 0)
    del keep
    return solver.energy
