"""Multi-GPU driver: the lattice rotations are independent solves (reference examples/e06:97-109 runs them in a
sequential loop), so they shard across ranks with no data-path collective; one all-gather of a < 3 KB record per
rotation merges the results (energy = min, degeneracy = max over the minimisers, as e06:107-109 does).

Second axis: inside one rotation the <= M branches of the beam are independent until the merge /
top-M of a site-step (tnac4o.py:437-537).  With `beam_shards` = B > 1 the ranks form sub-groups of B: the sub-group's
first rank computes the boundary-MPS sweep (a sequential chain, not shardable) and broadcasts rhoT to its partners; at
every site-step each rank evaluates the conditional probabilities of its contiguous slice of the branches and one
all-gather inside the sub-group rebuilds the full table, after which every rank runs the identical deterministic merge.

One process per GPU (torchrun); backend 'nccl' is RCCL over xGMI on ROCm, 'gloo' is used by the CPU tests.  The
solver is injected (`make_solver`) so that the sharding / gather / merge logic can be exercised without a GPU.
"""
import numpy as np
import torch
import torch.distributed as dist

_CHAIN_STREAMS = {}


def _make_chain_streams(n):
    """The chains' streams.  TN_CU_MASK=<k> (experiment): chain i gets a stream whose CU mask leaves out the i-th of n groups of k
    consecutive... see DESIGN.md 4.4; default: plain torch streams."""
    import ctypes as C
    import os
    spec = os.environ.get('TN_CU_MASK', '')
    if not spec:
        return [torch.cuda.Stream() for _ in range(n)]
    from ._lib import lib, check
    ncu = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    nwords = (ncu + 31) // 32
    out = []
    mode, _, arg = spec.partition(':')
    for i in range(n):
        bits = [1] * ncu
        if mode == 'block':                  # leave out a contiguous block of ncu / n CUs
            w = ncu // n
            for c in range(i * w, (i + 1) * w):
                bits[c] = 0
        elif mode == 'stride':               # leave out every n-th CU, offset i
            for c in range(i, ncu, n):
                bits[c] = 0
        elif mode == 'keep':                 # chain i keeps only `arg` per cent of the CUs, spread evenly, offset i
            frac = float(arg or 75) / 100.0
            bits = [0] * ncu
            for c in range(ncu):
                if ((c + i * 7) % 100) < frac * 100:
                    bits[c] = 1
        words = (C.c_uint32 * nwords)()
        for c, b in enumerate(bits):
            if b:
                words[c // 32] |= (1 << (c % 32))
        h = C.c_void_p()
        check(lib().tn_stream_create_masked(words, nwords, C.byref(h)))
        out.append(torch.cuda.ExternalStream(h.value))
    return out


_LAST_CALL_S = {}                   # (device, chains) -> wall time of the previous run_concurrent call (de-phased starts)


def run_concurrent(fns):
    """Run independent chains concurrently on one GPU: each callable gets its own host thread and its own HIP stream
    (one stream per rotation so that the latency-bound chains interleave on the device).  Returns the
    list of results; the first exception raised by a chain is re-raised here."""
    import threading
    n = len(fns)
    if n == 1:
        return [fns[0]()]
    # The chains own their streams for the life of the process (arenas, workspaces and torch's cached blocks are keyed by stream).
    # What decides whether 4 chains interleave (2.9 s per step) or pairs of them serialise (4.5 s) is the HARDWARE QUEUE behind each
    # stream: ROCm multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4), the legacy default stream
    # included, so with 4 chains + the default stream two chains end up on one queue (measured: 4 queues 4.48 s, 2 queues 4.65 s,
    # 8 or 16 queues 2.90 s per step; the round-2 form with fresh pool streams on every call happened to dodge the collision until
    # the pool wrapped, every 8th call).  tnac4o_amd/__init__.py therefore asks for 8 queues before the runtime starts.
    from . import ops
    import tnac4o_amd as _pkg
    if _pkg.HIP_STARTED_BEFORE_IMPORT and n > 3 and not getattr(run_concurrent, '_warned', False):
        run_concurrent._warned = True
        import warnings
        warnings.warn('tnac4o_amd was imported after the HIP runtime had started with its default of 4 hardware queues: %d interleaved chains '
                      'will share queues and run ~1.5x slower; export GPU_MAX_HW_QUEUES=8 (or import tnac4o_amd first)' % n)
    key = (torch.cuda.current_device(), n)
    streams = _CHAIN_STREAMS.get(key)
    if streams is None:
        streams = _CHAIN_STREAMS[key] = _make_chain_streams(n)
    # side streams (deferred Schmidt-value checks; tn_qr's look-ahead when enabled), taken right after the chains' own streams
    # so that the pairing with hardware queues is the same on every call (torch hands out pool streams round-robin, pool
    # stream k sits on hardware queue k mod 4): chain i's side stream is rotated by TN_AUX_ROT so that it does not share a
    # queue with its own chain
    if ops.LOOKAHEAD or ops.SCHMIDT_SIDE:
        import os
        rot = int(os.environ.get('TN_AUX_ROT', '2'))
        side = [torch.cuda.Stream() for _ in range(n)]
        for i in range(n):
            ops.register_aux_stream(streams[i], side[(i + rot) % n])
    out, err = [None] * n, [None] * n
    # The caller's pending work must be visible to the chains: wait for it on the HOST.  streams[i].wait_stream(current)
    # would record an event on the (legacy) default stream, after which every launch of the chain pays for a dependency on
    # it: 1.51 -> 1.62 s/sweep with fresh streams, 1.64 -> 2.36 with reused ones (tools/stream_regime_experiment.py).
    torch.cuda.current_stream().synchronize()

    # De-phased starts.  The chains are structurally identical and start together, so their device-filling phases (the first pass of every
    # row: half of a chain since round 5) coincide and serialise while nothing latency-bound is there to fill the gaps; started a fraction
    # of a row apart, one chain's first pass overlaps with the others' truncating passes (four chains at L = 2048: 419 -> 404 ms/sweep for
    # any offset between 5 and 30 ms).  The offset scales itself: chain i starts i x TN_STAGGER_FRAC (0.7 %) x the duration of the previous
    # call with the same number of chains -- 12 ms for the 1.7 s step of the headline workload, nothing for the first call and next to
    # nothing for small problems.  TN_STAGGER_MS=<ms> fixes the offset (0: all chains start together).
    import os as _os
    import time as _time
    env_ms = _os.environ.get('TN_STAGGER_MS')
    if env_ms is not None:
        stagger = float(env_ms) * 1e-3
    else:
        stagger = float(_os.environ.get('TN_STAGGER_FRAC', '0.007')) * _LAST_CALL_S.get(key, 0.0)
    t_call = _time.perf_counter()

    def work(i):
        try:
            if stagger > 0.0 and i:
                _time.sleep(i * stagger)
            with torch.cuda.stream(streams[i]):
                out[i] = fns[i]()
                streams[i].synchronize()
        except BaseException as e:      # noqa: BLE001 - re-raised in the caller's thread
            err[i] = e
    th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    _LAST_CALL_S[key] = _time.perf_counter() - t_call
    for e in err:
        if e is not None:
            raise e
    return out


# ---------------------------------------------------------------------------------------------- beam sharding
def shard_range(n, rank, world):
    """Contiguous slice [lo, hi) of n items owned by `rank` (the first n % world ranks get one more)."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _group_info(group):
    if group is None or not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def _comm_device(group):
    return torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')


def gather_branch_tables(compute, nb, q, group):
    """Conditional-probability tables of all nb branches from per-rank slices.

    compute(lo, hi) -> (P, minP): P (hi-lo, q) and minP (hi-lo,) as float64 torch tensors or numpy arrays, for the
    branches lo..hi-1.  Every rank of `group` gets the complete (nb, q) / (nb,) tables as numpy arrays, assembled in
    branch order, so whatever follows (cut-off, merge, top-M) is identical on all of them.  One all_gather per call."""
    rank, world = _group_info(group)
    if world == 1:
        P, mP = compute(0, nb)
        P = P.cpu().numpy() if torch.is_tensor(P) else np.asarray(P)
        mP = mP.cpu().numpy() if torch.is_tensor(mP) else np.asarray(mP)
        return P, mP
    lo, hi = shard_range(nb, rank, world)
    chunk = -(-int(nb) // world)                      # slices are padded to the largest one
    dev = _comm_device(group)
    buf = torch.zeros((chunk, q + 1), dtype=torch.float64, device=dev)
    if hi > lo:
        P, mP = compute(lo, hi)
        buf[:hi - lo, :q] = torch.as_tensor(P, dtype=torch.float64).to(dev)
        buf[:hi - lo, q] = torch.as_tensor(mP, dtype=torch.float64).to(dev)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    full = np.empty((nb, q + 1))
    for r in range(world):
        l, h = shard_range(nb, r, world)
        if h > l:
            full[l:h] = out[r][:h - l].cpu().numpy()
    return np.ascontiguousarray(full[:, :q]), np.ascontiguousarray(full[:, q])


def allreduce_minmax(mn, mx, group):
    """Global minimum of `mn` and maximum of `mx` (1-element float64 device tensors) over the ranks of `group`: one all-reduce."""
    dev = _comm_device(group)
    t = torch.cat([-mn.reshape(1), mx.reshape(1)]).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    t = t.to(mn.device)
    return (-t[0]).reshape(1), t[1].reshape(1)


def allgather_candidates(idx, vals, rest_max, group):
    """The pruned candidate exchange of a site-step: every rank contributes the candidates of ITS slice of the
    branches that survive the relative cut-off -- global flat indices `idx` (int64, ascending) and their log2 p `vals` -- plus the
    largest log2 p it cut (`rest_max`, 1 element).  Returns the concatenation in rank order (= ascending flat index, the canonical
    candidate order of tnac4o_amd.beam) and the global largest cut value, identical on every rank.  Two small collectives: the
    counts, then one padded (count + 1) x 16-byte payload per rank; nothing goes through host numpy on the nccl path."""
    rank, world = _group_info(group)
    dev = _comm_device(group)
    home = idx.device
    n = torch.tensor([idx.numel()], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    buf = torch.zeros((cap + 1, 2), dtype=torch.int64, device=dev)
    buf[0, 0] = rest_max.reshape(1).view(torch.int64).to(dev)[0]
    k = idx.numel()
    if k:
        buf[1:1 + k, 0] = idx.to(dev)
        buf[1:1 + k, 1] = vals.contiguous().view(torch.int64).to(dev)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    all_idx = torch.cat([out[r][1:1 + counts[r], 0] for r in range(world)]).to(home)
    all_vals = torch.cat([out[r][1:1 + counts[r], 1] for r in range(world)]).contiguous().view(torch.float64).to(home)
    rm = torch.stack([out[r][0, 0] for r in range(world)]).view(torch.float64).max().reshape(1).to(home)
    return all_idx, all_vals, rm


def broadcast_site_tensors(rows, group, rehearse=False):
    """Broadcast a list (rows) of lists (sites) of float64 arrays from the first rank of `group` to the others.

    rows: on the source the tensors (numpy arrays or torch tensors); ignored elsewhere.  Returns the same nested list on
    every rank (numpy arrays for gloo, device tensors for nccl; the source gets its own objects back).  One broadcast of
    the shapes and one flat buffer per row.  rehearse=True issues the collectives even in a group of one rank (bench.py
    --force-dist: the RCCL code path on a single GPU)."""
    rank, world = _group_info(group)
    if world == 1 and not (rehearse and dist.is_available() and dist.is_initialized()):
        return rows
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dev = _comm_device(group)
    meta = [[tuple(int(x) for x in a.shape) for a in row] for row in rows] if rank == 0 else None
    box = [meta]
    dist.broadcast_object_list(box, src=src, group=group)
    meta = box[0]
    out = []
    for i, shapes in enumerate(meta):
        n = sum(int(np.prod(sh)) for sh in shapes)
        if rank == 0:
            flat = torch.cat([torch.as_tensor(a, dtype=torch.float64).reshape(-1).to(dev) for a in rows[i]]) if n else \
                torch.empty(0, dtype=torch.float64, device=dev)
        else:
            flat = torch.empty(n, dtype=torch.float64, device=dev)
        if n:
            dist.broadcast(flat, src=src, group=group)
        if rank == 0:
            out.append(rows[i])
            continue
        row, off = [], 0
        for sh in shapes:
            k = int(np.prod(sh))
            t = flat[off:off + k].reshape(sh)
            row.append(t.clone() if dev.type == 'cuda' else t.numpy().copy())
            off += k
        out.append(row)
    return out


def broadcast_object(obj, group):
    """Small python object from the first rank of `group` to all of its ranks."""
    rank, world = _group_info(group)
    if world == 1:
        return obj
    box = [obj if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0), group=group)
    return box[0]


_HEAD = 6       # energy, degeneracy, log2 P, discarded log2 P, negative probability, rotation


def _pack(s, rot, ncell):
    """One solve as an int64 record: the four float64 results travel as their bit patterns, degeneracy and rotation as
    integers, the state as integers (i64 / i8 fields, exact for any degeneracy).  Slot 0 of an unused or
    non-owner record is _EMPTY."""
    rec = np.zeros(_HEAD + ncell, dtype=np.int64)
    f = np.array([float(s.energy[0]), float(s.probability[0]), float(s.discarded_probability),
                  float(s.negative_probability)], dtype=np.float64).view(np.int64)
    rec[0], rec[2], rec[3], rec[4] = f[0], f[1], f[2], f[3]
    rec[1] = int(s.degeneracy)
    rec[5] = int(rot)
    rec[_HEAD:] = np.asarray(s.states[0], dtype=np.int64)
    return rec


_EMPTY = np.array([np.nan]).view(np.int64)[0]      # bit pattern of NaN in the energy slot marks "no record"


def _unpack_floats(table):
    """(energy, log2 P, discarded, negative) columns of a record table as float64."""
    return [np.ascontiguousarray(table[:, c]).view(np.float64) for c in (0, 2, 3, 4)]


_BEAM_GROUPS = {}


def _beam_groups(world, B):
    """Sub-groups of B consecutive ranks, created once per (world, B) (torch.distributed requires every rank to create
    every group, and groups are never freed, so they are cached for the life of the process group)."""
    key = (id(dist.distributed_c10d._get_default_group()), world, B)     # a re-initialised process group gets new ones
    if key not in _BEAM_GROUPS:
        _BEAM_GROUPS[key] = [dist.new_group(list(range(g * B, (g + 1) * B))) for g in range(world // B)]
    return _BEAM_GROUPS[key]


def solve_rotations(make_solver, rotations=(0, 1, 2, 3), precondition=False, min_dEng=1e-12, group=None,
                    concurrent=False, beam_shards=1, **search_kwargs):
    """Solve the same instance from several lattice rotations, sharded round-robin over the ranks of `group`.

    beam_shards = B > 1 (world size divisible by B; `group` must be the default group): consecutive ranks form
    sub-groups of B that work on the same rotation with the beam split between them (see the module docstring); the
    rotations are then dealt round-robin to the world/B sub-groups.

    concurrent=True runs the rotations assigned to this rank at the same time (threads + streams, GPU only).
    make_solver() -> a fresh solver exposing rotate_graph / precondition / search_ground_state and the result
    attributes of tnac4o.tnac4o.  Returns a dict (identical on every rank):
      energy (min over rotations), degeneracy (max over the rotations reaching that energy, e06:107-109),
      rotation / state / probability of the best record, and the per-rotation table `records`.
    """
    ready = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if ready else 0
    world = dist.get_world_size(group) if ready else 1
    rotations = list(rotations)
    B = int(beam_shards)
    beam_group, owner = None, True
    if B > 1:
        if not ready or world % B:
            raise ValueError('beam_shards=%d needs an initialised process group whose size it divides' % B)
        if group is not None:
            raise ValueError('beam_shards > 1 works on the default process group')
        ngroups = world // B
        beam_group = _beam_groups(world, B)[rank // B]
        if concurrent:
            # the rotations of a rank would issue their collectives on the shared sub-group from several host threads in
            # an order that differs between ranks: refuse instead of hanging
            raise ValueError('concurrent=True cannot be combined with beam_shards > 1 (collectives of a team must be '
                             'issued in one order)')
        owner = (rank % B == 0)
        team, nteams = rank // B, ngroups
    else:
        team, nteams = rank, world
    slots = (len(rotations) + nteams - 1) // nteams
    mine = [r for i, r in enumerate(rotations) if i % nteams == team]
    local, ncell = [], None

    def one(rot):
        s = make_solver()
        if rot:
            s.rotate_graph(rot)
        if precondition:                          # deterministic, so the partners of a beam team repeat it identically
            s.precondition(mode='balancing')
        if beam_group is not None:
            s.search_ground_state(beam_group=beam_group, **search_kwargs)
        else:
            s.search_ground_state(**search_kwargs)
        rec = _pack(s, rot, s.states.shape[1])
        if not owner:
            rec[0] = _EMPTY                       # partners hold the same result; only the owner's record is counted
        return rec
    if concurrent and len(mine) > 1:        # this rank's rotations interleave on the GPU, one stream each
        local = run_concurrent([(lambda r=rot: one(r)) for rot in mine])
    else:
        local = [one(rot) for rot in mine]
    if local:
        ncell = len(local[0]) - _HEAD
    if ncell is None:                       # a rank without work still takes part in the gather
        probe = make_solver()
        ncell = probe.Nx * probe.Ny
    buf = np.zeros((slots, _HEAD + ncell), dtype=np.int64)
    buf[:, 0] = _EMPTY
    for i, rec in enumerate(local):
        buf[i] = rec
    if ready and world > 1:
        backend = dist.get_backend(group)
        dev = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
        mine_t = torch.as_tensor(buf, dtype=torch.int64).to(dev)
        out = [torch.empty_like(mine_t) for _ in range(world)]
        dist.all_gather(out, mine_t, group=group)          # the single exchange of the whole solve
        table = torch.stack(out).cpu().numpy().reshape(world * slots, -1)
    else:
        table = buf
    table = table[table[:, 0] != _EMPTY]
    table = table[np.argsort(table[:, 5], kind='stable')]
    E, logP, disc, neg = _unpack_floats(table)
    best = np.flatnonzero(E - E.min() <= min_dEng)
    top = best[np.argmax(logP[best])]                       # most probable among the minimisers
    return {
        'energy': float(E.min()),
        'degeneracy': int(table[best, 1].max()),
        'rotation': int(table[top, 5]),
        'probability': float(logP[top]),
        'state': table[top, _HEAD:].astype(np.int64),
        'records': [dict(rotation=int(r[5]), energy=float(E[i]), degeneracy=int(r[1]), probability=float(logP[i]),
                         discarded_probability=float(disc[i]), negative_probability=float(neg[i]))
                    for i, r in enumerate(table)],
    }
