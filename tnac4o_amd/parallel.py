"""Multi-GPU driver: the lattice rotations are independent solves (reference examples/e06:97-109 runs them in a
sequential loop), so they shard across ranks with no data-path collective; one all-gather of a < 3 KB record per
rotation merges the results (energy = min, degeneracy = max over the minimisers, as e06:107-109 does).

One process per GPU (torchrun); backend 'nccl' is RCCL over xGMI on ROCm, 'gloo' is used by the CPU tests.  The
solver is injected (`make_solver`) so that the sharding / gather / merge logic can be exercised without a GPU.
"""
import numpy as np
import torch
import torch.distributed as dist

def run_concurrent(fns):
    """Run independent chains concurrently on one GPU: each callable gets its own host thread and its own HIP stream
    (one stream per rotation so that the latency-bound chains interleave on the device).  Returns the
    list of results; the first exception raised by a chain is re-raised here."""
    import threading
    n = len(fns)
    if n == 1:
        return [fns[0]()]
    streams = [torch.cuda.Stream() for _ in range(n)]
    out, err = [None] * n, [None] * n
    cur = torch.cuda.current_stream()

    def work(i):
        try:
            with torch.cuda.stream(streams[i]):
                streams[i].wait_stream(cur)
                out[i] = fns[i]()
                streams[i].synchronize()
        except BaseException as e:      # noqa: BLE001 - re-raised in the caller's thread
            err[i] = e
    th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in err:
        if e is not None:
            raise e
    return out


_HEAD = 6       # energy, degeneracy, log2 P, discarded log2 P, negative probability, rotation


def _pack(s, rot, ncell):
    rec = np.full(_HEAD + ncell, np.nan)
    rec[0] = float(s.energy[0])
    rec[1] = float(s.degeneracy)
    rec[2] = float(s.probability[0])
    rec[3] = float(s.discarded_probability)
    rec[4] = float(s.negative_probability)
    rec[5] = float(rot)
    rec[_HEAD:] = np.asarray(s.states[0], dtype=np.float64)
    return rec


def solve_rotations(make_solver, rotations=(0, 1, 2, 3), precondition=False, min_dEng=1e-12, group=None,
                    concurrent=False, **search_kwargs):
    """Solve the same instance from several lattice rotations, sharded round-robin over the ranks of `group`.

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
    slots = (len(rotations) + world - 1) // world
    mine = [r for i, r in enumerate(rotations) if i % world == rank]
    local, ncell = [], None

    def one(rot):
        s = make_solver()
        if rot:
            s.rotate_graph(rot)
        if precondition:
            s.precondition(mode='balancing')
        s.search_ground_state(**search_kwargs)
        return _pack(s, rot, s.states.shape[1])
    if concurrent and len(mine) > 1:        # this rank's rotations interleave on the GPU, one stream each
        local = run_concurrent([(lambda r=rot: one(r)) for rot in mine])
    else:
        local = [one(rot) for rot in mine]
    if local:
        ncell = len(local[0]) - _HEAD
    if ncell is None:                       # a rank without work still takes part in the gather
        probe = make_solver()
        ncell = probe.Nx * probe.Ny
    buf = np.full((slots, _HEAD + ncell), np.nan)
    for i, rec in enumerate(local):
        buf[i] = rec
    if ready and world > 1:
        backend = dist.get_backend(group)
        dev = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
        mine_t = torch.as_tensor(buf, dtype=torch.float64).to(dev)
        out = [torch.empty_like(mine_t) for _ in range(world)]
        dist.all_gather(out, mine_t, group=group)          # the single exchange of the whole solve
        table = torch.stack(out).cpu().numpy().reshape(world * slots, -1)
    else:
        table = buf
    table = table[~np.isnan(table[:, 0])]
    table = table[np.argsort(table[:, 5], kind='stable')]
    E = table[:, 0]
    best = np.flatnonzero(E - E.min() <= min_dEng)
    top = best[np.argmax(table[best, 2])]                   # most probable among the minimisers
    return {
        'energy': float(E.min()),
        'degeneracy': int(table[best, 1].max()),
        'rotation': int(table[top, 5]),
        'probability': float(table[top, 2]),
        'state': table[top, _HEAD:].astype(np.int64),
        'records': [dict(rotation=int(r[5]), energy=float(r[0]), degeneracy=int(r[1]), probability=float(r[2]),
                         discarded_probability=float(r[3]), negative_probability=float(r[4])) for r in table],
    }
