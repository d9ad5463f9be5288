"""The rotation-sharded multi-rank path (tnac4o_amd.parallel) on CPU: world_size 2 over gloo, with the CPU oracle
injected as the per-rank solver (the product solver itself only runs on a GPU).  Checks the sharding, the single
all-gather and the merge rule against a serial run and against the reference's golden ground state."""
import os
import subprocess
import sys
import json

import numpy as np
import pytest

import golden_inputs as gi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import numpy as np, torch, torch.distributed as dist
import golden_inputs as gi
from oracle import solver_ref as sr
from tnac4o_amd.parallel import solve_rotations
dist.init_process_group('gloo')
J = gi.droplet_J(128, 1)
make = lambda: sr.RefSolver(mode='Ising', Nx=4, Ny=4, Nc=8, J=J, beta=3.0)
res = solve_rotations(make, rotations=%(rots)r, M=256, relative_P_cutoff=1e-6, Dmax=8)
res['state'] = [int(x) for x in res['state']]
res['rank'] = dist.get_rank()
print('RESULT ' + json.dumps(res), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def run_world(nproc, rots, port):
    code = WORKER % dict(root=ROOT, rots=tuple(rots))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), OMP_NUM_THREADS='2', OPENBLAS_NUM_THREADS='2')
    procs = []
    for r in range(nproc):
        e = dict(env, RANK=str(r), WORLD_SIZE=str(nproc), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, '-c', code], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-2000:]
        line = [l for l in out.splitlines() if l.startswith('RESULT ')][-1]
        outs.append(json.loads(line[7:]))
    return outs


@pytest.mark.parametrize('rots', [(0, 1, 2, 3), (0, 1, 2)])
def test_rotations_sharded_over_two_ranks(rots):
    outs = run_world(2, rots, 29530 + len(rots))
    a, b = outs
    for k in ('energy', 'degeneracy', 'rotation', 'probability', 'state', 'records'):
        assert a[k] == b[k]                                   # every rank holds the merged result
    assert [r['rotation'] for r in a['records']] == list(rots)
    E, bits = gi.golden_groundstate(128, 1)
    assert a['energy'] == pytest.approx(E, abs=1e-5)
    assert all(r['energy'] == pytest.approx(E, abs=1e-5) for r in a['records'])
    # serial run (no process group) gives the same merged record
    from oracle import solver_ref as sr
    from tnac4o_amd.parallel import solve_rotations
    J = gi.droplet_J(128, 1)
    ser = solve_rotations(lambda: sr.RefSolver(mode='Ising', Nx=4, Ny=4, Nc=8, J=J, beta=3.0), rotations=rots, M=256,
                          relative_P_cutoff=1e-6, Dmax=8)
    assert ser['energy'] == a['energy'] and ser['degeneracy'] == a['degeneracy'] and ser['rotation'] == a['rotation']
    assert [int(x) for x in ser['state']] == a['state']
    assert ser['probability'] == pytest.approx(a['probability'], abs=1e-12)


BEAM_WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import numpy as np, torch, torch.distributed as dist
import golden_inputs as gi
from sharded_ref import ShardedRef
from tnac4o_amd.parallel import solve_rotations
dist.init_process_group('gloo')
J = gi.droplet_J(128, %(inst)d)
make = lambda: ShardedRef(mode='Ising', Nx=4, Ny=4, Nc=8, J=J, beta=3.0)
res = solve_rotations(make, rotations=%(rots)r, beam_shards=%(shards)d, M=64, relative_P_cutoff=1e-6, Dmax=8)
res['state'] = [int(x) for x in res['state']]
res['rank'] = dist.get_rank()
print('RESULT ' + json.dumps(res), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def run_beam_world(nproc, rots, shards, inst, port):
    code = BEAM_WORKER % dict(root=ROOT, rots=tuple(rots), shards=shards, inst=inst)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), OMP_NUM_THREADS='2', OPENBLAS_NUM_THREADS='2')
    procs = []
    for r in range(nproc):
        e = dict(env, RANK=str(r), WORLD_SIZE=str(nproc), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, '-c', code], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=900)
        assert p.returncode == 0, err[-2000:]
        line = [l for l in out.splitlines() if l.startswith('RESULT ')][-1]
        outs.append(json.loads(line[7:]))
    return outs


@pytest.mark.parametrize('nproc,rots,inst', [(2, (0,), 1), (4, (0, 3), 2), (8, (0, 1, 2, 3), 1)])
def test_beam_sharded_inside_a_rotation(nproc, rots, inst):
    """teams of 2 ranks per rotation; the team's first rank sweeps and broadcasts rhoT, every site-step's
    branches are split between the two, one all-gather per step; every rank must end with the serial result.  The world-8 case is
    the exact layout `bench.py --gpus 8` runs on a node: 4 rotation teams x 2 beam partners, all 4 rotations of one instance."""
    outs = run_beam_world(nproc, rots, 2, inst, 29560 + nproc)
    for o in outs[1:]:
        for k in ('energy', 'degeneracy', 'rotation', 'probability', 'state', 'records'):
            assert o[k] == outs[0][k]
    assert [r['rotation'] for r in outs[0]['records']] == list(rots)       # one record per rotation (the owner's)
    from oracle import solver_ref as sr
    from tnac4o_amd.parallel import solve_rotations
    J = gi.droplet_J(128, inst)
    ser = solve_rotations(lambda: sr.RefSolver(mode='Ising', Nx=4, Ny=4, Nc=8, J=J, beta=3.0), rotations=rots, M=64,
                          relative_P_cutoff=1e-6, Dmax=8)
    a = outs[0]
    assert ser['energy'] == a['energy'] and ser['degeneracy'] == a['degeneracy'] and ser['rotation'] == a['rotation']
    assert [int(x) for x in ser['state']] == a['state']
    assert ser['probability'] == a['probability']               # same arithmetic on the same tables: bit-identical
    for r1, r2 in zip(ser['records'], a['records']):
        assert r1 == r2


def test_more_ranks_than_rotations_idle_partners():
    """The layout `bench.py --gpus 8` runs by default (--beam-shards auto): one working rank per lattice rotation, the other four
    ranks idle -- they only take part in the final gather; every rank ends with the serial result."""
    rots = (0, 1, 2, 3)
    outs = run_beam_world(8, rots, 1, 1, 29590)
    for o in outs[1:]:
        for k in ('energy', 'degeneracy', 'rotation', 'probability', 'state', 'records'):
            assert o[k] == outs[0][k]
    assert [r['rotation'] for r in outs[0]['records']] == list(rots)
    from oracle import solver_ref as sr
    from tnac4o_amd.parallel import solve_rotations
    J = gi.droplet_J(128, 1)
    ser = solve_rotations(lambda: sr.RefSolver(mode='Ising', Nx=4, Ny=4, Nc=8, J=J, beta=3.0), rotations=rots, M=64,
                          relative_P_cutoff=1e-6, Dmax=8)
    a = outs[0]
    assert ser['energy'] == a['energy'] and ser['degeneracy'] == a['degeneracy'] and ser['rotation'] == a['rotation']
    assert [int(x) for x in ser['state']] == a['state'] and ser['probability'] == a['probability']


EXCHANGE_WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from tnac4o_amd import parallel
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
grp = dist.new_group(list(range(world)))       # (the helpers take an explicit group: None means 'no sharding')
ok = True
for trial, (nb, q) in enumerate([(37, 16), (5, 8), (1, 4), (64, 256), (2, 3)]):
    rng = np.random.default_rng(100 + trial)                      # the same table on every rank
    tab = rng.standard_normal((nb, q)) * 8.0
    tab[rng.random((nb, q)) < 0.1] = -np.inf                      # zero-probability entries
    lo, hi = parallel.shard_range(nb, rank, world)
    flat = torch.as_tensor(tab[lo:hi].reshape(-1).copy())
    mn = flat.min().reshape(1) if flat.numel() else torch.full((1,), float('inf'), dtype=torch.float64)
    mx = flat.max().reshape(1) if flat.numel() else torch.full((1,), float('-inf'), dtype=torch.float64)
    gmn, gmx = parallel.allreduce_minmax(mn, mx, grp)
    ok &= float(gmn) == tab.min() and float(gmx) == tab.max()
    cutoff = gmx + float(np.log2(1e-3))
    mask = flat > cutoff
    idx = mask.nonzero().squeeze(1) + lo * q
    vals = flat[mask]
    rest = torch.where(mask, torch.full_like(flat, float('-inf')), flat)
    rest_max = rest.max().reshape(1) if flat.numel() else torch.full((1,), float('-inf'), dtype=torch.float64)
    gi_, gv, grm = parallel.allgather_candidates(idx, vals, rest_max, grp)
    full = tab.reshape(-1)
    want = np.flatnonzero(full > float(cutoff))
    ok &= np.array_equal(gi_.numpy(), want) and np.array_equal(gv.numpy(), full[want])
    cut = full[full <= float(cutoff)]
    ok &= float(grm) == (cut.max() if cut.size else -np.inf)
print('RESULT ' + json.dumps({'rank': rank, 'ok': bool(ok)}), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize('nproc', [2, 3])
def test_pruned_candidate_exchange(nproc):
    """The pruned exchange of a site-step (tnac4o_amd.parallel.allreduce_minmax / allgather_candidates): every rank
    cuts the log-probabilities of ITS slice of the branches against the global maximum and only the survivors travel; the
    concatenation in rank order must be exactly the candidates a single process keeps, in ascending flat index (the canonical order
    of tnac4o_amd/beam.py), with the same largest cut value -- including empty slices and -inf entries."""
    code = EXCHANGE_WORKER % dict(root=ROOT)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29590 + nproc), OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1')
    procs = []
    for r in range(nproc):
        e = dict(env, RANK=str(r), WORLD_SIZE=str(nproc), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, '-c', code], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err[-2000:]
        line = [l for l in out.splitlines() if l.startswith('RESULT ')][-1]
        assert json.loads(line[7:])['ok']


def test_shard_range_partitions():
    from tnac4o_amd.parallel import shard_range
    for n in (0, 1, 5, 64, 1000):
        for w in (1, 2, 3, 8):
            cuts = [shard_range(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            sizes = [h - l for l, h in cuts]
            assert max(sizes) - min(sizes) <= 1


TEAM_EXCHANGE_WORKER = r'''
import ctypes as C, json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from tnac4o_amd import beam, parallel
dist.init_process_group('gloo')
world, rank = dist.get_world_size(), dist.get_rank()
B = %(B)d
group = parallel._beam_groups(world, B)[rank // B]
team, trank = dist.get_world_size(group), dist.get_rank(group)
ok = True
ws = torch.zeros(1 << 16, dtype=torch.uint8)
fn, err = beam._team_exchange(ws, group)
for step, (nb, q) in enumerate([(1, 4), (2, 3), (5, 7), (64, 16), (3, 1)]):       # (fewer branches than ranks: empty slices)
    lp = ws[256:256 + nb * q * 8].view(torch.float64)
    mp = ws[32768:32768 + nb * 8].view(torch.float64)
    lp.fill_(float('nan')); mp.fill_(float('nan'))
    lo, hi = nb * trank // team, nb * (trank + 1) // team
    want_lp = torch.arange(nb * q, dtype=torch.float64) * 0.5 + 100.0 * step + 7.0 * (rank // B)
    want_mp = -torch.arange(nb, dtype=torch.float64) - step - 3.0 * (rank // B)
    lp[lo * q:hi * q] = want_lp[lo * q:hi * q]
    mp[lo:hi] = want_mp[lo:hi]
    # what tn_beam_search_team does at every site-step: call the hook with the DEVICE pointers of the two arrays
    rc = C.cast(fn, beam._EXCHANGE_FN)(None, lp.data_ptr(), mp.data_ptr(), nb, q, trank, team)
    ok = ok and rc == 0 and not err and torch.equal(lp, want_lp) and torch.equal(mp, want_mp)
print('RESULT ' + json.dumps({'ok': bool(ok), 'team': team, 'rank': rank}), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize('nproc,B', [(2, 2), (3, 3), (8, 2)])
def test_team_exchange_hook_of_the_library_beam_walk(nproc, B):
    """The exchange function tn_beam_search_team calls once per site-step (tnac4o_amd/beam.py: _team_exchange; include/tnpeps.h:
    tn_beam_exchange_fn) over gloo: every rank of a team fills its contiguous slice of the log2 p table and of the minima, the hook
    must leave the complete arrays on every rank -- teams of 2 and 3, the 4 x 2 layout of `bench.py --gpus 8 --beam-shards 2` (four
    teams exchanging independently), table sizes with empty slices.  (The walk itself needs the GPU library: the two-process GPU test
    tests/test_gpu_mps.py::test_beam_sharded_product_path_two_ranks runs it through this hook and compares it bit for bit.)"""
    code = TEAM_EXCHANGE_WORKER % dict(root=ROOT, B=B)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29560 + nproc + B), OMP_NUM_THREADS='1')
    procs = [subprocess.Popen([sys.executable, '-c', code], env=dict(env, RANK=str(r), WORLD_SIZE=str(nproc), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(nproc)]
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err[-2000:]
        res = json.loads([l for l in out.splitlines() if l.startswith('RESULT ')][-1][7:])
        assert res['ok'] and res['team'] == B, res
