"""Test adapter: the CPU oracle wrapped in the product's beam-sharding helpers (tnac4o_amd.parallel), so that the
multi-rank protocol of SURVEY.md 8e-ii (sweep on the team's first rank + broadcast, branch slices + all-gather,
identical merge on every rank) runs under gloo without a GPU.  The product solver plugs the same helpers around its
HIP kernels (tnac4o_amd/tnac4o.py: search_ground_state(beam_group=...))."""
import numpy as np
import torch.distributed as dist

from oracle import mps_ref as mr
from oracle import solver_ref as sr
from tnac4o_amd import parallel


class ShardedRef(sr.RefSolver):
    def search_ground_state(self, beam_group=None, **kw):
        if beam_group is None:
            return super().search_ground_state(**kw)

        def sweep_hook(solver, run_sweep):
            owner = dist.get_rank(beam_group) == 0
            if owner:
                run_sweep()
            rows = parallel.broadcast_site_tensors([m.A for m in solver.rhoT] if owner else None, beam_group)
            if not owner:
                solver.rhoT = []
                for A in rows:
                    m = mr.RefMPS(d=1, L=solver.Nx, Dmax=1)
                    m.A = [np.asarray(a) for a in A]
                    solver.rhoT.append(m)
        return super().search_ground_state(sweep_hook=sweep_hook,
                                           pn_gather=lambda f, nb, q: parallel.gather_branch_tables(f, nb, q, beam_group), **kw)
