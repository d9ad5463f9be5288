"""tnac4o_amd — MI355X-native boundary-MPS PEPS contraction engine behind the tnac4o call surface.

    import tnac4o_amd as tnac4o
    ins = tnac4o.tnac4o(mode='Ising', Nx=16, Ny=16, Nc=8, J=J, beta=3)
    ins.search_ground_state(M=1024, relative_P_cutoff=1e-8, Dmax=64)

Re-exports mirror the reference's tnac4o/__init__.py:1-2.  All compute runs in libtnpeps.so (hand-written HIP for
gfx950); importing works without a GPU, running anything does not.
"""
import os as _os

# One hardware queue per chain: the 4 lattice rotations of an instance run as 4 chains on 4 HIP streams of one GPU, and ROCm
# multiplexes all streams of a process (the default stream included) onto GPU_MAX_HW_QUEUES hardware queues, 4 unless told
# otherwise -- two chains on one queue serialise (4.5 s per 4-rotation sweep step instead of 2.9 s, parallel.run_concurrent).
# The runtime reads the variable when it initialises, i.e. at the first HIP call of the process; a value set by the user wins.
# The library itself does not rely on this side effect: csrc/cholqr.hip reads GPU_MAX_HW_QUEUES (4 when unset) and the device's CU
# count when it sizes the launches that spin on in-kernel barriers.  What an import after the runtime's start cannot do is raise the
# queue count: HIP_STARTED_BEFORE_IMPORT records that case and parallel.run_concurrent warns about the slow regime.
HIP_STARTED_BEFORE_IMPORT = False
try:
    import sys as _sys
    _t = _sys.modules.get('torch')
    if _t is not None and _t.cuda.is_initialized() and 'GPU_MAX_HW_QUEUES' not in _os.environ:
        HIP_STARTED_BEFORE_IMPORT = True
except Exception:                      # noqa: BLE001
    pass
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

from .auxx import load_Jij, round_Jij, minus_Jij, Jij_f2p, energy_Jij, energy_RMF  # noqa: F401
from . import mps  # noqa: F401
from .tnac4o import tnac4o, load  # noqa: F401
