"""Timing diagnostics of eig_small (third form): TN_EIG_DBG bit 0 = no J update, bit 1 = no G update, bit 2 = no look-ahead arithmetic
(rotations forced "on" so that every step runs).  Results are meaningless with a non-zero value; only the time per launch matters."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tnac4o_amd import ops
from tnac4o_amd._lib import lib
from svd_probe import case

R, _ = case(512, 500, 6, 5)
L = lib()
try:
    ops._svd_trunc_raw(R, 256, 1e-16)
except Exception as e:            # noqa: BLE001
    pass
torch.cuda.synchronize()
L.tn_profile_reset()
L.tn_profile_enable((1 << 15) - 1)
try:
    ops._svd_trunc_raw(R, 256, 1e-16)
except Exception as e:            # noqa: BLE001
    pass
torch.cuda.synchronize()
L.tn_profile_enable(0)
calls, ms, fl, by = C.c_uint64(0), C.c_double(0), C.c_double(0), C.c_double(0)
L.tn_profile_get(7, C.byref(calls), C.byref(ms), C.byref(fl), C.byref(by))
print('TN_EIG_DBG=%s  eig_small %d launches  %.1f us each' % (os.environ.get('TN_EIG_DBG', '0'), calls.value, 1e3 * ms.value / max(1, calls.value)), flush=True)
