"""Host side of libtnpeps under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU box: the library
is rebuilt with -fsanitize=address,undefined for the HOST code only (-fno-gpu-sanitize; GPU ASan is not available on this
pool) and every entry point is driven down its error paths in a child process with the sanitizer runtime preloaded:
argument errors must come back as rc < 0 with a message and no launch; with valid-looking arguments but no GPU the HIP
failure must come back as rc > 0 (never an abort, an exception or a sanitizer report)."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN_LIB = os.path.join(ROOT, 'tnac4o_amd', 'libtnpeps_asan.so')

CHILD = r'''
import ctypes as C, sys
sys.path.insert(0, %(root)r)
from tnac4o_amd import _lib
L = C.CDLL(%(lib)r)
for name, (res, args) in _lib.SIGNATURES.items():
    fn = getattr(L, name); fn.restype, fn.argtypes = res, args
buf = C.create_string_buffer(512)
def msg():
    L.tn_last_error(buf, 512); return buf.value.decode()
def expect_neg(rc, what):
    assert rc < 0, (what, rc); m = msg(); assert m, what; print('ok  %%-22s rc=%%d  %%s' %% (what, rc, m))
host = (C.c_double * 4096)()
P = C.cast(host, C.c_void_p)            # a valid HOST address standing in for device memory: nothing may dereference it
i64 = C.c_int64(0); f64 = C.c_double(0); i32 = C.c_int(0)
assert L.tn_version() == _lib.ABI_VERSION
expect_neg(L.tn_gemm(-1, 4, 4, 1.0, P, 4, 1, P, 4, 1, 0.0, P, 4, 1, 1, 0, 0, 0, None, 0, None), 'tn_gemm negative dim')
expect_neg(L.tn_gemm(4, 4, 4, 1.0, None, 4, 1, P, 4, 1, 0.0, P, 4, 1, 1, 0, 0, 0, None, 0, None), 'tn_gemm null')
expect_neg(L.tn_absorb(None, P, P, 1, 1, 1, 1, 1, 1, 1, 0, 1, 0, 0, 0, None), 'tn_absorb null')
expect_neg(L.tn_absorb(P, P, P, 1, 1, 1, 1, 1, 1, 1, 0, -2, 0, 0, 0, None), 'tn_absorb batch')
expect_neg(L.tn_qr_batched(P, 4, 1, 8, 4, P, 4, 1, P, 4, 1, 32, 0.0, None, 3, 32, 32, 16, P, 16, None, None, 0), 'tn_qr_batched ws')
expect_neg(L.tn_qr_batched(P, 4, 1, 8, 4, P, 4, 1, P, 4, 1, 32, 0.0, None, 3, 32, 32, 16, P, 1 << 24, None, None, 9), 'tn_qr_batched sides')
expect_neg(L.tn_svd_trunc_batched(P, 4, 1, 4, 4, 4, 0.0, P, 4, 1, P, P, 4, 1, None, None, None, None, 2, 16, 16, 4, 16, P, 1 << 20, None), 'tn_svd_trunc_batched keep')
expect_neg(L.tn_qr(P, 4, 1, 0, 4, P, 4, 1, P, 4, 1, 32, 0.0, None, P, 1 << 20, None, None), 'tn_qr empty')
expect_neg(L.tn_qr(P, 4, 1, 8, 4, P, 4, 1, P, 4, 1, 48, 0.0, None, P, 1 << 20, None, None), 'tn_qr bad nb')
expect_neg(L.tn_qr(P, 4, 1, 8, 4, P, 4, 1, P, 4, 1, 32, 0.0, None, P, 16, None, None), 'tn_qr small ws')
expect_neg(L.tn_qr(P, 4, 1, 8, 4, P, 4, 1, P, 4, 1, 32, 2.0, None, P, 1 << 20, None, None), 'tn_qr rank_tol')
expect_neg(L.tn_panel_orth(None, 4, 1, 64, 4, P, 4, 1, 0, None, None, P, 1 << 20, None), 'tn_panel_orth null')
expect_neg(L.tn_panel_orth(P, 4, 1, 64, 4, P, 4, 1, 2, None, None, P, 1 << 20, None), 'tn_panel_orth method')
expect_neg(L.tn_panel_orth(P, 4, 1, 64, 4, P, 4, 1, 0, None, None, P, 16, None), 'tn_panel_orth ws')
expect_neg(L.tn_panel_orth(P, 40, 1, 64, 40, C.cast(C.byref(host, 8), C.c_void_p), 40, 1, 1, None, None, P, 1 << 22, None), 'tn_panel_orth width')
expect_neg(L.tn_panel_stats(None, 0), 'tn_panel_stats null')
expect_neg(L.tn_panel_stats_stream(None, 0, None), 'tn_panel_stats_stream null')
expect_neg(L.tn_smallqr_stats(None, 0, None), 'tn_smallqr_stats null')
expect_neg(L.tn_fused_timeouts(None, None), 'tn_fused_timeouts null')
desc = (C.c_int64 * 10)(C.cast(host, C.c_void_p).value, 64, 1, 8, 65, C.cast(host, C.c_void_p).value, 64, 1, 9, 8)
expect_neg(L.tn_svdvals_small_batched(P, -1, desc, P, None), 'svdvals_small_batched batch')
expect_neg(L.tn_svdvals_small_batched(P, 1, desc, P, None), 'svdvals_small_batched length')
expect_neg(L.tn_svdvals_small_batched(P, 1, C.cast(C.byref(desc, 40), C.POINTER(C.c_int64)), P, None), 'svdvals_small_batched order')
assert L.tn_svdvals_small_batched(None, 0, None, None, None) == 0
expect_neg(L.tn_svd_trunc(P, 4, 1, 4, 4, 0, 0.0, P, 4, 1, P, P, 4, 1, C.byref(i64), None, None, None, P, 1 << 20, None), 'tn_svd_trunc Dmax')
expect_neg(L.tn_svd_trunc(P, 4, 1, 4, 4, 4, 0.0, P, 4, 1, P, P, 4, 1, None, None, None, None, P, 1 << 20, None), 'tn_svd_trunc keep')
expect_neg(L.tn_svd_trunc(P, 4, 1, 4, 4, 4, 0.0, P, 4, 1, P, P, 4, 1, C.byref(i64), None, None, None, P, 8, None), 'tn_svd_trunc ws')
expect_neg(L.tn_svdvals(P, 4, 1, 0, 4, host, None, None, P, 1 << 20, None), 'tn_svdvals dims')
expect_neg(L.tn_nfactor(P, 0, P, P, None), 'tn_nfactor empty')
expect_neg(L.tn_nfactor(None, 4, P, P, None), 'tn_nfactor null')
expect_neg(L.tn_normalize_pow2(P, 4, P, None, 8192, None), 'tn_normalize null')
expect_neg(L.tn_calc_pn(P, P, P, P, P, P, P, P, P, 4, 0, 1, 1, 1, 1, 1, P, P, None, None, None), 'tn_calc_pn q')
expect_neg(L.tn_calc_pn(P, P, P, P, P, P, P, P, P, 4, 256, 1, 1, 64, 512, 64, P, P, None, None, None), 'tn_calc_pn lds')
expect_neg(L.tn_calc_pn(P, P, P, P, P, P, P, P, P, 4, 16, 1, 1, 1, 1, 1, P, P, P, None, None), 'tn_calc_pn log2p pair')
expect_neg(L.tn_merge_groups(P, P, P, P, P, -1, 0.0, P, P, P, None), 'tn_merge_groups ngroups')
expect_neg(L.tn_compress_mps_arena_bytes(0, None, None, 8), 'tn_compress_mps_arena_bytes L')
class Cell(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('F', 'dmap', 'rmap', 'down', 'right', 'Es', 'E1', 'E4', 'left_map', 'up_map', 'A')] + \
               [(n, C.c_int64) for n in ('q', 'nl', 'nu', 'pd', 'br', 'e1cols', 'e4cols', 'Dl', 'p', 'Dr')]
cell = Cell(*([P.value] * 11 + [4, 2, 2, 2, 2, 2, 2, 1, 2, 1]))
cells = C.cast(C.pointer(cell), C.c_void_p)
bs = lambda cl, M, ws, wsb, nbp=C.byref(i64): L.tn_beam_search(1, 1, cl, M, 1, -20.0, 1e-12, 4, P, P, P, P, nbp, C.byref(f64), C.byref(f64), ws, wsb, None)
assert L.tn_beam_search_ws_bytes(4, 4, 64, 16, 64, 64, 256) > 0
expect_neg(bs(None, 8, P, 1 << 30), 'tn_beam_search cells')
expect_neg(bs(cells, 0, P, 1 << 30), 'tn_beam_search M')
expect_neg(bs(cells, 8, P, 1 << 30, None), 'tn_beam_search results')
expect_neg(bs(cells, 8, P, 1024), 'tn_beam_search ws')
cell.p = 3
expect_neg(bs(cells, 8, P, 1 << 30), 'tn_beam_search bond')
cell.p, cell.Dl, cell.nl = 2, 512, 16
expect_neg(bs(cells, 8, P, 1 << 30), 'tn_beam_search env')
expect_neg(L.tn_env_rr_batched(P, P, P, P, P, 4, 512, 16, 64, 16, 16, 16, P, None), 'tn_env_rr acc')
expect_neg(L.tn_env_rr_batched(P, P, P, P, P, -1, 4, 4, 4, 4, 4, 4, P, None), 'tn_env_rr nk')
expect_neg(L.tn_env_rl_batched(None, P, P, 4, 4, 4, P, None), 'tn_env_rl null')
expect_neg(L.tn_balance(P, 4, 1, 65, 0.0, P, None, None), 'tn_balance n')
expect_neg(L.tn_peps_factor(P, P, P, P, P, P, P, P, None, 4, 4, 4, P, None), 'tn_peps_factor null')
expect_neg(L.tn_profile_get(99, C.byref(C.c_uint64(0)), C.byref(f64), C.byref(f64), C.byref(f64)), 'tn_profile_get family')
# zero-sized work is a no-op
assert L.tn_env_rl_batched(None, None, None, 0, 4, 4, None, None) == 0
assert L.tn_nfactor_batched(P, 0, 4, None) == 0
# valid-looking arguments, no GPU in this container: the HIP failure is reported as rc > 0 with its text
rc = L.tn_scale_by(P, 16, P, None)
assert rc > 0 and msg(), rc
print('ok  no-device launch      rc=%%d  %%s' %% (rc, msg()))
assert L.tn_last_error(None, 0) >= 0
print('ASAN_CHILD_OK')
'''


def _runtime():
    libs = sorted(glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so'))
    return libs[-1] if libs else None


def test_c_abi_error_paths_under_asan_ubsan():
    from tnac4o_amd import _lib
    rt = _runtime()
    if rt is None or not os.path.exists(os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')):
        pytest.skip('ROCm clang sanitizer runtime not found')
    import torch
    if torch.cuda.is_available():
        pytest.skip('CPU-box test: with a GPU present the no-device branch would launch on garbage pointers')
    cmd = [os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-O1', '-g', '-std=c++17', '-fPIC', '-shared',
           '-fsanitize=address,undefined', '-fno-gpu-sanitize', '-fno-sanitize-recover=undefined', '-shared-libsan',
           '-DTN_SRC_HASH="%s"' % _lib.source_hash(), '-o', ASAN_LIB] + [os.path.join(_lib.CSRC, s) for s in _lib.SOURCES]
    subprocess.run(cmd, check=True)
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1', UBSAN_OPTIONS='print_stacktrace=1')
    out = subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT, lib=ASAN_LIB)], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and 'ASAN_CHILD_OK' in out.stdout, (out.stdout[-3000:], out.stderr[-3000:])
    assert 'runtime error' not in out.stderr and 'AddressSanitizer' not in out.stderr, out.stderr[-3000:]
