"""ctypes binding of libtnpeps.so (include/tnpeps.h).  The HIP library is the only compute backend:
if it is missing or fails to load this module raises — there is no CPU fallback."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libtnpeps.so')
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['api.hip', 'gemm_f64.hip', 'small.hip', 'qr.hip', 'svd.hip', 'absorb.hip', 'misc.hip', 'beam.hip', 'prof.hip', 'tsqr.hip', 'peps.hip']

_i64, _f64, _int, _ptr = C.c_int64, C.c_double, C.c_int, C.c_void_p

# name -> (restype, argtypes): every symbol include/tnpeps.h declares
SIGNATURES = {
    'tn_version': (_int, []),
    'tn_last_error': (_int, [C.c_char_p, _int]),
    'tn_gemm': (_int, [_i64, _i64, _i64, _f64, _ptr, _i64, _i64, _ptr, _i64, _i64, _f64, _ptr, _i64, _i64,
                       _i64, _i64, _i64, _i64, _ptr, _i64, _ptr]),
    'tn_gemm_ws_bytes': (_i64, [_i64, _i64, _i64, _i64]),
    'tn_absorb': (_int, [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _int, _ptr]),
    'tn_qr': (_int, [_ptr, _i64, _i64, _i64, _i64, _ptr, _i64, _i64, _ptr, _i64, _i64, _int, _f64, C.POINTER(_i64), _ptr, _i64,
              _ptr]),
    'tn_qr_ws_bytes': (_i64, [_i64, _i64, _int]),
    'tn_svd_trunc': (_int, [_ptr, _i64, _i64, _i64, _i64, _i64, _f64, _ptr, _i64, _i64, _ptr, _ptr, _i64, _i64,
                            C.POINTER(_i64), C.POINTER(_f64), C.POINTER(_int), C.POINTER(_int), _ptr, _i64, _ptr]),
    'tn_svdvals': (_int, [_ptr, _i64, _i64, _i64, _i64, C.POINTER(_f64), C.POINTER(_int), C.POINTER(_int), _ptr, _i64,
                          _ptr]),
    'tn_svd_ws_bytes': (_i64, [_i64, _i64, _int]),
    'tn_nfactor': (_int, [_ptr, _i64, _ptr, _ptr, _ptr]),
    'tn_scale_by': (_int, [_ptr, _i64, _ptr, _ptr]),
    'tn_normalize_pow2': (_int, [_ptr, _i64, _ptr, _ptr, _i64, _ptr]),
    'tn_scale_phys': (_int, [_ptr, _i64, _i64, _i64, _ptr, _int, _ptr]),
    'tn_calc_pn': (_int, [_ptr] * 9 + [_i64] * 7 + [_ptr, _ptr, _ptr]),
    'tn_nfactor_batched': (_int, [_ptr, _i64, _i64, _ptr]),
    'tn_peps_factor': (_int, [_ptr] * 9 + [_i64] * 3 + [_ptr, _ptr]),
    'tn_mpo_from_factor': (_int, [_ptr] * 3 + [_i64] * 5 + [_ptr, _ptr]),
    'tn_profile_enable': (None, [C.c_uint]),
    'tn_profile_reset': (None, []),
    'tn_profile_sample': (None, [C.c_uint]),
    'tn_profile_get': (_int, [_int, C.POINTER(C.c_uint64), C.POINTER(_f64), C.POINTER(_f64), C.POINTER(_f64)]),
    'tn_profile_get_phase': (_int, [_int, _int, C.POINTER(C.c_uint64), C.POINTER(_f64), C.POINTER(_f64), C.POINTER(_f64)]),
}


def build(verbose=False):
    """Compile libtnpeps.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-o', LIB_PATH] + \
          os.environ.get('TN_EXTRA_HIPCC_FLAGS', '').split() + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, '..', 'include', 'tnpeps.h')]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


_lib = None


def lib():
    """The loaded library (raises RuntimeError when it is absent — build it with tnac4o_amd._lib.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libtnpeps.so not found at %s: run `python -c "import __graft_entry__ as g; g.build()"` '
                               '(the HIP library is the only backend; there is no CPU fallback)' % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


class TnError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        buf = C.create_string_buffer(512)
        lib().tn_last_error(buf, 512)
        raise TnError('libtnpeps error %d: %s' % (rc, buf.value.decode(errors='replace')))
