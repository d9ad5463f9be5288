"""ctypes binding of libtnpeps.so (include/tnpeps.h).  The HIP library is the only compute backend:
if it is missing or fails to load this module raises — there is no CPU fallback."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libtnpeps.so')
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['api.hip', 'gemm_f64.hip', 'small.hip', 'qr.hip', 'svd.hip', 'absorb.hip', 'misc.hip', 'beam.hip', 'prof.hip', 'tsqr.hip', 'cholqr.hip', 'smallqr.hip', 'peps.hip', 'env.hip', 'batch.hip', 'site.hip', 'chain.hip', 'beamsearch.hip']

_i64, _f64, _int, _ptr = C.c_int64, C.c_double, C.c_int, C.c_void_p

# name -> (restype, argtypes): every symbol include/tnpeps.h declares
SIGNATURES = {
    'tn_version': (_int, []),
    'tn_last_error': (_int, [C.c_char_p, _int]),
    'tn_build_id': (_int, [C.c_char_p, _int]),
    'tn_stream_create_masked': (_int, [C.POINTER(C.c_uint32), _int, C.POINTER(_ptr)]),
    'tn_stream_destroy': (_int, [_ptr]),
    'tn_gemm': (_int, [_i64, _i64, _i64, _f64, _ptr, _i64, _i64, _ptr, _i64, _i64, _f64, _ptr, _i64, _i64,
                       _i64, _i64, _i64, _i64, _ptr, _i64, _ptr]),
    'tn_gemm_ws_bytes': (_i64, [_i64, _i64, _i64, _i64]),
    'tn_absorb': (_int, [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _int, _i64, _i64, _i64, _i64, _ptr]),
    'tn_qr': (_int, [_ptr, _i64, _i64, _i64, _i64, _ptr, _i64, _i64, _ptr, _i64, _i64, _int, _f64, C.POINTER(_i64), _ptr, _i64,
              _ptr, _ptr]),
    'tn_qr_ws_bytes': (_i64, [_i64, _i64, _int]),
    'tn_panel_orth_ws_bytes': (_i64, [_i64, _int]),
    'tn_panel_orth': (_int, [_ptr, _i64, _i64, _i64, _int, _ptr, _i64, _i64, _int, C.POINTER(_int), C.POINTER(_f64), _ptr, _i64, _ptr]),
    'tn_panel_stats': (_int, [C.POINTER(C.c_uint64), _int]),
    'tn_panel_stats_stream': (_int, [C.POINTER(C.c_uint64), _int, _ptr]),
    'tn_smallqr_stats': (_int, [C.POINTER(C.c_uint64), _int, _ptr]),
    'tn_fused_timeouts': (_int, [C.POINTER(_int), _ptr]),
    'tn_qr_batched': (_int, [_ptr, _i64, _i64, _i64, _i64, _ptr, _i64, _i64, _ptr, _i64, _i64, _int, _f64, C.POINTER(_i64), _i64, _i64, _i64,
                      _i64, _ptr, _i64, _ptr, C.POINTER(_ptr), _int]),
    'tn_svd_trunc': (_int, [_ptr, _i64, _i64, _i64, _i64, _i64, _f64, _ptr, _i64, _i64, _ptr, _ptr, _i64, _i64,
                            C.POINTER(_i64), C.POINTER(_f64), C.POINTER(_int), C.POINTER(_int), _ptr, _i64, _ptr]),
    'tn_svdvals': (_int, [_ptr, _i64, _i64, _i64, _i64, C.POINTER(_f64), C.POINTER(_int), C.POINTER(_int), _ptr, _i64,
                          _ptr]),
    'tn_svd_ws_bytes': (_i64, [_i64, _i64, _int]),
    'tn_svdvals_async': (_int, [_ptr, _i64, _i64, _i64, _i64, _ptr, _ptr]),
    'tn_svdvals_small_batched': (_int, [_ptr, _i64, _ptr, _ptr, _ptr]),
    'tn_svd_trunc_batched': (_int, [_ptr, _i64, _i64, _i64, _i64, _i64, _f64, _ptr, _i64, _i64, _ptr, _ptr, _i64, _i64, C.POINTER(_i64),
                             C.POINTER(_f64), C.POINTER(_int), C.POINTER(_int), _i64, _i64, _i64, _i64, _i64, _ptr, _i64, _ptr]),
    'tn_svdvals_batched': (_int, [_ptr, _i64, _i64, _i64, _i64, C.POINTER(_f64), C.POINTER(_int), C.POINTER(_int), _i64, _i64, _ptr, _i64,
                           _ptr]),
    'tn_nfactor': (_int, [_ptr, _i64, _ptr, _ptr, _ptr]),
    'tn_scale_by': (_int, [_ptr, _i64, _ptr, _ptr]),
    'tn_normalize_pow2': (_int, [_ptr, _i64, _ptr, _ptr, _i64, _ptr]),
    'tn_scale_phys': (_int, [_ptr, _i64, _i64, _i64, _ptr, _int, _ptr]),
    'tn_calc_pn': (_int, [_ptr] * 9 + [_i64] * 7 + [_ptr, _ptr, _ptr, _ptr, _ptr]),
    'tn_merge_groups': (_int, [_ptr] * 5 + [_i64, _f64, _ptr, _ptr, _ptr, _ptr]),
    'tn_nfactor_batched': (_int, [_ptr, _i64, _i64, _ptr]),
    'tn_env_rr_batched': (_int, [_ptr] * 5 + [_i64] * 7 + [_ptr, _ptr]),
    'tn_env_rl_batched': (_int, [_ptr] * 3 + [_i64] * 3 + [_ptr, _ptr]),
    'tn_balance': (_int, [_ptr, _i64, _i64, _i64, _f64, _ptr, _ptr, _ptr]),
    'tn_peps_factor': (_int, [_ptr] * 9 + [_i64] * 3 + [_ptr, _ptr]),
    'tn_mpo_from_factor': (_int, [_ptr] * 3 + [_i64] * 5 + [_ptr, _ptr]),
    'tn_site_qr_ws_bytes': (_i64, [_int, _i64, _i64, _i64, _i64, _int]),
    'tn_site_qr': (_int, [_int, _ptr, _i64, _i64, _i64, _ptr, _i64, _ptr, _ptr, _f64, C.POINTER(_i64), _ptr, C.POINTER(_int), C.POINTER(_f64),
                   _int, C.POINTER(_i64), _ptr, _i64, _ptr]),
    'tn_gram_weights': (_int, [_ptr, _i64, _f64, _ptr, _ptr, _ptr]),
    'tn_rows_norm2': (_int, [_ptr, _i64, _i64, _ptr, _ptr]),
    'tn_gather_scale_rows': (_int, [_ptr, _i64, _i64, _ptr, _ptr, _ptr, _int, _ptr]),
    'tn_bond_deflate': (_int, [_int, _ptr, _i64, _i64, _ptr, _i64, _ptr, _ptr, C.POINTER(_i64), C.POINTER(_f64), _ptr, _i64, _ptr]),
    'tn_rar_ws_bytes': (_i64, [_i64] * 5),
    'tn_rar': (_int, [_ptr, _ptr, _ptr] + [_i64] * 5 + [_ptr, _ptr, _i64, _ptr]),
    'tn_env_mix_ws_bytes': (_i64, [_int] + [_i64] * 5),
    'tn_env_mix': (_int, [_int, _ptr, _ptr, _ptr] + [_i64] * 5 + [_ptr, _ptr, _i64, _ptr]),
    'tn_apply_truncation_ws_bytes': (_i64, [_i64] * 5),
    'tn_apply_truncation': (_int, [_ptr, _i64, _i64, _ptr, _i64, _i64, _i64, _ptr, _i64, _i64, _ptr, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr,
                            _i64, _ptr]),
    'tn_compress_mps_arena_bytes': (_i64, [_i64, C.POINTER(_i64), C.POINTER(_i64), _i64]),
    'tn_compress_mps': (_int, [_i64, C.POINTER(_ptr), C.POINTER(_i64), C.POINTER(_ptr), C.POINTER(_i64), _int, _i64, _f64, _f64, _int, _int, _int,
                        _ptr, _i64, C.POINTER(_i64), C.POINTER(_f64), C.POINTER(_f64), C.POINTER(_f64), _i64, C.POINTER(_i64), _ptr, _i64,
                        C.POINTER(_i64), C.POINTER(_f64), _ptr, _i64, _ptr]),
    'tn_argsort_desc': (_int, [_ptr, _i64, _ptr, _ptr]),
    'tn_weighted_sum': (_int, [_ptr, _ptr, _i64, _ptr, _ptr, _ptr]),
    'tn_beam_search_ws_bytes': (_i64, [_i64] * 7),
    'tn_beam_search': (_int, [_i64, _i64, _ptr, _i64, _int, _f64, _f64, _i64, _ptr, _ptr, _ptr, _ptr, C.POINTER(_i64), C.POINTER(_f64),
                       C.POINTER(_f64), _ptr, _i64, _ptr]),
    'tn_beam_search_team': (_int, [_i64, _i64, _ptr, _i64, _int, _f64, _f64, _i64, _ptr, _ptr, _ptr, _ptr, C.POINTER(_i64), C.POINTER(_f64),
                            C.POINTER(_f64), _ptr, _i64, _ptr, _int, _int, _ptr, _ptr]),
    'tn_profile_enable': (None, [C.c_uint]),
    'tn_profile_reset': (None, []),
    'tn_profile_sample': (None, [C.c_uint]),
    'tn_profile_get': (_int, [_int, C.POINTER(C.c_uint64), C.POINTER(_f64), C.POINTER(_f64), C.POINTER(_f64)]),
    'tn_profile_get_phase': (_int, [_int, _int, C.POINTER(C.c_uint64), C.POINTER(_f64), C.POINTER(_f64), C.POINTER(_f64)]),
}


def source_hash():
    """sha256 over csrc/* and include/tnpeps.h (sorted by name): identifies the sources a library was built from."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.h')))
    files.append(os.path.join(HERE, '..', 'include', 'tnpeps.h'))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:32]


def build(verbose=False):
    """Compile libtnpeps.so in-tree for gfx950 (hipcc cross-compiles without a GPU).  The hash of the sources is compiled
    in (tn_build_id) so that lib() can refuse a library that does not match the sources next to it.  Every .hip file is a
    translation unit of its own (no relocatable device code): the objects are compiled in parallel and cached under
    build/obj keyed by the content of the file, of every header of csrc/ and include/, and of the flags, so that an edit
    recompiles one file."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC'] + os.environ.get('TN_EXTRA_HIPCC_FLAGS', '').split()
    objdir = os.path.join(HERE, '..', 'build', 'obj')
    os.makedirs(objdir, exist_ok=True)
    hh = hashlib.sha256(' '.join(flags).encode())
    hh.update(os.path.realpath(hipcc).encode())                 # the compiler is part of the key: a toolchain upgrade must not link stale objects
    try:
        hh.update(subprocess.run([hipcc, '--version'], capture_output=True, check=False).stdout)
    except OSError:
        pass
    for f in sorted(os.listdir(CSRC)):
        if f.endswith('.h'):
            hh.update(open(os.path.join(CSRC, f), 'rb').read())
    hh.update(open(os.path.join(HERE, '..', 'include', 'tnpeps.h'), 'rb').read())
    src_hash = source_hash()

    def one(s):
        src = os.path.join(CSRC, s)
        h = hh.copy()
        h.update(open(src, 'rb').read())
        extra = []
        if s == 'api.hip':           # the only file that sees the build id
            extra = ['-DTN_SRC_HASH="%s"' % src_hash]
            h.update(src_hash.encode())
        obj = os.path.join(objdir, '%s.%s.o' % (s[:-4], h.hexdigest()[:16]))
        if not os.path.exists(obj):
            for old in os.listdir(objdir):
                if old.startswith(s[:-4] + '.') and old.endswith('.o'):
                    os.unlink(os.path.join(objdir, old))
            import tempfile
            fd, tmp = tempfile.mkstemp(prefix=s[:-4] + '.', suffix='.o.part', dir=objdir)     # (two builds at once must not share a temp name)
            os.close(fd)
            cmd = [hipcc] + flags + extra + ['-c', src, '-o', tmp]
            if verbose:
                print(' '.join(cmd), flush=True)
            try:
                subprocess.run(cmd, check=True)
                os.replace(tmp, obj)
            finally:
                if os.path.exists(tmp):
                    os.unlink(tmp)
        return obj

    with ThreadPoolExecutor(max_workers=int(os.environ.get('TN_BUILD_JOBS', '6'))) as ex:
        objs = list(ex.map(one, SOURCES))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + os.environ.get('TN_EXTRA_HIPCC_FLAGS', '').split() + ['-o', LIB_PATH] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB_PATH


def _stale(L):
    """True when the loaded library was built from other sources than the ones in the tree (content hash, so copying
    the tree to another machine does not matter)."""
    buf = C.create_string_buffer(64)
    L.tn_build_id.restype, L.tn_build_id.argtypes = _int, [C.c_char_p, _int]
    L.tn_build_id(buf, 64)
    return buf.value.decode() != source_hash()


# short, non-blocking entry points (see lib())
SHORT_CALLS = ('tn_gemm', 'tn_gemm_ws_bytes', 'tn_qr_ws_bytes', 'tn_svd_ws_bytes', 'tn_absorb', 'tn_nfactor', 'tn_scale_by',
               'tn_normalize_pow2', 'tn_scale_phys', 'tn_calc_pn', 'tn_nfactor_batched', 'tn_env_rr_batched', 'tn_env_rl_batched',
               'tn_balance', 'tn_merge_groups', 'tn_svdvals_async', 'tn_rar', 'tn_rar_ws_bytes', 'tn_env_mix', 'tn_env_mix_ws_bytes',
               'tn_apply_truncation', 'tn_apply_truncation_ws_bytes', 'tn_site_qr_ws_bytes', 'tn_gram_weights', 'tn_argsort_desc', 'tn_weighted_sum', 'tn_rows_norm2', 'tn_gather_scale_rows', 'tn_peps_factor', 'tn_mpo_from_factor', 'tn_last_error')
_lib = None
ABI_VERSION = 9          # bumped whenever a signature of include/tnpeps.h changes; must equal tn_version()


def lib():
    """The loaded library (raises RuntimeError when it is absent — build it with tnac4o_amd._lib.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libtnpeps.so not found at %s: run `python -c "import __graft_entry__ as g; g.build()"` '
                               '(the HIP library is the only backend; there is no CPU fallback)' % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.tn_version.restype = _int
        if not hasattr(L, 'tn_build_id') or (_stale(L) and not os.environ.get('TN_ALLOW_STALE_LIB')):
            raise RuntimeError('libtnpeps.so was built from other sources than csrc/*.hip + include/tnpeps.h in this tree: '
                               'rebuild it with `python -c "import __graft_entry__ as g; g.build()"` -- a stale library can '
                               'disagree with the ctypes signatures declared here')
        if L.tn_version() != ABI_VERSION:
            raise RuntimeError('libtnpeps.so implements C-ABI version %d, this package binds version %d: rebuild the library'
                               % (L.tn_version(), ABI_VERSION))
        # Entry points that only enqueue a launch or two return within microseconds: they are bound through PyDLL, i.e.
        # called WITHOUT releasing the GIL.  With 4 chains driven by 4 host threads, releasing and re-taking the GIL around
        # every 5 us call costs a futex wake-up each time (measured: 0.9 s of "python + GIL" per chain and sweep against
        # 0.29 s for a single chain, tools/host_overhead.py); the calls that block or enqueue hundreds of launches (tn_qr*,
        # tn_svd*) keep releasing it.  TN_PYDLL=0 restores plain CDLL for A/B measurements.
        if os.environ.get('TN_PYDLL', '1') != '0':
            P = C.PyDLL(LIB_PATH)
            for name in SHORT_CALLS:
                setattr(L, name, getattr(P, name))
            L._pydll = P
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
