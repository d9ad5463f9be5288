"""Thin tensor-level wrappers over the C-ABI (torch tensors in, torch tensors out).  PyTorch is used for device
memory and streams only; every contraction/factorisation below runs in libtnpeps."""
import ctypes as C
import os

import numpy as np
import torch

from ._lib import lib, check, TnError

QR_NB = 32
_ws = {}
_aux = {}
SCHMIDT_SIDE = os.environ.get('TN_SCHMIDT_SIDE', '0') == '1'   # deferred Schmidt-value checks on a side stream (off: with 4
# chains a second stream per chain costs 6-10 % of the step, measured; on the chain's own stream the launch is still asynchronous)
LOOKAHEAD = os.environ.get('TN_QR_LOOKAHEAD', '0') == '1'      # tn_qr look-ahead on a second stream per chain (off: no gain measured)


def register_aux_stream(main, aux):
    """Pair a chain's stream with the side stream tn_qr may use for its look-ahead (parallel.run_concurrent does this for
    the streams it creates, so the mapping of chains to hardware queues is deterministic)."""
    _aux[(main.device.index, main.cuda_stream)] = aux


def side_stream():
    """The side stream paired with the current stream (created on first use): used for work that is off a chain's critical
    path (deferred Schmidt-value checks, tn_qr's look-ahead when enabled)."""
    cur = torch.cuda.current_stream()
    key = (cur.device.index, cur.cuda_stream)
    a = _aux.get(key)
    if a is None:
        a = _aux[key] = torch.cuda.Stream()
    return a


def aux_stream():
    """The side stream paired with the current stream (created on first use), or None when look-ahead is disabled."""
    if not LOOKAHEAD:
        return None
    cur = torch.cuda.current_stream()
    key = (cur.device.index, cur.cuda_stream)
    a = _aux.get(key)
    if a is None:
        a = _aux[key] = torch.cuda.Stream()
    return a


try:                                   # raw handle of the current stream without building a Stream object (and without the
    _raw_stream = torch._C._cuda_getCurrentRawStream      # GIL hand-over a full torch call costs when 4 chains run)
    _raw_device = torch._C._cuda_getDevice
except AttributeError:                 # pragma: no cover - other torch builds
    _raw_stream = _raw_device = None


def _stream_handle():
    if _raw_stream is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _stream():
    return C.c_void_p(_stream_handle())


def _need_gpu(t):
    if not t.is_cuda:
        raise RuntimeError('tnac4o_amd operates on GPU tensors only (got a %s tensor); there is no CPU path' % t.device)
    if t.dtype != torch.float64:
        raise TypeError('float64 required')


def workspace(nbytes, slot=0):
    """A persistent scratch buffer of at least nbytes (grown on demand), private to the (device, stream) pair so that
    independent chains running on different streams never share scratch memory."""
    dev = _raw_device() if _raw_device is not None else torch.cuda.current_device()
    key = (dev, _stream_handle(), slot)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device='cuda')
        _ws[key] = buf
    return buf


def mm(A, B, out=None, alpha=1.0, beta=0.0):
    """out = alpha * A @ B + beta * out for 2-D views with arbitrary strides (tn_gemm)."""
    _need_gpu(A)
    _need_gpu(B)
    M, K = A.shape
    K2, N = B.shape
    assert K == K2, (A.shape, B.shape)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float64, device=A.device)
    L = lib()
    wsb = L.tn_gemm_ws_bytes(M, N, K, 1)
    ws = workspace(wsb, 1) if wsb > 0 else None
    check(L.tn_gemm(M, N, K, alpha, A.data_ptr(), A.stride(0), A.stride(1), B.data_ptr(), B.stride(0), B.stride(1),
                    beta, out.data_ptr(), out.stride(0), out.stride(1), 1, 0, 0, 0,
                    ws.data_ptr() if ws is not None else None, wsb, _stream()))
    return out


def bmm(A, B, out=None):
    """Batched out[b] = A[b] @ B[b]; A (batch|1, M, K), B (batch|1, K, N) with arbitrary strides."""
    _need_gpu(A)
    _need_gpu(B)
    batch = max(A.shape[0], B.shape[0])
    M, K = A.shape[1:]
    N = B.shape[2]
    if out is None:
        out = torch.empty((batch, M, N), dtype=torch.float64, device=A.device)
    if batch == 0:
        return out
    bsa = A.stride(0) if A.shape[0] > 1 else 0
    bsb = B.stride(0) if B.shape[0] > 1 else 0
    check(lib().tn_gemm(M, N, K, 1.0, A.data_ptr(), A.stride(1), A.stride(2), B.data_ptr(), B.stride(1), B.stride(2), 0.0,
                        out.data_ptr(), out.stride(1), out.stride(2), batch, bsa, bsb, out.stride(0), None, 0, _stream()))
    return out


def absorb(A, W, hconj):
    """One site of MPO.MPS absorption (tn_absorb).  A (Dl,p,Dr), W (ba,po,bb,pi), both contiguous."""
    _need_gpu(A)
    _need_gpu(W)
    A, W = A.contiguous(), W.contiguous()
    Dl, p, Dr = A.shape
    ba, po, bb, pi = W.shape
    pnew = pi if hconj else po
    out = torch.empty((Dl * ba, pnew, Dr * bb), dtype=torch.float64, device=A.device)
    check(lib().tn_absorb(A.data_ptr(), W.data_ptr(), out.data_ptr(), Dl, p, Dr, ba, po, bb, pi, 1 if hconj else 0, 1, 0, 0, 0,
                          _stream()))
    return out


def absorb_batched(A, W, hconj):
    """`batch` equally shaped sites in one launch: A (batch, Dl, p, Dr), W (batch | 1, ba, po, bb, pi) -- one MPO site
    absorbed into the same site of several boundary MPS when W has a leading 1.  Returns (batch, Dl*ba, pnew, Dr*bb)."""
    _need_gpu(A)
    _need_gpu(W)
    A, W = A.contiguous(), W.contiguous()
    batch, Dl, p, Dr = A.shape
    bw, ba, po, bb, pi = W.shape
    assert bw in (1, batch)
    pnew = pi if hconj else po
    out = torch.empty((batch, Dl * ba, pnew, Dr * bb), dtype=torch.float64, device=A.device)
    check(lib().tn_absorb(A.data_ptr(), W.data_ptr(), out.data_ptr(), Dl, p, Dr, ba, po, bb, pi, 1 if hconj else 0, batch,
                          A.stride(0), W.stride(0) if bw > 1 else 0, out.stride(0), _stream()))
    return out


def qr_batched(T, side_streams=(), rank_tol=0.0, nb=None):
    """Economic QR of every T[i] (batch, m, n; contiguous, destroyed) -- tn_qr_batched.  side_streams: torch streams on which
    the items run concurrently (forked from / joined into the current stream inside the call).  Returns Q (batch, m, k),
    R (batch, k, n) and the list of revealed ranks."""
    _need_gpu(T)
    assert T.is_contiguous() and T.dim() == 3
    batch, m, n = T.shape
    k = min(m, n)
    nb = nb or QR_NB
    Q = torch.empty((batch, m, k), dtype=torch.float64, device=T.device)
    R = torch.empty((batch, k, n), dtype=torch.float64, device=T.device)
    L = lib()
    wsi = (L.tn_qr_ws_bytes(m, n, nb) + 255) // 256 * 256
    ws = workspace(wsi * batch, 4)
    keff = (C.c_int64 * batch)(*([k] * batch))
    sides = (C.c_void_p * max(1, len(side_streams)))(*[s.cuda_stream for s in side_streams])
    # the call destroys T; error -7 (a launch with in-kernel barriers gave up: the streams are on the multi-launch forms from then on)
    # asks for a rerun from a copy, so one is kept (this entry point is not on the contraction path)
    T0 = T.clone()
    for attempt in range(2):
        rc = L.tn_qr_batched(T.data_ptr(), T.stride(1), T.stride(2), m, n, Q.data_ptr(), Q.stride(1), Q.stride(2), R.data_ptr(),
                             R.stride(1), R.stride(2), nb, float(rank_tol), keff, batch, T.stride(0), Q.stride(0), R.stride(0),
                             ws.data_ptr(), wsi * batch, _stream(), sides, len(side_streams))
        if rc == -7 and attempt == 0:
            T.copy_(T0)
            for i in range(batch):
                keff[i] = k
            continue
        check(rc)
        break
    return Q, R, [int(x) for x in keff]


def svd_trunc_batched(Cm, Dmax, tol):
    """Truncated SVDs of every Cm[i] (batch, k, n) in one call (tn_svd_trunc_batched).  Returns a list of the per-item tuples
    ops.svd_trunc returns."""
    _need_gpu(Cm)
    assert Cm.dim() == 3
    batch, k, n = Cm.shape
    cap = int(min(k, n, Dmax))
    U = torch.empty((batch, k, cap), dtype=torch.float64, device=Cm.device)
    S = torch.empty((batch, cap), dtype=torch.float64, device=Cm.device)
    Vt = torch.empty((batch, cap, n), dtype=torch.float64, device=Cm.device)
    L = lib()
    wsb = L.tn_svd_ws_bytes(k, n, 1)
    ws = workspace(wsb, 0)
    keep, disc = (C.c_int64 * batch)(), (C.c_double * batch)()
    sweeps, info = (C.c_int * batch)(), (C.c_int * batch)()
    check(L.tn_svd_trunc_batched(Cm.data_ptr(), Cm.stride(1), Cm.stride(2), k, n, cap if Dmax >= cap else int(Dmax), float(tol),
                                 U.data_ptr(), U.stride(1), U.stride(2), S.data_ptr(), Vt.data_ptr(), Vt.stride(1), Vt.stride(2),
                                 keep, disc, sweeps, info, batch, Cm.stride(0), U.stride(0), S.stride(0), Vt.stride(0),
                                 ws.data_ptr(), wsb, _stream()))
    out = []
    for i in range(batch):
        if info[i] != 0:
            raise TnError('tn_svd_trunc_batched: item %d did not converge (%d sweeps)' % (i, sweeps[i]))
        kp = int(keep[i])
        out.append((U[i, :, :kp], S[i, :kp], Vt[i, :kp], kp, float(disc[i]), dict(sweeps=sweeps[i], info=info[i])))
    return out


RANK_TOL = 2.0 ** -56         # the deflation threshold of the Jacobi SVD (svd.hip)


def qr_into(T, Q, R, overwrite=False, nb=None, rank_tol=0.0):
    """Economic QR of the 2-D view T into the (strided) views Q (m x k) and R (k x n); diag(R) >= 0.
    rank_tol > 0 enables the early exit of tn_qr; the number of columns produced is returned as third value
    (use Q[:, :keff], R[:keff])."""
    _need_gpu(T)
    m, n = T.shape
    nb = nb or QR_NB
    L = lib()
    wsb = L.tn_qr_ws_bytes(m, n, nb)
    ws = workspace(wsb, 0)
    keff = C.c_int64(min(m, n))
    aux = aux_stream() if (nb == 32 and m >= 2048 and min(m, n) >= 128) else None
    src = T
    for attempt in range(2):
        if not overwrite:
            T = src.clone(memory_format=torch.preserve_format)
        rc = L.tn_qr(T.data_ptr(), T.stride(0), T.stride(1), m, n, Q.data_ptr(), Q.stride(0), Q.stride(1), R.data_ptr(),
                     R.stride(0), R.stride(1), nb, float(rank_tol), C.byref(keff), ws.data_ptr(), wsb, _stream(),
                     C.c_void_p(aux.cuda_stream) if aux is not None else None)
        # -7: a single-launch panel step gave up at a barrier (results invalid, input overwritten); the stream has been taken off
        # those launch forms, so a second run from the untouched source takes the six-launch chain
        if rc != -7 or overwrite or attempt == 1:
            check(rc)
            break
    return Q, R, int(keff.value)


def panel_orth(X, method=0, state=False, out=None):
    """The panel step of tn_qr on its own (tn_panel_orth): an orthonormal basis of the column space of the (strided) n x b panel X,
    b <= 32.  method 0: iterated Cholesky-QR with deferral (what tn_qr uses), 1: Householder TSQR.  With state=True also returns
    the panel's state record (list of 9 ints, see include/tnpeps.h) and max|X^T X - I| before each pass (synchronises)."""
    _need_gpu(X)
    n, b = X.shape
    L = lib()
    wsb = L.tn_panel_orth_ws_bytes(n, b)
    ws = workspace(wsb, 7)
    Y = out if out is not None else torch.empty_like(X)
    st9 = (C.c_int * 9)() if state else None
    dev = (C.c_double * 8)() if state else None
    check(L.tn_panel_orth(X.data_ptr(), X.stride(0), X.stride(1), n, b, Y.data_ptr(), Y.stride(0), Y.stride(1), int(method), st9, dev,
                          ws.data_ptr(), wsb, _stream()))
    if state:
        return Y, list(st9), [dev[i] for i in range(6)]
    return Y


PANEL_STAT_KEYS = ('panels', 'substitution_passes', 'deferred_pivots', 'refilled_columns', 'householder_fallbacks', 'panels_with_3_or_more_passes',
                   'panels_with_4_or_more_passes', 'pass_elements_six_launch_chain', 'pass_elements_single_launch', 'single_launch_panels')


def panel_stats(reset=False, stream=False):
    """Diagnostic counters of the Cholesky-QR panel step since the last reset: dict.  stream=False: summed over all streams
    (tn_panel_stats); stream=True: of the panels launched on the current stream only (tn_panel_stats_stream)."""
    st = (C.c_uint64 * 16)()
    if stream:
        check(lib().tn_panel_stats_stream(st, 1 if reset else 0, _stream()))
    else:
        check(lib().tn_panel_stats(st, 1 if reset else 0))
    return {k: int(st[i]) for i, k in enumerate(PANEL_STAT_KEYS)}


def smallqr_stats(reset=False):
    """Diagnostic counters of the one-launch factorisations (csrc/smallqr.hip) on the current stream: dict."""
    st = (C.c_uint64 * 4)()
    check(lib().tn_smallqr_stats(st, 1 if reset else 0, _stream()))
    return dict(calls=int(st[0]), passes=int(st[1]), householder_fallbacks=int(st[2]), timeouts=int(st[3]))


def fused_timeouts():
    """Launches with in-kernel barriers of the current stream that gave up since the last check (synchronises)."""
    n = C.c_int(0)
    check(lib().tn_fused_timeouts(C.byref(n), _stream()))
    return int(n.value)


def qr(T, overwrite=False, nb=None):
    """Economic QR (Q, R) of a 2-D view; T is preserved unless overwrite is set."""
    m, n = T.shape
    k = min(m, n)
    Q = torch.empty((m, k), dtype=torch.float64, device=T.device)
    R = torch.empty((k, n), dtype=torch.float64, device=T.device)
    qr_into(T, Q, R, overwrite, nb)
    return Q, R


def _svd_trunc_raw(Cm, Dmax, tol):
    k, n = Cm.shape
    cap = int(min(k, n, Dmax))
    U = torch.empty((k, cap), dtype=torch.float64, device=Cm.device)
    S = torch.empty((cap,), dtype=torch.float64, device=Cm.device)
    Vt = torch.empty((cap, n), dtype=torch.float64, device=Cm.device)
    L = lib()
    wsb = L.tn_svd_ws_bytes(k, n, 1)
    ws = workspace(wsb, 0)
    keep, disc, sweeps, info = C.c_int64(0), C.c_double(0.0), C.c_int(0), C.c_int(0)
    check(L.tn_svd_trunc(Cm.data_ptr(), Cm.stride(0), Cm.stride(1), k, n, cap if Dmax >= cap else int(Dmax), float(tol),
                         U.data_ptr(), U.stride(0), U.stride(1), S.data_ptr(), Vt.data_ptr(), Vt.stride(0), Vt.stride(1),
                         C.byref(keep), C.byref(disc), C.byref(sweeps), C.byref(info), ws.data_ptr(), wsb, _stream()))
    kp = int(keep.value)
    return U[:, :kp], S[:kp], Vt[:kp], kp, float(disc.value), dict(sweeps=sweeps.value, info=info.value)


def _sign_gauge_(U, Vt):
    """The reference's sign convention (mps.py:35-39) on a finished factorisation: flip the pairs (column of U, row of Vt)
    in which the most negative entry outweighs the most positive one in both."""
    if U.shape[1] == 0:
        return
    flip = (U.min(dim=0).values.abs() > U.max(dim=0).values) & (Vt.min(dim=1).values.abs() > Vt.max(dim=1).values)
    sg = torch.where(flip, -1.0, 1.0).to(torch.float64)
    U.mul_(sg[None, :])
    Vt.mul_(sg[:, None])


def svd_trunc(Cm, Dmax, tol):
    """Truncated SVD of the 2-D view Cm.  Returns (U[:, :keep], S[:keep], Vt[:keep], keep, discarded, info).

    The one-sided Jacobi kernel converges in a handful of sweeps on what the contraction path feeds it (triangular factors
    of a QR).  On a general ill-conditioned matrix (nearly parallel rows) it can exhaust its sweep cap; then the classical
    remedy is applied here: factor the tall orientation, A = Q1 R1, R1^T = Q2 R2 (two QRs leave a nearly diagonal
    triangle), take the Jacobi SVD of R2 and fold the orthogonal factors back.  The reference's LAPACK call raises
    LinAlgError when it does not converge (mps.py:31-34); this raises TnError if even the preconditioned run fails."""
    _need_gpu(Cm)
    out = _svd_trunc_raw(Cm, Dmax, tol)
    if out[5]['info'] == 0:
        return out
    k, n = Cm.shape
    tall = Cm if k >= n else Cm.t()
    Q1, R1 = qr(tall)                                     # tall = Q1 R1
    Q2, R2 = qr(R1.t().contiguous())                      # R1^T = Q2 R2   =>  tall = Q1 R2^T Q2^T
    U2, S, V2t, kp, disc, info = _svd_trunc_raw(R2, Dmax, tol)     # R2 = U2 S V2t  =>  tall = (Q1 V2) S (Q2 U2)^T
    if info['info'] != 0:
        raise TnError('tn_svd_trunc: Jacobi sweeps did not converge on a %d x %d matrix (%d sweeps, after QR preconditioning)'
                      % (k, n, info['sweeps']))
    left = mm(Q1, V2t.t())                                # (rows of tall) x keep
    right = mm(U2.t(), Q2.t())                            # keep x (cols of tall)
    U, Vt = (left, right) if k >= n else (right.t().contiguous(), left.t().contiguous())
    _sign_gauge_(U, Vt)
    info = dict(info, preconditioned=True, first_attempt_sweeps=out[5]['sweeps'])
    return U, S, Vt, kp, disc, info


def svdvals(Cm, _preconditioned=False):
    """Singular values (host numpy array, descending) of the 2-D view Cm (same fallback as svd_trunc)."""
    _need_gpu(Cm)
    k, n = Cm.shape
    out = np.empty(min(k, n), dtype=np.float64)
    L = lib()
    wsb = L.tn_svd_ws_bytes(k, n, 0)
    ws = workspace(wsb, 0)
    sweeps, info = C.c_int(0), C.c_int(0)
    check(L.tn_svdvals(Cm.data_ptr(), Cm.stride(0), Cm.stride(1), k, n, out.ctypes.data_as(C.POINTER(C.c_double)),
                       C.byref(sweeps), C.byref(info), ws.data_ptr(), wsb, _stream()))
    if info.value != 0:
        if _preconditioned:
            raise TnError('tn_svdvals: Jacobi sweeps did not converge on a %d x %d matrix (%d sweeps, after QR preconditioning)'
                          % (k, n, sweeps.value))
        _, R1 = qr(Cm if k >= n else Cm.t())
        _, R2 = qr(R1.t().contiguous())
        return svdvals(R2, _preconditioned=True)
    return out


def svdvals_async(Cm, out66, stream=None):
    """Schmidt values of a centre matrix with both dimensions <= 64 into the device buffer out66 (66 doubles: 64 values
    sorted descending, sweeps, converged flag) without synchronising (tn_svdvals_async).  `stream`: a torch stream to launch
    on instead of the current one (the caller orders it after the producer of Cm and keeps Cm alive until it joins)."""
    _need_gpu(Cm)
    k, n = Cm.shape
    st = C.c_void_p(stream.cuda_stream) if stream is not None else _stream()
    check(lib().tn_svdvals_async(Cm.data_ptr(), Cm.stride(0), Cm.stride(1), k, n, out66.data_ptr(), st))


def svdvals_small_batched(mats):
    """Schmidt values of several centre matrices (both dimensions <= 64 each) in one launch (tn_svdvals_small_batched).  Returns the
    device table (len(mats), 66): 64 values sorted descending, sweeps, converged flag per row; no synchronisation."""
    n = len(mats)
    desc = np.empty((n, 5), dtype=np.int64)
    for i, Cm in enumerate(mats):
        _need_gpu(Cm)
        k, m = Cm.shape
        desc[i] = (Cm.data_ptr(), Cm.stride(0), Cm.stride(1), k, m) if k <= m else (Cm.data_ptr(), Cm.stride(1), Cm.stride(0), m, k)
    dev = mats[0].device
    ddesc = torch.from_numpy(desc).to(dev)
    out = torch.empty((n, 66), dtype=torch.float64, device=dev)
    check(lib().tn_svdvals_small_batched(ddesc.data_ptr(), n, desc.ctypes.data_as(C.c_void_p), out.data_ptr(), _stream()))
    return out


# ---- fused site steps (csrc/site.hip) ------------------------------------------------------------------------------------
FUSED_SITE = os.environ.get('TN_FUSED_SITE', '1') != '0'
PASS1_WEIGHTED = os.environ.get('TN_PASS1_WEIGHTED', '1') != '0'     # weighted rank-revealing first canonisation pass (mps.py)
PASS1_PIVOT = os.environ.get('TN_PASS1_PIVOT', '1') != '0'           # ... with panel pivoting inside tn_qr
PASS1_TRACE = os.environ.get('TN_PASS1_TRACE', '0') == '1'
_wsq = {}


def _ws_query(name, *args):
    key = (name,) + args
    v = _wsq.get(key)
    if v is None:
        v = _wsq[key] = int(getattr(lib(), name)(*args))
    return v


def site_qr(side, A, Cm=None, rank_tol=0.0, normalise=True, info=None, frobenius_exit=False, pivot=False):
    """One canonisation step in one call (tn_site_qr): attach the centre matrix Cm (side 0: Cm . A, side 1: A . Cm; None = no
    attach, A is consumed), QR with diag(R) >= 0, power-of-two normalisation of the triangular factor.
    side 0 returns (Q (l p x k), R (k x Dr), k, nf);  side 1 returns (Q^T (k x p r), R^T (Dl x k), k, nf), nf = device [nf, 1/nf].
    After a rank-revealing early exit (k below the full rank) the factors are sliced and normalised here.
    normalise=False leaves the triangular factor as it is (nf = None); `info` (dict) receives 'dropped2', the squared
    Frobenius norm of the trailing block an early exit dropped."""
    Dl, p, Dr = A.shape
    attach = Cm is not None
    kc = (Cm.shape[0] if side == 0 else Cm.shape[1]) if attach else 0
    assert A.is_contiguous() and (not attach or (Cm.is_contiguous() and (Cm.shape[1] == Dl if side == 0 else Cm.shape[0] == Dr)))
    if side == 0:
        m, n = (kc if attach else Dl) * p, Dr
    else:
        m, n = p * (kc if attach else Dr), Dl
    kf = min(m, n)
    dev = A.device
    Q = torch.empty((m, kf) if side == 0 else (kf, m), dtype=torch.float64, device=dev)
    R = torch.empty((kf, n) if side == 0 else (n, kf), dtype=torch.float64, device=dev)
    nf = torch.empty(2, dtype=torch.float64, device=dev)
    wsb = _ws_query('tn_site_qr_ws_bytes', side, Dl, p, Dr, kc, 1 if attach else 0)
    ws = workspace(wsb, 0)
    keff, normd, drop2 = C.c_int64(kf), C.c_int(0), C.c_double(0.0)
    piv = (C.c_int64 * n)() if pivot else None               # panel pivoting: order of the factored matrix's columns
    # error -7 (a launch with in-kernel barriers gave up; the stream is on the multi-launch forms from then on) asks for a rerun: with
    # an attach the input is intact (the product lives in the workspace), without one the call consumes A, so a copy is kept
    A0 = None if attach else A.clone()
    for attempt in range(2):
        rc = lib().tn_site_qr(side, A.data_ptr(), Dl, p, Dr, Cm.data_ptr() if attach else None, kc, Q.data_ptr(), R.data_ptr(),
                              float(rank_tol), C.byref(keff), nf.data_ptr() if normalise else None, C.byref(normd), C.byref(drop2),
                              1 if frobenius_exit else 0, piv, ws.data_ptr(), wsb, _stream())
        if rc == -7 and attempt == 0:
            if A0 is not None:
                A.copy_(A0)
            keff, normd, drop2 = C.c_int64(kf), C.c_int(0), C.c_double(0.0)
            continue
        check(rc)
        break
    k = int(keff.value)
    if info is not None:
        info['dropped2'] = float(drop2.value)
        if pivot:
            info['perm'] = torch.as_tensor(np.frombuffer(piv, dtype=np.int64).copy()).to(A.device)
    if k < kf:
        if side == 0:
            Q, R = Q[:, :k].contiguous(), R[:k].contiguous()
        else:
            Q, R = Q[:k].contiguous(), R[:, :k].contiguous()
    if not normalise:
        nf = None
    elif not normd.value:
        nf = normalize_pow2_(R)
    return Q, R, k, nf


def gram_weights(G, floor_rel):
    """(d2, stats) of tn_gram_weights: floored squared weights of a bond's indices from the Gram matrix of the part on its
    other side, and the 65 statistics [ 64 partial sums of ||K||_F^2, max G_cc ] (device tensors)."""
    n = G.shape[0]
    assert G.is_contiguous() and G.shape == (n, n)
    d2 = torch.empty(n, dtype=torch.float64, device=G.device)
    st = torch.empty(65, dtype=torch.float64, device=G.device)
    check(lib().tn_gram_weights(G.data_ptr(), n, float(floor_rel), d2.data_ptr(), st.data_ptr(), _stream()))
    return d2, st


def rows_norm2(A2d):
    """Squared norms of the rows of a contiguous 2-D tensor (tn_rows_norm2)."""
    assert A2d.is_contiguous() and A2d.dim() == 2
    out = torch.empty(A2d.shape[0], dtype=torch.float64, device=A2d.device)
    check(lib().tn_rows_norm2(A2d.data_ptr(), A2d.shape[0], A2d.shape[1], out.data_ptr(), _stream()))
    return out


def gather_scale_rows(A2d, perm, w2, inverse=False):
    """inverse=False: out[j] = sqrt(w2[perm[j]]) A2d[perm[j]];  inverse=True: out[perm[j]] = A2d[j] / sqrt(w2[perm[j]])."""
    assert A2d.is_contiguous() and perm.dtype == torch.int64 and perm.is_contiguous() and w2.is_contiguous()
    out = torch.empty_like(A2d)
    check(lib().tn_gather_scale_rows(A2d.data_ptr(), A2d.shape[0], A2d.shape[1], perm.data_ptr(), w2.data_ptr(), out.data_ptr(),
                                     1 if inverse else 0, _stream()))
    return out


def argsort_desc(w):
    """Stable descending argsort of a contiguous device vector (tn_argsort_desc; NaN first, ties by increasing index)."""
    assert w.is_contiguous() and w.dim() == 1
    out = torch.empty(w.numel(), dtype=torch.int64, device=w.device)
    check(lib().tn_argsort_desc(w.data_ptr(), w.numel(), out.data_ptr(), _stream()))
    return out


def weighted_sum(a, b):
    """(a * b, its sum as a 1-element device tensor) with a fixed summation order (tn_weighted_sum)."""
    assert a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel()
    w = torch.empty_like(a)
    s = torch.empty(1, dtype=torch.float64, device=a.device)
    check(lib().tn_weighted_sum(a.data_ptr(), b.data_ptr(), a.numel(), w.data_ptr(), s.data_ptr(), _stream()))
    return w, s


NATIVE_CHAIN = os.environ.get('TN_NATIVE_CHAIN', '1') != '0'     # compress_mps (and the absorption in front of it) as one library call


def gauge_svd_mode():
    """TN_GAUGE_SVD: 1 keeps every decomposition of the intermediate passes that cannot truncate, 2 those of the 2 chi pass
    (chain.hip: gauge_svd_skippable; read per call)."""
    try:
        return int(os.environ.get('TN_GAUGE_SVD', '0'))
    except ValueError:
        return 0


def bond_deflate_on():
    return os.environ.get('TN_BOND_DEFLATE', '1') != '0'

_arena = {}


def chain_arena(nbytes):
    """The arena of tn_compress_mps for the current (device, stream): one persistent torch byte buffer, grown on demand."""
    dev = _raw_device() if _raw_device is not None else torch.cuda.current_device()
    key = (dev, _stream_handle())
    buf = _arena.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = None
        _arena.pop(key, None)
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device='cuda')
        _arena[key] = buf
    return buf


def compress_mps_native(sites, mpo_sites, hconj, Dmax, tolS, tolV, max_sweeps, graduate, weighted=True, structured=True, lazy=True):
    """apply_mpo + compress_mps of one boundary MPS in one library call (tn_compress_mps).  sites: list of contiguous (Dl, p, Dr)
    device tensors; mpo_sites: list of contiguous (ba, po, bb, pi) tensors / None per site, or None.  Returns a dict with the
    compressed sites `A`, `overlap`, `discarded`, `S` (list of numpy arrays / None per bond), `nfs` (device pairs), `info`."""
    L = len(sites)
    for t in sites:
        _need_gpu(t)
        assert t.is_contiguous() and t.dim() == 3
    sd = (C.c_int64 * (3 * L))(*[int(x) for t in sites for x in t.shape])
    sp = (C.c_void_p * L)(*[t.data_ptr() for t in sites])
    have_mpo = mpo_sites is not None and any(w is not None for w in mpo_sites)
    md = mp = None
    p_out = [int(t.shape[1]) for t in sites]
    if have_mpo:
        dims, ptrs = [], []
        for n, w in enumerate(mpo_sites):
            if w is None:
                dims += [0, 0, 0, 0]
                ptrs.append(None)
            else:
                _need_gpu(w)
                assert w.is_contiguous() and w.dim() == 4
                dims += [int(x) for x in w.shape]
                ptrs.append(w.data_ptr())
                p_out[n] = int(w.shape[3] if hconj else w.shape[1])
        md = (C.c_int64 * (4 * L))(*dims)
        mp = (C.c_void_p * L)(*ptrs)
    Lb = lib()
    need = int(Lb.tn_compress_mps_arena_bytes(L, sd, md, int(Dmax)))
    arena = chain_arena(need)
    dev = sites[0].device
    cap = int(Dmax)
    slot = max(cap * p * cap for p in p_out)
    out = torch.empty((L, slot), dtype=torch.float64, device=dev)
    od = (C.c_int64 * (3 * L))()
    overlap = C.c_double(0.0)
    disc = (C.c_double * (L + 1))()
    pitch = max(cap, 1)
    sch = (C.c_double * ((L + 1) * pitch))()
    slen = (C.c_int64 * (L + 1))()
    nfs_cap = max(64, 2 * int(max_sweeps) + 16) * L + 64         # ~2L pairs per variational sweep + ~5L for the canonisation passes; the library fails (-3) rather than reuse a slot
    nfs = torch.empty((nfs_cap, 2), dtype=torch.float64, device=dev)
    ncount = C.c_int64(0)
    info = (C.c_double * 8)()
    flags = (1 if weighted else 0) | (2 if structured else 0) | (4 if lazy else 0)
    ab = arena.data_ptr()
    off = (-ab) % 256
    for attempt in range(2):
        rc = Lb.tn_compress_mps(L, sp, sd, mp, md, 1 if hconj else 0, int(Dmax), float(tolS), float(tolV), int(max_sweeps), 1 if graduate else 0,
                                flags, out.data_ptr(), slot, od, C.byref(overlap), disc, sch, pitch, slen, nfs.data_ptr(), nfs_cap, C.byref(ncount),
                                info, C.c_void_p(ab + off), arena.numel() - off, _stream())
        if rc == -3 and attempt == 0 and arena.numel() < int(Lb.tn_compress_mps_arena_bytes(L, sd, md, -1)):
            # the typical arena did not do (e.g. the weighted first pass fell back to the plain one): once more with the conservative bound
            torch.cuda.current_stream().synchronize()
            arena = None
            arena = chain_arena(int(Lb.tn_compress_mps_arena_bytes(L, sd, md, -1)))
            ab = arena.data_ptr()
            off = (-ab) % 256
            continue
        check(rc)
        break
    A = []
    for n in range(L):
        a, b, c = int(od[3 * n]), int(od[3 * n + 1]), int(od[3 * n + 2])
        A.append(out[n, :a * b * c].view(a, b, c))
    S = []
    for i in range(L + 1):
        k = int(slen[i])
        S.append(None if k < 0 else np.array(sch[i * pitch:i * pitch + k], dtype=np.float64))
    return dict(A=A, overlap=float(overlap.value), discarded=[float(x) for x in disc], S=S, nfs=[nfs[i] for i in range(int(ncount.value))],
                info=dict(reveal_error_bound=float(info[0]), reveal_fallbacks=int(info[1]), weighted_used=bool(info[2]), arena_peak=int(info[3]), bonds_before=int(info[4]), bonds_after=int(info[5]), redone=int(info[6]), gauge_skipped=int(info[7]) % 65536, target_swapped=(int(info[7]) // 65536) % 2, var1_skipped=(int(info[7]) // 131072) % 2, attach_fused=int(info[7]) // 262144,
                          arena_bytes=int(arena.numel())))


def bond_deflate(side, Cm, site):
    """tn_bond_deflate: drop the bond indices between the centre matrix and its orthonormal site that carry at most eps^2 of the
    largest one's weight in total (see include/tnpeps.h).  side 0: Cm (k, n), site (Dl, p, k); side 1: Cm (n, k), site (k, p, Dr).
    Returns (Cm', site', k', dropped2_rel); the inputs themselves when nothing is dropped."""
    _need_gpu(Cm)
    Cm = Cm if Cm.is_contiguous() else Cm.contiguous()
    site = site if site.is_contiguous() else site.contiguous()
    if side == 0:
        k, n = Cm.shape
        m = site.shape[0] * site.shape[1]
        assert site.shape[2] == k
    else:
        n, k = Cm.shape
        m = site.shape[1] * site.shape[2]
        assert site.shape[0] == k
    if k < 2 or k > 256:
        return Cm, site, k, 0.0
    Co = torch.empty_like(Cm)
    So = torch.empty_like(site)
    ws = workspace(8192, 3)
    kk = C.c_int64(k)
    d2 = C.c_double(0.0)
    check(lib().tn_bond_deflate(side, Cm.data_ptr(), k, n, site.data_ptr(), m, Co.data_ptr(), So.data_ptr(), C.byref(kk), C.byref(d2),
                                ws.data_ptr(), 8192, _stream()))
    kk = int(kk.value)
    if kk == k:
        return Cm, site, k, 0.0
    if side == 0:
        return Co.view(-1)[:kk * n].view(kk, n), So.view(-1)[:m * kk].view(site.shape[0], site.shape[1], kk), kk, float(d2.value)
    return Co.view(-1)[:n * kk].view(n, kk), So.view(-1)[:kk * m].view(kk, site.shape[1], site.shape[2]), kk, float(d2.value)


def rar(RL, A, RR):
    """RL . A . RR -> (c, s, c2) (tn_rar; MPS._mps_RAR)."""
    a, s_, a2 = A.shape
    c, c2 = RL.shape[0], RR.shape[1]
    assert A.is_contiguous() and RL.shape[1] == a and RR.shape[0] == a2
    RL, RR = RL if RL.is_contiguous() else RL.contiguous(), RR if RR.is_contiguous() else RR.contiguous()
    out = torch.empty((c, s_, c2), dtype=torch.float64, device=A.device)
    wsb = _ws_query('tn_rar_ws_bytes', c, a, s_, a2, c2)
    ws = workspace(wsb, 1)
    check(lib().tn_rar(RL.data_ptr(), A.data_ptr(), RR.data_ptr(), c, a, s_, a2, c2, out.data_ptr(), ws.data_ptr(), wsb, _stream()))
    return out


def env_mix(side, Rm, A, Ac):
    """Mixed environment update (tn_env_mix; MPS._mps_RL for side 0, _mps_RR for side 1)."""
    a, s_, a2 = A.shape
    c, _, c2 = Ac.shape
    assert A.is_contiguous() and Ac.is_contiguous() and tuple(Rm.shape) == ((c, a) if side == 0 else (a2, c2))
    Rm = Rm if Rm.is_contiguous() else Rm.contiguous()
    out = torch.empty((c2, a2) if side == 0 else (a, c), dtype=torch.float64, device=A.device)
    wsb = _ws_query('tn_env_mix_ws_bytes', side, a, s_, a2, c, c2)
    ws = workspace(wsb, 1)
    check(lib().tn_env_mix(side, Rm.data_ptr(), A.data_ptr(), Ac.data_ptr(), a, s_, a2, c, c2, out.data_ptr(), ws.data_ptr(), wsb,
                           _stream()))
    return out


def apply_truncation(Al, U, S, Vt, Ar):
    """Projectors of a truncation into the neighbours + diagonal centre (tn_apply_truncation).  Al (Dl, p, k0), Ar (k1, p2, Dr),
    U (k0 x keep), Vt (keep x k1) as strided views.  Returns (Al_new (Dl, p, keep), Ar_new (keep, p2, Dr), diag(S))."""
    Dl, p, k0 = Al.shape
    k1, p2, Dr = Ar.shape
    keep = S.numel()
    dev = Al.device
    Aln = torch.empty((Dl, p, keep), dtype=torch.float64, device=dev)
    Arn = torch.empty((keep, p2, Dr), dtype=torch.float64, device=dev)
    Cd = torch.empty((keep, keep), dtype=torch.float64, device=dev)
    wsb = _ws_query('tn_apply_truncation_ws_bytes', Dl * p, k0, keep, k1, p2 * Dr)
    ws = workspace(max(wsb, 256), 1)
    check(lib().tn_apply_truncation(Al.data_ptr(), Dl * p, k0, U.data_ptr(), U.stride(0), U.stride(1), keep, Vt.data_ptr(), Vt.stride(0),
                                    Vt.stride(1), Ar.data_ptr(), k1, p2 * Dr, S.data_ptr(), Aln.data_ptr(), Arn.data_ptr(), Cd.data_ptr(),
                                    ws.data_ptr(), wsb, _stream()))
    return Aln, Arn, Cd


def nfactor_dev(T):
    """Device tensor [nf, 1/nf] with nf = 2^floor(log2 max|T|) (tn_nfactor).  T must be contiguous."""
    _need_gpu(T)
    assert T.is_contiguous()
    out = torch.empty(3, dtype=torch.float64, device=T.device)      # [nf, 1/nf, scratch slot]
    check(lib().tn_nfactor(T.data_ptr(), T.numel(), out.data_ptr(), out.data_ptr() + 16, _stream()))
    return out


def scale_(T, scalar_dev):
    """T *= scalar_dev[0] in place (T contiguous)."""
    assert T.is_contiguous()
    check(lib().tn_scale_by(T.data_ptr(), T.numel(), scalar_dev.data_ptr(), _stream()))
    return T


def normalize_pow2_(T):
    """T /= nfactor(T) in place; returns the device pair [nf, 1/nf] (tn_normalize_pow2)."""
    _need_gpu(T)
    assert T.is_contiguous()
    out = torch.empty(2, dtype=torch.float64, device=T.device)
    scratch = workspace(8192, 3)
    check(lib().tn_normalize_pow2(T.data_ptr(), T.numel(), out.data_ptr(), scratch.data_ptr(), 8192, _stream()))
    return out


def scale_phys_(A, diag, inv=False):
    assert A.is_contiguous() and diag.is_contiguous()
    Dl, p, Dr = A.shape
    assert diag.numel() >= p
    check(lib().tn_scale_phys(A.data_ptr(), Dl, p, Dr, diag.data_ptr(), 1 if inv else 0, _stream()))
    return A


def nfactor_batched_(X):
    """Each X[b] (contiguous) divided by its own nfactor."""
    assert X.is_contiguous()
    b = X.shape[0]
    if b:
        check(lib().tn_nfactor_batched(X.data_ptr(), b, X.numel() // b, _stream()))
    return X


def calc_pn(T1, RR, F, dmap, rmap, pref, suf, lidx, uidx, parent_log2p=None):
    """Batched conditional probabilities (tn_calc_pn).  Returns (P (nb,q), minP (nb)); with parent_log2p (nb) also the expanded
    log-probabilities log2(P) + parent_log2p[:, None] as third value."""
    nb = pref.numel()
    q, nl, nu = F.shape
    _, p, Dr = T1.shape
    br = RR.shape[2]
    for t in (T1, RR, F, dmap, rmap, pref, suf, lidx, uidx):
        assert t.is_contiguous() and t.is_cuda
    P = torch.empty((nb, q), dtype=torch.float64, device=T1.device)
    mP = torch.empty((nb,), dtype=torch.float64, device=T1.device)
    LP = None
    if parent_log2p is not None:
        assert parent_log2p.is_contiguous() and parent_log2p.numel() == nb and parent_log2p.dtype == torch.float64
        LP = torch.empty((nb, q), dtype=torch.float64, device=T1.device)
    check(lib().tn_calc_pn(T1.data_ptr(), RR.data_ptr(), F.data_ptr(), dmap.data_ptr(), rmap.data_ptr(), pref.data_ptr(),
                           suf.data_ptr(), lidx.data_ptr(), uidx.data_ptr(), nb, q, nl, nu, p, Dr, br, P.data_ptr(),
                           mP.data_ptr(), parent_log2p.data_ptr() if LP is not None else None, LP.data_ptr() if LP is not None else None,
                           _stream()))
    if LP is not None:
        return P, mP, LP
    return P, mP


def merge_groups(E, lp, deg, pos, starts, min_dEng):
    """Per-group merge of a site-step's candidates (tn_merge_groups): members sorted by group, `starts` the ngroups + 1 offsets.
    Returns (rep_pos, deg, log2p) per group, device tensors."""
    ng = starts.numel() - 1
    for t in (E, lp, deg, pos, starts):
        assert t.is_contiguous() and t.is_cuda
    rep = torch.empty(ng, dtype=torch.int64, device=E.device)
    dn = torch.empty(ng, dtype=torch.int64, device=E.device)
    ln = torch.empty(ng, dtype=torch.float64, device=E.device)
    check(lib().tn_merge_groups(E.data_ptr(), lp.data_ptr(), deg.data_ptr(), pos.data_ptr(), starts.data_ptr(), ng, float(min_dEng),
                                rep.data_ptr(), dn.data_ptr(), ln.data_ptr(), _stream()))
    return rep, dn, ln


def env_rr(A, RRprev, W, parent, uidx):
    """Right environments of all distinct suffixes of one site (tn_env_rr_batched).  A (Dl,p,Dr), RRprev (nprev,Dr,br),
    W (bl,p,br,pu), parent / uidx int32 device vectors (nk).  Returns (nk, Dl, bl), each block nfactor-normalised."""
    Dl, p, Dr = A.shape
    bl, p2, br, pu = W.shape
    assert p2 == p and RRprev.shape[1:] == (Dr, br), (A.shape, W.shape, RRprev.shape)
    for t in (A, RRprev, W, parent, uidx):
        assert t.is_contiguous() and t.is_cuda
    nk = parent.numel()
    out = torch.empty((nk, Dl, bl), dtype=torch.float64, device=A.device)
    check(lib().tn_env_rr_batched(A.data_ptr(), RRprev.data_ptr(), W.data_ptr(), parent.data_ptr(), uidx.data_ptr(), nk, Dl, p, Dr,
                                  bl, br, pu, out.data_ptr(), _stream()))
    return out


def env_rl(T1, par, didx):
    """Left environments of the new distinct prefixes: rows (par[k], didx[k]) of T1 (npref, p, Dr), nfactor-normalised."""
    _, p, Dr = T1.shape
    for t in (T1, par, didx):
        assert t.is_contiguous() and t.is_cuda
    nk = par.numel()
    out = torch.empty((nk, Dr), dtype=torch.float64, device=T1.device)
    check(lib().tn_env_rl_batched(T1.data_ptr(), par.data_ptr(), didx.data_ptr(), nk, p, Dr, out.data_ptr(), _stream()))
    return out


def balance(env, max_scale=0.0):
    """dgebal (job 'S') scaling of a small square device matrix, clamped to [1/max_scale, max_scale] (tn_balance).
    Returns the device vector of scale factors (powers of two)."""
    _need_gpu(env)
    n = env.shape[0]
    assert env.shape == (n, n)
    out = torch.empty(n, dtype=torch.float64, device=env.device)
    check(lib().tn_balance(env.data_ptr(), env.stride(0), env.stride(1), n, float(max_scale), out.data_ptr(), None, _stream()))
    return out


def peps_factor(Es, E1, E4, Xu, Xl, Xr, Xd, dmap, rmap):
    """F[s,l,u] on the device from the (beta-scaled, min-shifted) energy tables and gauge diagonals (tn_peps_factor)."""
    q, nl = E1.shape
    nu = E4.shape[1]
    for t in (Es, E1, E4, Xu, Xl, Xr, Xd, dmap, rmap):
        assert t.is_contiguous() and t.is_cuda
    F = torch.empty((q, nl, nu), dtype=torch.float64, device=Es.device)
    check(lib().tn_peps_factor(Es.data_ptr(), E1.data_ptr(), E4.data_ptr(), Xu.data_ptr(), Xl.data_ptr(), Xr.data_ptr(),
                               Xd.data_ptr(), dmap.data_ptr(), rmap.data_ptr(), q, nl, nu, F.data_ptr(), _stream()))
    return F


def mpo_from_factor(F, dmap, rmap, pd, br):
    """W[l,d,r,u] = sum over cell states of the PEPS factor (tn_mpo_from_factor)."""
    q, nl, nu = F.shape
    W = torch.empty((nl, pd, br, nu), dtype=torch.float64, device=F.device)
    check(lib().tn_mpo_from_factor(F.data_ptr(), dmap.data_ptr(), rmap.data_ptr(), q, nl, nu, pd, br, W.data_ptr(), _stream()))
    return W
