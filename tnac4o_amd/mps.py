"""Boundary-MPS algebra on MI355X — the call surface of the reference's ``tnac4o/mps.py`` that its solver
consumes, with every tensor resident in HBM and every contraction / factorisation running in
libtnpeps (hand-written HIP; include/tnpeps.h).  Method names, argument meaning and return values mirror the
reference (file:line cited per method) so the solver code reads the same; PyTorch only owns the memory.

Site tensors are contiguous float64 ``(Dl, p, Dr)`` CUDA tensors.  Host syncs happen only where the algorithm
needs a number on the host: the kept rank in ``truncateC``, the Schmidt values in ``update_S`` and the overlap.
"""
import math
import os
import weakref

import numpy as np
import torch

from . import ops

EPS = float(np.finfo(np.float64).eps)
# weighted rank-revealing first canonisation pass (MPS.canonise_right_weighted)
PASS1_ACCEPT = 2.0 ** -56        # accepted a-posteriori bound on the relative change of the state (the SVD's deflation level)
PASS1_FLOOR = 1e-14              # squared weights are floored at this fraction of the largest (what fp64 Gram sums resolve)
PASS1_MIN_BOND = 256             # bonds narrower than this are factored in full (nothing to gain)
LAZY_SCHMIDT = os.environ.get('TN_LAZY_SCHMIDT', '1') != '0'              # see _LazyS
BATCHED_SCHMIDT = os.environ.get('TN_BATCHED_SCHMIDT', '1') != '0'        # small centre matrices of a sweep in one launch
PASS1_STRUCTURED = os.environ.get('TN_PASS1_STRUCTURED', '1') != '0'     # Gram recursion through the MPS (x) MPO structure


def _attach_fused():
    """TN_ATTACH_FUSED=0: the attach of the weighted first pass multiplies with the absorbed tensor (chain.hip; read per call)."""
    return os.environ.get('TN_ATTACH_FUSED', '1') != '0'


def _var1_skip():
    """TN_VAR1_SKIP=0 runs the 4 chi stage's variational sweep even when the state is its own target (chain.hip; read per call)."""
    return os.environ.get('TN_VAR1_SKIP', '1') != '0'


def _var_target_phi():
    """TN_VAR_TARGET=phi keeps the first pass's tensors as the target of the variational sweeps (chain.hip; read per call)."""
    return os.environ.get('TN_VAR_TARGET', '')[:1] == 'p'



def _gram_step_structured(G, A, W, hconj):
    """One step of the left Gram recursion of the weighted first pass, G' = T^T (G (x) 1_t) T summed over the physical index,
    for an absorbed site T = A (x) W (tn_absorb) WITHOUT touching the absorbed tensor: with a = (alpha, l), b = (beta, r)

        T1[a, l', s', b'] = sum_alpha' G[a, (alpha', l')] A[alpha', s', beta']              (one GEMM, K = Dl)
        T2[a, t, r', b']  = sum_{l', s'} Wx[(t, r'), (l', s')] T1[a, (l', s'), b']           (batch over a)
        U[alpha, (s, r), (r', b')] = sum_{l, t} Wy[(s, r), (l, t)] T2[alpha, (l, t), (r', b')]   (batch over alpha)
        G'[beta, (r, r', beta')]   = sum_{alpha, s} A[(alpha, s), beta] U[(alpha, s), (r, r', beta')]   (one GEMM, K = Dl p)

    21 GFLOP per bulk site of the L = 2048, chi = 64 sweep instead of 69 for the two products with the absorbed tensor (the
    largest device-filling GEMMs of a sweep).  Every contraction is a plain or strided-batched tn_gemm on contiguous operands; only
    G (8 MB) and the MPO site are permuted.  A (Dl, s, Dr) MPS site, W (ba, po, bb, pi) MPO site; hconj as in tn_absorb: the MPS
    index is the major one of the fused bonds when hconj, the MPO index otherwise."""
    Dl, ps, Dr = A.shape
    ba, po, bb, pi = W.shape
    if hconj:                                  # contracted physical index s = po, new one t = pi
        Wl = W.permute(0, 1, 2, 3)             # (l, s, r, t)
        pt = pi
        assert ps == po
    else:                                      # s = pi, t = po
        Wl = W.permute(0, 3, 2, 1)             # (l, s, r, t)
        pt = po
        assert ps == pi
    na = Dl * ba
    assert G.shape == (na, na)
    G4 = G.view(Dl, ba, Dl, ba) if hconj else G.view(ba, Dl, ba, Dl).permute(1, 0, 3, 2)       # -> (alpha, l, alpha', l')
    Gp = G4.permute(0, 1, 3, 2).contiguous().view(na * ba, Dl)                                    # rows (alpha, l, l'), cols alpha'
    A = A.contiguous()
    T1 = ops.mm(Gp, A.view(Dl, ps * Dr))                                                          # (alpha, l, l', s', beta')
    Wx = Wl.permute(3, 2, 0, 1).contiguous().view(1, pt * bb, ba * ps)                            # [(t, r'), (l', s')]
    T2 = ops.bmm(Wx, T1.view(na, ba * ps, Dr))                                                    # (a, (t, r'), beta')
    Wy = Wl.permute(1, 2, 0, 3).contiguous().view(1, ps * bb, ba * pt)                            # [(s, r), (l, t)]
    U = ops.bmm(Wy, T2.view(Dl, ba * pt, bb * Dr))                                                # (alpha, (s, r), (r', beta'))
    O = ops.mm(A.view(Dl * ps, Dr).t(), U.view(Dl * ps, bb * bb * Dr))                            # (beta, (r, r', beta'))
    O4 = O.view(Dr, bb, bb, Dr)                                                                   # (beta, r, r', beta')
    out = O4.permute(0, 1, 3, 2) if hconj else O4.permute(1, 0, 2, 3)                             # (beta, r, beta', r') | (r, beta, r', beta')
    return out.contiguous().view(Dr * bb, Dr * bb)


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError('tnac4o_amd needs a ROCm GPU: the HIP library is the only backend (no CPU fallback)')
    return torch.device('cuda', torch.cuda.current_device())


def _t(x):
    """Accept numpy arrays for convenience; everything is float64 on the current GPU."""
    if isinstance(x, torch.Tensor):
        return x.to(device=_dev(), dtype=torch.float64)
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float64).to(_dev())


# ------------------------------------------------------------------------------------------ module functions
def nfactor(T):
    """Largest |entry| floored to a power of two, as a Python float (reference mps.py:76-85).  Syncs."""
    return float(ops.nfactor_dev(_t(T).contiguous())[0].item())


def qr(T):
    """Economic QR with diag(R) >= 0 (mps.py:43-59)."""
    return ops.qr(_t(T))


def svd(T):
    """Thin SVD with the reference's sign gauge (mps.py:24-40).  Returns U, S, V (V rows are right vectors).
    Vectors are returned for singular values above 2^-56 S0 (smaller ones are deflated by the Jacobi kernel)."""
    T = _t(T)
    U, S, V, _, _, _ = ops.svd_trunc(T, min(T.shape), 0.0)
    return U, S, V


def svd_S(T):
    """Singular values only (mps.py:62-73), numpy array on the host."""
    return ops.svdvals(_t(T))


def dot(phi, psi):
    """<phi|psi> (mps.py:88-93)."""
    RL = torch.ones((1, 1), dtype=torch.float64, device=_dev())
    for n in range(psi.L):
        RL = psi._mps_RL(RL, psi.A[n], phi.A[n])
    return float(RL.reshape(-1)[0].item())


class MPO:
    """MPO container (mps.py:818-865): ``W[n]`` has legs (left bond, out, right bond, in)."""

    def __init__(self, d=2, dout=None, L=2):
        self.L = L
        one = torch.ones((1, 1, 1, 1), dtype=torch.float64, device=_dev())
        self.W = [one] * L
        self.support = [0] * L

    def set_direct(self, W, n):
        """mps.py:859-865."""
        self.W[n] = _t(W).contiguous()
        self.support[n] = 1


class _Ident:
    """Identity of a tensor as recorded at some point: the object itself (weakly) and torch's in-place version counter.  matches(t) is
    true only for that very tensor, unmodified by any torch in-place operation since (kernels of libtnpeps that write a site in place
    go through MPS methods that drop the record)."""

    def __init__(self, t):
        self.ref = weakref.ref(t)
        self.version = t._version

    def matches(self, t):
        return self.ref() is t and t._version == self.version


class _LazyS:
    """Schmidt values that nothing has asked for yet: the centre matrix they would be computed from.  variational_compress
    only ever compares a bond's Schmidt values with the previous ones OF THE SAME LENGTH (mps.py:555-556 resets them otherwise),
    and the values taken during the last sweep a call is allowed to make cannot influence that call any more (the sweep limit
    returns the same overlap whatever the change was).  So in such a sweep update_S is recorded, not evaluated: the next
    update_S at that bond materialises the values if the bond still has the same dimension -- with the same kernel on the same
    data, i.e. bit-identical to the eager evaluation -- and drops them otherwise.  On the contraction path the graduated stage
    (one sweep at 4 chi) is followed by a stage at chi, so its 256 x 256 centre matrices are never decomposed (0.9 ms each)."""

    def __init__(self, Cm):
        self.Cm = Cm
        self.size = min(Cm.shape)

    def values(self):
        return ops.svdvals(self.Cm)


class _SchmidtList(list):
    """psi.S: reading an entry evaluates it if it is still a _LazyS (whoever looks at the attribute sees plain arrays, as in
    the reference); the sweeps themselves go through _previous_S, which drops a lazy entry of the wrong length unevaluated."""

    def __getitem__(self, i):
        v = list.__getitem__(self, i)
        if isinstance(v, _LazyS):
            v = v.values()
            list.__setitem__(self, i, v)
        elif isinstance(i, slice):
            v = [self[j] for j in range(*i.indices(len(self)))]
        return v

    def __iter__(self):
        return (self[i] for i in range(len(self)))


def _previous_S(psi, pC, size):
    """psi.S[pC] as update_S sees it before taking new values of length `size` (mps.py:555-556)."""
    old = list.__getitem__(psi.S, pC)
    if isinstance(old, _LazyS):
        old = old.values() if old.size == size else None
    if old is None or old.size != size:
        old = psi._one_S(size)
    return old


def _dist(a, b):
    """||a - b||_2 summed in index order (the C++ chain driver adds in the same order)."""
    acc = 0.0
    for x, y in zip(a, b):
        d = float(x) - float(y)
        acc += d * d
    return math.sqrt(acc)


class _DeferredSchmidt:
    """The update_S calls of one variational sweep (mps.py:550-560), taken asynchronously and replayed in order at the end.

    add(): the current centre matrix psi.C at bond psi.pC.  Up to 64 x 64 it is only kept: finish() decomposes all of them
    in ONE launch, one workgroup per matrix (tn_svdvals_small_batched; a launch per site cost the chain 130 us each --
    TN_BATCHED_SCHMIDT=0 restores that form, tn_svdvals_async into a device table); larger centre matrices go through the
    synchronous block-Jacobi path at once.  finish(): that launch, one device-to-host copy, then exactly the bookkeeping of
    update_S for every site in the order the reference performs it; returns max dS over the measured (left-to-right) half.
    lazy: see _LazyS."""

    def __init__(self, psi, lazy=False):
        self.psi = psi
        self.lazy = lazy                # last sweep the caller may make: record the centre matrices only (see _LazyS)
        self.items = []                 # (pC, measure, row index | None, host array | None, centre tensor kept alive)
        self.small = []                 # centre matrices up to 64 x 64 of this sweep, decomposed together in finish()
        self.table = None
        self.side = None
        cur = torch.cuda.current_stream()
        if cur.cuda_stream != 0 and ops.SCHMIDT_SIDE:   # never make the legacy default stream wait on / for others
            self.side = ops.side_stream()
        self.cur = cur

    def add(self, measure):
        psi = self.psi
        Cm = psi.C
        if self.lazy:
            psi.S[psi.pC] = _LazyS(Cm)
            return
        if max(Cm.shape) <= 64:
            if BATCHED_SCHMIDT:                            # kept until finish(): one launch for the whole sweep
                self.items.append((psi.pC, measure, len(self.small), None, Cm))
                self.small.append(Cm)
                return
            if self.table is None:
                self.table = torch.zeros((2 * psi.L + 2, 66), dtype=torch.float64, device=Cm.device)
            row = sum(1 for it in self.items if it[2] is not None)
            if self.side is not None:
                self.side.wait_stream(self.cur)
            ops.svdvals_async(Cm, self.table[row], stream=self.side)
            self.items.append((psi.pC, measure, row, None, Cm))
        else:
            self.items.append((psi.pC, measure, None, ops.svdvals(Cm), None))

    def finish(self):
        psi = self.psi
        host = None
        if self.small:
            host = ops.svdvals_small_batched(self.small).cpu().numpy()
            self.small = []
        elif self.table is not None:
            if self.side is not None:
                self.cur.wait_stream(self.side)
            host = self.table.cpu().numpy()
        diff = 0.0
        for pC, measure, row, S, Cm in self.items:
            if row is not None:
                k = min(Cm.shape)
                if host[row, 65] == 0.0 or not np.all(np.isfinite(host[row, :k])):
                    S = ops.svdvals(Cm)          # not converged in the fused kernel: the full path (raises if that fails too)
                else:
                    S = host[row, :k].copy()
            dS = _dist(_previous_S(psi, pC, S.size), S)
            psi.S[pC] = S
            if measure:
                diff = max(diff, dS)
        self.items = []
        return diff


class MPS:
    """Boundary MPS with an explicit orthogonality centre C at bond pC (mps.py:96-173).

    Only the maximally mixed start state ``initial='X'`` is provided — the one the solver uses (tnac4o.py:1682).
    """

    def __init__(self, d=2, L=2, Dmax=2, initial='X', canonise='left'):
        if initial != 'X':
            raise NotImplementedError("only initial='X' is on the contraction path")
        d = [d] if isinstance(d, int) else list(d)
        d = (d * ((L + len(d) - 1) // len(d)))[:L]
        self.L, self.d = L, d
        self.zero = EPS
        self.dtype = 'float64'
        dev = _dev()
        self.D = self._Dset(Dmax, d)
        self.A = []
        for n in range(L):                                                   # mps.py:635-638
            A = torch.zeros((self.D[n], d[n], self.D[n + 1]), dtype=torch.float64, device=dev)
            A[0, :, 0] = 1.0 / np.sqrt(d[n])
            self.A.append(A)
        self.C = torch.ones((1, 1), dtype=torch.float64, device=dev)
        self.pC = L
        self._nfs = []                   # device pairs [nf, 1/nf]; normC is their product (kept lazily)
        self.reset_R()
        self.reset_S()
        self.discarded = [0] * (L + 1)
        if canonise == 'left':
            self.canonise_left()
        elif canonise == 'right':
            self.canonise_right()
        self._nfs = []

    # -- bookkeeping --------------------------------------------------------------------------------------
    @property
    def normC(self):
        """Accumulated power-of-two norm factor (mps.py:122, 537, 546).  Syncs; not used on the hot path."""
        out = 1.0
        for f in self._nfs:
            out *= float(f[0].item())
        return out

    @staticmethod
    def _Dset(Dmax, d):
        """mps.py:644-653."""
        L = len(d)
        D = [1] * (L + 1)
        for n in range(L):
            D[n + 1] = min(D[n] * d[n], Dmax)
        D[-1] = 1
        for n in range(L - 1, -1, -1):
            D[n] = min(D[n + 1] * d[n], Dmax, D[n])
        return D

    @staticmethod
    def _one_S(D):
        S = np.zeros(D)
        S[0] = 1.0
        return S

    def reset_R(self):
        """mps.py:281-286."""
        one = torch.ones((1, 1), dtype=torch.float64, device=_dev())
        self.R = [one.clone() for _ in range(self.L + 2)]
        self.R[-1] = None

    def reset_S(self):
        """mps.py:295-299."""
        self.S = _SchmidtList(self._one_S(self.D[n]) for n in range(self.L + 1))

    def copy(self):
        """Deep copy of the tensors (mps.py:159-173); S and ``discarded`` start fresh as in the reference."""
        o = MPS(d=self.d, L=self.L, Dmax=1, initial='X', canonise=None)
        o.A = [a.clone() for a in self.A]
        o.C = self.C.clone()
        o.pC = self.pC
        o._nfs = list(self._nfs)
        o.D = self.D[:]
        o.R = self.R[:]
        return o

    # -- absorption ---------------------------------------------------------------------------------------
    def apply_mpo(self, M, Hconj=False):
        """psi <- H psi (or H^dag psi), site by site (mps.py:353-359 -> :753-763): K1 tn_absorb."""
        # the factors of every absorbed site are kept next to the product (both small): the weighted first pass of compress_mps
        # pushes its Gram matrices through the product structure instead of through the 134 MB absorbed tensor
        self._absorbed = {}
        for n in range(self.L):
            if M.support[n]:
                old = self.A[n]
                self.A[n] = ops.absorb(old, M.W[n], Hconj)
                self.D[n], self.d[n], self.D[n + 1] = self.A[n].shape
                # (a weak reference and the version counter of the product: identity, not an address that could be reused)
                self._absorbed[n] = (old, M.W[n], bool(Hconj), _Ident(self.A[n]))

    def apply_diagonalO(self, diagO, n):
        """mps.py:361-366."""
        self.scale_site_(n, _t(diagO).contiguous())

    def scale_site_(self, n, diag, inv=False):
        """A[n] scaled (or divided) along its physical leg IN PLACE by a device vector; any recorded absorption is stale afterwards."""
        self._absorbed = None
        ops.scale_phys_(self.A[n], diag, inv=inv)

    # -- gauge moves --------------------------------------------------------------------------------------
    def attach_AC(self):
        """A[pC-1] <- A[pC-1] . C (mps.py:368-373)."""
        n = self.pC - 1
        Dl, p, Dr = self.A[n].shape
        self.A[n] = ops.mm(self.A[n].view(Dl * p, Dr), self.C).view(Dl, p, self.C.shape[1])

    def attach_CA(self):
        """A[pC] <- C . A[pC] (mps.py:375-380)."""
        n = self.pC
        Dl, p, Dr = self.A[n].shape
        self.A[n] = ops.mm(self.C, self.A[n].view(Dl, p * Dr)).view(self.C.shape[0], p, Dr)

    def orth_left(self, n, rank_tol=0.0):
        """A[n] -> Q, C = R / nfactor(R) with QR of the (Dl p, Dr) matrix (mps.py:532-539, 772-785).
        rank_tol > 0 (truncating passes only): tn_qr stops once the rest of R is below rank_tol of its scale, so the new
        bond may be smaller than min(Dl p, Dr) — the rows dropped are the ones truncateC's SVD deflates anyway."""
        Dl, p, Dr = self.A[n].shape
        kf = min(Dl * p, Dr)
        Q = torch.empty((Dl * p, kf), dtype=torch.float64, device=self.A[n].device)
        R = torch.empty((kf, Dr), dtype=torch.float64, device=self.A[n].device)
        _, _, k = ops.qr_into(self.A[n].view(Dl * p, Dr), Q, R, overwrite=True, rank_tol=rank_tol)
        if k < kf:
            Q, R = Q[:, :k].contiguous(), R[:k].contiguous()
        self._nfs.append(ops.normalize_pow2_(R))
        if R.shape == (1, 1):            # mps.py:778-780 (diag(R) >= 0, so sign(C) = 1): the norm is dropped
            R = torch.ones_like(R)
        self.A[n] = Q.view(Dl, p, k)
        self.C = R
        self.D[n + 1] = k
        self.pC = n + 1

    def orth_right(self, n, rank_tol=0.0):
        """A[n] -> C Q with QR of the transposed (p Dr, Dl) view (mps.py:541-548, 787-800); rank_tol as in orth_left."""
        Dl, p, Dr = self.A[n].shape
        kf = min(p * Dr, Dl)
        dev = self.A[n].device
        Qt = torch.empty((kf, p * Dr), dtype=torch.float64, device=dev)     # Q^T, i.e. the new right-canonical site
        Ct = torch.empty((Dl, kf), dtype=torch.float64, device=dev)         # R^T
        _, _, k = ops.qr_into(self.A[n].view(Dl, p * Dr).t(), Qt.t(), Ct.t(), overwrite=True, rank_tol=rank_tol)
        if k < kf:
            Qt, Ct = Qt[:k].contiguous(), Ct[:, :k].contiguous()
        self._nfs.append(ops.normalize_pow2_(Ct))
        if Ct.shape == (1, 1):
            Ct = torch.ones_like(Ct)
        self.A[n] = Qt.view(k, p, Dr)
        self.C = Ct
        self.D[n] = k
        self.pC = n

    def _site_left(self, n, Cm, rank_tol=0.0):
        """attach_CA (Cm given) + orth_left at site n in one library call (tn_site_qr); same state changes."""
        Q, R, k, nf = ops.site_qr(0, self.A[n], Cm, rank_tol)
        self._nfs.append(nf)
        if R.shape == (1, 1):
            R = torch.ones_like(R)
        self.A[n] = Q.view(-1, self.A[n].shape[1], k)
        self.C = R
        self.D[n], self.D[n + 1] = self.A[n].shape[0], k
        self.pC = n + 1

    def _site_right(self, n, Cm, rank_tol=0.0):
        """attach_AC (Cm given) + orth_right at site n in one library call (tn_site_qr); same state changes."""
        Qt, Ct, k, nf = ops.site_qr(1, self.A[n], Cm, rank_tol)
        self._nfs.append(nf)
        if Ct.shape == (1, 1):
            self._c11 = Ct                       # the normalised 1 x 1 centre (its value enters the state norm of the pass)
            Ct = torch.ones_like(Ct)
        self.A[n] = Qt.view(k, self.A[n].shape[1], -1)
        self.C = Ct
        self.D[n], self.D[n + 1] = k, self.A[n].shape[2]
        self.pC = n

    def truncateC(self, Dmax, tol=None):
        """SVD-truncate the centre matrix and push the projectors into the neighbours (mps.py:562-585, 802-811)."""
        if 0 < self.pC < self.L:
            if tol is None:
                tol = self.zero
            Dcap = int(min(Dmax, min(self.C.shape)))
            if (getattr(self, '_intermediate_pass', False) and ops.gauge_svd_mode() != 1 and not (ops.gauge_svd_mode() == 2 and self._pass_side == 1) and tol <= np.finfo(float).eps
                    and min(self.C.shape) <= Dmax):
                # a truncation that cannot truncate (chain.hip: gauge_svd_skippable): it would only remove singular values below
                # eps S0 and turn the bond into the Schmidt basis, neither of which is visible outside an intermediate pass --
                # the centre matrix stays with the next site, less the bond indices that carry nothing (tn_bond_deflate)
                if ops.bond_deflate_on():
                    side = self._pass_side
                    ns = self.pC - 1 if side == 0 else self.pC
                    self.C, self.A[ns], kk, d2 = ops.bond_deflate(side, self.C, self.A[ns])
                    self.D[self.pC] = kk
                    self.discarded[self.pC] = max(self.discarded[self.pC], float(np.sqrt(d2)))
                return 0.0
            U, S, Vt, keep, disc, _ = ops.svd_trunc(self.C, Dcap, tol)
            if disc > 32.0 * np.finfo(float).eps:
                self._pass_truncated = True          # this pass has changed the state by more than rounding (chain.hip: pass_truncated)
            nl, nr = self.pC - 1, self.pC
            if ops.FUSED_SITE and keep > 0:
                self.A[nl], self.A[nr], self.C = ops.apply_truncation(self.A[nl], U, S, Vt, self.A[nr])
            else:
                Dl, p, _ = self.A[nl].shape
                self.A[nl] = ops.mm(self.A[nl].view(Dl * p, -1), U).view(Dl, p, keep)
                _, p2, Dr = self.A[nr].shape
                self.A[nr] = ops.mm(Vt, self.A[nr].view(-1, p2 * Dr)).view(keep, p2, Dr)
                self.C = torch.diag(S)
            self.D[self.pC] = keep
            self.discarded[self.pC] = max(self.discarded[self.pC], disc)
            return disc
        return 0.0

    def canonise_left(self, compress=False, Dmax=np.inf, tol=None):
        """mps.py:202-218."""
        self.C = torch.ones((1, 1), dtype=torch.float64, device=self.A[0].device)
        self.pC = 0
        self._pass_side = 0
        self._pass_C = [None] * (self.L + 1)
        for n in range(self.L):
            # truncating pass: left part canonical, right part canonical -> the scale of C is the Schmidt scale, so rows of
            # R below 2^-56 of it can be skipped already in the QR (they are deflated by the SVD of truncateC)
            rank_tol = ops.RANK_TOL if (compress and 0 < n + 1 < self.L) else 0.0
            if ops.FUSED_SITE:
                self._site_left(n, self.C, rank_tol)
            else:
                self.attach_CA()
                self.orth_left(n, rank_tol=rank_tol)
            if compress:
                self.truncateC(Dmax, tol)
                if getattr(self, '_intermediate_pass', False):
                    self._pass_C[self.pC] = self.C           # the centre matrix this pass leaves at the bond (chain.hip: pass_C)
        self.R[-1] = None

    def canonise_right(self, compress=False, Dmax=np.inf, tol=None):
        """mps.py:220-236."""
        self.C = torch.ones((1, 1), dtype=torch.float64, device=self.A[0].device)
        self.pC = self.L
        self._pass_side = 1
        for n in range(self.L - 1, -1, -1):
            rank_tol = ops.RANK_TOL if (compress and 0 < n < self.L) else 0.0
            if ops.FUSED_SITE:
                self._site_right(n, self.C, rank_tol)
            else:
                self.attach_AC()
                self.orth_right(n, rank_tol=rank_tol)
            if compress:
                self.truncateC(Dmax, tol)
        self.R[-1] = None

    def canonise_right_weighted(self):
        """The first (non-truncating) pass of compress_mps -- `canonise_right()` at mps.py:187 -- with a rank-revealing
        early exit made rigorous by a diagonal gauge.  Returns True when the error bound below holds (the caller falls back
        to the plain pass otherwise; it keeps the input tensors).

        The absorbed MPS has bonds of 1024 with a Schmidt rank of ~100 at 2^-56, but while sweeping right to left the part
        LEFT of the site is not canonical, so a column of the site matrix that looks negligible may still carry weight
        (dropping by plain norms moved log2 P by 3e-7 in round 1).  The weight of left-bond index c is the norm of the
        left part restricted to it, d_c = sqrt(G_L[c,c]), with G_L the Gram matrix of the left part -- obtained for every
        bond by two GEMMs per site from the absorbed tensors (left to right, before the pass; carried through the last
        site it also gives the norm of the whole state).  At site n the matrix M_n (left bond x (phys, right bond)) is
        scaled row-wise by d and its rows are sorted by decreasing weighted norm; tn_qr factors that matrix and stops once
        the Frobenius norm of the trailing block is below the threshold worked out below.  In the scaled gauge the left
        part L' = L diag(1/d) has unit columns, so dropping a block E changes the state by at most ||L'||_2 ||E||_F with
        ||L'||_2^2 = lambda_max(K) <= ||K||_F, K = G_L / (d d^T) (tn_gram_weights).  The norm N_n of the state in the units
        of step n is ||psi|| divided by the power-of-two factors of the sites already done, so the threshold of site n is
        set to  (2^-57 / #sites) N_n / (sqrt(g_n) ||K_n||_F^(1/2))  (g_n: scale removed from G_L(n)), i.e. every site may
        change the state by 2^-57 / #sites of its norm, 2^-57 in total -- below the level (2^-56) at which the Jacobi SVD of
        the next pass deflates.  The sum of the actual bounds is re-checked from the dropped norms tn_qr reports.
        Host synchronisations: one before the pass, one small read-back per weighted site."""
        L = self.L
        T = list(self.A)
        self._D_in = list(self.D)
        dev = T[0].device
        # Gram matrices of the left part, bond by bond (G_L(n) belongs to the left bond of site n; G_L(L) = ||psi||^2)
        G = torch.ones((1, 1), dtype=torch.float64, device=dev)
        weights = [None] * (L + 1)
        gfac = [None] * (L + 1)                            # device [nf, 1/nf] removed from G_L when it was normalised
        absorbed = getattr(self, '_absorbed', None) or {}
        self._absorbed = None                              # used once: the factors must not outlive the pass
        for n in range(L):
            Dl, p, Dr = T[n].shape
            fac = absorbed.get(n)
            if PASS1_STRUCTURED and fac is not None and fac[3].matches(T[n]) and Dl >= PASS1_MIN_BOND and G.shape[0] == Dl:
                G = _gram_step_structured(G, fac[0], fac[1], fac[2])
            else:
                X = ops.mm(G, T[n].view(Dl, p * Dr))
                G = ops.mm(T[n].view(Dl * p, Dr).t(), X.view(Dl * p, Dr))
            gfac[n + 1] = ops.normalize_pow2_(G)
            if n + 1 < L and Dr >= PASS1_MIN_BOND:
                weights[n + 1] = ops.gram_weights(G, PASS1_FLOOR)
        used = [n for n in range(L) if weights[n] is not None]
        if not used:
            self.canonise_right()
            self.reveal_error_bound = 0.0
            return True
        pack = torch.cat([gfac[m][:1] for m in range(1, L + 1)] + [G.reshape(1)] + [weights[n][1][:64] for n in used]).cpu().numpy()
        # (host arithmetic through libm's log2 / pow / sqrt in a fixed order: tn_compress_mps reproduces it bit for bit)
        logg = [0.0]
        for m in range(L):
            logg.append(logg[-1] + math.log2(float(pack[m])))                      # log2 of the scale removed up to bond n
        log_psi = 0.5 * (logg[L] + math.log2(float(pack[L])))                      # log2 ||psi||
        kparts = pack[L + 1:].reshape(len(used), 64)
        bound = {}
        for i, n in enumerate(used):                                               # ||L'||_2 <= ||K||_F^(1/2)
            acc = 0.0
            for x in kparts[i]:
                acc += float(x)
            bound[n] = math.pow(acc, 0.25)
        budget = 2.0 ** -57 / len(used)
        self.C = torch.ones((1, 1), dtype=torch.float64, device=dev)
        self.pC = L
        lognf_done, pending, err = 0.0, [], 0.0
        for n in range(L - 1, -1, -1):
            if weights[n] is None:
                self._site_right(n, self.C, 0.0)
                pending.append(self._nfs[-1])
                continue
            d2, _ = weights[n]
            Dl, p, Dr = self.A[n].shape
            r = self.C.shape[1]
            fac = absorbed.get(n)
            if (_attach_fused() and PASS1_STRUCTURED and fac is not None and fac[3].matches(T[n]) and max(self._D_in) >= 2 * PASS1_MIN_BOND):
                # the attach through the factors of the absorbed site (chain.hip: attach_through_factors): the native driver never forms
                # the absorbed tensor of such a site; here it exists (apply_mpo made it) but the product takes the same steps
                Af, Wf = fac[0].contiguous(), fac[1].contiguous()
                Dl0, ps, Dr0 = Af.shape
                ba, po, bb, pi = Wf.shape
                if fac[2]:                                                                       # Hconj: MPS-major fused bonds
                    Tm = ops.mm(Af.view(Dl0 * ps, Dr0), self.C.contiguous().view(Dr0, bb * r))                   # (alpha, s, rb, r')
                    Wq = Wf.permute(0, 3, 1, 2).contiguous().view(1, ba * pi, ps * bb)           # [(l, t), (s, rb)]
                    A = ops.bmm(Wq, Tm.view(Dl0, ps * bb, r)).view(Dl, p * r)
                else:                                                                            # MPO-major: one product per rb, result moved to (l, alpha)
                    Tm = torch.empty((Dl0 * ps, bb, r), dtype=torch.float64, device=dev)          # (alpha, s, rb, r')
                    ops.bmm(Af.view(1, Dl0 * ps, Dr0), self.C.contiguous().view(bb, Dr0, r), out=Tm.permute(1, 0, 2))
                    Wq = Wf.permute(0, 1, 3, 2).contiguous().view(1, ba * po, ps * bb)           # [(l, t), (s, rb)]
                    Mt = ops.bmm(Wq, Tm.view(Dl0, ps * bb, r))                                   # (alpha, (l, t), r')
                    A = Mt.view(Dl0, ba, po * r).permute(1, 0, 2).contiguous().view(Dl, p * r)
            else:
                A = ops.mm(self.A[n].view(Dl * p, Dr), self.C).view(Dl, p * r)      # M_n (attach_AC)
            w, wsum = ops.weighted_sum(d2, ops.rows_norm2(A))
            host = torch.cat([wsum] + [f[:1] for f in pending]).cpu().numpy()       # the per-site read-back
            for x in host[1:]:
                lognf_done += math.log2(float(x))
            pending = []
            logN = log_psi - lognf_done                                             # log2 of the state norm in this step's units
            scale = math.pow(2.0, 0.5 * logg[n] - logN) * bound[n]                  # relative change of the state per unit ||E||_F
            fro = math.sqrt(float(host[0]))
            rel_tol = min(2.0 ** -40, max(1e-30, budget / (scale * fro))) if fro > 0.0 else 0.0
            perm = ops.argsort_desc(w)                                              # 1024 keys: index plumbing
            B = ops.gather_scale_rows(A, perm, d2)
            info = {}
            Qt, Ctp, k, _ = ops.site_qr(1, B.view(Dl, p, r), None, rel_tol, normalise=False, info=info, frobenius_exit=True,
                                        pivot=ops.PASS1_PIVOT)
            if ops.PASS1_PIVOT:
                perm = perm[info['perm']]                                           # sort order followed by the panel pivoting
            Ct = ops.gather_scale_rows(Ctp, perm, d2, inverse=True)                 # rows back in place, weights removed
            self._nfs.append(ops.normalize_pow2_(Ct))
            pending.append(self._nfs[-1])
            self.A[n] = Qt.view(k, p, r)
            self.C = Ct
            self.D[n], self.D[n + 1] = k, r
            self.pC = n
            err += scale * math.sqrt(info['dropped2'])
        self.R[-1] = None
        self.reveal_error_bound = float(err)
        if ops.PASS1_TRACE:
            import sys
            print('[pass1] D=%s err_bound=%.3e accept=%s' % (self.D, err, err <= PASS1_ACCEPT), file=sys.stderr, flush=True)
        return bool(err <= PASS1_ACCEPT)

    # -- environments -------------------------------------------------------------------------------------
    @staticmethod
    def _mps_RL(RL, A, Ac):
        """out[c',a'] = sum_{c,s,a} Ac[c,s,c'] RL[c,a] A[a,s,a'] (mps.py:655-658)."""
        if ops.FUSED_SITE:
            return ops.env_mix(0, RL, A, Ac)
        a, s, a2 = A.shape
        c, _, c2 = Ac.shape
        T = ops.mm(RL, A.view(a, s * a2))                       # (c, s a')
        return ops.mm(Ac.view(c * s, c2).t(), T.view(c * s, a2))

    @staticmethod
    def _mps_RR(RR, A, Ac):
        """out[a,c] = sum A[a,s,a'] RR[a',c'] Ac[c,s,c'] (mps.py:660-663)."""
        if ops.FUSED_SITE:
            return ops.env_mix(1, RR, A, Ac)
        a, s, a2 = A.shape
        c, _, c2 = Ac.shape
        T = ops.mm(A.view(a * s, a2), RR)                       # (a s, c')
        return ops.mm(T.view(a, s * c2), Ac.view(c, s * c2).t())

    @staticmethod
    def _mps_RAR(RL, A, RR):
        """RL . A . RR (mps.py:748-751)."""
        if ops.FUSED_SITE:
            return ops.rar(RL, A, RR)
        a, s, a2 = A.shape
        T = ops.mm(RL, A.view(a, s * a2))                       # (c, s a')
        c = RL.shape[0]
        return ops.mm(T.view(c * s, a2), RR).view(c, s, RR.shape[1])

    def update_RL_mix(self, phi, n, keep_on_device=False):
        """mps.py:436-444.  At the last site the overlap goes to R[-1]: as a host float (one synchronisation), or with
        keep_on_device as the 1 x 1 device tensor (the preconditioner walks over the last site without needing the number)."""
        new = self._mps_RL(self.R[n], phi.A[n], self.A[n])
        if n == self.L - 1:
            self.R[self.L + 1] = new if keep_on_device else float(new.reshape(-1)[0].item())
        else:
            self.R[n + 1] = new

    def update_RR_mix(self, phi, n):
        """mps.py:418-426."""
        new = self._mps_RR(self.R[n + 1], phi.A[n], self.A[n])
        if n == 0:
            self.R[self.L + 1] = float(new.reshape(-1)[0].item())
        else:
            self.R[n] = new

    def setup_RL_mix(self, phi):
        """mps.py:446-452."""
        for n in range(self.L):
            self.update_RL_mix(phi, n)
        return self.R[-1]

    def setup_RR_mix(self, phi):
        """mps.py:428-434."""
        for n in range(self.L - 1, -1, -1):
            self.update_RR_mix(phi, n)
        return self.R[-1]

    def bond_env_mix(self, phi, n):
        """p x p environment of the physical leg of site n in <self|phi> (mps.py:454-458, 765-769)."""
        T2 = self._mps_RAR(self.R[n], phi.A[n], self.R[n + 1])          # (c, s, c')
        c, s, c2 = T2.shape
        Ac = self.A[n]
        # env[s, s'] = sum_{c,c'} T2[c,s,c'] Ac[c,s',c']
        return ops.mm(T2.permute(1, 0, 2).reshape(s, c * c2), Ac.permute(1, 0, 2).reshape(Ac.shape[1], c * c2).t())

    def expectation_mix_dev(self, phi, n):
        """<self| ... |phi> at site n given both environments, as a (1, 1) device tensor (no synchronisation)."""
        T2 = self._mps_RAR(self.R[n], phi.A[n], self.R[n + 1])
        return ops.mm(T2.reshape(1, -1), self.A[n].reshape(-1, 1))

    def expectation_mix(self, phi, n):
        """<self| ... |phi> at site n given both environments (mps.py:587-591, 694-698).  Syncs."""
        return float(self.expectation_mix_dev(phi, n).item())

    # -- variational compression --------------------------------------------------------------------------
    def optimise_site(self, phi, n):
        """mps.py:617-621."""
        self.A[n] = self._mps_RAR(self.R[n], phi.A[n], self.R[n + 1])

    def update_S(self):
        """Schmidt values of the centre matrix; returns ||S_old - S_new||_2 (mps.py:550-560)."""
        S = ops.svdvals(self.C)
        dS = _dist(_previous_S(self, self.pC, S.size), S)
        self.S[self.pC] = S
        return dS

    def variational_compress(self, phi, tol=None, max_sweeps=1, verbose=False):
        """mps.py:238-279.  The Schmidt spectra that update_S takes at every site (mps.py:550-560) are only consumed at the
        end of a sweep (convergence test, mps.py:255/270), so they are taken without synchronising -- tn_svdvals_async on a
        side stream for centre matrices up to 64 x 64 -- and folded in once per sweep (_DeferredSchmidt)."""
        if tol is None:
            tol = self.zero
        overlap = self.setup_RL_mix(phi)
        sweeps, diff = 0, 1.0
        while diff > tol:
            if sweeps >= max_sweeps:
                return overlap
            pend = _DeferredSchmidt(self, lazy=LAZY_SCHMIDT and sweeps + 1 >= max_sweeps)
            for n in range(self.L - 1, 0, -1):
                self.optimise_site(phi, n)
                if ops.FUSED_SITE:
                    self._site_right(n, None)
                else:
                    self.orth_right(n)
                pend.add(measure=False)
                self.update_RR_mix(phi, n)
            for n in range(self.L):
                self.optimise_site(phi, n)
                if ops.FUSED_SITE:
                    self._site_left(n, None)
                else:
                    self.orth_left(n)
                pend.add(measure=True)
                self.update_RL_mix(phi, n)
            diff = pend.finish()
            overlap = self.R[-1]
            sweeps += 1
        return overlap

    def _compress_native(self, mpo, Hconj, Dmax, tolS, tolV, max_sweeps, graduate_truncation):
        """apply_mpo (when mpo is given) + compress_mps through tn_compress_mps: the whole chain of this row in one library call,
        walked in C++ without the interpreter (and without the GIL: with the 4 lattice rotations on 4 host threads the Python
        between the ~600 steps of a row sets the pace).  Same kernels, same order: bit-identical to the Python driver."""
        W = None
        A = [a if a.is_contiguous() else a.contiguous() for a in self.A]
        if mpo is not None:
            W = [mpo.W[n] if mpo.support[n] else None for n in range(self.L)]
        else:
            # an absorption recorded by apply_mpo whose products are still the current sites: hand the factors over instead (the
            # driver absorbs them again, 16 launches, and pushes the Gram recursion of the weighted pass through them exactly as the
            # Python driver does)
            fac = getattr(self, '_absorbed', None) or {}
            if fac and all(f[3].matches(self.A[n]) for n, f in fac.items()) and len({f[2] for f in fac.values()}) == 1:
                A = [fac[n][0].contiguous() if n in fac else A[n] for n in range(self.L)]
                W = [fac[n][1].contiguous() if n in fac else None for n in range(self.L)]
                Hconj = next(iter(fac.values()))[2]
        res = ops.compress_mps_native(A, W, Hconj, int(Dmax), tolS if tolS is not None else self.zero, tolV if tolV is not None else self.zero,
                                      max_sweeps, graduate_truncation, weighted=ops.PASS1_WEIGHTED, structured=PASS1_STRUCTURED, lazy=LAZY_SCHMIDT)
        self.A = res['A']
        self.d = [int(a.shape[1]) for a in self.A]
        self.D = [int(self.A[0].shape[0])] + [int(a.shape[2]) for a in self.A]
        self.C = torch.ones((1, 1), dtype=torch.float64, device=self.A[0].device)
        self.pC = self.L
        self._nfs = list(self._nfs) + res['nfs']
        self._absorbed = None
        self.discarded = res['discarded']
        self.reset_R()
        self.R[-1] = res['overlap']
        self.S = _SchmidtList(res['S'][n] if res['S'][n] is not None else self._one_S(self.D[n]) for n in range(self.L + 1))
        if res['info']['weighted_used']:
            self.reveal_error_bound = res['info']['reveal_error_bound']
            if res['info']['reveal_fallbacks']:
                self.reveal_fallbacks = getattr(self, 'reveal_fallbacks', 0) + res['info']['reveal_fallbacks']
        self.native_info = res['info']
        return res['overlap']

    def apply_mpo_compress(self, M, Hconj=False, Dmax=np.inf, tolS=None, tolV=None, max_sweeps=4, graduate_truncation=True):
        """apply_mpo(M, Hconj) followed by compress_mps(...) -- the row step of the sweeps (tnac4o.py:1689-1693) -- as ONE library
        call when the native chain driver is enabled, the two separate steps otherwise.  Returns the overlap."""
        if ops.NATIVE_CHAIN and ops.FUSED_SITE and np.isfinite(Dmax):
            return self._compress_native(M, Hconj, Dmax, tolS, tolV, max_sweeps, graduate_truncation)
        self.apply_mpo(M, Hconj=Hconj)
        return self._compress_python(Dmax=Dmax, tolS=tolS, tolV=tolV, max_sweeps=max_sweeps, graduate_truncation=graduate_truncation)

    def compress_mps(self, Dmax=np.inf, tolS=None, tolV=None, max_sweeps=4, graduate_truncation=True, verbose=False):
        """Truncate: SVD initialisation + variational sweeps (mps.py:175-200).  Returns the overlap <psi|phi>.  Runs in the C++ chain
        driver (tn_compress_mps) unless TN_NATIVE_CHAIN=0 selects the Python driver below (same kernels, same order)."""
        if ops.NATIVE_CHAIN and ops.FUSED_SITE and np.isfinite(Dmax) and tolS is not None:
            return self._compress_native(None, False, Dmax, tolS, tolV, max_sweeps, graduate_truncation)
        return self._compress_python(Dmax=Dmax, tolS=tolS, tolV=tolV, max_sweeps=max_sweeps, graduate_truncation=graduate_truncation)

    def _compress_python(self, Dmax=np.inf, tolS=None, tolV=None, max_sweeps=4, graduate_truncation=True):
        """compress_mps step by step from Python (the first-generation driver; kept as the cross-check of tn_compress_mps)."""
        if ops.PASS1_WEIGHTED and ops.FUSED_SITE and max(self.D) >= 2 * PASS1_MIN_BOND:
            keep = (list(self.A), list(self.D), list(self._nfs))
            if not self.canonise_right_weighted():                   # bound not met: the plain pass on the kept input
                self.A, self.D, self._nfs = list(keep[0]), list(keep[1]), list(keep[2])
                self.reveal_fallbacks = getattr(self, 'reveal_fallbacks', 0) + 1
                self.canonise_right()
        else:
            self.canonise_right()
        self._absorbed = None
        phi = self.copy()
        self.discarded = [0] * (self.L + 1)
        if graduate_truncation:
            self._intermediate_pass = True
            try:
                self._pass_truncated = False
                self.canonise_left(compress=True, Dmax=Dmax * 4, tol=tolS / 10)
                # a 4 chi pass that truncated nothing but rounding noise leaves phi itself, with smaller bonds: the target of the variational
                # sweeps from here on (chain.hip, tn_compress_mps)
                swapped = not self._pass_truncated and not _var_target_phi() and tolS / 10 <= np.finfo(float).eps
                if swapped:
                    phi = self.copy()
                if swapped and LAZY_SCHMIDT and _var1_skip():
                    # ... and the one sweep of this stage has nothing to do (the state IS its target): only the Schmidt values it would
                    # leave for the final stage are recorded, unevaluated, from the centre matrices of the 4 chi pass
                    for b in range(1, self.L + 1):
                        if self._pass_C[b] is not None:
                            list.__setitem__(self.S, b, _LazyS(self._pass_C[b]))
                else:
                    self.variational_compress(phi, tol=tolV, max_sweeps=1)
                self._pass_C = None
                self.canonise_right(compress=True, Dmax=Dmax * 2, tol=tolS / 2)
            finally:
                self._intermediate_pass = False
        self.canonise_left(compress=True, Dmax=Dmax, tol=tolS)
        return self.variational_compress(phi, tol=tolV, max_sweeps=max_sweeps)
