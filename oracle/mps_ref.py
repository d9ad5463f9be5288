"""Oracle: boundary-MPS algebra on the CPU (numpy + LAPACK through scipy).

TEST INFRASTRUCTURE — see oracle/__init__.py.  Restates the subset of the reference's
``tnac4o/mps.py`` that its solver consumes.  The same
LAPACK drivers are used (``dgeqrf/dorgqr`` via scipy.linalg.qr, ``dgesdd`` with ``dgesvd``
fallback via scipy.linalg.svd) so that agreement with the reference is at rounding level.

Layout: site tensors are C-ordered ``(Dl, p, Dr)`` float64 arrays.
"""
import numpy as np
import scipy.linalg as sla

EPS = np.finfo(np.float64).eps


# --------------------------------------------------------------------------- module fns
def pow2_floor_max(T):
    """Largest |entry| floored to a power of two (reference mps.py:76-85, ``nfactor``).

    Uses the exponent field of the IEEE double, i.e. 2**floor(log2(max|T|)) for normal
    numbers and 2**-1023 for zero / subnormal input (the reference's behaviour).
    """
    x = np.float64(np.max(np.abs(T)))
    bits = np.abs(x).view(np.int64)
    return 2.0 ** ((int(bits) >> 52) - 1023)


nfactor = pow2_floor_max


def qr_pos(T):
    """Economic QR with a non-negative diagonal of R (reference mps.py:43-59)."""
    Q, R = sla.qr(T, mode='economic')
    s = np.sign(np.diag(R))
    s[s == 0] = 1
    return Q * s, s[:, None] * R


def svd_gauged(T):
    """Thin SVD with the reference's sign gauge (mps.py:24-40): a (U column, V row) pair is
    flipped when in both of them the most negative entry outweighs the most positive."""
    try:
        U, S, V = sla.svd(T, full_matrices=False)
    except sla.LinAlgError:
        U, S, V = sla.svd(T, full_matrices=False, lapack_driver='gesvd')
    flip = (np.abs(U.min(0)) > U.max(0)) & (np.abs(V.min(1)) > V.max(1))
    U[:, flip] *= -1
    V[flip] *= -1
    return U, S, V


def svdvals(T):
    """Singular values only (mps.py:62-73)."""
    try:
        return sla.svd(T, full_matrices=False, compute_uv=False)
    except sla.LinAlgError:
        return sla.svd(T, full_matrices=False, compute_uv=False, lapack_driver='gesvd')


def truncate_center(C, Dmax, tol):
    """SVD-truncate a centre matrix (mps.py:802-811).

    keep = min(#(S > S0*max(eps, tol)), Dmax); discarded = sqrt(sum S[keep:]^2)/S0.
    Returns (projL, diag(S_kept), projR, keep, discarded).
    """
    U, S, V = svd_gauged(C)
    tol = max(EPS, tol)
    keep = min(int(np.sum(S > S[0] * tol)), Dmax)
    disc = np.sqrt(np.sum(S[keep:] ** 2)) / S[0]
    return U[:, :keep], np.diag(S[:keep]), V[:keep, :], keep, disc


def absorb_site(A, W, hconj):
    """One site of MPO.MPS absorption (mps.py:753-763).

    W has legs (left b, out p, right b, in p).
      hconj=True : contract A's physical leg with W's *out* leg; new physical = W's in leg;
                   fused bonds are (MPS index major, MPO index minor).
      hconj=False: contract with W's *in* leg; new physical = out leg; (MPO major, MPS minor).
    """
    Dl, p, Dr = A.shape
    if hconj:
        # T[Dl,Dr,a,b,i] = sum_o A[Dl,o,Dr] W[a,o,b,i]
        T = np.tensordot(A, W, axes=(1, 1)).transpose(0, 2, 4, 1, 3)
    else:
        # T[a,o,b,Dl,Dr] = sum_i W[a,o,b,i] A[Dl,i,Dr]
        T = np.tensordot(W, A, axes=(3, 1)).transpose(0, 3, 1, 2, 4)
    s = T.shape
    return np.reshape(T, (s[0] * s[1], s[2], s[3] * s[4]))


def env_left(RL, A, Ac):
    """Left environment step (mps.py:655-658): out[c',a'] = sum Ac[c,s,c'] RL[c,a] A[a,s,a']."""
    return np.tensordot(Ac, np.tensordot(RL, A, axes=(1, 0)), axes=([0, 1], [0, 1]))


def env_right(RR, A, Ac):
    """Right environment step (mps.py:660-663): out[a,c] = sum A[a,s,a'] RR[a',c'] Ac[c,s,c']."""
    return np.tensordot(np.tensordot(A, RR, axes=(2, 0)), Ac, axes=([1, 2], [1, 2]))


# --------------------------------------------------------------------------- containers
class RefMPO:
    """MPO container (mps.py:818-865): ``W[n]`` legs (b_left, p_out, b_right, p_in)."""

    def __init__(self, L):
        self.L = L
        self.W = [np.ones((1, 1, 1, 1)) for _ in range(L)]
        self.support = [0] * L

    def set_direct(self, W, n):
        self.W[n] = W
        self.support[n] = 1


def bond_profile(Dmax, d):
    """Bond dimensions compatible with local dims and a cap (mps.py:644-653)."""
    L = len(d)
    D = [1] * (L + 1)
    for n in range(L):
        D[n + 1] = min(D[n] * d[n], Dmax)
    D[-1] = 1
    for n in range(L - 1, -1, -1):
        D[n] = min(D[n + 1] * d[n], Dmax, D[n])
    return D


class RefMPS:
    """Boundary MPS with an explicit orthogonality centre (mps.py:96-173).

    Only the 'X' start state is provided (that is the one the solver uses, tnac4o.py:1682).
    """

    def __init__(self, d=2, L=2, Dmax=2, canonise='left'):
        d = [d] if isinstance(d, int) else list(d)
        d = (d * ((L + len(d) - 1) // len(d)))[:L]
        self.L, self.d = L, d
        self.D = bond_profile(Dmax, d)
        self.A = []
        for n in range(L):          # mps.py:635-638
            A = np.zeros((self.D[n], d[n], self.D[n + 1]))
            A[0, :, 0] = 1.0 / np.sqrt(d[n])
            self.A.append(A)
        self.C = np.ones((1, 1))
        self.pC = L
        self.normC = 1.0
        self.R = [np.ones((1, 1)) for _ in range(L + 2)]
        self.R[-1] = None
        self.S = [self._unit_S(self.D[n]) for n in range(L + 1)]
        self.discarded = [0] * (L + 1)
        if canonise == 'left':
            self.canonise_left()
        elif canonise == 'right':
            self.canonise_right()
        self.normC = 1.0

    @staticmethod
    def _unit_S(D):
        S = np.zeros(D)
        S[0] = 1.0
        return S

    def copy(self):
        """mps.py:159-173 — note that S and ``discarded`` are *not* carried over."""
        o = RefMPS(d=self.d, L=self.L, Dmax=1, canonise=None)
        o.A = [a.copy() for a in self.A]
        o.C = self.C.copy()
        o.pC, o.normC = self.pC, self.normC
        o.D = self.D[:]
        o.R = self.R[:]
        return o

    # -- absorption -------------------------------------------------------------------
    def apply_mpo(self, M, Hconj=False):
        """mps.py:353-359."""
        for n in range(self.L):
            if M.support[n]:
                self.A[n] = absorb_site(self.A[n], M.W[n], Hconj)
                self.D[n], self.d[n], self.D[n + 1] = self.A[n].shape

    def apply_diagonalO(self, diag, n):
        """mps.py:361-366."""
        self.A[n] *= np.asarray(diag)[None, :, None]

    # -- gauge moves ------------------------------------------------------------------
    def attach_AC(self):
        """A[pC-1] <- A[pC-1].C (mps.py:368-373)."""
        n = self.pC - 1
        self.A[n] = np.tensordot(self.A[n], self.C, axes=(2, 0))

    def attach_CA(self):
        """A[pC] <- C.A[pC] (mps.py:375-380)."""
        n = self.pC
        self.A[n] = np.tensordot(self.C, self.A[n], axes=(1, 0))

    def orth_left(self, n):
        """QR of (Dl*p, Dr); C = R / nfactor(R) (mps.py:532-539, 772-785)."""
        Dl, p, Dr = self.A[n].shape
        Q, C = qr_pos(self.A[n].reshape(Dl * p, Dr))
        nC = pow2_floor_max(C)
        if C.shape == (1, 1):
            Q = Q * np.sign(C.flat[0])
            C = np.ones((1, 1))
        else:
            C = C / nC
        self.A[n] = Q.reshape(Dl, p, C.shape[0])
        self.C = C
        self.normC *= nC
        self.D[n + 1] = C.shape[0]
        self.pC = n + 1

    def orth_right(self, n):
        """QR of the transposed (p*Dr, Dl) matrix (mps.py:541-548, 787-800)."""
        Dl, p, Dr = self.A[n].shape
        Q, C = qr_pos(self.A[n].reshape(Dl, p * Dr).T)
        nC = pow2_floor_max(C)
        if C.shape == (1, 1):
            Q = Q * np.sign(C.flat[0])
            C = np.ones((1, 1))
        else:
            C = C.T / nC
        self.A[n] = Q.T.reshape(C.shape[1], p, Dr)
        self.C = C
        self.normC *= nC
        self.D[n] = C.shape[1]
        self.pC = n

    def truncateC(self, Dmax, tol=None):
        """mps.py:562-585."""
        if 0 < self.pC < self.L:
            if tol is None:
                tol = EPS
            pL, self.C, pR, keep, disc = truncate_center(self.C, Dmax, tol)
            self.A[self.pC - 1] = np.tensordot(self.A[self.pC - 1], pL, axes=(2, 0))
            self.A[self.pC] = np.tensordot(pR, self.A[self.pC], axes=(1, 0))
            self.D[self.pC] = keep
            self.discarded[self.pC] = max(self.discarded[self.pC], disc)
            return disc
        return 0.0

    def canonise_left(self, compress=False, Dmax=np.inf, tol=None):
        """mps.py:202-218."""
        self.C = np.ones((1, 1))
        self.pC = 0
        for n in range(self.L):
            self.attach_CA()
            self.orth_left(n)
            if compress:
                self.truncateC(Dmax, tol)
        self.R[-1] = None

    def canonise_right(self, compress=False, Dmax=np.inf, tol=None):
        """mps.py:220-236."""
        self.C = np.ones((1, 1))
        self.pC = self.L
        for n in range(self.L - 1, -1, -1):
            self.attach_AC()
            self.orth_right(n)
            if compress:
                self.truncateC(Dmax, tol)
        self.R[-1] = None

    # -- mixed environments <self|phi> --------------------------------------------------
    def update_RL_mix(self, phi, n):
        """mps.py:436-444."""
        new = env_left(self.R[n], phi.A[n], self.A[n])
        if n == self.L - 1:
            self.R[self.L + 1] = new.flat[0]
        else:
            self.R[n + 1] = new

    def update_RR_mix(self, phi, n):
        """mps.py:418-426."""
        new = env_right(self.R[n + 1], phi.A[n], self.A[n])
        if n == 0:
            self.R[self.L + 1] = new.flat[0]
        else:
            self.R[n] = new

    def setup_RL_mix(self, phi):
        for n in range(self.L):
            self.update_RL_mix(phi, n)
        return self.R[-1]

    def bond_env_mix(self, phi, n):
        """p x p environment of the physical leg of site n in <self|phi> (mps.py:454-458, 765-769)."""
        T = np.tensordot(np.tensordot(self.R[n], phi.A[n], axes=(1, 0)), self.R[n + 1], axes=(2, 0))
        return np.tensordot(T, self.A[n], axes=([0, 2], [0, 2]))

    def expectation_mix(self, phi, n):
        """mps.py:587-591, 694-698."""
        T = np.tensordot(np.tensordot(self.R[n], phi.A[n], axes=(1, 0)), self.R[n + 1], axes=(2, 0))
        return np.tensordot(T, self.A[n], axes=((0, 1, 2), (0, 1, 2)))

    # -- variational compression ---------------------------------------------------------
    def optimise_site(self, phi, n):
        """A[n] <- RL . phi.A[n] . RR (mps.py:617-621, 748-751)."""
        self.A[n] = np.tensordot(np.tensordot(self.R[n], phi.A[n], axes=(1, 0)), self.R[n + 1], axes=(2, 0))

    def update_S(self):
        """mps.py:550-560."""
        S = svdvals(self.C)
        if self.S[self.pC].size != S.size:
            self.S[self.pC] = self._unit_S(S.size)
        dS = np.sqrt(np.sum((self.S[self.pC] - S) ** 2))
        self.S[self.pC] = S
        return dS

    def variational_compress(self, phi, tol=None, max_sweeps=1):
        """mps.py:238-279."""
        if tol is None:
            tol = EPS
        overlap = self.setup_RL_mix(phi)
        sweeps, diff = 0, 1.0
        while diff > tol:
            if sweeps >= max_sweeps:
                return overlap
            for n in range(self.L - 1, 0, -1):
                self.optimise_site(phi, n)
                self.orth_right(n)
                self.update_S()
                self.update_RR_mix(phi, n)
            diff = 0.0
            for n in range(self.L):
                self.optimise_site(phi, n)
                self.orth_left(n)
                diff = max(diff, self.update_S())
                self.update_RL_mix(phi, n)
            overlap = self.R[-1]
            sweeps += 1
        return overlap

    def compress_mps(self, Dmax=np.inf, tolS=None, tolV=None, max_sweeps=4, graduate_truncation=True):
        """mps.py:175-200.  Returns the overlap <psi|phi> after compression."""
        self.canonise_right()
        phi = self.copy()
        self.discarded = [0] * (self.L + 1)
        if graduate_truncation:
            self.canonise_left(compress=True, Dmax=Dmax * 4, tol=tolS / 10)
            self.variational_compress(phi, tol=tolV, max_sweeps=1)
            self.canonise_right(compress=True, Dmax=Dmax * 2, tol=tolS / 2)
        self.canonise_left(compress=True, Dmax=Dmax, tol=tolS)
        return self.variational_compress(phi, tol=tolV, max_sweeps=max_sweeps)


def mps_dot(phi, psi):
    """<phi|psi> (mps.py:88-93)."""
    RL = np.ones((1, 1))
    for n in range(psi.L):
        RL = env_left(RL, psi.A[n], phi.A[n])
    return RL.flat[0]
