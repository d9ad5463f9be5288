"""CPU restatement of LAPACK's dgebal with job = 'S' (scaling only) — the algorithm behind
scipy.linalg.matrix_balance(env, permute=False, separate=True) at reference tnac4o/tnac4o.py:1845-1847.

TEST INFRASTRUCTURE (oracle): imported only by tests/.  LAPACK is a third-party dependency of the reference (scipy >= 1.3.0,
requirements.txt:2; this container has scipy 1.15.3 on OpenBLAS/LAPACK 3.x).  The published algorithm (LAPACK >= 3.5.0,
dgebal.f "Iterative loop for norm reduction"): 2-norms of row and column i over the full index range, radix 2, a step is
accepted when it shrinks c + r below 0.95 of its value.  Pinned against scipy itself in tests/test_oracle_golden.py."""
import numpy as np

RADIX = 2.0
FACTOR = 0.95
SFMIN1 = np.finfo(np.float64).tiny / np.finfo(np.float64).eps
SFMAX1 = 1.0 / SFMIN1
SFMIN2 = SFMIN1 * RADIX
SFMAX2 = 1.0 / SFMIN2


def gebal_scale(A, max_iter=1000):
    """Returns (scale, iterations): D = diag(scale) with D^-1 A D balanced, scale entries powers of two."""
    A = np.array(A, dtype=np.float64, copy=True)
    n = A.shape[0]
    scale = np.ones(n)
    noconv, it = True, 0
    while noconv and it < max_iter:
        noconv = False
        it += 1
        for i in range(n):
            c = np.linalg.norm(A[:, i])
            r = np.linalg.norm(A[i, :])
            ca = np.abs(A[:, i]).max()
            ra = np.abs(A[i, :]).max()
            if c == 0.0 or r == 0.0:
                continue
            g, f, s = r / RADIX, 1.0, c + r
            while c < g and max(f, c, ca) < SFMAX2 and min(r, g, ra) > SFMIN2:
                f *= RADIX; c *= RADIX; ca *= RADIX; r /= RADIX; g /= RADIX; ra /= RADIX
            g = c / RADIX
            while g >= r and max(r, ra) < SFMAX2 and min(f, c, g, ca) > SFMIN2:
                f /= RADIX; c /= RADIX; g /= RADIX; ca /= RADIX; r *= RADIX; ra *= RADIX
            if c + r >= FACTOR * s:
                continue
            if f < 1.0 and scale[i] < 1.0 and f * scale[i] <= SFMIN1:
                continue
            if f > 1.0 and scale[i] > 1.0 and scale[i] >= SFMAX1 / f:
                continue
            scale[i] *= f
            noconv = True
            A[i, :] *= 1.0 / f
            A[:, i] *= f
    return scale, it
