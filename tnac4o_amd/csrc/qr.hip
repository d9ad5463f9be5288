// K3 — economic QR with non-negative diagonal of R (reference mps.py:43-59; call sites orth_left/right :532-548).
//
// Blocked Householder QR designed for launch-level parallelism instead of LAPACK's column-serial panel:
//   for each panel (nb columns of the trailing matrix):
//     1. orthonormalise the panel by 4 rounds of "scaled Gram -> Jacobi eigenvectors -> W <- W D^-1 J"
//        (one-sided block Jacobi on the panel; columns that are exactly zero or collapse to rounding noise are
//        refilled with hash noise so the basis is complete), then normalise;
//     2. Householder reconstruction (Ballard et al. 2014): sign-choosing LU of the top block gives Y (unit lower
//        trapezoidal) and T with  H = I - Y T Y^T,  H[:, :nb] = Q1 S;
//     3. trailing update  A <- H^T A  as three GEMMs on the MFMA kernel.
//   Q = H_1 ... H_P [Z; 0] accumulated backwards with GEMMs; the dense nb x nb diagonal blocks left by step 1 are
//   triangularised by small Householder QRs (Z), which also fixes diag(R) >= 0.
// Every transformation applied to A is orthogonal to rounding (the Gram matrices only steer the rotations), so the
// factorisation is column-wise backward stable like dgeqrf even for the numerically rank-deficient, strongly graded
// matrices this path produces (validated against LAPACK on matrices captured from the droplet sweeps).
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "common.h"

namespace tn {

// ------------------------------------------------------------------------------------------ helpers
__global__ __launch_bounds__(256) void copy_mat_kernel(const double* __restrict__ S, int64_t srs, int64_t scs,
                                                       double* __restrict__ D, int64_t drs, int64_t dcs, int64_t m,
                                                       int64_t n, int colfast) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= m * n) return;
    const int64_t i = colfast ? e / n : e % m, j = colfast ? e % n : e / m;
    D[i * drs + j * dcs] = S[i * srs + j * scs];
}

int copy_mat(hipStream_t st, const double* S, int64_t srs, int64_t scs, double* D, int64_t drs, int64_t dcs, int64_t m,
             int64_t n) {
    if (m <= 0 || n <= 0) return 0;
    const int colfast = (dcs == 1 || scs == 1) ? 1 : 0;
    TN_PROF_LAUNCH(st, PROF_QR_AUX, hipLaunchKernelGGL(copy_mat_kernel, dim3((unsigned)cdiv(m * n, 256)), dim3(256), 0, st, S, srs, scs, D, drs, dcs, m,
                       n, colfast));
    TN_CHECK_LAUNCH("copy_mat_kernel");
    return 0;
}

__global__ __launch_bounds__(256) void fill_mat_kernel(double* __restrict__ D, int64_t drs, int64_t dcs, int64_t m, int64_t n, double v,
                                                       int colfast) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= m * n) return;
    const int64_t i = colfast ? e / n : e % m, j = colfast ? e % n : e / m;
    D[i * drs + j * dcs] = v;
}
static int fill_mat(hipStream_t st, double* D, int64_t drs, int64_t dcs, int64_t m, int64_t n, double v) {
    if (m <= 0 || n <= 0) return 0;
    TN_PROF_LAUNCH(st, PROF_QR_AUX, hipLaunchKernelGGL(fill_mat_kernel, dim3((unsigned)cdiv(m * n, 256)), dim3(256), 0, st, D, drs, dcs, m, n, v,
                       dcs == 1 ? 1 : 0));
    TN_CHECK_LAUNCH("fill_mat_kernel");
    return 0;
}

// column swaps of a panel-pivoting step, applied in order to every row: A(r, pairs[2t]) <-> A(r, pairs[2t+1]), t = 0 .. npairs-1
// The swaps of a pivoting step as ONE gather / scatter: the host composes the sequence of swaps into "column dst[i] receives the
// old column src[i]" over the (at most 64) columns they touch; a thread fetches its row's values of all those columns together and
// then stores them (the swaps applied one after the other cost two dependent round trips each: 12 us per panel).  The list travels
// by value in the kernel arguments: no host-to-device copy, no staging buffer to keep alive.
struct SwapList { int n; int dst[64]; int src[64]; };
__global__ __launch_bounds__(256) void swap_columns_kernel(double* __restrict__ A, int64_t rs, int64_t cs, int64_t m, SwapList sl) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    double* row = A + r * rs;
    double v[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = (i < sl.n) ? row[(int64_t)sl.src[i] * cs] : 0.0;
#pragma unroll
    for (int i = 0; i < 64; ++i)
        if (i < sl.n) row[(int64_t)sl.dst[i] * cs] = v[i];
}
// pairs: npairs swaps (a, b) applied in order -> the moves they amount to
static void compose_swaps(const int* pairs, int npairs, SwapList& sl) {
    int cols[64], from[64], nc = 0;                  // touched columns and, for each, the ORIGINAL column whose data it holds now
    auto slot = [&](int c) { for (int i = 0; i < nc; ++i) if (cols[i] == c) return i; cols[nc] = c; from[nc] = c; return nc++; };
    for (int t = 0; t < npairs; ++t) {
        const int a = pairs[2 * t], b = pairs[2 * t + 1];
        if (a == b) continue;
        const int ia = slot(a), ib = slot(b);
        const int tmp = from[ia]; from[ia] = from[ib]; from[ib] = tmp;
    }
    sl.n = 0;
    for (int i = 0; i < nc; ++i)
        if (from[i] != cols[i]) { sl.dst[sl.n] = cols[i]; sl.src[sl.n] = from[i]; ++sl.n; }
}

__device__ __forceinline__ double hash_unit(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return ((double)(x >> 11) * (1.0 / 9007199254740992.0)) - 0.5;
}

// columns of the panel flagged dead (exactly zero) get deterministic pseudo-random content
__global__ __launch_bounds__(256) void refill_dead_kernel(double* __restrict__ W, int64_t rs, int64_t cs, int64_t m,
                                                          int b, const int* __restrict__ dead, uint64_t seed) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    for (int j = 0; j < b; ++j)
        if (dead[j]) W[r * rs + j * cs] = hash_unit(seed + (uint64_t)r * 64 + j);
}

// ------------------------------------------------------------------------------------------ reconstruction
// One workgroup.  In: top b x b block of the orthonormal panel (strided).  Out: Y1 (unit lower triangular, written
// back in place with explicit zeros/ones), Uinv (b x b), T (b x b upper), and the products that fold T into the tall
// factors so that a block reflector is applied with two GEMMs instead of three:
//   W = Y T^T (trailing update A -= W (Y^T A)),  Wq = Y T (Q accumulation Q -= Wq (Y^T Q)),  Y = [Y1; Q1_below Uinv]:
//   rows below the top block use UT = Uinv T^T and UTq = Uinv T, the top block (Wtop, Wqtop) is produced here.
// Steps: (1) sign-choosing LU of Q1_top - S = L U (Ballard et al.), one barrier per pivot: every thread derives the pivot
// and its multipliers itself, the multipliers go to a separate array; (2) U^-1 and L^-1 by substitution, one column per
// lane with the column in registers (wave 0: U^-1, wave 1: L^-1); (3) T = -U S L^-T and (4) the four 32^3 products on the
// matrix cores, one product per wave.
typedef double d4l __attribute__((ext_vector_type(4)));

template <int NB>
__device__ __forceinline__ void small_mm(const double* __restrict__ Am, bool at, const double* __restrict__ Bm_, bool bt, int P,
                                         int lane, d4l (&acc)[NB / 16][NB / 16]) {
    // acc = op(A) op(B) for NB x NB matrices in LDS (pitch P); at / bt: use the transpose.  One wave.
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int ti = 0; ti < NB / 16; ++ti)
#pragma unroll
        for (int tj = 0; tj < NB / 16; ++tj) acc[ti][tj] = d4l{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < NB / 4; ++ks) {
        const int k = ks * 4 + lk;
        double fa[NB / 16], fb[NB / 16];
#pragma unroll
        for (int t = 0; t < NB / 16; ++t) {
            const int i = t * 16 + li;
            fa[t] = at ? Am[k * P + i] : Am[i * P + k];
            fb[t] = bt ? Bm_[i * P + k] : Bm_[k * P + i];
        }
#pragma unroll
        for (int ti = 0; ti < NB / 16; ++ti)
#pragma unroll
            for (int tj = 0; tj < NB / 16; ++tj)
                acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ti], fb[tj], acc[ti][tj], 0, 0, 0);
    }
}

template <int NB>
__global__ __launch_bounds__(256) void lu_reconstruct_kernel(double* __restrict__ Ytop, int64_t rs, int64_t cs, int b,
                                                             double* __restrict__ Uinv, double* __restrict__ T,
                                                             double* __restrict__ UT, double* __restrict__ UTq,
                                                             double* __restrict__ Wtop, int64_t wrs, int64_t wcs,
                                                             double* __restrict__ Wqtop) {
    constexpr int P = NB + 1;
    __shared__ double Um[NB * P];      // work matrix, then U (zeros below the diagonal), then U S, then T
    __shared__ double Lm[NB * P];      // L (unit lower, zeros above)
    __shared__ double Ui[NB * P];      // U^-1
    __shared__ double Li[NB * P];      // L^-1
    __shared__ double sg[NB];
    double* const Tm = Um;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        double yv[NB * NB / 256];
#pragma unroll
        for (int t = 0; t < NB * NB / 256; ++t) {            // one memory round trip
            const int e = tid + 256 * t, i = e / NB, j = e % NB;
            yv[t] = (i < b && j < b) ? Ytop[i * rs + j * cs] : ((i == j) ? 0.0 : 0.0);
        }
#pragma unroll
        for (int t = 0; t < NB * NB / 256; ++t) {
            const int e = tid + 256 * t, i = e / NB, j = e % NB;
            Um[i * P + j] = yv[t];
            Lm[i * P + j] = (i == j) ? 1.0 : 0.0;
            Ui[i * P + j] = 0.0;
            Li[i * P + j] = 0.0;
        }
    }
    __syncthreads();
    // (1) LU with the sign choice  s_i = -sign(u_ii):  u_ii <- u_ii - s_i  (|pivot| >= 1)
    for (int i = 0; i < b; ++i) {
        const double bii = Um[i * P + i];
        const double sgn = (bii >= 0.0) ? -1.0 : 1.0, piv = bii - sgn, rp = fast_rcp(piv);
        const int rem = b - i - 1;
        for (int e = tid; e < rem * rem; e += 256) {
            const int r = i + 1 + e / rem, c = i + 1 + e % rem;
            const double l = Um[r * P + i] * rp;
            Um[r * P + c] -= l * Um[i * P + c];
            if (c == i + 1) Lm[r * P + i] = l;
        }
        __syncthreads();
        if (tid == 0) { sg[i] = sgn; Um[i * P + i] = piv; }
        if (tid > i && tid < b) Um[tid * P + i] = 0.0;       // column i below the pivot is L's now
    }
    if (tid >= b && tid < NB) { sg[tid] = 1.0; Um[tid * P + tid] = 1.0; }      // padding: identity block
    __syncthreads();
    // (2) one column per lane, the column held in registers
    if (wave == 0 && lane < NB) {
        const int j = lane;
        double x[NB], rd[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) rd[i] = fast_rcp(Um[i * P + i]);
#pragma unroll
        for (int i = NB - 1; i >= 0; --i) {
            double sa[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = i + 1; k < NB; ++k) sa[k & 3] += Um[i * P + k] * x[k];
            const double t = (sa[0] + sa[1]) + (sa[2] + sa[3]);
            x[i] = (i > j) ? 0.0 : ((i == j) ? rd[i] : -t * rd[i]);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) Ui[i * P + j] = x[i];
    } else if (wave == 1 && lane < NB) {
        const int j = lane;
        double x[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            double sa[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < i; ++k) sa[k & 3] += Lm[i * P + k] * x[k];
            const double t = (sa[0] + sa[1]) + (sa[2] + sa[3]);
            x[i] = (i < j) ? 0.0 : ((i == j) ? 1.0 : -t);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) Li[i * P + j] = x[i];
    }
    __syncthreads();
    for (int e = tid; e < NB * NB; e += 256) Um[(e / NB) * P + e % NB] *= sg[e % NB];      // U S, in place (U is done)
    __syncthreads();
    // (3) T = -(U S) L^-T, upper triangle
    const int li = lane & 15, lk = lane >> 4;
    d4l acc[NB / 16][NB / 16];
    if (wave == 0) small_mm<NB>(Tm, false, Li, true, P, lane, acc);
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int ti = 0; ti < NB / 16; ++ti)
#pragma unroll
            for (int tj = 0; tj < NB / 16; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = ti * 16 + lk + 4 * r, j = tj * 16 + li;
                    const double t = (i <= j && j < b) ? -acc[ti][tj][r] : 0.0;
                    Tm[i * P + j] = t;
                    if (i < b && j < b) T[i * b + j] = t;
                }
    }
    __syncthreads();
    // (4) wave 0: UT = Uinv T^T, wave 1: UTq = Uinv T, wave 2: W_top = L T^T, wave 3: Wq_top = L T
    const double* left = (wave < 2) ? Ui : Lm;
    small_mm<NB>(left, false, Tm, (wave & 1) == 0, P, lane, acc);
#pragma unroll
    for (int ti = 0; ti < NB / 16; ++ti)
#pragma unroll
        for (int tj = 0; tj < NB / 16; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ti * 16 + lk + 4 * r, j = tj * 16 + li;
                if (i < b && j < b) {
                    const double v = acc[ti][tj][r];
                    if (wave == 0) UT[i * b + j] = v;
                    else if (wave == 1) UTq[i * b + j] = v;
                    else if (wave == 2) Wtop[i * wrs + j * wcs] = v;
                    else Wqtop[i * rs + j * cs] = v;
                }
            }
    for (int e = tid; e < b * b; e += 256) {
        const int i = e / b, j = e % b;
        Uinv[e] = Ui[i * P + j];
        Ytop[i * rs + j * cs] = Lm[i * P + j];
    }
}

// rows r < nrows:  x = X(r, :b);  X(r,:) <- x S0 (in place),  W1(r,:) <- x S1,  W2(r,:) <- x S2   (all b x b, row-major).
// One workgroup per 256 rows.  Row-major operands (cs == 1) go through an LDS tile so that global loads and stores are
// full 256-byte rows (a thread writing its own row element by element costs 5x the bytes in partial-line writes:
// WRITE_SIZE 65.7 MB vs 12.6 MB algorithmic on the 16384 x 32 panel, profiles/r01_pmc_*); column-major operands are
// already coalesced across the threads of a wave.
template <int NB>
__global__ __launch_bounds__(256) void rows_times_small3_kernel(double* __restrict__ X, int64_t rs, int64_t cs, int64_t nrows,
                                                                int b, const double* __restrict__ S0, const double* __restrict__ S1,
                                                                const double* __restrict__ S2, double* __restrict__ W1, int64_t w1rs,
                                                                int64_t w1cs, double* __restrict__ W2) {
    constexpr int P = NB + 1;
    constexpr bool STAGE = (NB == 32);                   // the 64-wide (first-generation) panels keep the direct form
    __shared__ double Ss[3][NB * NB];
    __shared__ double tile[STAGE ? 256 * P : 1];
    const int tid = threadIdx.x;
    for (int e = tid; e < NB * NB; e += 256) {
        const int i = e / NB, j = e % NB;
        const bool in = (i < b && j < b);
        Ss[0][e] = in ? S0[i * b + j] : 0.0;
        Ss[1][e] = in ? S1[i * b + j] : 0.0;
        Ss[2][e] = in ? S2[i * b + j] : 0.0;
    }
    const int64_t r0 = (int64_t)blockIdx.x * 256;
    const int nr = (int)((nrows - r0 < 256) ? nrows - r0 : 256);
    const int64_t r = r0 + tid;
    const bool live = tid < nr;
    const bool xrow = STAGE && (cs == 1), wrow = STAGE && (w1cs == 1);
    double x[NB];
    if (xrow) {
        for (int e = tid; e < nr * b; e += 256) tile[(e / b) * P + e % b] = X[(r0 + e / b) * rs + e % b];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NB; ++i) x[i] = (live && i < b) ? tile[tid * P + i] : 0.0;
    } else {
#pragma unroll
        for (int i = 0; i < NB; ++i) x[i] = (live && i < b) ? X[r * rs + i * cs] : 0.0;
    }
    __syncthreads();                                     // Ss complete; tile free again
#pragma unroll 1
    for (int o = 0; o < 3; ++o) {
        double* dst = (o == 0) ? X : (o == 1) ? W1 : W2;
        const int64_t drs = (o == 1) ? w1rs : rs, dcs = (o == 1) ? w1cs : cs;
        const bool drow = (o == 1) ? wrow : xrow;
        const double* Sm = Ss[o];
        if (drow) {
            for (int j = 0; j < b; ++j) {
                double y = 0.0;
#pragma unroll
                for (int i = 0; i < NB; ++i) y += x[i] * Sm[i * NB + j];
                tile[tid * P + j] = y;
            }
            __syncthreads();
            for (int e = tid; e < nr * b; e += 256) dst[(r0 + e / b) * drs + e % b] = tile[(e / b) * P + e % b];
            __syncthreads();
        } else if (live) {
            for (int j = 0; j < b; ++j) {
                double y = 0.0;
#pragma unroll
                for (int i = 0; i < NB; ++i) y += x[i] * Sm[i * NB + j];
                dst[r * drs + j * dcs] = y;
            }
        }
    }
}

// MFMA form of rows_times_small3 for 32-wide panels: one workgroup per 256 rows computes the (256 x 32) . (32 x 96)
// product [x S0 | x S1 | x S2] on v_mfma_f64_16x16x4_f64 (wave w owns rows 64w .. 64w+63 = 4 row tiles; its A fragments
// are read from the LDS tile once and kept in registers, so the tile can stage the three outputs one after the other).
// Every global access is a coalesced pass over the tile, whatever the operand layout.
typedef double d4q __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void rows_times_small3_mfma_kernel(double* __restrict__ X, int64_t rs, int64_t cs, int64_t nrows,
                                                                     int b, const double* __restrict__ S0,
                                                                     const double* __restrict__ S1, const double* __restrict__ S2,
                                                                     double* __restrict__ W1, int64_t w1rs, int64_t w1cs,
                                                                     double* __restrict__ W2) {
    constexpr int NB = 32, P = 36;
    __shared__ double Ss[3][NB * NB];
    __shared__ double tile[256 * P];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // every global load of the prologue is issued before the first one is consumed (one memory round trip)
    const int64_t r0 = (int64_t)blockIdx.x * 256;
    const int nr = (int)((nrows - r0 < 256) ? nrows - r0 : 256);
    const bool xrow = (cs == 1);
    double sv[3][NB * NB / 256], xv[NB];
#pragma unroll
    for (int t = 0; t < NB * NB / 256; ++t) {
        const int e = tid + 256 * t, i = e / NB, j = e % NB;
        const bool in = (i < b && j < b);
        sv[0][t] = in ? S0[i * b + j] : 0.0;
        sv[1][t] = in ? S1[i * b + j] : 0.0;
        sv[2][t] = (in && S2) ? S2[i * b + j] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int e = tid + 256 * u;
        const int i = xrow ? e / NB : e % 256, j = xrow ? e % NB : e / 256;
        xv[u] = (i < nr && j < b) ? X[(r0 + i) * rs + j * cs] : 0.0;
    }
#pragma unroll
    for (int t = 0; t < NB * NB / 256; ++t) {
        const int e = tid + 256 * t;
        Ss[0][e] = sv[0][t];
        Ss[1][e] = sv[1][t];
        Ss[2][e] = sv[2][t];
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int e = tid + 256 * u;
        const int i = xrow ? e / NB : e % 256, j = xrow ? e % NB : e / 256;
        tile[i * P + j] = xv[u];
    }
    __syncthreads();
    const int li = lane & 15, lk = lane >> 4;
    double fa[4][8];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) fa[mt][ks] = tile[(wave * 64 + mt * 16 + li) * P + ks * 4 + lk];
    __syncthreads();
#pragma unroll 1
    for (int o = 0; o < 3; ++o) {
        double* dst = (o == 0) ? X : (o == 1) ? W1 : W2;
        if (dst == nullptr) continue;                    // uniform
        const int64_t drs = (o == 1) ? w1rs : rs, dcs = (o == 1) ? w1cs : cs;
        const double* Sm = Ss[o];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            double fb[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) fb[ks] = Sm[(ks * 4 + lk) * NB + nt * 16 + li];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                d4q acc = d4q{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[mt][ks], fb[ks], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) tile[(wave * 64 + mt * 16 + lk + 4 * r) * P + nt * 16 + li] = acc[r];
            }
        }
        __syncthreads();
        const bool drow = (dcs == 1);
#pragma unroll
        for (int u0 = 0; u0 < NB; u0 += 8) {             // 8 LDS reads, then 8 stores
            double ov[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + 256 * (u0 + u);
                const int i = drow ? e / NB : e % 256, j = drow ? e % NB : e / 256;
                ov[u] = tile[i * P + j];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + 256 * (u0 + u);
                const int i = drow ? e / NB : e % 256, j = drow ? e % NB : e / 256;
                if (i < nr && j < b) dst[(r0 + i) * drs + j * dcs] = ov[u];
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ diagonal blocks
// One workgroup per diagonal block: Householder QR  D = Z Tri  with diag(Tri) >= 0.
template <int NB>
__global__ __launch_bounds__(256) void diag_qr_kernel(const double* __restrict__ A, int64_t rs, int64_t cs, int nb,
                                                      int64_t k, double* __restrict__ Zbuf, double* __restrict__ Tri, int* clear64) {
    constexpr int P = NB + 1;
    // the first launch after the last panel of the factorisation: leaves the stream's panel state block (64 words) clean for the
    // next call (cholqr_begin)
    if (clear64 && blockIdx.x == 0 && threadIdx.x < 64) clear64[threadIdx.x] = 0;
    __shared__ double D[NB * P];
    __shared__ double Z[NB * P];
    __shared__ double tau[NB];
    const int tid = threadIdx.x, p = blockIdx.x;
    const int64_t j0 = (int64_t)p * nb;
    const int b = (int)((k - j0 < nb) ? k - j0 : nb);
    for (int e = tid; e < b * b; e += 256) {
        const int i = e / b, j = e % b;
        D[i * P + j] = A[(j0 + i) * rs + (j0 + j) * cs];
        Z[i * P + j] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
    // Householder QR with all 256 threads: thread (c, g) = (tid % NB, tid / NB) owns rows g, g + NG, ... of column c.
    // Per column one pass gives sum_{r>j} a_rj a_rc for every c (partials meet in LDS), from which the reflector and
    // its action follow (same scheme as tsqr_factor_kernel); entries are first scaled to [0.5,1) by a power of two.
    constexpr int NG = 256 / NB;
    __shared__ double part[256];
    __shared__ double dscale;
    {
        double m = 0.0;
        for (int e = tid; e < b * b; e += 256) m = fmax(m, fabs(D[(e / b) * P + e % b]));
        part[tid] = m;
        __syncthreads();
        for (int k = 128; k > 0; k >>= 1) {
            if (tid < k) part[tid] = fmax(part[tid], part[tid + k]);
            __syncthreads();
        }
        if (tid == 0) {
            int ex = 0;
            if (part[0] > 0.0 && part[0] < 1.7e308) frexp(part[0], &ex);
            dscale = ldexp(1.0, ex);
        }
        __syncthreads();
        const double scl = 1.0 / dscale;
        for (int e = tid; e < b * b; e += 256) D[(e / b) * P + e % b] *= scl;
        __syncthreads();
    }
    const int c = tid % NB, g = tid / NB;
    // Fast path: the panel step hands over an orthonormal basis Q1 with span(Q1[:, :j]) = span(panel[:, :j]) (Cholesky-QR and the
    // Householder TSQR are both triangular orthogonalisations), so the block D = S Q1^T A_panel is upper triangular already, up to
    // rounding (|d_ij| <~ eps ||a_j|| below the diagonal).  Then Z = I: dropping entries below 4e-15 of their column's norm is a
    // column-wise backward error of the size every other step of the factorisation makes, and it saves the 2 x 32 serial
    // Householder steps (50 us of pure latency per tn_qr call).  Anything else (64-wide first-generation panels) takes the general path.
    __shared__ int notri;
    if (tid == 0) notri = 0;
    __syncthreads();
    if (tid < b) {
        double below = 0.0, all2 = 0.0;
        for (int r = 0; r < b; ++r) {
            const double x = D[r * P + tid];
            all2 += x * x;
            if (r > tid) below = fmax(below, fabs(x));
        }
        if (!(below * below <= 1.6e-29 * all2)) notri = 1;          // (4e-15)^2; NaN lands here too
    }
    __syncthreads();
    const bool tri = (notri == 0);
    for (int j = 0; j < b && !tri; ++j) {
        double s = 0.0;
        if (c < b)
            for (int r = j + 1 + g; r < b; r += NG) s += D[r * P + j] * D[r * P + c];
        part[tid] = s;
        const double alpha = D[j * P + j], ajc = (c < b) ? D[j * P + c] : 0.0;
        __syncthreads();
        double sc = 0.0, sj = 0.0;
#pragma unroll
        for (int k = 0; k < NG; ++k) { sc += part[k * NB + c]; sj += part[k * NB + j]; }
        const double wj = alpha * alpha + sj;
        double t = 0.0;
        if (sj > 0.0 && wj > 1e-290) {                    // nothing below the diagonal -> H = I (like dlarfg)
            const double rn = fast_rsqrt(wj), nrm = wj * rn;
            const double beta = -copysign(nrm, alpha), d = alpha - beta, invd = fast_rcp(d);
            t = 1.0 + fabs(alpha) * rn;
            if (c > j && c < b) {
                const double f = t * (ajc + sc * invd);
                for (int r = j + 1 + g; r < b; r += NG) D[r * P + c] -= D[r * P + j] * invd * f;
                if (g == 0) D[j * P + c] = ajc - f;
            }
            __syncthreads();                               // column j is still needed unscaled by the updates above
            if (c == j) {
                for (int r = j + 1 + g; r < b; r += NG) D[r * P + j] *= invd;
                if (g == 0) D[j * P + j] = beta;
            }
        }
        if (tid == 0) tau[j] = t;
        __syncthreads();
    }
    for (int j = b - 1; j >= 0 && !tri; --j) {    // Z = H_0 ... H_{b-1}
        const double t = tau[j];
        if (t != 0.0) {                            // uniform
            double s = 0.0;
            if (c < b)
                for (int r = j + g; r < b; r += NG) s += ((r == j) ? 1.0 : D[r * P + j]) * Z[r * P + c];
            part[tid] = s;
            __syncthreads();
            double w = 0.0;
#pragma unroll
            for (int k = 0; k < NG; ++k) w += part[k * NB + c];
            w *= t;
            if (c < b)
                for (int r = j + g; r < b; r += NG) Z[r * P + c] -= ((r == j) ? 1.0 : D[r * P + j]) * w;
            __syncthreads();
        }
    }
    double* zo = Zbuf + (int64_t)p * nb * nb;
    double* to = Tri + (int64_t)p * nb * nb;
    for (int e = tid; e < b * b; e += 256) {
        const int i = e / b, j = e % b;
        const double sj = (D[j * P + j] < 0.0) ? -1.0 : 1.0, si = (D[i * P + i] < 0.0) ? -1.0 : 1.0;
        zo[i * nb + j] = Z[i * P + j] * sj;                  // column j of Z scaled
        to[i * nb + j] = (i <= j) ? D[i * P + j] * si * dscale : 0.0; // row i of Tri scaled (and un-normalised)
    }
}

// R[i][j]: 0 left of the diagonal block, Tri inside it, Z^T A to the right;  Q starts as [Z; 0] (block diagonal Z).
// Both only wait for diag_qr_kernel, so they share one launch: the first nR workgroups write R, the rest Q.
// With fold_b > 0 the LAST panel's reflector is applied on the way (single-level path): its columns of Q are
//     H_p [Z_p; 0] = [Z_p; 0] - (Y_p T_p) (Ytop_p^T Z_p),
// a b x b product every workgroup forms for itself in LDS -- two GEMM launches less per factorisation, and for a one-panel
// factorisation no Q accumulation at all.  Y / Wq: the reflectors and Y T of all panels (strides yrs / ycs), j0f: first column of the
// last panel.
__global__ __launch_bounds__(256) void assemble_R_init_Q_kernel(const double* __restrict__ A, int64_t rs, int64_t cs, int nb, int64_t k, int64_t n,
                                                                const double* __restrict__ Zbuf, const double* __restrict__ Tri,
                                                                double* __restrict__ R, int64_t rrs, int64_t rcs, unsigned nR,
                                                                double* __restrict__ Q, int64_t qrs, int64_t qcs, int64_t m, int colfast,
                                                                const double* __restrict__ Y, const double* __restrict__ Wq, int64_t yrs,
                                                                int64_t ycs, int64_t j0f, int fold_b) {
    if (blockIdx.x < nR) {
        const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (e >= k * n) return;
        const int64_t i = e / n, j = e % n;
        const int64_t p = i / nb, j0 = p * nb;
        const int b = (int)((k - j0 < nb) ? k - j0 : nb);
        const int il = (int)(i - j0);
        double v = 0.0;
        if (j >= j0 + b) {
            const double* z = Zbuf + p * nb * nb;
            if (nb == 32) {                                  // fixed trip count: the 2 x 32 loads go out together (one round trip, not 32)
                double zv[32], av[32];
#pragma unroll
                for (int r = 0; r < 32; ++r) {
                    zv[r] = (r < b) ? z[r * 32 + il] : 0.0;
                    av[r] = (r < b) ? A[(j0 + r) * rs + j * cs] : 0.0;
                }
#pragma unroll
                for (int r = 0; r < 32; ++r) v += zv[r] * av[r];
            } else {
                for (int r = 0; r < b; ++r) v += z[r * nb + il] * A[(j0 + r) * rs + j * cs];
            }
        } else if (j >= j0) {
            v = Tri[p * nb * nb + il * nb + (j - j0)];
        }
        R[i * rrs + j * rcs] = v;
    } else {
        __shared__ double Ms[32 * 33];
        if (fold_b > 0) {                                   // M = Ytop^T Z of the last panel (uniform branch)
            const double* z = Zbuf + (j0f / nb) * nb * nb;
            for (int e = threadIdx.x; e < fold_b * fold_b; e += 256) {
                const int c = e / fold_b, jj = e % fold_b;
                double yv[32], zv[32];
#pragma unroll
                for (int r = 0; r < 32; ++r) {
                    yv[r] = (r < fold_b) ? Y[(j0f + r) * yrs + (j0f + c) * ycs] : 0.0;
                    zv[r] = (r < fold_b) ? z[r * nb + jj] : 0.0;
                }
                double acc = 0.0;
#pragma unroll
                for (int r = 0; r < 32; ++r) acc += yv[r] * zv[r];
                Ms[c * 33 + jj] = acc;
            }
            __syncthreads();
        }
        const int64_t e = (int64_t)(blockIdx.x - nR) * 256 + threadIdx.x;
        if (e >= m * k) return;
        const int64_t i = colfast ? e / k : e % m, j = colfast ? e % k : e / m;
        double v = 0.0;
        if (i < k && i / nb == j / nb) {
            const int64_t p = i / nb;
            v = Zbuf[p * nb * nb + (i - p * nb) * nb + (j - p * nb)];
        }
        if (fold_b > 0 && j >= j0f && i >= j0f) {
            const int jj = (int)(j - j0f);
            double wv[32];
#pragma unroll
            for (int c = 0; c < 32; ++c) wv[c] = (c < fold_b) ? Wq[i * yrs + (j0f + c) * ycs] : 0.0;      // together: one round trip
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < 32; ++c) acc += wv[c] * ((c < fold_b) ? Ms[c * 33 + jj] : 0.0);
            v -= acc;
        }
        Q[i * qrs + j * qcs] = v;
    }
}

// out[j] = sum_i A(i,j)^2 for the columns of an (m x n) block (the host takes the maximum: no atomics, nothing to pre-zero)
__global__ __launch_bounds__(256) void colnorm2_kernel(const double* __restrict__ A, int64_t rs, int64_t cs, int64_t m,
                                                       int64_t n, double* __restrict__ out) {
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int64_t j = blockIdx.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int64_t i = tid;
    for (; i + 768 < m; i += 1024) {                      // four independent loads in flight
        const double x0 = A[i * rs + j * cs], x1 = A[(i + 256) * rs + j * cs], x2 = A[(i + 512) * rs + j * cs],
                     x3 = A[(i + 768) * rs + j * cs];
        s0 += x0 * x0; s1 += x1 * x1; s2 += x2 * x2; s3 += x3 * x3;
    }
    for (; i < m; i += 256) { const double x = A[i * rs + j * cs]; s0 += x * x; }
    double s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) out[j] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- tiny factorisations in ONE workgroup -----------------------------------------------------------------------------------
// A sweep makes ~450 factorisations of at most 4096 elements (1 x 1 edge sites, 16 x 16, 256 x 16 ...): through the blocked path
// each is a dozen launches, ~100 us of pure launch latency.  Here: the matrix in LDS, Householder QR column by column with all 256
// threads (thread (c, g) = column c, row group g; per column one pass gives sum_{r>j} a_rj a_rc for every c, from which the
// reflector and its action follow -- the scheme of diag_qr_kernel), backward accumulation of Q = H_0 ... H_{k-1} [I; 0], signs
// chosen so that diag(R) >= 0.  Entries are first scaled into [0.5, 1) by a power of two.  m n <= 4096, k = min(m, n) <= 32.
__global__ __launch_bounds__(256) void tiny_qr_kernel(const double* __restrict__ A, int64_t rs, int64_t cs, int m, int n, double* __restrict__ Q,
                                                      int64_t qrs, int64_t qcs, double* __restrict__ R, int64_t rrs, int64_t rcs) {
    __shared__ double As[4096];          // m x n, pitch n; below the diagonal: the reflectors (unit diagonal implied)
    __shared__ double Qs[4096];          // m x k, pitch k
    __shared__ double scol[4096];        // per column: sum over the rows below the pivot row
    __shared__ double part[256];
    __shared__ double taus[32];
    __shared__ double dscale_s;
    const int tid = threadIdx.x;
    const int k = m < n ? m : n;
    {
        double mx = 0.0;
        double xv[16];                                       // m n <= 4096 = 16 x 256: one memory round trip for the whole matrix
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = tid + 256 * u;
            xv[u] = (e < m * n) ? A[(int64_t)(e / n) * rs + (int64_t)(e % n) * cs] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = tid + 256 * u;
            if (e < m * n) {
                As[e] = xv[u];
                const double a = fabs(xv[u]);
                mx = (a == a) ? fmax(mx, a) : 1.7e308;
            }
        }
        part[tid] = mx;
        __syncthreads();
        for (int h = 128; h > 0; h >>= 1) {
            if (tid < h) part[tid] = fmax(part[tid], part[tid + h]);
            __syncthreads();
        }
        if (tid == 0) {
            int ex = 0;
            if (part[0] > 0.0 && part[0] < 1.7e308) frexp(part[0], &ex);
            dscale_s = ldexp(1.0, ex);
        }
        __syncthreads();
        const double scl = 1.0 / dscale_s;
        for (int e = tid; e < m * n; e += 256) As[e] *= scl;
        __syncthreads();
    }
    // columns over NC thread-columns (a power of two >= n, at most 256), rows over NG = 256 / NC groups
    int NC = 16;
    while (NC < n && NC < 256) NC <<= 1;
    const int NG = 256 / NC;
    const int cc = tid % NC, g = tid / NC;
    for (int j = 0; j < k; ++j) {
        for (int c = cc; c < n; c += NC) {
            double acc = 0.0;
            if (c >= j)
                for (int r = j + 1 + g; r < m; r += NG) acc += As[r * n + j] * As[r * n + c];
            if (NG > 1) part[g * NC + cc] = acc;             // (NG > 1 means n <= NC: one column per thread-column)
            else scol[c] = acc;
        }
        __syncthreads();
        if (NG > 1 && g == 0 && cc < n) {
            double acc = 0.0;
            for (int h = 0; h < NG; ++h) acc += part[h * NC + cc];
            scol[cc] = acc;
        }
        __syncthreads();
        const double alpha = As[j * n + j], sigma = scol[j];
        double tau = 0.0, beta = alpha, inv = 0.0;
        if (sigma > 0.0 && alpha * alpha + sigma > 1e-290) {     // nothing below the diagonal -> H = I (like dlarfg)
            const double nrm = sqrt(alpha * alpha + sigma);
            beta = -copysign(nrm, alpha);
            tau = (beta - alpha) / beta;
            inv = 1.0 / (alpha - beta);
        }
        double wreg[16];                                          // w_c of this thread's columns (n <= 4096 = 16 x 256)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int c = cc + NC * u;
            wreg[u] = (c < n && c > j) ? tau * (As[j * n + c] + inv * scol[c]) : 0.0;
        }
        __syncthreads();                                          // row j is rewritten below
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int c = cc + NC * u;
            if (c < n && c > j && tau != 0.0) {
                const double w = wreg[u];
                for (int r = j + 1 + g; r < m; r += NG) As[r * n + c] -= As[r * n + j] * inv * w;
                if (g == 0) As[j * n + c] -= w;
            }
        }
        __syncthreads();                                          // column j was read unscaled by the updates above
        for (int r = j + 1 + tid; r < m; r += 256) As[r * n + j] *= inv;
        if (tid == 0) { As[j * n + j] = beta; taus[j] = tau; }
        __syncthreads();
    }
    // Q = H_0 ... H_{k-1} [I; 0]
    for (int e = tid; e < m * k; e += 256) Qs[e] = (e / k == e % k) ? 1.0 : 0.0;
    __syncthreads();
    int NQ = 16;
    while (NQ < k) NQ <<= 1;
    const int NGq = 256 / NQ, cq = tid % NQ, gq = tid / NQ;
    for (int j = k - 1; j >= 0; --j) {
        const double tau = taus[j];
        if (tau != 0.0) {                                         // uniform
            double acc = 0.0;
            if (cq >= j && cq < k)
                for (int r = j + 1 + gq; r < m; r += NGq) acc += As[r * n + j] * Qs[r * k + cq];
            part[gq * NQ + cq] = acc;
            __syncthreads();
            double w = 0.0;
            if (cq >= j && cq < k) {
                for (int h = 0; h < NGq; ++h) w += part[h * NQ + cq];
                w = tau * (Qs[j * k + cq] + w);
            }
            __syncthreads();
            if (cq >= j && cq < k) {
                for (int r = j + 1 + gq; r < m; r += NGq) Qs[r * k + cq] -= As[r * n + j] * w;
                if (gq == 0) Qs[j * k + cq] -= w;
            }
            __syncthreads();
        }
    }
    const double ds = dscale_s;
    for (int e = tid; e < k * n; e += 256) {
        const int i = e / n, c = e % n;
        const double sg = (As[i * n + i] < 0.0) ? -1.0 : 1.0;
        R[(int64_t)i * rrs + (int64_t)c * rcs] = (c >= i) ? sg * As[i * n + c] * ds : 0.0;
    }
    if (Q)
        for (int e = tid; e < m * k; e += 256) {
            const int r = e / k, j = e % k;
            const double sg = (As[j * n + j] < 0.0) ? -1.0 : 1.0;
            Q[(int64_t)r * qrs + (int64_t)j * qcs] = sg * Qs[e];
        }
}

// ------------------------------------------------------------------------------------------ driver
// TN_DEBUG=1: synchronise after each stage and report the first non-finite intermediate (diagnostics only)
static bool dbg_on() { static int v = -1; if (v < 0) { const char* e = getenv("TN_DEBUG"); v = (e && e[0] == '1') ? 1 : 0; } return v == 1; }
static void dbg_check(hipStream_t st, const double* p, int64_t rs, int64_t cs, int64_t m, int64_t n, const char* what, int panel, int it) {
    if (!dbg_on()) return;
    std::vector<double> h((size_t)(m * n));
    (void)hipStreamSynchronize(st);
    for (int64_t i = 0; i < m; ++i)
        for (int64_t j = 0; j < n; ++j) (void)hipMemcpy(&h[i * n + j], p + i * rs + j * cs, 8, hipMemcpyDeviceToHost);
    double mx = 0, mn = 1e300; int bad = 0;
    for (double v : h) { if (!(v == v) || v > 1e300 || v < -1e300) ++bad; else { double a = v < 0 ? -v : v; if (a > mx) mx = a; if (a < mn) mn = a; } }
    fprintf(stderr, "[tn_qr dbg] panel %d it %d %-10s %lldx%lld nonfinite=%d max=%.3e min|.|=%.3e\n", panel, it, what, (long long)m, (long long)n, bad, mx, mn);
}

// The panel step: iterated Cholesky-QR (cholqr.hip) by default, TN_PANEL=tsqr selects the Householder TSQR (A/B, cross-checks).
static bool panel_tsqr() {
    const char* e = getenv("TN_PANEL");                              // read per call: the tests switch it
    return e && e[0] == 't';
}
static int panel_orthonormalize(hipStream_t st, const double* Xin, int64_t irs, int64_t ics, double* X, int64_t rs, int64_t cs, int64_t nrows,
                                int b, void* ws, int64_t ws_bytes, bool tsqr, uint64_t seed, int* fused_base, void* state) {
    if (tsqr) return tsqr_orthonormalize(st, Xin, irs, ics, X, rs, cs, nrows, b, ws, ws_bytes);
    return cholqr_orthonormalize(st, Xin, irs, ics, X, rs, cs, nrows, b, ws, ws_bytes, seed, fused_base, state);
}

constexpr int QR_NBO_MAX = 256;       // widest outer block of the two-level factorisation
struct QrWs {
    double *Y, *Wq, *W, *W2, *UT, *UTq, *gemm_ws2;
    double *G, *Tblk, *Zo, *Zo2, *tmpT;
    double *T, *X, *X2, *part, *Js, *Uinv, *Z, *Tri, *gemm_ws, *cn;
    int* dead;
    int* pairs;          // swap list of the pivoted panel step (2 nb ints)
    char* piv;           // PivState + int permutation (device-side panel pivoting)
    int64_t gemm_ws_bytes;
    void* tsqr_ws;
    int64_t tsqr_bytes;
    void* cq_state = nullptr;      // the stream's panel state block (cholqr_begin), or NULL = head of tsqr_ws
};

static int64_t qr_layout(int64_t m, int64_t n, int nb, char* base, QrWs* w) {
    const int64_t k = m < n ? m : n, P = cdiv(k, nb);
    int64_t off = 0;
    auto take = [&](int64_t bytes) { int64_t o = off; off += align_up(bytes, 256); return base ? base + o : nullptr; };
    double* Y = (double*)take(m * k * 8);
    double* Wq = (double*)take(m * k * 8);            // Y T of every panel (Q accumulation)
    double* Wp = (double*)take(m * nb * 8);           // Y T^T of the current panel (trailing update)
    double* Wp2 = (double*)take(m * nb * 8);          // ... of the next one (look-ahead: the wide update of panel p still reads W_p)
    double* UT = (double*)take((int64_t)nb * nb * 8);
    double* UTq = (double*)take((int64_t)nb * nb * 8);
    double* T = (double*)take(P * nb * nb * 8);
    double* X = (double*)take((int64_t)nb * (n > k ? n : k) * 8);
    double* X2 = (double*)take((int64_t)nb * (n > k ? n : k) * 8);
    double* part = (double*)take((int64_t)64 * nb * nb * 8);
    double* Js = (double*)take((int64_t)nb * nb * 8);
    double* Uinv = (double*)take((int64_t)nb * nb * 8);
    double* Z = (double*)take(P * nb * nb * 8);
    double* Tri = (double*)take(P * nb * nb * 8);
    int* dead = (int*)take(nb * 4);
    double* cn = (double*)take(2 * n * 8);            // column norms^2: [input | current trailing block]
    int* pairs = (int*)take((int64_t)2 * nb * 4);
    char* piv = (char*)take(1024 + n * 4);           // PivState | permutation (int) of the device-side panel pivoting
    // split-K scratch for the tall TN products (b x n, K = m)
    int64_t gw = (int64_t)64 * nb * (n > k ? n : k) * 8;      // upper bound of pick_splitk's partial buffers
    {   // ... and of the outer-block products (NBO x n', K = m) of the two-level path: pick_splitk asks for about 512 tiles of
        // 128 x 128 whatever n' is (s = ceil(512 / tiles)), i.e. <= (512 + tiles) 128^2 doubles; take that bound for the widest case
        const int64_t tiles = cdiv(QR_NBO_MAX, 128) * cdiv(n > k ? n : k, 128);
        const int64_t g2 = (512 + 2 * tiles) * 128 * 128 * 8;
        if (g2 > gw) gw = g2;
    }
    double* gws = (double*)take(gw + 256);
    double* gws2 = (double*)take(gw + 256);           // split-K scratch of the look-ahead stream
    // two-level blocking: Gram of an outer block, its merged T factors (one per block), the two (NBO x n) operands of the
    // outer update and a scratch for the T recurrence
    const int64_t nblk_o = cdiv(k, QR_NBO_MAX / 2);           // enough for the narrowest outer width used (128)
    double* Gm = (double*)take((int64_t)QR_NBO_MAX * QR_NBO_MAX * 8);
    double* Tblk = (double*)take(nblk_o * QR_NBO_MAX * QR_NBO_MAX * 8);
    double* Zb = (double*)take((int64_t)QR_NBO_MAX * (n > k ? n : k) * 8);
    double* Zb2 = (double*)take((int64_t)QR_NBO_MAX * (n > k ? n : k) * 8);
    double* tmpT = (double*)take((int64_t)QR_NBO_MAX * 64 * 8);
    if (w) { w->G = Gm; w->Tblk = Tblk; w->Zo = Zb; w->Zo2 = Zb2; w->tmpT = tmpT; }
    const int64_t tsb = std::max(tsqr_ws_bytes(m, nb < 32 ? nb : 32), cholqr_ws_bytes(m, nb < 32 ? nb : 32));
    void* tsw = (void*)take(tsb);
    if (w) { w->tsqr_ws = tsw; w->tsqr_bytes = tsb; }
    if (w) { w->Wq = Wq; w->W = Wp; w->W2 = Wp2; w->UT = UT; w->UTq = UTq; w->gemm_ws2 = gws2; }
    if (w) { w->Y = Y; w->T = T; w->X = X; w->X2 = X2; w->part = part; w->Js = Js; w->Uinv = Uinv; w->Z = Z; w->Tri = Tri;
             w->dead = dead; w->gemm_ws = gws; w->gemm_ws_bytes = gw; w->cn = cn; w->pairs = pairs; w->piv = piv; }
    return off;
}

int64_t qr_ws_bytes(int64_t m, int64_t n, int nb) { return qr_layout(m, n, nb, nullptr, nullptr); }

// rank_tol > 0 enables the early exit: every second panel the largest column norm of the not-yet-factored trailing block
// is compared with the largest column norm of the input; once it is below rank_tol times that, the remaining rows of R
// would be negligible on the scale of the leading singular value and the factorisation stops with k_eff columns
// (A = Q[:, :k_eff] R[:k_eff, :] to rank_tol * max column norm).  Used by the truncating canonisation passes, whose centre
// matrix is SVD-truncated at eps * S0 right afterwards (the Jacobi SVD deflates rows below 2^-56 anyway); it needs one
// 16-byte read-back per check.  *keff_host receives the number of columns/rows produced.
// ---- two-level blocked factorisation (nb = 32 inside outer blocks of `nbo` columns) ---------------------------------------
// Single-level blocking applies every 32-wide reflector to the whole trailing matrix and again to Q: three passes over up to
// 134 MB per panel at K = 32, i.e. HBM-bound work that fills the device (12.9 GB for one 16384 x 1024 call against 0.28 GB
// compulsory) and, with several chains on the GPU, serialises them.  Here the panels of an outer block only update the
// columns of that block; the columns to its right (and, afterwards, Q) see the block once, through the merged reflector
//      H_1 ... H_q = I - Y_blk T_blk Y_blk^T,   T_blk = [[T_1, -T_1 (Y_1^T Y_2) T_2, ...], [0, T_2, ...], ...]   (dlarft by blocks)
// built from the panels' own (Y_p, T_p) and one Gram matrix G = Y_blk^T Y_blk: rank-nbo GEMMs (K = 128 or 256) instead of
// rank-32 ones, 4-8x fewer bytes.  Used for the plain factorisation only (rank_tol = 0): the rank-revealing early exit of the
// truncating passes checks the trailing block after every second panel, which needs it up to date.
static int qr_two_level(hipStream_t st, Mat Am, int64_t m, int64_t n, int64_t k, Mat Ym, QrWs& w, int nbo, int64_t rs, int64_t cs,
                        int64_t yrs, int64_t ycs, int64_t wrs, int64_t wcs, double* Q, int64_t qrs, int64_t qcs, double* R, int64_t rrs,
                        int64_t rcs, bool use_tsqr, int* fused_base) {
    const int nb = 32;
    int rc;
    const int nblk = (int)cdiv(k, nbo);
    for (int bi = 0; bi < nblk; ++bi) {
        const int64_t J0 = (int64_t)bi * nbo;
        const int bw = (int)((k - J0 < nbo) ? k - J0 : nbo);
        const int64_t Jend = J0 + bw, mb = m - J0;
        Mat Yb = sub(Ym, J0, J0);
        // rows of the block above each panel's own top block must read as zero in the merged reflector
        if ((rc = fill_mat(st, Yb.p, yrs, ycs, bw, bw, 0.0))) return rc;
        double* Tb = w.Tblk + (int64_t)bi * QR_NBO_MAX * QR_NBO_MAX;            // bw x bw, row-major, pitch bw
        if ((rc = fill_mat(st, Tb, bw, 1, bw, bw, 0.0))) return rc;
        const int npan = (int)cdiv(bw, nb);
        for (int q = 0; q < npan; ++q) {
            const int64_t j0 = J0 + (int64_t)q * nb;
            const int b = (int)((Jend - j0 < nb) ? Jend - j0 : nb);
            const int64_t mp = m - j0, nin = Jend - j0;                         // the panel's update stays inside the block
            const int p = (int)(j0 / nb);
            Mat Ap = sub(Am, j0, j0), Yp = sub(Ym, j0, j0);
            double* Tp = w.T + (int64_t)p * nb * nb;
            Mat Wp = mat(w.W, wrs, wcs);
            if (!use_tsqr) {
                // orthonormalisation + Householder reconstruction + the tall products in one chain of launches (cholqr.hip)
                if ((rc = cholqr_panel(st, Ap.p, rs, cs, Yp.p, yrs, ycs, mp, b, w.tsqr_ws, w.tsqr_bytes, (uint64_t)p + 1, 1, Tp, Wp.p, wrs, wcs,
                                       nullptr, fused_base, w.cq_state)))
                    return rc;
            } else {
            if ((rc = panel_orthonormalize(st, Ap.p, rs, cs, Yp.p, yrs, ycs, mp, b, w.tsqr_ws, w.tsqr_bytes, use_tsqr, (uint64_t)p + 1, fused_base, w.cq_state))) return rc;
            // Wq_top goes to a scratch corner of the (otherwise unused here) Wq buffer: only Y, T and W = Y T^T are needed
            TN_PROF_LAUNCH(st, PROF_LU, hipLaunchKernelGGL((lu_reconstruct_kernel<32>), dim3(1), dim3(256), 0, st, Yp.p, yrs, ycs, b, w.Uinv, Tp, w.UT,
                               w.UTq, Wp.p, wrs, wcs, w.Wq));
            TN_CHECK_LAUNCH("lu_reconstruct_kernel");
            if (mp > b) {
                dim3 grid((unsigned)cdiv(mp - b, 256));
                prof_begin(st, PROF_ROWS_SMALL);
                hipLaunchKernelGGL(rows_times_small3_mfma_kernel, grid, dim3(256), 0, st, sub(Yp, b, 0).p, yrs, ycs, mp - b, b, w.Uinv, w.UT,
                                   (const double*)nullptr, sub(Wp, b, 0).p, wrs, wcs, (double*)nullptr);
                TN_CHECK_LAUNCH("rows_times_small3_kernel");
                prof_end(st, PROF_ROWS_SMALL, 4.0 * (mp - b) * b * b, 24.0 * (mp - b) * b);
            }
            }
            Mat Xm = mat(w.X, nin, 1);
            if ((rc = gemm(st, b, nin, mp, 1.0, tr(Yp), Ap, 0.0, Xm, w.gemm_ws, w.gemm_ws_bytes))) return rc;
            if ((rc = gemm(st, mp, nin, b, -1.0, Wp, Xm, 1.0, Ap))) return rc;
            // T_blk: diagonal block = T_p (its pitch is b); column block from the recurrence once the Gram matrix exists (below)
            if ((rc = copy_mat(st, Tp, b, 1, Tb + (int64_t)(j0 - J0) * bw + (j0 - J0), bw, 1, b, b))) return rc;
        }
        // G = Y_blk^T Y_blk, then T_blk[0:c0, c0:c0+b] = -T_blk[0:c0, 0:c0] (G[0:c0, c0:c0+b] T_q)
        Mat Gm = mat(w.G, bw, 1);
        if (npan > 1) {
            if ((rc = gemm(st, bw, bw, mb, 1.0, tr(Yb), Yb, 0.0, Gm, w.gemm_ws, w.gemm_ws_bytes))) return rc;
            for (int q = 1; q < npan; ++q) {
                const int64_t c0 = (int64_t)q * nb;
                const int b = (int)((bw - c0 < nb) ? bw - c0 : nb);
                Mat Tq = mat(Tb + c0 * bw + c0, bw, 1), tmp = mat(w.tmpT, b, 1);
                if ((rc = gemm(st, c0, b, b, 1.0, sub(Gm, 0, c0), Tq, 0.0, tmp))) return rc;
                if ((rc = gemm(st, c0, b, c0, -1.0, mat(Tb, bw, 1), tmp, 0.0, mat(Tb + c0, bw, 1)))) return rc;
            }
        }
        // outer update of everything to the right of the block:  A_r -= Y_blk (T_blk^T (Y_blk^T A_r))
        const int64_t nr = n - Jend;
        if (nr > 0) {
            Mat Ar = sub(Am, J0, Jend), Z = mat(w.Zo, nr, 1), Z2 = mat(w.Zo2, nr, 1);
            if ((rc = gemm(st, bw, nr, mb, 1.0, tr(Yb), Ar, 0.0, Z, w.gemm_ws, w.gemm_ws_bytes))) return rc;
            if ((rc = gemm(st, bw, nr, bw, 1.0, tr(mat(Tb, bw, 1)), Z, 0.0, Z2))) return rc;
            if ((rc = gemm(st, mb, nr, bw, -1.0, Yb, Z2, 1.0, Ar))) return rc;
        }
    }
    // --- triangularise the diagonal blocks, assemble R (as in the single-level path)
    const int P = (int)cdiv(k, nb);
    TN_PROF_LAUNCH(st, PROF_QR_AUX, hipLaunchKernelGGL((diag_qr_kernel<32>), dim3(P), dim3(256), 0, st, Am.p, rs, cs, nb, k, w.Z, w.Tri, (int*)w.cq_state));
    TN_CHECK_LAUNCH("diag_qr_kernel");
    if (w.cq_state) cholqr_end_ok(st);
    // --- Q = H_blk1 ... H_blkB [Z; 0]:  Q[J0:, J0:] -= Y_blk (T_blk (Y_blk^T Q[J0:, J0:]))
    const int qcolfast = (qcs == 1) ? 1 : 0;
    {
        const unsigned nR = (unsigned)cdiv(k * n, 256), nQ = (unsigned)cdiv(m * k, 256);
        TN_PROF_LAUNCH(st, PROF_QR_AUX, hipLaunchKernelGGL(assemble_R_init_Q_kernel, dim3(nR + nQ), dim3(256), 0, st, Am.p, rs, cs, nb, k, n, w.Z, w.Tri, R,
                           rrs, rcs, nR, Q, qrs, qcs, m, qcolfast, (const double*)nullptr, (const double*)nullptr, (int64_t)0, (int64_t)0, (int64_t)0, 0));
        TN_CHECK_LAUNCH("assemble_R_init_Q_kernel");
    }
    Mat Qm = mat(Q, qrs, qcs);
    for (int bi = nblk - 1; bi >= 0; --bi) {
        const int64_t J0 = (int64_t)bi * nbo;
        const int bw = (int)((k - J0 < nbo) ? k - J0 : nbo);
        const int64_t mb = m - J0, nq = k - J0;
        Mat Yb = sub(Ym, J0, J0), Qb = sub(Qm, J0, J0), Z = mat(w.Zo, nq, 1), Z2 = mat(w.Zo2, nq, 1);
        double* Tb = w.Tblk + (int64_t)bi * QR_NBO_MAX * QR_NBO_MAX;
        // (the last block meets [Z; 0]: only its top bw rows are non-zero, the product over the rest adds zeros)
        if ((rc = gemm(st, bw, nq, bi == nblk - 1 ? (int64_t)bw : mb, 1.0, tr(Yb), Qb, 0.0, Z, w.gemm_ws, w.gemm_ws_bytes))) return rc;
        if ((rc = gemm(st, bw, nq, bw, 1.0, mat(Tb, bw, 1), Z, 0.0, Z2))) return rc;
        if ((rc = gemm(st, mb, nq, bw, -1.0, Yb, Z2, 1.0, Qb))) return rc;
    }
    return 0;
}

// Selection of the b largest residuals among positions 0 .. ntr-1 by successive swaps (step t: the first maximum of positions
// t .. ntr-1 goes to position t), through a tournament tree over the positions instead of b linear scans (the scans cost 30 us of
// host time per panel with the device idle): leaf i = position i, a node keeps the position of the larger value, the left one on
// ties (= the first maximum of a scan); a placed position is retired with -infinity.  hcn is permuted along; perm (offset j0) too;
// pairs receives the b swaps (absolute column numbers) for swap_columns_kernel.
static void select_pivots(double* hcn, int64_t ntr, int b, int64_t j0, int64_t* perm, int* pairs) {
    int sz = 1;
    while (sz < ntr) sz <<= 1;
    thread_local std::vector<int> tree;
    thread_local std::vector<double> val;
    tree.assign((size_t)2 * sz, -1);
    val.assign((size_t)sz, -1.0);                   // squared norms are >= 0: -1 never wins
    for (int64_t j = 0; j < ntr; ++j) { val[j] = (hcn[j] == hcn[j]) ? hcn[j] : -1.0; tree[sz + j] = (int)j; }   // NaN never wins (as in a scan)
    auto better = [&](int a, int c) { return (a >= 0 && (c < 0 || val[a] >= val[c])) ? a : c; };
    for (int i = sz - 1; i >= 1; --i) tree[i] = better(tree[2 * i], tree[2 * i + 1]);
    auto update = [&](int pos) { for (int i = (sz + pos) >> 1; i >= 1; i >>= 1) tree[i] = better(tree[2 * i], tree[2 * i + 1]); };
    for (int t = 0; t < b; ++t) {
        int arg = tree[1];                            // leftmost maximum of the live positions t .. ntr-1
        if (arg < 0) arg = t;
        std::swap(hcn[t], hcn[arg]);
        std::swap(perm[j0 + t], perm[j0 + arg]);
        pairs[2 * t] = (int)(j0 + t);
        pairs[2 * t + 1] = (int)(j0 + arg);
        if (arg != t) { val[arg] = (hcn[arg] == hcn[arg]) ? hcn[arg] : -1.0; update(arg); }
        val[t] = -2.0;                                // position t is placed
        tree[sz + t] = -1;
        update(t);
    }
}

// ---- panel pivoting decided on the device ---------------------------------------------------------------------------------------
// The host-side selection above cost a read-back of the residual norms, a stream synchronisation and the host's tournament per panel --
// ~25 us of idle device in the one phase of a row where the host was still inside the chain (2 550 panels per sweep).  Here ONE workgroup
// takes the decisions: it adds the residual norms up (fixed order), applies the exit test, ranks the columns (rank counting in LDS), turns
// the b largest into the same "successive swaps" the host made (step t: the t-th largest goes to position t) and leaves the moves they
// amount to, the permutation and the verdict in device memory.  Every later launch of the factorisation -- column swap, panel step,
// trailing update -- looks at `active` and returns at once after the exit.  The host runs one panel ahead: it reads the verdict of panel
// p - 1 (32 bytes, copied behind the selection launch) before it enqueues panel p, so an exit costs one panel's worth of empty launches
// instead of a synchronisation per panel.  Ties between equal norms go to the lower ORIGINAL position (the host took the lower CURRENT
// one): only exact ties differ.  TN_PIVOT_DEVICE=0 keeps the host selection.
struct PivHead { int active, k_exit, stamp, swap_n; double scale2, dropped2; };      // what the host reads back (32 bytes)
struct PivState { PivHead h; int ticket; int pad[3]; int sel[32]; int dst[64]; int src[64]; };
constexpr int PIV_MAXN = 4096;
__device__ __forceinline__ void piv_sti(int* q, int v) { __hip_atomic_store((__attribute__((address_space(1))) int*)q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int piv_ldi(const int* q) { return __hip_atomic_load((const __attribute__((address_space(1))) int*)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ __launch_bounds__(256) void pivot_init_kernel(PivState* S, int* __restrict__ perm, int n) {      // once per factorisation
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) perm[i] = i;
    if (i < (int)(sizeof(PivState) / 4)) ((int*)S)[i] = 0;
}
// grid = ceil(ntr / 16) workgroups.  Every workgroup adds up all norms (same order: same sum everywhere) and applies the exit test; then it
// ranks ITS 16 columns against all (16 threads per column, a sixteenth of the comparisons each -- fp64 compares: the loop is VALU-bound)
// and publishes the ones among the b largest; the last workgroup to finish (ticket) turns the selection into moves with one wave:
//   position t < b receives the t-th largest column; the front columns that are not selected go, in ascending order, to the positions
//   the selected columns from behind the front vacate, in ascending order; nothing else moves.
// The kernel is a chain of dependent memory round trips (~2 us each): state and norms are fetched together, the permutation is
// brought up to date by the column swap that follows (swap_columns_dev_kernel), not here.
// mail (page-locked HOST memory): the verdict for the host as ONE 8-byte word, (stamp << 32) | (active << 31) | columns accepted, written
// with a relaxed system-scope store -- no copy launch, no event, no release fence; the dropped norm is read at the end of the call
struct PivMail { unsigned long long word; };
__device__ __forceinline__ void piv_post(PivMail* mail, int active, int k_exit, unsigned stamp) {
    const unsigned long long w = ((unsigned long long)stamp << 32) | ((unsigned long long)(active ? 1u : 0u) << 31) | (unsigned long long)(unsigned)k_exit;
    __hip_atomic_store(&mail->word, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
constexpr int PIV_CPB = 16;         // columns per workgroup
__global__ __launch_bounds__(256) void pivot_select_kernel(const double* __restrict__ cn, int ntr, int b, int j0, int p, double tol2, PivState* S,
                                                           PivMail* mail, unsigned stamp) {
    __shared__ double v[PIV_MAXN];
    __shared__ double red[256];
    __shared__ int cnt[16][PIV_CPB];
    __shared__ int vpos[32];
    __shared__ int s_last;
    const int tid = threadIdx.x, blk = blockIdx.x, nblk = gridDim.x;
    // one round trip: the state words and this thread's norms
    const int act = (p > 0) ? S->h.active : 1;
    const double scale_prev = (p > 0) ? S->h.scale2 : 0.0;
    double part = 0.0;
    for (int j = tid; j < ntr; j += 256) {
        const double x = cn[j];
        part += x;                                            // (a NaN stays a NaN in the sum: the exit test then fails, as on the host)
        v[j] = (x == x) ? x : -1.0;                           // ... and never wins a selection
    }
    if (act == 0) {                                           // (written by an earlier launch: the same answer in every workgroup)
        if (blk == 0 && tid == 0) piv_post(mail, 0, S->h.k_exit, stamp);
        return;
    }
    red[tid] = part;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (tid < k) red[tid] += red[tid + k]; __syncthreads(); }
    const double fro2 = red[0];
    const double scale2 = (p == 0) ? fro2 : scale_prev;
    if (p > 0 && fro2 <= tol2 * scale2) {                     // what is left is below the threshold: stop before this panel
        if (blk == 0 && tid == 0) {
            S->h.k_exit = j0; S->h.dropped2 = fro2; S->h.stamp = p + 1; S->h.swap_n = 0; S->h.active = 0;
            piv_post(mail, 0, j0, stamp);
        }
        return;
    }
    {
        const int c = tid & (PIV_CPB - 1), q = tid / PIV_CPB, j = blk * PIV_CPB + c;       // q: 0 .. 15
        const double mine = j < ntr ? v[j] : -2.0;
        const int chunk = (ntr + 15) >> 4, i0 = q * chunk, i1 = (i0 + chunk < ntr) ? i0 + chunk : ntr;
        int r = 0;
        int i = i0;
        for (; i + 8 <= i1; i += 8) {                         // eight LDS reads in flight
            double o[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) o[u] = v[i + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) r += (o[u] > mine || (o[u] == mine && i + u < j)) ? 1 : 0;
        }
        for (; i < i1; ++i) { const double o = v[i]; r += (o > mine || (o == mine && i < j)) ? 1 : 0; }
        cnt[q][c] = r;
        __syncthreads();
        if (q == 0 && j < ntr) {
            int rank = 0;
#pragma unroll
            for (int u = 0; u < 16; ++u) rank += cnt[u][c];
            if (rank < b) piv_sti(&S->sel[rank], j);
        }
    }
    // (the selections are agent-scope stores: they go through to memory, the publisher only waits for their completion -- no release
    //  fence, which would write back the XCD's L2; the last workgroup reads them with agent-scope loads)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) s_last = (atomicAdd(&S->ticket, 1) == nblk - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    const int lane = tid;                                     // (the first wave decides; the barriers below are reached by all 256 threads)
    const int st = (lane < b) ? piv_ldi(&S->sel[lane]) : -1;  // original trailing position of the lane-th largest column
    bool selfront = false;
    int rankV = 0;
    const bool isV = lane < b && st >= b;
    if (lane < 64) {
        for (int t = 0; t < b; ++t) {
            const int x = __shfl(st, t, 64);
            selfront = selfront || x == lane;
            rankV += (isV && x >= b && x < st) ? 1 : 0;
        }
    }
    const bool isD = lane < b && !selfront;
    const bool mv1 = lane < b && st != lane;
    unsigned long long bD = 0ull, b1 = 0ull;
    if (lane < 64) {
        // safety net: the b selections must be b distinct trailing positions.  They feed the column moves of the next launch, where a
        // wrong index would be an out-of-range access; a selection that is not a valid set (never seen; it would mean a published
        // selection was not visible to this workgroup) stops the factorisation with an error mark instead (stamp < 0, read by the host)
        bool dup = false;
        for (int t = 0; t < b; ++t) { const int x = __shfl(st, t, 64); dup = dup || (t != lane && x == st); }
        const bool wrong = lane < b && (st < 0 || st >= ntr || dup);
        const unsigned long long bw = __ballot(wrong);
        if (lane == 0) s_last = (bw != 0ull) ? 2 : 1;
        bD = __ballot(isD); b1 = __ballot(mv1);
    }
    if (isV && rankV < 32) vpos[rankV] = st;
    __syncthreads();
    if (s_last == 2) {
        if (tid == 0) {
            S->ticket = 0;
            S->h.swap_n = 0; S->h.k_exit = j0; S->h.stamp = -(p + 1); S->h.dropped2 = 0.0; S->h.active = 0;
            piv_post(mail, 0, j0, stamp);
        }
        return;
    }
    const unsigned long long below = (lane < 64) ? ((1ull << lane) - 1ull) : 0ull;
    const int n1 = __popcll(b1), nD = __popcll(bD);
    if (mv1) { const int i = __popcll(b1 & below); S->dst[i] = j0 + lane; S->src[i] = j0 + st; }
    if (isD) { const int i = __popcll(bD & below); S->dst[n1 + i] = j0 + vpos[i]; S->src[n1 + i] = j0 + lane; }
    if (tid == 0) {
        S->ticket = 0;
        S->h.swap_n = n1 + nD;
        S->h.k_exit = j0 + b;
        S->h.stamp = p + 1;
        if (p == 0) { S->h.scale2 = fro2; S->h.dropped2 = 0.0; }
        S->h.active = 1;
        piv_post(mail, 1, j0 + b, stamp);
    }
}
// the moves of the selection applied to every row: 64 rows per workgroup, the (at most 64) moves dealt to 4 threads per row, all values
// of a workgroup's rows fetched before the first one is stored (the moves permute columns: sources and destinations overlap)
// (the workgroup after the last one applies the moves to the permutation: perm[dst] <- perm[src])
__global__ __launch_bounds__(256) void swap_columns_dev_kernel(double* __restrict__ A, int64_t rs, int64_t cs, int64_t m, const PivState* __restrict__ S,
                                                               int* __restrict__ perm) {
    // (header and move list are fetched together -- one memory round trip instead of two before the first row load; entries beyond swap_n
    //  are stale but in range of the array and never used)
    const int g = threadIdx.x >> 6;
    int src[16], dst[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { src[i] = S->src[g * 16 + i]; dst[i] = S->dst[g * 16 + i]; }
    const int active = S->h.active, ns = S->h.swap_n;
    if (active == 0 || ns <= 0 || ns > 64) return;
    if (blockIdx.x == gridDim.x - 1) {
        int old = 0;
        if ((int)threadIdx.x < ns) old = perm[S->src[threadIdx.x]];
        __syncthreads();
        if ((int)threadIdx.x < ns) perm[S->dst[threadIdx.x]] = old;
        return;
    }
    const int64_t r = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const bool in = r < m;
    double* row = A + (in ? r : 0) * rs;
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int t = g * 16 + i; v[i] = (in && t < ns) ? row[(int64_t)src[i] * cs] : 0.0; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int t = g * 16 + i; if (in && t < ns) row[(int64_t)dst[i] * cs] = v[i]; }
}
// Look-ahead (aux != nullptr, nb = 32): the trailing update of panel p is split.  The columns of the next panel (and the
// panel's own) are updated on the caller's stream, which then factors panel p+1 right away -- a chain of latency-bound
// single-workgroup kernels -- while the device-filling update of everything to the right of it runs on `aux`.  Ordering:
// aux waits for panel p's reflectors (ev.panel), the caller's stream waits for the wide update of panel p-1 before it touches
// the columns of panel p+1 (ev.wide); W = Y T^T and the split-K scratch are double-buffered between the two streams.
// ---- Q accumulation through merged reflectors (round 5) ---------------------------------------------------------------------
// The single-level paths used to apply every 32-wide reflector to Q on its own: per panel a split-K product Y_p^T Q, its reduction and a
// rank-32 update -- three launches and three passes over Q[j0:, j0:] per panel, ~35 us each in the pivoted factorisations of the first
// pass (2 550 panels per sweep).  Here QMB / 32 consecutive panels act at once,
//      H_1 ... H_q = I - Y_blk T_blk Y_blk^T,      T_blk^-1 = [[T_1^-1, Y_1^T Y_2, ...], [0, T_2^-1, ...], ...]
// (block upper triangular: the inverse of dlarft's merged factor has the panels' own T_p^-1 on the diagonal and the blocks of the Gram
// matrix G = Y_blk^T Y_blk above it), so X = T_blk (Y_blk^T Q) is a block back substitution,  X_q = T_q (Z_q - sum_{j > q} G_qj X_j),
// done by ONE small launch per outer block; T_blk itself is never formed.  Per outer block: G, Z = Y_blk^T Q, the substitution, and a
// rank-QMB update -- four products and three reductions less per four panels, a quarter of the passes over Q.
constexpr int QMB = 128;
// rows of an outer block above each panel's own top block must read as zero in the merged reflector (the workspace is reused)
__global__ __launch_bounds__(256) void zero_above_panels_kernel(double* __restrict__ Y, int64_t rs, int64_t cs, int64_t k, int nb) {
    const int64_t J0 = (int64_t)blockIdx.y * QMB;
    const int bw = (int)((k - J0 < QMB) ? k - J0 : QMB);
    for (int e = blockIdx.x * 256 + threadIdx.x; e < bw * bw; e += gridDim.x * 256) {
        const int i = e / bw, j = e % bw;
        if (i / nb < j / nb) Y[(J0 + i) * rs + (J0 + j) * cs] = 0.0;
    }
}
// X (bw x nq, row-major) = T_blk Z by block back substitution; G: bw x bw row-major (pitch bw); T: the block's first panel factor
// (b x b row-major each, pitch b, 32 x 32 doubles apart).  One workgroup = 32 columns of Z; the block rows of G and the T_q are staged in
// LDS and, panel by panel (last to first), the two small products run on the matrix cores: wave w owns the 16 x 16 tile (w / 2, w % 2)
// of the 32 x 32 result.  (A first version with one thread per 4 x 1 entries and G read from memory inside the dependent loop took 85 us.)
__global__ __launch_bounds__(256) void apply_merged_T_kernel(const double* __restrict__ Z, int64_t nq, const double* __restrict__ G, int bw,
                                                             const double* __restrict__ T, int npan, double* __restrict__ X) {
    constexpr int NP = QMB / 32;
    __shared__ double xs[QMB][33];                    // X so far (rows of the panels already substituted)
    __shared__ double vs[32][33];                     // Z_q - sum_j G_qj X_j
    __shared__ double gs[(NP - 1) * 32][QMB - 32 + 1];  // panel q's block row of G to the right of it: rows 32 q .. 32 q + 31, columns from 0
    __shared__ double ts[NP][32][33];                 // T_q
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4, ti = wave >> 1, tj = wave & 1;
    const int64_t c0 = (int64_t)blockIdx.x * 32;
    // everything the substitution reads goes to LDS / registers in ONE round of loads, issued back to back from fully unrolled loops with
    // masks instead of data-dependent trip counts (a loop whose body is "load, store to LDS" waits for every load before it issues the
    // next one: the per-panel staging spent 21 us of such round trips per launch in the sweep, a first "single round" version with
    // runtime loop bounds 25); the panels are then a chain of matrix-core steps and barriers only
    {
        double gv[12][3], tv[NP * 4];
#pragma unroll
        for (int u1 = 0; u1 < 12; ++u1)
#pragma unroll
            for (int u2 = 0; u2 < 3; ++u2) {
                const int i = (tid >> 5) + 8 * u1, j = (tid & 31) + 32 * u2;       // row i of the block, column 32 (q + 1) + j of G
                const int col = 32 * ((i >> 5) + 1) + j;
                gv[u1][u2] = (i < (npan - 1) * 32 && col < bw) ? G[(int64_t)i * bw + col] : 0.0;
            }
#pragma unroll
        for (int u = 0; u < NP * 4; ++u) {
            const int q = u >> 2, idx = (u & 3) * 256 + tid, i = idx >> 5, l = idx & 31;
            const int bq = (bw - 32 * q < 32) ? bw - 32 * q : 32;
            tv[u] = (q < npan && i < bq && l < bq) ? T[(int64_t)q * 1024 + i * bq + l] : 0.0;
        }
#pragma unroll
        for (int u1 = 0; u1 < 12; ++u1)
#pragma unroll
            for (int u2 = 0; u2 < 3; ++u2) gs[(tid >> 5) + 8 * u1][(tid & 31) + 32 * u2] = gv[u1][u2];
#pragma unroll
        for (int u = 0; u < NP * 4; ++u) { const int idx = (u & 3) * 256 + tid; ts[u >> 2][idx >> 5][idx & 31] = tv[u]; }
    }
    double zr[NP][4];
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 32 * q + 16 * ti + lk + 4 * r;
            const int64_t c = c0 + 16 * tj + lr;
            zr[q][r] = (q < npan && i < bw && c < nq) ? Z[(int64_t)i * nq + c] : 0.0;
        }
    __syncthreads();
#pragma unroll
    for (int qq = 0; qq < NP; ++qq) {
        const int q = NP - 1 - qq;
        if (q >= npan) continue;                                   // (uniform)
        const int r0 = q * 32, kr = bw - r0 - 32;                   // kr: columns of G right of the panel (<= 96; <= 0 for the last one)
        d4q acc = {zr[q][0], zr[q][1], zr[q][2], zr[q][3]};
        for (int kk = 0; kk < kr; kk += 4) {
            const int k = kk + lk;
            const double a = (k < kr) ? -gs[r0 + 16 * ti + lr][k] : 0.0;
            const double b = (k < kr) ? xs[r0 + 32 + k][16 * tj + lr] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) vs[16 * ti + lk + 4 * r][16 * tj + lr] = acc[r];
        __syncthreads();
        d4q x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 32; kk += 4) {
            const int k = kk + lk;
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(ts[q][16 * ti + lr][k], vs[k][16 * tj + lr], x, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) xs[r0 + 16 * ti + lk + 4 * r][16 * tj + lr] = x[r];
        __syncthreads();
    }
    for (int e = tid; e < bw * 32; e += 256) {
        const int i = e >> 5, cc = e & 31;
        if (c0 + cc < nq) X[(int64_t)i * nq + c0 + cc] = xs[i][cc];
    }
}

struct LookaheadEvents {
    hipEvent_t panel[2] = {nullptr, nullptr}, wide[2] = {nullptr, nullptr};
    bool ok = false;
    bool init() {
        if (ok) return true;
        for (int i = 0; i < 2; ++i) {
            if (hipEventCreateWithFlags(&panel[i], hipEventDisableTiming) != hipSuccess) return false;
            if (hipEventCreateWithFlags(&wide[i], hipEventDisableTiming) != hipSuccess) return false;
        }
        return ok = true;
    }
    ~LookaheadEvents() {
        if (!ok) return;
        for (int i = 0; i < 2; ++i) { (void)hipEventDestroy(panel[i]); (void)hipEventDestroy(wide[i]); }
    }
};

// TN_QR_TRACE=1 (diagnostics): every call is timed synchronously with a pair of events and booked under its shape and kind; the
// table is printed when the process exits.  Perturbs the run (one synchronisation per factorisation).
namespace {
struct QrTrace {
    bool on;
    std::mutex mu;
    std::map<std::tuple<int64_t, int64_t, int, int, int>, std::pair<double, long>> tab;
    std::map<std::tuple<int64_t, int64_t, int, int, int>, double> ranks;
    double cat_ms[8] = {0}, cat_rank[8] = {0}, cat_m[8] = {0};
    long cat_calls[8] = {0};
    QrTrace() { const char* e = getenv("TN_QR_TRACE"); on = e && e[0] == '1'; }
    ~QrTrace() {
        if (!on || tab.empty()) return;
        std::vector<std::pair<double, std::tuple<int64_t, int64_t, int, int, int>>> v;
        double tot = 0.0;
        long calls = 0;
        for (auto& kv : tab) { v.push_back({kv.second.first, kv.first}); tot += kv.second.first; calls += kv.second.second; }
        std::sort(v.begin(), v.end(), [](auto& a, auto& b) { return a.first > b.first; });
        fprintf(stderr, "[tn_qr trace] %zu shapes, %ld calls, %.1f ms in total; m n wantQ truncating pivoted : calls, ms, us/call, mean accepted rank\n", v.size(), calls, tot);
        const char* cn[8] = {"plain, n <= 32", "plain, n <= 64", "plain, n <= 128", "plain, n > 128", "truncating, unpivoted", "pivoted, m <= 8192", "pivoted, m > 8192", ""};
        for (int c = 0; c < 7; ++c)
            fprintf(stderr, "  [%s] %ld calls, %.1f ms, mean rows %.0f, mean accepted rank %.1f\n", cn[c], cat_calls[c], cat_ms[c],
                    cat_calls[c] ? cat_m[c] / cat_calls[c] : 0.0, cat_calls[c] ? cat_rank[c] / cat_calls[c] : 0.0);
        static const size_t rows = [] { const char* e = getenv("TN_QR_TRACE_ROWS"); return e ? (size_t)atol(e) : (size_t)60; }();
        for (size_t i = 0; i < v.size() && i < rows; ++i) {
            auto& k = v[i].second;
            auto& st = tab[k];
            fprintf(stderr, "  %6lld %6lld  %d %d %d : %6ld  %8.2f  %8.2f  %7.1f\n", (long long)std::get<0>(k), (long long)std::get<1>(k), std::get<2>(k),
                    std::get<3>(k), std::get<4>(k), st.second, st.first, 1e3 * st.first / st.second, ranks[k] / st.second);
        }
    }
};
QrTrace g_qr_trace;
}  // namespace

static int qr_factor_impl(hipStream_t st, double* A, int64_t rs, int64_t cs, int64_t m, int64_t n, double* Q, int64_t qrs,
                          int64_t qcs, double* R, int64_t rrs, int64_t rcs, int nb, void* ws, int64_t ws_bytes, double rank_tol,
                          int64_t* keff_host, hipStream_t aux, double* dropped2_host, int frob_exit, int64_t* pivot_perm_host, double* nf_out2,
                          int* nf_done, bool* input_intact);
static int qr_factor_traced(hipStream_t st, double* A, int64_t rs, int64_t cs, int64_t m, int64_t n, double* Q, int64_t qrs,
              int64_t qcs, double* R, int64_t rrs, int64_t rcs, int nb, void* ws, int64_t ws_bytes, double rank_tol,
              int64_t* keff_host, hipStream_t aux, double* dropped2_host, int frob_exit, int64_t* pivot_perm_host, double* nf_out2, int* nf_done,
              bool* input_intact);

// nf_out2 != NULL (device, 2 doubles): a path that can divide R by its power-of-two norm factor in the launch that produces it does
// so and sets *nf_done (the one-launch factorisation of smallqr.hip); otherwise *nf_done = 0 and the caller normalises.
// Launches with in-kernel barriers (single-launch panel steps, smallqr.hip) may give up when the co-residency they rely on does not
// hold; unless the caller has deferred the check (FusedDeferCheck: tn_compress_mps asks once per row) the call ends by asking
// fused_timeouts: a factorisation whose input is still intact is redone through the blocked path at once, otherwise the call fails
// with -7 (the input was overwritten: the caller must rerun from a copy; the stream no longer takes the single-launch forms).
int qr_factor(hipStream_t st, double* A, int64_t rs, int64_t cs, int64_t m, int64_t n, double* Q, int64_t qrs,
              int64_t qcs, double* R, int64_t rrs, int64_t rcs, int nb, void* ws, int64_t ws_bytes, double rank_tol,
              int64_t* keff_host, hipStream_t aux, double* dropped2_host, int frob_exit, int64_t* pivot_perm_host, double* nf_out2, int* nf_done) {
    bool intact = false;
    int rc = qr_factor_traced(st, A, rs, cs, m, n, Q, qrs, qcs, R, rrs, rcs, nb, ws, ws_bytes, rank_tol, keff_host, aux, dropped2_host, frob_exit,
                              pivot_perm_host, nf_out2, nf_done, &intact);
    if (fused_check_deferred() || !fused_check_needed()) return rc;
    int gave_up = 0;
    const int rc2 = fused_timeouts(st, &gave_up);
    if (rc2) return rc2;
    if (gave_up == 0) return rc;
    if (!intact) {
        set_error("tn_qr: %d launch(es) with in-kernel barriers gave up (co-residency budget exceeded: another tenant on the device?); the "
                  "results are invalid and the input was overwritten -- rerun from a copy (this stream now takes the six-launch panel chain)", gave_up);
        return -7;
    }
    return qr_factor_traced(st, A, rs, cs, m, n, Q, qrs, qcs, R, rrs, rcs, nb, ws, ws_bytes, rank_tol, keff_host, aux, dropped2_host, frob_exit,
                            pivot_perm_host, nf_out2, nf_done, &intact);
}

static int qr_factor_traced(hipStream_t st, double* A, int64_t rs, int64_t cs, int64_t m, int64_t n, double* Q, int64_t qrs,
              int64_t qcs, double* R, int64_t rrs, int64_t rcs, int nb, void* ws, int64_t ws_bytes, double rank_tol,
              int64_t* keff_host, hipStream_t aux, double* dropped2_host, int frob_exit, int64_t* pivot_perm_host, double* nf_out2, int* nf_done,
              bool* input_intact) {
    if (nf_done) *nf_done = 0;
    if (!g_qr_trace.on)
        return qr_factor_impl(st, A, rs, cs, m, n, Q, qrs, qcs, R, rrs, rcs, nb, ws, ws_bytes, rank_tol, keff_host, aux, dropped2_host, frob_exit,
                              pivot_perm_host, nf_out2, nf_done, input_intact);
    thread_local hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!e0) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); }
    (void)hipEventRecord(e0, st);
    const int rc = qr_factor_impl(st, A, rs, cs, m, n, Q, qrs, qcs, R, rrs, rcs, nb, ws, ws_bytes, rank_tol, keff_host, aux, dropped2_host,
                                  frob_exit, pivot_perm_host, nf_out2, nf_done, input_intact);
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::lock_guard<std::mutex> lk(g_qr_trace.mu);
    const auto key = std::make_tuple(m, n, Q ? 1 : 0, (rank_tol > 0.0 && keff_host) ? 1 : 0, pivot_perm_host ? 1 : 0);
    auto& ss = g_qr_trace.tab[key];
    ss.first += ms;
    ss.second += 1;
    const double kr = keff_host ? (double)*keff_host : (double)(m < n ? m : n);
    g_qr_trace.ranks[key] += kr;
    const int c = pivot_perm_host ? (m > 8192 ? 6 : 5) : (rank_tol > 0.0 && keff_host) ? 4 : (n <= 32 ? 0 : n <= 64 ? 1 : n <= 128 ? 2 : 3);
    g_qr_trace.cat_ms[c] += ms; g_qr_trace.cat_rank[c] += kr; g_qr_trace.cat_calls[c] += 1; g_qr_trace.cat_m[c] += (double)m;
    return rc;
}

static int qr_factor_impl(hipStream_t st, double* A, int64_t rs, int64_t cs, int64_t m, int64_t n, double* Q, int64_t qrs,
                          int64_t qcs, double* R, int64_t rrs, int64_t rcs, int nb, void* ws, int64_t ws_bytes, double rank_tol,
                          int64_t* keff_host, hipStream_t aux, double* dropped2_host, int frob_exit, int64_t* pivot_perm_host, double* nf_out2,
                          int* nf_done, bool* input_intact) {
    TN_CHECK_ARG(m >= 1 && n >= 1, "empty matrix");
    if (dropped2_host) *dropped2_host = 0.0;
    TN_CHECK_ARG(nb == 32 || nb == 64, "nb must be 32 or 64");
    TN_CHECK_ARG(ws_bytes >= qr_ws_bytes(m, n, nb), "workspace too small");
    int64_t k = m < n ? m : n;
    {   // tiny matrices: the whole factorisation in one workgroup (TN_QR_TINY=0: the blocked path; read per call: the tests switch it)
        const char* e_tiny = getenv("TN_QR_TINY");
        const bool tiny_on = !(e_tiny && e_tiny[0] == '0');
        if (tiny_on && nb == 32 && pivot_perm_host == nullptr && m * n <= 4096 && k <= 32) {
            TN_PROF_LAUNCH(st, PROF_QR_AUX, hipLaunchKernelGGL(tiny_qr_kernel, dim3(1), dim3(256), 0, st, (const double*)A, rs, cs, (int)m, (int)n, Q, qrs, qcs,
                               R, rrs, rcs));
            TN_CHECK_LAUNCH("tiny_qr_kernel");
            if (keff_host) *keff_host = k;
            if (input_intact) *input_intact = true;
            return 0;
        }
    }
    QrWs w;
    qr_layout(m, n, nb, (char*)ws, &w);
    // up to 64 columns (no pivoting; the rank-revealing exit only exists from three panels on): the whole factorisation in ONE launch,
    // explicit-Q iterated Cholesky-QR (smallqr.hip); it leaves the input untouched
    if (nb == 32 && pivot_perm_host == nullptr && Q != nullptr && smallqr_fits(m, n)) {
        const int rcs_ = smallqr_factor(st, A, rs, cs, m, n, Q, qrs, qcs, R, rrs, rcs, nf_out2, w.gemm_ws, w.gemm_ws_bytes);
        if (rcs_ == 0) {
            if (keff_host) *keff_host = k;
            if (nf_done) *nf_done = nf_out2 ? 1 : 0;
            if (input_intact) *input_intact = true;
            return 0;
        }
        if (rcs_ != 1) return rcs_;
    }
    int P = (int)cdiv(k, nb);
    const int64_t kfull = k;
    double scale2 = -1.0;                                            // largest squared column norm of the input (lazily read back)
    // Panel pivoting (pivot_perm_host != NULL; nb = 32, rank_tol > 0): before every panel the residual norms of all remaining
    // columns are read back, the factorisation stops when their Frobenius norm is below rank_tol x the input's, otherwise the
    // 32 columns with the largest residuals are swapped to the front and form the next panel (column-pivoted QR at panel
    // granularity: a strong rank revealer at one extra read-back per panel).  pivot_perm_host[j] = input column now at j.
    const bool pivot = pivot_perm_host != nullptr && rank_tol > 0.0 && keff_host != nullptr && nb == 32;
    if (pivot_perm_host)
        for (int64_t j = 0; j < n; ++j) pivot_perm_host[j] = j;
    const bool reveal = !pivot && rank_tol > 0.0 && keff_host != nullptr && P > 2 && nb == 32;
    if (reveal) {
        TN_PROF_LAUNCH(st, PROF_QR_AUX, hipLaunchKernelGGL(colnorm2_kernel, dim3((unsigned)n), dim3(256), 0, st, A, rs, cs, m, n, w.cn));
        TN_CHECK_LAUNCH("colnorm2_kernel");
    }
    // Y shares A's fast direction so panel kernels coalesce the same way
    const bool rowmajor = (cs == 1 && rs != 1);
    const int64_t yrs = rowmajor ? kfull : 1, ycs = rowmajor ? 1 : m;
    const int64_t wrs = rowmajor ? nb : 1, wcs = rowmajor ? 1 : m;
    Mat Am = mat(A, rs, cs), Ym = mat(w.Y, yrs, ycs), Wqm = mat(w.Wq, yrs, ycs);
    int rc;
    const bool use_tsqr = panel_tsqr();
    if (nb == 32 && !use_tsqr && (rc = cholqr_begin(st, w.tsqr_ws, &w.cq_state))) return rc;
    int fbase = 0;                            // arrivals booked by the single-launch panel steps of this call (cholqr.hip)
    {   // two-level blocking for the plain factorisation of matrices with several outer blocks (TN_QR_NBO = 0 disables it)
        const char* e_nbo = getenv("TN_QR_NBO");                      // read per call: the tests switch it
        const int v_nbo = e_nbo ? atoi(e_nbo) : 256, nbo = (v_nbo == 128 || v_nbo == 256) ? v_nbo : 0;
        if (nbo > 0 && nb == 32 && !(rank_tol > 0.0 && keff_host != nullptr) && k >= 2 * nbo && m >= 4 * nbo) {
            if (keff_host) *keff_host = k;
            return qr_two_level(st, Am, m, n, k, Ym, w, nbo, rs, cs, yrs, ycs, wrs, wcs, Q, qrs, qcs, R, rrs, rcs, use_tsqr, &fbase);
        }
    }
    thread_local LookaheadEvents ev;
    // worth it only when there is a wide part to overlap with (at least 4 panels) and the panel is tall enough to be slow
    const bool lookahead = aux != nullptr && aux != st && nb == 32 && P >= 4 && m >= 2048 && ev.init();
    int wide_pending = -1;                    // parity of the ev.wide event the caller's stream has not waited for yet
    hipError_t he;
    auto join_wide = [&]() -> int {           // caller's stream waits for the outstanding wide update
        if (wide_pending >= 0) {
            if ((he = hipStreamWaitEvent(st, ev.wide[wide_pending], 0)) != hipSuccess) return hip_fail(he, "wait wide update");
            wide_pending = -1;
        }
        return 0;
    };
    // device-side panel pivoting (see pivot_select_kernel): the host runs one panel ahead of the verdicts it reads back
    const bool piv_dev_on = [] { const char* e = getenv("TN_PIVOT_DEVICE"); return !(e && e[0] == '0'); }();      // (read per call: the tests switch it)
    const bool piv_dev = pivot && piv_dev_on && n <= PIV_MAXN && !lookahead && !use_tsqr;
    PivState* pst = (PivState*)w.piv;
    int* pperm = (int*)(w.piv + 1024);
    const int* active = piv_dev ? &pst->h.active : nullptr;
    PivMail* ring = piv_dev ? (PivMail*)pinned_host(4 * sizeof(PivMail), 7) : nullptr;
    if (piv_dev && !ring) { set_error("tn_qr: no page-locked memory for the pivoting verdicts"); return 1; }
    if (piv_dev) for (int i = 0; i < 4; ++i) __atomic_store_n(&ring[i].word, 0ull, __ATOMIC_RELAXED);       // (the slot is shared with other read-backs of this thread: no stale word may look like a stamp)
    thread_local unsigned piv_seq = 0;                        // stamps are unique per host thread (the ring is the thread's own; wrap-around after 2^32 panels is harmless: four entries)
    const unsigned seq0 = piv_seq;
    if (piv_dev) piv_seq += (unsigned)P + 1u;
    bool piv_stopped = false;
    GemmExtra gx_active;
    gx_active.skip = active;
    // verdict of panel q (the selection launch in front of it): true = the exit test fired there
    auto piv_verdict = [&](int q, bool& stop) -> int {
        const PivMail* h = &ring[q & 3];
        const unsigned want = seq0 + (unsigned)q + 1u;
        unsigned long long wd = __atomic_load_n(&h->word, __ATOMIC_ACQUIRE);
        for (long spins = 0; (unsigned)(wd >> 32) != want; ++spins) {
            if ((spins & 0xfffff) == 0xfffff && hipStreamQuery(st) != hipErrorNotReady) {          // the stream has drained (or failed): look once more, then give up
                wd = __atomic_load_n(&h->word, __ATOMIC_ACQUIRE);
                if ((unsigned)(wd >> 32) == want) break;
                set_error("tn_qr: the pivoting verdict of panel %d never arrived", q);
                return 1;
            }
            wd = __atomic_load_n(&h->word, __ATOMIC_ACQUIRE);
        }
        stop = ((wd >> 31) & 1ull) == 0ull;
        if (stop) {
            k = (int64_t)(wd & 0x7fffffffull);
            P = (int)(k / nb);
            piv_stopped = true;
        }
        return 0;
    };
    if (piv_dev) {
        TN_PROF_LAUNCH(st, PROF_QR_AUX, hipLaunchKernelGGL(pivot_init_kernel, dim3((unsigned)cdiv(std::max<int64_t>(n, 256), 256)), dim3(256), 0, st, pst, pperm, (int)n));
        TN_CHECK_LAUNCH("pivot_init_kernel");
    }
    for (int p = 0; p < P; ++p) {
        const int64_t j0 = (int64_t)p * nb;
        const int b = (int)((k - j0 < nb) ? k - j0 : nb);
        const int64_t mp = m - j0, ntr = n - j0;
        Mat Ap = sub(Am, j0, j0), Yp = sub(Ym, j0, j0);
        if (piv_dev) {
            if (p >= 1) {
                bool stop = false;
                if ((rc = piv_verdict(p - 1, stop))) return rc;
                if (stop) break;
            }
            TN_PROF_LAUNCH(st, PROF_QR_AUX, hipLaunchKernelGGL(colnorm2_kernel, dim3((unsigned)ntr), dim3(256), 0, st, Ap.p, rs, cs, mp, ntr, w.cn));
            TN_CHECK_LAUNCH("colnorm2_kernel");
            TN_PROF_LAUNCH(st, PROF_QR_AUX, hipLaunchKernelGGL(pivot_select_kernel, dim3((unsigned)cdiv(ntr, PIV_CPB)), dim3(256), 0, st, (const double*)w.cn, (int)ntr, b, (int)j0, p,
                               rank_tol * rank_tol, pst, &ring[p & 3], seq0 + (unsigned)p + 1u));
            TN_CHECK_LAUNCH("pivot_select_kernel");
            TN_PROF_LAUNCH(st, PROF_QR_AUX, hipLaunchKernelGGL(swap_columns_dev_kernel, dim3((unsigned)cdiv(m, 64) + 1), dim3(256), 0, st, A, rs, cs, m,
                               (const PivState*)pst, pperm));
            TN_CHECK_LAUNCH("swap_columns_dev_kernel");
        } else if (pivot) {
            TN_PROF_LAUNCH(st, PROF_QR_AUX, hipLaunchKernelGGL(colnorm2_kernel, dim3((unsigned)ntr), dim3(256), 0, st, Ap.p, rs, cs, mp, ntr, w.cn));
            TN_CHECK_LAUNCH("colnorm2_kernel");
            std::vector<double> hcn_pageable;
            double* hcn = (double*)pinned_host((size_t)ntr * 8, 0);
            if (!hcn) { hcn_pageable.resize((size_t)ntr); hcn = hcn_pageable.data(); }
            if ((he = hipMemcpyAsync(hcn, w.cn, (size_t)ntr * 8, hipMemcpyDeviceToHost, st)) != hipSuccess) return hip_fail(he, "memcpy norms");
            if ((he = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(he, "sync norms");
            double fro2 = 0.0;
            for (int64_t j = 0; j < ntr; ++j) fro2 += hcn[j];
            if (p == 0) scale2 = fro2;
            if (p > 0 && fro2 <= rank_tol * rank_tol * scale2) {     // what is left is below the threshold: stop before this panel
                if (dropped2_host) *dropped2_host = fro2;
                k = j0;
                P = p;
                break;
            }
            int pairs[64];
            select_pivots(hcn, ntr, b, j0, pivot_perm_host, pairs);
            SwapList sl = {};
            compose_swaps(pairs, b, sl);
            if (sl.n > 0)
                TN_PROF_LAUNCH(st, PROF_QR_AUX, hipLaunchKernelGGL(swap_columns_kernel, dim3((unsigned)cdiv(m, 256)), dim3(256), 0, st, A, rs, cs, m, sl));
            TN_CHECK_LAUNCH("swap_columns_kernel");
        }
        // --- panel orthonormalisation
        const bool fused_panel = (nb == 32 && !use_tsqr);     // orthonormalisation + reconstruction + tall products in one chain (cholqr.hip)
        if (fused_panel) {
            // W_p = Y_p T_p^T of EVERY panel is kept (in the m x k array that holds Y T on the other paths): besides the trailing update it
            // serves the Q accumulation, H_p Q = Q - Y_p (W_p^T Q), so Y T is never formed (a third of the panel step's output)
            double* Tpf = w.T + (int64_t)p * nb * nb;
            Mat Wpf = sub(Wqm, j0, j0);
            if ((rc = cholqr_panel(st, Ap.p, rs, cs, Yp.p, yrs, ycs, mp, b, w.tsqr_ws, w.tsqr_bytes, (uint64_t)p + 1, 1, Tpf, Wpf.p, yrs, ycs,
                                   nullptr, &fbase, w.cq_state, active)))
                return rc;
        } else if (nb == 32) {
            if ((rc = panel_orthonormalize(st, Ap.p, rs, cs, Yp.p, yrs, ycs, mp, b, w.tsqr_ws, w.tsqr_bytes, use_tsqr, (uint64_t)p + 1, &fbase, w.cq_state))) return rc;
        } else {
            if ((rc = copy_mat(st, Ap.p, rs, cs, Yp.p, yrs, ycs, mp, b))) return rc;
            const int nchunk = gram_nchunk(mp);
            for (int it = 0; it < 5; ++it) {
                if ((rc = gram_partial(st, Yp.p, ycs, yrs, mp, b, b, nullptr, 1, nchunk, w.part))) return rc;
                const int mode = (it < 4) ? 0 : 1;
                // after the first round the columns are images of unit columns under an orthogonal J: a squared norm below
                // 1e-26 there is rounding noise (possibly structured, e.g. all parallel), so the column is refilled
                if ((rc = eig_small(st, w.part, nchunk, b, 1, mode, 12, it == 0 ? 0.0 : 1e-26, w.Js, mode == 0 ? w.dead : nullptr,
                                    nullptr, nullptr)))
                    return rc;
                dbg_check(st, w.Js, b, 1, b, b, "Js", p, it);
                if ((rc = rows_times_small(st, Yp.p, yrs, ycs, mp, b, w.Js))) return rc;
                dbg_check(st, Yp.p, yrs, ycs, mp < 64 ? mp : 64, b, "W", p, it);
                if (mode == 0) {
                    TN_PROF_LAUNCH(st, PROF_QR_AUX, hipLaunchKernelGGL(refill_dead_kernel, dim3((unsigned)cdiv(mp, 256)), dim3(256), 0, st, Yp.p, yrs, ycs, mp,
                                       b, w.dead, (uint64_t)(0x9E3779B97F4A7C15ULL * (uint64_t)(p * 4 + it + 1))));
                    TN_CHECK_LAUNCH("refill_dead_kernel");
                }
            }
        }
        // --- Householder reconstruction
        double* Tp = w.T + (int64_t)p * nb * nb;
        Mat Wqp = sub(Wqm, j0, j0), Wp = fused_panel ? sub(Wqm, j0, j0) : mat((lookahead && (p & 1)) ? w.W2 : w.W, wrs, wcs);
        if (fused_panel) {
        } else if (nb == 32)
            TN_PROF_LAUNCH(st, PROF_LU, hipLaunchKernelGGL((lu_reconstruct_kernel<32>), dim3(1), dim3(256), 0, st, Yp.p, yrs, ycs, b, w.Uinv, Tp, w.UT, w.UTq,
                               Wp.p, wrs, wcs, Wqp.p));
        else
            TN_PROF_LAUNCH(st, PROF_LU, hipLaunchKernelGGL((lu_reconstruct_kernel<64>), dim3(1), dim3(256), 0, st, Yp.p, yrs, ycs, b, w.Uinv, Tp, w.UT, w.UTq,
                               Wp.p, wrs, wcs, Wqp.p));
        if (!fused_panel) TN_CHECK_LAUNCH("lu_reconstruct_kernel");
        dbg_check(st, Tp, b, 1, b, b, "T", p, 9);
        dbg_check(st, w.Uinv, b, 1, b, b, "Uinv", p, 9);
        if (mp > b && !fused_panel) {            // rows below the top block: Y <- Q1 Uinv,  W <- Q1 (Uinv T^T),  Wq <- Q1 (Uinv T)
            dim3 grid((unsigned)cdiv(mp - b, 256));
            prof_begin(st, PROF_ROWS_SMALL);
            if (nb == 32)
                hipLaunchKernelGGL(rows_times_small3_mfma_kernel, grid, dim3(256), 0, st, sub(Yp, b, 0).p, yrs, ycs, mp - b, b,
                                   w.Uinv, w.UT, w.UTq, sub(Wp, b, 0).p, wrs, wcs, sub(Wqp, b, 0).p);
            else
                hipLaunchKernelGGL((rows_times_small3_kernel<64>), grid, dim3(256), 0, st, sub(Yp, b, 0).p, yrs, ycs, mp - b, b,
                                   w.Uinv, w.UT, w.UTq, sub(Wp, b, 0).p, wrs, wcs, sub(Wqp, b, 0).p);
            TN_CHECK_LAUNCH("rows_times_small3_kernel");
            prof_end(st, PROF_ROWS_SMALL, 6.0 * (mp - b) * b * b, 32.0 * (mp - b) * b);
        }
        // --- trailing update  A[j0:, j0:] -= (Y T^T) (Y^T A[j0:, j0:])
        const int64_t nnar = lookahead ? std::min<int64_t>(ntr, 2 * (int64_t)nb) : ntr;     // columns updated on this stream
        if (lookahead && ntr > nnar) {
            // the wide part goes to the look-ahead stream as soon as this panel's reflectors exist
            if ((he = hipEventRecord(ev.panel[p & 1], st)) != hipSuccess) return hip_fail(he, "record panel");
            if ((he = hipStreamWaitEvent(aux, ev.panel[p & 1], 0)) != hipSuccess) return hip_fail(he, "wait panel");
            const int64_t nw = ntr - nnar;
            Mat Aw = sub(Ap, 0, nnar), Xw = mat(w.X2, nw, 1);
            if ((rc = gemm(aux, b, nw, mp, 1.0, tr(Yp), Aw, 0.0, Xw, w.gemm_ws2, w.gemm_ws_bytes))) return rc;
            if ((rc = gemm(aux, mp, nw, b, -1.0, Wp, Xw, 1.0, Aw))) return rc;
            // columns j0+nb .. j0+2nb (the next panel) were last written by the previous wide update
            if ((rc = join_wide())) return rc;
            if ((he = hipEventRecord(ev.wide[p & 1], aux)) != hipSuccess) return hip_fail(he, "record wide");
            wide_pending = p & 1;
        } else if (lookahead) {
            if ((rc = join_wide())) return rc;
        }
        Mat Xm = mat(w.X, nnar, 1);
        if (piv_dev) {         // (the same two products, skipped on the device once the exit test has fired)
            const Mat Yt = tr(Yp);
            if ((rc = gemm_ex(st, b, nnar, mp, 1.0, Yt.p, Yt.rs, Yt.cs, Ap.p, Ap.rs, Ap.cs, 0.0, Xm.p, Xm.rs, Xm.cs, 1, 0, 0, 0, w.gemm_ws, w.gemm_ws_bytes,
                              &gx_active)))
                return rc;
            if ((rc = gemm_ex(st, mp, nnar, b, -1.0, Wp.p, Wp.rs, Wp.cs, Xm.p, Xm.rs, Xm.cs, 1.0, Ap.p, Ap.rs, Ap.cs, 1, 0, 0, 0, nullptr, 0, &gx_active))) return rc;
        } else {
        if ((rc = gemm(st, b, nnar, mp, 1.0, tr(Yp), Ap, 0.0, Xm, w.gemm_ws, w.gemm_ws_bytes))) return rc;
        if ((rc = gemm(st, mp, nnar, b, -1.0, Wp, Xm, 1.0, Ap))) return rc;
        }
        if (reveal && (p & 1) == 1 && p + 1 < P) {
            if ((rc = join_wide())) return rc;                  // the norms below read the whole trailing block
            const int64_t j1 = j0 + b;
            const int64_t nt = n - j1;
            TN_PROF_LAUNCH(st, PROF_QR_AUX, hipLaunchKernelGGL(colnorm2_kernel, dim3((unsigned)nt), dim3(256), 0, st, sub(Am, j1, j1).p, rs, cs, m - j1,
                               nt, w.cn + n));
            TN_CHECK_LAUNCH("colnorm2_kernel");
            // one read-back: the trailing norms, and (first check only) the input norms stored in front of them
            const size_t nread = (size_t)(scale2 < 0.0 ? n + nt : nt);
            std::vector<double> hcn_pageable;
            double* hcn = (double*)pinned_host(nread * 8, 0);
            if (!hcn) { hcn_pageable.resize(nread); hcn = hcn_pageable.data(); }
            hipError_t e;
            if ((e = hipMemcpyAsync(hcn, scale2 < 0.0 ? w.cn : w.cn + n, nread * 8, hipMemcpyDeviceToHost, st)) != hipSuccess)
                return hip_fail(e, "memcpy norms");
            if ((e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(e, "sync norms");
            const double* tr2 = hcn;
            // measure of "what is left" against the input: largest column norm (default) or, frob_exit, the Frobenius norm
            if (scale2 < 0.0) {
                scale2 = 0.0;
                for (int64_t j = 0; j < n; ++j) scale2 = frob_exit ? scale2 + hcn[j] : std::max(scale2, hcn[j]);
                tr2 = hcn + n;
            }
            double h[2] = {scale2, 0.0};
            for (int64_t j = 0; j < nt; ++j) h[1] = frob_exit ? h[1] + tr2[j] : std::max(h[1], tr2[j]);
            if (h[1] <= rank_tol * rank_tol * scale2) {       // nothing left above the threshold: stop here
                if (dropped2_host) {                           // squared Frobenius norm of the block that is dropped
                    double fro2 = 0.0;
                    for (int64_t j = 0; j < nt; ++j) fro2 += tr2[j];
                    *dropped2_host = fro2;
                }
                k = j1;
                P = p + 1;
                break;
            }
        }
    }
    if ((rc = join_wide())) return rc;
    if (piv_dev) {
        // the last panel's own verdict (it may have stopped the factorisation in front of itself), then the permutation
        if (P >= 1 && k == kfull) {
            bool stop = false;
            if ((rc = piv_verdict(P - 1, stop))) return rc;
        }
        // state block (1024 bytes, for the dropped norm) and permutation are neighbours in the workspace: one copy
        std::vector<int> hp_pageable;
        int* hp = (int*)pinned_host(1024 + (size_t)n * 4, 0);
        if (!hp) { hp_pageable.resize(256 + (size_t)n); hp = hp_pageable.data(); }
        if ((he = hipMemcpyAsync(hp, w.piv, 1024 + (size_t)n * 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return hip_fail(he, "memcpy permutation");
        if ((he = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(he, "sync permutation");
        for (int64_t j = 0; j < n; ++j) pivot_perm_host[j] = hp[256 + j];
        if (((const PivState*)hp)->h.stamp < 0) {
            set_error("tn_qr: the device-side pivot selection of panel %d was not a valid set of columns (TN_PIVOT_DEVICE=0 selects on the host)",
                      -((const PivState*)hp)->h.stamp - 1);
            return 1;
        }
        if (piv_stopped && dropped2_host) *dropped2_host = ((const PivState*)hp)->h.dropped2;
    }
    if (keff_host) *keff_host = k;
    // --- triangularise the diagonal blocks, assemble R
    if (nb == 32) TN_PROF_LAUNCH(st, PROF_QR_AUX, hipLaunchKernelGGL((diag_qr_kernel<32>), dim3(P), dim3(256), 0, st, A, rs, cs, nb, k, w.Z, w.Tri, (int*)w.cq_state));
    else TN_PROF_LAUNCH(st, PROF_QR_AUX, hipLaunchKernelGGL((diag_qr_kernel<64>), dim3(P), dim3(256), 0, st, A, rs, cs, nb, k, w.Z, w.Tri, (int*)nullptr));
    TN_CHECK_LAUNCH("diag_qr_kernel");
    if (w.cq_state) cholqr_end_ok(st);
    dbg_check(st, w.Z, nb, 1, (int64_t)P * nb, nb, "Z", -1, 0);
    dbg_check(st, w.Tri, nb, 1, (int64_t)P * nb, nb, "Tri", -1, 0);
    // --- Q = H_1 ... H_P [Z; 0]; the last panel's reflector is applied by the launch that writes [Z; 0] (nb = 32)
    const int qcolfast = (qcs == 1) ? 1 : 0;
    const int64_t jf = (int64_t)(P - 1) * nb;
    // (every workgroup of the Q part recomputes the b x b product: worth it while there are few of them, i.e. for the small
    //  factorisations whose time is launches; the large ones keep the two GEMMs)
    const int fold_b = (nb == 32 && P >= 1 && m * k <= 256 * 512) ? (int)(k - jf) : 0;
    // fused panel step: the m x k array holds W = Y T^T of every panel and  H_p Q = Q - Y_p (W_p^T Q);  otherwise it holds Y T and
    // H_p Q = Q - (Y_p T_p) (Y_p^T Q).  The kernel below forms  second (first_top^T Z)  either way.
    const bool wform = (nb == 32 && !use_tsqr);
    {
        const unsigned nR = (unsigned)cdiv(k * n, 256), nQ = (unsigned)cdiv(m * k, 256);
        TN_PROF_LAUNCH(st, PROF_QR_AUX, hipLaunchKernelGGL(assemble_R_init_Q_kernel, dim3(nR + nQ), dim3(256), 0, st, A, rs, cs, nb, k, n, w.Z, w.Tri, R, rrs,
                           rcs, nR, Q, qrs, qcs, m, qcolfast, (const double*)(wform ? w.Wq : w.Y), (const double*)(wform ? w.Y : w.Wq), yrs, ycs, jf, fold_b));
        TN_CHECK_LAUNCH("assemble_R_init_Q_kernel");
    }
    Mat Qm = mat(Q, qrs, qcs);
    // merged reflectors (see apply_merged_T_kernel): from two panels on, unless the small-matrix fold above already took the last one
    // (TN_QR_MERGED_Q=0 keeps the panel-by-panel accumulation; read per call: the tests switch it)
    const bool merged_q = wform && Q != nullptr && P >= 2 && fold_b == 0 && [] { const char* e = getenv("TN_QR_MERGED_Q"); return !(e && e[0] == '0'); }();
    if (merged_q) {
        const int nblk = (int)cdiv(k, QMB);
        TN_PROF_LAUNCH(st, PROF_QR_AUX, hipLaunchKernelGGL(zero_above_panels_kernel, dim3(16, (unsigned)nblk), dim3(256), 0, st, w.Y, yrs, ycs, k, nb));
        TN_CHECK_LAUNCH("zero_above_panels_kernel");
        for (int bi = nblk - 1; bi >= 0; --bi) {
            const int64_t J0 = (int64_t)bi * QMB;
            const int bw = (int)((k - J0 < QMB) ? k - J0 : QMB);
            const int64_t mb = m - J0, nq = k - J0;
            const int npan = (int)cdiv(bw, nb);
            Mat Yb = sub(Ym, J0, J0), Qb = sub(Qm, J0, J0), Z = mat(w.Zo, nq, 1), X = mat(w.Zo2, nq, 1), Gm = mat(w.G, bw, 1);
            if (npan > 1 && (rc = gemm(st, bw, bw, mb, 1.0, tr(Yb), Yb, 0.0, Gm, w.gemm_ws, w.gemm_ws_bytes))) return rc;
            // (the last block meets [Z; 0]: only its top bw rows are non-zero, the product over the rest adds zeros)
            if ((rc = gemm(st, bw, nq, bi == nblk - 1 ? (int64_t)bw : mb, 1.0, tr(Yb), Qb, 0.0, Z, w.gemm_ws, w.gemm_ws_bytes))) return rc;
            TN_PROF_LAUNCH(st, PROF_QR_AUX, hipLaunchKernelGGL(apply_merged_T_kernel, dim3((unsigned)cdiv(nq, 32)), dim3(256), 0, st, (const double*)Z.p, nq,
                               (const double*)Gm.p, bw, (const double*)(w.T + (J0 / nb) * nb * nb), npan, X.p));
            TN_CHECK_LAUNCH("apply_merged_T_kernel");
            if ((rc = gemm(st, mb, nq, bw, -1.0, Yb, X, 1.0, Qb))) return rc;
        }
        return 0;
    }
    for (int p = P - 1 - (fold_b > 0 ? 1 : 0); p >= 0; --p) {
        const int64_t j0 = (int64_t)p * nb;
        const int b = (int)((k - j0 < nb) ? k - j0 : nb);
        const int64_t mp = m - j0, nq = k - j0;
        Mat Qp = sub(Qm, j0, j0), Yp = sub(Ym, j0, j0), Wqp = sub(Wqm, j0, j0);
        Mat Xm = mat(w.X, nq, 1);
        // Y^T Q (the last panel meets [Z; 0]: only its top b rows are non-zero, the product over the rest adds zeros)
        if ((rc = gemm(st, b, nq, p == P - 1 ? (int64_t)b : mp, 1.0, tr(wform ? Wqp : Yp), Qp, 0.0, Xm, w.gemm_ws, w.gemm_ws_bytes))) return rc;
        if ((rc = gemm(st, mp, nq, b, -1.0, wform ? Yp : Wqp, Xm, 1.0, Qp))) return rc;                     // Q -= (Y T) (Y^T Q) = Y (W^T Q)
    }
    return 0;
}

}  // namespace tn
