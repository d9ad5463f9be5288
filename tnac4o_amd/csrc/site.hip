// Site steps of the boundary-MPS sweeps as single entry points: each is a fixed composition of the kernels behind tn_gemm /
// tn_qr / tn_normalize_pow2 with its temporaries carved out of the caller's workspace.  They exist for the host side: with
// the 4 lattice rotations of an instance driven by 4 host threads, every separate library call (and every temporary torch
// tensor) is a GIL hand-over, and those -- not the GPU -- account for ~1 s per chain and sweep (tools/host_overhead.py).
// Results are bit-identical to the separate calls (same kernels, same order, same split-K decisions).
//
//   tn_site_qr           attach_CA / attach_AC (mps.py:368-380) + orth_left / orth_right (mps.py:532-548, 772-800)
//   tn_rar               MPS._mps_RAR  (mps.py:748-751)           optimise_site of the variational sweep
//   tn_env_mix           MPS._mps_RL / _mps_RR (mps.py:655-663)   mixed environments
//   tn_apply_truncation  projectors of truncateC into the neighbouring sites + diagonal centre (mps.py:579-583)
#include <algorithm>

#include "common.h"

namespace tn {

int qr_factor(hipStream_t, double*, int64_t, int64_t, int64_t, int64_t, double*, int64_t, int64_t, double*, int64_t, int64_t, int,
              void*, int64_t, double, int64_t*, hipStream_t, double* dropped2_host = nullptr, int frob_exit = 0,
              int64_t* pivot_perm_host = nullptr, double* nf_out2 = nullptr,
              int* nf_done = nullptr);
int64_t qr_ws_bytes(int64_t, int64_t, int);
int normalize_pow2(hipStream_t, double*, int64_t, double*, void*, int64_t);

static inline int64_t up256(int64_t b) { return align_up(b, 256); }

// ---- attach + QR + nfactor ---------------------------------------------------------------------------------------------
// side 0: M = C (kc x Dl) . A (Dl x p Dr), factored as the (kc p) x Dr matrix  -> Q (kc p x k), R (k x Dr), both row-major
// side 1: M = A (Dl p x Dr) . C (Dr x kc), factored through its transposed (p kc) x Dl view -> Qt (k x p kc) = Q^T, Ct (Dl x k) = R^T
// C == nullptr: no attach, A itself is factored (and destroyed); kc is then ignored (kc = Dl resp. Dr).
struct SiteQrDims { int64_t m, n, k, tmp, rows_l, cols_l; };
static SiteQrDims site_qr_dims(int side, int64_t Dl, int64_t p, int64_t Dr, int64_t kc, bool attach) {
    SiteQrDims d;
    if (side == 0) { const int64_t l = attach ? kc : Dl; d.m = l * p; d.n = Dr; d.tmp = attach ? l * p * Dr : 0; }
    else { const int64_t r = attach ? kc : Dr; d.m = p * r; d.n = Dl; d.tmp = attach ? Dl * p * r : 0; }
    d.k = d.m < d.n ? d.m : d.n;
    d.rows_l = 0; d.cols_l = 0;
    return d;
}

int64_t site_qr_ws_bytes(int side, int64_t Dl, int64_t p, int64_t Dr, int64_t kc, int attach) {
    const SiteQrDims d = site_qr_dims(side, Dl, p, Dr, kc, attach != 0);
    int64_t g = 0;
    if (attach) g = side == 0 ? gemm_ws_bytes(kc, p * Dr, Dl, 1) : gemm_ws_bytes(Dl * p, kc, Dr, 1);
    return up256(d.tmp * 8) + up256(g) + up256(qr_ws_bytes(d.m, d.n, 32)) + 8192;
}

int site_qr(hipStream_t st, int side, double* A, int64_t Dl, int64_t p, int64_t Dr, const double* C, int64_t kc, double* Q, double* R,
            double rank_tol, int64_t* keff_host, double* nf_out2, int* normalised_host, void* ws, int64_t ws_bytes,
            double* dropped2_host, int frob_exit, int64_t* pivot_perm_host) {
    TN_CHECK_ARG(side == 0 || side == 1, "side must be 0 (left sweep) or 1 (right sweep)");
    TN_CHECK_ARG(Dl >= 1 && p >= 1 && Dr >= 1 && (C == nullptr || kc >= 1), "non-positive dimension");
    const bool attach = C != nullptr;
    TN_CHECK_ARG(ws_bytes >= site_qr_ws_bytes(side, Dl, p, Dr, kc, attach ? 1 : 0), "workspace too small");
    const SiteQrDims d = site_qr_dims(side, Dl, p, Dr, kc, attach);
    char* w = (char*)ws;
    double* M = A;
    int rc;
    if (attach) {
        M = (double*)w;
        w += up256(d.tmp * 8);
        const int64_t g = side == 0 ? gemm_ws_bytes(kc, p * Dr, Dl, 1) : gemm_ws_bytes(Dl * p, kc, Dr, 1);
        double* gws = g > 0 ? (double*)w : nullptr;
        w += up256(g);
        if (side == 0) rc = gemm(st, kc, p * Dr, Dl, 1.0, C, Dl, 1, A, p * Dr, 1, 0.0, M, p * Dr, 1, 1, 0, 0, 0, gws, g);
        else rc = gemm(st, Dl * p, kc, Dr, 1.0, A, Dr, 1, C, kc, 1, 0.0, M, kc, 1, 1, 0, 0, 0, gws, g);
        if (rc) return rc;
    }
    const int64_t qws = up256(qr_ws_bytes(d.m, d.n, 32));
    void* qw = w;
    w += qws;
    int64_t keff = d.k;
    int nf_done = 0;
    {
        ProfPhase ph(PH_QR);
        const double dm = (double)d.m, dn = (double)d.k;
        prof_note(PROF_QR_NOMINAL, 1, 4.0 * dm * dn * dn - 4.0 / 3.0 * dn * dn * dn, 8.0 * (2.0 * dm * dn + dn * dn));
        if (side == 0)      // M (m x n) row-major; Q (m x k) row-major; R (k x n) row-major
            rc = qr_factor(st, M, d.n, 1, d.m, d.n, Q, d.k, 1, R, d.n, 1, 32, qw, qws, rank_tol, &keff, nullptr, dropped2_host, frob_exit, pivot_perm_host,
                           nf_out2, &nf_done);
        else                // the (p r) x Dl view of the row-major (Dl, p r) array; Q = Qt^T, R = Ct^T
            rc = qr_factor(st, M, 1, d.m, d.m, d.n, Q, 1, d.m, R, 1, d.k, 32, qw, qws, rank_tol, &keff, nullptr, dropped2_host, frob_exit, pivot_perm_host,
                           nf_out2, &nf_done);
    }
    if (rc) return rc;
    if (keff_host) *keff_host = keff;
    int normalised = nf_done;            // (the one-launch factorisation divides R by its norm factor itself)
    if (!normalised && nf_out2 && keff == d.k) {      // the triangular factor is complete and contiguous: C = R / nfactor(R) (mps.py:781-782, 796-797)
        if ((rc = normalize_pow2(st, R, d.k * d.n, nf_out2, w, 8192))) return rc;
        normalised = 1;
    }
    if (normalised_host) *normalised_host = normalised;
    return 0;
}

// ---- RL . A . RR -------------------------------------------------------------------------------------------------------
int64_t rar_ws_bytes(int64_t c, int64_t a, int64_t s, int64_t a2, int64_t c2) {
    const int64_t g1 = gemm_ws_bytes(c, s * a2, a, 1), g2 = gemm_ws_bytes(c * s, c2, a2, 1);
    return up256(c * s * a2 * 8) + up256(g1 > g2 ? g1 : g2);
}
int rar(hipStream_t st, const double* RL, const double* A, const double* RR, int64_t c, int64_t a, int64_t s, int64_t a2, int64_t c2,
        double* out, void* ws, int64_t ws_bytes) {
    TN_CHECK_ARG(c >= 1 && a >= 1 && s >= 1 && a2 >= 1 && c2 >= 1, "non-positive dimension");
    TN_CHECK_ARG(ws_bytes >= rar_ws_bytes(c, a, s, a2, c2), "workspace too small");
    double* T = (double*)ws;
    double* gws = (double*)((char*)ws + up256(c * s * a2 * 8));
    const int64_t g1 = gemm_ws_bytes(c, s * a2, a, 1), g2 = gemm_ws_bytes(c * s, c2, a2, 1);
    int rc;
    if ((rc = gemm(st, c, s * a2, a, 1.0, RL, a, 1, A, s * a2, 1, 0.0, T, s * a2, 1, 1, 0, 0, 0, g1 > 0 ? gws : nullptr, g1))) return rc;
    return gemm(st, c * s, c2, a2, 1.0, T, a2, 1, RR, c2, 1, 0.0, out, c2, 1, 1, 0, 0, 0, g2 > 0 ? gws : nullptr, g2);
}

// ---- mixed environments ------------------------------------------------------------------------------------------------
// side 0 (left):  out[c2, a2] = sum_{c,s,a} Ac[c,s,c2] R[c,a] A[a,s,a2]       R: (c x a)
// side 1 (right): out[a, c]   = sum_{s,a2,c2} A[a,s,a2] R[a2,c2] Ac[c,s,c2]   R: (a2 x c2)
int64_t env_mix_ws_bytes(int side, int64_t a, int64_t s, int64_t a2, int64_t c, int64_t c2) {
    int64_t t, g1, g2;
    if (side == 0) { t = c * s * a2; g1 = gemm_ws_bytes(c, s * a2, a, 1); g2 = gemm_ws_bytes(c2, a2, c * s, 1); }
    else { t = a * s * c2; g1 = gemm_ws_bytes(a * s, c2, a2, 1); g2 = gemm_ws_bytes(a, c, s * c2, 1); }
    return up256(t * 8) + up256(g1 > g2 ? g1 : g2);
}
int env_mix(hipStream_t st, int side, const double* R, const double* A, const double* Ac, int64_t a, int64_t s, int64_t a2, int64_t c,
            int64_t c2, double* out, void* ws, int64_t ws_bytes) {
    TN_CHECK_ARG(side == 0 || side == 1, "side must be 0 (left) or 1 (right)");
    TN_CHECK_ARG(a >= 1 && s >= 1 && a2 >= 1 && c >= 1 && c2 >= 1, "non-positive dimension");
    TN_CHECK_ARG(ws_bytes >= env_mix_ws_bytes(side, a, s, a2, c, c2), "workspace too small");
    double* T = (double*)ws;
    int rc;
    if (side == 0) {
        double* gws = (double*)((char*)ws + up256(c * s * a2 * 8));
        const int64_t g1 = gemm_ws_bytes(c, s * a2, a, 1), g2 = gemm_ws_bytes(c2, a2, c * s, 1);
        if ((rc = gemm(st, c, s * a2, a, 1.0, R, a, 1, A, s * a2, 1, 0.0, T, s * a2, 1, 1, 0, 0, 0, g1 > 0 ? gws : nullptr, g1))) return rc;
        // out = Ac(c s, c2)^T . T(c s, a2)
        return gemm(st, c2, a2, c * s, 1.0, Ac, 1, c2, T, a2, 1, 0.0, out, a2, 1, 1, 0, 0, 0, g2 > 0 ? gws : nullptr, g2);
    }
    double* gws = (double*)((char*)ws + up256(a * s * c2 * 8));
    const int64_t g1 = gemm_ws_bytes(a * s, c2, a2, 1), g2 = gemm_ws_bytes(a, c, s * c2, 1);
    if ((rc = gemm(st, a * s, c2, a2, 1.0, A, a2, 1, R, c2, 1, 0.0, T, c2, 1, 1, 0, 0, 0, g1 > 0 ? gws : nullptr, g1))) return rc;
    // out = T(a, s c2) . Ac(c, s c2)^T
    return gemm(st, a, c, s * c2, 1.0, T, s * c2, 1, Ac, 1, s * c2, 0.0, out, c, 1, 1, 0, 0, 0, g2 > 0 ? gws : nullptr, g2);
}

// ---- projectors of a truncation ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void diag_from_vec_kernel(const double* __restrict__ S, int64_t k, double* __restrict__ D) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < k * k) D[e] = (e / k == e % k) ? S[e / k] : 0.0;
}
int64_t apply_truncation_ws_bytes(int64_t ml, int64_t k0, int64_t keep, int64_t k1, int64_t nr) {
    const int64_t g1 = gemm_ws_bytes(ml, keep, k0, 1), g2 = gemm_ws_bytes(keep, nr, k1, 1);
    return up256(g1 > g2 ? g1 : g2);
}
int apply_truncation(hipStream_t st, const double* Al, int64_t ml, int64_t k0, const double* U, int64_t urs, int64_t ucs, int64_t keep,
                     const double* Vt, int64_t vrs, int64_t vcs, const double* Ar, int64_t k1, int64_t nr, const double* S, double* Al_new,
                     double* Ar_new, double* Cdiag, void* ws, int64_t ws_bytes) {
    TN_CHECK_ARG(ml >= 1 && k0 >= 1 && keep >= 1 && k1 >= 1 && nr >= 1, "non-positive dimension");
    TN_CHECK_ARG(ws_bytes >= apply_truncation_ws_bytes(ml, k0, keep, k1, nr), "workspace too small");
    const int64_t g1 = gemm_ws_bytes(ml, keep, k0, 1), g2 = gemm_ws_bytes(keep, nr, k1, 1);
    int rc;
    if ((rc = gemm(st, ml, keep, k0, 1.0, Al, k0, 1, U, urs, ucs, 0.0, Al_new, keep, 1, 1, 0, 0, 0, g1 > 0 ? (double*)ws : nullptr, g1))) return rc;
    if ((rc = gemm(st, keep, nr, k1, 1.0, Vt, vrs, vcs, Ar, nr, 1, 0.0, Ar_new, nr, 1, 1, 0, 0, 0, g2 > 0 ? (double*)ws : nullptr, g2))) return rc;
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(diag_from_vec_kernel, dim3((unsigned)cdiv(keep * keep, 256)), dim3(256), 0, st, S, keep, Cdiag));
    TN_CHECK_LAUNCH("diag_from_vec_kernel");
    return 0;
}

// ---- weights of a bond from the Gram matrix of the unfactored part on its other side (rank-revealing first pass) -----------
// G (n x n, row-major): Gram matrix of the part of MPO.psi left of a bond with respect to that bond's index (any positive
// overall scale).  d2[c] = max(G_cc, floor_rel * max_c G_cc): squared weight of index c, floored so that entries the recursion
// cannot resolve are over- rather than underestimated; stats[0..63] = partial sums of sum_{c,c'} (G_cc' / (d_c d_c'))^2, the squared
// Frobenius norm of the scaled Gram matrix K (lambda_max(K) <= ||K||_F bounds the operator norm of the scaled left part),
// stats[64] = max G_cc.
constexpr int GW_PARTS = 64;
__global__ __launch_bounds__(256) void gram_diag_kernel(const double* __restrict__ G, int n, double floor_rel, double* __restrict__ d2,
                                                        double* __restrict__ stats) {
    __shared__ double red[256];
    __shared__ double gmax;
    const int tid = threadIdx.x;
    double m = 0.0;
    for (int c = tid; c < n; c += 256) m = fmax(m, G[(int64_t)c * n + c]);
    red[tid] = m;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (tid < k) red[tid] = fmax(red[tid], red[tid + k]); __syncthreads(); }
    if (tid == 0) { gmax = red[0]; stats[GW_PARTS] = red[0]; }
    __syncthreads();
    const double fl = gmax * floor_rel;
    for (int c = tid; c < n; c += 256) d2[c] = fmax(G[(int64_t)c * n + c], fl);
}
// part[b] = sum over the rows owned by block b of (G_ij)^2 / (d2_i d2_j): fixed assignment and fixed order inside a block, so the
// figure is reproducible bit for bit (the caller adds the GW_PARTS partial sums in order)
__global__ __launch_bounds__(256) void gram_kfro_kernel(const double* __restrict__ G, int n, const double* __restrict__ d2,
                                                        double* __restrict__ part) {
    __shared__ double red[256];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int rows_per = (n + GW_PARTS - 1) / GW_PARTS;
    const int r0 = b * rows_per, r1 = (r0 + rows_per < n) ? r0 + rows_per : n;
    double s = 0.0;
    for (int i = r0; i < r1; ++i) {
        const double ri = 1.0 / d2[i];
        const double* g = G + (int64_t)i * n;
        for (int j = tid; j < n; j += 256) { const double x = g[j]; s += x * x * ri / d2[j]; }
    }
    red[tid] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (tid < k) red[tid] += red[tid + k]; __syncthreads(); }
    if (tid == 0) part[b] = red[0];
}
int gram_weights(hipStream_t st, const double* G, int64_t n, double floor_rel, double* d2, double* stats) {
    TN_CHECK_ARG(n >= 1 && n <= 65536 && floor_rel >= 0.0, "bad arguments");
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(gram_diag_kernel, dim3(1), dim3(256), 0, st, G, (int)n, floor_rel, d2, stats));
    TN_CHECK_LAUNCH("gram_diag_kernel");
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(gram_kfro_kernel, dim3(GW_PARTS), dim3(256), 0, st, G, (int)n, d2, stats));
    TN_CHECK_LAUNCH("gram_kfro_kernel");
    return 0;
}

// out[r] = sum_c A[r, c]^2  (rows of a row-major rows x cols matrix), one workgroup per row
__global__ __launch_bounds__(256) void rows_norm2_kernel(const double* __restrict__ A, int64_t cols, double* __restrict__ out) {
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const double* a = A + (int64_t)blockIdx.x * cols;
    double s0 = 0.0, s1 = 0.0;
    int64_t c = tid;
    for (; c + 256 < cols; c += 512) { const double x = a[c], y = a[c + 256]; s0 += x * x; s1 += y * y; }
    for (; c < cols; c += 256) { const double x = a[c]; s0 += x * x; }
    double s = s0 + s1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
int rows_norm2(hipStream_t st, const double* A, int64_t rows, int64_t cols, double* out) {
    if (rows <= 0) return 0;
    TN_CHECK_ARG(cols >= 1 && rows < 2147483647LL, "bad dimensions");
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(rows_norm2_kernel, dim3((unsigned)rows), dim3(256), 0, st, A, cols, out));
    TN_CHECK_LAUNCH("rows_norm2_kernel");
    return 0;
}

// out[j, :] = sqrt(w2[perm[j]]) * A[perm[j], :]      (gather + scale of the rows of a row-major rows x cols matrix)
__global__ __launch_bounds__(256) void gather_scale_rows_kernel(const double* __restrict__ A, int64_t cols, const int64_t* __restrict__ perm,
                                                                const double* __restrict__ w2, double* __restrict__ out) {
    const int64_t src = perm[blockIdx.x];
    const double sc = sqrt(w2[src]);
    const double* a = A + src * cols;
    double* o = out + (int64_t)blockIdx.x * cols;
    for (int64_t c = threadIdx.x; c < cols; c += 256) o[c] = sc * a[c];
}
// out[perm[j], :] = Cp[j, :] / sqrt(w2[perm[j]])     (inverse of the above on the small triangular factor, rows x cols row-major)
__global__ __launch_bounds__(256) void scatter_unscale_rows_kernel(const double* __restrict__ Cp, int64_t cols, const int64_t* __restrict__ perm,
                                                                   const double* __restrict__ w2, double* __restrict__ out) {
    const int64_t dst = perm[blockIdx.x];
    const double inv = 1.0 / sqrt(w2[dst]);
    const double* a = Cp + (int64_t)blockIdx.x * cols;
    double* o = out + dst * cols;
    for (int64_t c = threadIdx.x; c < cols; c += 256) o[c] = a[c] * inv;
}
int gather_scale_rows(hipStream_t st, const double* A, int64_t rows, int64_t cols, const int64_t* perm, const double* w2, double* out,
                      int inverse) {
    if (rows <= 0 || cols <= 0) return 0;
    TN_CHECK_ARG(rows < 2147483647LL, "too many rows");
    if (inverse)
        TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(scatter_unscale_rows_kernel, dim3((unsigned)rows), dim3(256), 0, st, A, cols, perm, w2, out));
    else
        TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(gather_scale_rows_kernel, dim3((unsigned)rows), dim3(256), 0, st, A, cols, perm, w2, out));
    TN_CHECK_LAUNCH("gather_scale_rows_kernel");
    return 0;
}

// ---- deflation of a bond whose centre matrix cannot be truncated (chain.hip: gauge_svd_skippable) ------------------------------
// An intermediate pass of graduate_truncation meets bonds at which mps.py:805-806 cannot truncate (min(C.shape) <= Dmax, tol <= eps): its
// SVD would only remove singular values below eps S0.  The same noise is removed here without a decomposition.  The centre matrix sits
// between an orthonormal site and a canonical rest, so the state's norm is ||C||_F and zeroing bond index i changes the state by exactly
// the norm of row (side 0) / column (side 1) i of C.  The indices are dropped in ascending order of that norm for as long as the dropped
// squares add up to at most eps^2 max_i ||C_i||^2 <= (eps S0)^2 -- no more than ONE singular value at the reference's threshold would
// carry -- and C and the neighbouring site are gathered to the kept indices (kept in their order).
//   side 0 (left sweep):  C (k x n) row-major, bond = rows;     Q (m x k) row-major, bond = columns  -> C_out (k' x n), Q_out (m x k')
//   side 1 (right sweep): C (n x k) row-major, bond = columns;  Q (k x m) row-major, bond = rows     -> C_out (n x k'), Q_out (k' x m)
// k <= 256.  *k_out = k' (host); nothing is written when k' == k.  ws: k doubles.  One read-back.
struct KeepList { int idx[256]; };
__global__ __launch_bounds__(256) void bond_norm2_kernel(const double* __restrict__ C, int64_t vs, int64_t es, int64_t len, double* __restrict__ out) {
    __shared__ double red[256];
    const int tid = threadIdx.x;
    const double* c = C + (int64_t)blockIdx.x * vs;
    double s = 0.0;
    for (int64_t j = tid; j < len; j += 256) { const double x = c[j * es]; s += x * x; }
    red[tid] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (tid < k) red[tid] += red[tid + k]; __syncthreads(); }
    if (tid == 0) out[blockIdx.x] = red[0];
}
// out[j, :] = in[idx[j], :]   (row-major, `cols` columns; one workgroup per kept row)
__global__ __launch_bounds__(256) void gather_rows_idx_kernel(const double* __restrict__ in, int64_t cols, KeepList kl, double* __restrict__ out) {
    const double* a = in + (int64_t)kl.idx[blockIdx.x] * cols;
    double* o = out + (int64_t)blockIdx.x * cols;
    for (int64_t c = threadIdx.x; c < cols; c += 256) o[c] = a[c];
}
// out[r, j] = in[r, idx[j]]   (row-major rows x kin -> rows x kout; the index list sits in LDS)
__global__ __launch_bounds__(256) void gather_cols_idx_kernel(const double* __restrict__ in, int64_t rows, int kin, int kout, KeepList kl,
                                                              double* __restrict__ out) {
    __shared__ int idx[256];
    if (threadIdx.x < kout) idx[threadIdx.x] = kl.idx[threadIdx.x];
    __syncthreads();
    const int64_t total = rows * kout;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / kout;
        const int j = (int)(e - r * kout);
        out[e] = in[r * kin + idx[j]];
    }
}
int bond_deflate(hipStream_t st, int side, const double* C, int64_t k, int64_t n, const double* Q, int64_t m, double* C_out, double* Q_out,
                 int64_t* k_out, double* dropped2_rel_out, void* ws, int64_t ws_bytes) {
    TN_CHECK_ARG(side == 0 || side == 1, "side must be 0 (left sweep) or 1 (right sweep)");
    TN_CHECK_ARG(C && Q && C_out && Q_out && k_out && ws, "null operand");
    TN_CHECK_ARG(k >= 1 && k <= 256 && n >= 1 && m >= 1, "bad dimensions (1 <= k <= 256)");
    TN_CHECK_ARG(ws_bytes >= k * 8, "workspace too small");
    double* dn = (double*)ws;
    if (side == 0) TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(bond_norm2_kernel, dim3((unsigned)k), dim3(256), 0, st, C, n, 1, n, dn));
    else TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(bond_norm2_kernel, dim3((unsigned)k), dim3(256), 0, st, C, 1, k, n, dn));
    TN_CHECK_LAUNCH("bond_norm2_kernel");
    double h[256];
    {
        hipError_t e;
        double* stage = (double*)pinned_host((size_t)k * 8, 2);
        if ((e = hipMemcpyAsync(stage ? stage : h, dn, (size_t)k * 8, hipMemcpyDeviceToHost, st)) != hipSuccess) return hip_fail(e, "memcpy norms");
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(e, "sync norms");
        if (stage) memcpy(h, stage, (size_t)k * 8);
    }
    double nmax = 0.0;
    for (int64_t i = 0; i < k; ++i) {
        if (!(h[i] == h[i]) || h[i] > 1.7e308) { set_error("bond_deflate: non-finite centre matrix"); return -2; }
        if (h[i] > nmax) nmax = h[i];
    }
    *k_out = k;
    if (dropped2_rel_out) *dropped2_rel_out = 0.0;
    if (!(nmax > 0.0) || k == 1) return 0;
    int order[256];
    for (int i = 0; i < (int)k; ++i) order[i] = i;
    // ascending norms, ties by index (insertion sort: k <= 256, and the list is nearly sorted for a triangular factor read backwards)
    for (int i = 1; i < (int)k; ++i) {
        const int v = order[i];
        int j = i - 1;
        while (j >= 0 && (h[order[j]] > h[v] || (h[order[j]] == h[v] && order[j] > v))) { order[j + 1] = order[j]; --j; }
        order[j + 1] = v;
    }
    const double eps = 2.220446049250313e-16;
    const double budget = eps * eps * nmax;
    bool drop[256];
    for (int i = 0; i < (int)k; ++i) drop[i] = false;
    double acc = 0.0;
    int ndrop = 0;
    for (int t = 0; t < (int)k - 1; ++t) {
        const double v = h[order[t]];
        if (!(acc + v <= budget)) break;
        acc += v;
        drop[order[t]] = true;
        ++ndrop;
    }
    if (ndrop == 0) return 0;
    KeepList kl;
    int kk = 0;
    for (int i = 0; i < (int)k; ++i) if (!drop[i]) kl.idx[kk++] = i;
    for (int i = kk; i < 256; ++i) kl.idx[i] = 0;
    const int64_t rowsA = side == 0 ? m : n;           // the operand gathered by columns: Q (m x k) resp. C (n x k)
    const double* colsrc = side == 0 ? Q : C;
    double* coldst = side == 0 ? Q_out : C_out;
    const double* rowsrc = side == 0 ? C : Q;          // the operand gathered by rows: C (k x n) resp. Q (k x m)
    double* rowdst = side == 0 ? C_out : Q_out;
    const int64_t rowlen = side == 0 ? n : m;
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(gather_rows_idx_kernel, dim3((unsigned)kk), dim3(256), 0, st, rowsrc, rowlen, kl, rowdst));
    TN_CHECK_LAUNCH("gather_rows_idx_kernel");
    const int64_t total = rowsA * kk;
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv(total, 256), 2048);
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(gather_cols_idx_kernel, dim3(grid), dim3(256), 0, st, colsrc, rowsA, (int)k, kk, kl, coldst));
    TN_CHECK_LAUNCH("gather_cols_idx_kernel");
    *k_out = kk;
    if (dropped2_rel_out) *dropped2_rel_out = acc / nmax;
    return 0;
}

}  // namespace tn
