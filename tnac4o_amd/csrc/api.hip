// extern "C" surface of libtnpeps (declared in include/tnpeps.h) + thread-local error text.
#include <stdarg.h>

#include "../../include/tnpeps.h"
#include "common.h"

namespace tn {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }

// implemented in the other translation units
int absorb(hipStream_t, const double*, const double*, double*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int, int64_t,
           int64_t, int64_t, int64_t);
int qr_factor(hipStream_t, double*, int64_t, int64_t, int64_t, int64_t, double*, int64_t, int64_t, double*, int64_t, int64_t, int,
              void*, int64_t, double, int64_t*, hipStream_t, double* dropped2_host = nullptr, int frob_exit = 0,
              int64_t* pivot_perm_host = nullptr, double* nf_out2 = nullptr,
              int* nf_done = nullptr);
int64_t qr_ws_bytes(int64_t, int64_t, int);
int svd_trunc(hipStream_t, const double*, int64_t, int64_t, int64_t, int64_t, int64_t, double, double*, int64_t, int64_t, double*,
              double*, int64_t, int64_t, int64_t*, double*, int*, int*, void*, int64_t);
int svd_vals(hipStream_t, const double*, int64_t, int64_t, int64_t, int64_t, double*, int*, int*, void*, int64_t);
int64_t svd_ws_bytes(int64_t, int64_t, int);
int svd_vals_small_async(hipStream_t, const double*, int64_t, int64_t, int64_t, int64_t, double*);
int svd_vals_small_batched(hipStream_t, const int64_t*, int64_t, double*);
int nfactor(hipStream_t, const double*, int64_t, double*, void*);
int scale_by(hipStream_t, double*, int64_t, const double*);
int normalize_pow2(hipStream_t, double*, int64_t, double*, void*, int64_t);
int scale_phys(hipStream_t, double*, int64_t, int64_t, int64_t, const double*, int);
int calc_pn(hipStream_t, const double*, const double*, const double*, const int32_t*, const int32_t*, const int32_t*,
            const int32_t*, const int32_t*, const int32_t*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, double*,
            double*, const double*, double*);
int merge_groups(hipStream_t, const double*, const double*, const int64_t*, const int64_t*, const int64_t*, int64_t, double, int64_t*, int64_t*,
                 double*);
int nfactor_batched(hipStream_t, double*, int64_t, int64_t);
int peps_factor(hipStream_t, const double*, const double*, const double*, const double*, const double*, const double*, const double*,
                const int32_t*, const int32_t*, int64_t, int64_t, int64_t, double*);
int mpo_from_factor(hipStream_t, const double*, const int32_t*, const int32_t*, int64_t, int64_t, int64_t, int64_t, int64_t, double*);
int env_rr_batched(hipStream_t, const double*, const double*, const double*, const int32_t*, const int32_t*, int64_t, int64_t, int64_t,
                   int64_t, int64_t, int64_t, int64_t, double*);
int env_rl_batched(hipStream_t, const double*, const int32_t*, const int32_t*, int64_t, int64_t, int64_t, double*);
int balance(hipStream_t, const double*, int64_t, int64_t, int64_t, double, double*, int*);
int64_t site_qr_ws_bytes(int, int64_t, int64_t, int64_t, int64_t, int);
int site_qr(hipStream_t, int, double*, int64_t, int64_t, int64_t, const double*, int64_t, double*, double*, double, int64_t*, double*, int*,
            void*, int64_t, double*, int, int64_t*);
int gram_weights(hipStream_t, const double*, int64_t, double, double*, double*);
int rows_norm2(hipStream_t, const double*, int64_t, int64_t, double*);
int bond_deflate(hipStream_t, int, const double*, int64_t, int64_t, const double*, int64_t, double*, double*, int64_t*, double*, void*, int64_t);
int gather_scale_rows(hipStream_t, const double*, int64_t, int64_t, const int64_t*, const double*, double*, int);
int64_t rar_ws_bytes(int64_t, int64_t, int64_t, int64_t, int64_t);
int rar(hipStream_t, const double*, const double*, const double*, int64_t, int64_t, int64_t, int64_t, int64_t, double*, void*, int64_t);
int64_t env_mix_ws_bytes(int, int64_t, int64_t, int64_t, int64_t, int64_t);
int env_mix(hipStream_t, int, const double*, const double*, const double*, int64_t, int64_t, int64_t, int64_t, int64_t, double*, void*,
            int64_t);
int64_t apply_truncation_ws_bytes(int64_t, int64_t, int64_t, int64_t, int64_t);
int apply_truncation(hipStream_t, const double*, int64_t, int64_t, const double*, int64_t, int64_t, int64_t, const double*, int64_t, int64_t,
                     const double*, int64_t, int64_t, const double*, double*, double*, double*, void*, int64_t);
int qr_batched(hipStream_t, double*, int64_t, int64_t, int64_t, int64_t, double*, int64_t, int64_t, double*, int64_t, int64_t, int, double,
               int64_t*, int64_t, int64_t, int64_t, int64_t, void*, int64_t, void* const*, int);
int svd_trunc_batched(hipStream_t, const double*, int64_t, int64_t, int64_t, int64_t, int64_t, double, double*, int64_t, int64_t, double*,
                      double*, int64_t, int64_t, int64_t*, double*, int*, int*, int64_t, int64_t, int64_t, int64_t, int64_t, void*, int64_t);
int svd_vals_batched(hipStream_t, const double*, int64_t, int64_t, int64_t, int64_t, double*, int*, int*, int64_t, int64_t, void*, int64_t);
int smallqr_stats(hipStream_t st, unsigned long long* out4, int reset);

}  // namespace tn

using namespace tn;
#define ST ((hipStream_t)stream)

extern "C" {

int tn_version(void) { return 9; }

#ifndef TN_SRC_HASH
#define TN_SRC_HASH "unknown"
#endif
int tn_build_id(char* buf, int n) {
    const char* e = TN_SRC_HASH;
    const int len = (int)strlen(e);
    if (buf && n > 0) {
        const int c = len < n - 1 ? len : n - 1;
        memcpy(buf, e, c);
        buf[c] = 0;
    }
    return len;
}

void tn_profile_enable(unsigned mask) { prof_set_mask(mask); }
void tn_profile_reset(void) { prof_reset(); }
void tn_profile_sample(unsigned every) { prof_set_sample(every); }
int tn_profile_get(int family, uint64_t* calls_host, double* ms_host, double* flops_host, double* bytes_host) {
    TN_CHECK_ARG(family >= 0 && family < PROF_NFAM, "unknown kernel family");
    TN_CHECK_ARG(calls_host && ms_host && flops_host && bytes_host, "null output");
    prof_get(-1, family, calls_host, ms_host, flops_host, bytes_host);
    return 0;
}
int tn_profile_get_phase(int phase, int family, uint64_t* calls_host, double* ms_host, double* flops_host, double* bytes_host) {
    TN_CHECK_ARG(phase >= -1 && phase < PH_N, "unknown phase");
    TN_CHECK_ARG(family >= 0 && family < PROF_NFAM, "unknown kernel family");
    TN_CHECK_ARG(calls_host && ms_host && flops_host && bytes_host, "null output");
    prof_get(phase, family, calls_host, ms_host, flops_host, bytes_host);
    return 0;
}

// A HIP stream restricted to the compute units whose bits are set in `mask` (nwords x 32 bits; hipExtStreamCreateWithCUMask).  Used by
// parallel.run_concurrent (TN_CU_MASK=1) to keep a share of the CUs out of every chain's reach, so that a device-filling GEMM of one
// chain cannot occupy every CU the latency-bound kernels of the other chains could use.  The caller destroys it with tn_stream_destroy.
int tn_stream_create_masked(const uint32_t* mask_host, int nwords, void** stream_out) {
    TN_CHECK_ARG(mask_host && nwords >= 1 && stream_out, "bad arguments");
    hipStream_t st = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)nwords, mask_host);
    if (e != hipSuccess) return hip_fail(e, "hipExtStreamCreateWithCUMask");
    fused_forms_disable(st);        // launches with in-kernel barriers count on whole-chip co-residency: not on a masked stream
    *stream_out = (void*)st;
    return 0;
}
int tn_stream_destroy(void* stream) {
    fused_stream_released((hipStream_t)stream);
    const hipError_t e = hipStreamDestroy((hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "hipStreamDestroy");
}

int tn_last_error(char* buf, int n) {
    const char* e = get_error();
    int len = (int)strlen(e);
    if (buf && n > 0) {
        int c = len < n - 1 ? len : n - 1;
        memcpy(buf, e, c);
        buf[c] = 0;
    }
    return len;
}

int tn_gemm(int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t rsa, int64_t csa, const double* B,
            int64_t rsb, int64_t csb, double beta, double* C, int64_t rsc, int64_t csc, int64_t batch, int64_t bsa,
            int64_t bsb, int64_t bsc, void* ws, int64_t ws_bytes, void* stream) {
    TN_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && batch >= 0, "negative dimension");
    TN_CHECK_ARG(A && B && C, "null operand");
    return gemm(ST, M, N, K, alpha, A, rsa, csa, B, rsb, csb, beta, C, rsc, csc, batch, bsa, bsb, bsc, (double*)ws, ws_bytes);
}
int64_t tn_gemm_ws_bytes(int64_t M, int64_t N, int64_t K, int64_t batch) { return gemm_ws_bytes(M, N, K, batch); }

int tn_absorb(const double* A, const double* W, double* out, int64_t Dl, int64_t pold, int64_t Dr, int64_t ba, int64_t po,
              int64_t bb, int64_t pi, int hconj, int64_t batch, int64_t bsA, int64_t bsW, int64_t bsOut, void* stream) {
    TN_CHECK_ARG(batch >= 0, "negative batch");
    TN_CHECK_ARG(batch == 0 || (A && W && out), "null operand");
    ProfPhase ph(PH_ABSORB);
    return absorb(ST, A, W, out, Dl, pold, Dr, ba, po, bb, pi, hconj, batch, bsA, bsW, bsOut);
}

int tn_qr(double* A, int64_t rs, int64_t cs, int64_t m, int64_t n, double* Q, int64_t qrs, int64_t qcs, double* R, int64_t rrs,
          int64_t rcs, int nb, double rank_tol, int64_t* keff_host, void* ws, int64_t ws_bytes, void* stream, void* aux_stream) {
    TN_CHECK_ARG(A && Q && R && ws, "null operand");
    TN_CHECK_ARG(rank_tol >= 0.0 && rank_tol < 1.0, "rank_tol out of range");
    ProfPhase ph(PH_QR);
    const double dm = (double)m, dn = (double)(n < m ? n : m);
    prof_note(PROF_QR_NOMINAL, 1, 4.0 * dm * dn * dn - 4.0 / 3.0 * dn * dn * dn, 8.0 * (2.0 * dm * dn + dn * dn));
    return qr_factor(ST, A, rs, cs, m, n, Q, qrs, qcs, R, rrs, rcs, nb, ws, ws_bytes, rank_tol, keff_host, (hipStream_t)aux_stream);
}
int64_t tn_qr_ws_bytes(int64_t m, int64_t n, int nb) { return qr_ws_bytes(m, n, nb); }

int tn_smallqr_stats(uint64_t* out4_host, int reset, void* stream) {
    TN_CHECK_ARG(out4_host, "null output");
    unsigned long long o[4];
    const int rc = smallqr_stats(ST, o, reset);
    for (int i = 0; i < 4; ++i) out4_host[i] = o[i];
    return rc;
}
int tn_fused_timeouts(int* count_host, void* stream) {
    TN_CHECK_ARG(count_host, "null output");
    return fused_timeouts(ST, count_host);
}

int64_t tn_panel_orth_ws_bytes(int64_t nrows, int b) {
    const int64_t a = tsqr_ws_bytes(nrows, b), c = cholqr_ws_bytes(nrows, b);
    return a > c ? a : c;
}
int tn_panel_orth(const double* X, int64_t rs, int64_t cs, int64_t nrows, int b, double* Y, int64_t yrs, int64_t ycs, int method,
                  int* state9_host, double* dev_host, void* ws, int64_t ws_bytes, void* stream) {
    TN_CHECK_ARG(X && Y && ws, "null operand");
    TN_CHECK_ARG(method == 0 || method == 1, "method must be 0 (Cholesky-QR) or 1 (Householder TSQR)");
    TN_CHECK_ARG(ws_bytes >= tn_panel_orth_ws_bytes(nrows, b), "workspace too small");
    ProfPhase ph(PH_QR);
    if (method == 1) return tsqr_orthonormalize(ST, X, rs, cs, Y, yrs, ycs, nrows, b, ws, ws_bytes);
    int rc = cholqr_reset(ST, ws);
    if (rc) return rc;
    int fbase = 0;
    if ((rc = cholqr_orthonormalize(ST, X, rs, cs, Y, yrs, ycs, nrows, b, ws, ws_bytes, 0x5DEECE66DULL, &fbase))) return rc;
    if (fused_check_needed() && !fused_check_deferred()) {      // the single-launch form may have given up at a barrier: never hand NaN back
        int gave_up = 0;
        if ((rc = fused_timeouts(ST, &gave_up))) return rc;
        if (gave_up > 0) {                                      // X is untouched: once more, through the six-launch chain the stream now takes
            if ((rc = cholqr_reset(ST, ws))) return rc;
            fbase = 0;
            if ((rc = cholqr_orthonormalize(ST, X, rs, cs, Y, yrs, ycs, nrows, b, ws, ws_bytes, 0x5DEECE66DULL, &fbase))) return rc;
        }
    }
    if (state9_host && dev_host) return cholqr_debug_state(ST, ws, state9_host, dev_host);
    return 0;
}
int tn_panel_stats(uint64_t* out16_host, int reset) {
    TN_CHECK_ARG(out16_host, "null output");
    return cholqr_stats((unsigned long long*)out16_host, reset, nullptr, 1);
}
int tn_panel_stats_stream(uint64_t* out16_host, int reset, void* stream) {
    TN_CHECK_ARG(out16_host, "null output");
    return cholqr_stats((unsigned long long*)out16_host, reset, (hipStream_t)stream, 0);
}

int tn_svd_trunc(const double* C, int64_t crs, int64_t ccs, int64_t k, int64_t n, int64_t Dmax, double tol, double* U,
                 int64_t urs, int64_t ucs, double* S, double* Vt, int64_t vrs, int64_t vcs, int64_t* keep_host,
                 double* discarded_host, int* sweeps_host, int* info_host, void* ws, int64_t ws_bytes, void* stream) {
    TN_CHECK_ARG(C && U && S && Vt && ws && keep_host, "null operand");
    ProfPhase ph(PH_SVD);
    int sweeps = 0;
    const int rc = svd_trunc(ST, C, crs, ccs, k, n, Dmax, tol, U, urs, ucs, S, Vt, vrs, vcs, keep_host, discarded_host, &sweeps,
                             info_host, ws, ws_bytes);
    if (sweeps_host) *sweeps_host = sweeps;
    // nominal counts of SURVEY.md 8(d) with m = max(k, n), n = min(k, n)
    const double dm = (double)(k > n ? k : n), dn = (double)(k > n ? n : k);
    prof_note(PROF_SVD_NOMINAL, 1, 14.0 * dm * dn * dn + 8.0 * dn * dn * dn, 8.0 * (2.0 * dm * dn + dn * dn + dn));
    prof_note(PROF_SVD_STREAM, sweeps, 0.0, (double)sweeps * (dn - 1.0) * 16.0 * dn * (dm + dn));
    return rc;
}
int tn_svdvals(const double* C, int64_t crs, int64_t ccs, int64_t k, int64_t n, double* S_host, int* sweeps_host,
               int* info_host, void* ws, int64_t ws_bytes, void* stream) {
    TN_CHECK_ARG(C && S_host && ws, "null operand");
    ProfPhase ph(PH_SVDVALS);
    const double dm = (double)(k > n ? k : n), dn = (double)(k > n ? n : k);
    prof_note(PROF_SVDVALS_NOMINAL, 1, 4.0 * dm * dn * dn - 4.0 / 3.0 * dn * dn * dn, 8.0 * (dm * dn + dn));
    return svd_vals(ST, C, crs, ccs, k, n, S_host, sweeps_host, info_host, ws, ws_bytes);
}
int64_t tn_svd_ws_bytes(int64_t k, int64_t n, int vectors) { return svd_ws_bytes(k, n, vectors); }
int tn_svdvals_async(const double* C, int64_t crs, int64_t ccs, int64_t k, int64_t n, double* out66_dev, void* stream) {
    TN_CHECK_ARG(C && out66_dev, "null operand");
    ProfPhase ph(PH_SVDVALS);
    const double dm = (double)(k > n ? k : n), dn = (double)(k > n ? n : k);
    prof_note(PROF_SVDVALS_NOMINAL, 1, 4.0 * dm * dn * dn - 4.0 / 3.0 * dn * dn * dn, 8.0 * (dm * dn + dn));
    return svd_vals_small_async(ST, C, crs, ccs, k, n, out66_dev);
}

int tn_svdvals_small_batched(const int64_t* desc_dev, int64_t batch, const int64_t* desc_host, double* out66_dev, void* stream) {
    TN_CHECK_ARG(batch >= 0, "negative batch");
    TN_CHECK_ARG(batch == 0 || (desc_dev && desc_host && out66_dev), "null operand");
    for (int64_t i = 0; i < batch; ++i) {            // the host copy of the descriptors is only validated
        const int64_t* d = desc_host + 5 * i;
        TN_CHECK_ARG(d[0] != 0, "null matrix");
        TN_CHECK_ARG(d[3] >= 1 && d[4] >= 1 && d[3] <= d[4] && d[4] <= 64, "both dimensions must be in 1..64 (vectors <= length)");
    }
    ProfPhase ph(PH_SVDVALS);
    return svd_vals_small_batched(ST, desc_dev, batch, out66_dev);
}

int tn_nfactor(const double* x, int64_t n, double* out2, void* slot8, void* stream) {
    TN_CHECK_ARG(x && out2 && slot8, "null operand");
    return nfactor(ST, x, n, out2, slot8);
}
int tn_normalize_pow2(double* x, int64_t n, double* out2, void* scratch, int64_t scratch_bytes, void* stream) {
    TN_CHECK_ARG(x && out2 && scratch, "null operand");
    return normalize_pow2(ST, x, n, out2, scratch, scratch_bytes);
}
int tn_scale_by(double* x, int64_t n, const double* scalar_dev, void* stream) {
    TN_CHECK_ARG(x && scalar_dev, "null operand");
    return scale_by(ST, x, n, scalar_dev);
}
int tn_scale_phys(double* A, int64_t Dl, int64_t p, int64_t Dr, const double* diag, int inv, void* stream) {
    TN_CHECK_ARG(A && diag, "null operand");
    return scale_phys(ST, A, Dl, p, Dr, diag, inv);
}

int tn_calc_pn(const double* T1, const double* RR, const double* F, const int32_t* dmap, const int32_t* rmap,
               const int32_t* pref, const int32_t* suf, const int32_t* lidx, const int32_t* uidx, int64_t nb, int64_t q,
               int64_t nl, int64_t nu, int64_t p, int64_t Dr, int64_t br, double* P, double* minP, const double* parent_log2p,
               double* log2p_out, void* stream) {
    TN_CHECK_ARG(T1 && RR && F && dmap && rmap && pref && suf && lidx && uidx && P && minP, "null operand");
    return calc_pn(ST, T1, RR, F, dmap, rmap, pref, suf, lidx, uidx, nb, q, nl, nu, p, Dr, br, P, minP, parent_log2p, log2p_out);
}
int tn_merge_groups(const double* E, const double* log2p, const int64_t* deg, const int64_t* pos, const int64_t* starts, int64_t ngroups,
                    double min_dEng, int64_t* rep_pos_out, int64_t* deg_out, double* log2p_out, void* stream) {
    TN_CHECK_ARG(ngroups >= 0, "negative group count");
    TN_CHECK_ARG(ngroups == 0 || (E && log2p && deg && pos && starts && rep_pos_out && deg_out && log2p_out), "null operand");
    return merge_groups(ST, E, log2p, deg, pos, starts, ngroups, min_dEng, rep_pos_out, deg_out, log2p_out);
}
int tn_peps_factor(const double* Es, const double* E1, const double* E4, const double* Xu, const double* Xl, const double* Xr,
                   const double* Xd, const int32_t* dmap, const int32_t* rmap, int64_t q, int64_t nl, int64_t nu, double* F,
                   void* stream) {
    TN_CHECK_ARG(Es && E1 && E4 && Xu && Xl && Xr && Xd && dmap && rmap && F, "null operand");
    ProfPhase ph(PH_BUILD);
    return peps_factor(ST, Es, E1, E4, Xu, Xl, Xr, Xd, dmap, rmap, q, nl, nu, F);
}
int tn_mpo_from_factor(const double* F, const int32_t* dmap, const int32_t* rmap, int64_t q, int64_t nl, int64_t nu, int64_t pd,
                       int64_t br, double* W, void* stream) {
    TN_CHECK_ARG(F && dmap && rmap && W, "null operand");
    ProfPhase ph(PH_BUILD);
    return mpo_from_factor(ST, F, dmap, rmap, q, nl, nu, pd, br, W);
}
int tn_nfactor_batched(double* x, int64_t batch, int64_t len, void* stream) {
    TN_CHECK_ARG(x, "null operand");
    return nfactor_batched(ST, x, batch, len);
}

int tn_env_rr_batched(const double* A, const double* RRprev, const double* W, const int32_t* parent, const int32_t* uidx,
                      int64_t nk, int64_t Dl, int64_t p, int64_t Dr, int64_t bl, int64_t br, int64_t pu, double* out,
                      void* stream) {
    TN_CHECK_ARG(nk >= 0, "negative key count");
    TN_CHECK_ARG(nk == 0 || (A && RRprev && W && parent && uidx && out), "null operand");
    return env_rr_batched(ST, A, RRprev, W, parent, uidx, nk, Dl, p, Dr, bl, br, pu, out);
}
int tn_env_rl_batched(const double* T1, const int32_t* par, const int32_t* didx, int64_t nk, int64_t p, int64_t Dr,
                      double* out, void* stream) {
    TN_CHECK_ARG(nk >= 0, "negative key count");
    TN_CHECK_ARG(nk == 0 || (T1 && par && didx && out), "null operand");
    return env_rl_batched(ST, T1, par, didx, nk, p, Dr, out);
}
int tn_balance(const double* A, int64_t rs, int64_t cs, int64_t n, double max_scale, double* scale_out, int* iters_out,
               void* stream) {
    TN_CHECK_ARG(A && scale_out, "null operand");
    return balance(ST, A, rs, cs, n, max_scale, scale_out, iters_out);
}

int tn_qr_batched(double* A, int64_t rs, int64_t cs, int64_t m, int64_t n, double* Q, int64_t qrs, int64_t qcs, double* R, int64_t rrs,
                  int64_t rcs, int nb, double rank_tol, int64_t* keff_host, int64_t batch, int64_t bsA, int64_t bsQ, int64_t bsR, void* ws,
                  int64_t ws_bytes, void* stream, void* const* side_streams, int nside) {
    TN_CHECK_ARG(batch >= 0, "negative batch");
    TN_CHECK_ARG(batch == 0 || (A && Q && R && ws), "null operand");
    TN_CHECK_ARG(rank_tol >= 0.0 && rank_tol < 1.0, "rank_tol out of range");
    ProfPhase ph(PH_QR);
    const double dm = (double)m, dn = (double)(n < m ? n : m);
    prof_note(PROF_QR_NOMINAL, (double)batch, batch * (4.0 * dm * dn * dn - 4.0 / 3.0 * dn * dn * dn), batch * 8.0 * (2.0 * dm * dn + dn * dn));
    return qr_batched(ST, A, rs, cs, m, n, Q, qrs, qcs, R, rrs, rcs, nb, rank_tol, keff_host, batch, bsA, bsQ, bsR, ws, ws_bytes,
                      side_streams, nside);
}
int tn_svd_trunc_batched(const double* C, int64_t crs, int64_t ccs, int64_t k, int64_t n, int64_t Dmax, double tol, double* U, int64_t urs,
                         int64_t ucs, double* S, double* Vt, int64_t vrs, int64_t vcs, int64_t* keep_host, double* discarded_host,
                         int* sweeps_host, int* info_host, int64_t batch, int64_t bsC, int64_t bsU, int64_t bsS, int64_t bsV, void* ws,
                         int64_t ws_bytes, void* stream) {
    TN_CHECK_ARG(batch >= 0, "negative batch");
    TN_CHECK_ARG(batch == 0 || (C && U && S && Vt && ws && keep_host), "null operand");
    ProfPhase ph(PH_SVD);
    return svd_trunc_batched(ST, C, crs, ccs, k, n, Dmax, tol, U, urs, ucs, S, Vt, vrs, vcs, keep_host, discarded_host, sweeps_host,
                             info_host, batch, bsC, bsU, bsS, bsV, ws, ws_bytes);
}
int tn_svdvals_batched(const double* C, int64_t crs, int64_t ccs, int64_t k, int64_t n, double* S_host, int* sweeps_host, int* info_host,
                       int64_t batch, int64_t bsC, void* ws, int64_t ws_bytes, void* stream) {
    TN_CHECK_ARG(batch >= 0, "negative batch");
    TN_CHECK_ARG(batch == 0 || (C && S_host && ws), "null operand");
    ProfPhase ph(PH_SVDVALS);
    return svd_vals_batched(ST, C, crs, ccs, k, n, S_host, sweeps_host, info_host, batch, bsC, ws, ws_bytes);
}

int64_t tn_site_qr_ws_bytes(int side, int64_t Dl, int64_t p, int64_t Dr, int64_t kc, int attach) {
    return site_qr_ws_bytes(side, Dl, p, Dr, kc, attach);
}
int tn_site_qr(int side, double* A, int64_t Dl, int64_t p, int64_t Dr, const double* C, int64_t kc, double* Q, double* R, double rank_tol,
               int64_t* keff_host, double* nf_out2, int* normalised_host, double* dropped2_host, int frobenius_exit,
               int64_t* pivot_perm_host, void* ws, int64_t ws_bytes, void* stream) {
    TN_CHECK_ARG(A && Q && R && ws, "null operand");
    TN_CHECK_ARG(rank_tol >= 0.0 && rank_tol < 1.0, "rank_tol out of range");
    return site_qr(ST, side, A, Dl, p, Dr, C, kc, Q, R, rank_tol, keff_host, nf_out2, normalised_host, ws, ws_bytes, dropped2_host,
                   frobenius_exit, pivot_perm_host);
}
int tn_gram_weights(const double* G, int64_t n, double floor_rel, double* d2_out, double* stats65_out, void* stream) {
    TN_CHECK_ARG(G && d2_out && stats65_out, "null operand");
    return gram_weights(ST, G, n, floor_rel, d2_out, stats65_out);
}
int tn_rows_norm2(const double* A, int64_t rows, int64_t cols, double* out, void* stream) {
    TN_CHECK_ARG(rows == 0 || (A && out), "null operand");
    return rows_norm2(ST, A, rows, cols, out);
}
int tn_bond_deflate(int side, const double* C, int64_t k, int64_t n, const double* Q, int64_t m, double* C_out, double* Q_out, int64_t* k_out_host,
                    double* dropped2_rel_host, void* ws, int64_t ws_bytes, void* stream) {
    return bond_deflate(ST, side, C, k, n, Q, m, C_out, Q_out, k_out_host, dropped2_rel_host, ws, ws_bytes);
}
int tn_gather_scale_rows(const double* A, int64_t rows, int64_t cols, const int64_t* perm, const double* w2, double* out, int inverse,
                         void* stream) {
    TN_CHECK_ARG(rows == 0 || (A && perm && w2 && out), "null operand");
    return gather_scale_rows(ST, A, rows, cols, perm, w2, out, inverse);
}
int64_t tn_rar_ws_bytes(int64_t c, int64_t a, int64_t s, int64_t a2, int64_t c2) { return rar_ws_bytes(c, a, s, a2, c2); }
int tn_rar(const double* RL, const double* A, const double* RR, int64_t c, int64_t a, int64_t s, int64_t a2, int64_t c2, double* out,
           void* ws, int64_t ws_bytes, void* stream) {
    TN_CHECK_ARG(RL && A && RR && out && ws, "null operand");
    return rar(ST, RL, A, RR, c, a, s, a2, c2, out, ws, ws_bytes);
}
int64_t tn_env_mix_ws_bytes(int side, int64_t a, int64_t s, int64_t a2, int64_t c, int64_t c2) {
    return env_mix_ws_bytes(side, a, s, a2, c, c2);
}
int tn_env_mix(int side, const double* R, const double* A, const double* Ac, int64_t a, int64_t s, int64_t a2, int64_t c, int64_t c2,
               double* out, void* ws, int64_t ws_bytes, void* stream) {
    TN_CHECK_ARG(R && A && Ac && out && ws, "null operand");
    return env_mix(ST, side, R, A, Ac, a, s, a2, c, c2, out, ws, ws_bytes);
}
int64_t tn_apply_truncation_ws_bytes(int64_t ml, int64_t k0, int64_t keep, int64_t k1, int64_t nr) {
    return apply_truncation_ws_bytes(ml, k0, keep, k1, nr);
}
int tn_apply_truncation(const double* Al, int64_t ml, int64_t k0, const double* U, int64_t urs, int64_t ucs, int64_t keep, const double* Vt,
                        int64_t vrs, int64_t vcs, const double* Ar, int64_t k1, int64_t nr, const double* S, double* Al_new, double* Ar_new,
                        double* Cdiag, void* ws, int64_t ws_bytes, void* stream) {
    TN_CHECK_ARG(Al && U && Vt && Ar && S && Al_new && Ar_new && Cdiag, "null operand");
    return apply_truncation(ST, Al, ml, k0, U, urs, ucs, keep, Vt, vrs, vcs, Ar, k1, nr, S, Al_new, Ar_new, Cdiag, ws, ws_bytes);
}

}  // extern "C"
