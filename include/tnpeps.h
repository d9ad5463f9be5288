/* libtnpeps — C-ABI of the MI355X boundary-MPS PEPS contraction kernels.
 *
 * The reference (marekrams/tnac4o) has no FFI layer; its hot path is the module surface of tnac4o/mps.py as
 * consumed by tnac4o/tnac4o.py.  Each entry point below names the reference code it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to float64 data unless its name ends in `_host`;
 *   - matrices are passed with explicit element strides (rs = row stride, cs = column stride), so transposed
 *     or sliced views never need a copy;
 *   - the caller owns every buffer including workspaces (sizes from the *_ws_bytes queries); nothing is
 *     allocated or freed by the library; outputs never alias inputs unless stated;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it.  tn_svd_trunc / tn_svdvals
 *     synchronise the stream internally (rank decisions are made on the host), the rest is asynchronous;
 *   - return 0 = ok, < 0 = argument error (nothing launched; -3 = a caller-sized buffer was too small, -7 = a launch with
 *     in-kernel barriers gave up and the inputs were consumed: rerun from a copy), > 0 = HIP runtime error code.  The
 *     message is available from tn_last_error() (thread-local).  Numerical events (non-convergence) are reported
 *     through `info` outputs, never as errors.
 *
 * State the library keeps (SURVEY.md 8b asks for none; what there is, is bookkeeping -- no result depends on it being there):
 *   - per host thread: the error text; a few page-locked staging buffers for small read-backs / uploads (grown on demand,
 *     freed when the thread exits); the event ring of the optional profiler (tn_profile_*); the stamp counter of the
 *     device-side panel pivoting (tn_qr / tn_site_qr with pivot_perm_host);
 *   - per (device, stream), created at the stream's first use and released by tn_stream_destroy (streams the caller
 *     destroys otherwise keep their slot: at most 64 slots, the streams beyond share the last one for statistics and do not
 *     take the single-launch forms): a slot number; a 256-byte __device__ state block for the panel step, one for the
 *     one-launch factorisation and one for the one-launch Jacobi rounds (barrier counters, cleared by the launch that ends
 *     a call); DIAGNOSTIC counters (tn_panel_stats*, tn_smallqr_stats); the STICKY count of launches that gave up at an
 *     in-kernel barrier (read by tn_fused_timeouts, never reset by the statistics calls); a flag "this stream is off the
 *     single-launch forms" set after such a time-out (from then on the multi-launch forms run: same results bit for bit);
 *     the event of a tall panel launch in flight (admission control of launches that need co-resident workgroups);
 *   - process-wide, read once: the co-residency budget (CUs of the device, GPU_MAX_HW_QUEUES, TN_PANEL_CU_BUDGET) and the
 *     switches of INTEGRATION.md section 3 (the ones marked "per call" are read at every call).
 * Thread-safety: every entry point may be called concurrently from different host threads on DIFFERENT streams (the
 * product drives one chain per thread and stream); the tables above are guarded by mutexes.  Two threads must not
 * enqueue on the SAME stream at once (workspaces and state blocks are per stream).  The library allocates no device
 * memory; the __device__ state pools are static.
 */
#ifndef TNPEPS_H
#define TNPEPS_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* version of this C-ABI (bumped whenever a signature below changes; the binding checks it at load time) */
int tn_version(void);
/* copies the hash of the sources the library was built from (set by the build recipe) into buf; returns its length */
int tn_build_id(char* buf, int n);
/* A stream confined to the compute units set in mask_host (nwords x 32 bits); see tnac4o_amd/parallel.py (TN_CU_MASK). */
int tn_stream_create_masked(const uint32_t* mask_host, int nwords, void** stream_out);
int tn_stream_destroy(void* stream);
/* copies the calling thread's last error text into buf (NUL-terminated); returns its length */
int tn_last_error(char* buf, int n);

/* ---- K2: strided (batched) GEMM.  C = alpha * A[M,K] * B[K,N] + beta * C.
 * Replaces every np.tensordot of mps.py (_mps_CA/_mps_AC :740-746, _mps_RL/_RR :655-663, _mps_RAR :748-751,
 * projector application :579-580, bond_env/expectation :694-698, :765-769) and tnac4o.py:532, 1779-1780, 1792-1794.
 * ws may be NULL (disables split-K). */
int tn_gemm(int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t rsa, int64_t csa,
            const double* B, int64_t rsb, int64_t csb, double beta, double* C, int64_t rsc, int64_t csc,
            int64_t batch, int64_t bsa, int64_t bsb, int64_t bsc, void* ws, int64_t ws_bytes, void* stream);
int64_t tn_gemm_ws_bytes(int64_t M, int64_t N, int64_t K, int64_t batch);

/* ---- K1: MPO.MPS absorption of one site.  Replaces MPS.apply_mpo (mps.py:353-359) -> _mps_HA (:753-763).
 * A: (Dl, pold, Dr) C-order.  W: (ba, po, bb, pi) C-order, legs (left, out, right, in).
 * hconj=1: out (Dl*ba, pi, Dr*bb), MPS index major;  hconj=0: out (ba*Dl, po, bb*Dr), MPO index major.
 * batch >= 1 equally shaped sites in one launch (item i at A + i*bsA, W + i*bsW, out + i*bsOut; a stride of 0 shares the
 * operand between the items, e.g. one MPO site absorbed into the same site of several boundary MPS). */
int tn_absorb(const double* A, const double* W, double* out, int64_t Dl, int64_t pold, int64_t Dr, int64_t ba,
              int64_t po, int64_t bb, int64_t pi, int hconj, int64_t batch, int64_t bsA, int64_t bsW, int64_t bsOut,
              void* stream);

/* ---- K3: economic QR, diag(R) >= 0.  Replaces mps.qr (mps.py:43-59) as used by _mps_decompose_AC/CA (:772-800).
 * A (m x n) is DESTROYED.  Q: m x min(m,n), R: min(m,n) x n.  nb in {32, 64} is the panel width.
 * rank_tol = 0: the plain factorisation (asynchronous).  rank_tol > 0 (nb = 32, keff_host != NULL): early exit for the
 * truncating canonisation passes — every second panel the largest column norm of the unfactored trailing block is read
 * back; once it is <= rank_tol x the largest column norm of A the factorisation stops and *keff_host (HOST) receives
 * the number of columns of Q / rows of R produced (A = Q[:, :keff] R[:keff, :] to rank_tol * max column norm).  With
 * rank_tol = 2^-56 this drops exactly the rows the Jacobi SVD (tn_svd_trunc) would deflate.  Synchronises the stream
 * at each check.  *keff_host = min(m, n) otherwise.
 * aux_stream (a second hipStream_t owned by the caller, or NULL): look-ahead.  The device-filling part of every trailing update
 * is enqueued on aux_stream while `stream` goes on factoring the next panel (a chain of latency-bound single-workgroup kernels);
 * the two are ordered by events inside the call, and on return all work the result depends on is ordered before anything
 * enqueued on `stream` afterwards.  Results are bit-identical with and without it (same kernels on the same data). */
int tn_qr(double* A, int64_t rs, int64_t cs, int64_t m, int64_t n, double* Q, int64_t qrs, int64_t qcs, double* R,
          int64_t rrs, int64_t rcs, int nb, double rank_tol, int64_t* keff_host, void* ws, int64_t ws_bytes, void* stream,
          void* aux_stream);
int64_t tn_qr_ws_bytes(int64_t m, int64_t n, int nb);
/* The panel step of tn_qr on its own (test and measurement hook): Y (nrows x b, b <= 32) <- an orthonormal basis of the column
 * space of the panel X (read only, must not overlap Y; completed arbitrarily where X is rank deficient).
 * method 0: iterated Cholesky-QR with deferral (csrc/cholqr.hip, what tn_qr uses); 1: Householder TSQR (csrc/tsqr.hip).
 * state9_host / dev_host (both or neither; method 0; synchronises the stream): {ticket counter, done, final_next, passes applied,
 * input exponent, refill mask, deferred pivots, refilled columns, Householder fallback taken} and max|X^T X - I| before each pass.
 * tn_panel_stats_stream: DIAGNOSTIC counters of the panels launched on `stream` since its last reset (the library keeps them per
 * stream, so that concurrent chains do not mix their counts; streams beyond the 64th share one slot); tn_panel_stats: their sum
 * over all streams (reset clears all).  16 words: {panels, substitution passes applied, deferred pivots,
 * refilled columns, Householder fallbacks, panels with >= 3 passes, panels with >= 4 passes, panel elements x passes applied by the
 * six-launch chain, the same for the single-launch form, panels handled by the single-launch form, 0...}.  They feed bench.py's
 * accounting of the work the passes really did and nothing else; no result depends on them. */
int64_t tn_panel_orth_ws_bytes(int64_t nrows, int b);
int tn_panel_orth(const double* X, int64_t rs, int64_t cs, int64_t nrows, int b, double* Y, int64_t yrs, int64_t ycs, int method,
                  int* state9_host, double* dev_host, void* ws, int64_t ws_bytes, void* stream);
int tn_panel_stats(uint64_t* out16_host, int reset);
int tn_panel_stats_stream(uint64_t* out16_host, int reset, void* stream);
/* Factorisations of up to 64 columns (m >= n, at most 32 workgroups of rows, no pivoting) -- the plain `qr` of mps.py:43-59 as the
 * variational sweeps (mps.py:238-279) and the canonisation passes (mps.py:202-236) call it on 1024 x 64-class site matrices -- run as
 * ONE launch inside tn_qr / tn_site_qr (csrc/smallqr.hip: explicit-Q iterated Cholesky-QR, the triangular factors multiplied up, the
 * norm factor of mps.py:76-85 taken in the same launch); TN_QR_SMALL=0 keeps the blocked path.  DIAGNOSTIC counters of `stream`:
 * {factorisations, substitution passes applied, Householder fallbacks, launches that gave up at an in-kernel barrier}.
 * tn_fused_timeouts: the launches with in-kernel barriers (this one and the single-launch panel step) rely on their workgroups being
 * co-resident; the budget is derived from the device's CU count, GPU_MAX_HW_QUEUES and TN_PANEL_CU_BUDGET (the CUs this process may
 * count on when the card is shared).  A launch that gives up all the same poisons its outputs with NaN and is counted; every entry
 * point that used such launches asks before it returns (tn_compress_mps: once per call) and redoes the work through the six-launch
 * panel chain / the blocked path (same bits), or fails with -7 when its input was overwritten.  The hook reports how many launches of
 * `stream` gave up since the last check (synchronises the stream; a positive count takes the stream off these launch forms). */
int tn_smallqr_stats(uint64_t* out4_host, int reset, void* stream);
int tn_fused_timeouts(int* count_host, void* stream);
/* Strided batch of `batch` equally shaped QR problems (the rotations of examples/e06:97-109 at one site): item i
 * at A + i*bsA, Q + i*bsQ, R + i*bsR, keff_host[i].  ws_bytes >= batch * roundup(tn_qr_ws_bytes(m,n,nb), 256).  A factorisation
 * is a chain of latency-bound single-workgroup kernels, so the items are made CONCURRENT rather than fused: item i is enqueued on
 * side_streams[i % nside] (hipStream_t owned by the caller, nside <= 8), forked from and joined back into `stream` with events;
 * nside = 0 runs them one after the other on `stream`.  Per-item results are bit-identical to tn_qr. */
int tn_qr_batched(double* A, int64_t rs, int64_t cs, int64_t m, int64_t n, double* Q, int64_t qrs, int64_t qcs, double* R,
                  int64_t rrs, int64_t rcs, int nb, double rank_tol, int64_t* keff_host, int64_t batch, int64_t bsA, int64_t bsQ,
                  int64_t bsR, void* ws, int64_t ws_bytes, void* stream, void* const* side_streams, int nside);

/* ---- K4: truncated SVD of a centre matrix.  Replaces mps.svd (mps.py:24-40, sign gauge included) +
 * _mps_truncateC (:802-811): keep = min(#(S > S0*max(eps,tol)), Dmax), discarded = sqrt(sum S[keep:]^2)/S0.
 * C: k x n.  U: k x keep, S: keep values, Vt: keep x n are written for the first `keep` vectors only; buffers must
 * hold min(k, n, Dmax) vectors.  keep_host/discarded_host/sweeps_host/info_host are HOST pointers (may be NULL
 * except keep_host).  info: 0 converged, 1 sweep cap reached.
 * With up to 256 live vectors all Jacobi rounds run in ONE launch with the vectors resident in LDS (svdl_kernel: in-kernel barriers,
 * within the co-residency budget of the panel step; TN_SVD_FUSED=0 keeps three launches per round): same results bit for bit.  A launch
 * in which a barrier gave up reports it, the rounds are redone as separate launches inside the same call and the stream stays off the
 * single-launch forms (message on stderr); nothing non-finite is returned. */
int tn_svd_trunc(const double* C, int64_t crs, int64_t ccs, int64_t k, int64_t n, int64_t Dmax, double tol, double* U,
                 int64_t urs, int64_t ucs, double* S, double* Vt, int64_t vrs, int64_t vcs, int64_t* keep_host,
                 double* discarded_host, int* sweeps_host, int* info_host, void* ws, int64_t ws_bytes, void* stream);
/* ---- K5: singular values only, sorted descending, min(k,n) values written to HOST memory.
 * Replaces mps.svd_S (mps.py:62-73) as used by MPS.update_S (:550-560). */
int tn_svdvals(const double* C, int64_t crs, int64_t ccs, int64_t k, int64_t n, double* S_host, int* sweeps_host,
               int* info_host, void* ws, int64_t ws_bytes, void* stream);
int64_t tn_svd_ws_bytes(int64_t k, int64_t n, int vectors);
/* K5 without the read-back, for centre matrices with both dimensions <= 64 (the Schmidt-value checks of the variational sweeps,
 * mps.py:550-560, whose results are only needed at the end of a sweep): one asynchronous launch; out66_dev (DEVICE) receives 64
 * values sorted descending (zero padded), then the executed sweeps and a convergence flag (1 = converged) as doubles. */
int tn_svdvals_async(const double* C, int64_t crs, int64_t ccs, int64_t k, int64_t n, double* out66_dev, void* stream);
/* The same for `batch` centre matrices in ONE launch (one workgroup each): the Schmidt spectra of a whole variational sweep
 * (mps.py:550-560) are only compared at its end, so they are taken together after it.  Item i is described by 5 int64
 * {device address of C_i, vector stride, element stride, number of vectors, vector length} with the shorter side of C_i as the
 * vectors (k <= n: {C, crs, ccs, k, n}, else {C, ccs, crs, n, k}); desc_dev is that table on the device, desc_host the caller's
 * host copy of it (validated here, nothing on the device is dereferenced by the host); out66_dev + 66 i receives item i. */
int tn_svdvals_small_batched(const int64_t* desc_dev, int64_t batch, const int64_t* desc_host, double* out66_dev, void* stream);
/* Strided batches of the two SVD entry points (item i at C + i*bsC, U + i*bsU, S + i*bsS, Vt + i*bsV; the *_host outputs are
 * arrays of `batch` entries, S_host holds batch * min(k,n) values).  Ranks and convergence are read back per item, so the items
 * are issued one after the other and share the workspace of a single problem. */
int tn_svd_trunc_batched(const double* C, int64_t crs, int64_t ccs, int64_t k, int64_t n, int64_t Dmax, double tol, double* U,
                         int64_t urs, int64_t ucs, double* S, double* Vt, int64_t vrs, int64_t vcs, int64_t* keep_host,
                         double* discarded_host, int* sweeps_host, int* info_host, int64_t batch, int64_t bsC, int64_t bsU,
                         int64_t bsS, int64_t bsV, void* ws, int64_t ws_bytes, void* stream);
int tn_svdvals_batched(const double* C, int64_t crs, int64_t ccs, int64_t k, int64_t n, double* S_host, int* sweeps_host,
                       int* info_host, int64_t batch, int64_t bsC, void* ws, int64_t ws_bytes, void* stream);

/* ---- K6: out2[0] = 2^floor(log2 max|x|), out2[1] = 1/out2[0] (device).  Replaces mps.nfactor (mps.py:76-85).
 * slot8: 8 bytes of device scratch. */
int tn_nfactor(const double* x, int64_t n, double* out2, void* slot8, void* stream);
/* x /= nfactor(x) in place and out2 = [nfactor, 1/nfactor] (the fused form of tn_nfactor + tn_scale_by used after every QR,
 * mps.py:781-782, 796-797): two launches, no memset.  scratch: >= 8 KiB of device memory owned by the caller. */
int tn_normalize_pow2(double* x, int64_t n, double* out2, void* scratch, int64_t scratch_bytes, void* stream);
/* x[i] *= scalar_dev[0]  (used with out2+1 of tn_nfactor: mps.py:782, 797; tnac4o.py:533, 1781) */
int tn_scale_by(double* x, int64_t n, const double* scalar_dev, void* stream);
/* A[dl, s, dr] *= diag[s] (inv=0) or /= diag[s] (inv=1).  Replaces MPS.apply_diagonalO (mps.py:361-366). */
int tn_scale_phys(double* A, int64_t Dl, int64_t p, int64_t Dr, const double* diag, int inv, void* stream);

/* ---- K7: structured PEPS-factor and MPO-site builder.  Replaces tnac4o._peps_tensor (tnac4o.py:1562-1672) and the sum
 * over the physical index at tnac4o.py:1686 without ever forming the dense (q,l,d,r,u) tensor:
 *   F[s,l,u] = exp((Es[s] + E1[s,l]) + E4[s,u]) * Xu[u] * Xl[l] * Xr[rmap[s]] * Xd[dmap[s]]     (q, nl, nu)
 *   W[l,d,r,u] = sum_{s: dmap[s]=d, rmap[s]=r} F[s,l,u]                                          (nl, pd, br, nu)
 * Es/E1/E4 are the beta-scaled min-shifted energy tables beta*(min E - E); dmap/rmap are int32. */
int tn_peps_factor(const double* Es, const double* E1, const double* E4, const double* Xu, const double* Xl, const double* Xr,
                   const double* Xd, const int32_t* dmap, const int32_t* rmap, int64_t q, int64_t nl, int64_t nu, double* F,
                   void* stream);
int tn_mpo_from_factor(const double* F, const int32_t* dmap, const int32_t* rmap, int64_t q, int64_t nl, int64_t nu, int64_t pd,
                       int64_t br, double* W, void* stream);

/* ---- K8: conditional probabilities of one cell for a batch of branches.  Replaces the per-branch loop
 * tnac4o.py:444-448 around _calculate_Pn (:1786-1807), including the negative-probability rule.
 *   T1: (npref, p, Dr)  left environment times the top MPS site, one row block per distinct prefix
 *   RR: (nsuf, Dr, br)  right environments per distinct suffix
 *   F : (q, nl, nu)     non-zero factor of the PEPS tensor, T[s,l,d,r,u] = F[s,l,u] [d=dmap[s]] [r=rmap[s]]
 *   per branch kk: pref[kk], suf[kk], lidx[kk], uidx[kk] (int32).  Out: P (nb, q) normalised, minP (nb).
 *   parent_log2p (nb) / log2p_out (nb, q), both or neither: log2p_out[kk, s] = log2(P[kk, s]) + parent_log2p[kk], the expansion of
 *   the branch log-probabilities of tnac4o.py:450-453 in the same launch. */
int tn_calc_pn(const double* T1, const double* RR, const double* F, const int32_t* dmap, const int32_t* rmap,
               const int32_t* pref, const int32_t* suf, const int32_t* lidx, const int32_t* uidx, int64_t nb,
               int64_t q, int64_t nl, int64_t nu, int64_t p, int64_t Dr, int64_t br, double* P, double* minP,
               const double* parent_log2p, double* log2p_out, void* stream);
/* ---- a12: merge of the branches of a site-step with identical boundary indices (tnac4o.py:481-509).  The candidates arrive sorted by
 * group, in candidate order inside a group: E, log2p, deg, pos (their position in the candidate list), group g = members starts[g] ..
 * starts[g+1]-1 (ngroups + 1 offsets).  Per group: rep_pos_out = position of the FIRST member of minimal energy, deg_out = sum of the
 * degeneracies of the members within min_dEng of that minimum, log2p_out = the representative's log2p when it is alone in that set,
 * else the mean over the set added up in member order. */
int tn_merge_groups(const double* E, const double* log2p, const int64_t* deg, const int64_t* pos, const int64_t* starts, int64_t ngroups,
                    double min_dEng, int64_t* rep_pos_out, int64_t* deg_out, double* log2p_out, void* stream);
/* ---- K9: per-item power-of-two normalisation of a batch of environments (tnac4o.py:533, 1781):
 * each of the `batch` contiguous blocks of `len` doubles is divided by its own nfactor. */
int tn_nfactor_batched(double* x, int64_t batch, int64_t len, void* stream);
/* ---- K9: right environments of every distinct boundary suffix of the beam for one site.  Replaces the body of the key loop
 * of tnac4o._setup_RR (tnac4o.py:1777-1783): gather of the parent environment, both contractions and the nfactor rescale,
 * one launch for all keys, no (Dl p) x br intermediate.
 *   A: (Dl, p, Dr) top boundary-MPS site;  RRprev: (nprev, Dr, br) environments of the previous level;
 *   W: (bl, p, br, pu) row-MPO site (legs left, down, right, up);  parent[k] (row of RRprev), uidx[k] (up index) int32;
 *   out[k, x, l] = sum_{d, x', r} A[x, d, x'] RRprev[parent[k], x', r] W[l, d, r, uidx[k]]  / nfactor(out[k]).   (nk, Dl, bl)
 * Limits (argument error otherwise): Dl * bl <= 2048, (Dr*br + bl*p*br + Dr*bl) * 8 <= 150 KiB. */
int tn_env_rr_batched(const double* A, const double* RRprev, const double* W, const int32_t* parent, const int32_t* uidx,
                      int64_t nk, int64_t Dl, int64_t p, int64_t Dr, int64_t bl, int64_t br, int64_t pu, double* out,
                      void* stream);
/* ---- K9: left environments of the new distinct prefixes (tnac4o.py:528-535).  T1 = RL . A (npref, p, Dr) is the product K8
 * already uses, so RL'[k] = RL[par] . A[:, d, :] is row (par[k], didx[k]) of it:  out[k, :] = T1[par[k], didx[k], :] / nfactor. */
int tn_env_rl_batched(const double* T1, const int32_t* par, const int32_t* didx, int64_t nk, int64_t p, int64_t Dr,
                      double* out, void* stream);
/* ---- bond balancing of the preconditioner.  Replaces scipy.linalg.matrix_balance(env, permute=False, separate=True) and the
 * clamp that follows it (tnac4o.py:1845-1847): LAPACK dgebal, job 'S' (radix 2, 2-norms), on the n x n matrix A (n <= 64,
 * element strides rs/cs).  scale_out[i] (device, n doubles) = min(max(scale[i], 1/max_scale), max_scale); max_scale <= 0
 * disables the clamp.  iters_out (device int, may be NULL) = passes of the outer loop. */
int tn_balance(const double* A, int64_t rs, int64_t cs, int64_t n, double max_scale, double* scale_out, int* iters_out,
               void* stream);

/* ---- site steps of the sweeps as single calls (compositions of the kernels above with their temporaries in `ws`; results are
 * bit-identical to the separate calls).  They exist for the host: with 4 chains driven by 4 host threads every separate call and
 * every temporary tensor is a GIL hand-over.  All matrices contiguous, row-major (C order).
 * tn_site_qr: attach + QR + nfactor of one canonisation step (mps.py:368-380, 532-548, 772-800).
 *   side 0 (left sweep):  M = C (kc x Dl) . A (Dl, p, Dr), QR of the (kc p) x Dr matrix: Q (kc p x k), R (k x Dr).
 *   side 1 (right sweep): M = A (Dl, p, Dr) . C (Dr x kc), QR of the transposed (p kc) x Dl view: Q receives Q^T (k x p kc),
 *                         R receives R^T (Dl x k).        k = min of the two matrix dimensions.
 *   C == NULL: no attach, A itself is factored and DESTROYED (kc ignored).  rank_tol / keff_host as in tn_qr.
 *   nf_out2 != NULL: when the factorisation ran to the end the triangular factor is divided by its nfactor and nf_out2 (device)
 *   = [nf, 1/nf]; *normalised_host tells whether that happened (it does not after an early exit: the caller slices first).
 *   dropped2_host (HOST, may be NULL): squared Frobenius norm of the trailing block an early exit dropped (0 otherwise).
 *   frobenius_exit = 1: the early exit compares the Frobenius norm of the trailing block with rank_tol x the Frobenius norm of
 *   the input (instead of the largest column norms of the two).
 *   pivot_perm_host (HOST, n int64 with n = columns of the factored matrix, may be NULL): panel pivoting.  Before every panel
 *   the residual norms of all remaining columns are read back; the factorisation stops when their Frobenius norm is below
 *   rank_tol x the input's, otherwise the 32 columns with the largest residuals form the next panel.  The triangular factor
 *   is returned in the pivoted column order; pivot_perm_host[j] = input column at position j. */
int64_t tn_site_qr_ws_bytes(int side, int64_t Dl, int64_t p, int64_t Dr, int64_t kc, int attach);
int tn_site_qr(int side, double* A, int64_t Dl, int64_t p, int64_t Dr, const double* C, int64_t kc, double* Q, double* R,
               double rank_tol, int64_t* keff_host, double* nf_out2, int* normalised_host, double* dropped2_host,
               int frobenius_exit, int64_t* pivot_perm_host, void* ws, int64_t ws_bytes, void* stream);
/* ---- helpers of the weighted rank-revealing first canonisation pass (tnac4o_amd/mps.py: canonise_right_weighted; no
 * counterpart in the reference, whose first pass factors every site in full, mps.py:187):
 * tn_gram_weights: from the Gram matrix G (n x n) of the unfactored part on the other side of a bond, the squared weight of
 *   every bond index, d2[c] = max(G_cc, floor_rel max G), and stats65 = [ 64 partial sums of ||K||_F^2 with K = G / (d d^T) (to be
 *   added in order: reproducible bit for bit), max_c G_cc ].
 * tn_rows_norm2: out[r] = sum_c A[r,c]^2 for a row-major rows x cols matrix.
 * tn_gather_scale_rows: inverse = 0: out[j,:] = sqrt(w2[perm[j]]) A[perm[j],:];  inverse = 1: out[perm[j],:] = A[j,:] / sqrt(w2[perm[j]])
 *   (perm: int64 device vector). */
int tn_gram_weights(const double* G, int64_t n, double floor_rel, double* d2_out, double* stats65_out, void* stream);
int tn_rows_norm2(const double* A, int64_t rows, int64_t cols, double* out, void* stream);
int tn_gather_scale_rows(const double* A, int64_t rows, int64_t cols, const int64_t* perm, const double* w2, double* out, int inverse,
                         void* stream);
/* ---- a truncation that cannot truncate (reference: MPS.truncateC, mps.py:562-585 -> _mps_truncateC, mps.py:802-811, as called by the
 * 4 chi / 2 chi passes of compress_mps, mps.py:192-195).  With min(C.shape) <= Dmax and tol <= eps the reference's rule only removes
 * singular values below eps S0 and turns the bond into the Schmidt basis; inside an intermediate pass neither is visible afterwards
 * (the next canonisation step's triangular factor does not depend on an orthogonal change of the bond, the variational sweep is
 * covariant under it).  tn_bond_deflate removes the same noise without a decomposition: C sits between an orthonormal site Q and a
 * canonical rest, so zeroing bond index i changes the state by exactly the norm of row (side 0) / column (side 1) i of C; indices are
 * dropped in ascending order of that norm while the dropped squares add up to at most eps^2 max_i ||C_i||^2 (<= (eps S0)^2), and C and
 * Q are gathered to the kept indices (in their order).
 *   side 0 (left sweep):  C (k x n), Q (m x k)  ->  C_out (k' x n), Q_out (m x k')        side 1: C (n x k), Q (k x m) -> (n x k'), (k' x m)
 * all row-major and contiguous, 1 <= k <= 256.  *k_out_host = k'; when k' == k nothing was written and the caller keeps C and Q.
 * *dropped2_rel_host = dropped squares / max_i ||C_i||^2.  ws: k doubles.  One read-back (synchronises the stream). */
int tn_bond_deflate(int side, const double* C, int64_t k, int64_t n, const double* Q, int64_t m, double* C_out, double* Q_out,
                    int64_t* k_out_host, double* dropped2_rel_host, void* ws, int64_t ws_bytes, void* stream);
/* out (c, s, c2) = RL (c x a) . A (a, s, a2) . RR (a2 x c2)      (MPS._mps_RAR, mps.py:748-751) */
int64_t tn_rar_ws_bytes(int64_t c, int64_t a, int64_t s, int64_t a2, int64_t c2);
int tn_rar(const double* RL, const double* A, const double* RR, int64_t c, int64_t a, int64_t s, int64_t a2, int64_t c2, double* out,
           void* ws, int64_t ws_bytes, void* stream);
/* mixed environments (MPS._mps_RL / _mps_RR, mps.py:655-663), A (a, s, a2), Ac (c, s, c2):
 *   side 0: out (c2 x a2) = sum_{c,s,a} Ac[c,s,c2] R[c,a] A[a,s,a2];   side 1: out (a x c) = sum A[a,s,a2] R[a2,c2] Ac[c,s,c2] */
int64_t tn_env_mix_ws_bytes(int side, int64_t a, int64_t s, int64_t a2, int64_t c, int64_t c2);
int tn_env_mix(int side, const double* R, const double* A, const double* Ac, int64_t a, int64_t s, int64_t a2, int64_t c, int64_t c2,
               double* out, void* ws, int64_t ws_bytes, void* stream);
/* projectors of a truncation pushed into the neighbouring sites and the diagonal centre (mps.py:579-583):
 *   Al_new (ml x keep) = Al (ml x k0) . U (k0 x keep, strides urs/ucs);  Ar_new (keep x nr) = Vt (keep x k1, strides vrs/vcs) . Ar (k1 x nr);
 *   Cdiag (keep x keep) = diag(S). */
int64_t tn_apply_truncation_ws_bytes(int64_t ml, int64_t k0, int64_t keep, int64_t k1, int64_t nr);
int tn_apply_truncation(const double* Al, int64_t ml, int64_t k0, const double* U, int64_t urs, int64_t ucs, int64_t keep,
                        const double* Vt, int64_t vrs, int64_t vcs, const double* Ar, int64_t k1, int64_t nr, const double* S,
                        double* Al_new, double* Ar_new, double* Cdiag, void* ws, int64_t ws_bytes, void* stream);

/* ---- measurement: bracket every launch of the selected kernel families with HIP events on the launch stream.
 * family ids: 0-3 gemm_kernel<128,128> / <128,32> / <32,128> / <64,64> (all operand layouts), 4 splitk_reduce,
 * 5 absorb, 6 gram_partial, 7 eig_small, 8 rows_times_small, 9 small_t_times_vecs, 10 tsqr_factor/apply,
 * 11 lu_reconstruct, 12 QR auxiliaries (diag_qr, assemble_R, init_Q, norms, copies), 13 SVD auxiliaries (norms, init,
 * gather), 14 misc (nfactor, scaling, builders, beam kernels).
 * mask bit f enables family f.  tn_profile_get synchronises the recorded events and returns totals since the last
 * reset: launches, summed duration (ms), algorithmic flops and bytes (SURVEY.md §8d counts).
 * Counter-only families (no events; active whenever any family is enabled): 15 tn_qr nominal (calls, 4mn^2-4/3n^3,
 * 8(2mn+n^2)), 16 tn_svd_trunc nominal (14mn^2+8n^3, 8(2mn+n^2+n)), 17 Jacobi streaming model (calls = executed
 * sweeps, bytes = sweeps*(n-1)*16*n*(m+n)), 18 tn_svdvals nominal (4mn^2-4/3n^3, 8(mn+n)), 19 block-Jacobi rounds (calls =
 * executed rounds of all tn_svd_trunc / tn_svdvals calls, flops = pair eigenproblems).
 * tn_profile_get_phase splits the same totals by the entry point that issued the launch: phase 0 other (attach /
 * projector / environment GEMMs, scaling), 1 tn_absorb, 2 tn_qr, 3 tn_svd_trunc, 4 tn_svdvals, 5 MPO builders;
 * phase -1 = all. */
void tn_profile_enable(unsigned mask);
void tn_profile_reset(void);
/* bracket only every `every`-th launch of an enabled family (default 1 = all): totals then cover the sampled launches */
void tn_profile_sample(unsigned every);
int tn_profile_get(int family, uint64_t* calls_host, double* ms_host, double* flops_host, double* bytes_host);
int tn_profile_get_phase(int phase, int family, uint64_t* calls_host, double* ms_host, double* flops_host, double* bytes_host);

/* ---- the chain driver: MPS.apply_mpo (mps.py:353-359) + MPS.compress_mps (mps.py:175-200) of ONE boundary MPS in ONE call ------------
 * What tnac4o.py:1688-1693 does per row (copy, apply_mpo, compress_mps) without a host interpreter between the ~600 kernel-launching
 * steps: the same kernels on the same operands in the same order as the per-step entry points above (bit-identical results), walked in
 * C++ on the caller's thread.  Every intermediate tensor lives in `arena` (DEVICE, 256-byte aligned, arena_bytes >=
 * tn_compress_mps_arena_bytes(...)), managed by a host-side allocator inside the call; the library still allocates nothing on the device.
 *   sites_host[n]   DEVICE pointer to MPS site n, (Dl, p, Dr) C-order, dims in site_dims_host[3n .. 3n+2]; read only
 *   mpo_host[n]     DEVICE pointer to MPO site n (ba, po, bb, pi) C-order (dims in mpo_dims_host[4n ..]) or NULL = no absorption at
 *                   that site; mpo_host itself may be NULL (compress only).  hconj as for tn_absorb
 *   Dmax, tolS, tolV, max_sweeps, graduate   the arguments of compress_mps (graduate_truncation as 0/1)
 *   flags           bit 0: weighted rank-revealing first pass (MPS.canonise_right_weighted), bit 1: its Gram recursion
 *                   through the MPS (x) MPO structure, bit 2: lazy Schmidt values in a stage's last sweep
 *   out             DEVICE, L slots of out_slot doubles: compressed site n (left-canonical, C-order) at out + n*out_slot, its dims in
 *                   out_dims_host[3n ..]
 *   overlap_host    <psi|phi> returned by compress_mps;  discarded_host[L+1]: MPS.discarded;  schmidt_host ((L+1) x schmidt_pitch) /
 *                   schmidt_len_host[L+1]: MPS.S (-1: bond never measured)
 *   nfs_dev         DEVICE table of nfs_cap pairs [nf, 1/nf]: the power-of-two factors taken out of the centre matrices (normC is
 *                   their product), *nfs_count_host of them written
 *   info_host[8]    {a-posteriori bound of the weighted pass, plain-pass fallbacks, weighted pass used, peak arena bytes, sum of the
 *                   bond dimensions before / after the first canonisation pass, attempts redone after a barrier time-out (see
 *                   tn_fused_timeouts: the call checks once, at its end, and redoes the row through the multi-launch forms), 0}
 * tn_compress_mps_arena_bytes: Dmax >= 0 -> what a call needs when the weighted first pass is accepted (~1.6x the measured peak at
 * L = 2048); Dmax < 0 -> the conservative bound that also covers the plain first pass on the kept input.  A call that runs out of arena
 * fails with -3 and has returned nothing: retry with the conservative size (tnac4o_amd.ops does).  nfs_cap too small: -3 as well (size
 * it (2 max_sweeps + 16) L + 64).
 * Synchronises `stream` wherever the algorithm needs a number on the host (kept ranks, convergence, overlaps).  On return the results
 * have been copied out of the arena by `stream`; the arena may be reused by the next call on the same stream.
 * Errors: -3 arena (or factor table) too small, -4 Jacobi sweeps did not converge even after QR preconditioning, -5 zero centre matrix,
 * -7 launches with in-kernel barriers gave up twice in a row. */
int64_t tn_compress_mps_arena_bytes(int64_t L, const int64_t* site_dims_host, const int64_t* mpo_dims_host, int64_t Dmax);
int tn_compress_mps(int64_t L, const double* const* sites_host, const int64_t* site_dims_host, const double* const* mpo_host,
                    const int64_t* mpo_dims_host, int hconj, int64_t Dmax, double tolS, double tolV, int max_sweeps, int graduate, int flags,
                    double* out, int64_t out_slot, int64_t* out_dims_host, double* overlap_host, double* discarded_host, double* schmidt_host,
                    int64_t schmidt_pitch, int64_t* schmidt_len_host, double* nfs_dev, int64_t nfs_cap, int64_t* nfs_count_host, double* info_host,
                    void* arena, int64_t arena_bytes, void* stream);
/* The two small reductions of the weighted first pass whose order is part of the result (both drivers call these):
 * perm_out (DEVICE int64[n]) = stable descending argsort of w (NaN first);  w_out[i] = a[i] * b[i] and sum_out[0] = their sum in a fixed order. */
int tn_argsort_desc(const double* w, int64_t n, int64_t* perm_out, void* stream);
int tn_weighted_sum(const double* a, const double* b, int64_t n, double* w_out, double* sum_out, void* stream);

/* ---- K8 driver: the beam search of search_ground_state walked in C++ (reference tnac4o.py:429-542) -------------------------
 * One lattice cell (ny, nx) as the search sees it; every pointer is a DEVICE pointer that stays valid for the call:
 *   F, dmap, rmap, q, nl, nu, pd, br   the PEPS factor of the cell as tn_peps_factor builds it (tnac4o.py:1562-1672)
 *   down, right (int64[q])             boundary index a cell state sends to the row below / the cell to its right (:1469-1489)
 *   Es (q), E1 (q x e1cols), E4 (q x e4cols)   energy tables of tnac4o._update_Eng (:1506-1558); E1 / E4 may be NULL in column / row 0
 *   left_map / up_map (int64)          Ising: state of the left / upper neighbour -> column of E1 / E4 (that neighbour's right / down
 *                                      table); NULL: the neighbour's state itself is the column
 *   A, Dl, p, Dr                       site nx of the boundary MPS above the row (rhoT[ny+1]), C-order; p must equal pd */
typedef struct tn_beam_cell {
    const double* F;
    const int32_t* dmap;
    const int32_t* rmap;
    const int64_t* down;
    const int64_t* right;
    const double* Es;
    const double* E1;
    const double* E4;
    const int64_t* left_map;
    const int64_t* up_map;
    const double* A;
    int64_t q, nl, nu, pd, br, e1cols, e4cols, Dl, p, Dr;
} tn_beam_cell;
/* tn_beam_search: rows ny = 0 .. Ny-1, sites nx = 0 .. Nx-1 (cells[ny * Nx + nx], HOST array), at most M branches, candidates cut at
 * log2 p <= max + log2_cutoff when has_cut, merge window min_dEng, B > every boundary index (radix of the row keys).  Canonical
 * order of tnac4o_amd/beam.py (ascending candidate index, lexicographic merge groups, first member of minimal energy, stable top M).
 * Results (DEVICE, room for M branches): states_out (M x Nx*Ny int16, lattice order of this rotation), energy_out, log2p_out, deg_out;
 * HOST: *nb_host branches returned, *pd_max_host largest log2 p cut or dropped, *globalmin_host smallest conditional-table flag.
 * Synchronises `stream` four times per site-step (counts).  Needs Dl * nl <= 2048 at every site (tn_env_rr_batched).
 * Errors: -3 workspace too small, -6 no candidate left. */
int64_t tn_beam_search_ws_bytes(int64_t Nx, int64_t Ny, int64_t M, int64_t qmax, int64_t max_env, int64_t max_t1, int64_t max_w);
int tn_beam_search(int64_t Nx, int64_t Ny, const tn_beam_cell* cells, int64_t M, int has_cut, double log2_cutoff, double min_dEng, int64_t B,
                   int16_t* states_out, double* energy_out, double* log2p_out, int64_t* deg_out, int64_t* nb_host, double* pd_max_host,
                   double* globalmin_host, void* ws, int64_t ws_bytes, void* stream);
/* The same walk shared by a TEAM of `team` processes (one per GPU; SURVEY.md 8e-ii: the beam shards of one lattice rotation).  Every rank
 * holds the whole beam and calls tn_beam_search_team with the same arguments and its own `rank`; the conditional tables of a site-step
 * (tn_calc_pn, tnac4o.py:444-453) are evaluated for the rank's contiguous slice of the branches, [nb rank / team, nb (rank + 1) / team),
 * then `exchange` is called on every rank, in the same order, once per site-step:
 *     exchange(ctx, log2p (DEVICE, nb x q), minp (DEVICE, nb), nb, q, rank, team)
 * and must return 0 with the slices of ALL ranks in place on every rank (row b of log2p / entry b of minp belongs to the rank whose slice
 * holds b) -- e.g. one broadcast per rank over the team's communicator, enqueued on `stream` or ordered behind it.  From there on all
 * ranks run the identical deterministic cut, merge and selection: the results are those of tn_beam_search bit for bit, on every rank.
 * team = 1 (exchange may be NULL) is tn_beam_search.  Errors as tn_beam_search; -8: the exchange function returned non-zero. */
typedef int (*tn_beam_exchange_fn)(void* ctx, double* log2p, double* minp, int64_t nb, int64_t q, int rank, int team);
int tn_beam_search_team(int64_t Nx, int64_t Ny, const tn_beam_cell* cells, int64_t M, int has_cut, double log2_cutoff, double min_dEng, int64_t B,
                        int16_t* states_out, double* energy_out, double* log2p_out, int64_t* deg_out, int64_t* nb_host, double* pd_max_host,
                        double* globalmin_host, void* ws, int64_t ws_bytes, void* stream, int rank, int team, tn_beam_exchange_fn exchange,
                        void* exchange_ctx);

#ifdef __cplusplus
}
#endif
#endif
