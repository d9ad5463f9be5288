// Shared helpers for libtnpeps (gfx950 only).  Internal header — the public C-ABI is include/tnpeps.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

namespace tn {

// thread-local error text, retrievable through tn_last_error()
void set_error(const char* fmt, ...);
const char* get_error();

inline int hip_fail(hipError_t e, const char* what) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e > 0 ? (int)e : 1;
}

#define TN_CHECK_ARG(cond, msg)                \
    do {                                       \
        if (!(cond)) {                         \
            tn::set_error("%s: %s", __func__, msg); \
            return -1;                         \
        }                                      \
    } while (0)

#define TN_CHECK_LAUNCH(what)                          \
    do {                                               \
        hipError_t e__ = hipGetLastError();            \
        if (e__ != hipSuccess) return tn::hip_fail(e__, what); \
    } while (0)

// Thread-local page-locked host buffers for the small read-backs / uploads around host decisions (slot 0..7: 0-1 tn_qr, 2-3 the
// Jacobi SVD, 4-6 the chain driver; grown on demand,
// released when the thread exits).  A copy into pageable memory makes hipMemcpyAsync drain the stream on the host first and only
// then enqueue the transfer (15-20 us of idle device per read-back, 15 k read-backs per sweep); with page-locked memory the
// transfer is queued right behind the producing kernel.  Returns nullptr if the allocation fails (callers fall back to pageable).
void* pinned_host(size_t bytes, int slot);

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t align_up(int64_t a, int64_t b) { return cdiv(a, b) * b; }

// Fast reciprocal square root / reciprocal with three Newton steps (full double accuracy to ~1 ulp; the hardware
// seeds are single-precision accurate).  They sit on the per-column critical path of the panel factorisation.
static __device__ __forceinline__ double fast_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = r * (1.5 - 0.5 * x * r * r);
    r = r * (1.5 - 0.5 * x * r * r);
    r = r * (1.5 - 0.5 * x * r * r);
    return r;
}
static __device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = r * (2.0 - x * r);
    r = r * (2.0 - x * r);
    r = r * (2.0 - x * r);
    return r;
}


// ---- optional event timing per kernel family (prof.hip) ---------------------------------------------
enum { PROF_GEMM_128x128 = 0, PROF_GEMM_128x32, PROF_GEMM_32x128, PROF_GEMM_64x64, PROF_SPLITK_REDUCE, PROF_ABSORB,
       PROF_GRAM, PROF_EIG, PROF_ROWS_SMALL, PROF_VECS_SMALL, PROF_TSQR,
       PROF_LU,        // lu_reconstruct_kernel (Householder reconstruction of a panel)
       PROF_QR_AUX,    // diag_qr, assemble_R, init_Q, column norms, panel copies
       PROF_SVD_AUX,   // vector norms, init, gather of the Jacobi SVD
       PROF_MISC,      // nfactor / scaling / builders / beam kernels
       PROF_NKERNEL,   // ---- families above bracket kernel launches with events; the ones below are pure counters ----
       PROF_QR_NOMINAL = PROF_NKERNEL,   // per tn_qr call: calls, 4mn^2 - 4/3 n^3 flops, 8(2mn + n^2) bytes
       PROF_SVD_NOMINAL,                 // per tn_svd_trunc call: 14mn^2 + 8n^3 flops, 8(2mn + n^2 + n) bytes
       PROF_SVD_STREAM,                  // per tn_svd_trunc call: calls += executed sweeps, bytes += sweeps*(n-1)*16*n*(m+n)
       PROF_SVDVALS_NOMINAL,             // per tn_svdvals call: 4mn^2 - 4/3 n^3 flops, 8(mn + n) bytes
       PROF_SVD_ROUNDS,                  // per block-Jacobi SVD: calls += executed rounds (sweeps x rounds per sweep), flops += pair eigenproblems
       PROF_NFAM };
// phase a launch is attributed to (thread-local; set by the entry points that own a phase)
enum { PH_OTHER = 0, PH_ABSORB, PH_QR, PH_SVD, PH_SVDVALS, PH_BUILD, PH_N };
bool prof_on(int fam);
void prof_begin(hipStream_t st, int fam);
void prof_end(hipStream_t st, int fam, double flops, double bytes);
void prof_note(int fam, double calls, double flops, double bytes);   // counter-only families
int prof_phase(int phase);                                           // returns the previous phase
struct ProfPhase {                                                    // scoped phase
    int prev;
    explicit ProfPhase(int ph) : prev(prof_phase(ph)) {}
    ~ProfPhase() { prof_phase(prev); }
};
// bracket one kernel launch of a family that has no flop/byte model
#define TN_PROF_LAUNCH(st, fam, launch) do { prof_begin(st, fam); launch; prof_end(st, fam, 0.0, 0.0); } while (0)
void prof_set_mask(unsigned mask);
void prof_set_sample(unsigned n);
void prof_reset();
void prof_get(int phase, int fam, uint64_t* calls, double* ms, double* flops, double* bytes);   // phase < 0: all phases

// ---- strided matrix view (element strides) -------------------------------------------------
struct Mat {
    double* p;
    int64_t rs, cs;     // row stride, column stride (in doubles)
};
static inline Mat mat(double* p, int64_t rs, int64_t cs) { return Mat{p, rs, cs}; }
static inline Mat sub(Mat a, int64_t i, int64_t j) { return Mat{a.p + i * a.rs + j * a.cs, a.rs, a.cs}; }
static inline Mat tr(Mat a) { return Mat{a.p, a.cs, a.rs}; }

// ---- internal GEMM entry (gemm_f64.hip) --------------------------------------------------------
// C[M,N] = alpha * A[M,K] * B[K,N] + beta * C, arbitrary element strides, optional batch.
// ws/ws_bytes: optional split-K scratch (may be null -> no split).
int gemm(hipStream_t st, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t rsa, int64_t csa,
         const double* B, int64_t rsb, int64_t csb, double beta, double* C, int64_t rsc, int64_t csc,
         int64_t batch = 1, int64_t bsa = 0, int64_t bsb = 0, int64_t bsc = 0, double* ws = nullptr,
         int64_t ws_bytes = 0);
int64_t gemm_ws_bytes(int64_t M, int64_t N, int64_t K, int64_t batch);
// Extended form: block-pair row indirection (see GemmP in gemm_f64.hip), forced split-K with the partial sums left in
// ws as [(batch*splitk + s)][M][N] (raw_partials), per-batch skip flags.
struct GemmExtra {
    const int* pairs = nullptr;
    const int* skip = nullptr;
    int pw = 0, mapA = 0, mapB = 0, mapC = 0;
    int force_splitk = 0;
    bool raw_partials = false;
    int* splitk_used = nullptr;
};
// how gemm_ex cuts K for force_splitk = s: returns the number of splits actually used, *kchunk = elements per split
int gemm_forced_split(int64_t K, int s, int64_t* kchunk);
int gemm_ex(hipStream_t st, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t rsa, int64_t csa,
            const double* B, int64_t rsb, int64_t csb, double beta, double* C, int64_t rsc, int64_t csc, int64_t batch,
            int64_t bsa, int64_t bsb, int64_t bsc, double* ws, int64_t ws_bytes, const GemmExtra* x);

static inline int gemm(hipStream_t st, int64_t M, int64_t N, int64_t K, double alpha, Mat A, Mat B, double beta, Mat C,
                       double* ws = nullptr, int64_t ws_bytes = 0) {
    return gemm(st, M, N, K, alpha, A.p, A.rs, A.cs, B.p, B.rs, B.cs, beta, C.p, C.rs, C.cs, 1, 0, 0, 0, ws, ws_bytes);
}

// ---- small dense kernels (small.hip) ------------------------------------------------------------
constexpr int NBMAX = 64;      // largest panel / Jacobi pair width handled by the single-workgroup kernels

// Partial Gram matrices of `nvec` vectors of length L:  G[i][j] = sum_c X(i,c) X(j,c).
// Vector v, element c lives at X + vec_off[v] + c*es  where vec_off is given by (blk0,blk1,w,vs):
//   v <  w : (blk0*w + v) * vs ;  v >= w : (blk1*w + v - w) * vs.   pairs==nullptr -> single group blk0=0,blk1=1.
// Output: part[(g*nchunk + chunk)*nvec*nvec + i*nvec + j].
int gram_partial(hipStream_t st, const double* X, int64_t vs, int64_t es, int64_t L, int nvec, int w, const int* pairs,
                 int ngroups, int nchunk, double* part);
int gram_nchunk(int64_t L);

// Sum partials, (optionally) scale to unit diagonal, diagonalise with parallel-order Jacobi.
// mode 0: QR panel step  -> out = D^-1 J         (columns with squared norm <= dead_thresh are flagged in
//         dead[g*nvec+i] and get a zero row/column in out)
// mode 1: normalise only -> out = D^-1
// mode 2: SVD pair step  -> out = J (orthogonal), unscaled Gram; nrot[g] = rotations applied, maxoff[g] = largest
//         relative off-diagonal seen before rotating.
// relevant2 (mode 2): vectors with squared norm <= relevant2 are left out of the convergence measure maxoff.
// allow_fast (mode 2, 64 vectors): pairs in the quadratic regime may take the near-diagonal steps on the matrix cores instead of the
// cyclic sweeps (TN_EIG_FAST; see eig_small3_kernel).
int eig_small(hipStream_t st, const double* part, int nchunk, int nvec, int ngroups, int mode, int max_sweeps,
              double dead_thresh, double* out, int* dead, int* nrot, double* maxoff, double relevant2 = 0.0, int allow_fast = 1);

// In place:  X(r, 0:b) <- X(r, 0:b) * S   for r < nrows  (S is b x b row-major in global memory).
int rows_times_small(hipStream_t st, double* X, int64_t rs, int64_t cs, int64_t nrows, int b, const double* S);

// Out(v, c) = sum_u S[u][v] * In(u, c)  for the nvec vectors of each group (same addressing as gram_partial),
// c < L; in place when Out == In.  If nrot != nullptr groups with nrot[g]==0 are skipped.
// All Jacobi rounds of a block-pair SVD (jacobi_core, svd.hip) in one launch (small.hip).  norms: nvp + 4 doubles.
struct SvdRoundsJob {
    double* X; int64_t pitch, L; int nvp, w;
    const int* pairs; int ng, nr, nchunk;
    double* part; int64_t part_bytes; double* Js; int* nrot; double* maxoff;
    double relevant2, last_tol; int inner_first, inner_later;
    double* norms;
};
int svd_rounds_fused(hipStream_t st, const SvdRoundsJob& j);       // 0 launched, 1 not taken
int small_t_times_vecs(hipStream_t st, const double* S, double* X, int64_t vs, int64_t es, int64_t L, int nvec, int w,
                       const int* pairs, int ngroups, const int* nrot);


// ---- TSQR panel orthonormalisation (tsqr.hip) -------------------------------------------------------------------
int64_t tsqr_ws_bytes(int64_t nrows, int b);
int tsqr_orthonormalize(hipStream_t st, const double* Xin, int64_t irs, int64_t ics, double* X, int64_t rs, int64_t cs,
                        int64_t nrows, int b, void* ws, int64_t ws_bytes);


// ---- iterated Cholesky-QR panel orthonormalisation (cholqr.hip), the default panel step -------------------------------
int64_t cholqr_ws_bytes(int64_t nrows, int b);
int cholqr_reset(hipStream_t st, void* ws);          // zero the state block at the head of ws once per call, before the first panel
// the stream's own state block (zero on return; see cholqr.hip) -- *state_out goes to the panel calls of this factorisation; the
// caller's first launch after the last panel clears the block (CQ_STATE_BYTES) and then tells cholqr_end_ok
int cholqr_begin(hipStream_t st, void* ws, void** state_out);
void cholqr_end_ok(hipStream_t st);
// fused_base: host counter of the call (0 at its start) for the single-launch form of small panels; NULL = six-launch chain
// state: the block from cholqr_begin, or NULL = the head of ws (cleared with cholqr_reset)
int cholqr_orthonormalize(hipStream_t st, const double* X, int64_t irs, int64_t ics, double* Y, int64_t rs, int64_t cs, int64_t nrows,
                          int b, void* ws, int64_t ws_bytes, uint64_t seed, int* fused_base, void* state = nullptr);
// the whole panel step: orthonormal basis, and with reconstruct != 0 the Householder reconstruction (Y, T, W = Y T^T, Wq = Y T)
int cholqr_panel(hipStream_t st, const double* X, int64_t irs, int64_t ics, double* Y, int64_t rs, int64_t cs, int64_t nrows, int b, void* ws,
                 int64_t ws_bytes, uint64_t seed, int reconstruct, double* Tp, double* W, int64_t wrs, int64_t wcs, double* Wq, int* fused_base,
                 void* state = nullptr, const int* active = nullptr);       // active (DEVICE, may be null): 0 = every launch of this panel returns at once
// launches with in-kernel barriers (cq_fused_kernel, sq_kernel): co-residency budget and time-outs, see cholqr.hip
bool fused_forms_allowed(hipStream_t st, int nwg);
void fused_forms_disable(hipStream_t st);
void fused_stream_released(hipStream_t st);
void fused_note_launch();
bool fused_check_needed();
bool fused_check_deferred();
void fused_defer_push();
void fused_defer_pop();
struct FusedDeferCheck { FusedDeferCheck() { fused_defer_push(); } ~FusedDeferCheck() { fused_defer_pop(); } };
int fused_timeouts(hipStream_t st, int* count_out);       // synchronises st; *count_out > 0: redo the work since the last check
// one-launch factorisation of m x n, n <= 64 (smallqr.hip): 0 done, 1 shape / stream not taken, else error
bool smallqr_fits(int64_t m, int64_t n);
int64_t smallqr_ws_bytes(int64_t m, int64_t n);
int smallqr_factor(hipStream_t st, const double* A, int64_t rs, int64_t cs, int64_t m, int64_t n, double* Q, int64_t qrs, int64_t qcs, double* R,
                   int64_t rrs, int64_t rcs, double* nf_out2, void* ws, int64_t ws_bytes);
int cholqr_stream_slot(hipStream_t st);     // statistics / state slot of a stream (CHOLQR_SLOTS = no slot of its own)
constexpr int CHOLQR_SLOTS = 64;
int cholqr_debug_state(hipStream_t st, const void* ws, int* ints9, double* dev_hist);
int cholqr_stats(unsigned long long* out16, int reset, hipStream_t st_or_null, int all_streams);

}  // namespace tn
