// Strided-batch forms of the factorisations ("strided-batched over rotations", the loop of reference
// examples/e06_search_gs_degeneracy_J124.py:97-109): `batch` equally shaped, independent problems per call.
//
// A factorisation here is a chain of latency-bound single-workgroup kernels, so the useful form of batching is not one fat
// launch per step but CONCURRENCY between the items: item i is enqueued on side stream i % nside (forked from and joined
// back into the caller's stream with events), so that the items' chains interleave on the device exactly like the lattice
// rotations of parallel.run_concurrent do -- without host threads.  tn_qr_batched with rank_tol = 0 is fully asynchronous.
// The SVD entry points read ranks / convergence back per item, so their items are issued one after the other; they are
// provided for API symmetry (one call per site for all rotations) and keep the per-item results bit-identical to single calls.
#include "common.h"

namespace tn {

int qr_factor(hipStream_t, double*, int64_t, int64_t, int64_t, int64_t, double*, int64_t, int64_t, double*, int64_t, int64_t, int,
              void*, int64_t, double, int64_t*, hipStream_t, double* dropped2_host = nullptr, int frob_exit = 0,
              int64_t* pivot_perm_host = nullptr, double* nf_out2 = nullptr,
              int* nf_done = nullptr);
int64_t qr_ws_bytes(int64_t, int64_t, int);
int svd_trunc(hipStream_t, const double*, int64_t, int64_t, int64_t, int64_t, int64_t, double, double*, int64_t, int64_t, double*,
              double*, int64_t, int64_t, int64_t*, double*, int*, int*, void*, int64_t);
int svd_vals(hipStream_t, const double*, int64_t, int64_t, int64_t, int64_t, double*, int*, int*, void*, int64_t);
int64_t svd_ws_bytes(int64_t, int64_t, int);

constexpr int MAX_SIDE = 8;

struct ForkJoin {                        // per host thread, created on first use, destroyed with the thread
    hipEvent_t fork = nullptr, done[MAX_SIDE] = {};
    bool ok = false;
    bool init() {
        if (ok) return true;
        if (hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess) return false;
        for (int i = 0; i < MAX_SIDE; ++i)
            if (hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess) return false;
        return ok = true;
    }
    ~ForkJoin() {
        if (!ok) return;
        (void)hipEventDestroy(fork);
        for (int i = 0; i < MAX_SIDE; ++i) (void)hipEventDestroy(done[i]);
    }
};

int qr_batched(hipStream_t st, double* A, int64_t rs, int64_t cs, int64_t m, int64_t n, double* Q, int64_t qrs, int64_t qcs, double* R,
               int64_t rrs, int64_t rcs, int nb, double rank_tol, int64_t* keff_host, int64_t batch, int64_t bsA, int64_t bsQ,
               int64_t bsR, void* ws, int64_t ws_bytes, void* const* side, int nside) {
    if (batch == 0) return 0;
    TN_CHECK_ARG(batch >= 1, "negative batch");
    TN_CHECK_ARG(nside >= 0 && nside <= MAX_SIDE && (nside == 0 || side != nullptr), "0..8 side streams");
    const int64_t wsi = align_up(qr_ws_bytes(m, n, nb), 256);
    TN_CHECK_ARG(ws_bytes >= wsi * batch, "workspace too small (batch x tn_qr_ws_bytes, each rounded up to 256 B)");
    thread_local ForkJoin fj;
    const int used = (int)(nside < batch ? nside : batch);
    hipError_t e;
    if (used > 0) {
        TN_CHECK_ARG(fj.init(), "event creation failed");
        if ((e = hipEventRecord(fj.fork, st)) != hipSuccess) return hip_fail(e, "record fork");
        for (int s = 0; s < used; ++s)
            if ((e = hipStreamWaitEvent((hipStream_t)side[s], fj.fork, 0)) != hipSuccess) return hip_fail(e, "wait fork");
    }
    int rc = 0;
    {
        // the items' own checks for launches that gave up at an in-kernel barrier would each synchronise its stream and so undo the
        // concurrency this entry point exists for: they are deferred, and every stream that was used is asked once after the join
        FusedDeferCheck defer;
        for (int64_t i = 0; i < batch && rc == 0; ++i) {
            hipStream_t si = used > 0 ? (hipStream_t)side[i % used] : st;
            rc = qr_factor(si, A + i * bsA, rs, cs, m, n, Q + i * bsQ, qrs, qcs, R + i * bsR, rrs, rcs, nb, (char*)ws + i * wsi, wsi, rank_tol,
                           keff_host ? keff_host + i : nullptr, nullptr);
        }
    }
    for (int s = 0; s < used; ++s) {      // join even after an error so that the caller's stream stays ordered
        if ((e = hipEventRecord(fj.done[s], (hipStream_t)side[s])) != hipSuccess) return hip_fail(e, "record join");
        if ((e = hipStreamWaitEvent(st, fj.done[s], 0)) != hipSuccess) return hip_fail(e, "wait join");
    }
    if (fused_check_needed()) {
        int total = 0;
        for (int s = 0; s < (used > 0 ? used : 1); ++s) {
            int gave_up = 0;
            const int rc2 = fused_timeouts(used > 0 ? (hipStream_t)side[s] : st, &gave_up);
            if (rc2) return rc2;
            total += gave_up;
        }
        if (total > 0) {
            set_error("tn_qr_batched: %d launch(es) with in-kernel barriers gave up; the results are invalid and the inputs may have been "
                      "overwritten -- rerun from copies (these streams now take the multi-launch forms)", total);
            return -7;
        }
    }
    return rc;
}

int svd_trunc_batched(hipStream_t st, const double* C, int64_t crs, int64_t ccs, int64_t k, int64_t n, int64_t Dmax, double tol, double* U,
                      int64_t urs, int64_t ucs, double* S, double* Vt, int64_t vrs, int64_t vcs, int64_t* keep_host,
                      double* discarded_host, int* sweeps_host, int* info_host, int64_t batch, int64_t bsC, int64_t bsU, int64_t bsS,
                      int64_t bsV, void* ws, int64_t ws_bytes) {
    if (batch == 0) return 0;
    TN_CHECK_ARG(batch >= 1, "negative batch");
    const int64_t wsi = align_up(svd_ws_bytes(k, n, 1), 256);
    TN_CHECK_ARG(ws_bytes >= wsi, "workspace too small");
    for (int64_t i = 0; i < batch; ++i) {  // items share one workspace: they run one after the other (rank read-backs)
        const int rc = svd_trunc(st, C + i * bsC, crs, ccs, k, n, Dmax, tol, U + i * bsU, urs, ucs, S + i * bsS, Vt + i * bsV, vrs, vcs,
                                 keep_host + i, discarded_host ? discarded_host + i : nullptr, sweeps_host ? sweeps_host + i : nullptr,
                                 info_host ? info_host + i : nullptr, ws, ws_bytes);
        if (rc) return rc;
    }
    return 0;
}

int svd_vals_batched(hipStream_t st, const double* C, int64_t crs, int64_t ccs, int64_t k, int64_t n, double* S_host, int* sweeps_host,
                     int* info_host, int64_t batch, int64_t bsC, void* ws, int64_t ws_bytes) {
    if (batch == 0) return 0;
    TN_CHECK_ARG(batch >= 1, "negative batch");
    const int64_t kn = k < n ? k : n;
    for (int64_t i = 0; i < batch; ++i) {
        const int rc = svd_vals(st, C + i * bsC, crs, ccs, k, n, S_host + i * kn, sweeps_host ? sweeps_host + i : nullptr,
                                info_host ? info_host + i : nullptr, ws, ws_bytes);
        if (rc) return rc;
    }
    return 0;
}

}  // namespace tn
