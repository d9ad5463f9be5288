// Host-free chain driver: MPS.apply_mpo + MPS.compress_mps (reference mps.py:353-359, 175-200) for ONE boundary MPS in ONE library
// call.  The Python driver (tnac4o_amd/mps.py) issues ~600 library calls and as many torch allocations per row from under the GIL; with
// the four lattice rotations of an instance on four host threads that host work, not the GPU, sets the pace.  Here the
// same steps -- the same kernels on the same operands in the same order, so the results are bit-identical to the Python driver's
// (tests/test_gpu_mps.py) -- are walked in C++ on the caller's thread without the GIL (ctypes releases it for the call), with every
// tensor carved out of ONE caller-owned arena by a host-side first-fit allocator (the stream is in order, so a block may be reused
// as soon as the host has enqueued its last reader).  Nothing is allocated on the device by the library.
//
//   tn_compress_mps        apply_mpo (optional, K1) + compress_mps: weighted rank-revealing first canonisation pass
//                          (MPS.canonise_right_weighted), SVD initialisation with graduated truncation, variational sweeps with lazy /
//                          batched Schmidt-value checks -- every decision the Python driver takes on the host is taken here on the host
//   tn_argsort_desc, tn_weighted_sum   the two small reductions of the weighted pass that the Python driver used torch for (their
//                          summation / tie-breaking order is part of the result, so both drivers now call these)
#include <math.h>
#include <stdlib.h>

#include <cmath>

#include <algorithm>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "common.h"

namespace tn {

// ---- other translation units ----------------------------------------------------------------------------------------------
int absorb(hipStream_t, const double*, const double*, double*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int, int64_t,
           int64_t, int64_t, int64_t);
int qr_factor(hipStream_t, double*, int64_t, int64_t, int64_t, int64_t, double*, int64_t, int64_t, double*, int64_t, int64_t, int, void*,
              int64_t, double, int64_t*, hipStream_t, double* dropped2_host = nullptr, int frob_exit = 0, int64_t* pivot_perm_host = nullptr, double* nf_out2 = nullptr,
              int* nf_done = nullptr);
int64_t qr_ws_bytes(int64_t, int64_t, int);
int copy_mat(hipStream_t, const double*, int64_t, int64_t, double*, int64_t, int64_t, int64_t, int64_t);
int svd_trunc(hipStream_t, const double*, int64_t, int64_t, int64_t, int64_t, int64_t, double, double*, int64_t, int64_t, double*, double*,
              int64_t, int64_t, int64_t*, double*, int*, int*, void*, int64_t);
int svd_vals(hipStream_t, const double*, int64_t, int64_t, int64_t, int64_t, double*, int*, int*, void*, int64_t);
int64_t svd_ws_bytes(int64_t, int64_t, int);
int svd_vals_small_batched(hipStream_t, const int64_t*, int64_t, double*);
int normalize_pow2(hipStream_t, double*, int64_t, double*, void*, int64_t);
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

// ---- small kernels --------------------------------------------------------------------------------------------------------
// out (n0, n1, n2, n3) contiguous  <-  in[i0 s0 + i1 s1 + i2 s2 + i3 s3]     (torch's permute(...).contiguous())
__global__ __launch_bounds__(256) void permute4_kernel(const double* __restrict__ in, int64_t s0, int64_t s1, int64_t s2, int64_t s3, int64_t n0,
                                                       int64_t n1, int64_t n2, int64_t n3, double* __restrict__ out) {
    const int64_t tot = n0 * n1 * n2 * n3;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < tot; e += (int64_t)gridDim.x * 256) {
        const int64_t i3 = e % n3, r = e / n3, i2 = r % n2, r2 = r / n2, i1 = r2 % n1, i0 = r2 / n1;
        out[e] = in[i0 * s0 + i1 * s1 + i2 * s2 + i3 * s3];
    }
}
static int permute4(hipStream_t st, const double* in, int64_t s0, int64_t s1, int64_t s2, int64_t s3, int64_t n0, int64_t n1, int64_t n2, int64_t n3,
                    double* out) {
    const int64_t tot = n0 * n1 * n2 * n3;
    if (tot <= 0) return 0;
    int64_t nb = cdiv(tot, 256 * 4);
    if (nb > 4096) nb = 4096;
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(permute4_kernel, dim3((unsigned)nb), dim3(256), 0, st, in, s0, s1, s2, s3, n0, n1, n2, n3, out));
    TN_CHECK_LAUNCH("permute4_kernel");
    return 0;
}

__global__ __launch_bounds__(256) void fill_kernel(double* __restrict__ x, int64_t n, double v) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] = v;
}
static int fill(hipStream_t st, double* x, int64_t n, double v) {
    if (n <= 0) return 0;
    int64_t nb = cdiv(n, 1024);
    if (nb > 1024) nb = 1024;
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(fill_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, n, v));
    TN_CHECK_LAUNCH("fill_kernel");
    return 0;
}

// w[i] = a[i] * b[i];  sum[0] = sum_i w[i] in a FIXED order (thread t adds i = t, t + 256, ...; then a binary tree over the threads)
__global__ __launch_bounds__(256) void weighted_sum_kernel(const double* __restrict__ a, const double* __restrict__ b, int64_t n, double* __restrict__ w,
                                                           double* __restrict__ sum) {
    __shared__ double red[256];
    const int tid = threadIdx.x;
    double s = 0.0;
    for (int64_t i = tid; i < n; i += 256) { const double x = a[i] * b[i]; w[i] = x; s += x; }
    red[tid] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (tid < k) red[tid] += red[tid + k];
        __syncthreads();
    }
    if (tid == 0) sum[0] = red[0];
}
int weighted_sum(hipStream_t st, const double* a, const double* b, int64_t n, double* w, double* sum) {
    TN_CHECK_ARG(n >= 1, "empty input");
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(256), 0, st, a, b, n, w, sum));
    TN_CHECK_LAUNCH("weighted_sum_kernel");
    return 0;
}

// perm[rank(i)] = i with rank(i) = #{ j : w[j] before w[i] } in the strict total order "larger value first, NaN before everything,
// equal values by increasing index" -- a stable descending argsort by counting (n is a bond dimension: <= a few thousand)
// (one workgroup per element: its 256 threads share the comparisons, so a 1024-key sort is one wave of short workgroups)
__global__ __launch_bounds__(256) void argsort_desc_kernel(const double* __restrict__ w, int64_t n, int64_t* __restrict__ perm) {
    __shared__ int red[4];
    const int tid = threadIdx.x;
    const int64_t i = blockIdx.x;
    const double mine = w[i];
    const bool mine_nan = !(mine == mine);
    int cnt = 0;
    for (int64_t j = tid; j < n; j += 256) {
        const double x = w[j];
        const bool x_nan = !(x == x);
        bool before;
        if (x_nan || mine_nan) before = x_nan && (!mine_nan || j < i);
        else before = (x > mine) || (x == mine && j < i);
        cnt += before ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) perm[(red[0] + red[1]) + (red[2] + red[3])] = i;
}
int argsort_desc(hipStream_t st, const double* w, int64_t n, int64_t* perm) {
    TN_CHECK_ARG(n >= 1 && n <= (1 << 20), "length out of range");
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(argsort_desc_kernel, dim3((unsigned)n), dim3(256), 0, st, w, n, perm));
    TN_CHECK_LAUNCH("argsort_desc_kernel");
    return 0;
}

// the reference's sign convention (mps.py:35-39) on a finished factorisation U (k x r, strides), Vt (r x n, strides): flip the pairs
// (column of U, row of Vt) in which the most negative entry outweighs the most positive one in both.  One workgroup per pair.
__global__ __launch_bounds__(256) void sign_gauge_kernel(double* __restrict__ U, int64_t urs, int64_t ucs, int64_t k, double* __restrict__ Vt, int64_t vrs,
                                                         int64_t vcs, int64_t n) {
    __shared__ double rmin[256], rmax[256];
    __shared__ int flip;
    const int j = blockIdx.x, tid = threadIdx.x;
    double lo = 1e308, hi = -1e308;
    for (int64_t i = tid; i < k; i += 256) { const double x = U[i * urs + j * ucs]; lo = fmin(lo, x); hi = fmax(hi, x); }
    rmin[tid] = lo; rmax[tid] = hi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) { rmin[tid] = fmin(rmin[tid], rmin[tid + s]); rmax[tid] = fmax(rmax[tid], rmax[tid + s]); } __syncthreads(); }
    const double umin = rmin[0], umax = rmax[0];
    __syncthreads();
    lo = 1e308; hi = -1e308;
    for (int64_t i = tid; i < n; i += 256) { const double x = Vt[j * vrs + i * vcs]; lo = fmin(lo, x); hi = fmax(hi, x); }
    rmin[tid] = lo; rmax[tid] = hi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) { rmin[tid] = fmin(rmin[tid], rmin[tid + s]); rmax[tid] = fmax(rmax[tid], rmax[tid + s]); } __syncthreads(); }
    if (tid == 0) flip = (fabs(umin) > umax && fabs(rmin[0]) > rmax[0]) ? 1 : 0;
    __syncthreads();
    if (!flip) return;
    for (int64_t i = tid; i < k; i += 256) U[i * urs + j * ucs] = -U[i * urs + j * ucs];
    for (int64_t i = tid; i < n; i += 256) Vt[j * vrs + i * vcs] = -Vt[j * vrs + i * vcs];
}

// ---- arena ------------------------------------------------------------------------------------------------------------------
// Host-side first-fit allocator over the caller's buffer.  Everything runs on one in-order stream, so a block can be handed out again
// as soon as the host has ENQUEUED the last kernel that touches it.
class Arena {
public:
    Arena(char* base, int64_t size) : base_(base), size_(size) { free_[0] = size; }
    char* alloc(int64_t bytes) {
        bytes = align_up(bytes > 0 ? bytes : 1, 256);
        for (auto it = free_.begin(); it != free_.end(); ++it) {
            if (it->second >= bytes) {
                const int64_t off = it->first, rest = it->second - bytes;
                free_.erase(it);
                if (rest > 0) free_[off + bytes] = rest;
                used_[off] = bytes;
                in_use_ += bytes;
                if (in_use_ > peak_) peak_ = in_use_;
                return base_ + off;
            }
        }
        return nullptr;
    }
    void release(char* p) {
        const int64_t off = p - base_;
        auto u = used_.find(off);
        if (u == used_.end()) return;
        int64_t start = off, len = u->second;
        in_use_ -= len;
        used_.erase(u);
        auto nx = free_.lower_bound(start);
        if (nx != free_.end() && nx->first == start + len) { len += nx->second; nx = free_.erase(nx); }
        if (nx != free_.begin()) {
            auto pv = std::prev(nx);
            if (pv->first + pv->second == start) { start = pv->first; len += pv->second; free_.erase(pv); }
        }
        free_[start] = len;
    }
    // give the tail of a block back (keeps the first `bytes`)
    void shrink(char* p, int64_t bytes) {
        const int64_t off = p - base_;
        auto u = used_.find(off);
        if (u == used_.end()) return;
        bytes = align_up(bytes > 0 ? bytes : 1, 256);
        if (bytes >= u->second) return;
        const int64_t tail = off + bytes, tlen = u->second - bytes;
        u->second = bytes;
        used_[tail] = tlen;
        release(base_ + tail);
    }
    int64_t peak() const { return peak_; }
    int64_t size() const { return size_; }
private:
    char* base_;
    int64_t size_, in_use_ = 0, peak_ = 0;
    std::map<int64_t, int64_t> free_, used_;
};

struct Block {                  // owner of one arena block (never copied: the destructor gives the block back)
    Arena* ar;
    char* p;
    Block(Arena* a, char* q) : ar(a), p(q) {}
    Block(const Block&) = delete;
    Block& operator=(const Block&) = delete;
    ~Block() { if (ar && p) ar->release(p); }
};
using Ref = std::shared_ptr<Block>;

struct T3 {                     // site tensor (a, b, c) contiguous; also used for matrices (a = 1)
    Ref blk;                    // owner (null: memory of the caller, never freed here)
    double* p = nullptr;
    int64_t a = 0, b = 0, c = 0;
    int64_t numel() const { return a * b * c; }
};
struct M2 {                     // matrix (r x c), contiguous
    Ref blk;
    double* p = nullptr;
    int64_t r = 0, c = 0;
};

#define CH(expr) do { const int rc__ = (expr); if (rc__) return rc__; } while (0)

constexpr double CH_EPS = 2.220446049250313e-16;
constexpr double CH_RANK_TOL = 1.3877787807814457e-17;      // 2^-56 (ops.RANK_TOL)
constexpr double CH_PASS1_ACCEPT = 1.3877787807814457e-17;  // 2^-56
constexpr double CH_PASS1_FLOOR = 1e-14;
constexpr int64_t CH_PASS1_MIN_BOND = 256;

// TN_CHAIN_PASSES=1 (diagnostics): the stream is synchronised at every pass boundary of compress_mps and the wall time booked per pass;
// table at exit (the synchronisations cost a few microseconds per pass: not for timed runs).
namespace {
struct PassClock {
    bool on;
    std::mutex mu;
    double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double tr[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};      // truncating passes 2 / 3 / 4: site factorisation | deflation | decomposition | projectors
    double sub[4] = {0, 0, 0, 0};      // inside pass 1: Gram recursion + weights | attach, norms, sort, gather | pivoted factorisation | un-weighting, normalisation
    long calls = 0;
    PassClock() { const char* e = getenv("TN_CHAIN_PASSES"); on = e && e[0] == '1'; }
    ~PassClock() {
        if (!on || !calls) return;
        static const char* nm[8] = {"absorb", "pass 1 (canonise_right, weighted)", "pass 2 (canonise_left, 4 chi)", "variational, 1 sweep",
                                    "pass 3 (canonise_right, 2 chi)", "pass 4 (canonise_left, chi)", "variational, final", "results"};
        double tot = 0.0;
        for (double v : ms) tot += v;
        fprintf(stderr, "[tn_compress_mps passes] %ld calls, %.1f ms in total\n", calls, tot);
        for (int i = 0; i < 8; ++i) fprintf(stderr, "  %-36s %9.1f ms  %5.1f %%\n", nm[i], ms[i], 100.0 * ms[i] / (tot > 0 ? tot : 1));
        for (int q = 0; q < 3; ++q)
            fprintf(stderr, "  inside pass %d: site factorisations %.1f ms | bond deflation %.1f ms | decompositions %.1f ms | projectors %.1f ms\n", q + 2, tr[q][0], tr[q][1],
                    tr[q][2], tr[q][3]);
        fprintf(stderr, "  inside pass 1: Gram recursion + weights %.1f ms | attach, norms, sort, gather %.1f ms | pivoted factorisation %.1f ms | un-weighting, "
                "normalisation %.1f ms\n", sub[0], sub[1], sub[2], sub[3]);
    }
};
PassClock g_pass_clock;
struct PassMark {
    hipStream_t st;
    std::chrono::steady_clock::time_point t0;
    explicit PassMark(hipStream_t s) : st(s) { if (g_pass_clock.on) { (void)hipStreamSynchronize(st); t0 = std::chrono::steady_clock::now(); } }
    void sublap(int k) {
        if (!g_pass_clock.on) return;
        (void)hipStreamSynchronize(st);
        const auto t1 = std::chrono::steady_clock::now();
        std::lock_guard<std::mutex> lk(g_pass_clock.mu);
        g_pass_clock.sub[k] += std::chrono::duration<double, std::milli>(t1 - ts).count();
        ts = t1;
    }
    void trlap(int pass, int k) {
        if (!g_pass_clock.on || pass < 2 || pass > 4) return;
        (void)hipStreamSynchronize(st);
        const auto t1 = std::chrono::steady_clock::now();
        std::lock_guard<std::mutex> lk(g_pass_clock.mu);
        g_pass_clock.tr[pass - 2][k] += std::chrono::duration<double, std::milli>(t1 - ts).count();
        ts = t1;
    }
    void substart() { if (g_pass_clock.on) { (void)hipStreamSynchronize(st); ts = std::chrono::steady_clock::now(); } }
    std::chrono::steady_clock::time_point ts;
    void lap(int k) {
        if (!g_pass_clock.on) return;
        (void)hipStreamSynchronize(st);
        const auto t1 = std::chrono::steady_clock::now();
        std::lock_guard<std::mutex> lk(g_pass_clock.mu);
        g_pass_clock.ms[k] += std::chrono::duration<double, std::milli>(t1 - t0).count();
        if (k == 7) g_pass_clock.calls += 1;
        t0 = t1;
    }
};
}  // namespace

class Chain {
public:
    Chain(hipStream_t st, Arena& ar, int64_t L) : st(st), ar(ar), L(L), A(L), D(L + 1, 1), discarded(L + 1, 0.0), R(L + 2), Sst(L + 1) {}
    hipStream_t st;
    Arena& ar;
    int64_t L;
    std::vector<T3> A;
    std::vector<int64_t> D;
    M2 C;
    int64_t pC = 0;
    std::vector<double> discarded;
    std::vector<M2> R;                       // mixed environments (index 0 .. L; the overlap lives in `overlap`)
    double overlap = 0.0;
    double* nfs_dev = nullptr;               // caller's table of [nf, 1/nf] pairs
    int64_t nfs_cap = 0, nfs_count = 0;
    double reveal_error_bound = 0.0;
    int reveal_fallbacks = 0;
    int weighted_used = 0;
    int64_t bonds_before = 0, bonds_after = 0;   // sum of the bond dimensions before / after the first canonisation pass
    // absorbed factors for the structured Gram recursion (may be null per site)
    std::vector<const double*> facA, facW;
    std::vector<int64_t> facdims;            // L x 7: Dl, ps, Dr, ba, po, bb, pi
    int hconj = 0;
    // Schmidt values per bond: concrete host values, or a centre matrix whose decomposition nobody has asked for yet (mps._LazyS)
    struct SState { std::vector<double> val; bool has = false; M2 lazy; bool is_lazy = false; };
    std::vector<SState> Sst;

    // ---- memory ----
    int out_of_memory(const char* what, int64_t bytes) {
        set_error("tn_compress_mps: arena too small (%s needs %lld more bytes; arena %lld, peak %lld)", what, (long long)bytes, (long long)ar.size(),
                  (long long)ar.peak());
        return -3;
    }
    int new_block(int64_t doubles, Ref& ref, double*& p, const char* what) {
        char* q = ar.alloc(doubles * 8);
        if (!q) return out_of_memory(what, doubles * 8);
        ref = std::make_shared<Block>(&ar, q);
        p = (double*)q;
        return 0;
    }
    int new_t3(int64_t a, int64_t b, int64_t c, T3& t, const char* what) {
        t.a = a; t.b = b; t.c = c;
        return new_block(a * b * c, t.blk, t.p, what);
    }
    int new_m2(int64_t r, int64_t c, M2& m, const char* what) {
        m.r = r; m.c = c;
        return new_block(r * c, m.blk, m.p, what);
    }
    // persistent scratch slots (ops.workspace): 0 = QR / SVD, 1 = GEMM split-K and site steps, 3 = normalize_pow2
    struct Scratch { Ref blk; char* p = nullptr; int64_t bytes = 0; } ws_[4];
    int scratch(int slot, int64_t bytes, void*& out) {
        Scratch& s = ws_[slot];
        if (s.bytes < bytes) {
            s.blk.reset();
            const int64_t want = bytes + bytes / 4 + 4096;
            char* q = ar.alloc(want);
            int64_t got = want;
            if (!q) { q = ar.alloc(bytes); got = bytes; }
            if (!q) return out_of_memory("scratch", bytes);
            s.blk = std::make_shared<Block>(&ar, q);
            s.p = q;
            s.bytes = got;
        }
        out = s.p;
        return 0;
    }
    bool nfs_overflow = false;               // the caller's table was too small: the call fails (checked at its end)
    double* next_nf() {                      // slot of the next [nf, 1/nf] pair in the caller's table
        if (nfs_count >= nfs_cap) { nfs_overflow = true; return nfs_dev + 2 * (nfs_cap - 1); }
        return nfs_dev + 2 * (nfs_count++);
    }

    // ---- GEMM wrappers (ops.mm / ops.bmm) ----
    int mm(int64_t M, int64_t N, int64_t K, const double* a, int64_t rsa, int64_t csa, const double* b, int64_t rsb, int64_t csb, double* c,
           int64_t rsc, int64_t csc) {
        const int64_t wsb = gemm_ws_bytes(M, N, K, 1);
        void* w = nullptr;
        if (wsb > 0) CH(scratch(1, wsb, w));
        return gemm(st, M, N, K, 1.0, a, rsa, csa, b, rsb, csb, 0.0, c, rsc, csc, 1, 0, 0, 0, (double*)w, wsb);
    }
    int bmm(int64_t batch, int64_t M, int64_t N, int64_t K, const double* a, int64_t rsa, int64_t csa, int64_t bsa, const double* b, int64_t rsb,
            int64_t csb, int64_t bsb, double* c, int64_t rsc, int64_t csc, int64_t bsc) {
        if (batch <= 0) return 0;
        return gemm(st, M, N, K, 1.0, a, rsa, csa, b, rsb, csb, 0.0, c, rsc, csc, batch, bsa, bsb, bsc, nullptr, 0);
    }
    int normalize(double* x, int64_t n, double* nf2) {
        void* sc = nullptr;
        CH(scratch(3, 8192, sc));
        return normalize_pow2(st, x, n, nf2, sc, 8192);
    }
    int ones11(M2& m) {
        CH(new_m2(1, 1, m, "1 x 1 centre"));
        return fill(st, m.p, 1, 1.0);
    }
    int sync(const char* what) {
        const hipError_t e = hipStreamSynchronize(st);
        return e == hipSuccess ? 0 : hip_fail(e, what);
    }
    int d2h(void* host, const void* dev, size_t bytes) {
        const hipError_t e = hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st);
        return e == hipSuccess ? 0 : hip_fail(e, "device to host copy");
    }
    int h2d(void* dev, const void* host, size_t bytes) {
        const hipError_t e = hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st);
        return e == hipSuccess ? 0 : hip_fail(e, "host to device copy");
    }

    // ---- one canonisation step (ops.site_qr + MPS._site_left / _site_right) ----
    // side 0: Q (m x k) row-major, Rm (k x n);  side 1: Q = Q^T (k x m), Rm = R^T (n x k).  Cm == nullptr: no attach, A[n] is consumed.
    struct SiteOut { T3 Q; M2 Rm; int64_t k = 0; double dropped2 = 0.0; std::vector<int64_t> piv; };
    int site_qr_step(int side, const T3& Ain, const M2* Cm, double rank_tol, bool normalise, bool frob_exit, bool pivot, SiteOut& o) {
        const int64_t Dl = Ain.a, p = Ain.b, Dr = Ain.c;
        const bool attach = Cm != nullptr;
        const int64_t kc = attach ? (side == 0 ? Cm->r : Cm->c) : 0;
        if (attach && (side == 0 ? Cm->c != Dl : Cm->r != Dr)) { set_error("tn_compress_mps: centre matrix does not fit the site"); return -1; }
        int64_t m, n;
        if (side == 0) { m = (attach ? kc : Dl) * p; n = Dr; } else { m = p * (attach ? kc : Dr); n = Dl; }
        const int64_t kf = m < n ? m : n;
        Ref qb, rb;
        double *Q = nullptr, *Rp = nullptr;
        CH(new_block(m * kf, qb, Q, "Q factor"));
        CH(new_block(kf * n, rb, Rp, "R factor"));
        const int64_t wsb = site_qr_ws_bytes(side, Dl, p, Dr, kc, attach ? 1 : 0);
        void* w = nullptr;
        CH(scratch(0, wsb, w));
        int64_t keff = kf;
        int normd = 0;
        double drop2 = 0.0;
        double* nf = normalise ? next_nf() : nullptr;
        if (pivot) o.piv.assign((size_t)n, 0);
        CH(site_qr(st, side, Ain.p, Dl, p, Dr, attach ? Cm->p : nullptr, kc, Q, Rp, rank_tol, &keff, nf, &normd, w, wsb, &drop2, frob_exit ? 1 : 0,
                   pivot ? o.piv.data() : nullptr));
        const int64_t k = keff;
        o.k = k;
        o.dropped2 = drop2;
        if (k < kf) {                        // rank-revealing early exit: slice the factors (Q[:, :k] / R[:k] resp. Q[:k] / R[:, :k])
            if (side == 0) {
                Ref q2; double* Q2 = nullptr;
                CH(new_block(m * k, q2, Q2, "Q factor (sliced)"));
                CH(copy_mat(st, Q, kf, 1, Q2, k, 1, m, k));
                qb = q2; Q = Q2;
                ar.shrink((char*)Rp, k * n * 8);
            } else {
                ar.shrink((char*)Q, k * m * 8);
                Ref r2; double* R2 = nullptr;
                CH(new_block(n * k, r2, R2, "R factor (sliced)"));
                CH(copy_mat(st, Rp, kf, 1, R2, k, 1, n, k));
                rb = r2; Rp = R2;
            }
        }
        if (normalise && !normd) CH(normalize(Rp, k * n, nf));
        o.Q.blk = qb; o.Q.p = Q;
        if (side == 0) { o.Q.a = m / p; o.Q.b = p; o.Q.c = k; o.Rm.r = k; o.Rm.c = n; }
        else { o.Q.a = k; o.Q.b = p; o.Q.c = m / p; o.Rm.r = n; o.Rm.c = k; }
        o.Rm.blk = rb; o.Rm.p = Rp;
        return 0;
    }
    int site_left(int64_t n, const M2* Cm, double rank_tol) {
        SiteOut o;
        const T3 Ain = A[n];
        CH(site_qr_step(0, Ain, Cm, rank_tol, true, false, false, o));
        if (o.Rm.r == 1 && o.Rm.c == 1) CH(ones11(o.Rm));          // mps.py:778-780: the norm of a 1 x 1 centre is dropped
        A[n] = o.Q;
        C = o.Rm;
        D[n] = A[n].a; D[n + 1] = o.k;
        pC = n + 1;
        return 0;
    }
    M2 c11;                                                      // the normalised 1 x 1 centre of the last right step (weighted pass)
    int site_right(int64_t n, const M2* Cm, double rank_tol) {
        SiteOut o;
        const T3 Ain = A[n];
        CH(site_qr_step(1, Ain, Cm, rank_tol, true, false, false, o));
        if (o.Rm.r == 1 && o.Rm.c == 1) { c11 = o.Rm; CH(ones11(o.Rm)); }
        A[n] = o.Q;
        C = o.Rm;
        D[n] = o.k; D[n + 1] = A[n].c;
        pC = n;
        return 0;
    }

    // ---- truncated SVD (ops.svd_trunc, with the QR-preconditioned retry) ----
    struct SvdOut { M2 U, Vt; Ref sblk; double* S = nullptr; int64_t keep = 0, cap = 0; double disc = 0.0; };
    int svd_raw(const double* Cp, int64_t crs, int64_t ccs, int64_t k, int64_t n, int64_t Dmax, double tol, SvdOut& o, int& info, int& sweeps) {
        const int64_t cap = std::min(std::min(k, n), Dmax);
        o.cap = cap;
        CH(new_m2(k, cap, o.U, "U"));
        CH(new_block(cap, o.sblk, o.S, "S"));
        CH(new_m2(cap, n, o.Vt, "Vt"));
        const int64_t wsb = svd_ws_bytes(k, n, 1);
        void* w = nullptr;
        CH(scratch(0, wsb, w));
        int64_t keep = 0;
        double disc = 0.0;
        {
            ProfPhase ph(PH_SVD);
            CH(svd_trunc(st, Cp, crs, ccs, k, n, cap, tol, o.U.p, cap, 1, o.S, o.Vt.p, n, 1, &keep, &disc, &sweeps, &info, w, wsb));
            const double dm = (double)(k > n ? k : n), dn = (double)(k > n ? n : k);
            prof_note(PROF_SVD_NOMINAL, 1, 14.0 * dm * dn * dn + 8.0 * dn * dn * dn, 8.0 * (2.0 * dm * dn + dn * dn + dn));
            prof_note(PROF_SVD_STREAM, sweeps, 0.0, (double)sweeps * (dn - 1.0) * 16.0 * dn * (dm + dn));
        }
        o.keep = keep;
        o.disc = disc;
        return 0;
    }
    int plain_qr(const double* T, int64_t rs, int64_t cs, int64_t m, int64_t n, M2& Q, M2& Rm) {      // ops.qr: T preserved
        const int64_t k = m < n ? m : n;
        Ref tb; double* Tc = nullptr;
        CH(new_block(m * n, tb, Tc, "QR input copy"));
        CH(copy_mat(st, T, rs, cs, Tc, n, 1, m, n));              // (torch clones with preserved strides; values are what matters)
        CH(new_m2(m, k, Q, "Q"));
        CH(new_m2(k, n, Rm, "R"));
        const int64_t wsb = qr_ws_bytes(m, n, 32);
        void* w = nullptr;
        CH(scratch(0, wsb, w));
        int64_t keff = k;
        ProfPhase ph(PH_QR);
        return qr_factor(st, Tc, n, 1, m, n, Q.p, k, 1, Rm.p, n, 1, 32, w, wsb, 0.0, &keff, nullptr);
    }
    // U (k x keep, leading dimension o.U.c), S, Vt (keep x n) of the centre matrix Cm
    int svd_trunc_full(const M2& Cm, int64_t Dmax, double tol, SvdOut& o) {
        int info = 0, sweeps = 0;
        CH(svd_raw(Cm.p, Cm.c, 1, Cm.r, Cm.c, Dmax, tol, o, info, sweeps));
        if (info == 0) return 0;
        // the sweep cap was reached: A = Q1 R1, R1^T = Q2 R2, Jacobi on R2, fold the orthogonal factors back (ops.svd_trunc)
        const int64_t k = Cm.r, n = Cm.c;
        const bool tall_is_C = k >= n;
        const int64_t tm = tall_is_C ? k : n, tn_ = tall_is_C ? n : k;
        M2 Q1, R1, Q2, R2;
        CH(plain_qr(Cm.p, tall_is_C ? Cm.c : 1, tall_is_C ? 1 : Cm.c, tm, tn_, Q1, R1));            // tall = Q1 R1  (R1: tn x tn)
        CH(plain_qr(R1.p, 1, R1.c, tn_, tn_, Q2, R2));                                              // R1^T = Q2 R2
        SvdOut o2;
        CH(svd_raw(R2.p, R2.c, 1, R2.r, R2.c, Dmax, tol, o2, info, sweeps));
        if (info != 0) { set_error("tn_compress_mps: Jacobi sweeps did not converge on a %lld x %lld matrix (%d sweeps, after QR preconditioning)", (long long)k, (long long)n, sweeps); return -4; }
        const int64_t kp = o2.keep, cap2 = o2.cap;
        // left = Q1 V2^T  (tm x kp),  right = U2^T Q2^T  (kp x tn)
        M2 left, right;
        CH(new_m2(tm, kp, left, "left vectors"));
        CH(new_m2(kp, tn_, right, "right vectors"));
        CH(mm(tm, kp, tn_, Q1.p, Q1.c, 1, o2.Vt.p, 1, tn_, left.p, kp, 1));
        CH(mm(kp, tn_, tn_, o2.U.p, 1, cap2, Q2.p, 1, Q2.c, right.p, tn_, 1));
        SvdOut res;
        res.keep = kp; res.cap = kp; res.disc = o2.disc; res.sblk = o2.sblk; res.S = o2.S;
        if (tall_is_C) { res.U = left; res.Vt = right; }
        else {                                                     // U = right^T (k x kp), Vt = left^T (kp x n), made contiguous
            CH(new_m2(k, kp, res.U, "U"));
            CH(new_m2(kp, n, res.Vt, "Vt"));
            CH(copy_mat(st, right.p, 1, tn_, res.U.p, kp, 1, k, kp));
            CH(copy_mat(st, left.p, 1, kp, res.Vt.p, n, 1, kp, n));
        }
        if (kp > 0) {
            TN_PROF_LAUNCH(st, PROF_SVD_AUX, hipLaunchKernelGGL(sign_gauge_kernel, dim3((unsigned)kp), dim3(256), 0, st, res.U.p, res.U.c, 1, k, res.Vt.p,
                               res.Vt.c, 1, n));
            TN_CHECK_LAUNCH("sign_gauge_kernel");
        }
        o = res;
        return 0;
    }
    int pass_id = 0;
    PassMark* trmark = nullptr;              // TN_CHAIN_PASSES: clock of the truncating pass in progress
    bool pass_truncated = false;             // a truncation of the current pass discarded more than rounding noise (> 32 eps S0)
    bool intermediate_pass = false;          // the 4 chi / 2 chi passes of graduate_truncation (their bonds are internal to compress_mps)
    int truncateC(int64_t Dmax, double tol) {
        if (!(0 < pC && pC < L)) return 0;
        const int64_t Dcap = std::min(Dmax, std::min(C.r, C.c));
        if (intermediate_pass && gauge_svd_skippable(Dmax, tol)) {
            gauge_skipped += 1;
            const int rcd = deflate_bond();
            if (trmark) trmark->trlap(pass_id, 1);
            return rcd;
        }
        SvdOut o;
        CH(svd_trunc_full(C, Dcap, tol, o));
        const int64_t keep = o.keep;
        if (o.disc > 32.0 * CH_EPS) pass_truncated = true;          // this pass has changed the state by more than rounding
        if (trmark) trmark->trlap(pass_id, 2);
        {   // TN_DEFLATE_TRACE=1 (diagnostics): what every truncation kept, next to what tn_bond_deflate would keep of the same bond
            static const bool trace = [] { const char* e = getenv("TN_DEFLATE_TRACE"); return e && e[0] == '1'; }();
            if (trace) {
                const int64_t k = pass_side == 0 ? C.r : C.c, n = pass_side == 0 ? C.c : C.r;
                int64_t kk = k;
                if (k >= 2 && k <= 256) {
                    T3& site = pass_side == 0 ? A[pC - 1] : A[pC];
                    const int64_t m = pass_side == 0 ? site.a * site.b : site.b * site.c;
                    M2 Cn; T3 Sn;
                    CH(new_m2(C.r, C.c, Cn, "trace")); CH(new_t3(site.a, site.b, site.c, Sn, "trace"));
                    void* w = nullptr;
                    CH(scratch(3, 8192, w));
                    double d2 = 0.0;
                    CH(bond_deflate(st, pass_side, C.p, k, n, site.p, m, Cn.p, Sn.p, &kk, &d2, w, 8192));
                }
                fprintf(stderr, "[deflate trace] pass %d bond %lld C %lld x %lld Dmax %lld svd keep %lld deflate keep %lld\n", pass_id, (long long)pC,
                        (long long)C.r, (long long)C.c, (long long)Dmax, (long long)keep, (long long)kk);
            }
        }
        if (keep <= 0) { set_error("tn_compress_mps: centre matrix at bond %lld is zero", (long long)pC); return -5; }
        const int64_t nl = pC - 1, nr = pC;
        const T3 Al = A[nl], Ar = A[nr];
        T3 Aln, Arn;
        M2 Cd;
        CH(new_t3(Al.a, Al.b, keep, Aln, "left site (truncated)"));
        CH(new_t3(keep, Ar.b, Ar.c, Arn, "right site (truncated)"));
        CH(new_m2(keep, keep, Cd, "diagonal centre"));
        const int64_t wsb = apply_truncation_ws_bytes(Al.a * Al.b, Al.c, keep, Ar.a, Ar.b * Ar.c);
        void* w = nullptr;
        CH(scratch(1, wsb > 256 ? wsb : 256, w));
        CH(apply_truncation(st, Al.p, Al.a * Al.b, Al.c, o.U.p, o.U.c, 1, keep, o.Vt.p, o.Vt.c, 1, Ar.p, Ar.a, Ar.b * Ar.c, o.S, Aln.p, Arn.p, Cd.p, w,
                            wsb));
        A[nl] = Aln; A[nr] = Arn; C = Cd;
        if (trmark) trmark->trlap(pass_id, 3);
        D[pC] = keep;
        discarded[pC] = std::max(discarded[pC], o.disc);
        return 0;
    }
    // A truncation that cannot truncate.  With min(C.shape) <= Dmax and tol <= eps the rule of mps.py:805-806 only removes singular values
    // below eps S0 -- rounding noise of the factorisations before it (LAPACK itself resolves a singular value to eps S0) -- and turns the
    // bond into the Schmidt basis.  Both are invisible outside an INTERMEDIATE pass of graduate_truncation: the next canonisation step
    // factors C A[n+1], whose triangular factor does not depend on an orthogonal change of the bond between A[n] and C A[n+1] (uniqueness of
    // QR with diag >= 0), the variational sweep is covariant under it, and the noise directions carry <= sqrt(k) eps of the state's norm,
    // which the next real truncation removes.  The centre matrix then simply stays with the next site (C = R, no projectors), the bond
    // keeps the rank the rank-revealing QR accepted.  The final pass (Dmax = chi) always decomposes: its bonds are the result.
    // TN_GAUGE_SVD=1 keeps every decomposition (A/B and the tests that compare the two forms).
    int64_t gauge_skipped = 0;
    int target_swapped = 0, var1_skipped = 0;
    int64_t attach_fused = 0;                // sites whose absorption went through the factors (never materialised)
    bool gauge_svd_skippable(int64_t Dmax, double tol) const {
        const int keep_mode = [] { const char* e = getenv("TN_GAUGE_SVD"); return e ? atoi(e) : 0; }();   // 1: all, 2: those of the 2 chi pass (read per call: the tests switch it)
        if (keep_mode == 1 || (keep_mode == 2 && pass_id == 3)) return false;
        return tol <= CH_EPS && std::min(C.r, C.c) <= Dmax;
    }
    // ... what is left to do at such a bond: drop the bond indices that carry nothing (tn_bond_deflate; TN_BOND_DEFLATE=0: keep them all)
    int pass_side = 0;                       // 0: left sweep (C = R, bond = rows of C / columns of A[pC-1]), 1: right sweep (C = R^T, bond = columns of C / rows of A[pC])
    int deflate_bond() {
        const bool off = [] { const char* e = getenv("TN_BOND_DEFLATE"); return e && e[0] == '0'; }();
        const int64_t k = pass_side == 0 ? C.r : C.c, n = pass_side == 0 ? C.c : C.r;
        if (off || k < 2 || k > 256) return 0;
        T3& site = pass_side == 0 ? A[pC - 1] : A[pC];
        const int64_t m = pass_side == 0 ? site.a * site.b : site.b * site.c;
        if ((pass_side == 0 ? site.c : site.a) != k) { set_error("tn_compress_mps: centre matrix does not fit its site"); return -1; }
        M2 Cn;
        T3 Sn;
        if (pass_side == 0) { CH(new_m2(k, n, Cn, "deflated centre")); CH(new_t3(site.a, site.b, k, Sn, "deflated site")); }
        else { CH(new_m2(n, k, Cn, "deflated centre")); CH(new_t3(k, site.b, site.c, Sn, "deflated site")); }
        void* w = nullptr;
        CH(scratch(3, 8192, w));
        int64_t kk = k;
        double d2 = 0.0;
        CH(bond_deflate(st, pass_side, C.p, k, n, site.p, m, Cn.p, Sn.p, &kk, &d2, w, 8192));
        if (kk == k) return 0;
        if (pass_side == 0) { Cn.r = kk; Sn.c = kk; ar.shrink((char*)Cn.p, kk * n * 8); ar.shrink((char*)Sn.p, m * kk * 8); }
        else { Cn.c = kk; Sn.a = kk; ar.shrink((char*)Cn.p, n * kk * 8); ar.shrink((char*)Sn.p, kk * m * 8); }
        site = Sn;
        C = Cn;
        D[pC] = kk;
        discarded[pC] = std::max(discarded[pC], std::sqrt(d2));
        bonds_deflated += k - kk;
        return 0;
    }
    int64_t bonds_deflated = 0;
    std::vector<M2> pass_C;                  // the centre matrix a truncating left pass leaves at every bond (index = bond)
    int canonise_left(bool compress, int64_t Dmax, double tol) {
        pass_side = 0;
        CH(ones11(C));
        pC = 0;
        pass_C.assign((size_t)L + 1, M2());
        for (int64_t n = 0; n < L; ++n) {
            const double rank_tol = (compress && 0 < n + 1 && n + 1 < L) ? CH_RANK_TOL : 0.0;
            const M2 Cm = C;
            PassMark tm(st);
            if (compress) tm.substart();
            CH(site_left(n, &Cm, rank_tol));
            if (compress) { tm.trlap(pass_id, 0); trmark = &tm; CH(truncateC(Dmax, tol)); trmark = nullptr; }
            if (compress && intermediate_pass) pass_C[pC] = C;
        }
        return 0;
    }
    int canonise_right(bool compress, int64_t Dmax, double tol) {
        pass_side = 1;
        CH(ones11(C));
        pC = L;
        for (int64_t n = L - 1; n >= 0; --n) {
            const double rank_tol = (compress && 0 < n && n < L) ? CH_RANK_TOL : 0.0;
            const M2 Cm = C;
            PassMark tm(st);
            if (compress) tm.substart();
            CH(site_right(n, &Cm, rank_tol));
            if (compress) { tm.trlap(pass_id, 0); trmark = &tm; CH(truncateC(Dmax, tol)); trmark = nullptr; }
        }
        return 0;
    }

    // ---- weighted rank-revealing first pass (MPS.canonise_right_weighted) ----
    int gram_step_structured(M2& G, int64_t n) {
        const int64_t* fd = &facdims[7 * n];
        const int64_t Dl = fd[0], ps = fd[1], Dr = fd[2], ba = fd[3], po = fd[4], bb = fd[5], pi = fd[6];
        const double* Af = facA[n];
        const double* W = facW[n];
        // Wl (l, s, r, t) as strides into W (ba, po, bb, pi): hconj: s = po, t = pi; else s = pi, t = po
        const int64_t wl = po * bb * pi, wr = pi;
        const int64_t wsd = hconj ? bb * pi : 1, wtd = hconj ? 1 : bb * pi;
        const int64_t pt = hconj ? pi : po;
        const int64_t na = Dl * ba;
        // Gp (alpha, l, l', alpha') <- G4 (alpha, l, alpha', l')
        Ref gpb; double* Gp = nullptr;
        CH(new_block(na * ba * Dl, gpb, Gp, "Gram (permuted)"));
        if (hconj) CH(permute4(st, G.p, ba * na, na, 1, ba, Dl, ba, ba, Dl, Gp));          // G[(alpha ba + l) na + alpha' ba + l']
        else CH(permute4(st, G.p, na, Dl * na, Dl, 1, Dl, ba, ba, Dl, Gp));                // G[(l Dl + alpha) na + l' Dl + alpha']
        Ref t1b; double* T1 = nullptr;
        CH(new_block(na * ba * ps * Dr, t1b, T1, "Gram step T1"));
        CH(mm(na * ba, ps * Dr, Dl, Gp, Dl, 1, Af, ps * Dr, 1, T1, ps * Dr, 1));
        gpb.reset();
        // Wx[(t, r'), (l', s')] = Wl[l', s', r', t]
        Ref wxb; double* Wx = nullptr;
        CH(new_block(pt * bb * ba * ps, wxb, Wx, "Wx"));
        CH(permute4(st, W, wtd, wr, wl, wsd, pt, bb, ba, ps, Wx));
        Ref t2b; double* T2 = nullptr;
        CH(new_block(na * pt * bb * Dr, t2b, T2, "Gram step T2"));
        CH(bmm(na, pt * bb, Dr, ba * ps, Wx, ba * ps, 1, 0, T1, Dr, 1, ba * ps * Dr, T2, Dr, 1, pt * bb * Dr));
        t1b.reset();
        // Wy[(s, r), (l, t)] = Wl[l, s, r, t]
        Ref wyb; double* Wy = nullptr;
        CH(new_block(ps * bb * ba * pt, wyb, Wy, "Wy"));
        CH(permute4(st, W, wsd, wr, wl, wtd, ps, bb, ba, pt, Wy));
        Ref ub; double* U = nullptr;
        CH(new_block(Dl * ps * bb * bb * Dr, ub, U, "Gram step U"));
        CH(bmm(Dl, ps * bb, bb * Dr, ba * pt, Wy, ba * pt, 1, 0, T2, bb * Dr, 1, ba * pt * bb * Dr, U, bb * Dr, 1, ps * bb * bb * Dr));
        t2b.reset();
        Ref ob; double* O = nullptr;
        CH(new_block(Dr * bb * bb * Dr, ob, O, "Gram step O"));
        CH(mm(Dr, bb * bb * Dr, Dl * ps, Af, 1, Dr, U, bb * bb * Dr, 1, O, bb * bb * Dr, 1));
        ub.reset();
        M2 Gn;
        CH(new_m2(Dr * bb, Dr * bb, Gn, "Gram matrix"));
        // O4 (beta, r, r', beta'): hconj out (beta, r, beta', r'); else out (r, beta, r', beta')
        if (hconj) CH(permute4(st, O, bb * bb * Dr, bb * Dr, 1, Dr, Dr, bb, Dr, bb, Gn.p));
        else CH(permute4(st, O, bb * Dr, bb * bb * Dr, Dr, 1, bb, Dr, bb, Dr, Gn.p));
        G = Gn;
        return 0;
    }
    // ---- absorption fused into the first contraction (round 5) ----
    // On the weighted path an absorbed bulk site (A (x) W, 134 MB at chi = 64) is only ever read by ONE product, the attach M_n = A'_n C.
    // With Hconj the absorbed index order is MPS-major, A'[(alpha l), t, (beta rb)] = sum_s A[alpha, s, beta] W[l, s, rb, t], and the attach
    // goes through the factors in two products that never form A':
    //     T[alpha, s, rb, r'] = sum_beta A[alpha, s, beta] C[(beta rb), r']                  (Dl ps) x (bb r) x Dr   -- C read as Dr x (bb r)
    //     M[(alpha l), t, r'] = sum_{s, rb} W[l, s, rb, t] T[alpha, s, rb, r']               batched over alpha: (ba pt) x r x (ps bb)
    // 3.2x fewer multiply-adds than the product with the absorbed tensor (K = 64 and 256 instead of 1024), no 134 MB written and read back
    // per site.  Such sites stay un-absorbed (`p == nullptr`, dimensions set) until something else asks for the tensor (ensure_absorbed:
    // the plain fallback pass).  TN_ATTACH_FUSED=0 absorbs every site up front (the round-4 form; the tests compare the two).
    int ensure_absorbed(int64_t n) {
        if (A[n].p) return 0;
        if (facA.empty() || !facA[n]) { set_error("tn_compress_mps: site %lld has neither a tensor nor its factors", (long long)n); return -1; }
        const int64_t* fd = &facdims[7 * n];
        T3 t;
        CH(new_t3(A[n].a, A[n].b, A[n].c, t, "absorbed site"));
        {
            ProfPhase ph(PH_ABSORB);
            CH(absorb(st, facA[n], facW[n], t.p, fd[0], fd[1], fd[2], fd[3], fd[4], fd[5], fd[6], hconj, 1, 0, 0, 0));
        }
        A[n] = t;
        return 0;
    }
    // Am (Dl_abs p x r) = A'_n C for an un-absorbed site, C: (Dr_abs x r).  Without Hconj the fused bonds are MPO-major,
    //     A'[(l alpha), t, (rb beta)] = sum_s W[l, t, rb, s] A[alpha, s, beta]:
    // the first product runs per rb (C read as bb matrices Dr x r, T written in the same (alpha, s, rb, r') layout), the second is the same
    // batched product with W[(l, t), (s, rb)], and its (alpha, l, t, r') result is moved to (l, alpha, t, r') by one more pass over it.
    int attach_through_factors(int64_t n, const M2& Cm, double* Am) {
        const int64_t* fd = &facdims[7 * n];
        const int64_t Dl = fd[0], ps = fd[1], Dr = fd[2], ba = fd[3], po = fd[4], bb = fd[5], pi = fd[6], r = Cm.c;
        const int64_t pt = hconj ? pi : po;                                     // hconj: s = po, t = pi; else s = pi, t = po
        if (Cm.r != Dr * bb || (hconj ? po : pi) != ps) { set_error("tn_compress_mps: centre matrix does not fit the factors of site %lld", (long long)n); return -1; }
        Ref tb; double* T = nullptr;
        CH(new_block(Dl * ps * bb * r, tb, T, "attach through the factors: T"));
        if (hconj) CH(mm(Dl * ps, bb * r, Dr, facA[n], Dr, 1, Cm.p, bb * r, 1, T, bb * r, 1));
        else CH(bmm(bb, Dl * ps, r, Dr, facA[n], Dr, 1, 0, Cm.p, r, 1, Dr * r, T, bb * r, 1, r));
        // Wq[(l, t), (s, rb)] = W[l, s, rb, t] (hconj) | W[l, t, rb, s]   (W: (ba, po, bb, pi) contiguous)
        Ref wb; double* Wq = nullptr;
        CH(new_block(ba * pt * ps * bb, wb, Wq, "attach through the factors: W"));
        if (hconj) CH(permute4(st, facW[n], po * bb * pi, 1, bb * pi, pi, ba, pt, ps, bb, Wq));
        else CH(permute4(st, facW[n], po * bb * pi, bb * pi, 1, pi, ba, pt, ps, bb, Wq));
        if (hconj) return bmm(Dl, ba * pt, r, ps * bb, Wq, ps * bb, 1, 0, T, r, 1, ps * bb * r, Am, r, 1, ba * pt * r);
        Ref mb; double* Mt = nullptr;
        CH(new_block(Dl * ba * pt * r, mb, Mt, "attach through the factors: M (MPS-major)"));
        CH(bmm(Dl, ba * pt, r, ps * bb, Wq, ps * bb, 1, 0, T, r, 1, ps * bb * r, Mt, r, 1, ba * pt * r));
        return permute4(st, Mt, pt * r, ba * pt * r, r, 1, ba, Dl, pt, r, Am);        // Mt (alpha, l, t, r') -> Am (l, alpha, t, r')
    }
    int canonise_right_weighted(bool& accepted) {
        struct Wt { Ref blk; double* d2 = nullptr; double* st65 = nullptr; bool on = false; };
        std::vector<Wt> wts(L + 1);
        std::vector<double*> gfac(L + 1, nullptr);
        M2 G;
        PassMark sm(st);
        sm.substart();
        CH(ones11(G));
        for (int64_t n = 0; n < L; ++n) {
            const int64_t Dl = A[n].a, p = A[n].b, Dr = A[n].c;
            const bool structured = !facA.empty() && facA[n] != nullptr && Dl >= CH_PASS1_MIN_BOND && G.r == Dl;
            if (structured) CH(gram_step_structured(G, n));
            else {
                CH(ensure_absorbed(n));
                Ref xb; double* X = nullptr;
                CH(new_block(Dl * p * Dr, xb, X, "Gram step X"));
                CH(mm(Dl, p * Dr, Dl, G.p, Dl, 1, A[n].p, p * Dr, 1, X, p * Dr, 1));
                M2 Gn;
                CH(new_m2(Dr, Dr, Gn, "Gram matrix"));
                CH(mm(Dr, Dr, Dl * p, A[n].p, 1, Dr, X, Dr, 1, Gn.p, Dr, 1));
                G = Gn;
            }
            gfac[n + 1] = next_nf();
            CH(normalize(G.p, G.r * G.c, gfac[n + 1]));
            if (n + 1 < L && Dr >= CH_PASS1_MIN_BOND) {
                Wt& w = wts[n + 1];
                double* q = nullptr;
                CH(new_block(Dr + 65, w.blk, q, "bond weights"));
                w.d2 = q; w.st65 = q + Dr; w.on = true;
                CH(gram_weights(st, G.p, Dr, CH_PASS1_FLOOR, w.d2, w.st65));
            }
        }
        sm.sublap(0);
        std::vector<int64_t> used;
        for (int64_t n = 0; n < L; ++n) if (wts[n].on) used.push_back(n);
        if (used.empty()) {
            CH(canonise_right(false, 0, 0.0));
            reveal_error_bound = 0.0;
            accepted = true;
            return 0;
        }
        // one read-back: the scale factors of the Gram matrices, the norm of the state, the 64 partial sums of ||K||_F^2 per bond
        const size_t npack = (size_t)L + 1 + 64 * used.size();
        std::vector<double> pack_pageable;
        double* pack = (double*)pinned_host(npack * 8, 4);
        if (!pack) { pack_pageable.resize(npack); pack = pack_pageable.data(); }
        for (int64_t m = 1; m <= L; ++m) CH(d2h(pack + (m - 1), gfac[m], 8));
        CH(d2h(pack + L, G.p, 8));
        for (size_t i = 0; i < used.size(); ++i) CH(d2h(pack + L + 1 + 64 * i, wts[used[i]].st65, 64 * 8));
        CH(sync("weighted pass: statistics"));
        std::vector<double> logg(L + 1, 0.0);
        for (int64_t m = 1; m <= L; ++m) logg[m] = logg[m - 1] + log2(pack[m - 1]);
        const double log_psi = 0.5 * (logg[L] + log2(pack[L]));
        std::vector<double> bound(L + 1, 0.0);
        for (size_t i = 0; i < used.size(); ++i) {
            double s = 0.0;
            for (int q = 0; q < 64; ++q) s += pack[L + 1 + 64 * i + q];
            bound[used[i]] = pow(s, 0.25);
        }
        const double budget = ldexp(1.0, -57) / (double)used.size();
        G.blk.reset();
        CH(ones11(C));
        pC = L;
        double lognf_done = 0.0, err = 0.0;
        std::vector<double*> pending;
        for (int64_t n = L - 1; n >= 0; --n) {
            if (!wts[n].on) {
                CH(ensure_absorbed(n));
                const M2 Cm = C;
                CH(site_right(n, &Cm, 0.0));
                pending.push_back(nfs_dev + 2 * (nfs_count - 1));
                continue;
            }
            const Wt& w = wts[n];
            const int64_t Dl = A[n].a, p = A[n].b, Dr = A[n].c, r = C.c;
            Ref ab; double* Am = nullptr;
            CH(new_block(Dl * p * r, ab, Am, "attached site"));
            if (A[n].p) CH(mm(Dl * p, r, Dr, A[n].p, Dr, 1, C.p, r, 1, Am, r, 1));          // M_n (attach_AC), viewed (Dl, p r)
            else { CH(attach_through_factors(n, C, Am)); attach_fused += 1; }
            Ref wb; double* wv = nullptr;
            CH(new_block(2 * Dl + 2, wb, wv, "row weights"));
            double* rn = wv + Dl;
            double* wsum = wv + 2 * Dl;
            CH(rows_norm2(st, Am, Dl, p * r, rn));
            CH(weighted_sum(st, w.d2, rn, Dl, wv, wsum));
            const size_t nh = 1 + pending.size();
            std::vector<double> host_pageable;
            double* host = (double*)pinned_host(nh * 8 + (size_t)Dl * 8, 4);
            if (!host) { host_pageable.resize(nh + (size_t)Dl); host = host_pageable.data(); }
            CH(d2h(host, wsum, 8));
            for (size_t i = 0; i < pending.size(); ++i) CH(d2h(host + 1 + i, pending[i], 8));
            // the sort order of the weighted row norms (stable, descending), needed on both sides
            Ref pb; double* permd = nullptr;
            CH(new_block(2 * Dl, pb, permd, "row order"));
            int64_t* perm = (int64_t*)permd;
            CH(argsort_desc(st, wv, Dl, perm));
            int64_t* hperm = (int64_t*)(host + nh);
            CH(d2h(hperm, perm, (size_t)Dl * 8));
            CH(sync("weighted pass: site read-back"));
            for (size_t i = 0; i < pending.size(); ++i) lognf_done += log2(host[1 + i]);
            pending.clear();
            const double logN = log_psi - lognf_done;
            const double scale = pow(2.0, 0.5 * logg[n] - logN) * bound[n];
            const double fro = sqrt(host[0]);
            double rel_tol = 0.0;
            if (fro > 0.0) rel_tol = std::min(ldexp(1.0, -40), std::max(1e-30, budget / (scale * fro)));
            std::vector<int64_t> order(hperm, hperm + Dl);
            Ref bb_; double* B = nullptr;
            CH(new_block(Dl * p * r, bb_, B, "sorted scaled site"));
            CH(gather_scale_rows(st, Am, Dl, p * r, perm, w.d2, B, 0));
            ab.reset();
            T3 Bt; Bt.blk = bb_; Bt.p = B; Bt.a = Dl; Bt.b = p; Bt.c = r;
            SiteOut o;
            sm.sublap(1);
            CH(site_qr_step(1, Bt, nullptr, rel_tol, false, true, true, o));
            sm.sublap(2);
            // sort order followed by the panel pivoting: perm <- perm[piv]
            std::vector<int64_t> comp((size_t)Dl);
            for (int64_t j = 0; j < Dl; ++j) comp[j] = order[(size_t)o.piv[j]];
            int64_t* hup = (int64_t*)pinned_host((size_t)Dl * 8, 5);
            std::vector<int64_t>* keep_alive = nullptr;
            if (hup) std::copy(comp.begin(), comp.end(), hup);
            else { keep_alive = new std::vector<int64_t>(comp); hup = keep_alive->data(); }
            CH(h2d(perm, hup, (size_t)Dl * 8));
            if (keep_alive) { CH(sync("weighted pass: order upload")); delete keep_alive; }
            M2 Ct;
            CH(new_m2(Dl, o.k, Ct, "centre (weights removed)"));
            CH(gather_scale_rows(st, o.Rm.p, Dl, o.k, perm, w.d2, Ct.p, 1));
            double* nf = next_nf();
            CH(normalize(Ct.p, Dl * o.k, nf));
            pending.push_back(nf);
            A[n] = o.Q;                                            // (k, p, r)
            C = Ct;
            D[n] = o.k; D[n + 1] = r;
            pC = n;
            err += scale * sqrt(o.dropped2);
            sm.sublap(3);
        }
        reveal_error_bound = err;
        accepted = err <= CH_PASS1_ACCEPT;
        return 0;
    }

    // ---- mixed environments and the variational sweeps ----
    int env_update(int side, const M2& Rm, const T3& Aphi, const T3& Ac, M2& out) {
        const int64_t a = Aphi.a, s = Aphi.b, a2 = Aphi.c, c = Ac.a, c2 = Ac.c;
        if (side == 0 ? (Rm.r != c || Rm.c != a) : (Rm.r != a2 || Rm.c != c2)) { set_error("tn_compress_mps: environment does not fit the sites"); return -1; }
        if (side == 0) CH(new_m2(c2, a2, out, "left environment")); else CH(new_m2(a, c, out, "right environment"));
        const int64_t wsb = env_mix_ws_bytes(side, a, s, a2, c, c2);
        void* w = nullptr;
        CH(scratch(1, wsb, w));
        return env_mix(st, side, Rm.p, Aphi.p, Ac.p, a, s, a2, c, c2, out.p, w, wsb);
    }
    int read_scalar(const double* dev, double& v) {
        double* stage = (double*)pinned_host(8, 6);
        double tmp = 0.0;
        CH(d2h(stage ? stage : &tmp, dev, 8));
        CH(sync("overlap"));
        v = stage ? *stage : tmp;
        return 0;
    }
    int update_RL(const std::vector<T3>& phi, int64_t n) {
        M2 nw;
        CH(env_update(0, R[n], phi[n], A[n], nw));
        if (n == L - 1) return read_scalar(nw.p, overlap);
        R[n + 1] = nw;
        return 0;
    }
    int update_RR(const std::vector<T3>& phi, int64_t n) {
        M2 nw;
        CH(env_update(1, R[n + 1], phi[n], A[n], nw));
        if (n == 0) return read_scalar(nw.p, overlap);
        R[n] = nw;
        return 0;
    }
    int optimise_site(const std::vector<T3>& phi, int64_t n) {
        const M2& RL = R[n];
        const M2& RR = R[n + 1];
        const T3& P = phi[n];
        if (RL.c != P.a || RR.r != P.c) { set_error("tn_compress_mps: environments do not fit the site"); return -1; }
        T3 out;
        CH(new_t3(RL.r, P.b, RR.c, out, "optimised site"));
        const int64_t wsb = rar_ws_bytes(RL.r, P.a, P.b, P.c, RR.c);
        void* w = nullptr;
        CH(scratch(1, wsb, w));
        CH(rar(st, RL.p, P.p, RR.p, RL.r, P.a, P.b, P.c, RR.c, out.p, w, wsb));
        A[n] = out;
        return 0;
    }
    int svdvals_host(const M2& Cm, std::vector<double>& S) {       // ops.svdvals (with its QR-preconditioned retry)
        const int64_t k = Cm.r, n = Cm.c, nv = std::min(k, n);
        S.assign((size_t)nv, 0.0);
        const int64_t wsb = svd_ws_bytes(k, n, 0);
        void* w = nullptr;
        CH(scratch(0, wsb, w));
        int sweeps = 0, info = 0;
        {
            ProfPhase ph(PH_SVDVALS);
            const double dm = (double)(k > n ? k : n), dn = (double)(k > n ? n : k);
            prof_note(PROF_SVDVALS_NOMINAL, 1, 4.0 * dm * dn * dn - 4.0 / 3.0 * dn * dn * dn, 8.0 * (dm * dn + dn));
            CH(svd_vals(st, Cm.p, Cm.c, 1, k, n, S.data(), &sweeps, &info, w, wsb));
        }
        if (info == 0) return 0;
        M2 Q1, R1, Q2, R2;
        const bool tall = k >= n;
        CH(plain_qr(Cm.p, tall ? Cm.c : 1, tall ? 1 : Cm.c, tall ? k : n, tall ? n : k, Q1, R1));
        CH(plain_qr(R1.p, 1, R1.c, R1.c, R1.r, Q2, R2));
        const int64_t wsb2 = svd_ws_bytes(R2.r, R2.c, 0);
        CH(scratch(0, wsb2, w));
        ProfPhase ph(PH_SVDVALS);
        CH(svd_vals(st, R2.p, R2.c, 1, R2.r, R2.c, S.data(), &sweeps, &info, w, wsb2));
        if (info != 0) { set_error("tn_compress_mps: Jacobi sweeps did not converge on a %lld x %lld matrix (values only, after QR preconditioning)", (long long)k, (long long)n); return -4; }
        return 0;
    }
    // psi.S[pC] as update_S sees it before taking new values of length `size` (mps._previous_S)
    int previous_S(int64_t bond, int64_t size, std::vector<double>& old) {
        SState& s = Sst[bond];
        bool have = false;
        if (s.is_lazy) {
            if (std::min(s.lazy.r, s.lazy.c) == size) {
                std::vector<double> v;
                CH(svdvals_host(s.lazy, v));
                s.val = v; s.has = true;
                have = true;
            }
            s.is_lazy = false;
            s.lazy = M2();
            if (!have) s.has = false;
        } else if (s.has && (int64_t)s.val.size() == size) have = true;
        if (have) old = s.val;
        else { old.assign((size_t)size, 0.0); old[0] = 1.0; }
        return 0;
    }
    struct Pending { int64_t bond; bool measure; int row; std::vector<double> S; M2 Cm; };
    int variational_compress(const std::vector<T3>& phi, double tol, int max_sweeps, bool lazy_enabled) {
        for (int64_t n = 0; n < L; ++n) CH(update_RL(phi, n));
        double ov = overlap;
        int sweeps = 0;
        double diff = 1.0;
        while (diff > tol) {
            if (sweeps >= max_sweeps) { overlap = ov; return 0; }
            const bool lazy = lazy_enabled && sweeps + 1 >= max_sweeps;
            std::vector<Pending> items;
            std::vector<M2> small;
            auto add = [&](bool measure) -> int {
                if (lazy) { SState& s = Sst[pC]; s.is_lazy = true; s.lazy = C; s.has = false; s.val.clear(); return 0; }
                Pending it;
                it.bond = pC; it.measure = measure; it.row = -1;
                if (std::max(C.r, C.c) <= 64) { it.row = (int)small.size(); it.Cm = C; small.push_back(C); }
                else CH(svdvals_host(C, it.S));
                items.push_back(std::move(it));
                return 0;
            };
            for (int64_t n = L - 1; n >= 1; --n) {
                CH(optimise_site(phi, n));
                CH(site_right(n, nullptr, 0.0));
                CH(add(false));
                CH(update_RR(phi, n));
            }
            for (int64_t n = 0; n < L; ++n) {
                CH(optimise_site(phi, n));
                CH(site_left(n, nullptr, 0.0));
                CH(add(true));
                CH(update_RL(phi, n));
            }
            // finish(): all small centre matrices in one launch, one read-back, then the bookkeeping of update_S in the reference's order.
            // Centre matrices recorded lazily by an earlier stage that the bookkeeping below is going to ask for (first use of the bond
            // in this sweep, same length) ride along in the same launch instead of one synchronous decomposition each (same kernel
            // body, same values).
            std::vector<double> table;
            std::vector<int> lazy_row((size_t)L + 1, -1);
            {
                std::vector<char> seen((size_t)L + 1, 0);
                for (const Pending& it : items) {
                    if (seen[it.bond]) continue;
                    seen[it.bond] = 1;
                    const SState& ss = Sst[it.bond];
                    const int64_t size = it.row >= 0 ? std::min(it.Cm.r, it.Cm.c) : (int64_t)it.S.size();
                    if (ss.is_lazy && std::min(ss.lazy.r, ss.lazy.c) == size && std::max(ss.lazy.r, ss.lazy.c) <= 64) {
                        lazy_row[it.bond] = (int)small.size();
                        small.push_back(ss.lazy);
                    }
                }
            }
            if (!small.empty()) {
                const size_t ns = small.size();
                std::vector<int64_t> desc(5 * ns);
                for (size_t i = 0; i < ns; ++i) {
                    const M2& m = small[i];
                    if (m.r <= m.c) { desc[5 * i] = (int64_t)(intptr_t)m.p; desc[5 * i + 1] = m.c; desc[5 * i + 2] = 1; desc[5 * i + 3] = m.r; desc[5 * i + 4] = m.c; }
                    else { desc[5 * i] = (int64_t)(intptr_t)m.p; desc[5 * i + 1] = 1; desc[5 * i + 2] = m.c; desc[5 * i + 3] = m.c; desc[5 * i + 4] = m.r; }
                }
                Ref db; double* dd = nullptr;
                CH(new_block(5 * ns + 66 * ns, db, dd, "Schmidt table"));
                int64_t* ddesc = (int64_t*)dd;
                double* out66 = dd + 5 * ns;
                int64_t* hst = (int64_t*)pinned_host(5 * ns * 8, 5);
                if (hst) std::copy(desc.begin(), desc.end(), hst);
                CH(h2d(ddesc, hst ? hst : desc.data(), 5 * ns * 8));
                if (!hst) CH(sync("Schmidt descriptors"));
                {
                    ProfPhase ph(PH_SVDVALS);
                    CH(svd_vals_small_batched(st, ddesc, (int64_t)ns, out66));
                }
                table.resize(66 * ns);
                double* stage = (double*)pinned_host(66 * ns * 8, 6);
                CH(d2h(stage ? stage : table.data(), out66, 66 * ns * 8));
                CH(sync("Schmidt values"));
                if (stage) std::copy(stage, stage + 66 * ns, table.begin());
            }
            for (int64_t bnd = 0; bnd <= L; ++bnd) {
                if (lazy_row[bnd] < 0) continue;
                SState& ss = Sst[bnd];
                const int64_t k = std::min(ss.lazy.r, ss.lazy.c);
                const double* row = &table[66 * (size_t)lazy_row[bnd]];
                bool good = row[65] != 0.0;
                for (int64_t i = 0; i < k && good; ++i) good = std::isfinite(row[i]);
                if (good) { ss.val.assign(row, row + k); ss.has = true; ss.is_lazy = false; ss.lazy = M2(); }      // else: previous_S decomposes it
            }
            diff = 0.0;
            for (Pending& it : items) {
                std::vector<double> S;
                if (it.row >= 0) {
                    const int64_t k = std::min(it.Cm.r, it.Cm.c);
                    const double* row = &table[66 * (size_t)it.row];
                    bool good = row[65] != 0.0;
                    for (int64_t i = 0; i < k && good; ++i) good = std::isfinite(row[i]);
                    if (!good) CH(svdvals_host(it.Cm, S));
                    else S.assign(row, row + k);
                } else S = it.S;
                std::vector<double> old;
                CH(previous_S(it.bond, (int64_t)S.size(), old));
                double s2 = 0.0;
                for (size_t i = 0; i < S.size(); ++i) { const double d = old[i] - S[i]; s2 += d * d; }
                const double dS = sqrt(s2);
                SState& ss = Sst[it.bond];
                ss.val = S; ss.has = true; ss.is_lazy = false; ss.lazy = M2();
                if (it.measure) diff = std::max(diff, dS);
            }
            ov = overlap;
            ++sweeps;
        }
        overlap = ov;
        return 0;
    }
};

}  // namespace tn

using namespace tn;

extern "C" {

// Size of the arena for tn_compress_mps.  Dmax >= 0: what a call needs when the weighted first pass is accepted (always, so far): the
// absorbed sites (consumed site by site as the first pass walks over them, their factors are much smaller) plus the attach result, Q, the
// reflector panels and trailing scratch of the largest site's factorisation, the Gram-recursion temporaries and a fixed reserve for
// the small tensors -- about 1.6x the measured peak at L = 2048 (2.15 GB).  Dmax < 0: the conservative bound that also covers the plain
// first pass on the kept input (full-size factors of every site next to the absorbed ones).  A call that runs out of arena fails with
// -3 before anything is returned; tnac4o_amd.ops then retries once with the conservative size.
int64_t tn_compress_mps_arena_bytes(int64_t L, const int64_t* site_dims_host, const int64_t* mpo_dims_host, int64_t Dmax) {
    if (L < 1 || !site_dims_host) { set_error("tn_compress_mps_arena_bytes: bad arguments"); return -1; }
    int64_t total = 0, biggest = 0, bmax = 1;
    for (int64_t n = 0; n < L; ++n) {
        int64_t Dl = site_dims_host[3 * n], p = site_dims_host[3 * n + 1], Dr = site_dims_host[3 * n + 2];
        if (mpo_dims_host && mpo_dims_host[4 * n] > 0) {
            const int64_t ba = mpo_dims_host[4 * n], po = mpo_dims_host[4 * n + 1], bb = mpo_dims_host[4 * n + 2], pi = mpo_dims_host[4 * n + 3];
            Dl *= ba; Dr *= bb; p = po > pi ? po : pi;
            bmax = std::max(bmax, std::max(ba, bb));
        }
        const int64_t bytes = Dl * p * Dr * 8;
        total += bytes;
        biggest = std::max(biggest, bytes);
    }
    // absorbed input + pass-1 output (shared with its copy) + second-pass output, attach result + Q + Y + Wq + trailing scratch of the
    // largest QR, the Gram-recursion temporaries (up to b x a site), SVD workspaces; plus a fixed reserve for the small tensors
    if (Dmax >= 0) return total + 8 * biggest + (int64_t)256 * 1024 * 1024;
    return 3 * total + (10 + 2 * bmax) * biggest + (int64_t)512 * 1024 * 1024;
}

static int compress_mps_once(int64_t L, const double* const* sites_host, const int64_t* site_dims_host, const double* const* mpo_host,
                    const int64_t* mpo_dims_host, int hconj, int64_t Dmax, double tolS, double tolV, int max_sweeps, int graduate, int flags,
                    double* out, int64_t out_slot, int64_t* out_dims_host, double* overlap_host, double* discarded_host, double* schmidt_host,
                    int64_t schmidt_pitch, int64_t* schmidt_len_host, double* nfs_dev, int64_t nfs_cap, int64_t* nfs_count_host, double* info_host,
                    void* arena, int64_t arena_bytes, void* stream);

// apply_mpo (when mpo_host[n] != NULL) + compress_mps of one boundary MPS.  See include/tnpeps.h.
// The row's factorisations use launches with in-kernel barriers (cholqr.hip, smallqr.hip); instead of asking after each of them, the
// row asks ONCE, at its end (or when a step fails on the NaN a launch that gave up leaves behind), whether any of them gave up
// (fused_timeouts: 16 bytes read back).  If so the whole row is redone from its untouched inputs -- the stream has been taken off
// those launch forms by then, so the second attempt runs the six-launch panel chain and the blocked small factorisations, whose
// results are the same bit for bit; info_host[6] counts the redone attempts.
int tn_compress_mps(int64_t L, const double* const* sites_host, const int64_t* site_dims_host, const double* const* mpo_host,
                    const int64_t* mpo_dims_host, int hconj, int64_t Dmax, double tolS, double tolV, int max_sweeps, int graduate, int flags,
                    double* out, int64_t out_slot, int64_t* out_dims_host, double* overlap_host, double* discarded_host, double* schmidt_host,
                    int64_t schmidt_pitch, int64_t* schmidt_len_host, double* nfs_dev, int64_t nfs_cap, int64_t* nfs_count_host, double* info_host,
                    void* arena, int64_t arena_bytes, void* stream) {
    for (int attempt = 0; attempt < 2; ++attempt) {
        int rc;
        {
            FusedDeferCheck defer;
            rc = compress_mps_once(L, sites_host, site_dims_host, mpo_host, mpo_dims_host, hconj, Dmax, tolS, tolV, max_sweeps, graduate, flags, out, out_slot,
                                   out_dims_host, overlap_host, discarded_host, schmidt_host, schmidt_pitch, schmidt_len_host, nfs_dev, nfs_cap,
                                   nfs_count_host, info_host, arena, arena_bytes, stream);
        }
        if (rc == -1) return rc;                                   // argument error: nothing was launched
        int gave_up = 0;
        if (fused_check_needed()) {
            char keep[512];
            strncpy(keep, get_error(), sizeof(keep) - 1);          // (the check may overwrite the row's own error text)
            keep[sizeof(keep) - 1] = 0;
            const int rc2 = fused_timeouts((hipStream_t)stream, &gave_up);
            if (rc2) return rc2;
            if (gave_up == 0 && rc) set_error("%s", keep);
        }
        if (gave_up == 0) {
            if (rc == 0 && info_host) info_host[6] = (double)attempt;
            return rc;
        }
    }
    set_error("tn_compress_mps: launches with in-kernel barriers gave up twice in a row");
    return -7;
}

static int compress_mps_once(int64_t L, const double* const* sites_host, const int64_t* site_dims_host, const double* const* mpo_host,
                    const int64_t* mpo_dims_host, int hconj, int64_t Dmax, double tolS, double tolV, int max_sweeps, int graduate, int flags,
                    double* out, int64_t out_slot, int64_t* out_dims_host, double* overlap_host, double* discarded_host, double* schmidt_host,
                    int64_t schmidt_pitch, int64_t* schmidt_len_host, double* nfs_dev, int64_t nfs_cap, int64_t* nfs_count_host, double* info_host,
                    void* arena, int64_t arena_bytes, void* stream) {
    TN_CHECK_ARG(L >= 1 && sites_host && site_dims_host && out && out_dims_host && arena && nfs_dev, "null operand");
    TN_CHECK_ARG(Dmax >= 1 && max_sweeps >= 0 && nfs_cap >= 8, "bad parameters");
    TN_CHECK_ARG(tolS > 0.0 && tolV > 0.0, "tolerances must be positive");
    TN_CHECK_ARG(((uintptr_t)arena & 255) == 0, "arena must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    Arena ar((char*)arena, arena_bytes);
    Chain ch(st, ar, L);
    ch.nfs_dev = nfs_dev;
    ch.nfs_cap = nfs_cap;
    ch.hconj = hconj;
    const bool weighted = (flags & 1) != 0, structured = (flags & 2) != 0, lazy = (flags & 4) != 0;
    int rc;
    PassMark pm(st);
    // ---- apply_mpo (mps.py:353-359): K1 into the arena; the factors are kept for the structured Gram recursion
    bool any_mpo = false;
    if (mpo_host && mpo_dims_host) {
        ch.facA.assign((size_t)L, nullptr);
        ch.facW.assign((size_t)L, nullptr);
        ch.facdims.assign((size_t)7 * L, 0);
    }
    // (absorbed bonds, to know up front whether the weighted first pass will run: only then may bulk sites stay un-absorbed)
    int64_t Dbig_abs = 1;
    for (int64_t n = 0; n < L; ++n) {
        const bool has = mpo_host && mpo_dims_host && mpo_host[n];
        Dbig_abs = std::max(Dbig_abs, std::max(site_dims_host[3 * n] * (has ? mpo_dims_host[4 * n] : 1), site_dims_host[3 * n + 2] * (has ? mpo_dims_host[4 * n + 2] : 1)));
    }
    const bool fuse_attach = [] { const char* e = getenv("TN_ATTACH_FUSED"); return !(e && e[0] == '0'); }() && weighted && structured &&
                             Dbig_abs >= 2 * CH_PASS1_MIN_BOND;
    for (int64_t n = 0; n < L; ++n) {
        const int64_t Dl = site_dims_host[3 * n], p = site_dims_host[3 * n + 1], Dr = site_dims_host[3 * n + 2];
        TN_CHECK_ARG(Dl >= 1 && p >= 1 && Dr >= 1 && sites_host[n], "bad site");
        if (mpo_host && mpo_dims_host && mpo_host[n]) {
            const int64_t ba = mpo_dims_host[4 * n], po = mpo_dims_host[4 * n + 1], bb = mpo_dims_host[4 * n + 2], pi = mpo_dims_host[4 * n + 3];
            TN_CHECK_ARG(ba >= 1 && po >= 1 && bb >= 1 && pi >= 1 && p == (hconj ? po : pi), "MPO site does not fit the MPS site");
            const int64_t pnew = hconj ? pi : po;
            T3 t;
            if (fuse_attach && n >= 1 && Dl * ba >= CH_PASS1_MIN_BOND) {           // a weighted site: its absorption rides on the attach (attach_through_factors)
                t.p = nullptr; t.a = Dl * ba; t.b = pnew; t.c = Dr * bb;
            } else {
                if ((rc = ch.new_t3(Dl * ba, pnew, Dr * bb, t, "absorbed site"))) return rc;
                ProfPhase ph(PH_ABSORB);
                if ((rc = absorb(st, sites_host[n], mpo_host[n], t.p, Dl, p, Dr, ba, po, bb, pi, hconj, 1, 0, 0, 0))) return rc;
            }
            ch.A[n] = t;
            if (structured) {
                ch.facA[n] = sites_host[n];
                ch.facW[n] = mpo_host[n];
                int64_t* fd = &ch.facdims[7 * n];
                fd[0] = Dl; fd[1] = p; fd[2] = Dr; fd[3] = ba; fd[4] = po; fd[5] = bb; fd[6] = pi;
            }
            any_mpo = true;
        } else {
            T3 t;
            t.p = const_cast<double*>(sites_host[n]);              // read only: every step that consumes its input works on arena copies
            t.a = Dl; t.b = p; t.c = Dr;
            ch.A[n] = t;
        }
        ch.D[n] = ch.A[n].a;
        ch.D[n + 1] = ch.A[n].c;
    }
    if (!any_mpo || !structured) { ch.facA.clear(); ch.facW.clear(); }
    pm.lap(0);
    // ---- compress_mps (mps.py:175-200)
    const int64_t Dbig = *std::max_element(ch.D.begin(), ch.D.end());
    for (int64_t d : ch.D) ch.bonds_before += d;
    if (weighted && Dbig >= 2 * CH_PASS1_MIN_BOND) {
        const std::vector<T3> keepA = ch.A;
        const std::vector<int64_t> keepD = ch.D;
        const int64_t keep_nfs = ch.nfs_count;
        bool ok = false;
        if ((rc = ch.canonise_right_weighted(ok))) return rc;
        ch.weighted_used = 1;
        if (!ok) {                                                 // bound not met: the plain pass on the kept input
            ch.A = keepA; ch.D = keepD; ch.nfs_count = keep_nfs;
            ch.reveal_fallbacks += 1;
            for (int64_t n = 0; n < L; ++n) if ((rc = ch.ensure_absorbed(n))) return rc;
            if ((rc = ch.canonise_right(false, 0, 0.0))) return rc;
        }
    } else if ((rc = ch.canonise_right(false, 0, 0.0))) return rc;
    ch.facA.clear(); ch.facW.clear();
    for (int64_t d : ch.D) ch.bonds_after += d;
    const std::vector<T3> phi = ch.A;                              // shares the buffers: later passes replace psi's sites, never write them
    std::vector<T3> phi_small;
    const std::vector<T3>* target = &phi;
    std::fill(ch.discarded.begin(), ch.discarded.end(), 0.0);
    for (int64_t i = 0; i <= L; ++i) if ((rc = ch.ones11(ch.R[i]))) return rc;
    pm.lap(1);
    if (graduate) {
        ch.intermediate_pass = true;
        ch.pass_id = 2;
        ch.pass_truncated = false;
        if ((rc = ch.canonise_left(true, Dmax * 4, tolS / 10))) return rc;
        pm.lap(2);
        // The target of the variational sweeps.  When the 4 chi pass has truncated nothing but rounding noise (no bond above 4 chi: every
        // step either kept the centre matrix or dropped at most eps of it, and the rank-revealing factorisations stop at 2^-56), the state
        // it leaves IS phi up to ~L eps -- in left-canonical form and with the bonds that pass found (~100 where the first pass, which
        // cannot see the left part's null space, had to keep ~500).  Overlaps and optimised sites only depend on the STATE the target
        // represents, so that copy serves as the target from here on: the environment and projector products shrink with the square of
        // its bonds.  TN_VAR_TARGET=phi keeps the first pass's tensors (A/B, tests).
        const bool target_phi = [] { const char* e = getenv("TN_VAR_TARGET"); return e && e[0] == 'p'; }();          // (read per call: the tests switch it)
        if (!ch.pass_truncated && !target_phi && tolS / 10 <= CH_EPS) { phi_small = ch.A; target = &phi_small; ch.target_swapped = 1; }
        // ... and then the one variational sweep of this stage has nothing to do: it would optimise every site of the state towards the
        // state itself (overlap 1 - O(L eps)), i.e. return it in another gauge of the same left-canonical form, which the 2 chi pass
        // does not see (it factors from the right end).  What the sweep leaves behind for the final stage are the Schmidt values of
        // every bond (update_S, mps.py:550-560, compared only with later values of the same length): they are the singular values of
        // the centre matrices the 4 chi pass itself met at those bonds, recorded here the way the sweep records them -- unevaluated
        // (_LazyS).  Needs the lazy bookkeeping; TN_VAR1_SKIP=0 runs the sweep.
        const bool var1_run = [] { const char* e = getenv("TN_VAR1_SKIP"); return e && e[0] == '0'; }();
        if (ch.target_swapped && lazy && !var1_run) {
            for (int64_t b = 1; b <= L; ++b) {
                if (!ch.pass_C[b].p) continue;
                Chain::SState& ss = ch.Sst[b];
                ss.is_lazy = true; ss.lazy = ch.pass_C[b]; ss.has = false; ss.val.clear();
            }
            ch.var1_skipped = 1;
        } else if ((rc = ch.variational_compress(*target, tolV, 1, lazy))) return rc;
        ch.pass_C.clear();
        pm.lap(3);
        ch.pass_id = 3;
        if ((rc = ch.canonise_right(true, Dmax * 2, tolS / 2))) return rc;
        ch.intermediate_pass = false;
        pm.lap(4);
    }
    ch.pass_id = 4;
    if ((rc = ch.canonise_left(true, Dmax, tolS))) return rc;
    pm.lap(5);
    if ((rc = ch.variational_compress(*target, tolV, max_sweeps, lazy))) return rc;
    pm.lap(6);
    // ---- results
    for (int64_t n = 0; n < L; ++n) {
        const T3& t = ch.A[n];
        TN_CHECK_ARG(t.numel() <= out_slot, "output slot too small for a compressed site");
        const hipError_t e = hipMemcpyAsync(out + n * out_slot, t.p, (size_t)t.numel() * 8, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return hip_fail(e, "copy result");
        out_dims_host[3 * n] = t.a; out_dims_host[3 * n + 1] = t.b; out_dims_host[3 * n + 2] = t.c;
    }
    pm.lap(7);
    if (overlap_host) *overlap_host = ch.overlap;
    if (discarded_host) for (int64_t i = 0; i <= L; ++i) discarded_host[i] = ch.discarded[i];
    if (schmidt_host && schmidt_len_host) {
        for (int64_t i = 0; i <= L; ++i) {
            Chain::SState& s = ch.Sst[i];
            if (s.is_lazy) {                                       // nobody asked during the call: evaluate now (what reading psi.S does)
                std::vector<double> v;
                if ((rc = ch.svdvals_host(s.lazy, v))) return rc;
                s.val = v; s.has = true; s.is_lazy = false; s.lazy = M2();
            }
            const int64_t len = s.has ? std::min<int64_t>((int64_t)s.val.size(), schmidt_pitch) : 0;
            schmidt_len_host[i] = s.has ? len : -1;
            for (int64_t j = 0; j < len; ++j) schmidt_host[i * schmidt_pitch + j] = s.val[j];
        }
    }
    if (ch.nfs_overflow) {
        set_error("tn_compress_mps: the table of normalisation factors is too small (%lld pairs): size it (2 max_sweeps + 16) L + 64", (long long)nfs_cap);
        return -3;
    }
    if (nfs_count_host) *nfs_count_host = ch.nfs_count;
    if (info_host) {
        info_host[0] = ch.reveal_error_bound;
        info_host[1] = (double)ch.reveal_fallbacks;
        info_host[2] = (double)ch.weighted_used;
        info_host[3] = (double)ar.peak();
        info_host[4] = (double)ch.bonds_before;
        info_host[5] = (double)ch.bonds_after;
        info_host[6] = 0.0;
        info_host[7] = (double)ch.gauge_skipped + 65536.0 * (double)ch.target_swapped + 131072.0 * (double)ch.var1_skipped + 262144.0 * (double)ch.attach_fused;    // (four diagnostics in one word)
    }
    // the results are copied out of the arena by the stream; the caller may reuse the arena for the next call on the SAME stream at once
    return 0;
}

int tn_argsort_desc(const double* w, int64_t n, int64_t* perm_out, void* stream) {
    TN_CHECK_ARG(w && perm_out, "null operand");
    return argsort_desc((hipStream_t)stream, w, n, perm_out);
}
int tn_weighted_sum(const double* a, const double* b, int64_t n, double* w_out, double* sum_out, void* stream) {
    TN_CHECK_ARG(a && b && w_out && sum_out, "null operand");
    return weighted_sum((hipStream_t)stream, a, b, n, w_out, sum_out);
}

}  // extern "C"
