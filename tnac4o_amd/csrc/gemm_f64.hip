// fp64 GEMM on the gfx950 matrix cores (v_mfma_f64_16x16x4_f64), arbitrary element strides.
//
// K2 of SURVEY.md §8a: every tensordot of the reference's mps.py (attach_CA/AC :740-746, _mps_RL/_RR
// :655-663, _mps_RAR :748-751, projector application :579-580) is a strided view of this kernel, so no
// transposed copies are ever materialised.
//
// Tile: BM x BN per 256-thread workgroup (4 waves as 2 x 2), K step BKT = 16.  Operands are staged through two
// LDS stages, k-major (As[k][m], Bs[k][n]) with pitch = B? + 16 doubles and an XOR swizzle of the low column bits
// by k, which makes both the fragment reads (ds_read_b64, 16 consecutive doubles per k) and the transposed stores
// of k-contiguous operands bank-conflict free.  The next K-tile is prefetched into registers while the current one
// is multiplied and lands in the other stage: one barrier per K step.  (BKT = 32 was measured: a 64 x 64 tile makes
// 0.43 us of MFMA work per 16 of K on one CU and its global loads are in flight meanwhile, so a deeper step saves
// next to nothing on the small products -- 31 -> 29 us at 64 x 64 x 416 -- and costs the large ones occupancy:
// 16384 x 1024 x 1024 drops from 56 to 43 TFLOP/s.)  MFMA f64 lane maps (cdna_hip_programming.md §3): A[l&15][l>>4],
// B[l>>4][l&15], D reg r -> row (l>>4)+4r, col l&15.
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "common.h"

namespace tn {

typedef double d4 __attribute__((ext_vector_type(4)));

struct GemmP {
    const double* A;
    const double* B;
    double* C;
    int64_t M, N, K;
    int64_t rsa, csa, rsb, csb, rsc, csc;
    int64_t bsa, bsb, bsc;
    double alpha, beta;
    int splitk;
    int64_t kchunk;     // K range per split (multiple of 16)
    double* ws;         // split-K partials [(batch*splitk+s)][M][N]
    int tiles_m, tiles_n;
    // optional block-pair indirection (Jacobi SVD): a "vector index" v of batch z lives in physical row
    //   pairs[2z + (v >= pw)] * pw + (v mod pw).  mapA: on A's row index; mapB: 1 on B's k index, 2 on B's column index;
    //   mapC: on C's row index.  skip[z] == 0 -> the whole batch item is a no-op.
    const int* pairs;
    const int* skip;
    int pw, mapA, mapB, mapC;
};

constexpr int BK = 16;          // granularity of the split-K chunks (multiples of 32 for the BKT = 32 kernels)

template <int BM, int BN, int BKT, bool AKFAST, bool BKFAST, bool MAPPED>
__global__ __launch_bounds__(256, (BM == 128 && BN == 128 && BKT == 16) ? 2 : 1) void gemm_kernel(GemmP g) {
    constexpr int BK = BKT;
    constexpr int PA = BM + 16, PB = BN + 16;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
    constexpr int EA = BM * BK / 256, EB = BN * BK / 256;     // elements per thread per tile
    __shared__ double As[2][BK * PA];
    __shared__ double Bs[2][BK * PB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: consecutive tiles of one output row-panel land on the same XCD (L2 reuse of A)
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = bid % 8, loc = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tm0 = (bid / g.tiles_n) * BM, tn0 = (bid % g.tiles_n) * BN;
    const int nsk = g.splitk < 0 ? 1 : g.splitk;
    const int zb = blockIdx.z / nsk, zs = blockIdx.z % nsk;
    if (g.skip && g.skip[zb] == 0) return;
    int blk0 = 0, blk1 = 0;
    if (MAPPED && g.pairs) { blk0 = g.pairs[2 * zb]; blk1 = g.pairs[2 * zb + 1]; }
    auto remap = [&](int64_t v) -> int64_t { return v < g.pw ? (int64_t)blk0 * g.pw + v : (int64_t)blk1 * g.pw + (v - g.pw); };
    const double* A = g.A + zb * g.bsa;
    const double* B = g.B + zb * g.bsb;
    const int64_t k_lo = zs * g.kchunk;
    const int64_t k_hi = (k_lo + g.kchunk < g.K) ? k_lo + g.kchunk : g.K;

    // Element e of this thread's share of a tile sits at (m0 + e DM, kA0 + e DKA) of the A tile and (n0 + e DN,
    // kB0 + e DKB) of the B tile (the fast index of the thread id follows the operand's unit stride, so the loads
    // coalesce).  Without block-pair indirection the global element offsets are affine in e and in the K position:
    // one 64-bit offset per operand is kept and advanced, instead of re-deriving 16 addresses per K step.
    constexpr int DM = AKFAST ? 256 / BK : 0, DKA = AKFAST ? 0 : 256 / BM;
    constexpr int DN = BKFAST ? 256 / BK : 0, DKB = BKFAST ? 0 : 256 / BN;
    const int m0 = AKFAST ? tid / BK : tid % BM, kA0 = AKFAST ? tid % BK : tid / BM;
    const int n0 = BKFAST ? tid / BK : tid % BN, kB0 = BKFAST ? tid % BK : tid / BN;
    const int64_t offA0 = (int64_t)(tm0 + m0) * g.rsa + (int64_t)kA0 * g.csa, stepA = (int64_t)DM * g.rsa + (int64_t)DKA * g.csa;
    const int64_t offB0 = (int64_t)(tn0 + n0) * g.csb + (int64_t)kB0 * g.rsb, stepB = (int64_t)DN * g.csb + (int64_t)DKB * g.rsb;

    double ra[EA], rb[EB];
    auto load_tiles = [&](int64_t k0) {
        if constexpr (!MAPPED) {
            const double* Ak = A + k0 * g.csa + offA0;
            const double* Bk = B + k0 * g.rsb + offB0;
#pragma unroll
            for (int e = 0; e < EA; ++e)
                ra[e] = (tm0 + m0 + e * DM < g.M && k0 + kA0 + e * DKA < k_hi) ? Ak[e * stepA] : 0.0;
#pragma unroll
            for (int e = 0; e < EB; ++e)
                rb[e] = (tn0 + n0 + e * DN < g.N && k0 + kB0 + e * DKB < k_hi) ? Bk[e * stepB] : 0.0;
        } else {
#pragma unroll
            for (int e = 0; e < EA; ++e) {
                const int64_t gm = tm0 + m0 + e * DM, gk = k0 + kA0 + e * DKA;
                ra[e] = (gm < g.M && gk < k_hi) ? A[(g.mapA ? remap(gm) : gm) * g.rsa + gk * g.csa] : 0.0;
            }
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                const int64_t gn = tn0 + n0 + e * DN, gk = k0 + kB0 + e * DKB;
                rb[e] = (gn < g.N && gk < k_hi) ? B[(g.mapB == 1 ? remap(gk) : gk) * g.rsb + (g.mapB == 2 ? remap(gn) : gn) * g.csb] : 0.0;
            }
        }
    };
    auto store_tiles = [&](int stage) {
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            const int m = m0 + e * DM, k = kA0 + e * DKA;
            As[stage][k * PA + (m ^ k)] = ra[e];
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const int n = n0 + e * DN, k = kB0 + e * DKB;
            Bs[stage][k * PB + (n ^ k)] = rb[e];
        }
    };

    d4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

    const int lr = lane & 15, lk = lane >> 4;
    int cur = 0;
    if (k_lo < k_hi) {
        load_tiles(k_lo);
        store_tiles(0);
    }
    __syncthreads();
    for (int64_t k0 = k_lo; k0 < k_hi; k0 += BK) {
        const bool more = k0 + BK < k_hi;
        if (more) load_tiles(k0 + BK);                    // in flight while this stage is multiplied
        const double* Ac = As[cur];
        const double* Bc = Bs[cur];
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            const int k = kk * 4 + lk;
            double a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = Ac[k * PA + ((wm * WM + i * 16 + lr) ^ k)];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bc[k * PB + ((wn * WN + j * 16 + lr) ^ k)];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) store_tiles(cur ^ 1);                    // the other stage: nobody reads it before the barrier
        __syncthreads();
        cur ^= 1;
    }

    if (g.splitk > 1 || g.splitk < 0) {
        double* W = g.ws + (int64_t)blockIdx.z * g.M * g.N;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = tm0 + wm * WM + i * 16 + lk + 4 * r, col = tn0 + wn * WN + j * 16 + lr;
                    if (row < g.M && col < g.N) W[row * g.N + col] = acc[i][j][r];
                }
    } else {
        double* C = g.C + zb * g.bsc;
        const bool hb = g.beta != 0.0;
        // beta != 0: the old values of a 16-row slab are fetched TOGETHER before any of them is used (a load whose value is consumed
        // right away costs a memory round trip per element: 64 in a row for a 128 x 128 tile, which made the rank-32 updates
        // A -= W X latency-bound in their epilogue)
        if (hb) {
            double cv[2][TN][4];
            auto fetch = [&](int i, int buf) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t row = tm0 + wm * WM + i * 16 + lk + 4 * r, col = tn0 + wn * WN + j * 16 + lr;
                        cv[buf][j][r] = (row < g.M && col < g.N) ? C[((MAPPED && g.mapC) ? remap(row) : row) * g.rsc + col * g.csc] : 0.0;
                    }
            };
            fetch(0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (i + 1 < TM) fetch(i + 1, (i + 1) & 1);       // the next slab's values are in flight while this one is stored
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t row = tm0 + wm * WM + i * 16 + lk + 4 * r, col = tn0 + wn * WN + j * 16 + lr;
                        if (row < g.M && col < g.N) {
                            double v = g.alpha * acc[i][j][r];
                            v += g.beta * cv[i & 1][j][r];
                            C[((MAPPED && g.mapC) ? remap(row) : row) * g.rsc + col * g.csc] = v;
                        }
                    }
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t row = tm0 + wm * WM + i * 16 + lk + 4 * r, col = tn0 + wn * WN + j * 16 + lr;
                        if (row < g.M && col < g.N) C[((MAPPED && g.mapC) ? remap(row) : row) * g.rsc + col * g.csc] = g.alpha * acc[i][j][r];
                    }
        }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmP g) {
    const int64_t mn = g.M * g.N;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int zb = blockIdx.y;
    if (e >= mn) return;
    if (g.skip && g.skip[zb] == 0) return;
    const double* W = g.ws + (int64_t)zb * g.splitk * mn + e;
    double s = 0.0;
    int k = 0;
    for (; k + 8 <= g.splitk; k += 8) {              // eight partials in flight per round trip, summed in a fixed order
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = W[(int64_t)(k + u) * mn];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < g.splitk; ++k) s += W[(int64_t)k * mn];
    const int64_t row = e / g.N, col = e % g.N;
    double* c = g.C + zb * g.bsc + row * g.rsc + col * g.csc;
    double v = g.alpha * s;
    if (g.beta != 0.0) v += g.beta * *c;
    *c = v;
}

template <int BM, int BN, int BKT>
static void launch_tile(hipStream_t st, const GemmP& g, dim3 grid, bool ak, bool bk) {
    if constexpr (BM == 64 && BN == 64) {          // the block-pair indirection of the Jacobi SVD only ever uses this tile
        if (g.pairs) {
            if (ak && bk) hipLaunchKernelGGL((gemm_kernel<BM, BN, BKT, true, true, true>), grid, dim3(256), 0, st, g);
            else if (ak) hipLaunchKernelGGL((gemm_kernel<BM, BN, BKT, true, false, true>), grid, dim3(256), 0, st, g);
            else if (bk) hipLaunchKernelGGL((gemm_kernel<BM, BN, BKT, false, true, true>), grid, dim3(256), 0, st, g);
            else hipLaunchKernelGGL((gemm_kernel<BM, BN, BKT, false, false, true>), grid, dim3(256), 0, st, g);
            return;
        }
    }
    if (ak && bk) hipLaunchKernelGGL((gemm_kernel<BM, BN, BKT, true, true, false>), grid, dim3(256), 0, st, g);
    else if (ak) hipLaunchKernelGGL((gemm_kernel<BM, BN, BKT, true, false, false>), grid, dim3(256), 0, st, g);
    else if (bk) hipLaunchKernelGGL((gemm_kernel<BM, BN, BKT, false, true, false>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_kernel<BM, BN, BKT, false, false, false>), grid, dim3(256), 0, st, g);
}

// ---- launch plan --------------------------------------------------------------------------------------------------
// Almost every product of a sweep is small: its tiles fill a few of the 256 CUs, and one CU makes 64 x 64 x 16 of fp64 MFMA work
// in ~0.9 us (a 128 x 128 tile: ~1.9 us), so the time of such a product is (K steps per workgroup) x that, plus the launches.
// Products of less than TN_GEMM_SMALLWORK (2^30) multiply-adds are therefore spread over about one workgroup per CU: 64 x 64
// tiles instead of 128 x 128 while those would leave half the chip idle, and K split down to chunks of TN_GEMM_MINCHUNK (32, at most TN_GEMM_SMAX = 64 of them),
// the partial sums added up by splitk_reduce_kernel (an in-kernel reduction by the last workgroup of a tile to arrive was
// measured slower: it pulls all the partials of a tile through one CU, 22 us against 12 us for the two launches at 64 x 64 x 1024).
// Larger products are bound by MFMA throughput: big tiles, K split only to fill the chip, chunks of at least 128.
struct GemmPlan { int bm, bn, s; };
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static GemmPlan plan_gemm(int64_t M, int64_t N, int64_t K, int64_t batch) {
    static const int64_t small_work = (int64_t)1 << env_int("TN_GEMM_SMALLWORK_LOG2", 30);
    // small products: 128 x 128 tiles from this many workgroups on.  Round 5: never (was 128) -- the small products with hundreds of big
    // tiles are the rank-32 updates of the panel loops, HBM round trips of the trailing matrix with two K steps of arithmetic: four times
    // the workgroups hide their prologues behind each other's epilogues (first pass -2.4 %, four chains 425.8 -> 418.1 ms/sweep)
    static const int big_tiles = env_int("TN_GEMM_BIGTILES", 1 << 30);
    static const int target_wg = env_int("TN_GEMM_TARGETWG", 256);
    static const int min_chunk = env_int("TN_GEMM_MINCHUNK", 32);
    static const int smax = env_int("TN_GEMM_SMAX", 64);
    static const int bigsplit_wgs = env_int("TN_GEMM_BIGSPLIT_WGS", 192), bigsplit_target = env_int("TN_GEMM_BIGSPLIT_TARGET", 512);
    const bool small = (double)M * (double)N * (double)K * (double)batch < (double)small_work;
    GemmPlan p;
    if (M > 64 && N > 64) {
        const bool big = !small || cdiv(M, 128) * cdiv(N, 128) * batch >= big_tiles;
        p.bm = p.bn = big ? 128 : 64;
    } else if (N <= 32 && M > 64) { p.bm = 128; p.bn = 32; }
    else if (M <= 32 && N > 64) { p.bm = 32; p.bn = 128; }
    else { p.bm = 64; p.bn = 64; }
    const int64_t wgs = cdiv(M, p.bm) * cdiv(N, p.bn) * batch;
    p.s = 1;
    if (small) {
        if (wgs < target_wg * 3 / 4 && K >= 2 * min_chunk) {
            // (up to 16 tiles: the split depends on K alone, so that a product and the same product cut into column ranges -- the
            //  look-ahead of tn_qr -- add up in the same order)
            int64_t s = wgs <= 16 ? smax : cdiv(target_wg, wgs);
            if (s > K / min_chunk) s = K / min_chunk;
            if (s > smax) s = smax;
            p.s = s < 2 ? 1 : (int)s;
        }
    } else if (wgs < bigsplit_wgs && K >= 512) {
        int64_t s = cdiv(bigsplit_target, wgs);
        if (s > K / 128) s = K / 128;
        if (s > 64) s = 64;
        p.s = s < 2 ? 1 : (int)s;
    }
    return p;
}

int gemm_forced_split(int64_t K, int s, int64_t* kchunk) {
    *kchunk = s > 1 ? align_up(cdiv(K, s), 2 * BK) : (K > 0 ? align_up(K, 2 * BK) : 2 * BK);      // whole steps of either K depth
    return s > 1 ? (int)cdiv(K, *kchunk) : s;
}

int64_t gemm_ws_bytes(int64_t M, int64_t N, int64_t K, int64_t batch) {
    const GemmPlan p = plan_gemm(M, N, K, batch);
    return p.s > 1 ? (int64_t)p.s * batch * M * N * 8 : 0;
}

int gemm(hipStream_t st, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t rsa, int64_t csa,
         const double* B, int64_t rsb, int64_t csb, double beta, double* C, int64_t rsc, int64_t csc, int64_t batch,
         int64_t bsa, int64_t bsb, int64_t bsc, double* ws, int64_t ws_bytes) {
    return gemm_ex(st, M, N, K, alpha, A, rsa, csa, B, rsb, csb, beta, C, rsc, csc, batch, bsa, bsb, bsc, ws, ws_bytes, nullptr);
}

// TN_GEMM_TRACE=1 (diagnostics): every call is timed synchronously with a pair of events and booked under its shape; the table of
// the most expensive shapes is printed when the process exits.  Perturbs the run (one synchronisation per GEMM).
namespace {
struct ShapeStat { double ms = 0.0; long calls = 0; };
struct GemmTrace {
    bool on;
    std::mutex mu;
    std::map<std::tuple<int64_t, int64_t, int64_t, int64_t, int, int, int>, ShapeStat> tab;
    GemmTrace() { const char* e = getenv("TN_GEMM_TRACE"); on = e && e[0] == '1'; }
    ~GemmTrace() {
        if (!on || tab.empty()) return;
        std::vector<std::pair<double, std::tuple<int64_t, int64_t, int64_t, int64_t, int, int, int>>> v;
        double tot = 0.0;
        for (auto& kv : tab) { v.push_back({kv.second.ms, kv.first}); tot += kv.second.ms; }
        std::sort(v.begin(), v.end(), [](auto& a, auto& b) { return a.first > b.first; });
        {   // classes: split-K or not x depth K x tiles of the launch plan
            std::map<std::tuple<int, int, int>, ShapeStat> cls;
            for (auto& kv : tab) {
                const int64_t M = std::get<0>(kv.first), N = std::get<1>(kv.first), K = std::get<2>(kv.first), b = std::get<3>(kv.first);
                const int sk = std::get<6>(kv.first);
                const int kb = K <= 32 ? 32 : K <= 64 ? 64 : K <= 128 ? 128 : K <= 256 ? 256 : K <= 512 ? 512 : K <= 1024 ? 1024 : K <= 4096 ? 4096 : 65536;
                const int64_t t = cdiv(M, 64) * cdiv(N, 64) * b;
                const int tb = t <= 1 ? 1 : t <= 4 ? 4 : t <= 16 ? 16 : t <= 64 ? 64 : t <= 256 ? 256 : 100000;
                ShapeStat& c = cls[std::make_tuple(sk > 1 ? 1 : (sk < 0 ? -1 : 0), kb, tb)];
                c.ms += kv.second.ms; c.calls += kv.second.calls;
            }
            fprintf(stderr, "[tn_gemm classes] split(1)/plain(0)/jacobi(-1)  K<=  64x64-tiles<= : calls  ms  us/call\n");
            for (auto& kv : cls)
                fprintf(stderr, "  %2d %6d %7d : %7ld %9.2f %8.2f\n", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), kv.second.calls, kv.second.ms,
                        1e3 * kv.second.ms / kv.second.calls);
        }
        fprintf(stderr, "[tn_gemm trace] %zu shapes, %.1f ms in total; M N K batch transA transB splitk : calls, ms, us/call, TFLOP/s\n", v.size(), tot);
        for (size_t i = 0; i < v.size() && i < 40; ++i) {
            auto& k = v[i].second;
            const ShapeStat& st = tab[k];
            const double fl = 2.0 * std::get<0>(k) * std::get<1>(k) * std::get<2>(k) * std::get<3>(k) * st.calls;
            fprintf(stderr, "  %6lld %6lld %6lld %5lld  %d %d %3d : %6ld  %8.2f  %8.2f  %6.2f\n", (long long)std::get<0>(k), (long long)std::get<1>(k),
                    (long long)std::get<2>(k), (long long)std::get<3>(k), std::get<4>(k), std::get<5>(k), std::get<6>(k), st.calls, st.ms,
                    1e3 * st.ms / st.calls, fl / (st.ms * 1e-3) / 1e12);
        }
    }
};
GemmTrace g_trace;
}  // namespace

static int gemm_ex_impl(hipStream_t st, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t rsa, int64_t csa,
                        const double* B, int64_t rsb, int64_t csb, double beta, double* C, int64_t rsc, int64_t csc, int64_t batch,
                        int64_t bsa, int64_t bsb, int64_t bsc, double* ws, int64_t ws_bytes, const GemmExtra* x);

int gemm_ex(hipStream_t st, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t rsa, int64_t csa,
            const double* B, int64_t rsb, int64_t csb, double beta, double* C, int64_t rsc, int64_t csc, int64_t batch,
            int64_t bsa, int64_t bsb, int64_t bsc, double* ws, int64_t ws_bytes, const GemmExtra* x) {
    if (!g_trace.on) return gemm_ex_impl(st, M, N, K, alpha, A, rsa, csa, B, rsb, csb, beta, C, rsc, csc, batch, bsa, bsb, bsc, ws, ws_bytes, x);
    thread_local hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!e0) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); }
    (void)hipEventRecord(e0, st);
    const int rc = gemm_ex_impl(st, M, N, K, alpha, A, rsa, csa, B, rsb, csb, beta, C, rsc, csc, batch, bsa, bsb, bsc, ws, ws_bytes, x);
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const int sk = (x && x->force_splitk > 0) ? x->force_splitk : plan_gemm(M, N, K, batch).s;
    std::lock_guard<std::mutex> lk(g_trace.mu);
    ShapeStat& ss = g_trace.tab[std::make_tuple(M, N, K, batch, (csa == 1 && rsa != 1) ? 1 : 0, (rsb == 1 && csb != 1) ? 1 : 0, (x && x->pairs) ? -sk : sk)];
    ss.ms += ms;
    ss.calls += 1;
    return rc;
}

static int gemm_ex_impl(hipStream_t st, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t rsa, int64_t csa,
                        const double* B, int64_t rsb, int64_t csb, double beta, double* C, int64_t rsc, int64_t csc, int64_t batch,
                        int64_t bsa, int64_t bsb, int64_t bsc, double* ws, int64_t ws_bytes, const GemmExtra* x) {
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    TN_CHECK_ARG(K >= 0, "negative K");
    // Column-major C (unit ROW stride): the MFMA result layout puts 16 consecutive COLUMNS of a row in the lanes of a store, i.e.
    // 16 separate 32-byte pieces per instruction there.  The transposed product C^T = B^T A^T is the same arithmetic in the same
    // order (element by element: the same K sequence, the same splits) with C^T row-major -- full 128-byte segments.  Not for the
    // block-pair indirection / raw partials of the Jacobi SVD, whose index maps and partial layout are tied to the operand roles.
    static const bool swap_on = [] { const char* e = getenv("TN_GEMM_SWAP"); return !(e && e[0] == '0'); }();
    if (swap_on && rsc == 1 && csc != 1 && N > 1 && !(x && (x->pairs || x->raw_partials)))          // (skip flags are per batch item: they follow)
        return gemm_ex_impl(st, N, M, K, alpha, B, csb, rsb, A, csa, rsa, beta, C, csc, rsc, batch, bsb, bsa, bsc, ws, ws_bytes, x);
    GemmP g;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K;
    g.rsa = rsa; g.csa = csa; g.rsb = rsb; g.csb = csb; g.rsc = rsc; g.csc = csc;
    g.bsa = bsa; g.bsb = bsb; g.bsc = bsc; g.alpha = alpha; g.beta = beta; g.ws = ws;
    g.pairs = x ? x->pairs : nullptr; g.skip = x ? x->skip : nullptr;
    g.pw = x ? x->pw : 0; g.mapA = x ? x->mapA : 0; g.mapB = x ? x->mapB : 0; g.mapC = x ? x->mapC : 0;
    const GemmPlan pl = plan_gemm(M, N, K, batch);
    const int bm = pl.bm, bn = pl.bn;
    TN_CHECK_ARG(g.pairs == nullptr || (bm == 64 && bn == 64), "block-pair indirection is built for the 64 x 64 tile only");
    g.tiles_m = (int)cdiv(M, bm); g.tiles_n = (int)cdiv(N, bn);
    int s = (x && x->force_splitk > 0) ? x->force_splitk : pl.s;
    if (s > 1) {                                   // a smaller scratch than gemm_ws_bytes asks for: as many splits as fit
        const int64_t fit = ws ? ws_bytes / (batch * M * N * 8) : 0;
        if (fit < s) s = fit < 2 ? 1 : (int)fit;
    }
    const bool raw = x && x->raw_partials;
    TN_CHECK_ARG(!raw || ws != nullptr, "raw partials need a workspace");
    g.splitk = s = gemm_forced_split(K, s, &g.kchunk);
    if (raw && s == 1) g.splitk = -1;          // single "partial": still written to ws (handled below)
    if (x && x->splitk_used) *x->splitk_used = s;
    if (batch * s > 65535) {      // grid.z limit: run the batch in slices (same stream, so the split-K scratch can be reused)
        TN_CHECK_ARG(x == nullptr && s <= 65535, "batch*splitk exceeds grid.z");
        const int64_t cb = 65535 / s;
        for (int64_t b0 = 0; b0 < batch; b0 += cb) {
            const int64_t nbt = batch - b0 < cb ? batch - b0 : cb;
            const int rc = gemm_ex_impl(st, M, N, K, alpha, A + b0 * bsa, rsa, csa, B + b0 * bsb, rsb, csb, beta, C + b0 * bsc, rsc, csc,
                                        nbt, bsa, bsb, bsc, ws, ws_bytes, nullptr);
            if (rc) return rc;
        }
        return 0;
    }
    const bool ak = (csa == 1 && rsa != 1), bk = (rsb == 1 && csb != 1);
    dim3 grid(g.tiles_m * g.tiles_n, 1, (unsigned)(batch * s));
    const int fam = (bm == 128 && bn == 128) ? PROF_GEMM_128x128 : (bm == 128) ? PROF_GEMM_128x32
                    : (bm == 32) ? PROF_GEMM_32x128 : PROF_GEMM_64x64;
    prof_begin(st, fam);
    if (bm == 128 && bn == 128) launch_tile<128, 128, 16>(st, g, grid, ak, bk);
    else if (bm == 128 && bn == 32) launch_tile<128, 32, 16>(st, g, grid, ak, bk);
    else if (bm == 32 && bn == 128) launch_tile<32, 128, 16>(st, g, grid, ak, bk);
    else launch_tile<64, 64, 16>(st, g, grid, ak, bk);
    TN_CHECK_LAUNCH("gemm_kernel");
    // algorithmic work of SURVEY.md §8d: 2MNK flops, 8(MK + KN + MN) bytes
    prof_end(st, fam, 2.0 * M * N * K * batch, 8.0 * batch * ((double)M * K + (double)K * N + (double)M * N));
    if (s > 1 && !raw) {
        dim3 rg((unsigned)cdiv(M * N, 256), (unsigned)batch);
        prof_begin(st, PROF_SPLITK_REDUCE);
        hipLaunchKernelGGL(splitk_reduce_kernel, rg, dim3(256), 0, st, g);
        TN_CHECK_LAUNCH("splitk_reduce_kernel");
        prof_end(st, PROF_SPLITK_REDUCE, (double)s * M * N * batch, 8.0 * batch * ((double)s + 1.0) * M * N);
    }
    return 0;
}

}  // namespace tn
