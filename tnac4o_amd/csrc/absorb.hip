// K1 — MPO.MPS absorption (reference mps.py:353-359 apply_mpo -> :753-763 _mps_HA), both orientations.
//
//   hconj = 1:  out[(dl*ba + a), i, (dr*bb + b)] = sum_o A[dl, o, dr] * W[a, o, b, i]     (MPS index major)
//   hconj = 0:  out[(a*Dl + dl), o, (b*Dr + dr)] = sum_i W[a, o, b, i] * A[dl, i, dr]     (MPO index major)
//
// W legs are (left bond ba, out p_o, right bond bb, in p_i).  Arithmetic intensity is ~4 flop/B (the output is
// 134 MB at chi=64 for 0.5 MB of inputs), so this is an HBM-write-bound streaming kernel: one workgroup owns one
// (left MPS index, left MPO index) pair, i.e. `pnew` consecutive output rows = one contiguous slab; the A slab
// A[dl, :, :] and the W slab W[a, :, :, :] are staged in LDS once and every thread produces consecutive elements of
// the fused right bond, so each wave store is a full 512-byte line.
#include "common.h"

namespace tn {

template <bool HCONJ>
__global__ __launch_bounds__(256) void absorb_kernel(const double* __restrict__ A_, const double* __restrict__ W_,
                                                     double* __restrict__ out_, int Dl, int pold, int Dr, int ba, int po,
                                                     int bb, int pi, int64_t bsA, int64_t bsW, int64_t bsO) {
    extern __shared__ double lds[];
    // blockIdx.y = item of a strided batch of equally shaped sites
    const double* __restrict__ A = A_ + (int64_t)blockIdx.y * bsA;
    const double* __restrict__ W = W_ + (int64_t)blockIdx.y * bsW;
    double* __restrict__ out = out_ + (int64_t)blockIdx.y * bsO;
    // contracted / surviving physical legs of W
    const int pc = HCONJ ? po : pi;        // == pold
    const int pnew = HCONJ ? pi : po;
    double* sA = lds;                      // [pc][Dr]
    double* sW = lds + (int64_t)pc * Dr;   // [pc][pnew][bb]
    const int tid = threadIdx.x;
    const int dl = HCONJ ? blockIdx.x / ba : blockIdx.x % Dl;
    const int a = HCONJ ? blockIdx.x % ba : blockIdx.x / Dl;
    for (int e = tid; e < pc * Dr; e += 256) sA[e] = A[(int64_t)dl * pold * Dr + e];
    for (int e = tid; e < pc * pnew * bb; e += 256) {
        const int b = e % bb, q = (e / bb) % pnew, c = e / (bb * pnew);
        // W[a, o, b, i]: o = contracted (HCONJ) or surviving
        const int o = HCONJ ? c : q, i = HCONJ ? q : c;
        sW[e] = W[(((int64_t)a * po + o) * bb + b) * pi + i];
    }
    __syncthreads();
    const int ncol = Dr * bb;                              // fused right bond
    double* orow = out + (int64_t)blockIdx.x * pnew * ncol;
    // A thread owns one element (dr, b) of the fused right bond at a time and walks the pnew output rows: the index
    // arithmetic (two integer divisions) and the pc values of A are per (thread, column), not per output element;
    // for fixed q consecutive threads store consecutive addresses (512 B per wave).
    for (int col = tid; col < ncol; col += 256) {
        const int dr = HCONJ ? col / bb : col % Dr;
        const int b = HCONJ ? col % bb : col / Dr;
        if (pc <= 16) {
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = (c < pc) ? sA[c * Dr + dr] : 0.0;
            for (int q = 0; q < pnew; ++q) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int c = 0; c < 16; c += 2) {
                    if (c < pc) s0 += a[c] * sW[(c * pnew + q) * bb + b];
                    if (c + 1 < pc) s1 += a[c + 1] * sW[((c + 1) * pnew + q) * bb + b];
                }
                orow[(int64_t)q * ncol + col] = s0 + s1;
            }
        } else {
            for (int q = 0; q < pnew; ++q) {
                double s = 0.0;
                for (int c = 0; c < pc; ++c) s += sA[c * Dr + dr] * sW[(c * pnew + q) * bb + b];
                orow[(int64_t)q * ncol + col] = s;
            }
        }
    }
}

// MFMA form for the bulk shapes (pc % 4 == 0, Dr % 16 == 0, bb % 16 == 0): per surviving physical index q the slab
// out_q[slow, fast] is a (slow x pc) . (pc x fast) product on v_mfma_f64_16x16x4_f64, with the contiguous ("fast") half
// of the fused bond on the MFMA column index so that a wave's accumulator registers are stored as full 128-byte rows
// (hconj: fast = MPO bond b, slow = MPS bond dr; otherwise the other way round).  The operand that does not depend on q
// (the A slab) is loaded into registers once; each of the 4 waves walks q = wave, wave + 4, ...
typedef double d4a __attribute__((ext_vector_type(4)));

template <bool HCONJ, int TS, int TF, int KS>   // slow tiles, fast tiles, k-steps (pc / 4)
__global__ __launch_bounds__(256) void absorb_mfma_kernel(const double* __restrict__ A_, const double* __restrict__ W_,
                                                          double* __restrict__ out_, int Dl, int pold, int Dr, int ba, int po,
                                                          int bb, int pi, int64_t bsA, int64_t bsW, int64_t bsO) {
    extern __shared__ double lds[];
    const double* __restrict__ A = A_ + (int64_t)blockIdx.y * bsA;
    const double* __restrict__ W = W_ + (int64_t)blockIdx.y * bsW;
    double* __restrict__ out = out_ + (int64_t)blockIdx.y * bsO;
    const int pc = pold, pnew = HCONJ ? pi : po;
    // the right bond of the MPS site is padded to whole 16-wide tiles in LDS (zeros): edge sites have any Dr (13, 23, 45, 58 ...)
    constexpr int DrP = (HCONJ ? TS : TF) * 16;
    double* sA = lds;                      // [pc][DrP]
    double* sW = lds + (int64_t)pc * DrP;  // [pc][pnew][bb]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dl = HCONJ ? blockIdx.x / ba : blockIdx.x % Dl;
    const int a = HCONJ ? blockIdx.x % ba : blockIdx.x / Dl;
    for (int e = tid; e < pc * DrP; e += 256) {
        const int c = e / DrP, dr = e % DrP;
        sA[e] = dr < Dr ? A[((int64_t)dl * pold + c) * Dr + dr] : 0.0;
    }
    for (int e = tid; e < pc * pnew * bb; e += 256) {
        const int b = e % bb, q = (e / bb) % pnew, c = e / (bb * pnew);
        const int o = HCONJ ? c : q, i = HCONJ ? q : c;
        sW[e] = W[(((int64_t)a * po + o) * bb + b) * pi + i];
    }
    __syncthreads();
    const int li = lane & 15, lk = lane >> 4;
    const int fastdim = HCONJ ? bb : Dr;
    const int64_t ncol = (int64_t)Dr * bb;
    double* oblk = out + (int64_t)blockIdx.x * pnew * ncol;
    // q-independent operand: A slab.  hconj: MFMA "A" operand (rows = dr), TS tiles; else "B" operand (cols = dr), TF tiles
    constexpr int TA = HCONJ ? TS : TF;
    double fa[TA][KS];
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) fa[t][ks] = sA[(ks * 4 + lk) * DrP + t * 16 + li];
    for (int q = wave; q < pnew; q += 4) {
        constexpr int TWn = HCONJ ? TF : TS;
        double fw[TWn][KS];
#pragma unroll
        for (int t = 0; t < TWn; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) fw[t][ks] = sW[((ks * 4 + lk) * pnew + q) * bb + t * 16 + li];
#pragma unroll
        for (int ts = 0; ts < TS; ++ts)
#pragma unroll
            for (int tf = 0; tf < TF; ++tf) {
                d4a acc = d4a{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const double x = HCONJ ? fa[ts][ks] : fw[ts][ks];      // rows  = slow index
                    const double y = HCONJ ? fw[tf][ks] : fa[tf][ks];      // cols  = fast index
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int slow = ts * 16 + lk + 4 * r, fast = tf * 16 + li;
                    if ((HCONJ ? slow : fast) < Dr) oblk[(int64_t)q * ncol + (int64_t)slow * fastdim + fast] = acc[r];
                }
            }
    }
}

template <bool HCONJ>
static bool launch_absorb_mfma(hipStream_t st, dim3 grid, size_t lds, const double* A, const double* W, double* out, int Dl, int pold,
                               int Dr, int ba, int po, int bb, int pi, int64_t bsA, int64_t bsW, int64_t bsO) {
    const int slow = HCONJ ? Dr : bb, fast = HCONJ ? bb : Dr;
    if (pold % 4 || (HCONJ ? bb : bb) % 16) return false;              // the MPO bond must be whole tiles; the MPS bond Dr is padded in LDS
    const int ts = (slow + 15) / 16, tf = (fast + 15) / 16, ks = pold / 4;
    lds = ((size_t)pold * (HCONJ ? ts : tf) * 16 + (size_t)pold * (HCONJ ? pi : po) * bb) * 8;
#define TN_ABS(TS_, TF_, KS_)                                                                                             \
    if (ts == TS_ && tf == TF_ && ks == KS_) {                                                                            \
        if (lds > 64 * 1024)                                                                                              \
            (void)hipFuncSetAttribute((const void*)absorb_mfma_kernel<HCONJ, TS_, TF_, KS_>,                             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
        hipLaunchKernelGGL((absorb_mfma_kernel<HCONJ, TS_, TF_, KS_>), grid, dim3(256), lds, st, A, W, out, Dl, pold, Dr, ba,  \
                           po, bb, pi, bsA, bsW, bsO);                                                                    \
        return true;                                                                                                      \
    }
    // (slow tiles, fast tiles, k steps) of the configurations on the benchmark path: chi in {16,32,64,128}, b = p in {8,16}
    TN_ABS(4, 1, 4) TN_ABS(1, 4, 4) TN_ABS(2, 1, 4) TN_ABS(1, 2, 4) TN_ABS(1, 1, 4) TN_ABS(8, 1, 4) TN_ABS(1, 8, 4)
    TN_ABS(3, 1, 4) TN_ABS(1, 3, 4) TN_ABS(5, 1, 4) TN_ABS(1, 5, 4) TN_ABS(6, 1, 4) TN_ABS(1, 6, 4) TN_ABS(7, 1, 4) TN_ABS(1, 7, 4)
#undef TN_ABS
    return false;
}

int absorb(hipStream_t st, const double* A, const double* W, double* out, int64_t Dl, int64_t pold, int64_t Dr, int64_t ba,
           int64_t po, int64_t bb, int64_t pi, int hconj, int64_t batch, int64_t bsA, int64_t bsW, int64_t bsO) {
    if (batch == 0) return 0;
    TN_CHECK_ARG(batch >= 1 && batch <= 65535, "batch must be in 1..65535");
    TN_CHECK_ARG(Dl >= 1 && pold >= 1 && Dr >= 1 && ba >= 1 && po >= 1 && bb >= 1 && pi >= 1, "non-positive dimension");
    TN_CHECK_ARG(pold == (hconj ? po : pi), "MPS physical leg does not match the contracted MPO leg");
    const int64_t pnew = hconj ? pi : po;
    const int64_t lds = (pold * Dr + pold * pnew * bb) * 8;
    TN_CHECK_ARG(lds <= 160 * 1024, "site too large for the LDS-staged absorb kernel");
    TN_CHECK_ARG(Dl * ba < 2147483647LL, "too many output slabs");
    dim3 grid((unsigned)(Dl * ba), (unsigned)batch);
    prof_begin(st, PROF_ABSORB);
    const bool done = hconj ? launch_absorb_mfma<true>(st, grid, (size_t)lds, A, W, out, (int)Dl, (int)pold, (int)Dr, (int)ba, (int)po,
                                                       (int)bb, (int)pi, bsA, bsW, bsO)
                            : launch_absorb_mfma<false>(st, grid, (size_t)lds, A, W, out, (int)Dl, (int)pold, (int)Dr, (int)ba, (int)po,
                                                        (int)bb, (int)pi, bsA, bsW, bsO);
    if (done) {
        // fall through to the bookkeeping below
    } else if (hconj) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)absorb_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((absorb_kernel<true>), grid, dim3(256), (size_t)lds, st, A, W, out, (int)Dl, (int)pold, (int)Dr,
                           (int)ba, (int)po, (int)bb, (int)pi, bsA, bsW, bsO);
    } else {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)absorb_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((absorb_kernel<false>), grid, dim3(256), (size_t)lds, st, A, W, out, (int)Dl, (int)pold, (int)Dr,
                           (int)ba, (int)po, (int)bb, (int)pi, bsA, bsW, bsO);
    }
    TN_CHECK_LAUNCH("absorb_kernel");
    {   // as a GEMM (Dl Dr) x (ba bb pnew) x pold: 2MNK flops, 8(MK + KN + MN) bytes
        const double Mg = (double)Dl * Dr, Ng = (double)ba * bb * pnew, Kg = (double)pold;
        prof_end(st, PROF_ABSORB, 2.0 * Mg * Ng * Kg * batch, 8.0 * (Mg * Kg + Kg * Ng + Mg * Ng) * batch);
    }
    return 0;
}

}  // namespace tn
