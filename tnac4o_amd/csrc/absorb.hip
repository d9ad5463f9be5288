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
__global__ __launch_bounds__(256) void absorb_kernel(const double* __restrict__ A, const double* __restrict__ W,
                                                     double* __restrict__ out, int Dl, int pold, int Dr, int ba, int po,
                                                     int bb, int pi) {
    extern __shared__ double lds[];
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
    const int64_t ncol = (int64_t)Dr * bb;                 // fused right bond
    double* orow = out + (int64_t)blockIdx.x * pnew * ncol;
    for (int64_t e = tid; e < (int64_t)pnew * ncol; e += 256) {
        const int q = (int)(e / ncol);
        const int col = (int)(e % ncol);
        const int dr = HCONJ ? col / bb : col % Dr;
        const int b = HCONJ ? col % bb : col / Dr;
        double s = 0.0;
        for (int c = 0; c < pc; ++c) s += sA[c * Dr + dr] * sW[(c * pnew + q) * bb + b];
        orow[e] = s;
    }
}

int absorb(hipStream_t st, const double* A, const double* W, double* out, int64_t Dl, int64_t pold, int64_t Dr, int64_t ba,
           int64_t po, int64_t bb, int64_t pi, int hconj) {
    TN_CHECK_ARG(Dl >= 1 && pold >= 1 && Dr >= 1 && ba >= 1 && po >= 1 && bb >= 1 && pi >= 1, "non-positive dimension");
    TN_CHECK_ARG(pold == (hconj ? po : pi), "MPS physical leg does not match the contracted MPO leg");
    const int64_t pnew = hconj ? pi : po;
    const int64_t lds = (pold * Dr + pold * pnew * bb) * 8;
    TN_CHECK_ARG(lds <= 160 * 1024, "site too large for the LDS-staged absorb kernel");
    TN_CHECK_ARG(Dl * ba < 2147483647LL, "too many output slabs");
    dim3 grid((unsigned)(Dl * ba));
    prof_begin(st, PROF_ABSORB);
    if (hconj) {
        if (lds > 64 * 1024) hipFuncSetAttribute((const void*)absorb_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((absorb_kernel<true>), grid, dim3(256), (size_t)lds, st, A, W, out, (int)Dl, (int)pold, (int)Dr,
                           (int)ba, (int)po, (int)bb, (int)pi);
    } else {
        if (lds > 64 * 1024) hipFuncSetAttribute((const void*)absorb_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((absorb_kernel<false>), grid, dim3(256), (size_t)lds, st, A, W, out, (int)Dl, (int)pold, (int)Dr,
                           (int)ba, (int)po, (int)bb, (int)pi);
    }
    TN_CHECK_LAUNCH("absorb_kernel");
    {   // as a GEMM (Dl Dr) x (ba bb pnew) x pold: 2MNK flops, 8(MK + KN + MN) bytes
        const double Mg = (double)Dl * Dr, Ng = (double)ba * bb * pnew, Kg = (double)pold;
        prof_end(st, PROF_ABSORB, 2.0 * Mg * Ng * Kg, 8.0 * (Mg * Kg + Kg * Ng + Mg * Ng));
    }
    return 0;
}

}  // namespace tn
