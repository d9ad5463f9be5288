// K8 driver -- the beam search over the rows and sites of the lattice (reference tnac4o.py:429-542, search_ground_state's loop)
// walked in C++ on one stream: what tnac4o_amd/beam.py does with ~90 torch calls per site-step under the GIL (four rotations
// search at once and serialise there) is here a fixed sequence of launches and four scalar read-backs per site-step.
//
// Same canonical order as beam.py (so the results agree bit for bit, tests/test_gpu_configs.py):
//   * candidates of a site-step in ascending flat index f = branch * q + state, kept when log2 p > max + log2(cutoff);
//   * merge groups (equal boundary rows) in lexicographic order of the row, compared through ONE int64 key built from
//     order-preserving ranks (prefix rank, down index, right index, suffix rank); members in candidate order; representative =
//     first member of minimal energy (tn_merge_groups);
//   * the M survivors = the M largest group log2 p (ties to the smaller group index), kept in group order.
// Sorting is rocPRIM's radix sort (stable; called directly, no CUDA-compat layer); "unique" = sort, head flags, prefix sum.  Index rows live column-major (a column of
// boundary indices is contiguous: it is what tn_calc_pn and the keys read).
#include <string.h>

#include <cstring>
#include <limits>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <vector>

#include "../../include/tnpeps.h"
#include "common.h"

namespace tn {

int gemm(hipStream_t st, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t rsa, int64_t csa, const double* B, int64_t rsb,
         int64_t csb, double beta, double* C, int64_t rsc, int64_t csc, int64_t batch, int64_t bsa, int64_t bsb, int64_t bsc, double* ws, int64_t ws_bytes);
int64_t gemm_ws_bytes(int64_t M, int64_t N, int64_t K, int64_t batch);
int calc_pn(hipStream_t st, const double* T1, const double* RR, const double* F, const int32_t* dmap, const int32_t* rmap, const int32_t* pref,
            const int32_t* suf, const int32_t* lidx, const int32_t* uidx, int64_t nb, int64_t q, int64_t nl, int64_t nu, int64_t p, int64_t Dr, int64_t br,
            double* P, double* minP, const double* parent_log2p, double* log2p_out);
int merge_groups(hipStream_t st, const double* E, const double* lp, const int64_t* deg, const int64_t* pos, const int64_t* starts, int64_t ng,
                 double min_dEng, int64_t* rep_pos, int64_t* degn, double* lpn);
int env_rr_batched(hipStream_t st, const double* A, const double* RRprev, const double* W, const int32_t* parent, const int32_t* uidx, int64_t nk,
                   int64_t Dl, int64_t p, int64_t Dr, int64_t bl, int64_t br, int64_t pu, double* out);
int env_rl_batched(hipStream_t st, const double* T1, const int32_t* par, const int32_t* didx, int64_t nk, int64_t p, int64_t Dr, double* out);
int mpo_from_factor(hipStream_t st, const double* F, const int32_t* dmap, const int32_t* rmap, int64_t q, int64_t nl, int64_t nu, int64_t pd, int64_t br,
                    double* W);

namespace {

#define BS(call)                   \
    do {                           \
        const int rc__ = (call);   \
        if (rc__) return rc__;     \
    } while (0)
#define BSH(call, what)                                   \
    do {                                                  \
        const hipError_t e__ = (call);                    \
        if (e__ != hipSuccess) return hip_fail(e__, what); \
    } while (0)

constexpr double NEG_INF = -__builtin_huge_val();

// ---- small kernels -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void iota_kernel(int32_t* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (int32_t)i;
}
__global__ __launch_bounds__(256) void fill_i32_kernel(int32_t* out, int64_t n, int32_t v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = v;
}
// key of the suffix vind[:, c:] = (vind[:, c], rank of vind[:, c+1:])
__global__ __launch_bounds__(256) void suffix_key_kernel(const int32_t* col, const int32_t* suf_prev, int64_t nkeys_prev, int64_t n, int64_t* key) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) key[i] = (int64_t)col[i] * nkeys_prev + suf_prev[i];
}
__global__ __launch_bounds__(256) void heads_kernel(const int64_t* skey, int64_t n, int32_t* head) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) head[i] = (i == 0 || skey[i] != skey[i - 1]) ? 1 : 0;
}
// after the prefix sum of the head flags: inverse (group of every element), first member of every group (stable sort: the head
// of a group is its smallest original index), offsets of the groups in the sorted order (starts[ng] = n)
__global__ __launch_bounds__(256) void unique_scatter_kernel(const int32_t* sidx, const int32_t* head, const int32_t* gid, int64_t n, int32_t* inv,
                                                            int32_t* first, int64_t* starts) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t g = gid[i] - 1;
    if (inv) inv[sidx[i]] = g;
    if (head[i]) {
        if (first) first[g] = sidx[i];
        if (starts) starts[g] = i;
    }
    if (i == n - 1 && starts) starts[g + 1] = n;
}
__global__ __launch_bounds__(256) void level_gather_kernel(const int32_t* first, const int32_t* suf_prev, const int32_t* col, int64_t nk, int32_t* parent,
                                                          int32_t* uidx) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= nk) return;
    const int32_t f = first[k];
    parent[k] = suf_prev[f];
    uidx[k] = col[f];
}
__global__ __launch_bounds__(256) void ones_kernel(double* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = 1.0;
}
// mask of the kept candidates and the masked copy of the cut ones (tnac4o.py:455-465)
__global__ __launch_bounds__(256) void cut_flags_kernel(const double* lp, int64_t n, const double* lmax, double log_cut, int32_t* flag, double* rest) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double cutoff = *lmax + log_cut;
    const bool keep = lp[i] > cutoff;
    flag[i] = keep ? 1 : 0;
    rest[i] = keep ? NEG_INF : lp[i];
}
// scal[dst] = max / min (scal[dst], *src)
__global__ void scalar_max_kernel(double* scal, int dst, const double* src) { scal[dst] = fmax(scal[dst], *src); }
__global__ void scalar_min_kernel(double* scal, int dst, const double* src) { scal[dst] = fmin(scal[dst], *src); }

struct CellDev {                       // what the expansion of a site-step needs of its cell (device pointers)
    const int64_t* down;
    const int64_t* right;
    const double* Es;
    const double* E1;
    const double* E4;
    const int64_t* left_map;
    const int64_t* up_map;
    int64_t q, e1cols, e4cols;
};
// one kept candidate = (parent branch, state of the cell): new boundary indices, energy, keys (tnac4o.py:467-479, 1506-1558)
__global__ __launch_bounds__(256) void expand_kernel(const int32_t* idx, const double* lp, int64_t keep, CellDev c, const int16_t* states, int64_t nsites,
                                                    int64_t pos, int64_t Nx, int has_left, int has_up, const double* Eng, const int32_t* pref,
                                                    const int32_t* suf, int64_t B, int64_t nsuf, int32_t* parent_o, int32_t* child_o, int32_t* down_o,
                                                    int32_t* right_o, double* vals_o, double* E_o, int64_t* pkey_o, int64_t* key_o) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= keep) return;
    const int64_t f = idx[j];
    const int64_t par = f / c.q, ch = f - par * c.q;
    const int64_t dn = c.down[ch], rt = c.right[ch];
    double dE = 1.0 * c.Es[ch];
    if (has_left) {
        const int64_t left = states[par * nsites + pos - 1];
        dE = dE + c.E1[ch * c.e1cols + (c.left_map ? c.left_map[left] : left)];
    }
    if (has_up) {
        const int64_t up = states[par * nsites + pos - Nx];
        dE = dE + c.E4[ch * c.e4cols + (c.up_map ? c.up_map[up] : up)];
    }
    parent_o[j] = (int32_t)par;
    child_o[j] = (int32_t)ch;
    down_o[j] = (int32_t)dn;
    right_o[j] = (int32_t)rt;
    vals_o[j] = lp[f];
    E_o[j] = Eng[par] + dE;
    const int64_t pk = (int64_t)pref[par] * B + dn;
    pkey_o[j] = pk;
    key_o[j] = (pk * B + rt) * nsuf + suf[par];
}
// members of the merge groups in sorted order: energy, log2 p, degeneracy of the parent, candidate position
__global__ __launch_bounds__(256) void merge_gather_kernel(const int32_t* perm, int64_t keep, const double* E, const double* vals, const int32_t* parent,
                                                          const int64_t* deg, double* Es, double* ls, int64_t* ds, int64_t* pos) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= keep) return;
    const int32_t j = perm[i];
    Es[i] = E[j];
    ls[i] = vals[j];
    ds[i] = deg[parent[j]];
    pos[i] = j;
}
__global__ __launch_bounds__(256) void copy_i32_kernel(const int32_t* in, int32_t* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}
// the survivors become the branches of the next site-step (tnac4o.py:511-526): one workgroup per new branch
__global__ __launch_bounds__(256) void commit_kernel(const int32_t* sel, const int64_t* rep, const int64_t* degn, const double* lpn, const int32_t* parent,
                                                    const int32_t* child, const int32_t* down, const int32_t* right, const double* E, const int64_t* pkey,
                                                    const int32_t* vind, const int32_t* sufmat, const int32_t* pref, const int16_t* states, const int64_t cap,
                                                    int64_t ncol, int64_t nlev, int64_t nsites, int64_t nx, int64_t pos, int32_t* vind_n, int32_t* sufmat_n,
                                                    int16_t* states_n, double* Eng_n, double* prob_n, int64_t* deg_n, int64_t* pkey_n, int32_t* prefc_n) {
    const int64_t j = blockIdx.x;
    const int32_t g = sel[j];
    const int64_t r = rep[g];
    const int64_t pr = parent[r];
    const int tid = threadIdx.x;
    for (int64_t c = tid; c < ncol; c += 256) vind_n[c * cap + j] = (c == nx) ? down[r] : (c == nx + 1) ? right[r] : vind[c * cap + pr];
    for (int64_t c = tid; c < nlev; c += 256) sufmat_n[c * cap + j] = sufmat[c * cap + pr];
    for (int64_t c = tid; c < nsites; c += 256) states_n[j * nsites + c] = (c == pos) ? (int16_t)child[r] : states[pr * nsites + c];
    if (tid == 0) {
        Eng_n[j] = E[r];
        prob_n[j] = lpn[g];
        deg_n[j] = degn[g];
        pkey_n[j] = pkey[r];
        prefc_n[j] = pref[pr];
    }
}
__global__ __launch_bounds__(256) void prefix_gather_kernel(const int32_t* nfirst, const int32_t* prefc, const int32_t* col, int64_t npref, int32_t* par,
                                                           int32_t* didx) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= npref) return;
    const int32_t f = nfirst[g];
    par[g] = prefc[f];
    didx[g] = col[f];
}
// end of a row (tnac4o.py:540-542): the down indices of the row become the up indices of the next, column 0 is the open left edge
__global__ __launch_bounds__(256) void shift_columns_kernel(const int32_t* vind, int32_t* vind_n, int64_t cap, int64_t ncol, int64_t nb) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nb * ncol) return;
    const int64_t c = i / nb, b = i % nb;
    vind_n[c * cap + b] = (c == 0) ? 0 : vind[(c - 1) * cap + b];
}

struct Bump {
    char* base = nullptr;
    int64_t cap = 0, off = 0;
    template <typename T>
    T* take(int64_t count) {
        const int64_t o = align_up(off, 256), bytes = count * (int64_t)sizeof(T);
        if (o + bytes > cap) return nullptr;
        off = o + bytes;
        return (T*)(base + o);
    }
};
#define TAKE(ptr, T, bump, count, what)                                                     \
    T* ptr = (bump).take<T>(count);                                                         \
    if (!ptr) { set_error("tn_beam_search: workspace too small (%s)", what); return -3; }

struct Branches {                      // the beam: index rows (column-major), suffix ranks per level, prefix ranks, records
    int32_t* vind;
    int32_t* sufmat;
    int32_t* pref;
    int16_t* states;
    double* Eng;
    double* prob;
    int64_t* deg;
};

struct Search {
    hipStream_t st;
    int64_t Nx, Ny, M, cap, qmax, B;
    void* cub_tmp = nullptr;
    size_t cub_bytes = 0;
    int32_t* iota = nullptr;            // 0 .. cap*qmax-1
    double* scal = nullptr;             // device scalars: [0] pd_max, [1] globalmin, [2] local max, [3] rest max, [4] min of minP
    int32_t* counter = nullptr;         // device: number of selected items

    int read_i32(const int32_t* dev, int32_t& v) {
        int32_t* stage = (int32_t*)pinned_host(8, 7);
        int32_t tmp = 0;
        BSH(hipMemcpyAsync(stage ? stage : &tmp, dev, 4, hipMemcpyDeviceToHost, st), "beam search: read-back");
        BSH(hipStreamSynchronize(st), "beam search: synchronise");
        v = stage ? *stage : tmp;
        return 0;
    }
    // sorted unique of n int64 keys: number of groups (host), inverse, first members, sorted order, group offsets (any may be NULL)
    int unique(Bump scratch, const int64_t* key, int64_t n, int64_t& ng, int32_t* inv, int32_t* first, int32_t* sidx_out, int64_t* starts) {
        TAKE(skey, int64_t, scratch, n, "sorted keys");
        int32_t* sidx = sidx_out;
        if (!sidx) { sidx = scratch.take<int32_t>(n); if (!sidx) { set_error("tn_beam_search: workspace too small (sort order)"); return -3; } }
        TAKE(head, int32_t, scratch, n, "head flags");
        TAKE(gid, int32_t, scratch, n, "group ids");
        size_t tb = cub_bytes;
        BSH(rocprim::radix_sort_pairs(cub_tmp, tb, key, skey, iota, sidx, (int)n, 0, 64, st), "beam search: sort keys");
        const unsigned nblk = (unsigned)cdiv(n, 256);
        hipLaunchKernelGGL(heads_kernel, dim3(nblk), dim3(256), 0, st, skey, n, head);
        TN_CHECK_LAUNCH("heads_kernel");
        tb = cub_bytes;
        BSH(rocprim::inclusive_scan(cub_tmp, tb, head, gid, (size_t)n, rocprim::plus<int32_t>(), st), "beam search: scan");
        hipLaunchKernelGGL(unique_scatter_kernel, dim3(nblk), dim3(256), 0, st, sidx, head, gid, n, inv, first, starts);
        TN_CHECK_LAUNCH("unique_scatter_kernel");
        int32_t g = 0;
        BS(read_i32(gid + (n - 1), g));
        ng = g;
        return 0;
    }
};

}  // namespace

}  // namespace tn

using namespace tn;

extern "C" {

int64_t tn_beam_search_ws_bytes(int64_t Nx, int64_t Ny, int64_t M, int64_t qmax, int64_t max_env, int64_t max_t1, int64_t max_w) {
    const int64_t cap = M, cand = M * qmax, nsites = Nx * Ny;
    int64_t b = 0;
    auto add = [&](int64_t bytes) { b = align_up(b, 256) + bytes; };
    for (int gen = 0; gen < 2; ++gen) {              // the beam, two generations
        add((Nx + 1) * cap * 4); add(Nx * cap * 4); add(cap * 4); add(cap * nsites * 2); add(cap * 8); add(cap * 8); add(cap * 8);
    }
    add(cand * 4); add(64); add(64);                 // iota, scalars, counter
    add((int64_t)64 << 20);                          // rocPRIM temporary storage (checked against its queries at run time)
    // a row: right environments and MPO site of every level, the levels' index scratch, two generations of left environments
    add(256);
    for (int64_t l = 0; l < Nx; ++l) { add(cap * max_env * 8); add(max_w * 8); add(cap * 64 + 4096); }
    add(cap * max_env * 8); add(cap * max_env * 8);
    // a site-step
    add(cap * max_t1 * 8);                           // T1
    add((int64_t)96 << 20);                          // split-K scratch of T1 = RL . A (the product runs unsplit when a larger request does not fit)
    add(cand * 8); add(cand * 8); add(cap * 8);      // conditional tables, log2 p, minima
    add(cand * 4); add(cand * 8); add(cand * 4);     // flags, cut candidates, kept indices
    for (int i = 0; i < 4; ++i) add(cand * 4);       // parent, child, down, right
    for (int i = 0; i < 4; ++i) add(cand * 8);       // log2 p, energy, prefix key, row key
    add(cand * 4); add((cand + 1) * 8);              // group order, group offsets
    add(cand * 8); add(cand * 4); add(cand * 4); add(cand * 4);     // unique: sorted keys, (order), heads, group ids
    for (int i = 0; i < 4; ++i) add(cand * 8);       // members in group order: energy, log2 p, degeneracy, position
    for (int i = 0; i < 3; ++i) add(cand * 8);       // representatives, merged degeneracies, merged log2 p
    add(cand * 8); add(cand * 4); add(cap * 4);      // top M: sorted log2 p, groups by probability, selection
    add(cap * 8); add(cap * 4); add(cap * 4);        // survivors: prefix keys, prefix ranks, first members
    add(cap * 8); add(cap * 4 * 3);                  // unique of the survivors' prefixes
    add(cap * 4); add(cap * 4);                      // prefix parents, down indices
    return b + (1 << 20);
}

int tn_beam_search(int64_t Nx, int64_t Ny, const tn_beam_cell* cells, int64_t M, int has_cut, double log2_cutoff, double min_dEng, int64_t B,
                   int16_t* states_out, double* energy_out, double* log2p_out, int64_t* deg_out, int64_t* nb_host, double* pd_max_host,
                   double* globalmin_host, void* ws, int64_t ws_bytes, void* stream) {
    return tn_beam_search_team(Nx, Ny, cells, M, has_cut, log2_cutoff, min_dEng, B, states_out, energy_out, log2p_out, deg_out, nb_host, pd_max_host,
                               globalmin_host, ws, ws_bytes, stream, 0, 1, nullptr, nullptr);
}

// The walk shared by a team of `team` ranks (one process per GPU, tnac4o.py:444-453 is what they split): every rank holds the whole beam
// and walks every site-step; the conditional tables -- the one part of a site-step whose cost grows with the beam -- are evaluated for the
// rank's contiguous slice of the branches only, then `exchange` (the caller's collective: RCCL / gloo through torch.distributed in
// tnac4o_amd/beam.py) completes log2 p and the minima on every rank; from there on all ranks run the identical deterministic cut, merge
// and selection, so the beams stay replicas without any further traffic.
int tn_beam_search_team(int64_t Nx, int64_t Ny, const tn_beam_cell* cells, int64_t M, int has_cut, double log2_cutoff, double min_dEng, int64_t B,
                        int16_t* states_out, double* energy_out, double* log2p_out, int64_t* deg_out, int64_t* nb_host, double* pd_max_host,
                        double* globalmin_host, void* ws, int64_t ws_bytes, void* stream, int rank, int team, tn_beam_exchange_fn exchange,
                        void* exchange_ctx) {
    TN_CHECK_ARG(Nx >= 1 && Ny >= 1 && M >= 1 && B >= 1 && cells && ws, "bad arguments");
    TN_CHECK_ARG(team >= 1 && rank >= 0 && rank < team && (team == 1 || exchange != nullptr), "bad team (a team of several ranks needs an exchange function)");
    TN_CHECK_ARG(states_out && energy_out && log2p_out && deg_out && nb_host && pd_max_host && globalmin_host, "null result pointer");
    hipStream_t st = (hipStream_t)stream;
    const int64_t nsites = Nx * Ny, cap = M, ncol = Nx + 1;
    int64_t qmax = 1, max_env = 1, max_t1 = 1, max_w = 1;
    for (int64_t i = 0; i < nsites; ++i) {
        const tn_beam_cell& c = cells[i];
        TN_CHECK_ARG(c.q >= 1 && c.q <= 32767 && c.nl >= 1 && c.nu >= 1 && c.pd >= 1 && c.br >= 1 && c.Dl >= 1 && c.p >= 1 && c.Dr >= 1, "bad cell");
        TN_CHECK_ARG(c.p == c.pd, "boundary MPS and PEPS cell disagree on the vertical bond");
        TN_CHECK_ARG(c.Dl * c.nl <= 2048, "Dl x (left PEPS bond) exceeds 2048 (tn_env_rr): use the Python path");
        qmax = std::max(qmax, c.q);
        max_env = std::max(max_env, std::max(c.Dl * c.nl, c.Dr * c.br));
        max_t1 = std::max(max_t1, c.p * c.Dr);
        max_w = std::max(max_w, c.nl * c.pd * c.br * c.nu);
    }
    TN_CHECK_ARG(ws_bytes >= tn_beam_search_ws_bytes(Nx, Ny, M, qmax, max_env, max_t1, max_w), "workspace too small");
    TN_CHECK_ARG(M * qmax < ((int64_t)1 << 31), "too many candidates per site-step");
    const int64_t cand = M * qmax;
    Bump bump;
    bump.base = (char*)ws; bump.cap = ws_bytes;
    Branches gen[2];
    for (int g = 0; g < 2; ++g) {
        gen[g].vind = bump.take<int32_t>(ncol * cap); gen[g].sufmat = bump.take<int32_t>(Nx * cap); gen[g].pref = bump.take<int32_t>(cap);
        gen[g].states = bump.take<int16_t>(cap * nsites); gen[g].Eng = bump.take<double>(cap); gen[g].prob = bump.take<double>(cap);
        gen[g].deg = bump.take<int64_t>(cap);
        TN_CHECK_ARG(gen[g].deg != nullptr, "workspace too small");
    }
    Search S;
    S.st = st; S.Nx = Nx; S.Ny = Ny; S.M = M; S.cap = cap; S.qmax = qmax; S.B = B;
    S.iota = bump.take<int32_t>(cand);
    S.scal = bump.take<double>(8);
    S.counter = bump.take<int32_t>(16);
    S.cub_bytes = (size_t)64 << 20;
    S.cub_tmp = bump.take<char>((int64_t)S.cub_bytes);
    TN_CHECK_ARG(S.cub_tmp != nullptr, "workspace too small");
    {   // the temporary storage rocPRIM asks for at the largest sizes must fit the slot
        size_t need = 0, t = 0;
        (void)rocprim::radix_sort_pairs(nullptr, t, (const int64_t*)nullptr, (int64_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr, (int)cand, 0, 64, st);
        need = std::max(need, t);
        (void)rocprim::radix_sort_pairs_desc(nullptr, t, (const double*)nullptr, (double*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr, (int)cand, 0, 64, st);
        need = std::max(need, t);
        (void)rocprim::inclusive_scan(nullptr, t, (const int32_t*)nullptr, (int32_t*)nullptr, (size_t)cand, rocprim::plus<int32_t>(), st);
        need = std::max(need, t);
        (void)rocprim::select(nullptr, t, (const int32_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int)cand, st);
        need = std::max(need, t);
        (void)rocprim::reduce(nullptr, t, (const double*)nullptr, (double*)nullptr, std::numeric_limits<double>::lowest(), (size_t)cand, rocprim::maximum<double>(), st);
        need = std::max(need, t);
        TN_CHECK_ARG(need <= S.cub_bytes, "rocPRIM temporary storage exceeds its slot");
    }
    hipLaunchKernelGGL(iota_kernel, dim3((unsigned)cdiv(cand, 256)), dim3(256), 0, st, S.iota, cand);
    TN_CHECK_LAUNCH("iota_kernel");
    {   // the root: one branch, all indices 0, log2 p = 0, degeneracy 1; pd_max = -inf, globalmin = 0
        const double h[8] = {NEG_INF, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        double* stage = (double*)pinned_host(64, 7);
        if (stage) memcpy(stage, h, 64);
        BSH(hipMemcpyAsync(S.scal, stage ? stage : h, 64, hipMemcpyHostToDevice, st), "beam search: scalars");
        if (!stage) BSH(hipStreamSynchronize(st), "beam search: synchronise");
        BSH(hipMemsetAsync(gen[0].vind, 0, (size_t)ncol * cap * 4, st), "beam search: clear");
        BSH(hipMemsetAsync(gen[0].states, 0, (size_t)cap * nsites * 2, st), "beam search: clear");
        BSH(hipMemsetAsync(gen[0].Eng, 0, 8, st), "beam search: clear");
        BSH(hipMemsetAsync(gen[0].prob, 0, 8, st), "beam search: clear");
        hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(256), 0, st, (int32_t*)gen[0].deg, (int64_t)2, 0);     // int64 1 = words (1, 0)
        hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(256), 0, st, (int32_t*)gen[0].deg, (int64_t)1, 1);
        TN_CHECK_LAUNCH("fill_i32_kernel");
    }
    int cur = 0;
    int64_t nb = 1;
    const int64_t row_mark = bump.off;
    for (int64_t ny = 0; ny < Ny; ++ny) {
        bump.off = row_mark;
        Branches& br = gen[cur];
        const tn_beam_cell* row = cells + ny * Nx;
        // ---- right environments of every distinct suffix (tnac4o._setup_RR, tnac4o.py:1768-1784): level j serves site Nx-1-j
        std::vector<double*> RRs((size_t)Nx, nullptr);
        std::vector<int64_t> nsuf((size_t)Nx, 1);
        TAKE(rr0, double, bump, 1, "right edge");
        hipLaunchKernelGGL(ones_kernel, dim3(1), dim3(256), 0, st, rr0, (int64_t)1);
        TN_CHECK_LAUNCH("ones_kernel");
        RRs[0] = rr0;
        hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)cdiv(nb, 256)), dim3(256), 0, st, br.sufmat, nb, 0);
        TN_CHECK_LAUNCH("fill_i32_kernel");
        int64_t nkeys_prev = 1;
        for (int64_t nx = Nx - 1; nx >= 1; --nx) {
            const int64_t lvl = Nx - nx;
            const tn_beam_cell& c = row[nx];
            const int32_t* col = br.vind + (nx + 1) * cap;
            const int32_t* suf_prev = br.sufmat + (lvl - 1) * cap;
            int32_t* suf_new = br.sufmat + lvl * cap;
            Bump scratch = bump;                                   // released at the end of the level (a copy: the row keeps bump)
            TAKE(key, int64_t, scratch, nb, "suffix keys");
            TAKE(first, int32_t, scratch, nb, "first members");
            hipLaunchKernelGGL(suffix_key_kernel, dim3((unsigned)cdiv(nb, 256)), dim3(256), 0, st, col, suf_prev, nkeys_prev, nb, key);
            TN_CHECK_LAUNCH("suffix_key_kernel");
            int64_t nk = 0;
            BS(S.unique(scratch, key, nb, nk, suf_new, first, nullptr, nullptr));
            TAKE(parent, int32_t, scratch, nk, "level parents");
            TAKE(uidx, int32_t, scratch, nk, "level up indices");
            hipLaunchKernelGGL(level_gather_kernel, dim3((unsigned)cdiv(nk, 256)), dim3(256), 0, st, first, suf_prev, col, nk, parent, uidx);
            TN_CHECK_LAUNCH("level_gather_kernel");
            // the level's results live until the end of the row: take them from the row's allocator, past the scratch in use
            bump.off = scratch.off;
            TAKE(W, double, bump, c.nl * c.pd * c.br * c.nu, "MPO site");
            TAKE(RR, double, bump, nk * c.Dl * c.nl, "right environments");
            BS(mpo_from_factor(st, c.F, c.dmap, c.rmap, c.q, c.nl, c.nu, c.pd, c.br, W));
            BS(env_rr_batched(st, c.A, RRs[(size_t)lvl - 1], W, parent, uidx, nk, c.Dl, c.p, c.Dr, c.nl, c.br, c.nu, RR));
            RRs[(size_t)lvl] = RR;
            nsuf[(size_t)lvl] = nk;
            nkeys_prev = nk;
        }
        hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)cdiv(nb, 256)), dim3(256), 0, st, br.pref, nb, 0);
        TN_CHECK_LAUNCH("fill_i32_kernel");
        TAKE(RLa, double, bump, cap * max_env, "left environments");
        TAKE(RLb, double, bump, cap * max_env, "left environments");
        double* RL = RLa;
        double* RLn = RLb;
        hipLaunchKernelGGL(ones_kernel, dim3(1), dim3(256), 0, st, RL, (int64_t)1);
        TN_CHECK_LAUNCH("ones_kernel");
        int64_t npref = 1;
        const int64_t step_mark = bump.off;
        for (int64_t nx = 0; nx < Nx; ++nx) {
            bump.off = step_mark;
            Branches& b0 = gen[cur];
            Branches& b1 = gen[cur ^ 1];
            const tn_beam_cell& c = row[nx];
            const int64_t q = c.q, pos = ny * Nx + nx, lvl = Nx - nx - 1, total = nb * q;
            // T1[prefix] = RL[prefix] . A   (tnac4o.py:437-441)
            TAKE(T1, double, bump, npref * c.p * c.Dr, "T1");
            {
                const int64_t gwb = gemm_ws_bytes(npref, c.p * c.Dr, c.Dl, 1);
                double* gws = (gwb > 0 && gwb <= ((int64_t)96 << 20)) ? (double*)bump.take<char>(gwb) : nullptr;      // (else: one pass over K)
                BS(gemm(st, npref, c.p * c.Dr, c.Dl, 1.0, RL, c.Dl, 1, c.A, c.p * c.Dr, 1, 0.0, T1, c.p * c.Dr, 1, 1, 0, 0, 0, gws, gws ? gwb : 0));
            }
            TAKE(Pn, double, bump, total, "conditional tables");
            TAKE(LP, double, bump, total, "log2 p");
            TAKE(mP, double, bump, nb, "minima");
            {
                // this rank's slice of the branches (the whole beam for a team of one); the exchange fills in the partners' slices
                const int64_t lo = team > 1 ? nb * rank / team : 0, hi = team > 1 ? nb * (rank + 1) / team : nb;
                if (hi > lo)
                    BS(calc_pn(st, T1, RRs[(size_t)lvl], c.F, c.dmap, c.rmap, b0.pref + lo, b0.sufmat + lvl * cap + lo, b0.vind + nx * cap + lo,
                               b0.vind + (nx + 1) * cap + lo, hi - lo, q, c.nl, c.nu, c.p, c.Dr, c.br, Pn + lo * q, mP + lo, b0.prob + lo, LP + lo * q));
                if (team > 1) {
                    const int rcx = exchange(exchange_ctx, LP, mP, nb, q, rank, team);
                    if (rcx != 0) { set_error("tn_beam_search_team: the exchange function failed (%d) at site (%lld, %lld)", rcx, (long long)ny, (long long)nx); return -8; }
                }
            }
            size_t tb = S.cub_bytes;
            BSH(rocprim::reduce(S.cub_tmp, tb, mP, S.scal + 4, std::numeric_limits<double>::max(), (size_t)nb, rocprim::minimum<double>(), st), "beam search: minimum");
            hipLaunchKernelGGL(scalar_min_kernel, dim3(1), dim3(1), 0, st, S.scal, 1, S.scal + 4);
            TN_CHECK_LAUNCH("scalar_min_kernel");
            // ---- cut-off against the largest candidate (tnac4o.py:455-465)
            int64_t keep = total;
            const int32_t* idx = S.iota;
            if (has_cut) {
                tb = S.cub_bytes;
                BSH(rocprim::reduce(S.cub_tmp, tb, LP, S.scal + 2, std::numeric_limits<double>::lowest(), (size_t)total, rocprim::maximum<double>(), st), "beam search: maximum");
                TAKE(flag, int32_t, bump, total, "flags");
                TAKE(rest, double, bump, total, "cut candidates");
                TAKE(kept, int32_t, bump, total, "kept candidates");
                hipLaunchKernelGGL(cut_flags_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, LP, total, S.scal + 2, log2_cutoff, flag, rest);
                TN_CHECK_LAUNCH("cut_flags_kernel");
                tb = S.cub_bytes;
                BSH(rocprim::select(S.cub_tmp, tb, S.iota, flag, kept, S.counter, (int)total, st), "beam search: compaction");
                tb = S.cub_bytes;
                BSH(rocprim::reduce(S.cub_tmp, tb, rest, S.scal + 3, std::numeric_limits<double>::lowest(), (size_t)total, rocprim::maximum<double>(), st), "beam search: maximum of the cut");
                int32_t k32 = 0;
                BS(S.read_i32(S.counter, k32));
                keep = k32;
                idx = kept;
                if (keep < total) {
                    hipLaunchKernelGGL(scalar_max_kernel, dim3(1), dim3(1), 0, st, S.scal, 0, S.scal + 3);
                    TN_CHECK_LAUNCH("scalar_max_kernel");
                }
            }
            if (keep <= 0) { set_error("tn_beam_search: no candidate survives the cut-off at site (%lld, %lld)", (long long)ny, (long long)nx); return -6; }
            // ---- the kept candidates: boundary indices, energies, keys
            TAKE(parent, int32_t, bump, keep, "parents");
            TAKE(child, int32_t, bump, keep, "children");
            TAKE(down, int32_t, bump, keep, "down indices");
            TAKE(right, int32_t, bump, keep, "right indices");
            TAKE(vals, double, bump, keep, "log2 p of the kept");
            TAKE(Ec, double, bump, keep, "energies");
            TAKE(pkey, int64_t, bump, keep, "prefix keys");
            TAKE(key, int64_t, bump, keep, "row keys");
            CellDev cd;
            cd.down = c.down; cd.right = c.right; cd.Es = c.Es; cd.E1 = c.E1; cd.E4 = c.E4; cd.left_map = c.left_map; cd.up_map = c.up_map;
            cd.q = q; cd.e1cols = c.e1cols; cd.e4cols = c.e4cols;
            hipLaunchKernelGGL(expand_kernel, dim3((unsigned)cdiv(keep, 256)), dim3(256), 0, st, idx, LP, keep, cd, b0.states, nsites, pos, Nx, nx > 0 ? 1 : 0,
                               ny > 0 ? 1 : 0, b0.Eng, b0.pref, b0.sufmat + lvl * cap, B, nsuf[(size_t)lvl], parent, child, down, right, vals, Ec, pkey, key);
            TN_CHECK_LAUNCH("expand_kernel");
            // ---- merge of equal boundary rows (tnac4o.py:481-515)
            TAKE(perm, int32_t, bump, keep, "group order");
            TAKE(starts, int64_t, bump, keep + 1, "group offsets");
            int64_t ng = 0;
            BS(S.unique(bump, key, keep, ng, nullptr, nullptr, perm, starts));
            TAKE(Es, double, bump, keep, "sorted energies");
            TAKE(ls, double, bump, keep, "sorted log2 p");
            TAKE(ds, int64_t, bump, keep, "sorted degeneracies");
            TAKE(ps, int64_t, bump, keep, "sorted positions");
            hipLaunchKernelGGL(merge_gather_kernel, dim3((unsigned)cdiv(keep, 256)), dim3(256), 0, st, perm, keep, Ec, vals, parent, b0.deg, Es, ls, ds, ps);
            TN_CHECK_LAUNCH("merge_gather_kernel");
            TAKE(rep, int64_t, bump, ng, "representatives");
            TAKE(degn, int64_t, bump, ng, "merged degeneracies");
            TAKE(lpn, double, bump, ng, "merged log2 p");
            BS(merge_groups(st, Es, ls, ds, ps, starts, ng, min_dEng, rep, degn, lpn));
            // ---- the M most probable groups, in group order (tnac4o.py:518-526)
            const int32_t* sel = S.iota;
            int64_t nbn = ng;
            if (ng > M) {
                TAKE(sv, double, bump, ng, "sorted log2 p of the groups");
                TAKE(six, int32_t, bump, ng, "groups by probability");
                TAKE(selb, int32_t, bump, M, "selection");
                tb = S.cub_bytes;
                BSH(rocprim::radix_sort_pairs_desc(S.cub_tmp, tb, lpn, sv, S.iota, six, (int)ng, 0, 64, st), "beam search: sort groups");
                hipLaunchKernelGGL(scalar_max_kernel, dim3(1), dim3(1), 0, st, S.scal, 0, sv + M);
                TN_CHECK_LAUNCH("scalar_max_kernel");
                tb = S.cub_bytes;
                BSH(rocprim::radix_sort_keys(S.cub_tmp, tb, six, selb, (int)M, 0, 32, st), "beam search: selection order");
                sel = selb;
                nbn = M;
            }
            TAKE(pkey_n, int64_t, bump, nbn, "prefix keys of the survivors");
            TAKE(prefc_n, int32_t, bump, nbn, "prefix ranks of the survivors");
            hipLaunchKernelGGL(commit_kernel, dim3((unsigned)nbn), dim3(256), 0, st, sel, rep, degn, lpn, parent, child, down, right, Ec, pkey, b0.vind,
                               b0.sufmat, b0.pref, b0.states, cap, ncol, Nx, nsites, nx, pos, b1.vind, b1.sufmat, b1.states, b1.Eng, b1.prob, b1.deg, pkey_n,
                               prefc_n);
            TN_CHECK_LAUNCH("commit_kernel");
            // ---- left environments of the new distinct prefixes: rows of T1 (tnac4o.py:528-535)
            TAKE(nfirst, int32_t, bump, nbn, "first members of the prefixes");
            int64_t np2 = 0;
            BS(S.unique(bump, pkey_n, nbn, np2, b1.pref, nfirst, nullptr, nullptr));
            TAKE(par, int32_t, bump, np2, "prefix parents");
            TAKE(didx, int32_t, bump, np2, "prefix down indices");
            hipLaunchKernelGGL(prefix_gather_kernel, dim3((unsigned)cdiv(np2, 256)), dim3(256), 0, st, nfirst, prefc_n, b1.vind + nx * cap, np2, par, didx);
            TN_CHECK_LAUNCH("prefix_gather_kernel");
            BS(env_rl_batched(st, T1, par, didx, np2, c.p, c.Dr, RLn));
            std::swap(RL, RLn);
            npref = np2;
            nb = nbn;
            cur ^= 1;
        }
        {   // tnac4o.py:540-542
            Branches& b0 = gen[cur];
            Branches& b1 = gen[cur ^ 1];
            hipLaunchKernelGGL(shift_columns_kernel, dim3((unsigned)cdiv(nb * ncol, 256)), dim3(256), 0, st, b0.vind, b1.vind, cap, ncol, nb);
            TN_CHECK_LAUNCH("shift_columns_kernel");
            std::swap(b0.vind, b1.vind);
        }
    }
    // ---- results
    const Branches& f = gen[cur];
    BSH(hipMemcpyAsync(states_out, f.states, (size_t)nb * nsites * 2, hipMemcpyDeviceToDevice, st), "beam search: results");
    BSH(hipMemcpyAsync(energy_out, f.Eng, (size_t)nb * 8, hipMemcpyDeviceToDevice, st), "beam search: results");
    BSH(hipMemcpyAsync(log2p_out, f.prob, (size_t)nb * 8, hipMemcpyDeviceToDevice, st), "beam search: results");
    BSH(hipMemcpyAsync(deg_out, f.deg, (size_t)nb * 8, hipMemcpyDeviceToDevice, st), "beam search: results");
    double hs[2] = {0.0, 0.0};
    double* stage = (double*)pinned_host(16, 7);
    BSH(hipMemcpyAsync(stage ? stage : hs, S.scal, 16, hipMemcpyDeviceToHost, st), "beam search: scalars");
    BSH(hipStreamSynchronize(st), "beam search: synchronise");
    if (stage) { hs[0] = stage[0]; hs[1] = stage[1]; }
    *nb_host = nb;
    *pd_max_host = hs[0];
    *globalmin_host = hs[1];
    return 0;
}

}  // extern "C"
