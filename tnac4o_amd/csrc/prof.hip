// Optional per-kernel-family timing with HIP events on the launch stream (used by bench.py for the roofline line).
// Disabled by default: prof_begin/prof_end are then two predictable branches.  When enabled for a family, every
// launch of that family is bracketed by an event pair on its own stream; pairs are recycled through a ring.
#include <vector>

#include "common.h"

namespace tn {

struct Fam {
    uint64_t calls = 0;
    double ms = 0.0, flops = 0.0, bytes = 0.0;
};
static Fam g_fam[PROF_NFAM];
static unsigned g_mask = 0;
struct Pair { hipEvent_t a, b; int fam; };
static std::vector<Pair> g_ring;
static size_t g_used = 0;

static void drain() {
    if (g_used == 0) return;
    hipEventSynchronize(g_ring[g_used - 1].b);
    for (size_t i = 0; i < g_used; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_ring[i].a, g_ring[i].b) == hipSuccess) g_fam[g_ring[i].fam].ms += ms;
    }
    g_used = 0;
}

bool prof_on(int fam) { return (g_mask >> fam) & 1u; }

void prof_begin(hipStream_t st, int fam) {
    if (!prof_on(fam)) return;
    if (g_ring.empty()) {
        g_ring.resize(8192);
        for (auto& p : g_ring) { hipEventCreate(&p.a); hipEventCreate(&p.b); }
    }
    if (g_used == g_ring.size()) drain();
    g_ring[g_used].fam = fam;
    hipEventRecord(g_ring[g_used].a, st);
}

void prof_end(hipStream_t st, int fam, double flops, double bytes) {
    if (!prof_on(fam)) return;
    hipEventRecord(g_ring[g_used].b, st);
    ++g_used;
    g_fam[fam].calls += 1;
    g_fam[fam].flops += flops;
    g_fam[fam].bytes += bytes;
}

void prof_set_mask(unsigned mask) { drain(); g_mask = mask; }
void prof_reset() { drain(); for (auto& f : g_fam) f = Fam(); }
void prof_get(int fam, uint64_t* calls, double* ms, double* flops, double* bytes) {
    drain();
    *calls = g_fam[fam].calls; *ms = g_fam[fam].ms; *flops = g_fam[fam].flops; *bytes = g_fam[fam].bytes;
}

}  // namespace tn
