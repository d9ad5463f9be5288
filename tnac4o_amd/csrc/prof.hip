// Optional per-kernel-family timing with HIP events on the launch stream (used by bench.py for the roofline line).
// Disabled by default: prof_begin/prof_end are then two predictable branches.  When a family is enabled, every launch
// of it is bracketed by an event pair recorded on its own stream.  State is per host thread (one thread drives one
// stream / one chain), so concurrent chains never share a ring; totals are summed over threads on request.
#include <mutex>
#include <vector>

#include "common.h"

namespace tn {

struct Fam {
    uint64_t calls = 0;
    double ms = 0.0, flops = 0.0, bytes = 0.0;
};
struct Pair { hipEvent_t a, b; int fam; };
struct ThreadProf {
    Fam fam[PROF_NFAM];
    std::vector<Pair> ring;
    size_t used = 0;
    void drain() {
        if (used == 0) return;
        hipEventSynchronize(ring[used - 1].b);
        for (size_t i = 0; i < used; ++i) {
            float ms = 0.f;
            hipEventSynchronize(ring[i].b);
            if (hipEventElapsedTime(&ms, ring[i].a, ring[i].b) == hipSuccess) fam[ring[i].fam].ms += ms;
        }
        used = 0;
    }
};

static unsigned g_mask = 0;
static std::mutex g_mu;
static std::vector<ThreadProf*> g_all;

static ThreadProf& mine() {
    thread_local ThreadProf* tp = nullptr;
    if (!tp) {
        tp = new ThreadProf();          // lives for the process (registered below); a handful of threads at most
        std::lock_guard<std::mutex> lk(g_mu);
        g_all.push_back(tp);
    }
    return *tp;
}

bool prof_on(int fam) { return (g_mask >> fam) & 1u; }

void prof_begin(hipStream_t st, int fam) {
    if (!prof_on(fam)) return;
    ThreadProf& t = mine();
    if (t.ring.empty()) {
        t.ring.resize(8192);
        for (auto& p : t.ring) { hipEventCreate(&p.a); hipEventCreate(&p.b); }
    }
    if (t.used == t.ring.size()) t.drain();
    t.ring[t.used].fam = fam;
    hipEventRecord(t.ring[t.used].a, st);
}

void prof_end(hipStream_t st, int fam, double flops, double bytes) {
    if (!prof_on(fam)) return;
    ThreadProf& t = mine();
    hipEventRecord(t.ring[t.used].b, st);
    ++t.used;
    t.fam[fam].calls += 1;
    t.fam[fam].flops += flops;
    t.fam[fam].bytes += bytes;
}

// The three calls below are made while no chain is running (bench.py calls them between phases).
void prof_set_mask(unsigned mask) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto* t : g_all) t->drain();
    g_mask = mask;
}
void prof_reset() {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto* t : g_all) { t->drain(); for (auto& f : t->fam) f = Fam(); }
}
void prof_get(int fam, uint64_t* calls, double* ms, double* flops, double* bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    *calls = 0; *ms = 0; *flops = 0; *bytes = 0;
    for (auto* t : g_all) {
        t->drain();
        *calls += t->fam[fam].calls; *ms += t->fam[fam].ms; *flops += t->fam[fam].flops; *bytes += t->fam[fam].bytes;
    }
}

}  // namespace tn
