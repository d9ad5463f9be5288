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
struct Pair { hipEvent_t a, b; int fam, ph; };
struct ThreadProf {
    Fam fam[PH_N][PROF_NFAM];
    int phase = PH_OTHER;
    unsigned seq[PROF_NFAM] = {};      // launches seen per family (for 1-in-n sampling)
    bool armed = false;                // the launch in flight is bracketed
    std::vector<Pair> ring;
    size_t used = 0;
    void drain() {
        if (used == 0) return;
        (void)hipEventSynchronize(ring[used - 1].b);
        for (size_t i = 0; i < used; ++i) {
            float ms = 0.f;
            (void)hipEventSynchronize(ring[i].b);
            if (hipEventElapsedTime(&ms, ring[i].a, ring[i].b) == hipSuccess) fam[ring[i].ph][ring[i].fam].ms += ms;
        }
        used = 0;
    }
};

static unsigned g_mask = 0;
static unsigned g_sample = 1;          // bracket every g_sample-th launch of an enabled family
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

bool prof_on(int fam) { return fam < PROF_NKERNEL && ((g_mask >> fam) & 1u); }

void prof_begin(hipStream_t st, int fam) {
    if (!prof_on(fam)) return;
    ThreadProf& t = mine();
    t.armed = (t.seq[fam]++ % g_sample) == 0;
    if (!t.armed) return;
    if (t.ring.empty()) {
        t.ring.resize(8192);
        for (auto& p : t.ring) { (void)hipEventCreate(&p.a); (void)hipEventCreate(&p.b); }
    }
    if (t.used == t.ring.size()) t.drain();
    t.ring[t.used].fam = fam;
    t.ring[t.used].ph = t.phase;
    (void)hipEventRecord(t.ring[t.used].a, st);
}

void prof_end(hipStream_t st, int fam, double flops, double bytes) {
    if (!prof_on(fam)) return;
    ThreadProf& t = mine();
    if (!t.armed) return;
    t.armed = false;
    (void)hipEventRecord(t.ring[t.used].b, st);
    ++t.used;
    Fam& f = t.fam[t.phase][fam];
    f.calls += 1;
    f.flops += flops;
    f.bytes += bytes;
}

void prof_note(int fam, double calls, double flops, double bytes) {
    if (g_mask == 0) return;
    ThreadProf& t = mine();
    Fam& f = t.fam[t.phase][fam];
    f.calls += (uint64_t)calls;
    f.flops += flops;
    f.bytes += bytes;
}

int prof_phase(int phase) {
    if (g_mask == 0) return PH_OTHER;      // profiling off: no thread-local traffic
    ThreadProf& t = mine();
    const int prev = t.phase;
    t.phase = phase;
    return prev;
}

// The three calls below are made while no chain is running (bench.py calls them between phases).
void prof_set_mask(unsigned mask) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto* t : g_all) t->drain();
    g_mask = mask;
}
void prof_set_sample(unsigned n) { g_sample = n ? n : 1; }
void prof_reset() {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto* t : g_all) { t->drain(); for (auto& ph : t->fam) for (auto& f : ph) f = Fam(); }
}
void prof_get(int phase, int fam, uint64_t* calls, double* ms, double* flops, double* bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    *calls = 0; *ms = 0; *flops = 0; *bytes = 0;
    for (auto* t : g_all) {
        t->drain();
        for (int ph = 0; ph < PH_N; ++ph) {
            if (phase >= 0 && ph != phase) continue;
            const Fam& f = t->fam[ph][fam];
            *calls += f.calls; *ms += f.ms; *flops += f.flops; *bytes += f.bytes;
        }
    }
}

}  // namespace tn
