// Optional per-kernel-family timing with HIP events on the launch stream (used by bench.py for the roofline line).
// Disabled by default: prof_begin/prof_end are then two predictable branches.  When a family is enabled, every launch
// of it is bracketed by an event pair recorded on its own stream.  State is per host thread (one thread drives one
// stream / one chain), so concurrent chains never share a ring; totals are summed over threads on request.  A thread
// that exits folds its totals into a process-wide accumulator and destroys its events (run_concurrent spawns fresh
// threads per call); the event ring starts at 256 pairs and doubles on demand up to 8192.
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
static Fam g_dead[PH_N][PROF_NFAM];    // totals of threads that have exited

struct ThreadSlot {                    // owns the calling thread's state; released when the thread exits
    ThreadProf* tp = nullptr;
    ~ThreadSlot() {
        if (!tp) return;
        std::lock_guard<std::mutex> lk(g_mu);
        tp->drain();
        for (int ph = 0; ph < PH_N; ++ph)
            for (int f = 0; f < PROF_NFAM; ++f) {
                g_dead[ph][f].calls += tp->fam[ph][f].calls; g_dead[ph][f].ms += tp->fam[ph][f].ms;
                g_dead[ph][f].flops += tp->fam[ph][f].flops; g_dead[ph][f].bytes += tp->fam[ph][f].bytes;
            }
        for (auto& p : tp->ring) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        for (size_t i = 0; i < g_all.size(); ++i)
            if (g_all[i] == tp) { g_all[i] = g_all.back(); g_all.pop_back(); break; }
        delete tp;
    }
};

static ThreadProf& mine() {
    thread_local ThreadSlot slot;
    if (!slot.tp) {
        slot.tp = new ThreadProf();
        std::lock_guard<std::mutex> lk(g_mu);
        g_all.push_back(slot.tp);
    }
    return *slot.tp;
}

bool prof_on(int fam) { return fam < PROF_NKERNEL && ((g_mask >> fam) & 1u); }

void prof_begin(hipStream_t st, int fam) {
    if (!prof_on(fam)) return;
    ThreadProf& t = mine();
    t.armed = (t.seq[fam]++ % g_sample) == 0;
    if (!t.armed) return;
    if (t.used == t.ring.size()) {
        if (t.ring.size() < 8192) {                       // grow: 256, 512, ... 8192 event pairs
            const size_t old = t.ring.size(), now = old ? old * 2 : 256;
            t.ring.resize(now);
            for (size_t i = old; i < now; ++i) { (void)hipEventCreate(&t.ring[i].a); (void)hipEventCreate(&t.ring[i].b); }
        } else {
            t.drain();
        }
    }
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
    for (auto& ph : g_dead) for (auto& f : ph) f = Fam();
}
void prof_get(int phase, int fam, uint64_t* calls, double* ms, double* flops, double* bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    *calls = 0; *ms = 0; *flops = 0; *bytes = 0;
    for (int ph = 0; ph < PH_N; ++ph) {
        if (phase >= 0 && ph != phase) continue;
        const Fam& f = g_dead[ph][fam];
        *calls += f.calls; *ms += f.ms; *flops += f.flops; *bytes += f.bytes;
    }
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
