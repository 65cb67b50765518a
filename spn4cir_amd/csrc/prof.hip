#include "prof.h"
#include "../../include/spn4cir_hip.h"
#include <vector>

namespace spn {

bool g_prof_on = false;

struct ProfRec { int kid; double work; hipEvent_t a, b; bool closed; };
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static size_t g_pool_next = 0;
static unsigned g_mask = ~0u;          // kernel classes being recorded
static int g_every = 1;                // ... every g_every-th launch of each
static int g_seen[PK_COUNT];
static bool g_open[PK_COUNT];          // the current launch of the class is being recorded

void prof_record(int kid, double work, hipStream_t st, bool end) {
    if (kid < 0 || kid >= PK_COUNT || !((g_mask >> kid) & 1u)) return;
    if (!end) {
        g_open[kid] = (g_seen[kid]++ % g_every) == 0;
        if (!g_open[kid]) return;
        if (g_pool_next + 2 > g_pool.size()) { g_open[kid] = false; return; }   // pool exhausted: stop recording
        ProfRec r{kid, work, g_pool[g_pool_next], g_pool[g_pool_next + 1], false};
        g_pool_next += 2;
        (void)hipEventRecord(r.a, st);
        g_recs.push_back(r);
    } else {
        if (!g_open[kid]) return;
        g_open[kid] = false;
        for (size_t i = g_recs.size(); i-- > 0;) {
            if (g_recs[i].kid == kid && !g_recs[i].closed) {
                (void)hipEventRecord(g_recs[i].b, st);
                g_recs[i].closed = true;
                return;
            }
        }
    }
}

}  // namespace spn

using namespace spn;

extern "C" {

int spn_prof_enable(int max_records) {
    if (max_records <= 0) return SPN_ERR_ARG;
    while (g_pool.size() < (size_t)max_records * 2) {
        hipEvent_t e;
        hipError_t rc = hipEventCreate(&e);
        if (rc != hipSuccess) return (int)rc;
        g_pool.push_back(e);
    }
    g_recs.clear();
    g_pool_next = 0;
    for (int i = 0; i < PK_COUNT; ++i) { g_seen[i] = 0; g_open[i] = false; }
    g_prof_on = true;
    return 0;
}

/* Restrict the recording to the kernel classes in `mask` (bit = ProfKernel id) and to every `sample_every`-th
 * launch of each: an event pair around a launch keeps it from overlapping its neighbours' ramp-up / drain, so a
 * bench that times every kernel slows the step it measures (7 % on config 2). */
int spn_prof_select(unsigned mask, int sample_every) {
    if (sample_every <= 0) return SPN_ERR_ARG;
    g_mask = mask;
    g_every = sample_every;
    for (int i = 0; i < PK_COUNT; ++i) { g_seen[i] = 0; g_open[i] = false; }
    return 0;
}

int spn_prof_disable(void) {
    g_prof_on = false;
    return 0;
}

int spn_prof_reset(void) {
    g_recs.clear();
    g_pool_next = 0;
    return 0;
}

/* Sums over the closed records of one kernel class; synchronises on their events. */
int spn_prof_collect(int kernel_id, double* total_ms, double* total_work, int* count) {
    double ms = 0, work = 0;
    int n = 0;
    for (auto& r : g_recs) {
        if (r.kid != kernel_id || !r.closed) continue;
        hipError_t rc = hipEventSynchronize(r.b);
        if (rc != hipSuccess) return (int)rc;
        float t = 0.f;
        rc = hipEventElapsedTime(&t, r.a, r.b);
        if (rc != hipSuccess) return (int)rc;
        ms += t; work += r.work; ++n;
    }
    if (total_ms) *total_ms = ms;
    if (total_work) *total_work = work;
    if (count) *count = n;
    return 0;
}

}  // extern "C"
