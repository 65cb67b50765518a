#include "prof.h"
#include "../../include/spn4cir_hip.h"
#include <vector>

namespace spn {

bool g_prof_on = false;

struct ProfRec { int kid; double work; hipEvent_t a, b; bool closed; };
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static size_t g_pool_next = 0;

void prof_record(int kid, double work, hipStream_t st, bool end) {
    if (!end) {
        if (g_pool_next + 2 > g_pool.size()) return;   // pool exhausted: stop recording
        ProfRec r{kid, work, g_pool[g_pool_next], g_pool[g_pool_next + 1], false};
        g_pool_next += 2;
        (void)hipEventRecord(r.a, st);
        g_recs.push_back(r);
    } else {
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
    g_prof_on = true;
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
