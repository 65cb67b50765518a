// Opt-in per-kernel timing with HIP events recorded on the launch stream (bench.py's live
// roofline measurement).  Disabled by default: zero work on the hot path beyond one branch.
#pragma once
#include <hip/hip_runtime.h>

namespace spn {

enum ProfKernel {
    PK_GEMM_NT = 0, PK_GEMM_TN = 1, PK_ATTN_FWD = 2, PK_ATTN_BWD = 3, PK_BANK_FWD = 4, PK_BANK_BWD = 5,
    PK_LAYERNORM = 6, PK_ADAMW = 7, PK_COUNT = 8
};

extern bool g_prof_on;
void prof_record(int kid, double work, hipStream_t st, bool end);

struct ProfScope {
    int kid; double work; hipStream_t st; bool on;
    ProfScope(int k, double w, hipStream_t s) : kid(k), work(w), st(s), on(g_prof_on) {
        if (on) prof_record(kid, work, st, false);
    }
    ~ProfScope() {
        if (on) prof_record(kid, work, st, true);
    }
};

}  // namespace spn
