// prof.h -- optional per-kernel HIP-event timing on the launch stream (used by bench.py's roofline leg).
// Disabled by default: zero events are created unless envgs_prof_enable(1) was called.
#pragma once
#include <hip/hip_runtime.h>

namespace envgs {
enum KernelId {
    K_PROJECT = 0, K_SCAN, K_EMIT_KEYS, K_SORT, K_RANGES, K_COMPOSITE_FWD, K_COMPOSITE_BWD, K_PROJECT_BWD,
    K_BVH_BUILD, K_TRACE_FWD, K_TRACE_BWD, K_TRACE_COLLECT, K_TRACE_SORT, K_TRACE_COMPOSITE, K_TRACE_KBUF_FWD, K_TRACE_LIST_BWD,
    K_TRACE_KBUF_BWD, K_TRACE_REDUCE, K_TRACE_REGISTER, K_FUSED_ADAM, K_LOSS_FWD, K_LOSS_BWD, K_COUNT
};
void prof_begin(int id, hipStream_t stream);
void prof_end(int id, hipStream_t stream);
struct ProfScope {
    int id; hipStream_t s;
    ProfScope(int id_, hipStream_t s_) : id(id_), s(s_) { prof_begin(id, s); }
    ~ProfScope() { prof_end(id, s); }
};
// Process-global diagnostic switches (envgs_debug_set, include/envgs_raster.h); 0 = production.  The library never reads the environment.
int debug_switch(int which);
// Zero up to ZERO_MAX float buffers in ONE launch (the tracer backward clears eleven gradient outputs: as hipMemsetAsync calls that is eleven
// fill kernels and eleven launch gaps in front of its first kernel).
constexpr int ZERO_MAX = 12;
struct ZeroBatch { float *ptr[ZERO_MAX]; unsigned long long n[ZERO_MAX]; int count; };
int launch_zero_many(const ZeroBatch &b, hipStream_t stream);

}  // namespace envgs
