// prof.hip -- HIP-event kernel timers behind envgs_prof_* (include/envgs_raster.h).
#include "prof.h"

#include <atomic>
#include <mutex>
#include <vector>

#include "../../include/envgs_raster.h"

namespace envgs {
namespace {
struct Pair { hipEvent_t a, b; };
bool g_on = false;
unsigned long long g_mask = ~0ull;    // envgs_prof_select: the scopes that record while the timers are on
std::mutex g_mu;
std::vector<Pair> g_open[K_COUNT];       // recorded, not yet read
std::vector<Pair> g_pool;                // recycled events
hipEvent_t g_cur[K_COUNT];
bool g_has_cur[K_COUNT];

Pair get_pair()
{
    if (!g_pool.empty()) { Pair p = g_pool.back(); g_pool.pop_back(); return p; }
    Pair p;
    (void)hipEventCreate(&p.a);
    (void)hipEventCreate(&p.b);
    return p;
}
}  // namespace

namespace { std::atomic<int> g_dbg[ENVGS_DBG_COUNT]; }
int debug_switch(int which) { return (which >= 0 && which < ENVGS_DBG_COUNT) ? g_dbg[which].load(std::memory_order_relaxed) : 0; }

void prof_begin(int id, hipStream_t stream)
{
    if (!g_on || !((g_mask >> id) & 1ull)) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Pair p = get_pair();
    (void)hipEventRecord(p.a, stream);
    g_open[id].push_back(p);
    g_has_cur[id] = true;
}

void prof_end(int id, hipStream_t stream)
{
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_has_cur[id] || g_open[id].empty()) return;
    (void)hipEventRecord(g_open[id].back().b, stream);
    g_has_cur[id] = false;
}
__global__ void __launch_bounds__(256) zero_many(const ZeroBatch b)
{
    float *p = b.ptr[blockIdx.y];
    const unsigned long long n = b.n[blockIdx.y];
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) p[i] = 0.f;
}

int launch_zero_many(const ZeroBatch &b, hipStream_t stream)
{
    if (b.count <= 0) return 0;
    hipLaunchKernelGGL(zero_many, dim3(512, b.count), dim3(256), 0, stream, b);
    return (int)hipGetLastError();
}

}  // namespace envgs

using namespace envgs;

extern "C" {

void envgs_prof_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
}

void envgs_prof_select(uint64_t kernel_mask)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_mask = kernel_mask;
}

int envgs_prof_read(int kernel_id, double *total_ms, int *launches)
{
    if (kernel_id < 0 || kernel_id >= K_COUNT || !total_ms || !launches) return ENVGS_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_mu);
    double tot = 0.0;
    int n = 0;
    for (Pair &p : g_open[kernel_id]) {
        if (hipEventSynchronize(p.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { tot += ms; n++; }
        g_pool.push_back(p);
    }
    g_open[kernel_id].clear();
    *total_ms = tot;
    *launches = n;
    return 0;
}

void envgs_debug_set(int32_t which, int32_t value)
{
    if (which >= 0 && which < ENVGS_DBG_COUNT) g_dbg[which].store(value, std::memory_order_relaxed);
}

int32_t envgs_debug_get(int32_t which) { return debug_switch(which); }

const char *envgs_prof_kernel_name(int kernel_id)
{
    static const char *names[K_COUNT] = {"project_surfels", "scan_tiles_touched", "bin_tile_pairs", "sort_tile_lists",
                                         "find_tile_ranges(unused)", "composite_fwd", "composite_bwd", "project_surfels_bwd",
                                         "bvh_build", "trace_fwd", "trace_bwd", "trace.collect_hits", "trace.sort_composite_fwd",
                                         "trace.composite_lists_fwd(unused)", "trace.kbuffer_fwd", "trace.batch_surfel_bwd", "trace.kbuffer_bwd", "trace.reduce_surfel_records", "trace.register_hits", "fused_adam_multi", "l1_ssim_fwd", "l1_ssim_bwd"};
    return (kernel_id >= 0 && kernel_id < K_COUNT) ? names[kernel_id] : "";
}

}  // extern "C"
