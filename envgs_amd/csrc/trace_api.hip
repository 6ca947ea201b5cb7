// trace_api.hip -- the C-ABI of the tracer (include/envgs_trace.h): argument checks, scratch carving, the launch sequences of the forward
// (two batch segments on two streams) and the backward.
#include "trace_common.h"

#include <mutex>


namespace envgs {

constexpr int MAX_SEG = 4;      // forward batch segments that may run concurrently (own stream and fetch counters each)

static int persistent_grid(int R, int per_cu = 8)
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int want = (R + 63) / 64;
    const int cap = cus * per_cu;
    return want < cap ? (want > 0 ? want : 1) : cap;
}

// grid for a grid-stride kernel that handles `per_block` rays per block iteration
static int stride_grid(int R, int per_block)
{
    const int want = (R + per_block - 1) / per_block;
    const int cap = 256 * 32;
    const int g = want < cap ? (want > 0 ? want : 1) : cap;
    return (g + 7) & ~7;                 // multiple of 8: xcd_block() needs it
}

// The list path packs surfel ids, ray slots and per-surfel entry counts into 24-bit fields: beyond 2^24 surfels or rays both directions take
// the K-buffer kernels (correct at any size, slower).
static bool lists_wanted(const envgs_trace_cfg *cfg, const envgs_trace_lists *L)
{
    return L && L->cap > 0 && cfg->max_trace_depth == 0 && cfg->P > 0 && cfg->P < (1 << 24) && cfg->num_rays < (1 << 24);
}
static bool lists_usable(const envgs_trace_cfg *cfg, const envgs_trace_lists *L)
{
    return lists_wanted(cfg, L) && L->hit_lists && L->hit_cnt && L->n_used && L->stack_spill && L->surf_cnt && L->surf_off && L->surf_acc && L->scan_temp;
}
// The backward of the list path reads the hit COUNTS, the per-hit state, the entries / pairs and the per-surfel offsets -- not the lists
// themselves (rays x capacity x 8 B, by far the largest buffer of a call) nor the forward's scratch: the caller may have released those
// (hit_lists == NULL) between the two calls.  Which path a call takes is decided by lists_wanted() ALONE in both directions (a forward that
// fell back to the K-buffer because a scratch pointer was NULL, followed by a backward that found its own pointers complete, read counts nobody
// had written): a struct that asks for lists (cap > 0, sizes in range) with a buffer of its direction missing is ENVGS_ERR_BAD_ARG, never a silent fallback.
static bool lists_usable_bwd(const envgs_trace_cfg *cfg, const envgs_trace_lists *L)
{
    return lists_wanted(cfg, L) && L->hit_cnt && L->n_used && L->surf_cnt && L->surf_off;
}


}  // namespace envgs

using namespace envgs;

static void ray_layout(const envgs_trace_cfg *cfg, int *rh, int *rw)
{
    const bool ok = cfg->ray_h > 0 && cfg->ray_w > 0 && (long long)cfg->ray_h * cfg->ray_w == cfg->num_rays;
    *rh = ok ? cfg->ray_h : 0;
    *rw = ok ? cfg->ray_w : 0;
}

static hipStream_t s_aux[16][MAX_SEG] = {};
static hipEvent_t s_fork[16] = {}, s_join[16][MAX_SEG] = {};
static std::mutex s_mu;                               // the per-device streams / events are created once, under a lock
static hipEvent_t s_def_go[16] = {}, s_def_done[16] = {};
static bool s_def_pending[16] = {};                   // a backward's deferred tail (envgs_trace_lists::defer_reduce) has been queued and nobody has joined it yet

// auxiliary streams + fork / join events of the current device, created on first use (one process drives one GPU in this design, but nothing
// here assumes it); false = not available (the callers then stay on the caller's stream)
static bool aux_objects(int *dev_out)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
    std::lock_guard<std::mutex> lk(s_mu);
    if (!s_fork[dev]) {
        bool ok = hipEventCreateWithFlags(&s_fork[dev], hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&s_def_go[dev], hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&s_def_done[dev], hipEventDisableTiming) == hipSuccess;
        for (int i = 1; i < MAX_SEG && ok; i++)
            ok = hipStreamCreateWithFlags(&s_aux[dev][i], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&s_join[dev][i], hipEventDisableTiming) == hipSuccess;
        if (!ok) { s_fork[dev] = nullptr; return false; }
    }
    *dev_out = dev;
    return true;
}

// `stream` waits for the deferred tail of the last backward, if one is still unjoined (every traced call does this first: the tail reads and
// writes buffers the next call reuses)
static int join_deferred(hipStream_t stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
    std::lock_guard<std::mutex> lk(s_mu);
    if (!s_def_pending[dev]) return 0;
    s_def_pending[dev] = false;
    return hipStreamWaitEvent(stream, s_def_done[dev], 0) == hipSuccess ? 0 : ENVGS_ERR_BAD_ARG;
}

extern "C" {

size_t envgs_trace_ray_sort_temp_bytes(int32_t num_rays) { return ray_sort_temp_bytes(num_rays); }

int envgs_trace_ray_order(int32_t num_rays, const float *ray_o, const float *ray_d, const float *nodes, int32_t P, uint64_t *pairs,
                          uint32_t *order, void *temp, size_t temp_bytes, void *stream)
{
    if (num_rays < 0 || P < 0) return ENVGS_ERR_BAD_ARG;
    if (num_rays == 0) return 0;
    if (!ray_o || !ray_d || !pairs || !order || !temp || (P > 0 && !nodes)) return ENVGS_ERR_BAD_ARG;
    return launch_ray_sort(num_rays, ray_o, ray_d, (const float4 *)nodes, P, pairs, order, temp, temp_bytes, (hipStream_t)stream);
}

// one slab per persistent wavefront of the collection kernels, for each of the (at most two) batch segments that run concurrently
size_t envgs_trace_stack_spill_ints(int32_t num_rays) { return (size_t)2 * persistent_grid(num_rays, 24) * STACK * 64; }

int envgs_trace_forward(const envgs_trace_cfg *cfg, const float *nodes, const float *ray_o, const float *ray_d,
                        const float *means3D, const float *scales, const float *rotations, const float *opacities,
                        const float *shs, const float *colors_precomp, const float *others_precomp, const float *bg,
                        float *srec, uint32_t *counters, float *rgb, float *dpt, float *acc, float *norm, float *dist,
                        float *aux, float *mid, float *wet, float *final_T, const envgs_trace_lists *L, void *stream_)
{
    if (!cfg || cfg->P < 0 || cfg->num_rays < 0 || cfg->sh_degree < 0 || cfg->sh_degree > 3 || cfg->max_trace_depth < 0 || cfg->max_trace_depth > 7)
        return ENVGS_ERR_BAD_ARG;
    if (cfg->num_rays == 0) return 0;
    if (!ray_o || !ray_d || !bg || !counters || !rgb || !dpt || !acc || !norm || !dist || !aux || !mid || !final_T) return ENVGS_ERR_BAD_ARG;
    if (cfg->P > 0 && (!nodes || !means3D || !scales || !rotations || !opacities || !srec || !wet)) return ENVGS_ERR_BAD_ARG;
    if (cfg->P > 0 && (cfg->sh_coeffs > 0 ? (!shs || cfg->sh_coeffs < (cfg->sh_degree + 1) * (cfg->sh_degree + 1)) : !colors_precomp)) return ENVGS_ERR_BAD_ARG;
    if (cfg->has_others && cfg->P > 0 && !others_precomp) return ENVGS_ERR_BAD_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    if (join_deferred(stream)) return ENVGS_ERR_BAD_ARG;
    envgs_raster_cfg dbg; dbg.debug = cfg->debug;
    const envgs_raster_cfg *dcfg = &dbg;
    hipError_t e;
    ForwardPrepare FP;                      // queued as ONE launch further down, once the list path has said whether it wants permuted SH blocks
    std::memset(&FP, 0, sizeof(FP));
    {   // counters, wet, mid and (list path) the per-surfel accumulators
        ZeroBatch &zb = FP.zero; zb.count = 0;
        zb.ptr[zb.count] = reinterpret_cast<float *>(counters); zb.n[zb.count++] = 96;
        if (cfg->P > 0) { zb.ptr[zb.count] = wet; zb.n[zb.count++] = (unsigned long long)cfg->P; }
        zb.ptr[zb.count] = mid; zb.n[zb.count++] = (unsigned long long)cfg->num_rays * MID * (cfg->max_trace_depth + 1);
        if (lists_usable(cfg, L)) { zb.ptr[zb.count] = reinterpret_cast<float *>(L->surf_acc); zb.n[zb.count++] = (unsigned long long)cfg->P * NCOPY * 2; }
        FP.zero_blocks = 512 * zb.count;
    }
    if (cfg->P > 0) {
        FP.P = cfg->P; FP.rec_blocks = (cfg->P + 255) / 256; FP.mod = cfg->scale_modifier;
        FP.means = means3D; FP.scales = scales; FP.rots = rotations; FP.opac = opacities; FP.srec = srec;
    }
    auto launch_prepare = [&](hipStream_t st) -> int {
        hipLaunchKernelGGL(forward_prepare, dim3((unsigned)(FP.zero_blocks + FP.rec_blocks + FP.perm_blocks)), dim3(256), 0, st, FP);
        return (int)hipGetLastError();
    };
    TraceArgs A;
    A = TraceArgs{};
    A.P = cfg->P; A.R = cfg->num_rays; A.D = cfg->sh_degree; A.M = cfg->sh_coeffs; A.ND = cfg->max_trace_depth + 1;
    A.start_from_first = cfg->start_from_first; A.has_others = cfg->has_others; A.bg_len = cfg->bg_len; A.spec_thr = cfg->specular_threshold;
    A.nodes = (const float4 *)nodes; A.srec = (const float4 *)srec; A.shs = shs; A.colors = colors_precomp; A.others = others_precomp;
    A.bg = bg; A.ray_o = ray_o; A.ray_d = ray_d; A.counter = counters; A.stats = (unsigned long long *)(counters + 2);
    A.rgb = rgb; A.dpt = dpt; A.acc = acc; A.norm = norm; A.dist = dist; A.aux = aux; A.mid = mid; A.wet = wet; A.final_T = final_T;
    A.mod = cfg->scale_modifier;
    A.f16 = cfg->feature_f16;
    A.exp = debug_switch(ENVGS_DBG_TRACE);
#ifndef ENVGS_DIAG
    if (A.exp & (8 | 16 | 512 | 2048 | 4096 | 8192)) return ENVGS_ERR_BAD_ARG;      // A/B kernels of the diagnostic build (libenvgs_hip_diag.so) were requested
#endif
    int rh, rw; ray_layout(cfg, &rh, &rw);
    const bool lists = lists_usable(cfg, L);
    if (lists_wanted(cfg, L) && !lists) return ENVGS_ERR_BAD_ARG;      // (see lists_usable_bwd: no silent fallback to the K-buffer path)
    if (L && L->cap > SORT_MAX) return ENVGS_ERR_BAD_ARG;
    ProfScope prof_(K_TRACE_FWD, stream);
    if (lists) {
        if (L->scan_temp_bytes < scan_temp_bytes(cfg->P * NCOPY)) return ENVGS_ERR_TEMP_TOO_SMALL;
        A.hits = (uint2 *)L->hit_lists; A.hit_cnt = L->hit_cnt; A.n_used = L->n_used; A.cap = L->cap; A.stack_spill = L->stack_spill;
        A.surf_cnt = L->surf_cnt; A.surf_off = L->surf_off; A.surf_acc = (unsigned long long *)L->surf_acc;
        const bool will_sort = L->ray_keys && L->ray_order && L->ray_sort_temp && !(A.exp & 64);
        {   // 40-bit fixed-point weight: enough integer bits that even a surfel seen with w = 1 by every ray cannot overflow
            int ib = 1;
            while ((1ll << ib) <= (long long)cfg->num_rays) ib++;
            A.wfrac = 40 - ib > 30 ? 30 : 40 - ib;
        }
        A.state = (float4 *)L->hit_state; A.entries = (unsigned long long *)L->entries; A.pairs = L->pairs; A.n_entries = L->n_entries;
        const bool compact = L->compact_rows > 0 && L->row_off && L->batch_rows && L->row_blk && L->hit_state;
        if (L->compact_rows > 0 && !compact) return ENVGS_ERR_BAD_ARG;
        A.state_plane = compact ? (size_t)L->compact_rows : (size_t)cfg->num_rays * (size_t)L->cap;
        A.colour_state = L->state_planes == 1 ? 1 : 0;
        if (compact) { A.row_off = L->row_off; A.batch_rows = (const uint2 *)L->batch_rows; A.batch_cnt = L->row_blk; }
        if (L->sparse_hits && L->sparse_cap > 0 && L->entries && L->pairs && L->hit_state) {
            // sparse entries (envgs_trace.h: sparse_hits): entries of at most 4 hits are filed per hit (ENVGS_DBG_SPARSE: value - 1 overrides, 1 = off)
            const int sw = debug_switch(ENVGS_DBG_SPARSE);
            A.sparse = (uint4 *)L->sparse_hits;
            A.sparse_cap = (unsigned)(L->sparse_cap > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : L->sparse_cap);
            A.sparse_max = sw > 0 ? (sw - 1 > 64 ? 64 : sw - 1) : 4;
        }
        if (L->sh_perm && shs && cfg->sh_coeffs == 16) {
            const size_t nw = (size_t)cfg->P * 48;
            FP.perm_blocks = (int)((nw + 255) / 256); FP.nb = (cfg->sh_degree + 1) * (cfg->sh_degree + 1); FP.f16 = cfg->feature_f16;
            FP.shs = (const void *)shs; FP.shp = L->sh_perm;
            A.shp = L->sh_perm;
        }
        // The ray batches are split into two segments that run collect -> sort+composite -> register on two streams: the collection
        // kernel is a persistent grid whose wavefronts drain over the time of one whole batch, and the second segment's wavefronts
        // (and the first segment's next kernel) move into the CUs it leaves idle.
        const int nbatch_all = (cfg->num_rays + 63) / 64;
        int nseg = 2;                                         // measured (round 1): 1 -> 18.3 ms / step, 2 -> 17.5, 4 -> 19.4 (each collection launch lasts at least one batch)
        if (debug_switch(ENVGS_DBG_SEGMENTS) > 0) nseg = debug_switch(ENVGS_DBG_SEGMENTS);
        if (nseg > MAX_SEG) nseg = MAX_SEG;                   // (fetch counters: 8 words per segment from counters[32])
        // the per-ray collection kernel (diagnostic: exp & 512, or no coherence sort) spills its stacks into a slab that is sized for two
        // segments (envgs_trace_stack_spill_ints)
        if (nseg > 2 && !(will_sort && !(A.exp & 512))) nseg = 2;
        while (nseg > 1 && nbatch_all / nseg < 256) nseg--;
        if (nseg < 1) nseg = 1;
        hipStream_t aux[MAX_SEG] = {};
        hipEvent_t ev_fork = nullptr, ev_join[MAX_SEG] = {};
        if (nseg > 1) {
            int dev = 0;
            if (!aux_objects(&dev)) nseg = 1;
            if (nseg > 1) {
                ev_fork = s_fork[dev];
                for (int i = 1; i < nseg; i++) { aux[i] = s_aux[dev][i]; ev_join[i] = s_join[dev][i]; }
            }
        }
        // forward_prepare (zero fills, surfel records, permuted SH blocks) reads nothing of the rays: with a second stream at hand it runs
        // there, beside the coherence sort's seven small launches, instead of after them
        const bool prepare_aside = nseg > 1 && will_sort && !(A.exp & 16384);
        if (prepare_aside) {
            if (hipEventRecord(ev_fork, stream) != hipSuccess || hipStreamWaitEvent(aux[1], ev_fork, 0) != hipSuccess) return ENVGS_ERR_BAD_ARG;
            const int rcp = launch_prepare(aux[1]); if (rcp) return rcp;
            if (hipEventRecord(ev_join[1], aux[1]) != hipSuccess) return ENVGS_ERR_BAD_ARG;
            ENVGS_CHECK_LAUNCH(dcfg, aux[1]);
        }
        if (will_sort) {
            // coherence sort of the rays: (key, ray id) pairs bucketed and sorted per bucket (raster_bin.hip: launch_ray_sort); the pairs use
            // ray_keys (2R words = R pairs), the order lands in the second half of ray_order
            const int R = cfg->num_rays;
            const int rc = launch_ray_sort(R, ray_o, ray_d, A.nodes, cfg->P, (uint64_t *)L->ray_keys, L->ray_order + R, L->ray_sort_temp,
                                           L->ray_sort_temp_bytes, stream);
            if (rc) return rc;
            ENVGS_CHECK_LAUNCH(dcfg, stream);
            A.order = L->ray_order + R;
            A.long_list = L->ray_keys;                        // the pairs are dead now: scratch for the queue of long rays
        }
        if (prepare_aside) { if (hipStreamWaitEvent(stream, ev_join[1], 0) != hipSuccess) return ENVGS_ERR_BAD_ARG; }
        else { const int rcp = launch_prepare(stream); if (rcp) return rcp; ENVGS_CHECK_LAUNCH(dcfg, stream); }
        if (nseg > 1) {
            if (hipEventRecord(ev_fork, stream) != hipSuccess) return ENVGS_ERR_BAD_ARG;
            for (int i = 1; i < nseg; i++)
                if (hipStreamWaitEvent(aux[i], ev_fork, 0) != hipSuccess) return ENVGS_ERR_BAD_ARG;
        }
        // Workgroups per CU of the collection's persistent grid.  8 fill the CU -- and then nothing else gets a wavefront slot until the
        // collection drains: with two segments in flight the other segment's sort / register kernels could only move into the CUs a finishing
        // collection leaves.  With HALF the slots (4 workgroups = 16 of 32 waves per CU) the collection itself is slower (1.73 -> 1.89 ms per
        // launch) but the two segments really run side by side: tracer forward 4.80 -> 4.62 ms, step 10.08 -> 9.90 ms (2 / 3 / 4 / 5 / 6 / 7 / 8
        // workgroups: 10.48 / 10.03 / 9.89 / 10.06 / 10.05 / 10.04 / 10.09 ms).  A single segment has nobody to share with: 8.
        const int coop_wgs = (debug_switch(ENVGS_DBG_COLLECT_WGS) > 0 && debug_switch(ENVGS_DBG_COLLECT_WGS) <= 8) ? debug_switch(ENVGS_DBG_COLLECT_WGS)
                                                                                                                  : (nseg > 1 ? 4 : 8);
        for (int sg = 0; sg < nseg; sg++) {                   // segment 0 on the caller's stream, the others on auxiliary streams
            hipStream_t st = sg ? aux[sg] : stream;
            TraceArgs S = A;
            S.seg = sg;
            S.spill_stride = persistent_grid(cfg->num_rays, 24);     // >= this segment's grid; matches envgs_trace_stack_spill_ints
            S.batch0 = (int)((long long)nbatch_all * sg / nseg);
            S.batch1 = (int)((long long)nbatch_all * (sg + 1) / nseg);
            const int rays_seg = (S.batch1 - S.batch0) * 64;
            // (round 3 gave the later segment's collection one more workgroup per CU -- 4+4 / 4+5 / 4+6 / 3+5 / 5+4 workgroups: 9.54 / 9.48 / 9.57 / 9.62 /
            //  9.59 ms per step then.  Round 6, with the collection 15 % faster than the sort pass it runs beside: 4+4 wins, three alternating pairs
            //  7.91 / 7.93 / 7.92 ms against 7.94 / 7.94 / 7.97, configs[4] 50.7 / 50.8 against 50.9 / 51.2 -- the sort pass gets the slots)
            const int seg_wgs = coop_wgs;
            {
                ProfScope p1(K_TRACE_COLLECT, st);
#ifndef ENVGS_DIAG
                // product library: the cooperative collection is the only collection kernel (without a coherence sort its batches are the
                // rays in the order given: correct, slower)
                const dim3 cg(persistent_grid(rays_seg, seg_wgs));
                const float4 *n4 = S.nodes + (size_t)(cfg->P > 1 ? cfg->P - 1 : 1) * 4;
                hipLaunchKernelGGL((collect_hits_coop<false, 8>), cg, dim3(256), 0, st, S, S.nodes, n4, S.srec);
#else
                const dim3 cg(persistent_grid(rays_seg, seg_wgs));
                const float4 *n4 = S.nodes + (size_t)(cfg->P > 1 ? cfg->P - 1 : 1) * 4;
                const bool coop = S.order && !(S.exp & 512) && !(S.exp & 16) && !(S.exp & 2048);
                if (coop && (S.exp & 4096)) hipLaunchKernelGGL((collect_hits_coop<true, 8>), cg, dim3(256), 0, st, S, S.nodes, n4, S.srec);       // A/B: deferred exact tests
                else if (coop && (S.exp & 8192)) hipLaunchKernelGGL((collect_hits_coop<true, 6>), cg, dim3(256), 0, st, S, S.nodes, n4, S.srec);
                else if (coop) hipLaunchKernelGGL((collect_hits_coop<false, 8>), cg, dim3(256), 0, st, S, S.nodes, n4, S.srec);
                else if (S.order && !(S.exp & 512) && !(S.exp & 16))
                    hipLaunchKernelGGL(collect_hits_packet4, dim3(persistent_grid(rays_seg, 24)), dim3(64), 0, st, S, S.nodes,
                                       S.nodes + (size_t)(cfg->P > 1 ? cfg->P - 1 : 1) * 4, S.srec);
                else if (S.order && !(S.exp & 512))
                    hipLaunchKernelGGL(collect_hits_packet, dim3(persistent_grid(rays_seg, 24)), dim3(64), 0, st, S, S.nodes, S.srec);
                else
                    hipLaunchKernelGGL(collect_hits, dim3(persistent_grid(rays_seg, 24)), dim3(64), 0, st, S);
#endif
            }
            ENVGS_CHECK_LAUNCH(dcfg, st);
            if (compact) {
                // row offsets of the compact per-hit buffers: scan of this segment's hit counts in sorted order; the segment owns the share of
                // the rows that corresponds to its share of the batches
                const int nb_seg = S.batch1 - S.batch0, nblk = (rays_seg + 255) / 256;
                unsigned *cntb = L->row_blk;                                       // (batches) row counts, scanned in place per segment
                // the segment claims its rows from a counter shared by the call's segments (counters[22]); its first row lands in counters[28 + seg]
                const unsigned long long limit = (unsigned long long)L->compact_rows;
#ifdef ENVGS_DIAG
                if (!(S.order && !(S.exp & 512) && !(S.exp & 16) && !(S.exp & 2048)))     // the A/B collection kernels do not write batch counts
                    hipLaunchKernelGGL(row_count, dim3(nblk), dim3(256), 0, st, S, cntb);
#endif
                hipLaunchKernelGGL(row_scan_blocks, dim3(1), dim3(256), 0, st, cntb + S.batch0, nb_seg, counters + 22, counters + 28 + sg);
                hipLaunchKernelGGL(row_offsets, dim3(nblk), dim3(256), 0, st, S, cntb, L->row_off, (uint2 *)L->batch_rows, (const unsigned *)(counters + 28 + sg), limit);
                ENVGS_CHECK_LAUNCH(dcfg, st);
            }
            {
                ProfScope p2(K_TRACE_SORT, st);
                const dim3 g(stride_grid(rays_seg, 4)), b(256);
                if (S.shp) hipLaunchKernelGGL((sort_composite_fwd<4, false, true>), g, b, 0, st, S);
                else hipLaunchKernelGGL((sort_composite_fwd<4, false, false>), g, b, 0, st, S);
                if (S.cap > 256) {
                    const dim3 gl(min(stride_grid(rays_seg, 256), 512));
                    if (S.cap <= 512) { if (S.shp) hipLaunchKernelGGL((sort_composite_fwd<8, true, true>), gl, b, 0, st, S); else hipLaunchKernelGGL((sort_composite_fwd<8, true, false>), gl, b, 0, st, S); }
                    else { if (S.shp) hipLaunchKernelGGL((sort_composite_fwd<16, true, true>), gl, b, 0, st, S); else hipLaunchKernelGGL((sort_composite_fwd<16, true, false>), gl, b, 0, st, S); }
                }
            }
            ENVGS_CHECK_LAUNCH(dcfg, st);
            {
                ProfScope p8(K_TRACE_REGISTER, st);
                const dim3 gr(stride_grid(rays_seg, 64)), br(64 * RH_W);
                // (lists of at most 256 hits: each hit's table slot and rank stay in registers between the kernel's phases)
                if (S.pairs && S.cap <= 256) hipLaunchKernelGGL(register_hits<true>, gr, br, 0, st, S);
                else hipLaunchKernelGGL(register_hits<false>, gr, br, 0, st, S);
            }
            ENVGS_CHECK_LAUNCH(dcfg, st);
        }
        for (int i = 1; i < nseg; i++)
            if (hipEventRecord(ev_join[i], aux[i]) != hipSuccess || hipStreamWaitEvent(stream, ev_join[i], 0) != hipSuccess) return ENVGS_ERR_BAD_ARG;
        hipLaunchKernelGGL(unpack_surfel_acc, dim3((cfg->P + 255) / 256), dim3(256), 0, stream, cfg->P, A.wfrac, A.surf_acc, L->surf_cnt, wet,
                           counters);                                          // (also clears the ray-fetch counter of the overflow pass)
        ENVGS_CHECK_LAUNCH(dcfg, stream);
        {   // records of the backward are addressed through the inclusive scan of the per-surfel hit counts
            const int rc = launch_scan(L->surf_cnt, L->surf_off, cfg->P * NCOPY, L->scan_temp, L->scan_temp_bytes, stream);
            if (rc) return rc;
        }
        A.only_overflow = 1;
    } else {
        const int rcp = launch_prepare(stream);
        if (rcp) return rcp;
        ENVGS_CHECK_LAUNCH(dcfg, stream);
    }
    { ProfScope p4(K_TRACE_KBUF_FWD, stream); hipLaunchKernelGGL(trace_fwd, dim3(persistent_grid(cfg->num_rays)), dim3(64), 0, stream, A, rh, rw); }
    ENVGS_CHECK_LAUNCH(dcfg, stream);
    return 0;
}

int envgs_trace_backward(const envgs_trace_cfg *cfg, const float *nodes, const float *ray_o, const float *ray_d,
                         const float *means3D, const float *scales, const float *rotations, const float *opacities,
                         const float *shs, const float *colors_precomp, const float *others_precomp, const float *bg,
                         const float *srec, uint32_t *counters, const float *rgb, const float *dpt, const float *acc,
                         const float *norm, const float *aux, const float *final_T, const float *dL_drgb, const float *dL_ddpt,
                         const float *dL_dacc, const float *dL_dnorm, const float *dL_daux, float *geo_rec, float *dmeans3D,
                         float *dgrads3D, float *dscales, float *drots, float *dopacities, float *dshs, float *dcolors,
                         float *dothers, float *dray_o, float *dray_d, const envgs_trace_lists *L, void *stream_)
{
    if (!cfg || cfg->P < 0 || cfg->num_rays < 0) return ENVGS_ERR_BAD_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    // envgs_trace.h: defer_reduce bits (1 = tail off the caller's stream, 2 = the surfel accumulators hold earlier calls' sums, 4 = leave them unconverted)
    const unsigned dflags = L ? L->defer_reduce : 0u;
    const bool accumulate = (dflags & ENVGS_TRACE_ACCUMULATE) != 0, no_finish = (dflags & ENVGS_TRACE_NO_FINISH) != 0;
    // (a call that continues a chain leaves the previous tail running under its own record kernels and waits for it before its K-buffer pass)
    if (!accumulate && join_deferred(stream)) return ENVGS_ERR_BAD_ARG;
    envgs_raster_cfg dbg; dbg.debug = cfg->debug;
    const envgs_raster_cfg *dcfg = &dbg;
    const size_t P = (size_t)cfg->P, R = (size_t)cfg->num_rays;
    hipError_t e;
    {
        ZeroBatch zb; zb.count = 0;
#define ZERO(buf_, nn) do { if ((buf_) && (nn) > 0) { zb.ptr[zb.count] = (buf_); zb.n[zb.count] = (unsigned long long)(nn); zb.count++; } } while (0)
        if (!accumulate) { ZERO(geo_rec, P * ENVGS_GEOREC_STRIDE); if (cfg->sh_coeffs > 0) ZERO(dshs, P * cfg->sh_coeffs * 3); else ZERO(dcolors, P * 3); }
        if (!no_finish) { ZERO(dmeans3D, P * 3); ZERO(dgrads3D, P * 3); ZERO(dscales, P * 2); ZERO(drots, P * 4); ZERO(dopacities, P); }
        ZERO(dothers, P * 2); ZERO(dray_o, R * 3); ZERO(dray_d, R * 3);
        if (counters && cfg->num_rays > 0 && cfg->P > 0) ZERO(reinterpret_cast<float *>(counters), 1);      // only the ray-fetch counter: [1] (largest list) and the stats stay readable
#undef ZERO
        const int rcz = launch_zero_many(zb, stream);              // one launch instead of eleven fills
        if (rcz) return rcz;
    }
    if (cfg->num_rays == 0 || cfg->P == 0) return 0;
    if (!nodes || !ray_o || !ray_d || !srec || !counters || !rgb || !dpt || !acc || !norm || !aux || !final_T || !geo_rec || !dray_o || !dray_d || !rotations || !bg)
        return ENVGS_ERR_BAD_ARG;
    if (!no_finish && (!dmeans3D || !dscales || !drots || !dopacities)) return ENVGS_ERR_BAD_ARG;
    if (cfg->sh_coeffs > 0 ? (!shs || !dshs) : (!colors_precomp || !dcolors)) return ENVGS_ERR_BAD_ARG;
    TraceArgs A;
    A = TraceArgs{};
    A.P = cfg->P; A.R = cfg->num_rays; A.D = cfg->sh_degree; A.M = cfg->sh_coeffs; A.ND = 1;
    A.start_from_first = cfg->start_from_first; A.has_others = cfg->has_others; A.bg_len = cfg->bg_len; A.spec_thr = cfg->specular_threshold;
    A.nodes = (const float4 *)nodes; A.srec = (const float4 *)srec; A.shs = shs; A.colors = colors_precomp; A.others = others_precomp;
    A.bg = bg; A.ray_o = ray_o; A.ray_d = ray_d; A.counter = counters;
    A.f_rgb = rgb; A.f_dpt = dpt; A.f_acc = acc; A.f_norm = norm; A.f_aux = aux; A.f_T = final_T;
    A.g_rgb = dL_drgb; A.g_dpt = dL_ddpt; A.g_acc = dL_dacc; A.g_norm = dL_dnorm; A.g_aux = dL_daux;
    A.geo_rec = geo_rec; A.dshs = dshs; A.dcolors = dcolors;
    A.exp = debug_switch(ENVGS_DBG_TRACE);
#ifndef ENVGS_DIAG
    if (A.exp & (8 | 16 | 512 | 2048 | 4096 | 8192)) return ENVGS_ERR_BAD_ARG;
#endif
    A.f16 = cfg->feature_f16;
    A.dothers = dothers; A.dray_o = dray_o; A.dray_d = dray_d; A.mod = cfg->scale_modifier;
    int rh, rw; ray_layout(cfg, &rh, &rw);
    bool deferred = false, have_records = false;
    int def_dev = 0;
    {
        ProfScope prof_(K_TRACE_BWD, stream);
        if (lists_wanted(cfg, L) && !lists_usable_bwd(cfg, L)) return ENVGS_ERR_BAD_ARG;
        if (lists_usable_bwd(cfg, L)) {                // lists_wanted(): the forward took the list path exactly when it holds
            A.hits = (uint2 *)L->hit_lists; A.hit_cnt = L->hit_cnt; A.n_used = L->n_used; A.cap = L->cap;
            if (L->ray_keys && L->ray_order && L->ray_sort_temp && !(A.exp & 64)) A.order = L->ray_order + cfg->num_rays;
            if (L->records && L->num_records > 0 && L->surf_cnt && L->surf_off && L->hit_state && L->entries && L->pairs && L->n_entries && !(A.exp & 8)) {
                // atomic-free: one record per (batch, surfel) entry, grouped by surfel; then each surfel's records are summed
                A.surf_cnt = L->surf_cnt; A.surf_off = L->surf_off; A.records = L->records; A.num_records = L->num_records;
                A.state = (float4 *)L->hit_state; A.entries = (unsigned long long *)L->entries; A.pairs = L->pairs; A.n_entries = L->n_entries;
                if (L->compact_rows > 0) {
                    if (!L->row_off || !L->batch_rows) return ENVGS_ERR_BAD_ARG;
                    A.row_off = L->row_off; A.batch_rows = (const uint2 *)L->batch_rows;
                }
                A.state_plane = L->compact_rows > 0 ? (size_t)L->compact_rows : (size_t)cfg->num_rays * (size_t)L->cap;
                {
                    ProfScope p5(K_TRACE_LIST_BWD, stream);
                    // the colour is the only output the loss uses (the EnvGS step): the specialisation that drops the other outputs' terms and state planes
                    const bool rgb_only = dL_drgb && !dL_ddpt && !dL_dacc && !dL_dnorm && !dL_daux;
                    if (L->state_planes == 1 && !rgb_only) return ENVGS_ERR_BAD_ARG;      // the forward was told to keep the colour's plane only
                    const dim3 g(stride_grid((cfg->num_rays + 63) / 64, 1));
                    if (rgb_only) hipLaunchKernelGGL((batch_surfel_bwd<true, false>), g, dim3(64), 0, stream, A);
                    else if (cfg->has_others) hipLaunchKernelGGL((batch_surfel_bwd<false, true>), g, dim3(64), 0, stream, A);
                    else hipLaunchKernelGGL((batch_surfel_bwd<false, false>), g, dim3(64), 0, stream, A);
                    if (L->sparse_hits && L->sparse_cap > 0) {      // the hits of sparse entries, one lane each (adds to the ray gradients stored above)
                        A.sparse = (uint4 *)L->sparse_hits;
                        A.sparse_cap = (unsigned)(L->sparse_cap > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : L->sparse_cap);
                        hipLaunchKernelGGL(sparse_hits_bwd, dim3(2048), dim3(256), 0, stream, A, rgb_only ? 1 : 0);
                    }
                }
                // defer_reduce (envgs_trace.h): the sum of the records and the conversion of the surfel gradients leave the caller's stream -- the
                // ray gradients are complete without them -- and run beside whatever the caller queues next; the K-buffer pass then runs BEFORE
                // the sum, which adds to what it finds
                have_records = true;
                if ((dflags & ENVGS_TRACE_DEFER) && aux_objects(&def_dev)) deferred = true;
                else if (!accumulate) { ProfScope p7(K_TRACE_REDUCE, stream); hipLaunchKernelGGL(reduce_surfel_records, dim3(stride_grid(cfg->P, 16)), dim3(256), 0, stream, A); }
            } else {
#ifdef ENVGS_DIAG
                if (!L->hit_lists) return ENVGS_ERR_BAD_ARG;       // the per-ray atomic-flush backward walks the lists themselves
                ProfScope p5(K_TRACE_LIST_BWD, stream);
                hipLaunchKernelGGL(composite_lists_bwd, dim3(stride_grid(cfg->num_rays, 64)), dim3(64), 0, stream, A);
#else
                // product library: the record backward is the only list backward.  No records = the forward composited nothing on the list
                // path (num_records is its device-side count) -- anything else is a caller error, not a reason to differentiate nothing silently
                if (!(L->surf_cnt && L->surf_off && L->hit_state && L->entries && L->pairs && L->n_entries) || (L->records == nullptr && L->num_records > 0))
                    return ENVGS_ERR_BAD_ARG;
#endif
            }
            A.only_overflow = 1;
        }
        // (the K-buffer pass adds to the accumulators atomically: the previous call's tail, which adds to them with plain read-modify-writes, must be through)
        if (accumulate && join_deferred(stream)) return ENVGS_ERR_BAD_ARG;
        { ProfScope p6(K_TRACE_KBUF_BWD, stream); hipLaunchKernelGGL(trace_bwd, dim3(persistent_grid(cfg->num_rays)), dim3(64), 0, stream, A, rh, rw); }
    }
    ENVGS_CHECK_LAUNCH(dcfg, stream);
    if ((dflags & ENVGS_TRACE_DEFER) && !deferred && aux_objects(&def_dev)) deferred = true;       // (no records: the tail is the conversion alone)
    hipStream_t tail = stream;
    if (deferred) {
        tail = s_aux[def_dev][1];       // (a stream of the lowest priority instead: the sum starves -- 0.27 -> 0.74 ms -- and the join comes later: step 7.57 -> 7.85 ms)
        if (hipEventRecord(s_def_go[def_dev], stream) != hipSuccess || hipStreamWaitEvent(tail, s_def_go[def_dev], 0) != hipSuccess) return ENVGS_ERR_BAD_ARG;
    }
    if (have_records && (deferred || accumulate)) {          // after the K-buffer pass: the sums are added to what it left (and to earlier calls' sums)
        A.reduce_adds = 1;
        // off the caller's stream the sum shares the chip with what the caller queued next (the base pass's compositing backward): a grid of two
        // workgroups per CU instead of one per 16 surfels stretches it (0.26 -> 0.41 ms) and leaves the neighbour its slots -- R7 0.84 -> 0.79 ms,
        // step 7.51 -> 7.45 ms (256 / 512 / 1024 workgroups: 7.44 / 7.45 / 7.48; configs[4] 49.06 / 49.08 against 49.64)
        // (... for the ~2 M records of such a view; the grid grows with the records -- 20 M of them, an incoherent bounce stage over the base set,
        //  took 2.8 ms on 512 workgroups and were still running when the caller joined)
        const unsigned long long rwant = (A.num_records / 4096ull + 7ull) & ~7ull;
        const int rfull = stride_grid(cfg->P, 16);
        const int rg = deferred ? min(rfull, max(512, (int)(rwant < (unsigned long long)rfull ? rwant : (unsigned long long)rfull))) : rfull;
        { ProfScope p7(K_TRACE_REDUCE, tail); hipLaunchKernelGGL(reduce_surfel_records, dim3(rg), dim3(256), 0, tail, A); }
        ENVGS_CHECK_LAUNCH(dcfg, tail);
    }
    if (!no_finish) {
        hipLaunchKernelGGL(finish_surfel_grads, dim3((cfg->P + 255) / 256), dim3(256), 0, tail, cfg->P, rotations, geo_rec, dmeans3D, dscales,
                           dopacities, drots, dgrads3D);
        ENVGS_CHECK_LAUNCH(dcfg, tail);
    }
    if (deferred) {
        std::lock_guard<std::mutex> lk(s_mu);
        if (hipEventRecord(s_def_done[def_dev], tail) != hipSuccess) return ENVGS_ERR_BAD_ARG;
        s_def_pending[def_dev] = true;
    }
    return 0;
}

int envgs_trace_backward_join(void *stream) { return join_deferred((hipStream_t)stream); }

}  // extern "C"

