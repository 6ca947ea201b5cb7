// raster_bin.hip -- R2..R5, hand-written for gfx950: prefix sum of tiles_touched, and the per-tile depth-ordered surfel lists.
//
// The reference's extension bins with CUB: scan -> emit (tile id << 32 | depth bits, surfel id) -> device-wide stable radix sort of the
// N pairs over 32 + log2(tiles) key bits -> range detection.  The RESULT that contract fixes is: per tile, the surfels whose 3-sigma
// rectangle covers it, ordered by (depth bits, emission order) -- and emission order within one tile is surfel-index order.  So the list
// of tile t is exactly "its instances sorted by the 64-bit value (depth bits << 32 | surfel id)", and that is what is built here,
// without a device-wide sort (N = millions of 12 B pairs through ~6 radix passes = 144 B per instance):
//
//   bin_pass<count>     each workgroup histograms ITS slice of the surfels over the tiles in LDS (LDS atomics only) and stores the
//                       histogram row                                   hist[w][t]           (no global atomics anywhere)
//   bin_column_scan     per tile, exclusive scan down the rows          hist[w][t] -> first slot of workgroup w inside tile t's segment
//   bin_tile_scan       exclusive scan over the tile totals             tile_start[t], ranges[t] = [start, start + count)   (= R5)
//   bin_pass<scatter>   the same walk again; an LDS cursor per tile (tile_start + row prefix) hands out slots: pair -> its tile's segment
//   sort_tile_lists     one workgroup per tile: the segment is sorted in LDS by a bitonic network whose compare-exchange steps run four at
//                       a time in registers (tile_sort.h), and written out as point_list (ids) + keys_sorted (tile id << 32 | depth);
//                       the LDS array (2048 / 4096 / 8192 entries) is picked from the average list length, longer segments go to
//                       sort_long_lists (16 384 entries in LDS; beyond that chunks go through LDS and only the wide merge steps run in HBM)
//
// Per instance that is 8 B written + 8 B read + 12 B written.  All integer work: point_list / keys_sorted / ranges are bit-exact against
// the oracle's stable sort (tests/test_raster_parity.py, tests/test_tile_binning.py).
//
// The same machinery (histogram rows -> column scan -> bucket scan -> scatter -> per-bucket LDS sort) also orders the tracer's rays
// (launch_ray_sort: buckets = top bits of the coherence key) and the LBVH's Morton keys (launch_key_sort), at the end of this file: the
// library contains no other scan or sort.
#include "common.h"
#include "ray_key.h"
#include "tile_sort.h"

namespace envgs {

// ---- wave64 integer scan / sum (DPP, same lane patterns as common.h's float versions) ------------
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_fill_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_scan_add_u32(uint32_t v) {
    v += dpp_fill_u32<0x111>(v); v += dpp_fill_u32<0x112>(v); v += dpp_fill_u32<0x114>(v); v += dpp_fill_u32<0x118>(v);
    v += dpp_fill_u32<0x142, 0xa>(v);
    v += dpp_fill_u32<0x143, 0xc>(v);
    return v;
}

// Exclusive prefix of `v` over the workgroup's NW wavefronts (thread order), and the workgroup total.  s_w: NW + 1 words of LDS.
template <int NW>
__device__ __forceinline__ uint32_t block_exclusive_u32(uint32_t v, uint32_t *s_w, uint32_t &total)
{
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t inc = wave_scan_add_u32(v);
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) { const uint32_t t = s_w[k]; all += t; before += (k < wave) ? t : 0u; }
    total = all;
    __syncthreads();
    return before + inc - v;
}

// ---- R2: inclusive prefix sum over n uint32 counters (two launches: block sums, then apply) ------
constexpr int SCAN_ITEMS = 1024;                 // per workgroup: 256 lanes x 4 consecutive counters

size_t scan_temp_bytes(int n) { return sizeof(uint32_t) * (size_t)(((n > 0 ? n : 1) + SCAN_ITEMS - 1) / SCAN_ITEMS) + 64; }

__global__ void __launch_bounds__(256)
scan_block_sums(const uint32_t *__restrict__ in, uint32_t *__restrict__ sums, int n)
{
    __shared__ uint32_t s_w[5];
    const int base = blockIdx.x * SCAN_ITEMS + threadIdx.x * 4;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) s += (base + k < n) ? in[base + k] : 0u;
    uint32_t total;
    (void)block_exclusive_u32<4>(s, s_w, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256)
scan_apply(const uint32_t *in, uint32_t *out, const uint32_t *__restrict__ sums, int n)       // in == out allowed (each lane loads its counters before it stores)
{
    __shared__ uint32_t s_w[5];
    uint32_t pre = 0;
    for (int k = threadIdx.x; k < (int)blockIdx.x; k += 256) pre += sums[k];
    uint32_t before;
    (void)block_exclusive_u32<4>(pre, s_w, before);                                            // `before` = sum of all earlier workgroups
    const int base = blockIdx.x * SCAN_ITEMS + threadIdx.x * 4;
    uint32_t a[4];
#pragma unroll
    for (int k = 0; k < 4; k++) a[k] = (base + k < n) ? in[base + k] : 0u;
    a[1] += a[0]; a[2] += a[1]; a[3] += a[2];
    uint32_t total;
    const uint32_t ex = block_exclusive_u32<4>(a[3], s_w, total) + before;
#pragma unroll
    for (int k = 0; k < 4; k++) if (base + k < n) out[base + k] = ex + a[k];
}

int launch_scan(const uint32_t *in, uint32_t *out, int n, void *temp, size_t temp_bytes, hipStream_t stream)
{
    if (n <= 0) return 0;
    if (temp_bytes < scan_temp_bytes(n)) return ENVGS_ERR_TEMP_TOO_SMALL;
    ProfScope prof_(K_SCAN, stream);
    const int nb = (n + SCAN_ITEMS - 1) / SCAN_ITEMS;
    hipLaunchKernelGGL(scan_block_sums, dim3(nb), dim3(256), 0, stream, in, (uint32_t *)temp, n);
    hipLaunchKernelGGL(scan_apply, dim3(nb), dim3(256), 0, stream, in, out, (const uint32_t *)temp, n);
    return (int)hipGetLastError();
}

// ---- R3..R5 ---------------------------------------------------------------------------------------
constexpr int BIN_ROWS_MAX = 512;                // histogram rows (= workgroups of bin_pass) at most
constexpr int BIN_SLICE_MIN = 1024;              // surfels per workgroup at least
constexpr int BIN_BAND = 16384;                  // tiles per LDS histogram (64 KB); larger images are walked band by band
constexpr int BIG_RECT = 32;                     // rectangles above this many tiles are walked by the whole wavefront
constexpr int SORT_LONG_N = 16384;               // sort_long_lists: 132 KB of LDS
constexpr int SORT_LONG_WGS = 256;               // one per CU

struct BinPlan { int rows, slice, ntiles; };
static BinPlan bin_plan(int P, int W, int H)
{
    BinPlan p;
    p.ntiles = ((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    int rows = (P + BIN_SLICE_MIN - 1) / BIN_SLICE_MIN;
    rows = rows < 1 ? 1 : (rows > BIN_ROWS_MAX ? BIN_ROWS_MAX : rows);
    p.slice = (((P + rows - 1) / rows + 255) / 256) * 256;
    p.rows = p.slice > 0 ? (P + p.slice - 1) / p.slice : 1;
    if (p.rows < 1) p.rows = 1;
    return p;
}

// Scratch of one binning call, carved from the caller's temp buffer (uint32 words):
//   hist (BIN_ROWS_MAX x ntiles) | tile_count (ntiles) | tile_start (ntiles + 1) | long_list (ntiles) | hdr (4: total, long count, -, -)
size_t sort_temp_bytes(uint32_t, int width, int height)
{
    const size_t ntiles = (size_t)((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
    return sizeof(uint32_t) * ((size_t)BIN_ROWS_MAX * ntiles + 3 * ntiles + 1 + 4) + 256;
}

// The tile rectangle of surfel i, recomputed from the stored centre and INTEGER radius exactly as R1 did.
__device__ __forceinline__ bool tile_rect(int i, int W, int H, const float *__restrict__ geom, const int32_t *__restrict__ radii,
                                          int &x0, int &y0, int &x1, int &y1, uint32_t &dbits)
{
    const int rad = radii[i];
    if (rad <= 0) return false;
    const float cx = geom[(size_t)i * GEOM + 9], cy = geom[(size_t)i * GEOM + 10];
    dbits = __float_as_uint(geom[(size_t)i * GEOM + 15]);
    const float radius = (float)rad;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    x0 = (int)((cx - radius) / (float)TILE); x0 = x0 < 0 ? 0 : (x0 > gx ? gx : x0);
    y0 = (int)((cy - radius) / (float)TILE); y0 = y0 < 0 ? 0 : (y0 > gy ? gy : y0);
    x1 = (int)((cx + radius + (float)(TILE - 1)) / (float)TILE); x1 = x1 < 0 ? 0 : (x1 > gx ? gx : x1);
    y1 = (int)((cy + radius + (float)(TILE - 1)) / (float)TILE); y1 = y1 < 0 ? 0 : (y1 > gy ? gy : y1);
    return x1 > x0 && y1 > y0;
}

// One workgroup per slice of `slice` surfels.  SCATTER == false: hist[w][t] = instances of this slice in tile t.  SCATTER == true: the
// LDS counters start at the slice's first slot inside each tile's segment and every instance takes the next one.
struct SurfelRect { int x0, y0, x1, y1; uint32_t dbits; bool vis; };
__device__ __forceinline__ SurfelRect load_rect(int i, int g1, int W, int H, const float *__restrict__ geom, const int32_t *__restrict__ radii)
{
    SurfelRect r;
    r.x0 = r.y0 = r.x1 = r.y1 = 0; r.dbits = 0;
    r.vis = i < g1 && tile_rect(i, W, H, geom, radii, r.x0, r.y0, r.x1, r.y1, r.dbits);
    return r;
}

template <bool SCATTER>
__global__ void __launch_bounds__(256)
bin_pass(int P, int W, int H, int slice, int ntiles, const float *__restrict__ geom, const int32_t *__restrict__ radii,
         uint32_t *hist, const uint32_t *__restrict__ tile_start, uint64_t *__restrict__ pairs, uint32_t cap)
{
    extern __shared__ uint32_t s_bin[];
    const int gx = (W + TILE - 1) / TILE;
    const int g0 = blockIdx.x * slice, g1 = min(P, g0 + slice);
    uint32_t *row = hist + (size_t)blockIdx.x * ntiles;
    const int lane = lane_id();
    for (int band0 = 0; band0 < ntiles; band0 += BIN_BAND) {
        const int bn = min(BIN_BAND, ntiles - band0);
        SurfelRect cur = load_rect(g0 + (int)threadIdx.x, g1, W, H, geom, radii);          // (in flight while the counters are set up)
        for (int t = threadIdx.x; t < bn; t += 256) s_bin[t] = SCATTER ? tile_start[band0 + t] + row[band0 + t] : 0u;
        __syncthreads();
        auto visit = [&](int t, uint64_t pair) {
            t -= band0;
            if ((unsigned)t >= (unsigned)bn) return;
            const uint32_t slot = atomicAdd(&s_bin[t], 1u);
            if (SCATTER && slot < cap) pairs[slot] = pair;           // (cap below the instance count: the caller repeats the call; nothing out of bounds)
        };
        for (int ib = g0; ib < g1; ib += 256) {
            const int i = ib + (int)threadIdx.x;
            const SurfelRect c = cur;
            cur = load_rect(i + 256, g1, W, H, geom, radii);          // the next surfel's record is fetched before this one's tiles are walked
            const int x0 = c.x0, y0 = c.y0, x1 = c.x1, y1 = c.y1;
            const uint32_t dbits = c.dbits;
            const bool vis = c.vis;
            const uint64_t pair = ((uint64_t)dbits << 32) | (uint32_t)i;
            const bool big = vis && (x1 - x0) * (y1 - y0) > BIG_RECT;
            if (vis && !big)
                for (int y = y0; y < y1; y++)
                    for (int x = x0; x < x1; x++) visit(y * gx + x, pair);
            uint64_t m = __builtin_amdgcn_ballot_w64(big);
            while (m) {                                               // a large splat would stall its 63 neighbours: all lanes share its tiles
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                const int bx0 = __builtin_amdgcn_readlane(x0, l), by0 = __builtin_amdgcn_readlane(y0, l);
                const int bw = __builtin_amdgcn_readlane(x1, l) - bx0, bh = __builtin_amdgcn_readlane(y1, l) - by0;
                const uint64_t bp = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)dbits, l) << 32) | (uint32_t)__builtin_amdgcn_readlane(i, l);
                for (int k = lane; k < bw * bh; k += 64) visit((by0 + k / bw) * gx + bx0 + k % bw, bp);
            }
        }
        __syncthreads();
        if (!SCATTER)
            for (int t = threadIdx.x; t < bn; t += 256) row[band0 + t] = s_bin[t];
        __syncthreads();
    }
}

// 16 tiles x 16 row segments per workgroup: hist[w][t] becomes the number of instances of tile t in the slices before w.  Every lane keeps
// its (at most 32) rows in registers: one read of the matrix, all loads in flight before the first store.
constexpr int CS_SEG = 16, CS_ROWS = BIN_ROWS_MAX / CS_SEG;
__global__ void __launch_bounds__(256)
bin_column_scan(int rows, int ntiles, uint32_t *hist, uint32_t *__restrict__ tile_count)
{
    __shared__ uint32_t s_seg[CS_SEG][16];
    const int tl = threadIdx.x & 15, seg = threadIdx.x >> 4;
    const int t = blockIdx.x * 16 + tl;
    const int per = (rows + CS_SEG - 1) / CS_SEG, r0 = seg * per;
    uint32_t v[CS_ROWS];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < CS_ROWS; k++) {
        const int r = r0 + k;
        v[k] = (t < ntiles && k < per && r < rows) ? hist[(size_t)r * ntiles + t] : 0u;
        s += v[k];
    }
    s_seg[seg][tl] = s;
    __syncthreads();
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < CS_SEG; k++) acc += (k < seg) ? s_seg[k][tl] : 0u;
    if (t >= ntiles) return;
#pragma unroll
    for (int k = 0; k < CS_ROWS; k++) {
        const int r = r0 + k;
        if (k < per && r < rows) hist[(size_t)r * ntiles + t] = acc;
        acc += v[k];
    }
    if (seg == CS_SEG - 1) tile_count[t] = acc;
}

// One workgroup: exclusive scan of the tile totals -> tile_start, ranges (R5; empty tiles read [0,0) like the reference's zeroed buffer).
__global__ void __launch_bounds__(1024)
bin_tile_scan(int ntiles, const uint32_t *__restrict__ tile_count, uint32_t *__restrict__ tile_start, uint32_t *__restrict__ ranges,
              uint32_t cap, uint32_t *__restrict__ hdr)
{
    __shared__ uint32_t s_w[17];
    const int per = (ntiles + 1023) / 1024, b = min(ntiles, (int)threadIdx.x * per), e = min(ntiles, b + per);
    uint32_t s = 0;
    for (int k = b; k < e; k++) s += tile_count[k];
    uint32_t total;
    uint32_t run = block_exclusive_u32<16>(s, s_w, total);
    const bool ok = total <= cap;                      // otherwise the caller repeats the call with the exact size: leave nothing to composite
    for (int k = b; k < e; k++) {
        const uint32_t c = tile_count[k];
        tile_start[k] = run;
        ranges[2 * k] = (c && ok) ? run : 0u;
        ranges[2 * k + 1] = (c && ok) ? run + c : 0u;
        run += c;
    }
    if (threadIdx.x == 0) { tile_start[ntiles] = total; hdr[0] = total; hdr[1] = 0u; hdr[2] = 0u; hdr[3] = 0u; }      // [1] long lists; sort_long_lists: [2] items claimed, [3] items done
}

// Sort the power-of-two padded LDS array of 1 << lp entries (tile_sort.h: register-blocked bitonic network, one barrier per 4 steps;
// entry i lives in slot ts_slot(i)).
template <int NT>
__device__ __forceinline__ void sort_padded_lds(uint64_t *s, int lp, int tid)
{
    const int npad = 1 << lp;
    for (int g = tid; g < (npad >> TS_S); g += NT) bitonic_first(s, g, lp);
    __syncthreads();
    for (int lk = TS_S + 1; lk <= lp; lk++)
        for (int a = lk - 1; a >= 0;) {
            const int S = ts_chunk(a);
            if (S == 4) { for (int g = tid; g < (npad >> 4); g += NT) bitonic_group<4>(s, g, lk, a, lp); }
            else if (S == 3) { for (int g = tid; g < (npad >> 3); g += NT) bitonic_group<3>(s, g, lk, a, lp); }
            else if (S == 2) { for (int g = tid; g < (npad >> 2); g += NT) bitonic_group<2>(s, g, lk, a, lp); }
            else { for (int g = tid; g < (npad >> 1); g += NT) bitonic_group<1>(s, g, lk, a, lp); }
            __syncthreads();
            a -= S;
        }
}

// The ascending merge of a padded LDS array whose two halves are each sorted after a mirror step further up: distances 2^(lc-1) .. 1.
template <int NT>
__device__ __forceinline__ void merge_padded_lds(uint64_t *s, int lc, int tid)
{
    const int npad = 1 << lc;
    for (int a = lc - 1; a >= 0;) {
        const int S = ts_chunk(a);
        if (S == 4) { for (int g = tid; g < (npad >> 4); g += NT) bitonic_group<4>(s, g, lc, a, lc); }
        else if (S == 3) { for (int g = tid; g < (npad >> 3); g += NT) bitonic_group<3>(s, g, lc, a, lc); }
        else if (S == 2) { for (int g = tid; g < (npad >> 2); g += NT) bitonic_group<2>(s, g, lc, a, lc); }
        else { for (int g = tid; g < (npad >> 1); g += NT) bitonic_group<1>(s, g, lc, a, lc); }
        __syncthreads();
        a -= S;
    }
}

constexpr int lds_slots(int lp_cap) { return (1 << lp_cap) + (1 << (lp_cap - 5)); }

// Output of a sorted segment: point_list = the low words (surfel / ray ids), keys_sorted (optional) = tile id << 32 | high word -- or, with
// full64, keys_sorted = the 64-bit values themselves (which may be the very buffer the segment was read from: every workgroup has its
// whole segment in LDS before it stores).
__device__ __forceinline__ void store_sorted(uint64_t v, uint32_t tile, size_t at, uint64_t *keys_sorted, uint32_t *point_list, int full64)
{
    if (full64) { keys_sorted[at] = v; return; }
    point_list[at] = (uint32_t)v;
    if (keys_sorted) keys_sorted[at] = ((uint64_t)tile << 32) | (v >> 32);
}

template <int NT>
__device__ __forceinline__ void load_sort_write(uint64_t *s, const uint64_t *pairs, int n, uint32_t tile, uint32_t b,
                                                uint64_t *keys_sorted, uint32_t *point_list, int tid, int full64)
{
    const int lp = ts_log2_padded(n);
    for (int i = tid; i < (1 << lp); i += NT) s[ts_slot(i)] = i < n ? pairs[b + i] : ~0ull;       // (a real pair is below 2^63: view depths are positive floats)
    __syncthreads();
    sort_padded_lds<NT>(s, lp, tid);
    for (int i = tid; i < n; i += NT) store_sorted(s[ts_slot(i)], tile, (size_t)b + i, keys_sorted, point_list, full64);
}

// One workgroup per tile; lists longer than the LDS array of this instantiation are handed to sort_long_lists.
template <int LP_CAP, int NT>
__global__ void __launch_bounds__(NT)
sort_tile_lists(const uint32_t *__restrict__ ranges, const uint64_t *pairs, uint64_t *keys_sorted, uint32_t *__restrict__ point_list,
                uint32_t *__restrict__ hdr, uint32_t *__restrict__ long_list, int full64)
{
    __shared__ uint64_t s_k[lds_slots(LP_CAP)];
    const uint32_t t = blockIdx.x, b = ranges[2 * t];
    const int n = (int)(ranges[2 * t + 1] - b);
    if (n == 0) return;
    if (n > (1 << LP_CAP)) {
        if (threadIdx.x == 0) long_list[atomicAdd(&hdr[1], 1u)] = t;
        return;
    }
    load_sort_write<NT>(s_k, pairs, n, t, b, keys_sorted, point_list, (int)threadIdx.x, full64);
}

// The long lists, one persistent workgroup per CU.  Up to 16 384 entries: one workgroup sorts the list in LDS (the lists are dealt round robin).
// Beyond that (a few far outlier surfels collapse the Morton codes, a narrow cone of rays collapses the direction cells, a tile with
// > 16 k instances -- and ONE workgroup merging a list through HBM is slow: ~20 ms for 600 k entries, 76 ms for the 120 k equal Morton codes of
// round 4's single-workgroup shortcut): the whole grid works on ALL such lists AT ONCE.  The sort of a list is a fixed sequence of PHASES
// (tile_sort.h: ts_sort_hybrid is the schedule) -- sort every 16 384-entry chunk in LDS; then per merge stage the steps wider than a chunk on
// the segment in HBM, and an ascending merge of every chunk in LDS again -- and every phase is a number of independent ITEMS (a chunk; 4 096
// comparators of a wide step).  The schedule is walked PHASE-MAJOR over a batch of up to 256 lists: phase p of every list of the batch that has
// a phase p (a list of 2^lp entries has none for stages beyond lp) is one pool of items, so thirty ray buckets of 17-40 k entries (a bounce
// stage of the 1200x1600 configuration) cost the ~8 grid-wide phases of ONE such list, not thirty times that.  Items are numbered through all
// phases and CLAIMED from one counter (hdr[2]); an item of phase p starts when the completion counter (hdr[3]) says every item of the earlier
// phases is done.  No workgroup ever waits for a workgroup that has not started: whatever is unfinished was claimed by a RUNNING workgroup, so
// the scheme cannot deadlock however many of the grid's workgroups are resident (two such kernels on two streams, several processes on one
// GPU), and a late workgroup finds its counters exhausted and falls through.  Lists that need no cooperation touch neither counter.
constexpr int LONG_ITEM = 4096;       // comparators of a wide step per item
constexpr int LONG_BATCH = 256;       // lists per phase-major batch

__device__ __forceinline__ uint32_t long_claim(uint32_t *hdr, uint32_t *s_item)
{
    __syncthreads();
    if (threadIdx.x == 0) *s_item = atomicAdd(&hdr[2], 1u);
    __syncthreads();
    return *s_item;
}
__device__ __forceinline__ void long_wait(uint32_t *hdr, const uint32_t base)
{
    if (threadIdx.x == 0)
        while (__hip_atomic_load(&hdr[3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < base) __builtin_amdgcn_s_sleep(4);
    __syncthreads();
    __threadfence();                    // (every lane: the segment was written by other workgroups, possibly behind another XCD's L2)
}
__device__ __forceinline__ void long_done(uint32_t *hdr)
{
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&hdr[3], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(1024)
sort_long_lists(const uint32_t *__restrict__ ranges, uint64_t *pairs, uint64_t *keys_sorted, uint32_t *__restrict__ point_list,
                uint32_t *hdr, const uint32_t *__restrict__ long_list, int full64)
{
    __shared__ uint64_t s_long[lds_slots(14)];
    __shared__ uint32_t s_item;
    __shared__ uint32_t s_bt[LONG_BATCH], s_bb[LONG_BATCH], s_bn[LONG_BATCH], s_pref[LONG_BATCH + 1];
    const uint32_t count = hdr[1];
    const int tid = (int)threadIdx.x, G = (int)gridDim.x;
    constexpr int LC = 14, C = 1 << LC;
    const bool copy_out = !(full64 && keys_sorted == pairs);
    // ---- the lists of one chunk: round robin, each sorted in LDS by one workgroup; every workgroup also counts the longer ones
    uint32_t small_seen = 0u, big = 0u;
    for (uint32_t w = 0; w < count; w++) {
        const uint32_t t = long_list[w], b = ranges[2 * t];
        const int n = (int)(ranges[2 * t + 1] - b);
        if (n > SORT_LONG_N) { big++; continue; }
        if ((small_seen++ % (uint32_t)G) != blockIdx.x) continue;
        load_sort_write<1024>(s_long, pairs, n, t, b, keys_sorted, point_list, tid, full64);
        __syncthreads();
    }
    if (big == 0u) return;
    // ---- the longer lists, phase-major in batches (every workgroup builds the same tables: the order of long_list is fixed by now)
    auto lp_of = [&](const int n) { int lp = LC; while ((1 << lp) < n) lp++; return lp; };
    // phase kinds: 0 = chunk sort / merge (lk), 1 = wide step (lk, q), 2 = copy the sorted segment out
    auto items_of = [&](const int kind, const int lk, const int n) -> uint32_t {
        const int lp = lp_of(n);
        if (kind == 1) return lk <= lp ? (uint32_t)(((1ll << (lp - 1)) + LONG_ITEM - 1) / LONG_ITEM) : 0u;
        return (kind == 2 || lk <= lp) ? (uint32_t)((n + C - 1) / C) : 0u;
    };
    auto do_item = [&](const int kind, const int lk, const int q, const int l, const uint32_t it) {
        const uint32_t t = s_bt[l], b = s_bb[l];
        const int n = (int)s_bn[l];
        uint64_t *seg = pairs + b;           // all-ascending network: entries beyond n never move, so nothing is padded in memory
        if (kind == 0) {
            const int c = (int)it;
            for (int i = tid; i < C; i += 1024) s_long[ts_slot(i)] = ((long long)c * C + i) < n ? seg[(size_t)c * C + i] : ~0ull;
            __syncthreads();
            if (lk == LC) sort_padded_lds<1024>(s_long, LC, tid);
            else merge_padded_lds<1024>(s_long, LC, tid);
            for (int i = tid; i < C; i += 1024)
                if ((long long)c * C + i < n) seg[(size_t)c * C + i] = s_long[ts_slot(i)];
        } else if (kind == 1) {
            const int lp = lp_of(n);
            const long long i0 = (long long)it * LONG_ITEM, i1 = min(i0 + LONG_ITEM, 1ll << (lp - 1));
            for (long long idx = i0 + tid; idx < i1; idx += 1024) ascending_step(seg, n, lk, q, (int)idx);
        } else {
            const long long i0 = (long long)it * C, i1 = min(i0 + C, (long long)n);
            for (long long i = i0 + tid; i < i1; i += 1024) store_sorted(seg[i], t, (size_t)b + (size_t)i, keys_sorted, point_list, full64);
        }
    };
    uint32_t base = 0u, my = long_claim(hdr, &s_item);
    uint32_t w = 0;
    while (w < count) {
        __syncthreads();                                 // (the previous batch's tables are no longer read)
        int nl = 0, lp_max = LC;
        for (; w < count && nl < LONG_BATCH; w++) {      // uniform over the workgroup: every lane walks the list, lane 0 fills the table
            const uint32_t t = long_list[w], b = ranges[2 * t];
            const int n = (int)(ranges[2 * t + 1] - b);
            if (n <= SORT_LONG_N) continue;
            if (tid == 0) { s_bt[nl] = t; s_bb[nl] = b; s_bn[nl] = (uint32_t)n; }
            lp_max = max(lp_max, lp_of(n));
            nl++;
        }
        if (nl == 0) break;
        auto run_phase = [&](const int kind, const int lk, const int q) {
            __syncthreads();                             // (s_pref of the previous phase is no longer read; the batch table is written)
            if (tid < nl) {
                uint32_t acc = 0u;
                for (int l = 0; l < tid; l++) acc += items_of(kind, lk, (int)s_bn[l]);
                s_pref[tid] = acc;
                if (tid == nl - 1) s_pref[nl] = acc + items_of(kind, lk, (int)s_bn[tid]);
            }
            __syncthreads();
            const uint32_t items = s_pref[nl];
            while (my - base < items) {                  // (unsigned: my >= base always -- items are claimed in order)
                const uint32_t j = my - base;
                int lo = 0, hi = nl - 1;                 // the list whose items [s_pref[l], s_pref[l + 1]) hold j
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_pref[mid] <= j) lo = mid; else hi = mid - 1; }
                long_wait(hdr, base);
                do_item(kind, lk, q, lo, j - s_pref[lo]);
                long_done(hdr);
                my = long_claim(hdr, &s_item);
            }
            base += items;
        };
        run_phase(0, LC, 0);
        for (int lk = LC + 1; lk <= lp_max; lk++) {
            for (int q = 0; q <= lk - LC - 1; q++) run_phase(1, lk, q);
            run_phase(0, lk, 0);
        }
        if (copy_out) run_phase(2, 0, 0);
    }
}

// sort_long_lists' grid: one workgroup per CU of the device the stream runs on
static int long_sort_grid()
{
    static int cached[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 64;
    if (cached[dev] == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 64;
        cached[dev] = cus < SORT_LONG_WGS ? cus : SORT_LONG_WGS;
    }
    return cached[dev];
}

int launch_bin(const envgs_raster_cfg *cfg, uint32_t N, const float *geom, const int32_t *radii, uint64_t *tile_pairs,
               uint64_t *keys_sorted, uint32_t *point_list, void *bin_temp, size_t bin_temp_bytes, uint32_t *ranges, hipStream_t stream)
{
    // N is the CAPACITY of the N-sized buffers: the exact instance count when the caller waited for it, or a guess made before the count was
    // known on the host (no host sync between projection and binning).  The count itself is re-derived here (sum of the tile totals); a
    // capacity below it leaves every range empty and writes nothing out of bounds.
    const BinPlan pl = bin_plan(cfg->P, cfg->width, cfg->height);
    if (N == 0 || cfg->P <= 0) return (int)hipMemsetAsync(ranges, 0, sizeof(uint32_t) * 2 * (size_t)pl.ntiles, stream);
    if (bin_temp_bytes < sort_temp_bytes(N, cfg->width, cfg->height)) return ENVGS_ERR_TEMP_TOO_SMALL;
    uint32_t *hist = (uint32_t *)bin_temp;
    uint32_t *tile_count = hist + (size_t)BIN_ROWS_MAX * pl.ntiles;
    uint32_t *tile_start = tile_count + pl.ntiles;
    uint32_t *long_list = tile_start + pl.ntiles + 1;
    uint32_t *hdr = long_list + pl.ntiles;
    const size_t lds = sizeof(uint32_t) * (size_t)(pl.ntiles < BIN_BAND ? pl.ntiles : BIN_BAND);
    prof_begin(K_EMIT_KEYS, stream);
    hipLaunchKernelGGL(bin_pass<false>, dim3(pl.rows), dim3(256), lds, stream, cfg->P, cfg->width, cfg->height, pl.slice, pl.ntiles,
                       geom, radii, hist, (const uint32_t *)nullptr, (uint64_t *)nullptr, 0u);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    hipLaunchKernelGGL(bin_column_scan, dim3((pl.ntiles + 15) / 16), dim3(256), 0, stream, pl.rows, pl.ntiles, hist, tile_count);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    hipLaunchKernelGGL(bin_tile_scan, dim3(1), dim3(1024), 0, stream, pl.ntiles, tile_count, tile_start, ranges, N, hdr);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    hipLaunchKernelGGL(bin_pass<true>, dim3(pl.rows), dim3(256), lds, stream, cfg->P, cfg->width, cfg->height, pl.slice, pl.ntiles,
                       geom, radii, hist, tile_start, tile_pairs, N);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    prof_end(K_EMIT_KEYS, stream);
    ProfScope prof_(K_SORT, stream);
    // LDS array per tile: room for 1.6x the average list length the capacity N allows, 16.5 / 33 / 66 KB (the smaller the array, the more
    // tiles are in flight per CU: 46 us vs 53 us for the 300 k / 800 x 800 lists); what does not fit is a long list
    const uint64_t want = (uint64_t)N * 8 / ((uint64_t)pl.ntiles * 5);
    if (want <= 2048)
        hipLaunchKernelGGL((sort_tile_lists<11, 128>), dim3(pl.ntiles), dim3(128), 0, stream, ranges, tile_pairs, keys_sorted, point_list, hdr, long_list, 0);
    else if (want <= 4096)
        hipLaunchKernelGGL((sort_tile_lists<12, 256>), dim3(pl.ntiles), dim3(256), 0, stream, ranges, tile_pairs, keys_sorted, point_list, hdr, long_list, 0);
    else
        hipLaunchKernelGGL((sort_tile_lists<13, 512>), dim3(pl.ntiles), dim3(512), 0, stream, ranges, tile_pairs, keys_sorted, point_list, hdr, long_list, 0);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    hipLaunchKernelGGL(sort_long_lists, dim3(long_sort_grid()), dim3(1024), 0, stream, ranges, tile_pairs,
                       keys_sorted, point_list, hdr, long_list, 0);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    return 0;
}

// ---- the tracer's ray coherence sort on the same machinery ---------------------------------------------------------------------------------
// R (key, ray id) pairs in (key, id) order -- what a stable radix sort of the 31-bit keys gives.  The top bits of the key are the bucket
// (the "tile"): LDS histograms per slice of rays, column / bucket scans, scatter into bucket segments, one LDS sort per bucket; the sorted
// low words are the ray order.  Five short launches instead of the ~22 dependent merge passes a library pair sort of 640 k items ran as
// (0.145 ms in front of the collection, on the critical path of the step).
// The per-bucket sorts of n items in nb buckets (LDS array picked from the average bucket, then the long lists).
static void launch_bucket_sorts(uint64_t n, int nb, const uint32_t *ranges, uint64_t *pairs, uint64_t *keys_sorted, uint32_t *point_list,
                                uint32_t *hdr, uint32_t *long_list, int full64, hipStream_t stream)
{
    const uint64_t want = n * 8 / ((uint64_t)nb * 5);
    if (want <= 2048)
        hipLaunchKernelGGL((sort_tile_lists<11, 128>), dim3(nb), dim3(128), 0, stream, ranges, pairs, keys_sorted, point_list, hdr, long_list, full64);
    else if (want <= 4096)
        hipLaunchKernelGGL((sort_tile_lists<12, 256>), dim3(nb), dim3(256), 0, stream, ranges, pairs, keys_sorted, point_list, hdr, long_list, full64);
    else
        hipLaunchKernelGGL((sort_tile_lists<13, 512>), dim3(nb), dim3(512), 0, stream, ranges, pairs, keys_sorted, point_list, hdr, long_list, full64);
    hipLaunchKernelGGL(sort_long_lists, dim3(long_sort_grid()), dim3(1024), 0, stream, ranges, pairs, keys_sorted, point_list, hdr, long_list, full64);
}

static int ray_bucket_bits(int R)
{
    int b = 4;
    while (b < 13 && ((long long)160 << b) < R) b++;
    return b;
}

size_t ray_sort_temp_bytes(int R)
{
    const size_t nb = (size_t)1 << ray_bucket_bits(R);
    // hist (BIN_ROWS_MAX x nb) | count (nb) | start (nb + 1) | long_list (nb) | hdr (4) | ranges (2 nb) | keys (R) | origin-bounds partials (BIN_ROWS_MAX x 6)
    return sizeof(uint32_t) * ((size_t)BIN_ROWS_MAX * nb + 5 * nb + 1 + 4 + (size_t)(R > 0 ? R : 1) + 6 * (size_t)BIN_ROWS_MAX) + 256;
}

// Bounding box of the ray origins, one partial (lo[3], hi[3]) per workgroup: the key pass reduces the <= BIN_ROWS_MAX partials itself (no atomics,
// nothing to initialise).
__global__ void __launch_bounds__(256)
ray_origin_bounds(int R, int slice, const float *__restrict__ ray_o, float *__restrict__ partial)
{
    __shared__ float s_red[4][6];
    const int g0 = blockIdx.x * slice, g1 = min(R, g0 + slice);
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = g0 + (int)threadIdx.x; i < g1; i += 256) {
#pragma unroll
        for (int c = 0; c < 3; c++) { const float x = ray_o[3 * i + c]; if (x == x && fabsf(x) < 1.0e30f) { lo[c] = fminf(lo[c], x); hi[c] = fmaxf(hi[c], x); } }
    }
#pragma unroll
    for (int c = 0; c < 3; c++)
        for (int o = 32; o > 0; o >>= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], o)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o)); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { for (int c = 0; c < 3; c++) { s_red[wave][c] = lo[c]; s_red[wave][3 + c] = hi[c]; } }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int c = threadIdx.x;
        float v = s_red[0][c];
        for (int w = 1; w < 4; w++) v = c < 3 ? fminf(v, s_red[w][c]) : fmaxf(v, s_red[w][c]);
        partial[blockIdx.x * 6 + c] = v;
    }
}

template <bool SCATTER>
__global__ void __launch_bounds__(256)
ray_bucket_pass(int R, int slice, int nb, int shift, const float *__restrict__ ray_o, const float *__restrict__ ray_d,
                const float *__restrict__ bounds_partial, int nrows, int lead, uint32_t *__restrict__ keys, uint32_t *hist, const uint32_t *__restrict__ bucket_start,
                uint64_t *__restrict__ pairs)
{
    extern __shared__ uint32_t s_bin[];
    __shared__ float s_box[6];
    const int g0 = blockIdx.x * slice, g1 = min(R, g0 + slice);
    uint32_t *row = hist + (size_t)blockIdx.x * nb;
    for (int t = threadIdx.x; t < nb; t += 256) s_bin[t] = SCATTER ? bucket_start[t] + row[t] : 0u;
    if (!SCATTER && threadIdx.x < 64) {          // the origins' bounding box: min / max over the partials (6 lanes x strided rows, then across the wavefront)
        const int c = threadIdx.x % 6, j0 = threadIdx.x / 6;
        float v = c < 3 ? 3.0e38f : -3.0e38f;
        if (j0 < 10)
            for (int j = j0; j < nrows; j += 10) { const float x = bounds_partial[j * 6 + c]; v = c < 3 ? fminf(v, x) : fmaxf(v, x); }
        // lanes c, c + 6, ..., c + 54 hold component c: gather them in lane c
        float acc = v;
        for (int k = 1; k < 10; k++) { const float x = __shfl(v, (int)(threadIdx.x % 6) + 6 * k); if (threadIdx.x < 6) acc = c < 3 ? fminf(acc, x) : fmaxf(acc, x); }
        if (threadIdx.x < 6) s_box[c] = acc;
    }
    __syncthreads();
    float lo0 = 0.f, lo1 = 0.f, lo2 = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if (!SCATTER) {
        lo0 = s_box[0]; lo1 = s_box[1]; lo2 = s_box[2];
        const float e0 = s_box[3] - lo0, e1 = s_box[4] - lo1, e2 = s_box[5] - lo2;
        s0 = e0 > 0.f ? 32.0f / e0 : 0.f; s1 = e1 > 0.f ? 32.0f / e1 : 0.f; s2 = e2 > 0.f ? 32.0f / e2 : 0.f;
    }
    for (int i = g0 + (int)threadIdx.x; i < g1; i += 256) {
        uint32_t key;
        if (SCATTER) key = keys[i];
        else keys[i] = key = ray_coherence_key(i, ray_o, ray_d, lo0, lo1, lo2, s0, s1, s2, lead);
        const uint32_t slot = atomicAdd(&s_bin[key >> shift], 1u);
        if (SCATTER) pairs[slot] = ((uint64_t)key << 32) | (uint32_t)i;
    }
    __syncthreads();
    if (!SCATTER)
        for (int t = threadIdx.x; t < nb; t += 256) row[t] = s_bin[t];
}

int launch_ray_sort(int R, const float *ray_o, const float *ray_d, const float4 *nodes, int P, uint64_t *pairs, uint32_t *order,
                    void *temp, size_t temp_bytes, hipStream_t stream)
{
    (void)nodes; (void)P;           // (rounds 1-3 quantised the origins inside the scene box; the key now uses the rays' own bounding box)
    if (R <= 0) return 0;
    if (temp_bytes < ray_sort_temp_bytes(R)) return ENVGS_ERR_TEMP_TOO_SMALL;
    const int bits = ray_bucket_bits(R), nb = 1 << bits, shift = 31 - bits;
    int rows = (R + BIN_SLICE_MIN - 1) / BIN_SLICE_MIN;
    rows = rows < 1 ? 1 : (rows > BIN_ROWS_MAX ? BIN_ROWS_MAX : rows);
    const int slice = (((R + rows - 1) / rows + 255) / 256) * 256;
    rows = (R + slice - 1) / slice;
    uint32_t *hist = (uint32_t *)temp;
    uint32_t *count = hist + (size_t)BIN_ROWS_MAX * nb;
    uint32_t *start = count + nb;
    uint32_t *long_list = start + nb + 1;
    uint32_t *hdr = long_list + nb;
    uint32_t *ranges = hdr + 4;
    uint32_t *keys = ranges + 2 * (size_t)nb;
    float *partial = (float *)(keys + (size_t)R);                          // (BIN_ROWS_MAX, 6) bounds partials
    const size_t lds = sizeof(uint32_t) * (size_t)nb;
    int lead = debug_switch(ENVGS_DBG_RAYKEY) > 0 ? debug_switch(ENVGS_DBG_RAYKEY) - 1 : RAY_KEY_LEAD;
    lead = lead > 8 ? 8 : lead;
    hipLaunchKernelGGL(ray_origin_bounds, dim3(rows), dim3(256), 0, stream, R, slice, ray_o, partial);
    hipLaunchKernelGGL(ray_bucket_pass<false>, dim3(rows), dim3(256), lds, stream, R, slice, nb, shift, ray_o, ray_d, (const float *)partial, rows, lead, keys, hist,
                       (const uint32_t *)nullptr, (uint64_t *)nullptr);
    hipLaunchKernelGGL(bin_column_scan, dim3((nb + 15) / 16), dim3(256), 0, stream, rows, nb, hist, count);
    hipLaunchKernelGGL(bin_tile_scan, dim3(1), dim3(1024), 0, stream, nb, count, start, ranges, (uint32_t)R, hdr);
    hipLaunchKernelGGL(ray_bucket_pass<true>, dim3(rows), dim3(256), lds, stream, R, slice, nb, shift, ray_o, ray_d, (const float *)partial, rows, lead, keys, hist, start, pairs);
    launch_bucket_sorts(R, nb, ranges, pairs, (uint64_t *)nullptr, order, hdr, long_list, 0, stream);
    return (int)hipGetLastError();
}

// ---- n 64-bit keys in ascending order (the LBVH's Morton code << 32 | surfel id): buckets = the top bits below `top_bit` -----------------
size_t key_sort_temp_bytes(int n)
{
    const size_t nb = (size_t)1 << ray_bucket_bits(n);
    return sizeof(uint32_t) * ((size_t)BIN_ROWS_MAX * nb + 5 * nb + 1 + 4) + 256;
}

template <bool SCATTER>
__global__ void __launch_bounds__(256)
key_bucket_pass(int n, int slice, int nb, int shift, const uint64_t *__restrict__ keys, uint32_t *hist, const uint32_t *__restrict__ bucket_start,
                uint64_t *__restrict__ out)
{
    extern __shared__ uint32_t s_bin[];
    const int g0 = blockIdx.x * slice, g1 = min(n, g0 + slice);
    uint32_t *row = hist + (size_t)blockIdx.x * nb;
    for (int t = threadIdx.x; t < nb; t += 256) s_bin[t] = SCATTER ? bucket_start[t] + row[t] : 0u;
    __syncthreads();
    for (int i = g0 + (int)threadIdx.x; i < g1; i += 256) {
        const uint64_t k = keys[i];
        const uint32_t slot = atomicAdd(&s_bin[(uint32_t)(k >> shift) & (uint32_t)(nb - 1)], 1u);
        if (SCATTER) out[slot] = k;
    }
    __syncthreads();
    if (!SCATTER)
        for (int t = threadIdx.x; t < nb; t += 256) row[t] = s_bin[t];
}

// keys_in (n) -> keys_out (n) ascending; only the bits below top_bit take part in the bucket choice (bits at and above it must be zero).
int launch_key_sort(int n, const uint64_t *keys_in, uint64_t *keys_out, int top_bit, void *temp, size_t temp_bytes, hipStream_t stream)
{
    if (n <= 0) return 0;
    if (temp_bytes < key_sort_temp_bytes(n)) return ENVGS_ERR_TEMP_TOO_SMALL;
    const int bits = ray_bucket_bits(n), nb = 1 << bits, shift = top_bit - bits;
    int rows = (n + BIN_SLICE_MIN - 1) / BIN_SLICE_MIN;
    rows = rows < 1 ? 1 : (rows > BIN_ROWS_MAX ? BIN_ROWS_MAX : rows);
    const int slice = (((n + rows - 1) / rows + 255) / 256) * 256;
    rows = (n + slice - 1) / slice;
    uint32_t *hist = (uint32_t *)temp;
    uint32_t *count = hist + (size_t)BIN_ROWS_MAX * nb;
    uint32_t *start = count + nb;
    uint32_t *long_list = start + nb + 1;
    uint32_t *hdr = long_list + nb;
    uint32_t *ranges = hdr + 4;
    const size_t lds = sizeof(uint32_t) * (size_t)nb;
    hipLaunchKernelGGL(key_bucket_pass<false>, dim3(rows), dim3(256), lds, stream, n, slice, nb, shift, keys_in, hist, (const uint32_t *)nullptr, (uint64_t *)nullptr);
    hipLaunchKernelGGL(bin_column_scan, dim3((nb + 15) / 16), dim3(256), 0, stream, rows, nb, hist, count);
    hipLaunchKernelGGL(bin_tile_scan, dim3(1), dim3(1024), 0, stream, nb, count, start, ranges, (uint32_t)n, hdr);
    hipLaunchKernelGGL(key_bucket_pass<true>, dim3(rows), dim3(256), lds, stream, n, slice, nb, shift, keys_in, hist, start, keys_out);
    launch_bucket_sorts((uint64_t)n, nb, ranges, keys_out, keys_out, (uint32_t *)nullptr, hdr, long_list, 1, stream);      // in place
    return (int)hipGetLastError();
}

}  // namespace envgs
