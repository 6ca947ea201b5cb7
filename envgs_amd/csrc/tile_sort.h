// tile_sort.h -- the sorting networks of raster_bin.hip's per-tile sort, as plain index arithmetic (host + device), so that
// tests/test_tile_sort_network.py can run the same functions on the CPU over every list length.
//
// Two networks over 64-bit keys:
//  * bitonic_group<S>: the classic bitonic network on a power-of-two array, S (up to 4) consecutive compare-exchange steps of one merge
//    stage done in REGISTERS on the 2^S elements they connect (one LDS read + one LDS write per element per S steps instead of per step,
//    and one workgroup barrier per S steps).  Stage k = 2^lk merges runs of k/2; step 2^a compares elements whose indices differ in bit a; the
//    run direction is bit lk of the index (the last stage, lk = log2(length), is ascending everywhere).
//  * ascending_step: the same sorter with every comparator ascending (each merge stage starts by comparing element i of the lower run
//    with its MIRROR in the upper run), so entries at or beyond n never take part: no padding, any n.  Used on segments too long for LDS,
//    as a hybrid: chunks of 2^lc entries are sorted in LDS; of every later stage only the steps whose distance is at least a chunk run
//    on the segment in HBM, the rest of the stage is an ascending merge of each chunk in LDS again (ts_sort_hybrid below is the schedule).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define TS_FN __host__ __device__ __forceinline__
#else
#define TS_FN inline
#endif

namespace envgs {

TS_FN void ts_cex(uint64_t &a, uint64_t &b, bool asc)
{
    const bool sw = (a > b) == asc;
    const uint64_t lo = sw ? b : a, hi = sw ? a : b;
    a = lo; b = hi;
}

// LDS layout of the padded array: one spare slot after every 32 entries, so that the strided register loads of the low-distance steps
// (lane stride 2, 4, 8, 16 entries) spread over the 64 banks instead of hitting the same few.
TS_FN int ts_slot(int i) { return i + (i >> 5); }
constexpr int TS_S = 4;                           // compare-exchange steps done in registers per LDS round trip (2^TS_S entries per lane)

// Stages 1..TS_S on the 16 contiguous entries [16g, 16g+16): they end up sorted, ascending when bit TS_S of the index is clear.
TS_FN void bitonic_first(uint64_t *s, int g, int lp)
{
    constexpr int G = 1 << TS_S;
    uint64_t v[G];
    const int base = g << TS_S;
#pragma unroll
    for (int m = 0; m < G; m++) v[m] = s[ts_slot(base + m)];
#pragma unroll
    for (int lk = 1; lk <= TS_S; lk++) {
#pragma unroll
        for (int st = lk - 1; st >= 0; st--) {
#pragma unroll
            for (int m = 0; m < G; m++) {
                if (m & (1 << st)) continue;
                const bool asc = lk >= lp ? true : ((((base + m) >> lk) & 1) == 0);
                ts_cex(v[m], v[m | (1 << st)], asc);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < G; m++) s[ts_slot(base + m)] = v[m];
}

// Steps 2^a, 2^(a-1), .. 2^(a-S+1) of stage lk on group g of 2^S entries (stride 2^(a-S+1)); groups are disjoint, g in [0, length >> S).
template <int S>
TS_FN void bitonic_group(uint64_t *s, int g, int lk, int a, int lp)
{
    constexpr int G = 1 << S;
    const int sh = a - S + 1;
    const int base = ((g >> sh) << (a + 1)) | (g & ((1 << sh) - 1));
    const bool asc = lk >= lp ? true : (((base >> lk) & 1) == 0);
    uint64_t v[G];
#pragma unroll
    for (int m = 0; m < G; m++) v[m] = s[ts_slot(base + (m << sh))];
#pragma unroll
    for (int st = S - 1; st >= 0; st--) {
#pragma unroll
        for (int m = 0; m < G; m++) {
            if (m & (1 << st)) continue;
            ts_cex(v[m], v[m | (1 << st)], asc);
        }
    }
#pragma unroll
    for (int m = 0; m < G; m++) s[ts_slot(base + (m << sh))] = v[m];
}

// The schedule: bitonic_first on every group of 16, then for lk = TS_S+1 .. lp: a = lk - 1; while a >= 0: S = min(TS_S, a + 1), all groups,
// barrier, a -= S.  Lengths are padded to at least 2^TS_S.
TS_FN int ts_chunk(int a) { return a + 1 < TS_S ? a + 1 : TS_S; }
TS_FN int ts_log2_padded(int n) { int lp = TS_S; while ((1 << lp) < n) lp++; return lp; }

// One step of the all-ascending network on s[0..n): stage lk, step index q (q == 0: the mirror step; q >= 1: distance 2^(lk-1-q)),
// comparator idx in [0, half) with half = 2^(lp-1).
TS_FN void ascending_step(uint64_t *s, int n, int lk, int q, int idx)
{
    int i, p;
    if (q == 0) {
        const int hk = 1 << (lk - 1);
        const int base = (idx >> (lk - 1)) << lk, off = idx & (hk - 1);
        i = base + off; p = base + 2 * hk - 1 - off;
    } else {
        const int j = 1 << (lk - 1 - q);
        i = ((idx & ~(j - 1)) << 1) | (idx & (j - 1)); p = i + j;
    }
    if (p < n) {
        const uint64_t a = s[i], b = s[p];
        if (a > b) { s[i] = b; s[p] = a; }
    }
}

}  // namespace envgs

#if !defined(__HIPCC__)
// C entry points for the CPU test (compiled by tests/test_tile_sort_network.py with g++ into a scratch .so; not part of the product library)
extern "C" {
// sort s[0..n) through the padded register-blocked network exactly as sort_tile_lists schedules it; pad = scratch of ts_slot(2^lp) entries
void ts_sort_blocked(uint64_t *s, int n, uint64_t *pad)
{
    using namespace envgs;
    const int lp = ts_log2_padded(n), npad = 1 << lp;
    for (int i = 0; i < npad; i++) pad[ts_slot(i)] = i < n ? s[i] : ~0ull;
    for (int g = 0; g < (npad >> TS_S); g++) bitonic_first(pad, g, lp);
    for (int lk = TS_S + 1; lk <= lp; lk++)
        for (int a = lk - 1; a >= 0;) {
            const int S = ts_chunk(a);
            for (int g = 0; g < (npad >> S); g++) {
                if (S == 4) bitonic_group<4>(pad, g, lk, a, lp);
                else if (S == 3) bitonic_group<3>(pad, g, lk, a, lp);
                else if (S == 2) bitonic_group<2>(pad, g, lk, a, lp);
                else bitonic_group<1>(pad, g, lk, a, lp);
            }
            a -= S;
        }
    for (int i = 0; i < n; i++) s[i] = pad[ts_slot(i)];
}
// the hybrid schedule of sort_long_lists' HBM branch with chunks of 2^lc entries (the kernel uses lc = 14); pad = scratch of ts_slot(2^lc)
void ts_sort_hybrid(uint64_t *s, int n, uint64_t *pad, int lc)
{
    using namespace envgs;
    const int C = 1 << lc;
    int lp = lc;
    while ((1 << lp) < n) lp++;
    auto load = [&](int c) { for (int i = 0; i < C; i++) pad[ts_slot(i)] = (c * C + i) < n ? s[c * C + i] : ~0ull; };
    auto store = [&](int c) { for (int i = 0; i < C; i++) if (c * C + i < n) s[c * C + i] = pad[ts_slot(i)]; };
    auto lds_steps = [&](bool full) {          // the chunk in LDS: sorted in full, or (later stages) only the ascending merge of its distances below a chunk
        if (full) {
            for (int g = 0; g < (C >> TS_S); g++) bitonic_first(pad, g, lc);
            for (int lk = TS_S + 1; lk <= lc; lk++)
                for (int a = lk - 1; a >= 0;) { const int S = ts_chunk(a);
                    for (int g = 0; g < (C >> S); g++) { if (S == 4) bitonic_group<4>(pad, g, lk, a, lc); else if (S == 3) bitonic_group<3>(pad, g, lk, a, lc); else if (S == 2) bitonic_group<2>(pad, g, lk, a, lc); else bitonic_group<1>(pad, g, lk, a, lc); }
                    a -= S; }
        } else {
            for (int a = lc - 1; a >= 0;) { const int S = ts_chunk(a);
                for (int g = 0; g < (C >> S); g++) { if (S == 4) bitonic_group<4>(pad, g, lc, a, lc); else if (S == 3) bitonic_group<3>(pad, g, lc, a, lc); else if (S == 2) bitonic_group<2>(pad, g, lc, a, lc); else bitonic_group<1>(pad, g, lc, a, lc); }
                a -= S; }
        }
    };
    const int nchunks = (n + C - 1) / C;
    for (int c = 0; c < nchunks; c++) { load(c); lds_steps(true); store(c); }
    for (int lk = lc + 1; lk <= lp; lk++) {
        for (int q = 0; q <= lk - lc - 1; q++)
            for (int idx = 0; idx < (1 << (lp - 1)); idx++) ascending_step(s, n, lk, q, idx);
        for (int c = 0; c < nchunks; c++) { load(c); lds_steps(false); store(c); }
    }
}
void ts_sort_ascending(uint64_t *s, int n)
{
    using namespace envgs;
    int lp = 1;
    while ((1 << lp) < n) lp++;
    const int half = 1 << (lp - 1);
    for (int lk = 1; lk <= lp; lk++)
        for (int q = 0; q < lk; q++)
            for (int idx = 0; idx < half; idx++) ascending_step(s, n, lk, q, idx);
}
}
#endif
