/*
 * envgs_trace.h -- C-ABI of the MI355X-native surfel ray tracer (libenvgs_hip.so).
 *
 * Drop-in boundary for the reference's `diff_surfel_tracing` extension (CUDA + OptiX 7), whose Python call
 * sites are easyvolcap/utils/optix_utils.py:24 (SurfelTracer()), :78 (build_acceleration_structure),
 * :104-119 (SurfelTracingSettings) and :188-201 (the traced call).  OptiX's GAS + any-hit pipeline is replaced
 * by a hand-written LBVH: Morton codes of the surfel proxies -> radix sort -> Karras hierarchy -> bottom-up AABB
 * fit, and a persistent-wavefront traversal with the per-lane node stack in LDS and a K-nearest hit buffer in
 * registers (front-to-back compositing in rounds of K hits).
 *
 * Plain pointers and sizes; every pointer is a DEVICE pointer unless it ends in _host; `stream` is a hipStream_t
 * passed as void*.  Returns 0, a negative envgs_status, or a positive hipError_t.  The caller owns all memory.
 *
 * HBM layouts (fp32 unless noted):
 *   vertices  (4P,3)   the quad vertices of optix_utils.py:39-69 (get_disks); 4 consecutive vertices = one surfel
 *   nodes     (max(P-1,1),16)  LBVH internal nodes, 64 B each:
 *                      [0..5] left child AABB (min xyz, max xyz)  [6..11] right child AABB
 *                      [12] left child  [13] right child  (int32 bits; >= 0 internal node, < 0 leaf = ~surfel id)
 *                      [14] parent (int32 bits)  [15] unused
 *   srec      (P,16)   per-surfel trace record, rebuilt every forward from the current parameters:
 *                      [0..2] centre  [3] opacity  [4..6] tangent a / s_u  [7] s_u  [8..10] tangent b / s_v  [11] s_v
 *                      [12..14] normal  [15] unused
 *   geo_rec   (P,16)   backward accumulator, one 64 B record per surfel: [0..2] dL/dcentre  [3..5] dL/da  [6..8] dL/db
 *                      [9..11] dL/dn (rotation columns, chained to the quaternion)  [12..13] dL/dscale  [14] dL/dopacity
 *   outputs   rgb (R,3) dpt (R) acc (R) norm (R,3) dist (R) aux (R,2) mid (R,16*(max_trace_depth+1)) wet (P)
 *             mid layout per bounce: rayo 0:3, rayd 3:6, dpt 6, acc 7, norm 8:11, aux 11:13, rgb 13:16
 *             (optix_utils.py:30-37)
 */
#ifndef ENVGS_TRACE_H
#define ENVGS_TRACE_H

#include <stddef.h>
#include <stdint.h>

#include "envgs_raster.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ENVGS_NODE_STRIDE 16
#define ENVGS_SREC_STRIDE 16
#define ENVGS_GEOREC_STRIDE 16
#define ENVGS_MID_CHANNELS 16
#define ENVGS_WIDE_EMPTY ((int32_t)0x80000000)   /* `ref` of an unused slot of a 4-wide node (a child that is a leaf has no grandchildren) */

/* Mirrors SurfelTracingSettings (optix_utils.py:104-119) minus the tensors, plus the call's start_from_first. */
typedef struct envgs_trace_cfg {
    int32_t P;                 /* surfels */
    int32_t num_rays;          /* R */
    int32_t sh_degree;         /* active degree 0..3 */
    int32_t sh_coeffs;         /* coefficients stored per surfel (shs is (P, sh_coeffs, 3)); 0 => colors_precomp (P,3) */
    int32_t max_trace_depth;   /* specular bounces after the primary stage */
    int32_t start_from_first;  /* 1: rays are camera rays (t_min = 0.2); 0: rays are already-reflected rays (t > 0);
                                  2: secondary rays of a bounce traced as a call of their own (t_min = 1e-3) */
    int32_t has_others;        /* others_precomp (P,2) present */
    int32_t bg_len;
    int32_t debug;
    int32_t ray_h, ray_w;      /* if the rays are an (H,W) image (ray_h*ray_w == num_rays) wavefronts take 8x8 pixel blocks; else 0 */
    float scale_modifier;
    float specular_threshold;
    int32_t feature_f16;       /* 1: shs / colors_precomp are stored as IEEE half (same shapes); arithmetic and gradient outputs stay fp32 */
} envgs_trace_cfg;

/*
 * Scratch of the list path (bounce-free tracing; all DEVICE pointers, caller-allocated).  HBM is used deliberately here
 * (288 GB per MI355X): per-ray hit lists make the backward traversal-free, per-(batch, surfel) gradient records grouped by surfel make
 * it atomic-free (the L2 atomic units retire ~0.15 T dword-atomics/s, which is what bounded a scatter-add backward).
 * Pass the SAME struct to envgs_trace_forward and envgs_trace_backward.  The list path serves max_trace_depth == 0 with fewer than 2^24
 * surfels and 2^24 rays (ids, ray slots and entry counts are packed into 24-bit fields); anything else -- or a NULL struct, or cap == 0 --
 * takes the K-buffer kernels in both directions, which have no such limits.
 */
typedef struct envgs_trace_lists {
    uint32_t *hit_lists;     /* (R, cap, 2): after the forward the first n_used entries are sorted by (t, id); word 1 = surfel id.  Forward only:
                                envgs_trace_backward reads hit_cnt / n_used / hit_state / entries / pairs / surf_cnt / surf_off, so a caller may
                                release this buffer -- the largest of a call -- and pass NULL to the backward */
    int32_t *hit_cnt;        /* (R) hits found; > cap => that ray took the K-buffer path */
    int32_t *n_used;         /* (R) hits composited before termination */
    int32_t cap;             /* list capacity per ray, <= 1024; 0 disables the list path.  With cap > 0 (and max_trace_depth == 0, P and R below 2^24) BOTH
                                calls take the list path and every buffer of their direction must be present: a missing one is ENVGS_ERR_BAD_ARG, never
                                a silent fallback to the K-buffer kernels (the two calls could otherwise disagree about the path) */
    int32_t *stack_spill;    /* envgs_trace_stack_spill_ints(R) int32 */
    uint64_t *surf_acc;      /* (P,8) packed accumulators of the forward (8 copies per surfel, chosen by ray index, spread same-address
                                atomics): low 24 bits hit count, high 40 bits fixed-point weight */
    uint32_t *surf_cnt;      /* (P,8) (batch, surfel) entries per (surfel, copy) (list path only); a batch = 64 consecutive rays of the
                                coherence-sorted order */
    uint32_t *surf_off;      /* (P,8) inclusive prefix sum of surf_cnt; the last entry = number of gradient records */
    void *scan_temp;         /* envgs_raster_scan_temp_bytes(8*P) bytes */
    size_t scan_temp_bytes;
    uint32_t *ray_keys;      /* (2R words, 8 B aligned) scratch of the ray coherence sort: R (key << 32 | ray) pairs; NULL disables the sort */
    uint32_t *ray_order;     /* (2R) the second half receives the ray permutation the kernels use (the first half is spare) */
    void *ray_sort_temp;     /* envgs_trace_ray_sort_temp_bytes(R) bytes */
    size_t ray_sort_temp_bytes;
    float *records;          /* backward only: (num_records, 64) one 256 B gradient record per (batch, surfel) entry, grouped by surfel */
    uint64_t num_records;    /* backward only: capacity of `records` in records (>= surf_off[P-1]) */
    float *hit_state;        /* two PLANES of per-hit rows, written by the forward for the backward; rows = compact_rows (or R * cap).  Plane 0 (16 B rows, at
                                float index 0): (transmittance before the composited hit, the three colour prefix sums after it); plane 1 (at float index
                                4 * rows): (depth, normal prefix sums) in 16 B rows -- with has_others (depth, normal, the two aux sums) in 24 B rows
                                (until round 6 the aux sums were a third plane).  8 floats per row and hit, 10 with has_others.  A backward whose only
                                upstream gradient is the colour's reads plane 0 alone */
    uint64_t *entries;       /* (ceil(R/64), 64*cap) distinct surfels of every batch, packed id | hits-1 << 24 | slot << 32 */
    uint32_t *pairs;         /* (ceil(R/64), 64*cap) (lane << 16 | list position) of every composited hit, grouped by entry */
    int32_t *n_entries;      /* (ceil(R/64), 2) entries merged in the batch's table, single entries filed from the top */
    /* COMPACT per-hit buffers (optional; compact_rows == 0 selects the (R, cap) layouts documented above).  A ray uses a third of its list
     * capacity on average, so hit_state / entries / pairs are addressed through per-ray ROW offsets instead: after the collection the rays'
     * hit counts are scanned in coherence-sorted order (a batch's rows are contiguous), ray at sorted slot s owns rows
     * [row_off[s], row_off[s] + min(hit_cnt, cap)) of hit_state (compact_rows x 8|10 floats), and batch b owns the same row range
     * batch_rows[b] = {first row, rows} of entries / pairs (compact_rows elements each).  Each forward segment CLAIMS its rows from a counter
     * shared by the call's segments once its hit counts are known (round 4; a fixed share per segment starved whichever held the busier rays);
     * rays that do not fit are handed to the K-buffer kernels like rays whose list overflowed
     * (hit_cnt := cap + 1, counters[21] counts them) -- slower, never wrong.  The caller sizes compact_rows from the previous call's
     * total of hits found (counters[8..9]). */
    uint64_t compact_rows;   /* rows of hit_state / elements of entries and pairs; 0 = (R, cap) layouts */
    uint32_t *row_off;       /* (R) by sorted slot */
    uint32_t *batch_rows;    /* (ceil(R/64), 2) */
    uint32_t *row_blk;       /* (ceil(R/64)) scratch: per-batch row counts (written by the collection), scanned in place per segment */
    void *sh_perm;           /* optional scratch, (P, 48) elements of the shs storage type (used when sh_coeffs == 16): a quad-permuted copy of the SH
                                blocks, rebuilt by every forward, that lets four lanes fetch one surfel's block as contiguous 64 B runs; NULL = each
                                lane gathers its own block from shs */
    int32_t state_planes;    /* 0 (or >= 2): hit_state holds every plane.  1: the caller DECLARES that the backward of this forward will receive the
                                colour's upstream gradient only (dL_ddpt = dL_dacc = dL_dnorm = dL_daux = NULL) -- what EnvGS trains with, and the last
                                stage of a bounce chain: the forward writes plane 0 alone (hit_state: 4 floats per row) -- half of the per-hit bytes it
                                stores.  envgs_trace_backward returns ENVGS_ERR_BAD_ARG if another gradient arrives after all */
    uint32_t *sparse_hits;   /* optional (round 6), with sparse_cap: (sparse_cap, 4) uint32 scratch of the record backward -- SPARSE entries.  A (batch, surfel)
                                entry costs batch_surfel_bwd one 64-lane pass however few of the batch's rays blended the surfel, and a fifth of the entries
                                of the benchmark view hold 1-4 hits (1.4 % of the hits): the forward files the hits of such entries here as
                                (sorted ray slot, list position, surfel id, record slot) -- one gradient record PER HIT -- instead of creating the entry,
                                and envgs_trace_backward differentiates them one lane per hit (sparse_hits_bwd).  NULL / 0 = every entry takes the batch
                                kernel; hits that find no room here do too.  counters[64] = hits filed, counters[65] = entries left to the batch kernel.
                                Worth it for INCOHERENT batches only (a filed hit costs ~12x a hit of a full entry: bounce rays off rough geometry,
                                1-2 hits per entry: 59 -> 48 ms per step; the coherent benchmark views: +-0): envgs_amd.tracing passes the buffer when
                                the tracer's previous call averaged fewer than 6 composited hits per entry */
    uint64_t sparse_cap;     /* capacity of sparse_hits in hits */
    uint32_t defer_reduce;   /* optional (round 6), read by envgs_trace_backward: bit mask of ENVGS_TRACE_DEFER / _ACCUMULATE / _NO_FINISH (below).  0 = everything
                                on `stream`, outputs zeroed and complete on return (the reference's semantics).
                                DEFER: the SURFEL gradients (dmeans3D, dgrads3D, dscales, drots, dopacities, dshs / dcolors, and geo_rec) are finished on a
                                stream of the library's own -- the sum of the records and the conversion are a fifth of a millisecond that nothing the caller
                                queues next depends on (the base pass's backward wants the RAY gradients, which are complete on `stream` when the call
                                returns, as is dothers).  The caller must not touch the surfel gradients (read, free or reuse their memory, or the records /
                                surf_cnt / surf_off scratch) before envgs_trace_backward_join() has made its stream wait; every later envgs_trace_forward /
                                _backward of the device joins first by itself.
                                ACCUMULATE + NO_FINISH: the stages of a bounce chain differentiate the SAME surfels; instead of one set of gradient tensors
                                per stage and sums in the caller, the stages share the accumulators geo_rec and dshs / dcolors: every call but the first
                                passes ACCUMULATE (they are not zeroed, the record sums are added; the call does not join a pending tail on entry but
                                before its K-buffer pass, so the previous stage's sums run under this stage's record kernels), every call but the last
                                passes NO_FINISH (geo_rec stays unconverted; dmeans3D .. dopacities may be NULL).  The conversion is linear in geo_rec, so
                                converting the sum once equals the sum of the conversions */
    uint32_t reserved0;
} envgs_trace_lists;
#define ENVGS_TRACE_DEFER 1u
#define ENVGS_TRACE_ACCUMULATE 2u
#define ENVGS_TRACE_NO_FINISH 4u

/* Scratch bytes for the Morton sort + build of P surfels. */
ENVGS_API size_t envgs_bvh_temp_bytes(int32_t P);
/* Floats of the `nodes` buffer for P surfels: max(P-1,1) binary nodes of 16 floats (words 0-11 the two child boxes, 12-13 the child
 * references, 14 the parent, 15 the other end of the node's run of sorted leaves), followed by as many 4-wide nodes of 32 floats (the
 * grandchildren of each binary node, 8 floats per slot: lo.x hi.x lo.y hi.y lo.z hi.z ref 0; what the packet traversal of coherence-sorted
 * rays walks), followed by the sorted leaf order (P int32).  Topology words + order are what envgs_bvh_refit reads. */
ENVGS_API size_t envgs_bvh_node_floats(int32_t P);

/*
 * SurfelTracer.build_acceleration_structure(vertices, faces, rebuild) (optix_utils.py:78).
 * Builds the LBVH over the P quads (faces are implied by the get_disks layout: 2 triangles per 4 vertices).
 * opacities (P) is optional (NULL = plain quad boxes): when given, each leaf box is tightened to the part of the quad where the
 * surfel can still reach alpha >= 1/255 (disc of radius sqrt(2 ln(255 o)) sigma), and surfels with o <= 1/255 shrink to a point -- exact for every ray, and the reason the drop-in module defers the build to the first trace after a rebuild request
 * (only then are the opacities known).  nodes: envgs_bvh_node_floats(P) floats out.  temp: envgs_bvh_temp_bytes(P).
 */
ENVGS_API int envgs_bvh_build(int32_t P, const float *vertices, const float *opacities, float *nodes, void *temp, size_t temp_bytes,
                              int32_t debug, void *stream);

/*
 * build_acceleration_structure(vertices, faces, rebuild=False) -- OptiX's "update" (optix_utils.py:71-85 passes the flag through): REFIT the
 * structure a previous envgs_bvh_build left in `nodes` (same P) to moved vertices / changed opacities.  The topology and the leaf order are
 * kept (read from nodes_prev, carried over into `nodes`; the two may be the same buffer -- the drop-in module writes a fresh one because an
 * earlier forward whose backward is outstanding may still hold the old one); the leaf boxes, the range-union tables and every node box are
 * recomputed (no Morton keys, no sort, no hierarchy pass).  Exact: the
 * boxes are the unions of the NEW leaf boxes, so the hit sets are those of a fresh build; only the tree's quality follows the old positions.
 * Same temp size as the build.
 */
ENVGS_API int envgs_bvh_refit(int32_t P, const float *vertices, const float *opacities, const float *nodes_prev, float *nodes, void *temp,
                              size_t temp_bytes, int32_t debug, void *stream);

/*
 * Quality of a structure under its CURRENT boxes, for the module's build-or-refit decision (the reference's caller asks for a rebuild on every
 * training step, optix_utils.py:73-78; a refit is exact, so the module may answer such a request with one as long as the tree has not aged):
 * out2[0] = sum over the binary nodes of the surface areas (half areas: xy + yz + zx) of their two child boxes, out2[1] = the root's.  The
 * ratio out2[0] / out2[1] against its value right after the last full build is the growth of the SAH cost through refits.  Asynchronous
 * (memset + one launch on `stream`); out2 is device memory.
 */
ENVGS_API int envgs_bvh_quality(int32_t P, const float *nodes, float *out2, void *stream);

/*
 * SurfelTracer.forward (optix_utils.py:188-201): trace R rays through the surfel set, composite front to back.
 * srec (P,16) is scratch written here (and read again by the backward).  counters: 96 uint32 of scratch; after the
 * forward, words [2..13] hold six uint64 totals: composited hits, (unused), K-buffer traversal rounds, hits found, wide nodes
 * visited by the packet traversal, leaf tests of the packet traversal (diagnostics that the roofline accounting of bench.py
 * needs: BASELINE.md section 4 "hits / node_visits are data dependent").  Word [20] counts the 64-ray batches whose packet
 * traversal ran out of its shared stack (their rays were handed to the K-buffer kernels, nothing is dropped); words [24..27]
 * are the per-segment queues of rays with more than 256 hits.
 * final_T (R): stage-0 transmittance, kept for the backward.
 * List path (lists != NULL, lists->cap > 0, max_trace_depth == 0): each ray's hits are collected in ONE unordered traversal,
 * sorted by (t, id) in LDS and walked front to back; the backward walks the same lists and never touches the BVH.  Rays with
 * more than `cap` hits, and all rays when lists == NULL or bounces are requested, use the K-nearest-buffer traversal instead.
 * After the call counters[1] holds the largest hit_cnt (so the caller can size `cap` for the next call).
 */
ENVGS_API size_t envgs_trace_stack_spill_ints(int32_t num_rays);
ENVGS_API size_t envgs_trace_ray_sort_temp_bytes(int32_t num_rays);
/* The coherence order envgs_trace_forward forms its 64-ray batches in, on its own (parity tests; a caller that wants to reuse one order
 * for several traces): pairs (num_rays x uint64, scratch) ends up holding every ray's (key << 32 | ray id), order (num_rays) the ray ids
 * sorted by (key, id).  nodes / P: the acceleration structure whose root box normalises the origins (P == 0: the unit box). */
ENVGS_API int envgs_trace_ray_order(int32_t num_rays, const float *ray_o, const float *ray_d, const float *nodes, int32_t P,
                                    uint64_t *pairs, uint32_t *order, void *temp, size_t temp_bytes, void *stream);
ENVGS_API int envgs_trace_forward(const envgs_trace_cfg *cfg, const float *nodes,
                                  const float *ray_o, const float *ray_d,
                                  const float *means3D, const float *scales, const float *rotations, const float *opacities,
                                  const float *shs, const float *colors_precomp, const float *others_precomp, const float *bg,
                                  float *srec, uint32_t *counters,
                                  float *rgb, float *dpt, float *acc, float *norm, float *dist, float *aux, float *mid,
                                  float *wet, float *final_T, const envgs_trace_lists *lists, void *stream);

/*
 * SurfelTracer backward: gradients of stage 0 w.r.t. the surfel parameters AND the rays (reflected rays are
 * differentiable: easyvolcap/models/samplers/envgs_sampler.py:454-455).  Bounce stages are detached.
 * Outputs are zeroed here.  dshs (P,sh_coeffs,3) or dcolors (P,3); dothers may be NULL; dgrads3D (P,3) receives the
 * densification signal (= dL/dmeans3D; consumer: envgs_sampler.py:357-361).
 * Any of the five upstream gradients dL_drgb .. dL_daux may be NULL: an output the loss does not use has a zero gradient.
 */
ENVGS_API int envgs_trace_backward(const envgs_trace_cfg *cfg, const float *nodes,
                                   const float *ray_o, const float *ray_d,
                                   const float *means3D, const float *scales, const float *rotations, const float *opacities,
                                   const float *shs, const float *colors_precomp, const float *others_precomp, const float *bg,
                                   const float *srec, uint32_t *counters,
                                   const float *rgb, const float *dpt, const float *acc, const float *norm, const float *aux,
                                   const float *final_T,
                                   const float *dL_drgb, const float *dL_ddpt, const float *dL_dacc, const float *dL_dnorm,
                                   const float *dL_daux,
                                   float *geo_rec,
                                   float *dmeans3D, float *dgrads3D, float *dscales, float *drots, float *dopacities,
                                   float *dshs, float *dcolors, float *dothers, float *dray_o, float *dray_d,
                                   const envgs_trace_lists *lists, void *stream);

/* Makes `stream` wait for the deferred part of the device's last envgs_trace_backward (envgs_trace_lists::defer_reduce); nothing pending = no-op.
 * No reference counterpart (the reference's backward is one stream-ordered call: diff_surfel_tracing/__init__.py:144-204). */
ENVGS_API int envgs_trace_backward_join(void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ENVGS_TRACE_H */
