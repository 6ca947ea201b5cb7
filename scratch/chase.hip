// Microbenchmark: how fast can a wavefront walk a chain of DEPENDENT uniform records (a BVH node, then a leaf record) on gfx950 --
// through the scalar cache (s_load, what collect_hits_coop does) or through the vector path (a few lanes fetch, LDS broadcast)?
// Decides whether the collection kernel is bound by per-wave latency or by the scalar cache's miss throughput (round 5).
//   hipcc --offload-arch=gfx950 -O3 scratch/chase.hip -o /tmp/chase && /tmp/chase
// Model of one traversal step: the node (128 B) names the next node and a leaf record (64 B); ~72 VALU of "slab tests" depend on the node and
// gate the next index, ~60 VALU of "exact test" depend on the leaf.  Regions: each XCD walks its own RN nodes + RN leaves (L2 resident).
//   MODE 0  scalar, sequential   : node -> slab -> leaf -> exact -> next node          (the shipped kernel's chain)
//   MODE 1  scalar, pipelined    : node -> slab -> [next node + leaf in flight] -> exact
//   MODE 2  vector+LDS, pipelined: lanes 0..7 fetch the node and lanes 8..11 the leaf in ONE global_load_dwordx4, ds_write, broadcast ds_read
//   MODE 3  vector+LDS, sequential
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ int xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return (int)(v & 7u); }

__device__ __forceinline__ float slab(float acc, const float4 a, const float4 b, float ox)      // ~18 VALU per child slot
{
    float t0 = (a.x - ox) * acc, t1 = (a.y - ox) * acc, t2 = (a.z - ox) * acc, t3 = (a.w - ox) * acc, t4 = (b.x - ox) * acc, t5 = (b.y - ox) * acc;
    return fmaxf(fmaxf(fminf(t0, t1), fminf(t2, t3)), fminf(t4, t5)) - fminf(fminf(fmaxf(t0, t1), fmaxf(t2, t3)), fmaxf(t4, t5));
}
__device__ __forceinline__ float exact(float acc, const float4 s0, const float4 s1, const float4 s2, const float4 s3, float ox)   // ~50 VALU
{
    const float den = s3.x * ox + s3.y * acc + s3.z;
    const float t = (s3.x * (s0.x - ox) + s3.y * (s0.y - acc) + s3.z * (s0.z - ox)) / den;
    const float qx = ox + t * acc - s0.x, qy = acc + t * ox - s0.y, qz = ox + t - s0.z;
    const float u = s1.x * qx + s1.y * qy + s1.z * qz, v = s2.x * qx + s2.y * qy + s2.z * qz;
    return s0.w * __expf(-0.5f * (u * u + v * v)) + t * 1e-9f;
}
__device__ __forceinline__ int zero_of(float x)        // a wave-uniform 0 the compiler cannot see through: ties the next index to the VALU result
{
    int r = __builtin_amdgcn_readfirstlane(__float_as_int(x));
    asm volatile("s_and_b32 %0, %0, 0" : "+s"(r));
    return r;
}

typedef float f8 __attribute__((ext_vector_type(8)));
#define SLOAD8(dst, ptr, off) asm volatile("s_load_dwordx8 %0, %1, " #off : "=s"(dst) : "s"(ptr))

template <int MODE>
__global__ void __launch_bounds__(256, 8) chase(const float4 *__restrict__ nodes, const float4 *__restrict__ leaves, int RN, int steps, float *out)
{
    __shared__ float4 stage[4][12];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int region = xcc_id();
    const float4 *N = nodes + (size_t)region * RN * 8, *L = leaves + (size_t)region * RN * 4;
    int cur = (int)((blockIdx.x * 4u + wave) * 2654435761u % (unsigned)RN);
    const float ox = 0.001f * lane;
    float acc = 1.0f + 0.01f * lane, res = 0.f;
    if (MODE == 0) {
        for (int s = 0; s < steps; s++) {
            const float4 *nd = N + (size_t)cur * 8;
            float4 q[8];
#pragma unroll
            for (int c = 0; c < 8; c++) q[c] = nd[c];
            float r = 0.f;
#pragma unroll
            for (int c = 0; c < 4; c++) r += slab(acc, q[2 * c], q[2 * c + 1], ox);
            const int z = zero_of(r);
            const int nxt = (__float_as_int(q[1].z) | z), lf = (__float_as_int(q[1].w) | z);
            const float4 *lr = L + (size_t)lf * 4;
            res += exact(acc, lr[0], lr[1], lr[2], lr[3], ox) + r * 1e-9f;
            cur = nxt | zero_of(res);
        }
    } else if (MODE == 1) {
        // the compiler sinks plain loads of the NEXT node to the loop header (i.e. behind the leaf's exact test): the pipelined order needs the
        // loads as volatile asm, and the consumers tied to an explicit s_waitcnt through "+s" operands
        f8 n0, n1, n2, n3, l0, l1;
        {
            const float4 *nd = N + (size_t)cur * 8;
            SLOAD8(n0, nd, 0); SLOAD8(n1, nd, 32); SLOAD8(n2, nd, 64); SLOAD8(n3, nd, 96);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(n0), "+s"(n1), "+s"(n2), "+s"(n3));
        }
        for (int s = 0; s < steps; s++) {
            float r = slab(acc, make_float4(n0[0], n0[1], n0[2], n0[3]), make_float4(n0[4], n0[5], n0[6], n0[7]), ox);
            r += slab(acc, make_float4(n1[0], n1[1], n1[2], n1[3]), make_float4(n1[4], n1[5], n1[6], n1[7]), ox);
            r += slab(acc, make_float4(n2[0], n2[1], n2[2], n2[3]), make_float4(n2[4], n2[5], n2[6], n2[7]), ox);
            r += slab(acc, make_float4(n3[0], n3[1], n3[2], n3[3]), make_float4(n3[4], n3[5], n3[6], n3[7]), ox);
            const int z = zero_of(r);
            const int nxt = (__float_as_int(n0[6]) | z), lf = (__float_as_int(n0[7]) | z);
            const float4 *nd = N + (size_t)nxt * 8, *lr = L + (size_t)lf * 4;
            SLOAD8(l0, lr, 0); SLOAD8(l1, lr, 32);
            SLOAD8(n0, nd, 0); SLOAD8(n1, nd, 32); SLOAD8(n2, nd, 64); SLOAD8(n3, nd, 96);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(n0), "+s"(n1), "+s"(n2), "+s"(n3), "+s"(l0), "+s"(l1));
            res += exact(acc, make_float4(l0[0], l0[1], l0[2], l0[3]), make_float4(l0[4], l0[5], l0[6], l0[7]), make_float4(l1[0], l1[1], l1[2], l1[3]),
                         make_float4(l1[4], l1[5], l1[6], l1[7]), ox) + r * 1e-9f;
        }
    } else {
        float4 *st = stage[wave];
        auto fetch = [&](int node, int leaf) -> float4 {       // lanes 0..7: the node's eight float4, lanes 8..11: the leaf's four (leaf < 0: none)
            const float4 *p = lane < 8 ? N + (size_t)node * 8 + lane : L + (size_t)(leaf < 0 ? 0 : leaf) * 4 + (lane - 8);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lane < (leaf < 0 ? 8 : 12)) v = *p;
            return v;
        };
        float4 v = fetch(cur, -1);
        if (lane < 8) st[lane] = v;
        for (int s = 0; s < steps; s++) {
            float r = 0.f;
#pragma unroll
            for (int c = 0; c < 4; c++) r += slab(acc, st[2 * c], st[2 * c + 1], ox);     // broadcast ds_read_b128
            const int z = zero_of(r);
            const float4 q1 = st[1];
            const int nxt = (__builtin_amdgcn_readfirstlane(__float_as_int(q1.z)) | z), lf = (__builtin_amdgcn_readfirstlane(__float_as_int(q1.w)) | z);
            if (MODE == 2) {
                v = fetch(nxt, lf);
                if (lane < 12) st[lane] = v;
                res += exact(acc, st[8], st[9], st[10], st[11], ox) + r * 1e-9f;
            } else {
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (lane >= 8 && lane < 12) w = L[(size_t)lf * 4 + (lane - 8)];
                if (lane >= 8 && lane < 12) st[lane] = w;
                res += exact(acc, st[8], st[9], st[10], st[11], ox) + r * 1e-9f;
                const int n2 = nxt | zero_of(res);
                v = fetch(n2, -1);
                if (lane < 8) st[lane] = v;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = res + (float)cur;
}

template <int MODE>
static void run(const char *name, const float4 *nodes, const float4 *leaves, int RN, int wgs_per_cu, float *out)
{
    const int blocks = 256 * wgs_per_cu, steps = 1500;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    chase<MODE><<<blocks, 256>>>(nodes, leaves, RN, 50, out);
    hipEventRecord(e0);
    chase<MODE><<<blocks, 256>>>(nodes, leaves, RN, steps, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double total = (double)blocks * 4 * steps;
    printf("%-34s RN %6d  WGs/CU %d (%2d waves/SIMD): %7.1f M steps/s   %6.0f ns per step per wave   (%.3f ms)\n", name, RN, wgs_per_cu, wgs_per_cu, total / ms / 1e3,
           ms * 1e6 / steps, ms);
}

int main()
{
    float *out; hipMalloc(&out, sizeof(float) * 256 * 16 * 256);
    for (int RN : {4096, 16384, 65536}) {
        const size_t nn = (size_t)8 * RN;
        std::vector<float> hn(nn * 32), hl(nn * 16);
        srand(7);
        for (size_t i = 0; i < nn; i++) {
            for (int k = 0; k < 32; k++) hn[i * 32 + k] = 0.5f + (float)(rand() % 1000) * 1e-3f;
            for (int k = 0; k < 16; k++) hl[i * 16 + k] = 0.5f + (float)(rand() % 1000) * 1e-3f;
            const int nxt = (int)(((unsigned)rand() * 2654435761u) % (unsigned)RN), lf = (int)(((unsigned)rand() * 40503u + 17u) % (unsigned)RN);
            ((int *)hn.data())[i * 32 + 6] = nxt;       // q[1].z
            ((int *)hn.data())[i * 32 + 7] = lf;        // q[1].w
        }
        float4 *dn, *dl;
        hipMalloc(&dn, hn.size() * 4); hipMalloc(&dl, hl.size() * 4);
        hipMemcpy(dn, hn.data(), hn.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dl, hl.data(), hl.size() * 4, hipMemcpyHostToDevice);
        for (int w : {2, 4, 8}) {
            run<0>("scalar sequential (shipped chain)", dn, dl, RN, w, out);
            run<1>("scalar pipelined", dn, dl, RN, w, out);
            run<3>("vector+LDS sequential", dn, dl, RN, w, out);
            run<2>("vector+LDS pipelined", dn, dl, RN, w, out);
        }
        hipFree(dn); hipFree(dl);
    }
    return 0;
}
