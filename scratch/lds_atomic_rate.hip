// Round 6 microbenchmark: ds_add_f32 (LDS float atomic add, no return) wave-instruction rate of gfx950 under the three address patterns a
// pair-mapped R7 (one lane per live (pixel, splat) pair, per-splat gradient accumulators in LDS) would produce -- to price VERDICT r5 item 3.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scratch/lds_atomic_rate.hip -o scratch/lds_atomic_rate.bin && scratch/lds_atomic_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ACC = 192 * 20;           // accumulators of one 192-splat batch: 20 gradient words per splat (15 + C at C = 5) = 15 KB
template <int MODE>
__global__ __launch_bounds__(256) void spin(float *out, int iters) {
    __shared__ float acc[ACC];
    for (int i = threadIdx.x; i < ACC; i += 256) acc[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    float v = 1.0f;
    for (int i = 0; i < iters; i++) {
        s = s * 1664525u + 1013904223u;
        int splat;
        if (MODE == 0) splat = lane;                                  // 64 different splats, consecutive: conflict-free
        else if (MODE == 1) splat = (s >> 8) % 192;                   // pixel-major lanes: every lane another splat of the batch, at random
        else if (MODE == 2) splat = ((s >> 8) % 192) & ~1 | (lane & 1);   // (pairs of lanes on neighbouring splats)
        else splat = (lane / 28) + (i & 63);                          // splat-major lanes: ~28 lanes (one splat's live pixels) on ONE accumulator
#pragma unroll
        for (int w = 0; w < 20; w++) atomicAdd(&acc[splat * 20 + w], v);
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = acc[threadIdx.x];
}
// ds_add_u32 vs ds_add_f32 with 64 / 14 live lanes (the collection's per-ray optical-depth bins: ~14 accepted hits per leaf test)
template <int INT, int LIVE>
__global__ __launch_bounds__(256) void spin2(float *out, int iters) {
    __shared__ float accf[64 * 36];
    __shared__ unsigned acci[64 * 36];
    for (int i = threadIdx.x; i < 64 * 36; i += 256) { accf[i] = 0.f; acci[i] = 0u; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int w = 0; w < 20; w++) {
            s = s * 1664525u + 1013904223u;
            const int b = (s >> 10) & 31;
            if (lane < LIVE) { if (INT) atomicAdd(&acci[lane * 36 + b], 3u); else atomicAdd(&accf[lane * 36 + b], 1.0f); }
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = accf[threadIdx.x] + (float)acci[threadIdx.x];
}
template <int INT, int LIVE>
void run2(const char *name, float *out) {
    int blocks = 256 * 4, iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    spin2<INT, LIVE><<<blocks, 256>>>(out, 10);
    hipEventRecord(e0);
    spin2<INT, LIVE><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)blocks * 4 * iters * 20;
    printf("%-44s %7.1f G wave-inst/s = %.3f per cycle per CU (%.0f cycles each)\n", name, insts / ms / 1e6, insts / ms / 1e6 / 256 / 2.4, 256 * 2.4e9 / (insts / ms * 1e3));
}
// the other LDS atomics the tracer uses: 64-bit add (register_hits' per-surfel accumulators), returning forms, compare-and-swap (its hash keys)
template <int KIND, int LIVE>
__global__ __launch_bounds__(256) void spin3(float *out, int iters) {
    __shared__ unsigned long long a64[1024];
    __shared__ int a32[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) { a64[i] = 0ull; a32[i] = -1; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    unsigned long long sink = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int w = 0; w < 20; w++) {
            s = s * 1664525u + 1013904223u;
            const int h = (s >> 10) & 1023;
            if (lane < LIVE) {
                if (KIND == 0) atomicAdd(&a64[h], 257ull);
                else if (KIND == 1) sink += atomicAdd(&a64[h], 257ull);
                else if (KIND == 2) sink += (unsigned)atomicAdd(&a32[h], 1);
                else sink += (unsigned)atomicCAS(&a32[h], -1, (int)s);
            }
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = (float)a64[threadIdx.x] + (float)a32[threadIdx.x] + (float)sink;
}
template <int KIND, int LIVE>
void run3(const char *name, float *out) {
    int blocks = 256 * 4, iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    spin3<KIND, LIVE><<<blocks, 256>>>(out, 10);
    hipEventRecord(e0);
    spin3<KIND, LIVE><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)blocks * 4 * iters * 20;
    printf("%-44s %7.1f G wave-inst/s (%.0f cycles each)\n", name, insts / ms / 1e6, 256 * 2.4e9 / (insts / ms * 1e3));
}
template <int MODE>
void run(const char *name, int wgs_per_cu, float *out) {
    int blocks = 256 * wgs_per_cu, iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    spin<MODE><<<blocks, 256>>>(out, 10);
    hipEventRecord(e0);
    spin<MODE><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)blocks * 4 * iters * 20;
    printf("%-44s %d workgroups/CU: %7.1f G ds_add wave-inst/s = %.3f per cycle per CU at 2.4 GHz  (%.2f ms per 13.4 M instructions)\n", name, wgs_per_cu,
           insts / ms / 1e6, insts / ms / 1e6 / 256 / 2.4, 13.4e6 / (insts / ms));
}
int main() {
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {2, 4}) {
        run<0>("64 consecutive splats (conflict-free)", w, out);
        run<1>("pixel-major: random splat per lane", w, out);
        run<3>("splat-major: ~28 lanes per accumulator", w, out);
    }
    run2<0, 64>("ds_add_f32, 64 live lanes, per-lane rows", out);
    run2<1, 64>("ds_add_u32, 64 live lanes, per-lane rows", out);
    run2<0, 14>("ds_add_f32, 14 live lanes", out);
    run2<1, 14>("ds_add_u32, 14 live lanes", out);
    run3<0, 64>("ds_add_u64, 64 lanes, random slots", out);
    run3<1, 64>("ds_add_rtn_u64, 64 lanes", out);
    run3<2, 64>("ds_add_rtn_u32, 64 lanes", out);
    run3<3, 64>("ds_cmpst_rtn_b32 (CAS), 64 lanes", out);
    return 0;
}
