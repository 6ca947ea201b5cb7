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
    return 0;
}
