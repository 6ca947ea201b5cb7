// Microbenchmark: VALU wave-instruction issue rate of gfx950 (to price the issue-rate roofline of bench.py).
// hipcc --offload-arch=gfx950 -O3 scratch/valu_peak.hip -o /tmp/valu_peak && /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void spin(float *out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = 1.0001f, c = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pm = {m, m}, pc = {c, c};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 8; k++)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 16; k++)
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pc));
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++)
                asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_max_f32 %2, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_and_b32 %4, %4, %8\n v_add_u32 %5, %5, %9\n v_lshlrev_b32 %6, 1, %6\n v_mov_b32 %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE>
void run(const char *name, int waves_per_simd, float *out) {
    int blocks = 256 * waves_per_simd, iters = 20000;      // 256-thread blocks = 4 waves = one per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    spin<MODE><<<blocks, 256>>>(out, 100);
    hipEventRecord(e0);
    spin<MODE><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)blocks * 4 * iters * 64;
    printf("%-28s waves/SIMD %d: %.1f G wave-inst/s  (%.2f per cycle per CU at 2.4 GHz)\n", name, waves_per_simd, insts / ms / 1e6, insts / ms / 1e6 / 256 / 2.4);
}
int main() {
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4 * 2);
    for (int w : {1, 2, 4, 8}) { run<0>("v_fma_f32", w, out); run<1>("v_pk_fma_f32", w, out); run<2>("mixed int/fp VALU", w, out); }
    return 0;
}
