// fetch_calib.hip -- what rocprofv3's FETCH_SIZE reports on gfx950 for the access patterns of the tracer backward, against known byte counts
// (MI355X_MICROARCH.md, HBM section: "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//   stream16   : coalesced 16 B per lane streaming read of the whole buffer (the guide's case: FETCH_SIZE = 1/2 of the bytes)
//   gather32   : every lane reads 32 B (two 16 B loads) at a random 32 B-aligned offset of a buffer far larger than the 256 MB Infinity Cache
//   gather32x2 : the same offsets, but every 64 B line is visited twice -- by two different wavefronts, far apart in time (the per-hit
//                state of batch_surfel_bwd: the two 32 B halves of a line belong to consecutive hits of one ray, used by different entries)
//   gather16   : 16 B at random 16 B-aligned offsets
// Build: hipcc --offload-arch=gfx950 -O3 -o scratch/calib/fetch_calib scratch/calib/fetch_calib.hip ; run under rocprofv3 --pmc FETCH_SIZE --kernel-trace
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

__global__ void stream16(const float4 *__restrict__ p, size_t n16, float *out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}
// count gathers in total; slot = mix(index) mod nslots; SHARE = 1: every line's two halves are both visited (index i and i ^ (count/2) map to the two halves)
template <int BYTES, int SHARE>
__global__ void gather(const float4 *__restrict__ p, size_t nslots, size_t count, float *out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        size_t slot;
        if (SHARE) { const size_t half = count / 2, j = i < half ? i : i - half; slot = 2 * (mix(j) % (nslots / 2)) + (i < half ? 0 : 1); }
        else slot = mix(i) % nslots;
        const float4 *q = p + slot * (BYTES / 16);
        float4 v = q[0]; acc += v.x + v.w;
        if (BYTES == 32) { v = q[1]; acc += v.y + v.z; }
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main()
{
    const size_t bytes = 4ull << 30;                     // 4 GiB buffer: 16x the Infinity Cache
    float4 *buf; float *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 0, bytes);
    (void)hipDeviceSynchronize();
    const size_t count = 32ull << 20;                    // 32 Mi gathers per launch
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(stream16, dim3(4096), dim3(256), 0, 0, buf, bytes / 16, out);
        hipLaunchKernelGGL((gather<32, 0>), dim3(4096), dim3(256), 0, 0, buf, bytes / 32, count, out);
        hipLaunchKernelGGL((gather<32, 1>), dim3(4096), dim3(256), 0, 0, buf, bytes / 32, count, out);
        hipLaunchKernelGGL((gather<16, 0>), dim3(4096), dim3(256), 0, 0, buf, bytes / 16, count, out);
    }
    (void)hipDeviceSynchronize();
    printf("expected useful bytes per launch: stream16 %zu | gather32 %zu (distinct 64 B lines ~%zu B) | gather32x2 %zu (lines %zu B) | gather16 %zu (lines ~%zu B)\n",
           bytes, count * 32, count * 64, count * 32, count / 2 * 64, count * 16, count * 64);
    return 0;
}
