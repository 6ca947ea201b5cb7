// densify.hip -- prune compaction of the per-Gaussian SoA and the 3-NN initialisation helper (include/envgs_densify.h).
#include "common.h"

#include "../../include/envgs_densify.h"

namespace envgs {

struct CompactBatch {
    envgs_rows_tensor t[ENVGS_COMPACT_MAX_TENSORS];
    long long chunk_start[ENVGS_COMPACT_MAX_TENSORS + 1];     // prefix of 256-word chunks
    int count;
};

__global__ void __launch_bounds__(256)
mask_to_flags(long long P, const uint8_t *__restrict__ keep, uint32_t *__restrict__ flags)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < P) flags[i] = keep[i] ? 1u : 0u;
}

// inclusive -> exclusive in place, and the total
__global__ void __launch_bounds__(256)
finish_positions(long long P, const uint8_t *__restrict__ keep, uint32_t *__restrict__ pos, uint32_t *__restrict__ n_kept)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const uint32_t incl = pos[i];
    if (i == P - 1) *n_kept = incl;
    pos[i] = incl - (keep[i] ? 1u : 0u);
}

// one launch for all tensors: a workgroup copies 256 consecutive 4-byte words of one tensor (coalesced reads; the writes of kept rows
// are contiguous too, because kept rows stay in order)
__global__ void __launch_bounds__(256)
compact_gather(const CompactBatch B, const long long P, const uint8_t *__restrict__ keep, const uint32_t *__restrict__ pos)
{
    const long long chunk = blockIdx.x;
    int ti = 0;
    while (ti + 1 < B.count && chunk >= B.chunk_start[ti + 1]) ti++;
    const envgs_rows_tensor T = B.t[ti];
    const long long w = T.row_bytes >> 2;
    const long long e = (chunk - B.chunk_start[ti]) * 256 + threadIdx.x;
    if (e >= P * w) return;
    const long long row = e / w;
    if (!keep[row]) return;
    const long long col = e - row * w;
    reinterpret_cast<uint32_t *>(T.dst)[(long long)pos[row] * w + col] = reinterpret_cast<const uint32_t *>(T.src)[e];
}

// exact 3 nearest neighbours by brute force: 256 query points per workgroup, candidates streamed through LDS in tiles of 256
__global__ void __launch_bounds__(256)
knn3_mean_dist2(int P, const float *__restrict__ xyz, float *__restrict__ out)
{
    __shared__ float sx[256], sy[256], sz[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < P;
    const float px = valid ? xyz[3 * i] : 0.f, py = valid ? xyz[3 * i + 1] : 0.f, pz = valid ? xyz[3 * i + 2] : 0.f;
    float b0 = 3.0e38f, b1 = 3.0e38f, b2 = 3.0e38f;
    for (int base = 0; base < P; base += 256) {
        const int j = base + threadIdx.x;
        __syncthreads();
        sx[threadIdx.x] = j < P ? xyz[3 * j] : 0.f; sy[threadIdx.x] = j < P ? xyz[3 * j + 1] : 0.f; sz[threadIdx.x] = j < P ? xyz[3 * j + 2] : 0.f;
        __syncthreads();
        const int nt = min(256, P - base);
        for (int k = 0; k < nt; k++) {
            const float dx = sx[k] - px, dy = sy[k] - py, dz = sz[k] - pz;
            float d = dx * dx + dy * dy + dz * dz;
            if (base + k == i) d = 3.0e38f;                       // not its own neighbour
            // insert into the sorted triple
            const float m0 = fminf(b0, d), x0 = fmaxf(b0, d);
            const float m1 = fminf(b1, x0), x1 = fmaxf(b1, x0);
            b0 = m0; b1 = m1; b2 = fminf(b2, x1);
        }
    }
    if (valid) {
        float s = 0.f; int c = 0;
        if (b0 < 1.0e38f) { s += b0; c++; }
        if (b1 < 1.0e38f) { s += b1; c++; }
        if (b2 < 1.0e38f) { s += b2; c++; }
        out[i] = c ? s / (float)c : 0.f;
    }
}

}  // namespace envgs

using namespace envgs;

extern "C" {

size_t envgs_compact_temp_bytes(int64_t P) { return scan_temp_bytes((int)(P > 0 ? P : 1)); }

int envgs_compact_scan(int64_t P, const uint8_t *keep, uint32_t *positions, uint32_t *n_kept, void *temp, size_t temp_bytes, void *stream_)
{
    if (P < 0 || P >= (1ll << 31) || !n_kept) return ENVGS_ERR_BAD_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    if (P == 0) return (int)hipMemsetAsync(n_kept, 0, sizeof(uint32_t), stream);
    if (!keep || !positions || !temp) return ENVGS_ERR_BAD_ARG;
    if (temp_bytes < scan_temp_bytes((int)P)) return ENVGS_ERR_TEMP_TOO_SMALL;
    const unsigned nb = (unsigned)((P + 255) / 256);
    hipLaunchKernelGGL(mask_to_flags, dim3(nb), dim3(256), 0, stream, (long long)P, keep, positions);
    const int rc = launch_scan(positions, positions, (int)P, temp, temp_bytes, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(finish_positions, dim3(nb), dim3(256), 0, stream, (long long)P, keep, positions, n_kept);
    return (int)hipGetLastError();
}

int envgs_compact_gather(int32_t count, const envgs_rows_tensor *tensors, int64_t P, const uint8_t *keep, const uint32_t *positions, void *stream_)
{
    if (count < 0 || count > ENVGS_COMPACT_MAX_TENSORS || P < 0 || (count > 0 && !tensors)) return ENVGS_ERR_BAD_ARG;
    if (P == 0 || count == 0) return 0;
    if (!keep || !positions) return ENVGS_ERR_BAD_ARG;
    CompactBatch B;
    B.count = 0;
    long long chunks = 0;
    for (int i = 0; i < count; i++) {
        if (tensors[i].row_bytes <= 0) continue;
        if ((tensors[i].row_bytes & 3) || !tensors[i].src || !tensors[i].dst) return ENVGS_ERR_BAD_ARG;
        B.t[B.count] = tensors[i];
        B.chunk_start[B.count] = chunks;
        chunks += ((long long)P * (tensors[i].row_bytes >> 2) + 255) / 256;
        B.count++;
    }
    B.chunk_start[B.count] = chunks;
    if (chunks == 0) return 0;
    if (chunks >= (1ll << 31)) return ENVGS_ERR_BAD_ARG;
    hipLaunchKernelGGL(compact_gather, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream_, B, (long long)P, keep, positions);
    return (int)hipGetLastError();
}

int envgs_knn3_mean_dist2(int32_t P, const float *xyz, float *out, void *stream_)
{
    if (P < 0) return ENVGS_ERR_BAD_ARG;
    if (P == 0) return 0;
    if (!xyz || !out) return ENVGS_ERR_BAD_ARG;
    hipLaunchKernelGGL(knn3_mean_dist2, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream_, P, xyz, out);
    return (int)hipGetLastError();
}

}  // extern "C"
