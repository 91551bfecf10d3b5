// Memory-system probe for the batched-affine design question (development tool): how fast can
// MI355X gather random 64-byte / 32-byte records (the shape of a base gather) from tables of
// 64 MiB ... 4 GiB, beside plain streaming reads and writes.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mem_probe.hip -o tools/mem_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
struct alignas(16) U4 { uint32_t x, y, z, w; };
struct alignas(8) U2 { uint32_t x, y; };

__device__ inline uint32_t mix(const U4 &a) { return a.x ^ a.y ^ a.z ^ a.w; }

// every lane gathers whole records of REC bytes, U records in flight
template <int REC, int U>
__global__ void k_gather(const unsigned char *__restrict__ table, const uint32_t *__restrict__ idx, uint32_t per_lane, uint32_t *out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, T = gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < per_lane; i += U) {
        U4 v[U][REC / 16];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const U4 *p = reinterpret_cast<const U4 *>(table + (size_t)idx[(size_t)(i + u) * T + t] * REC);
#pragma unroll
            for (int k = 0; k < REC / 16; k++) v[u][k] = p[k];
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int k = 0; k < REC / 16; k++) acc += mix(v[u][k]);
    }
    out[t] = acc;
}
// four lanes share a 64-byte record (16 B each): 16 records per wave instruction
template <int U>
__global__ void k_gather_coop(const unsigned char *__restrict__ table, const uint32_t *__restrict__ idx, uint32_t per_group, uint32_t *out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, T = gridDim.x * blockDim.x;
    const uint32_t grp = t >> 2, sub = t & 3, G = T >> 2;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < per_group; i += U) {
        U4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = *reinterpret_cast<const U4 *>(table + (size_t)idx[(size_t)(i + u) * G + grp] * 64 + sub * 16);
#pragma unroll
        for (int u = 0; u < U; u++) acc += mix(v[u]);
    }
    out[t] = acc;
}
__global__ void k_read(const U4 *__restrict__ src, size_t n16, uint32_t *out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) acc ^= src[i].x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k_write(U4 *__restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = U4{(uint32_t)i, 1, 2, 3};
}
__global__ void k_copy(const U4 *__restrict__ src, U4 *__restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void k_fill_idx(uint32_t *idx, size_t n, uint32_t nrec, uint64_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = seed + i * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        idx[i] = (uint32_t)(z % nrec);
    }
}

template <class Fn> static float time_ms(Fn launch, int reps = 5) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(a, 0); launch(); hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const size_t max_bytes = 4ull << 30;
    unsigned char *table; uint32_t *idx, *out; U4 *dst;
    const uint32_t lanes = 256 * 4 * 8 * 64;                    // 8 waves per SIMD
    const uint32_t per_lane = 32;
    const size_t nidx = (size_t)lanes * per_lane;               // 16.8 M gathers per launch
    hipMalloc(&table, max_bytes); hipMalloc(&dst, 1ull << 30); hipMalloc(&idx, nidx * 4); hipMalloc(&out, (size_t)8192 * 256 * 4);   // k_read runs 8192 x 256 lanes, the gathers `lanes`
    hipMemset(table, 1, max_bytes);
    printf("stream: ");
    {
        size_t n16 = (1ull << 30) / 16;
        float r = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, (const U4 *)table, n16, out); });
        float w = time_ms([&] { hipLaunchKernelGGL(k_write, dim3(8192), dim3(256), 0, 0, dst, n16); });
        float c = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, (const U4 *)table, dst, n16); });
        printf("read 1 GiB %.3f ms (%.0f GB/s)  write %.3f ms (%.0f GB/s)  copy %.3f ms (%.0f GB/s r+w)\n", r, 1.0737e3 / r * 1e0, w, 1.0737e3 / w, c, 2 * 1.0737e3 / c);
        size_t big = max_bytes / 16;
        float r4 = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, (const U4 *)table, big, out); });
        printf("        read 4 GiB %.3f ms (%.0f GB/s)\n", r4, 4 * 1.0737e3 / r4);
    }
    const size_t sizes[] = {64ull << 20, 128ull << 20, 256ull << 20, 512ull << 20, 1ull << 30, 4ull << 30};
    for (size_t bytes : sizes) {
        for (int rec : {64, 32}) {
            uint32_t nrec = (uint32_t)(bytes / rec);
            hipLaunchKernelGGL(k_fill_idx, dim3(4096), dim3(256), 0, 0, idx, nidx, nrec, 0x1234 + bytes);
            float t1, t2, t4;
            if (rec == 64) {
                t1 = time_ms([&] { hipLaunchKernelGGL((k_gather<64, 1>), dim3(lanes / 256), dim3(256), 0, 0, table, idx, per_lane, out); });
                t2 = time_ms([&] { hipLaunchKernelGGL((k_gather<64, 2>), dim3(lanes / 256), dim3(256), 0, 0, table, idx, per_lane, out); });
                t4 = time_ms([&] { hipLaunchKernelGGL((k_gather<64, 4>), dim3(lanes / 256), dim3(256), 0, 0, table, idx, per_lane, out); });
            } else {
                t1 = time_ms([&] { hipLaunchKernelGGL((k_gather<32, 1>), dim3(lanes / 256), dim3(256), 0, 0, table, idx, per_lane, out); });
                t2 = time_ms([&] { hipLaunchKernelGGL((k_gather<32, 2>), dim3(lanes / 256), dim3(256), 0, 0, table, idx, per_lane, out); });
                t4 = time_ms([&] { hipLaunchKernelGGL((k_gather<32, 4>), dim3(lanes / 256), dim3(256), 0, 0, table, idx, per_lane, out); });
            }
            double g = (double)nidx / 1e6;
            printf("table %5zu MiB rec %2d B: U=1 %.3f ms (%.1f G rec/s, %.0f GB/s)  U=2 %.3f ms (%.1f, %.0f)  U=4 %.3f ms (%.1f, %.0f)\n", bytes >> 20, rec,
                   t1, g / t1, g * rec / t1, t2, g / t2, g * rec / t2, t4, g / t4, g * rec / t4);
        }
        uint32_t nrec = (uint32_t)(bytes / 64);
        hipLaunchKernelGGL(k_fill_idx, dim3(4096), dim3(256), 0, 0, idx, nidx, nrec, 0x777 + bytes);
        // cooperative: lanes/4 groups x per_lane*4 records = the same number of records
        float c1 = time_ms([&] { hipLaunchKernelGGL((k_gather_coop<1>), dim3(lanes / 256), dim3(256), 0, 0, table, idx, per_lane * 4, out); });
        float c4 = time_ms([&] { hipLaunchKernelGGL((k_gather_coop<4>), dim3(lanes / 256), dim3(256), 0, 0, table, idx, per_lane * 4, out); });
        double g = (double)nidx / 1e6;
        printf("table %5zu MiB rec 64 B, 4 lanes per record: U=1 %.3f ms (%.1f G rec/s, %.0f GB/s)  U=4 %.3f ms (%.1f, %.0f)\n", bytes >> 20, c1, g / c1, g * 64 / c1, c4, g / c4,
               g * 64 / c4);
    }
    return 0;
}
