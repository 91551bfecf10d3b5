// Calibration of the FETCH_SIZE counter for the access pattern of k_accumulate: every lane reads
// 64-byte records (four 16-byte loads) at random 64-byte-aligned offsets of a 256 MiB table, the
// shape of the base gather.  Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE`: the kernel
// moves exactly records * 64 bytes (development tool).
// Build: hipcc -O3 --offload-arch=gfx950 tools/gather_probe.hip -o tools/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct alignas(16) U4 { uint32_t x, y, z, w; };
__global__ void k_gather64(const U4 *table, uint32_t nrec, uint32_t per_lane, uint32_t *out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, acc = 0;
    uint64_t s = 0x9E3779B97F4A7C15ull * (t + 1);
    for (uint32_t i = 0; i < per_lane; i++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const U4 *p = table + (size_t)((uint32_t)(s >> 33) % nrec) * 4;
        U4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    out[t] = acc;
}
__global__ void k_stream(const U4 *table, size_t n16, uint32_t *out) {          // reference: a plain streaming read
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) acc ^= table[i].x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    const size_t bytes = 256ull << 20;
    const uint32_t nrec = (uint32_t)(bytes / 64), lanes = 256 * 4 * 3 * 64, per_lane = 64;
    U4 *table; uint32_t *out;
    hipMalloc(&table, bytes); hipMalloc(&out, (size_t)lanes * 4 + 4096 * 256 * 4);
    hipMemset(table, 1, bytes);
    for (int r = 0; r < 3; r++) {
        hipLaunchKernelGGL(k_gather64, dim3(lanes / 128), dim3(128), 0, 0, table, nrec, per_lane, out);
        hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, table, bytes / 16, out);
    }
    hipDeviceSynchronize();
    printf("k_gather64 moves %.1f MiB per launch (%u records of 64 B); k_stream reads %.1f MiB\n", (double)lanes * per_lane * 64 / (1 << 20), lanes * per_lane, (double)bytes / (1 << 20));
    return 0;
}
