// Issue rates of the 64-bit VALU helpers the 29-bit-limb multiplier leans on besides
// v_mad_u64_u32 (development tool).  Build: hipcc -O3 --offload-arch=gfx950 tools/microbench64.hip -o tools/microbench64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int ITERS = 4096;
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
__global__ void mb_shr64(uint64_t *out, uint64_t seed) {
    uint64_t x[8];
    for (int k = 0; k < 8; k++) x[k] = seed * (threadIdx.x + 3 + k) | (1ull << 63);
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) { uint64_t t; asm volatile("v_lshrrev_b64 %0, 29, %1" : "=v"(t) : "v"(x[k])); x[k] = t | (1ull << 63); }
    }
    uint64_t r = 0; for (int k = 0; k < 8; k++) r ^= x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void mb_add64(uint64_t *out, uint64_t seed) {
    uint64_t x[8], y = seed + threadIdx.x;
    for (int k = 0; k < 8; k++) x[k] = seed * (threadIdx.x + 3 + k);
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) { uint64_t t; asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(t) : "v"(x[k]), "v"(y)); x[k] = t; }
    }
    uint64_t r = 0; for (int k = 0; k < 8; k++) r ^= x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void mb_addc(uint64_t *out, uint64_t seed) {            // the same 64-bit add as two 32-bit halves
    uint32_t lo[8], hi[8], yl = (uint32_t)seed + threadIdx.x, yh = (uint32_t)(seed >> 32);
    for (int k = 0; k < 8; k++) { lo[k] = (uint32_t)seed * (threadIdx.x + 3 + k); hi[k] = k; }
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo[k]), "+v"(hi[k]) : "v"(yl), "v"(yh) : "vcc");
    }
    uint64_t r = 0; for (int k = 0; k < 8; k++) r ^= ((uint64_t)hi[k] << 32) | lo[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void mb_and32(uint32_t *out, uint32_t seed) {
    uint32_t x[8];
    for (int k = 0; k < 8; k++) x[k] = seed * (threadIdx.x + 3 + k);
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) asm volatile("v_and_b32 %0, 0x1fffffff, %0\n\tv_add_u32 %0, %0, %1" : "+v"(x[k]) : "v"(seed));
    }
    uint32_t r = 0; for (int k = 0; k < 8; k++) r ^= x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <class K, class T> float run(K kern, T *buf, int blocks, int threads) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, buf, (decltype(+*buf))0x9E3779B97F4A7C15ull);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, buf, (decltype(+*buf))0x9E3779B97F4A7C15ull); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
    }
    return best;
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int CU = p.multiProcessorCount, blocks = CU * 8, threads = 256;      // 8 waves per SIMD
    uint64_t *buf; hipMalloc(&buf, (size_t)blocks * threads * 8);
    const double waves_per_simd_instr = (double)blocks * threads / 64 / (CU * 4) * ITERS * 8;
    auto cyc = [&](float ms, double instr_per_iter) { return ms * 1e-3 * 2.1e9 / (waves_per_simd_instr * instr_per_iter); };
    printf("cycles per wave-instruction per SIMD at 2.1 GHz, 8 waves per SIMD\n");
    printf("  v_lshrrev_b64 (+ v_or pair)      : %.2f per triple\n", cyc(run(mb_shr64, buf, blocks, threads), 1));
    printf("  v_lshl_add_u64                   : %.2f\n", cyc(run(mb_add64, buf, blocks, threads), 1));
    printf("  v_add_co_u32 + v_addc_co_u32     : %.2f per pair\n", cyc(run(mb_addc, buf, blocks, threads), 1));
    printf("  v_and_b32 + v_add_u32            : %.2f per pair\n", cyc(run(mb_and32, (uint32_t *)buf, blocks, threads), 1));
    return 0;
}
