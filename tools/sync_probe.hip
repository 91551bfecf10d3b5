// Development: what the end of a commit costs on the host.  A tiny kernel, then (a) hipStreamSynchronize, (b) a copy of 4 KiB to
// pinned memory + hipStreamSynchronize (the shipped epilogue), (c) the kernel writing its 4 KiB into pinned memory itself and
// raising a flag there that the host spins on.   hipcc -O3 --offload-arch=gfx950 tools/sync_probe.hip -o tools/sync_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_work(uint32_t *out, uint32_t v, int spin) {
    uint32_t x = v;
    for (int i = 0; i < spin; i++) x = x * 1664525u + 1013904223u;
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
}
__global__ void k_work_flag(uint32_t *out, uint32_t v, int spin, volatile uint32_t *flag, uint32_t *counter) {
    uint32_t x = v;
    for (int i = 0; i < spin; i++) x = x * 1664525u + 1013904223u;
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t done = atomicAdd(counter, 1u) + 1;
        if (done == gridDim.x) { *counter = 0; __threadfence_system(); *flag = v; }
    }
}
template <class Fn> static double med_us(Fn fn) {
    std::vector<double> t;
    for (int i = 0; i < 300; i++) {
        auto t0 = std::chrono::steady_clock::now();
        fn(i + 1);
        t.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    std::sort(t.begin(), t.end());
    return t[150];
}
int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    uint32_t *d, *h, *counter;
    volatile uint32_t *flag;
    CK(hipMalloc(&d, 4096)); CK(hipMalloc(&counter, 4)); CK(hipMemset(counter, 0, 4));
    CK(hipHostMalloc(&h, 4096 + 64, hipHostMallocDefault));
    flag = h + 1024;
    *flag = 0;
    for (int spin : {0, 20000}) {                            // an empty kernel, and one of ~50 us
        double a = med_us([&](int) { hipLaunchKernelGGL(k_work, 16, 64, 0, st, d, 1u, spin); (void)hipStreamSynchronize(st); });
        double b = med_us([&](int) { hipLaunchKernelGGL(k_work, 16, 64, 0, st, d, 1u, spin); (void)hipMemcpyAsync(h, d, 4096, hipMemcpyDeviceToHost, st); (void)hipStreamSynchronize(st); });
        double c = med_us([&](int i) {
            hipLaunchKernelGGL(k_work_flag, 16, 64, 0, st, h, (uint32_t)i + 7u, spin, flag, counter);
            while (*flag != (uint32_t)i + 7u) { }
        });
        (void)hipStreamSynchronize(st);
        printf("kernel loop %5d: launch + sync %.1f us | launch + copy 4 KiB + sync %.1f us | kernel writes pinned memory, host spins on a flag %.1f us\n", spin, a, b, c);
    }
    return 0;
}
