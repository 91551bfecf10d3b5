// Development probe: does a host -> device copy run beside a kernel that fills every wave slot?
// pinned source, pageable source, and hipHostRegister of the caller's pageable buffer.
// Build: hipcc -O3 --offload-arch=gfx950 tools/overlap_probe.hip -o tools/overlap_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void __launch_bounds__(256) k_busy(uint64_t *out, int iters) {
    uint64_t a = threadIdx.x + 1, b = blockIdx.x + 3, c = 7;
    for (int i = 0; i < iters; i++) { a = a * b + c; b = b * c + a; c = c * a + b; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c;
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t bytes = 128u << 20;
    void *pageable = malloc(bytes), *pinned = nullptr, *dev = nullptr;
    uint64_t *out = nullptr;
    memset(pageable, 1, bytes);
    CK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
    memset(pinned, 2, bytes);
    CK(hipMalloc(&dev, bytes));
    const int blocks = 256 * 8;                                     // 8 workgroups of 256 per CU: every wave slot taken
    CK(hipMalloc(&out, (size_t)blocks * 256 * 8));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    int iters = 200000;
    for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(256), 0, s1, out, iters); CK(hipStreamSynchronize(s1)); }
    double t0 = now(); hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(256), 0, s1, out, iters); CK(hipStreamSynchronize(s1)); double tk = now() - t0;
    t0 = now(); CK(hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, s2)); CK(hipStreamSynchronize(s2)); double tp = now() - t0;
    t0 = now(); CK(hipMemcpyAsync(dev, pageable, bytes, hipMemcpyHostToDevice, s2)); CK(hipStreamSynchronize(s2)); double tg = now() - t0;
    printf("kernel alone %.3f ms; copy alone: pinned %.3f ms, pageable %.3f ms\n", tk, tp, tg);
    for (int which = 0; which < 2; which++) {
        const void *src = which ? pageable : pinned;
        t0 = now();
        hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(256), 0, s1, out, iters);
        double t1 = now();
        CK(hipMemcpyAsync(dev, src, bytes, hipMemcpyHostToDevice, s2));
        double t2 = now();
        CK(hipStreamSynchronize(s2));
        double t3 = now();
        CK(hipStreamSynchronize(s1));
        double t4 = now();
        printf("%s copy beside the kernel: memcpyAsync call returned after %.3f ms, copy done at %.3f ms, kernel done at %.3f ms (serial would be %.3f)\n",
               which ? "pageable" : "pinned", t2 - t1, t3 - t0, t4 - t0, tk + (which ? tg : tp));
    }
    // 8 chunks of 16 MiB from pinned memory beside the kernel
    t0 = now();
    hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(256), 0, s1, out, iters);
    for (int k = 0; k < 8; k++) CK(hipMemcpyAsync((char *)dev + (size_t)k * (bytes / 8), (char *)pinned + (size_t)k * (bytes / 8), bytes / 8, hipMemcpyHostToDevice, s2));
    CK(hipStreamSynchronize(s2)); double tc = now() - t0;
    CK(hipStreamSynchronize(s1));
    printf("8 pinned chunks beside the kernel: copies done at %.3f ms, kernel done at %.3f ms\n", tc, now() - t0);
    // the pipeline of msm_host.cuh: copy chunk k (pageable, copy stream), event, compute stream waits, kernels of chunk k
    {
        hipEvent_t ev[8];
        for (int k = 0; k < 8; k++) CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
        const int it2 = iters / 28;                                  // ~1.5 ms per chunk
        hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(256), 0, s1, out, it2); CK(hipStreamSynchronize(s1));
        t0 = now(); hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(256), 0, s1, out, it2); CK(hipStreamSynchronize(s1)); double t1k = now() - t0;
        for (int which = 0; which < 2; which++) {
            const char *src = (const char *)(which ? pageable : pinned);
            t0 = now();
            for (int k = 0; k < 8; k++) {
                CK(hipMemcpyAsync((char *)dev + (size_t)k * (bytes / 8), src + (size_t)k * (bytes / 8), bytes / 8, hipMemcpyHostToDevice, s2));
                CK(hipEventRecord(ev[k], s2));
                CK(hipStreamWaitEvent(s1, ev[k], 0));
                hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(256), 0, s1, out, it2);
            }
            CK(hipStreamSynchronize(s1));
            printf("pipeline of 8 x (16 MiB %s copy -> %.3f ms kernel): %.3f ms (kernels alone %.3f, copies alone ~2.4)\n", which ? "pageable" : "pinned", t1k, now() - t0, 8 * t1k);
        }
    }
    // pin the caller's buffer in place
    for (int r = 0; r < 3; r++) {
        t0 = now(); CK(hipHostRegister(pageable, bytes, hipHostRegisterDefault)); double tr = now() - t0;
        t0 = now(); CK(hipMemcpyAsync(dev, pageable, bytes, hipMemcpyHostToDevice, s2)); CK(hipStreamSynchronize(s2)); double tcopy = now() - t0;
        t0 = now(); CK(hipHostUnregister(pageable)); double tu = now() - t0;
        printf("hipHostRegister 128 MiB %.3f ms, copy %.3f ms, unregister %.3f ms\n", tr, tcopy, tu);
    }
    // host memcpy into pinned memory, one thread
    t0 = now(); memcpy(pinned, pageable, bytes); double tm = now() - t0;
    printf("host memcpy pageable -> pinned, one thread: %.3f ms (%.1f GB/s)\n", tm, bytes / tm / 1e6);
    return 0;
}
