// Development probe (not shipped): what the host side of one commit costs -- the field and point operations of the Horner
// epilogue (host_field.hpp) on this box's CPU, and the two ways a few kilobytes of results can reach the host behind a kernel:
// hipMemcpyAsync into pinned memory + hipStreamSynchronize, or the kernel writing into mapped pinned memory with a flag the
// host spins on.
// Build: hipcc -O3 --offload-arch=gfx950 -I mira_amd/csrc tools/host_epilogue_probe.hip -o tools/host_epilogue_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

#include "host_field.hpp"
using namespace hostf;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_produce(uint64_t *out, uint32_t words, uint64_t stamp) {
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) out[i] = stamp + i;
}
// results straight into mapped host memory, then the flag (system-scope release)
__global__ void k_produce_flag(uint64_t *out, uint32_t words, uint64_t stamp, volatile uint64_t *flag) {
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) out[i] = stamp + i;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { __atomic_store_n((uint64_t *)flag, stamp, __ATOMIC_RELEASE); }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    {
        HXyzz<FqP> g; g.x = from_u64<FqP>(1); g.y = from_u64<FqP>(2); g.zz = one<FqP>(); g.zzz = one<FqP>();
        HXyzz<FqP> acc = dbl_pt(g), q = add_pt(acc, g);
        const double t0 = now_us();
        for (int i = 0; i < 100000; i++) acc = dbl_pt(acc);
        const double t1 = now_us();
        for (int i = 0; i < 100000; i++) acc = add_pt(acc, q);
        const double t2 = now_us();
        HFe<FqP> a = acc.x;
        for (int i = 0; i < 1000000; i++) a = mul(a, acc.y);
        const double t3 = now_us();
        uint64_t out[8];
        for (int i = 0; i < 1000; i++) { to_affine(acc, out); acc.x.l[0] ^= out[0] & 1; }
        const double t4 = now_us();
        printf("host: dbl %.1f ns  add %.1f ns  mul %.2f ns  to_affine %.2f us   (%llx %llx)\n", (t1 - t0) * 1e3 / 1e5, (t2 - t1) * 1e3 / 1e5, (t3 - t2) * 1e3 / 1e6, (t4 - t3) / 1e3,
               (unsigned long long)a.l[0], (unsigned long long)out[0]);
    }
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const uint32_t words = 30 * 16 + 128;                       // 30 points + statistics
    uint64_t *d_buf, *h_pinned, *h_mapped, *d_mapped, *h_flag, *d_flag;
    CK(hipMalloc(&d_buf, words * 8));
    CK(hipHostMalloc(&h_pinned, words * 8, hipHostMallocDefault));
    CK(hipHostMalloc(&h_mapped, words * 8, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc(&h_flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void **)&d_mapped, h_mapped, 0));
    CK(hipHostGetDevicePointer((void **)&d_flag, h_flag, 0));
    *h_flag = 0;
    const int reps = 2000;
    for (int mode = 0; mode < 3; mode++) {
        double sum = 0;
        for (int r = 0; r < reps + 100; r++) {
            const uint64_t stamp = ((uint64_t)(mode + 1) << 40) + (uint64_t)r * 4096 + 1;
            const double t0 = now_us();
            if (mode == 0) {
                hipLaunchKernelGGL(k_produce, 1, 256, 0, st, d_buf, words, stamp);
                CK(hipMemcpyAsync(h_pinned, d_buf, words * 8, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
                if (h_pinned[words - 1] != stamp + words - 1) { printf("mode 0: wrong data\n"); return 1; }
            } else if (mode == 1) {                              // mapped memory, still hipStreamSynchronize
                hipLaunchKernelGGL(k_produce, 1, 256, 0, st, d_mapped, words, stamp);
                CK(hipStreamSynchronize(st));
                if (((volatile uint64_t *)h_mapped)[words - 1] != stamp + words - 1) { printf("mode 1: wrong data\n"); return 1; }
            } else {                                             // mapped memory + flag, host spins
                hipLaunchKernelGGL(k_produce_flag, 1, 256, 0, st, d_mapped, words, stamp, d_flag);
                while (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != stamp) { }
                if (((volatile uint64_t *)h_mapped)[words - 1] != stamp + words - 1) { printf("mode 2: wrong data\n"); return 1; }
            }
            const double t1 = now_us();
            if (r >= 100) sum += t1 - t0;
        }
        CK(hipStreamSynchronize(st));
        printf("%s: %.2f us per launch + results on the host\n", mode == 0 ? "kernel + hipMemcpyAsync(pinned) + hipStreamSynchronize" : mode == 1 ? "kernel writes mapped memory + hipStreamSynchronize" : "kernel writes mapped memory + flag, host spins", sum / reps);
    }
    return 0;
}
