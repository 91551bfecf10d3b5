// Instruction-rate and field-op microbenchmarks for gfx950 (development tool; not shipped in
// libmira_gpu.so).  Build: hipcc -O3 --offload-arch=gfx950 -I mira_amd/csrc tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "curve.cuh"
#include "curve29.cuh"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

__global__ void mb_mad64(uint64_t *out, uint32_t a, uint32_t b) {
    uint64_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    uint32_t m = a + threadIdx.x;
    for (int i = 0; i < ITERS; i++) {
        x0 = (uint64_t)(uint32_t)x0 * m + x0; x1 = (uint64_t)(uint32_t)x1 * m + x1; x2 = (uint64_t)(uint32_t)x2 * m + x2; x3 = (uint64_t)(uint32_t)x3 * m + x3;
        x4 = (uint64_t)(uint32_t)x4 * m + x4; x5 = (uint64_t)(uint32_t)x5 * m + x5; x6 = (uint64_t)(uint32_t)x6 * m + x6; x7 = (uint64_t)(uint32_t)x7 * m + x7;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7 ^ b;
}
__global__ void mb_mullo(uint32_t *out, uint32_t a) {
    uint32_t x0 = threadIdx.x | 1, x1 = x0 + 2, x2 = x0 + 4, x3 = x0 + 6, x4 = x0 + 8, x5 = x0 + 10, x6 = x0 + 12, x7 = x0 + 14;
    uint32_t m = a | 1;
    for (int i = 0; i < ITERS; i++) {
        x0 *= m; x1 *= m; x2 *= m; x3 *= m; x4 *= m; x5 *= m; x6 *= m; x7 *= m;
        m += x0 & 2;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
__global__ void mb_mulhi(uint32_t *out, uint32_t a) {
    uint32_t x0 = ~threadIdx.x, x1 = x0 - 2, x2 = x0 - 4, x3 = x0 - 6, x4 = x0 - 8, x5 = x0 - 10, x6 = x0 - 12, x7 = x0 - 14;
    uint32_t m = ~a;
    for (int i = 0; i < ITERS; i++) {
        x0 = __umulhi(x0, m) | 0x80000000u; x1 = __umulhi(x1, m) | 0x80000000u; x2 = __umulhi(x2, m) | 0x80000000u; x3 = __umulhi(x3, m) | 0x80000000u;
        x4 = __umulhi(x4, m) | 0x80000000u; x5 = __umulhi(x5, m) | 0x80000000u; x6 = __umulhi(x6, m) | 0x80000000u; x7 = __umulhi(x7, m) | 0x80000000u;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
__global__ void mb_mul24(uint32_t *out, uint32_t a) {
    uint32_t x0 = threadIdx.x | 1, x1 = x0 + 2, x2 = x0 + 4, x3 = x0 + 6, x4 = x0 + 8, x5 = x0 + 10, x6 = x0 + 12, x7 = x0 + 14;
    uint32_t m = a | 1;
    for (int i = 0; i < ITERS; i++) {
        x0 = __umul24(x0, m) + 1; x1 = __umul24(x1, m) + 1; x2 = __umul24(x2, m) + 1; x3 = __umul24(x3, m) + 1;
        x4 = __umul24(x4, m) + 1; x5 = __umul24(x5, m) + 1; x6 = __umul24(x6, m) + 1; x7 = __umul24(x7, m) + 1;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
__global__ void mb_add32(uint32_t *out, uint32_t a) {
    uint32_t x0 = threadIdx.x, x1 = x0 + 2, x2 = x0 + 4, x3 = x0 + 6, x4 = x0 + 8, x5 = x0 + 10, x6 = x0 + 12, x7 = x0 + 14;
    for (int i = 0; i < ITERS; i++) {
        x0 = (x0 + a) ^ x1; x1 = (x1 + a) ^ x2; x2 = (x2 + a) ^ x3; x3 = (x3 + a) ^ x4;
        x4 = (x4 + a) ^ x5; x5 = (x5 + a) ^ x6; x6 = (x6 + a) ^ x7; x7 = (x7 + a) ^ x0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
__global__ void mb_dfma(double *out, double a) {
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; i++) {
        x0 = fma(x0, a, x0); x1 = fma(x1, a, x1); x2 = fma(x2, a, x2); x3 = fma(x3, a, x3);
        x4 = fma(x4, a, x4); x5 = fma(x5, a, x5); x6 = fma(x6, a, x6); x7 = fma(x7, a, x7);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void mb_ffma(float *out, float a) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; i++) {
        x0 = fmaf(x0, a, x0); x1 = fmaf(x1, a, x1); x2 = fmaf(x2, a, x2); x3 = fmaf(x3, a, x3);
        x4 = fmaf(x4, a, x4); x5 = fmaf(x5, a, x5); x6 = fmaf(x6, a, x6); x7 = fmaf(x7, a, x7);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
constexpr int FE_ITERS = 512;
template <class FP> __global__ void mb_femul(unsigned char *out) {
    Fe<FP> x = fe_one<FP>(), y = fe_one<FP>();
    x.l[0] += threadIdx.x; y.l[1] += blockIdx.x;
    for (int i = 0; i < FE_ITERS; i++) { x = fe_mul(x, y); y = fe_mul(y, x); }
    fe_store(out + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 32, fe_add(x, y));
}
template <class FP> __global__ void __launch_bounds__(128) mb_madd(unsigned char *out) {
    Aff<FP> p; p.x = fe_one<FP>(); p.y = fe_one<FP>(); p.x.l[0] += threadIdx.x; p.y.l[0] += 3;
    Xyzz<FP> acc = xyzz_from_affine(p);
    p.x.l[1] += 7;
    for (int i = 0; i < FE_ITERS; i++) { xyzz_add_affine(acc, p); p.x.l[2] += 1; }
    xyzz_store(out + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 128, acc);
}

template <class F> __global__ void mb_f29mul(unsigned char *out) {
    Fe29<F> x = f29_one<F>(), y = f29_one<F>();
    x.l[0] += threadIdx.x; y.l[1] += blockIdx.x;
    for (int i = 0; i < FE_ITERS; i++) { x = f29_mul(x, y); y = f29_mul(y, x); }
    f29_store_raw(out + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 36, f29_add(x, y));
}
template <class F> __global__ void mb_f29sqr(unsigned char *out) {
    Fe29<F> x = f29_one<F>(), y = f29_one<F>();
    x.l[0] += threadIdx.x; y.l[1] += blockIdx.x;
    for (int i = 0; i < FE_ITERS; i++) { x = f29_sqr(x); y = f29_sqr(y); }
    f29_store_raw(out + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 36, f29_add(x, y));
}
template <class F, int LB> __global__ void __launch_bounds__(128, LB) mb_madd29(unsigned char *out, const unsigned char *pts) {
    Aff29<F> p = aff29_load<F>(pts + (threadIdx.x & 7) * 64, false);
    Xyzz29<F> acc = xyzz29_identity<F>();
    xyzz29_add_affine(acc, p);
    p.x.l[1] += 7;
    for (int i = 0; i < FE_ITERS; i++) { xyzz29_add_affine(acc, p); p.x.l[2] = (p.x.l[2] + 1) & M29; }
    xyzz29_store(out + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * XYZZ29_BYTES, acc);
}

// one full XYZZ add per lane, executed `iters` times (iters = 1: the code runs once, cold)
template <class F> __global__ void __launch_bounds__(128) mb_fulladd(unsigned char *buf, int iters) {
    size_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Xyzz29<F> a = xyzz29_load<F>(buf + (t % 4096) * XYZZ29_BYTES), b = xyzz29_load<F>(buf + ((t + 7) % 4096) * XYZZ29_BYTES);
    for (int i = 0; i < iters; i++) xyzz29_add(a, b);
    xyzz29_store(buf + (4096 + t) * XYZZ29_BYTES, a);
}
template <class F> __global__ void mb_init_points(unsigned char *buf, const unsigned char *pts) {
    size_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Aff29<F> p = aff29_load<F>(pts + (t & 7) * 64, false);
    p.x.l[1] = (p.x.l[1] + (uint32_t)t * 977u) & M29;
    Xyzz29<F> acc = xyzz29_identity<F>();
    xyzz29_add_affine(acc, p);
    p.x.l[2] = (p.x.l[2] + 5) & M29;
    xyzz29_add_affine(acc, p);
    xyzz29_store(buf + t * XYZZ29_BYTES, acc);
}

template <class F> float time_kernel(F launch, int reps = 5) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    const int CU = prop.multiProcessorCount;
    void *buf; CK(hipMalloc(&buf, (size_t)1 << 28));
    struct { const char *name; int wpsimd; } cfgs[] = {{"1 wave/SIMD", 1}, {"2 waves/SIMD", 2}, {"4 waves/SIMD", 4}, {"8 waves/SIMD", 8}};
    for (auto &cfg : cfgs) {
        int blocks = CU * cfg.wpsimd, threads = 256;   // 4 waves per block = 1 per SIMD
        double lanes = (double)blocks * threads;
        double ops = lanes * ITERS * 8;
        float t;
        printf("-- %s (%d blocks x %d)\n", cfg.name, blocks, threads);
        t = time_kernel([&] { mb_mad64<<<blocks, threads>>>((uint64_t *)buf, 12345u, 1u); });
        printf("  v_mad_u64_u32 : %8.1f Gop/s  (%.2f cyc/wave-instr/SIMD @2.4GHz)\n", ops / t / 1e6, 2.4e9 * CU * 4 / (ops / 64 / (t / 1e3)));
        t = time_kernel([&] { mb_mullo<<<blocks, threads>>>((uint32_t *)buf, 12345u); });
        printf("  v_mul_lo_u32  : %8.1f Gop/s  (%.2f cyc)\n", ops / t / 1e6, 2.4e9 * CU * 4 / (ops / 64 / (t / 1e3)));
        t = time_kernel([&] { mb_mulhi<<<blocks, threads>>>((uint32_t *)buf, 12345u); });
        printf("  v_mul_hi_u32  : %8.1f Gop/s  (%.2f cyc, incl. v_or)\n", ops / t / 1e6, 2.4e9 * CU * 4 / (ops / 64 / (t / 1e3)));
        t = time_kernel([&] { mb_mul24<<<blocks, threads>>>((uint32_t *)buf, 12345u); });
        printf("  v_mad_u32_u24 : %8.1f Gop/s  (%.2f cyc)\n", ops / t / 1e6, 2.4e9 * CU * 4 / (ops / 64 / (t / 1e3)));
        t = time_kernel([&] { mb_add32<<<blocks, threads>>>((uint32_t *)buf, 12345u); });
        printf("  add+xor pair  : %8.1f Gpair/s (%.2f cyc/pair)\n", ops / t / 1e6, 2.4e9 * CU * 4 / (ops / 64 / (t / 1e3)));
        t = time_kernel([&] { mb_dfma<<<blocks, threads>>>((double *)buf, 1.0000001); });
        printf("  v_fma_f64     : %8.1f Gop/s  (%.2f cyc)\n", ops / t / 1e6, 2.4e9 * CU * 4 / (ops / 64 / (t / 1e3)));
        t = time_kernel([&] { mb_ffma<<<blocks, threads>>>((float *)buf, 1.0000001f); });
        printf("  v_fma_f32     : %8.1f Gop/s  (%.2f cyc)\n", ops / t / 1e6, 2.4e9 * CU * 4 / (ops / 64 / (t / 1e3)));
    }
    for (int wps : {1, 2, 3, 4}) {
        int blocks = CU * wps * 2, threads = 128;
        double muls = (double)blocks * threads * FE_ITERS * 2;
        float t = time_kernel([&] { mb_femul<FqP><<<blocks, threads>>>((unsigned char *)buf); });
        printf("fe_mul<Fq> %d waves/SIMD: %7.2f G modmul/s  (%.0f cyc per wave-modmul per SIMD)\n", wps, muls / t / 1e6, 2.4e9 * CU * 4 / (muls / 64 / (t / 1e3)));
        t = time_kernel([&] { mb_femul<FrP><<<blocks, threads>>>((unsigned char *)buf); });
        printf("fe_mul<Fr> %d waves/SIMD: %7.2f G modmul/s\n", wps, muls / t / 1e6);
        double adds = (double)blocks * threads * FE_ITERS;
        t = time_kernel([&] { mb_madd<FqP><<<blocks, threads>>>((unsigned char *)buf); });
        printf("xyzz madd<Fq> %d waves/SIMD: %7.2f G add/s  (= %.2f G modmul-equiv/s at 10 per add)\n", wps, adds / t / 1e6, adds * 10 / t / 1e6);
    }
    CK(hipMemset(buf, 1, 4096));
    for (int wps : {1, 2, 3, 4, 6, 8}) {
        int blocks = CU * wps * 2, threads = 128;
        double muls = (double)blocks * threads * FE_ITERS * 2;
        float t = time_kernel([&] { mb_f29mul<Fq29><<<blocks, threads>>>((unsigned char *)buf + 4096); });
        printf("f29_mul<Fq> %d waves/SIMD: %7.2f G modmul/s  (%.0f cyc per wave-modmul per SIMD)\n", wps, muls / t / 1e6, 2.4e9 * CU * 4 / (muls / 64 / (t / 1e3)));
        t = time_kernel([&] { mb_f29sqr<Fq29><<<blocks, threads>>>((unsigned char *)buf + 4096); });
        printf("f29_sqr<Fq> %d waves/SIMD: %7.2f G modsqr/s\n", wps, muls / t / 1e6);
        double adds = (double)blocks * threads * FE_ITERS;
        t = time_kernel([&] { mb_madd29<Fq29, 2><<<blocks, threads>>>((unsigned char *)buf + 4096, (const unsigned char *)buf); });
        printf("xyzz29 madd<Fq> min2w %d waves/SIMD: %7.2f G add/s  (= %.2f G modmul-equiv/s at 10 per add)\n", wps, adds / t / 1e6, adds * 10 / t / 1e6);
        t = time_kernel([&] { mb_madd29<Fq29, 3><<<blocks, threads>>>((unsigned char *)buf + 4096, (const unsigned char *)buf); });
        printf("xyzz29 madd<Fq> min3w %d waves/SIMD: %7.2f G add/s\n", wps, adds / t / 1e6);
        t = time_kernel([&] { mb_madd29<Fq29, 4><<<blocks, threads>>>((unsigned char *)buf + 4096, (const unsigned char *)buf); });
        printf("xyzz29 madd<Fq> min4w %d waves/SIMD: %7.2f G add/s\n", wps, adds / t / 1e6);
    }
    {
        float t = time_kernel([&] { mb_f29mul<Fq29><<<1, 64>>>((unsigned char *)buf + 4096); });
        printf("single wave: %.3f us per dependent f29_mul\n", t * 1e3 / (FE_ITERS * 2));
        t = time_kernel([&] { mb_madd29<Fq29, 2><<<1, 64>>>((unsigned char *)buf + 4096, (const unsigned char *)buf); });
        printf("single wave: %.3f us per dependent xyzz29 madd\n", t * 1e3 / FE_ITERS);
    }
    {
        unsigned char *pb = (unsigned char *)buf + (1 << 20);
        mb_init_points<Fq29><<<32, 128>>>(pb, (const unsigned char *)buf);
        CK(hipDeviceSynchronize());
        for (int waves : {1, 80, 640, 2560, 3072}) {
            for (int iters : {1, 2, 8}) {
                int blocks = (waves + 1) / 2;
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                float best = 1e9;
                for (int rep = 0; rep < 3; rep++) {
                    // a different kernel in between evicts the instruction cache, as in the real pipeline
                    mb_f29mul<Fq29><<<CU * 8, 128>>>((unsigned char *)buf + 4096);
                    hipEventRecord(e0);
                    mb_fulladd<Fq29><<<blocks, 128>>>(pb, iters);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    best = ms < best ? ms : best;
                }
                printf("full add x%d per lane, %4d waves: %8.1f us\n", iters, waves, best * 1e3);
            }
        }
    }
    // single-wave latency of one fe_mul chain
    {
        float t = time_kernel([&] { mb_femul<FqP><<<1, 64>>>((unsigned char *)buf); });
        printf("single wave: %.3f us per dependent fe_mul\n", t * 1e3 / (FE_ITERS * 2));
        t = time_kernel([&] { mb_madd<FqP><<<1, 64>>>((unsigned char *)buf); });
        printf("single wave: %.3f us per dependent xyzz madd\n", t * 1e3 / FE_ITERS);
    }
    return 0;
}
