// Development: the shipped 9 x 29-bit Montgomery multiplier (field29.cuh, 81 + 81 v_mad_u64_u32) against a variant whose
// a * b half is a 3-way Karatsuba over blocks of three limbs (6 block products = 54 multiply-adds instead of 81, paid for
// with 18 limb additions and 64-bit column additions / subtractions).  Both compute the same value (checked on device).
// Build: hipcc -O3 --offload-arch=gfx950 -I mira_amd/csrc tools/microbench_mul.hip -o tools/microbench_mul
#include <hip/hip_runtime.h>

#include <cstdio>

#include "curve29.cuh"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// 3 x 3 limbs -> 5 columns, accumulated into c[0..4]
__device__ __forceinline__ void blk(uint64_t *c, const uint32_t *a, const uint32_t *b) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) c[i + j] += (uint64_t)a[i] * b[j];
}
template <class F> __device__ __forceinline__ Fe29<F> f29_mul_k3(const Fe29<F> &a, const Fe29<F> &b) {
    uint32_t a01[3], a02[3], a12[3], b01[3], b02[3], b12[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        a01[i] = a.l[i] + a.l[3 + i]; a02[i] = a.l[i] + a.l[6 + i]; a12[i] = a.l[3 + i] + a.l[6 + i];
        b01[i] = b.l[i] + b.l[3 + i]; b02[i] = b.l[i] + b.l[6 + i]; b12[i] = b.l[3 + i] + b.l[6 + i];
    }
    uint64_t d0[5] = {0, 0, 0, 0, 0}, d1[5] = {0, 0, 0, 0, 0}, d2[5] = {0, 0, 0, 0, 0};
    blk(d0, a.l, b.l); blk(d1, a.l + 3, b.l + 3); blk(d2, a.l + 6, b.l + 6);
    uint64_t c[18];
#pragma unroll
    for (int k = 0; k < 18; k++) c[k] = 0;
    blk(c + 3, a01, b01); blk(c + 6, a02, b02); blk(c + 9, a12, b12);        // the three cross sums, in place
#pragma unroll
    for (int k = 0; k < 5; k++) {
        c[k] += d0[k];
        c[3 + k] -= d0[k] + d1[k];
        c[6 + k] += d1[k] - d0[k] - d2[k];
        c[9 + k] -= d1[k] + d2[k];
        c[12 + k] += d2[k];
    }
#pragma unroll
    for (int k = 0; k < 9; k++) {
        uint32_t m = ((uint32_t)c[k] * F::N0) & M29;
#pragma unroll
        for (int j = 0; j < 9; j++) c[k + j] += (uint64_t)m * F::P[j];
        c[k + 1] += c[k] >> 29;
    }
    Fe29<F> r;
#pragma unroll
    for (int i = 9; i < 17; i++) {
        r.l[i - 9] = (uint32_t)c[i] & M29;
        c[i + 1] += c[i] >> 29;
    }
    r.l[8] = (uint32_t)c[17];
    return r;
}

constexpr int FE_ITERS = 512;
template <class F, int V> __global__ void mb_mul(unsigned char *out) {
    Fe29<F> x = f29_one<F>(), y = f29_one<F>();
    x.l[0] += threadIdx.x; y.l[1] += blockIdx.x;
    for (int i = 0; i < FE_ITERS; i++) {
        if (V == 0) { x = f29_mul(x, y); y = f29_mul(y, x); }
        else { x = f29_mul_k3(x, y); y = f29_mul_k3(y, x); }
    }
    f29_store_raw(out + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 36, f29_add(x, y));
}
template <class F> __global__ void mb_check(uint32_t *bad) {
    Fe29<F> x = f29_one<F>(), y = f29_one<F>();
    x.l[0] += threadIdx.x * 977u + 1; y.l[1] += blockIdx.x * 31u + 5; y.l[7] ^= threadIdx.x;
    for (int i = 0; i < 64; i++) {
        const Fe29<F> p = f29_mul(x, y), q = f29_mul_k3(x, y);
        for (int k = 0; k < 9; k++) if (p.l[k] != q.l[k]) atomicAdd(bad, 1u);
        x = f29_add(p, y); y = p;                      // carried sums: limbs < 2^29 + 8 on both sides, as in the curve formulas
    }
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
    const int CU = prop.multiProcessorCount;
    void *buf; CK(hipMalloc(&buf, (size_t)1 << 28));
    uint32_t *bad; CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
    mb_check<Fq29><<<64, 64>>>(bad); mb_check<Fr29><<<64, 64>>>(bad);
    uint32_t h = 1; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
    printf("karatsuba variant vs shipped multiplier: %u mismatching limbs (8192 lanes x 64 products x 2 fields)\n", h);
    for (int wps : {2, 3, 4, 8}) {
        int blocks = CU * wps * 2, threads = 128;
        double muls = (double)blocks * threads * FE_ITERS * 2;
        float t0 = time_kernel([&] { mb_mul<Fq29, 0><<<blocks, threads>>>((unsigned char *)buf); });
        float t1 = time_kernel([&] { mb_mul<Fq29, 1><<<blocks, threads>>>((unsigned char *)buf); });
        printf("%d waves/SIMD: f29_mul %7.2f G/s (%.0f cyc)   3-way Karatsuba %7.2f G/s (%.0f cyc)\n", wps, muls / t0 / 1e6, 2.4e9 * CU * 4 / (muls / 64 / (t0 / 1e3)),
               muls / t1 / 1e6, 2.4e9 * CU * 4 / (muls / 64 / (t1 / 1e3)));
    }
    return 0;
}
