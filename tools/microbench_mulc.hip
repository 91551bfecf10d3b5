// Development (round 3, measured, NOT adopted): products with a CONSTANT factor -- every product of the NTT has a table
// entry for a factor.  The shipped 9 x 29-bit Montgomery multiplier (81 + 81 v_mad_u64_u32) against Shoup's form: w < P plain,
// wq = floor(w 2^261 / P) precomputed,
//     q = floor(a wq / 2^261)    from columns 7 .. 16 of a wq (53 multiply-adds; q is the true quotient or up to 2 below),
//     r = a w - q P              taken mod 2^261: the low nine columns of a w + q (2^261 - P) (45 + 45 multiply-adds), r < 3 P.
// 143 multiply-adds, no quotient digits, one carry chain of nine columns.  Checked against the Montgomery product of the same
// value (a * rho * 2^-261 with rho = w 2^261 mod P).  In isolation 15 - 25 % faster at every occupancy, with two independent
// chains per lane and with one; inside k_ntt_wave (tables of (w, wq) pairs, 80 B per entry, twice the LDS twiddle reads) the
// 2^24 transform went 2.27 -> 2.36 - 2.45 ms: see DESIGN.md section 5 and profiles/r03_d_shoup.txt.
// Build: hipcc -O3 --offload-arch=gfx950 -I mira_amd/csrc tools/microbench_mulc.hip -o tools/microbench_mulc
#include <hip/hip_runtime.h>

#include <cstdio>

#include "curve29.cuh"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <class F> struct ShoupK;
template <> struct ShoupK<Fq29> {
    static constexpr uint32_t PINV261[9] = {0x1b799c77u, 0x016fc3e8u, 0x0d654d9eu, 0x030535c2u, 0x0257f612u, 0x1a17f3e6u, 0x0e509d40u, 0x090dceeeu, 0x100a85ddu};   // P^-1 mod 2^261
    static constexpr uint32_t PNEG261[9] = {0x078302b9u, 0x1efb9f49u, 0x038d5cb0u, 0x1d2add2fu, 0x0a7a2687u, 0x1d24bf3fu, 0x1f591ebeu, 0x11a3d9cbu, 0x1fcf9bb1u};   // 2^261 - P
};
template <> struct ShoupK<Fr29> {
    static constexpr uint32_t PINV261[9] = {0x10000001u, 0x08f05360u, 0x05bb930fu, 0x12f36967u, 0x1dc6e9a7u, 0x13ebb37cu, 0x19347195u, 0x1c5e4f97u, 0x0d8c07d0u};
    static constexpr uint32_t PNEG261[9] = {0x0fffffffu, 0x00f05360u, 0x11a3dbafu, 0x182f6f0cu, 0x0a7a2d7cu, 0x1d24bf3fu, 0x1f591ebeu, 0x11a3d9cbu, 0x1fcf9bb1u};
};
template <class F> struct Shoup29 {
    uint32_t w[9], wq[9];
};
// a: value < 2^261, limbs up to 2.5 * 2^30 + 8 (what the uncarried butterflies hand over); w, wq: limbs < 2^29.  Result: a w - q P in [0, 3 P), limbs < 2^29.
template <class F> __device__ __forceinline__ Fe29<F> f29_mulc(const Fe29<F> &a, const Shoup29<F> &k) {
    uint64_t h[10];                                      // columns 7 .. 16 of a * wq
#pragma unroll
    for (int c = 0; c < 10; c++) h[c] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++)
            if (i + j >= 7) h[i + j - 7] += (uint64_t)a.l[i] * k.wq[j];
    uint64_t c[9];                                       // the low half of a w: 45 independent multiply-adds to lay over the carry chain of q
#pragma unroll
    for (int t = 0; t < 9; t++) c[t] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j + i < 9; j++) c[i + j] += (uint64_t)a.l[i] * k.w[j];
    uint32_t q[9];
    h[1] += h[0] >> 29;
    h[2] += h[1] >> 29;
#pragma unroll
    for (int t = 2; t < 9; t++) {
        q[t - 2] = (uint32_t)h[t] & M29;
        h[t + 1] += h[t] >> 29;
    }
    q[7] = (uint32_t)h[9] & M29;
    q[8] = (uint32_t)(h[9] >> 29);
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j + i < 9; j++) c[i + j] += (uint64_t)q[i] * ShoupK<F>::PNEG261[j];
    Fe29<F> r;
#pragma unroll
    for (int t = 0; t < 8; t++) {
        r.l[t] = (uint32_t)c[t] & M29;
        c[t + 1] += c[t] >> 29;
    }
    r.l[8] = (uint32_t)c[8] & M29;
    return r;
}
// rho = w 2^261 mod P, CANONICAL  ->  w canonical and wq = floor(w 2^261 / P) = (w 2^261 - rho) / P = - rho P^-1 mod 2^261
template <class F> __device__ __forceinline__ Shoup29<F> f29_shoup_pair(const Fe29<F> &rho) {
    Shoup29<F> k;
    const Fe29<F> w = f29_unpack_canonical<F>(reduce_once(f29_pack(f29_redc(rho))));
    uint64_t c[9];
#pragma unroll
    for (int t = 0; t < 9; t++) { c[t] = 0; k.w[t] = w.l[t]; }
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j + i < 9; j++) c[i + j] += (uint64_t)rho.l[i] * ShoupK<F>::PINV261[j];
    uint32_t t9[9], borrow = 0;
#pragma unroll
    for (int t = 0; t < 8; t++) { t9[t] = (uint32_t)c[t] & M29; c[t + 1] += c[t] >> 29; }
    t9[8] = (uint32_t)c[8] & M29;
#pragma unroll
    for (int t = 0; t < 9; t++) {                        // negate mod 2^261
        k.wq[t] = (0u - t9[t] - borrow) & M29;
        borrow = (t9[t] | borrow) != 0 ? 1u : 0u;
    }
    return k;
}

constexpr int FE_ITERS = 512;
template <class F, int V> __global__ void mb_mul(unsigned char *out) {
    Fe29<F> x = f29_one<F>(), y = f29_one<F>(), w = f29_one<F>();
    Shoup29<F> k;
    for (int i = 0; i < 9; i++) { k.w[i] = F::ONE[i] + (i == 2 ? threadIdx.x : 0); k.wq[i] = F::ONE[i] + (i == 3 ? blockIdx.x : 0); }
    x.l[0] += threadIdx.x; y.l[1] += blockIdx.x; w.l[2] += threadIdx.x;
    for (int i = 0; i < FE_ITERS; i++) {
        if (V == 0) { x = f29_mul(x, w); y = f29_mul(y, w); }                 // two independent chains
        else if (V == 1) { x = f29_mulc(x, k); y = f29_mulc(y, k); }
        else if (V == 2) { x = f29_mul(x, w); x = f29_mul(x, w); }            // one chain: every product waits for the one before
        else { x = f29_mulc(x, k); x = f29_mulc(x, k); }
    }
    f29_store_raw(out + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 36, f29_add(x, y));
}
template <class F> __global__ void mb_check(uint32_t *bad) {
    using S = typename F::Sat;
    Fe29<F> rho = f29_one<F>(), a = f29_one<F>();
    rho.l[0] += threadIdx.x * 977u + 1; rho.l[5] ^= blockIdx.x * 131u; a.l[1] += blockIdx.x * 31u + 5; a.l[7] ^= threadIdx.x;
    for (int i = 0; i < 64; i++) {
        const Fe29<F> rc = f29_unpack_canonical<F>(reduce_once(reduce_once(f29_pack(f29_mul(rho, f29_one<F>())))));     // canonical rho
        const Shoup29<F> k = f29_shoup_pair(rc);
        Fe29<F> au = a;                                   // uncarried operand as the butterflies hand it over: limbs up to ~2^31.3
        for (int k = 0; k < 8; k++) au.l[k] += (a.l[(k + 3) % 9] & 3u) << 29;
        const Fe29<F> ac = f29_carry(au);
        const Fe<S> want = reduce_once(reduce_once(f29_pack(f29_mul(ac, rc))));           // a * w mod P, canonical
        const Fe29<F> r = f29_mulc(au, k);
        Fe<S> got = f29_pack(r);
        for (int t = 0; t < 3; t++) got = reduce_once(got);
        for (int k = 0; k < 8; k++) if (got.l[k] != want.l[k]) atomicAdd(bad, 1u);
        a = f29_add(f29_mul(a, rho), a); rho = f29_mul(rho, a);
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
    printf("Shoup product vs Montgomery product of the same value: %u mismatching limbs (8192 lanes x 64 products x 2 fields)\n", h);
    for (int wps : {2, 3, 4, 8}) {
        int blocks = CU * wps * 2, threads = 128;
        double muls = (double)blocks * threads * FE_ITERS * 2;
        float t0 = time_kernel([&] { mb_mul<Fr29, 0><<<blocks, threads>>>((unsigned char *)buf); });
        float t1 = time_kernel([&] { mb_mul<Fr29, 1><<<blocks, threads>>>((unsigned char *)buf); });
        float t2 = time_kernel([&] { mb_mul<Fr29, 2><<<blocks, threads>>>((unsigned char *)buf); });
        float t3 = time_kernel([&] { mb_mul<Fr29, 3><<<blocks, threads>>>((unsigned char *)buf); });
        printf("%d waves/SIMD: f29_mul (Montgomery) %7.2f G/s (%.0f cyc)   f29_mulc (Shoup) %7.2f G/s (%.0f cyc)   one dependent chain: %7.2f G/s, %7.2f G/s\n", wps, muls / t0 / 1e6, 2.4e9 * CU * 4 / (muls / 64 / (t0 / 1e3)),
               muls / t1 / 1e6, 2.4e9 * CU * 4 / (muls / 64 / (t1 / 1e3)), muls / t2 / 1e6, muls / t3 / 1e6);
    }
    return 0;
}
