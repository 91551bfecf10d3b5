// Development probe (not shipped): what ONE wave on an otherwise idle SIMD pays for a dependent chain -- the regime of the
// MSM's tail kernels (fix-up trees, bucket reduction) -- and what the cross-lane primitives of a limb-parallel field
// multiplier would cost.  Cycles from s_memtime (shader clock), nanoseconds from wall_clock64 (100 MHz): their ratio is the
// clock the chip actually runs these light kernels at.
// Build: hipcc -O3 --offload-arch=gfx950 -I mira_amd/csrc tools/latency_probe.hip -o tools/latency_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "quad29.cuh"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 2048;

struct Stamp { uint64_t cycles, ns; };
__device__ __forceinline__ uint64_t now_cycles() { return __builtin_readcyclecounter(); }
__device__ __forceinline__ uint64_t now_wall() { return wall_clock64(); }

// 1 dependent v_mad_u64_u32 chain / 4 / 8 independent chains, one wave
template <int CHAINS> __global__ void p_mad(uint64_t *out, Stamp *st, uint32_t m) {
    uint64_t x[CHAINS];
    for (int k = 0; k < CHAINS; k++) x[k] = threadIdx.x + k;
    const uint64_t c0 = now_cycles(), w0 = now_wall();
    for (int i = 0; i < ITERS; i++)
#pragma unroll
        for (int k = 0; k < CHAINS; k++) x[k] = (uint64_t)(uint32_t)x[k] * m + x[k];
    const uint64_t c1 = now_cycles(), w1 = now_wall();
    uint64_t s = 0;
    for (int k = 0; k < CHAINS; k++) s ^= x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *st = Stamp{c1 - c0, (w1 - w0) * 10};
}
// dependent chain of DPP row shifts + add (the carry / column exchange step of a limb-parallel multiplier)
__global__ void p_dpp(uint32_t *out, Stamp *st) {
    uint32_t x = threadIdx.x * 2654435761u;
    const uint64_t c0 = now_cycles(), w0 = now_wall();
    for (int i = 0; i < ITERS; i++) x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111 /* row_shr:1 */, 0xF, 0xF, true);
    const uint64_t c1 = now_cycles(), w1 = now_wall();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) *st = Stamp{c1 - c0, (w1 - w0) * 10};
}
// dependent chain of ds_swizzle broadcasts within 16-lane rows (and_mask 0x10, or_mask k: the per-row broadcast of one
// limb -- gfx9 DPP has no row_share) + add
__global__ void p_swizzle(uint32_t *out, Stamp *st) {
    uint32_t x = threadIdx.x * 2654435761u;
    const uint64_t c0 = now_cycles(), w0 = now_wall();
    for (int i = 0; i < ITERS; i++) x += (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, 0x10 | (3 << 5));
    const uint64_t c1 = now_cycles(), w1 = now_wall();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) *st = Stamp{c1 - c0, (w1 - w0) * 10};
}
__global__ void p_bpermute(uint32_t *out, Stamp *st) {
    uint32_t x = threadIdx.x * 2654435761u;
    const uint64_t c0 = now_cycles(), w0 = now_wall();
    for (int i = 0; i < ITERS; i++) x += (uint32_t)__builtin_amdgcn_ds_bpermute((int)((threadIdx.x & 48u) | 5u) << 2, (int)x);
    const uint64_t c1 = now_cycles(), w1 = now_wall();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) *st = Stamp{c1 - c0, (w1 - w0) * 10};
}
// dependent chain of field multiplications (the shipped 9 x 29-bit multiplier), `waves` waves per workgroup on one CU
__global__ void p_mul(uint32_t *out, Stamp *st, int iters) {
    Fe29<Fq29> a = f29_one<Fq29>(), b = f29_one<Fq29>();
    a.l[0] += threadIdx.x; b.l[1] += 3;
    const uint64_t c0 = now_cycles(), w0 = now_wall();
    for (int i = 0; i < iters; i++) a = f29_mul(a, b);
    const uint64_t c1 = now_cycles(), w1 = now_wall();
    uint32_t s = 0;
    for (int k = 0; k < 9; k++) s ^= a.l[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *st = Stamp{c1 - c0, (w1 - w0) * 10};
}
// dependent chain of quad-cooperative general additions / doublings and of single-lane general additions
template <int MODE> __global__ void p_add(uint32_t *out, Stamp *st, const unsigned char *pts, int iters) {
    Xyzz29<Fq29> acc = xyzz29_load<Fq29>(pts), q = xyzz29_load<Fq29>(pts + XYZZ29_BYTES);
    const uint64_t c0 = now_cycles(), w0 = now_wall();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) xyzz29_add_quad(acc, q);
        else if (MODE == 1) acc = xyzz29_double_quad(acc);
        else xyzz29_add(acc, q);
    }
    const uint64_t c1 = now_cycles(), w1 = now_wall();
    uint32_t s = 0;
    for (int k = 0; k < 9; k++) s ^= acc.x.l[k] ^ acc.zzz.l[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *st = Stamp{c1 - c0, (w1 - w0) * 10};
}

int main() {
    uint64_t *out;
    Stamp *st;
    unsigned char *pts;
    CK(hipMalloc(&out, 1 << 22));
    CK(hipMalloc(&st, sizeof(Stamp)));
    CK(hipMalloc(&pts, 2 * XYZZ29_BYTES));
    // two points of the curve in raw XYZZ form: G = (1, 2) and 2 G, Z = 1 (x * 2^261 mod p: take f29_one multiples)
    {
        std::vector<uint32_t> h(72, 0);
        auto put = [&](int slot, const uint32_t *limbs) { for (int k = 0; k < 9; k++) h[slot * 9 + k] = limbs[k]; };
        uint32_t one[9], two[9];
        for (int k = 0; k < 9; k++) { one[k] = Fq29::ONE[k]; two[k] = 2 * Fq29::ONE[k]; }
        put(0, one); put(1, two); put(2, one); put(3, one);          // G
        // a second point: (x, y) with x = 2 is not on the curve, but the probe only needs the arithmetic's latency: the formulas do not branch on curve membership
        put(4, two); put(5, one); put(6, one); put(7, one);
        CK(hipMemcpy(pts, h.data(), 288, hipMemcpyHostToDevice));
    }
    Stamp h;
    auto report = [&](const char *name, double per, const char *unit) {
        (void)hipMemcpy(&h, st, sizeof h, hipMemcpyDeviceToHost);
        printf("%-58s %9.1f cycles per %s   %8.1f ns   clock %.2f GHz\n", name, (double)h.cycles / per, unit, (double)h.ns / per, (double)h.cycles / (double)h.ns);
    };
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(p_mad<1>, 1, 64, 0, 0, out, st, 12345u); CK(hipDeviceSynchronize()); report("v_mad_u64_u32, 1 dependent chain, 1 wave", ITERS, "mad");
        hipLaunchKernelGGL(p_mad<4>, 1, 64, 0, 0, out, st, 12345u); CK(hipDeviceSynchronize()); report("v_mad_u64_u32, 4 independent chains, 1 wave", ITERS * 4, "mad");
        hipLaunchKernelGGL(p_mad<8>, 1, 64, 0, 0, out, st, 12345u); CK(hipDeviceSynchronize()); report("v_mad_u64_u32, 8 independent chains, 1 wave", ITERS * 8, "mad");
        hipLaunchKernelGGL(p_mad<8>, 1, 256, 0, 0, out, st, 12345u); CK(hipDeviceSynchronize()); report("v_mad_u64_u32, 8 chains, 4 waves (1 per SIMD)", ITERS * 8, "mad");
        hipLaunchKernelGGL(p_mad<8>, 1, 512, 0, 0, out, st, 12345u); CK(hipDeviceSynchronize()); report("v_mad_u64_u32, 8 chains, 8 waves (2 per SIMD)", ITERS * 8, "mad");
        hipLaunchKernelGGL(p_dpp, 1, 64, 0, 0, (uint32_t *)out, st); CK(hipDeviceSynchronize()); report("v_mov_dpp row_shr:1 + v_add, dependent", ITERS, "step");
        hipLaunchKernelGGL(p_swizzle, 1, 64, 0, 0, (uint32_t *)out, st); CK(hipDeviceSynchronize()); report("ds_swizzle (row broadcast) + v_add, dependent", ITERS, "step");
        hipLaunchKernelGGL(p_bpermute, 1, 64, 0, 0, (uint32_t *)out, st); CK(hipDeviceSynchronize()); report("ds_bpermute + v_add, dependent", ITERS, "step");
        hipLaunchKernelGGL(p_mul, 1, 64, 0, 0, (uint32_t *)out, st, 512); CK(hipDeviceSynchronize()); report("f29_mul dependent chain, 1 wave", 512, "mul");
        hipLaunchKernelGGL(p_mul, 1, 512, 0, 0, (uint32_t *)out, st, 512); CK(hipDeviceSynchronize()); report("f29_mul dependent chain, 2 waves per SIMD", 512, "mul");
        hipLaunchKernelGGL(p_mul, 1024, 256, 0, 0, (uint32_t *)out, st, 512); CK(hipDeviceSynchronize()); report("f29_mul dependent chain, whole chip, 4 waves per SIMD", 512, "mul");
        hipLaunchKernelGGL(p_add<0>, 1, 64, 0, 0, (uint32_t *)out, st, (const unsigned char *)pts, 256); CK(hipDeviceSynchronize()); report("xyzz29_add_quad dependent chain, 1 wave", 256, "add");
        hipLaunchKernelGGL(p_add<0>, 1, 512, 0, 0, (uint32_t *)out, st, (const unsigned char *)pts, 256); CK(hipDeviceSynchronize()); report("xyzz29_add_quad dependent chain, 2 waves per SIMD", 256, "add");
        hipLaunchKernelGGL(p_add<1>, 1, 64, 0, 0, (uint32_t *)out, st, (const unsigned char *)pts, 256); CK(hipDeviceSynchronize()); report("xyzz29_double_quad dependent chain, 1 wave", 256, "dbl");
        hipLaunchKernelGGL(p_add<2>, 1, 64, 0, 0, (uint32_t *)out, st, (const unsigned char *)pts, 256); CK(hipDeviceSynchronize()); report("xyzz29_add (single lane) dependent chain, 1 wave", 256, "add");
        hipLaunchKernelGGL(p_add<2>, 1024, 256, 0, 0, (uint32_t *)out, st, (const unsigned char *)pts, 64); CK(hipDeviceSynchronize()); report("xyzz29_add (single lane), whole chip, 4 waves per SIMD", 64, "add");
        printf("\n");
    }
    return 0;
}
