// Fixed-base ("window table") mode of the MSM, sized for 288 GB of HBM3E.
//
// The commitment key is immutable and reused by every fold step (reference
// src/ivc/public_params.rs:50), so the engine may spend memory on it once: for a window width c it
// keeps W = ceil(256 / c) tables T_w[i] = 2^(c w) * P_i (affine, 64 B, resident layout).  Then
//     sum_i k_i P_i = sum_i sum_w d_{i,w} T_w[i]
// needs ONE set of 2^(c-1) buckets instead of one per window.  With a single set the bucket
// reduction stops growing with W, so c can rise from 16 to 20: 13 mixed additions per pair
// instead of 16 (k_accumulate is 72 % of the step and ALU-bound), no host Horner, and the window
// sums come back as 64 plain partial sums.  Cost: W x the key size in HBM (13 x 64 B per point:
// 3.3 GiB at 2^22, 52 GiB at 2^26) and 12 x 20 doublings per point at registration.
//
// With 2^19 buckets a per-tile LDS histogram over the raw digits no longer fits; the LDS-staged
// sort (sort_kernels.cuh) partitions by the top 9 bits first, counts the buckets from that
// coarse-sorted stream (a tile then spans only ~2000 of them) and places by bucket.
#pragma once
#include "curve29.cuh"

// Window width of table mode, per key: 20 bits (13 tables, 2^19 buckets) or 22 bits (12 tables, 2^21
// buckets: one addition per pair less, four times the buckets to reduce -- pays from ~2^24 pairs).
// Either way the level-1 sort partitions into 512 coarse bins (1024 or 4096 buckets each, the most a
// level-2 tile ranks in LDS).
struct TableCfg {
    uint32_t c, W, B, fine_bits;
};
static inline TableCfg table_cfg(uint32_t c) { return c == 22 ? TableCfg{22, 12, 1u << 21, 12} : TableCfg{20, 13, 1u << 19, 10}; }
static constexpr uint32_t TABLE_CB = 512;                // coarse bins
static constexpr uint32_t TABLE_SUMS = 64;               // partial sums returned to the host

// a^(P-2) on loose values (any input bound <= 12: every intermediate is a product < 2 P)
template <class F> HD Fe29<F> f29_inv(const Fe29<F> &a) {
    Fe29<F> acc = f29_one<F>(), base = f29_mul(a, f29_one<F>());   // bring the bound down to < 2
    // exponent P - 2, little-endian 29-bit limbs of P (P[0] is odd and > 2: no borrow)
#pragma unroll 1
    for (int i = 0; i < 9; i++) {
        uint32_t e = F::P[i] - (i == 0 ? 2u : 0u);
        const int bits = (i < 8) ? 29 : 22;              // P < 2^254 = 2^(8*29 + 22)
#pragma unroll 1
        for (int k = 0; k < bits; k++) {
            if ((e >> k) & 1) acc = f29_mul(acc, base);
            base = f29_sqr(base);
        }
    }
    return acc;
}

// dst[i] = 2^c * src[i]  (affine resident layout in and out; identity stays identity)
template <class F>
KERNEL void __launch_bounds__(64) k_table_step(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst, uint64_t n, uint32_t c) {
    using S = typename F::Sat;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Aff29<F> p = aff29_load<F>(src + i * 64, false);
    if (aff29_is_identity(p)) {
        fe_store(dst + i * 64, fe_zero<S>()); fe_store(dst + i * 64 + 32, fe_zero<S>());
        return;
    }
    Xyzz29<F> q = xyzz29_double_affine(p);
    for (uint32_t k = 1; k < c; k++) q = xyzz29_double(q);
    Fe<S> x = fe_zero<S>(), y = fe_zero<S>();
    if (!xyzz29_is_identity(q)) {
        Fe29<F> zi = f29_inv(q.zzz);                      // 1 / ZZZ
        Fe29<F> zzi = f29_sqr(f29_mul(zi, q.zz));         // (ZZ / ZZZ)^2 = 1 / ZZ
        x = reduce_once(f29_pack(f29_mul(q.x, zzi)));
        y = reduce_once(f29_pack(f29_mul(q.y, zi)));
    }
    fe_store(dst + i * 64, x);
    fe_store(dst + i * 64 + 32, y);
}

// scalar -> W signed c-bit digits (c = 20 or 22), window-major int32
template <class FS>
KERNEL void k_digits32(const unsigned char *__restrict__ scalars, uint32_t n, uint32_t TABLE_C, uint32_t TABLE_W, int32_t *__restrict__ digits) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<FS> s = fe_from_mont(fe_load<FS>(scalars + (size_t)i * 32));
    const uint32_t mask = (1u << TABLE_C) - 1u, half = 1u << (TABLE_C - 1);
    uint32_t carry = 0;
    for (uint32_t w = 0; w < TABLE_W; w++) {
        uint32_t raw = (s.l[0] & mask) + carry;
#pragma unroll
        for (int k = 0; k < 7; k++) s.l[k] = (s.l[k] >> TABLE_C) | (s.l[k + 1] << (32 - TABLE_C));
        s.l[7] >>= TABLE_C;
        int32_t d;
        if (raw >= half) { d = (int32_t)raw - (int32_t)(1u << TABLE_C); carry = 1; }
        else { d = (int32_t)raw; carry = 0; }
        digits[(size_t)w * n + i] = d;
    }
}

// coarse-bin histogram over the digits: grid = (ntiles, TABLE_W); LDS = TABLE_CB counters
KERNEL void k_thist_coarse(const int32_t *__restrict__ digits, uint32_t n, uint32_t tile, uint32_t TABLE_FINE_BITS, uint32_t *__restrict__ counts) {
    __shared__ uint32_t bins[TABLE_CB];
    const uint32_t w = blockIdx.y;
    for (uint32_t b = threadIdx.x; b < TABLE_CB; b += blockDim.x) bins[b] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * tile, end = (base + tile < n) ? base + tile : n;
    const int32_t *dw = digits + (size_t)w * n;
    for (uint32_t i = base + threadIdx.x; i < end; i += blockDim.x) {
        int32_t d = dw[i];
        if (d != 0) atomicAdd(&bins[((uint32_t)(d < 0 ? -d : d) - 1) >> TABLE_FINE_BITS], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < TABLE_CB; b += blockDim.x) {
        uint32_t cnt = bins[b];
        if (cnt) atomicAdd(&counts[b], cnt);
    }
}
