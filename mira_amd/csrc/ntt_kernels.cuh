// Radix-2 NTT over bn256::Fr for gfx950.  Replaces best_fft (reference src/fft.rs:51-115,
// 118-155): natural order in, natural order out, out[k] = sum_j a[j] * omega^(j k).
//
// The reference walks log2(n) butterfly layers over the whole array.  Here a transform of up to
// 4096 points (128 KiB as two 16-byte limb planes) lives in one CU's LDS for all of its layers,
// and larger n = n1 * n2 uses the four-step split: n2 column transforms of length n1, one
// multiplication by omega^(i2 k1), n1 row transforms of length n2 -- two trips through HBM
// instead of log2(n).  The field results are canonical, so every element is bit-identical to
// the reference's whatever order the butterflies run in.
#pragma once
#include "field.cuh"

static constexpr int NTT_MAX_LOG_LINE = 12;   // 4096 elements * 32 B = 128 KiB of the 160 KiB LDS

// table[j] = base^(j * stride), j < count   (base, table entries in Montgomery form)
template <class FP>
KERNEL void k_pow_table(const unsigned char *__restrict__ base, uint64_t stride, uint32_t count,
                        const unsigned char *__restrict__ scale, unsigned char *__restrict__ table) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    Fe<FP> v = fe_pow_u64(fe_load<FP>(base), (uint64_t)j * stride);
    if (scale) v = fe_mul(v, fe_load<FP>(scale));
    fe_store(table + (size_t)j * 32, v);
}

struct NttPass {
    uint32_t log_len;          // line length N = 1 << log_len (<= 4096)
    uint32_t nlines;           // number of lines = grid size
    uint64_t in_line_stride;   // element index of line l, element q: l * in_line_stride + q * in_elem_stride
    uint64_t in_elem_stride;
    uint64_t out_line_stride;
    uint64_t out_elem_stride;
    uint32_t tw_shift;         // post-twiddle omega^(l * k): T_hi[e >> tw_shift] * T_lo[e & mask]; 0xFFFFFFFF = none
    uint32_t has_scale;        // multiply every output by *scale
};

// One workgroup per line.  blockDim.x = max(64, N/2) capped at 1024.  dynamic LDS = N * 32 B.
template <class FP>
KERNEL void k_ntt_lines(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst, NttPass ps,
                        const unsigned char *__restrict__ line_tw,   // omega_N^j, j < N/2
                        const unsigned char *__restrict__ t_lo, const unsigned char *__restrict__ t_hi,
                        const unsigned char *__restrict__ scale) {
    DYN_SHARED(U4, lds);
    const uint32_t N = 1u << ps.log_len;
    U4 *plane0 = lds, *plane1 = lds + N;
    // XCD-aware line order: workgroups b and b+8 share an XCD (and its L2); give each XCD a
    // contiguous range of lines so neighbouring strided lines (which share 128-B lines) meet in
    // one L2.  Speed only; any mapping is correct.
    uint32_t line = blockIdx.x;
    if ((ps.nlines & 7u) == 0) line = (blockIdx.x & 7u) * (ps.nlines >> 3) + (blockIdx.x >> 3);

    const unsigned char *in = src + (size_t)line * ps.in_line_stride * 32;
    for (uint32_t q = threadIdx.x; q < N; q += blockDim.x) {
        const U4 *g = reinterpret_cast<const U4 *>(in + (size_t)q * ps.in_elem_stride * 32);
        uint32_t r = ps.log_len ? (__brev(q) >> (32 - ps.log_len)) : 0;
        plane0[r] = g[0];
        plane1[r] = g[1];
    }
    __syncthreads();
    for (uint32_t s = 0; s < ps.log_len; s++) {
        const uint32_t half = 1u << s;
        for (uint32_t bf = threadIdx.x; bf < N / 2; bf += blockDim.x) {
            const uint32_t j = bf & (half - 1);
            const uint32_t i0 = ((bf >> s) << (s + 1)) + j, i1 = i0 + half;
            U4 a0 = plane0[i0], a1 = plane1[i0], b0 = plane0[i1], b1 = plane1[i1];
            Fe<FP> u, v;
            u.l[0] = a0.x; u.l[1] = a0.y; u.l[2] = a0.z; u.l[3] = a0.w; u.l[4] = a1.x; u.l[5] = a1.y; u.l[6] = a1.z; u.l[7] = a1.w;
            v.l[0] = b0.x; v.l[1] = b0.y; v.l[2] = b0.z; v.l[3] = b0.w; v.l[4] = b1.x; v.l[5] = b1.y; v.l[6] = b1.z; v.l[7] = b1.w;
            if (s != 0) v = fe_mul(v, fe_load<FP>(line_tw + (size_t)(j << (ps.log_len - 1 - s)) * 32));
            Fe<FP> hi = fe_add(u, v), lo = fe_sub(u, v);
            plane0[i0] = U4{hi.l[0], hi.l[1], hi.l[2], hi.l[3]};
            plane1[i0] = U4{hi.l[4], hi.l[5], hi.l[6], hi.l[7]};
            plane0[i1] = U4{lo.l[0], lo.l[1], lo.l[2], lo.l[3]};
            plane1[i1] = U4{lo.l[4], lo.l[5], lo.l[6], lo.l[7]};
        }
        __syncthreads();
    }
    unsigned char *out = dst + (size_t)line * ps.out_line_stride * 32;
    const uint32_t lo_mask = ps.tw_shift == 0xFFFFFFFFu ? 0 : ((1u << ps.tw_shift) - 1u);
    for (uint32_t k = threadIdx.x; k < N; k += blockDim.x) {
        U4 a0 = plane0[k], a1 = plane1[k];
        Fe<FP> v;
        v.l[0] = a0.x; v.l[1] = a0.y; v.l[2] = a0.z; v.l[3] = a0.w; v.l[4] = a1.x; v.l[5] = a1.y; v.l[6] = a1.z; v.l[7] = a1.w;
        if (ps.tw_shift != 0xFFFFFFFFu) {
            uint64_t e = (uint64_t)line * k;
            Fe<FP> tw = fe_mul(fe_load<FP>(t_hi + (size_t)(e >> ps.tw_shift) * 32), fe_load<FP>(t_lo + (size_t)(e & lo_mask) * 32));
            v = fe_mul(v, tw);
        }
        if (ps.has_scale) v = fe_mul(v, fe_load<FP>(scale));
        fe_store(out + (size_t)k * ps.out_elem_stride * 32, v);
    }
}

// a[i] *= powers[i % 3 - 1] for i % 3 != 0   (distribute_powers_zeta, reference src/fft.rs:205-226)
template <class FP>
KERNEL void k_distribute_powers(unsigned char *__restrict__ a, uint64_t n, const unsigned char *__restrict__ powers2) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r = (uint32_t)(i % 3);
    if (r == 0) return;
    Fe<FP> v = fe_mul(fe_load<FP>(a + i * 32), fe_load<FP>(powers2 + (size_t)(r - 1) * 32));
    fe_store(a + i * 32, v);
}
