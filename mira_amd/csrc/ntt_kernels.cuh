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
#include "field29.cuh"

static constexpr int NTT_MAX_LOG_LINE = 12;   // 4096 elements * 36 B = 144 KiB of the 160 KiB LDS
static constexpr int NTT_LDS_TW_LOG = 8;      // + the twiddles of the first 9 layers (256 * 36 B) -- see k_ntt_lines

// 9 raw limbs (padded to 48 B) per table entry: canonical value in R' = 2^261 Montgomery form, ready to be a
// multiplier operand.  Multiplying a reference-form element (x * 2^256) by such an entry with
// f29_mul (which divides by 2^261) leaves the product in the reference form again.
static constexpr int TW_BYTES = 48;   // 9 limbs padded to three 16-byte loads
template <class F> HD void tw_store(unsigned char *p, const Fe29<F> &v) {
    U4 *q = reinterpret_cast<U4 *>(p);
    q[0] = U4{v.l[0], v.l[1], v.l[2], v.l[3]};
    q[1] = U4{v.l[4], v.l[5], v.l[6], v.l[7]};
    q[2] = U4{v.l[8], 0, 0, 0};
}
template <class F> HD Fe29<F> tw_load(const unsigned char *p) {
    const U4 *q = reinterpret_cast<const U4 *>(p);
    U4 a = q[0], b = q[1], c = q[2];
    Fe29<F> r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w; r.l[8] = c.x;
    F29_SET(r, 1.0);
    return r;
}
// loose value < 2 P -> canonical saturated limbs of the same residue
template <class F> HD Fe<typename F::Sat> f29_canonical(const Fe29<F> &v) {
    F29_ASSERT(F29_GET(v) <= 2.0);
    return reduce_once(f29_pack(v));
}
// table[j] = base^(j * stride) * scale, j < count   (base, scale in reference form; table in R' form)
template <class F>
KERNEL void k_pow_table(const unsigned char *__restrict__ base, uint64_t stride, uint32_t count,
                        const unsigned char *__restrict__ scale, unsigned char *__restrict__ table) {
    using S = typename F::Sat;
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    Fe<S> v = fe_pow_u64(fe_load<S>(base), (uint64_t)j * stride);
    if (scale) v = fe_mul(v, fe_load<S>(scale));
    tw_store(table + (size_t)j * TW_BYTES, f29_unpack_canonical<F>(f29_canonical(f29_from_r256<F>(v))));
}

struct NttPass {
    uint32_t log_len;          // line length N = 1 << log_len (<= 4096)
    uint32_t nlines;           // number of lines
    // line l = q << split | r starts at element q * hi + r * lo; its elements are elem_stride apart
    uint32_t split;
    uint32_t tw_line_shift;    // post-twiddle exponent (l >> tw_line_shift) * k
    uint64_t in_hi, in_lo, in_elem_stride;
    uint64_t out_hi, out_lo, out_elem_stride;
    uint32_t tw_shift;         // post-twiddle w^e = T_hi[e >> tw_shift] * T_lo[e & mask]; 0xFFFFFFFF = none
    uint32_t reserved;
};

// One workgroup per line.  blockDim.x = max(64, N/2) capped at 1024.  dynamic LDS = N * 36 B + 16 + 256 * 36 B.
//
// Inside the line every value stays a loose 9 x 29-bit element (field29.cuh): a butterfly is one
// multiplication (whose result is < 2 P whatever its input), one carry-free addition and one
// biased subtraction; entering layer s all values are < (1 + 3 s) P, 37 P after 12 layers, well
// inside the multiplier's input budget.  Values are made canonical once, when they leave the line
// -- by the four-step twiddle product in pass 1, by the scale (ifft) or a multiplication by one
// in the last pass.
static constexpr int NTT_LDS_BYTES_PER_ELEM = 36;
template <class F> struct NttLds {
    U4 *p0, *p1;
    uint32_t *p2;
    DEV Fe29<F> load(uint32_t i, double bound) const {
        U4 a = p0[i], b = p1[i];
        Fe29<F> r;
        r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w; r.l[8] = p2[i];
        F29_SET(r, bound);
        (void)bound;
        return r;
    }
    DEV void store(uint32_t i, const Fe29<F> &v) const {
        p0[i] = U4{v.l[0], v.l[1], v.l[2], v.l[3]};
        p1[i] = U4{v.l[4], v.l[5], v.l[6], v.l[7]};
        p2[i] = v.l[8];
    }
};

template <class F>
KERNEL void __launch_bounds__(1024) k_ntt_lines(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst, NttPass ps,
                        const unsigned char *__restrict__ line_tw,   // omega_N^j, j < N/2, TW_BYTES each
                        const unsigned char *__restrict__ t_lo, const unsigned char *__restrict__ t_hi,
                        const unsigned char *__restrict__ scale) {   // multiplier-form scale, or one, for the last pass
    using S = typename F::Sat;
    DYN_SHARED(U4, lds);
    const uint32_t N = 1u << ps.log_len;
    NttLds<F> L{lds, lds + N, reinterpret_cast<uint32_t *>(lds + 2 * (size_t)N)};
    const uint32_t lo_mask = ps.tw_shift == 0xFFFFFFFFu ? 0 : ((1u << ps.tw_shift) - 1u);
    const uint32_t split_mask = (1u << ps.split) - 1u;
    // Twiddles of layers 0..tw_layers-1 (at most 256 values) sit in LDS behind the line.  Vector
    // memory operations retire in order, so a twiddle fetched from global memory in a butterfly
    // layer first waits for the whole prefetch of the next line issued before it; with the early
    // layers fed from LDS the prefetch has 9 of the 12 layers to land (measured: 2 % per pass).
    const uint32_t tw_layers = ps.log_len < NTT_LDS_TW_LOG + 1 ? ps.log_len : NTT_LDS_TW_LOG + 1;
    U4 *tbase = lds + 2 * (size_t)N + (N + 3) / 4;
    NttLds<F> T{tbase, tbase + (1u << NTT_LDS_TW_LOG), reinterpret_cast<uint32_t *>(tbase + 2 * (size_t)(1u << NTT_LDS_TW_LOG))};
    if (tw_layers >= 2) {
        const uint32_t cnt = 1u << (tw_layers - 1);      // omega_N^(j * N / (2 cnt)), j < cnt
        for (uint32_t j = threadIdx.x; j < cnt; j += blockDim.x)
            T.store(j, tw_load<F>(line_tw + (size_t)(j << (ps.log_len - tw_layers)) * TW_BYTES));
    }
    // Persistent workgroups: each walks lines blockIdx.x, blockIdx.x + gridDim.x, ...  The next
    // line's elements are fetched into registers before the butterfly layers of the current line
    // start, so the strided HBM gather is hidden behind ~100k cycles of arithmetic.
    // XCD-aware order: workgroups b and b+8 share an XCD (and its L2); each XCD gets a contiguous
    // range of lines so neighbouring strided lines (which share 128-B lines) meet in one L2.
    // Speed only; any mapping is correct.
    auto line_of = [&](uint32_t idx) {
        return ((ps.nlines & 7u) == 0) ? (idx & 7u) * (ps.nlines >> 3) + (idx >> 3) : idx;
    };
    constexpr int PRE = 4;                               // N / blockDim.x <= 4 (4096 points, 1024 lanes)
    Fe<S> pre[PRE];
    auto fetch = [&](uint32_t idx) {
        const uint32_t fl = line_of(idx);
        const unsigned char *in = src + ((size_t)(fl >> ps.split) * ps.in_hi + (size_t)(fl & split_mask) * ps.in_lo) * 32;
#pragma unroll
        for (int k = 0; k < PRE; k++) {
            uint32_t q = threadIdx.x + k * blockDim.x;
            if (q < N) pre[k] = fe_load<S>(in + (size_t)q * ps.in_elem_stride * 32);
        }
    };
    uint32_t idx = blockIdx.x;
    if (idx < ps.nlines) fetch(idx);
    for (; idx < ps.nlines; idx += gridDim.x) {
        const uint32_t line = line_of(idx);
#pragma unroll
        for (int k = 0; k < PRE; k++) {
            uint32_t q = threadIdx.x + k * blockDim.x;
            if (q < N) {
                uint32_t r = ps.log_len ? (__brev(q) >> (32 - ps.log_len)) : 0;
                L.store(r, f29_unpack_canonical<F>(pre[k]));
            }
        }
        __syncthreads();
        if (idx + gridDim.x < ps.nlines) fetch(idx + gridDim.x);
        for (uint32_t s = 0; s < ps.log_len; s++) {
            const uint32_t half = 1u << s;
            const double bound = 1.0 + 3.0 * s;
            for (uint32_t bf = threadIdx.x; bf < N / 2; bf += blockDim.x) {
                const uint32_t j = bf & (half - 1);
                const uint32_t i0 = ((bf >> s) << (s + 1)) + j, i1 = i0 + half;
                Fe29<F> u = L.load(i0, bound), v = L.load(i1, bound);
                if (s != 0) {
                    Fe29<F> tw = s < tw_layers ? T.load(j << (tw_layers - 1 - s), 1.0)
                                               : tw_load<F>(line_tw + (size_t)(j << (ps.log_len - 1 - s)) * TW_BYTES);
                    v = f29_mul(v, tw);
                }
                L.store(i0, f29_add(u, v));
                L.store(i1, f29_sub<3>(u, v));
            }
            __syncthreads();
        }
        unsigned char *out = dst + ((size_t)(line >> ps.split) * ps.out_hi + (size_t)(line & split_mask) * ps.out_lo) * 32;
        const double bound = 1.0 + 3.0 * ps.log_len;
        for (uint32_t k = threadIdx.x; k < N; k += blockDim.x) {
            Fe29<F> v = L.load(k, bound);
            if (ps.tw_shift != 0xFFFFFFFFu) {
                uint64_t e = (uint64_t)(line >> ps.tw_line_shift) * k;
                Fe29<F> tw = f29_mul(tw_load<F>(t_hi + (size_t)(e >> ps.tw_shift) * TW_BYTES), tw_load<F>(t_lo + (size_t)(e & lo_mask) * TW_BYTES));
                v = f29_mul(v, tw);
            } else {
                v = f29_mul(v, tw_load<F>(scale));
            }
            fe_store(out + (size_t)k * ps.out_elem_stride * 32, f29_canonical(v));
        }
        __syncthreads();                                 // LDS is rewritten by the next line
    }
}

// a[i] *= powers[i % 3 - 1] for i % 3 != 0   (distribute_powers_zeta, reference src/fft.rs:205-226)
template <class F>
KERNEL void k_distribute_powers(unsigned char *__restrict__ a, uint64_t n, const unsigned char *__restrict__ powers2) {
    using S = typename F::Sat;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r = (uint32_t)(i % 3);
    if (r == 0) return;
    Fe<S> v = fe_mul(fe_load<S>(a + i * 32), fe_load<S>(powers2 + (size_t)(r - 1) * 32));
    fe_store(a + i * 32, v);
}
