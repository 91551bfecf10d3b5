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
// the same for a value straight out of the multiplier (limbs masked: f29_pack_product)
template <class F> HD Fe<typename F::Sat> ntt_canonical_product(const Fe29<F> &v) {
    F29_ASSERT(F29_GET(v) <= 2.0);
    return reduce_once(f29_pack_product(v));
}
// an element as a pass reads it: canonical from the caller, or what the previous pass left (< 2 P, see NttwIo::finish)
template <class F> HD Fe29<F> ntt_unpack(const Fe<typename F::Sat> &s) {
    Fe29<F> r = f29_unpack<F>(s);
    F29_SET(r, 2.0);
    return r;
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

// full[line << log_len | k] = hi[e >> h] * lo[e & (2^h - 1)], e = line * k: the first post-twiddle of every
// element of a transform, so that the pass multiplies by ONE table entry (the product costs a
// multiplication per element in every transform; here once per table)
template <class F>
KERNEL void k_tw_full(const unsigned char *__restrict__ t_lo, const unsigned char *__restrict__ t_hi, uint32_t h, uint32_t log_len, uint64_t count,
                      unsigned char *__restrict__ full) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    const uint64_t e = (idx >> log_len) * (idx & (((uint64_t)1 << log_len) - 1));
    const Fe29<F> v = f29_mul(tw_load<F>(t_hi + (size_t)(e >> h) * TW_BYTES), tw_load<F>(t_lo + (size_t)(e & (((uint64_t)1 << h) - 1)) * TW_BYTES));
    // the product of two multiplier-form entries (x 2^261)(y 2^261) / 2^261 is in multiplier form again
    tw_store(full + (size_t)idx * TW_BYTES, f29_unpack_canonical<F>(f29_canonical(v)));
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
    uint32_t tw_single;        // 1: the exponent range fits T_lo alone (w^e = T_lo[e], no product); 2: T_lo holds the twiddle of EVERY element, T_lo[line << log_len | k]
    uint32_t coop;             // k_ntt_wave: bit 0 / bit 1 = the lines of a workgroup are adjacent in the input / output
    uint32_t prescaled;        // last pass only: the pass before left every element times 2^261 (times the ifft scale): finish with f29_redc, not a product
};

// Work distribution: the block-groups of a pass are cut into NTTW_RANGES contiguous ranges, one per XCD (workgroups b and
// b + 8 share an XCD and its L2: neighbouring strided lines meet there), each with a counter in global memory (a 128-byte
// line of its own).  A workgroup takes the next block-group of its range when it has finished one, and walks on through
// the other ranges when its own is empty.  (A static stride per workgroup was measured: the SIMD issues for the oldest
// wave first, so of the workgroups of a CU the one in wave slot 0 ran three times as fast as the one in slot 2 and left it
// to finish the pass alone.)
static constexpr uint32_t NTTW_RANGES = 8, NTTW_CTR_STRIDE = 32, NTTW_DONE = 0xFFFFFFFFu;
// thread 0 of a workgroup: the next block-group, or NTTW_DONE.  r = how many ranges this workgroup has left behind.
// The FIRST block-group of a workgroup costs no counter: workgroup b is at home in range b mod 8 and takes that range's
// block-group number b / 8; the counter of a range hands out what lies behind the first ones of its home workgroups (a
// transform of 2^20 points has hardly more block-groups than the grid has workgroups: with every first block-group
// drawn from a counter, 96 workgroups queued on one address before anything started, 0.14 -> 0.20 ms).
// Passes with fewer than NTTW_DYNAMIC_MIN block-groups per workgroup keep the static stride (block-groups b, b + grid, ...
// in the XCD-aware order): nothing to balance, and the counters only cost (2^21 points: 0.26 -> 0.28 ms with them).
static constexpr uint32_t NTTW_DYNAMIC_MIN = 4;
// the first block-group of this workgroup: a function of its number alone, so every lane may evaluate it (k_ntt_lines does).
// A grid never has more workgroups than the pass has block-groups, and under the counters a range holds at least half a grid of
// them: every workgroup has a first one.
DEV uint32_t nttw_first(uint32_t nbg) {
    if (nbg < NTTW_DYNAMIC_MIN * gridDim.x) {
        const uint32_t i = blockIdx.x;
        return i >= nbg ? NTTW_DONE : (nbg & 7u) == 0 ? (i & 7u) * (nbg >> 3) + (i >> 3) : i;
    }
    const uint32_t per = (nbg + NTTW_RANGES - 1) / NTTW_RANGES;
    const uint32_t x = blockIdx.x % NTTW_RANGES, w = blockIdx.x / NTTW_RANGES, lo = x * per, hi = lo + per < nbg ? lo + per : nbg;
    return lo + w < hi ? lo + w : NTTW_DONE;
}
// every further one (ONE lane of the workgroup calls this)
DEV uint32_t nttw_grab(uint32_t *ctr, uint32_t &r, uint32_t nbg) {
    if (nbg < NTTW_DYNAMIC_MIN * gridDim.x) {                      // r = the rounds of the stride behind this workgroup
        const uint32_t i = blockIdx.x + ++r * gridDim.x;
        if (i >= nbg) return NTTW_DONE;
        return (nbg & 7u) == 0 ? (i & 7u) * (nbg >> 3) + (i >> 3) : i;
    }
    const uint32_t per = (nbg + NTTW_RANGES - 1) / NTTW_RANGES;
    while (r < NTTW_RANGES) {
        const uint32_t x = (blockIdx.x + r) % NTTW_RANGES, lo = x * per, hi = lo + per < nbg ? lo + per : nbg;
        const uint32_t at_home = gridDim.x > x ? (gridDim.x - x + NTTW_RANGES - 1) / NTTW_RANGES : 0u;   // each took lo + its number
        if (lo + at_home < hi) {
            const uint32_t i = atomicAdd(&ctr[x * NTTW_CTR_STRIDE], 1u);
            if (lo + at_home + i < hi) return lo + at_home + i;
        }
        r++;
    }
    return NTTW_DONE;
}

// One workgroup per line.  blockDim.x = max(64, N/2) capped at 1024.  dynamic LDS = N * 36 B + 16 + 256 * 36 B.
//
// Inside the line every value stays a loose 9 x 29-bit element (field29.cuh): a butterfly is one
// multiplication (whose result is < 2 P whatever its input), one carry-free addition and one
// biased subtraction; entering layer s all values are < (2 + 3 s) P, 38 P after 12 layers, well
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

// COUNTERS: lines come from the per-XCD counters (the host asks for it where a workgroup has four lines or more: 2^25 and up);
// without, the static stride, and not a word of LDS or an instruction more than it needs -- the two-pass transforms of 2^9 ..
// 2^19 points are one line per workgroup and 20 us per pass (measured with the counter path's bookkeeping in: + 1 us per pass)
template <class F, bool COUNTERS>
KERNEL void __launch_bounds__(1024) k_ntt_lines(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst, NttPass ps,
                        const unsigned char *__restrict__ line_tw,   // omega_N^j, j < N/2, TW_BYTES each
                        const unsigned char *__restrict__ t_lo, const unsigned char *__restrict__ t_hi,
                        const unsigned char *__restrict__ scale,     // multiplier-form scale, or one, for the last pass
                        uint32_t *__restrict__ ctr) {                // NTTW_RANGES work counters of this launch, zero
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
    // Persistent workgroups.  Lines are handed out like the block-groups of k_ntt_wave (nttw_grab: a static stride while a
    // workgroup has fewer than four lines, per-XCD counters above: the workgroups of a CU run at different speeds, the SIMD
    // issues for its oldest wave first).  The next line's elements are fetched into registers before the butterfly layers
    // of the current line start, so the strided HBM gather is hidden behind ~100k cycles of arithmetic -- hence the line
    // after that is asked for beside that prefetch, a whole iteration before its number is needed.
    uint32_t *NEXT = reinterpret_cast<uint32_t *>(tbase + 2 * (size_t)(1u << NTT_LDS_TW_LOG) + (1u << NTT_LDS_TW_LOG) / 4);   // two words, by iteration parity
    constexpr int PRE = 4;                               // N / blockDim.x <= 4 (4096 points, 1024 lanes)
    Fe<S> pre[PRE];
    auto fetch = [&](uint32_t fl) {
        const unsigned char *in = src + ((size_t)(fl >> ps.split) * ps.in_hi + (size_t)(fl & split_mask) * ps.in_lo) * 32;
#pragma unroll
        for (int k = 0; k < PRE; k++) {
            uint32_t q = threadIdx.x + k * blockDim.x;
            if (q < N) pre[k] = fe_load<S>(in + (size_t)q * ps.in_elem_stride * 32);
        }
    };
    // every lane works out the workgroup's first line for itself (no counter is involved); lane 0 asks for the second one
    // beside the first line's loads and publishes it behind the first barrier
    uint32_t ranges_left = 0, par = 0, pend = NTTW_DONE;           // pend (thread 0): the line to fetch in the coming iteration
    uint32_t stride_i = blockIdx.x;                                // (static stride) lines stride_i, stride_i + grid, ... in the XCD-aware order
    auto stride_line = [&](uint32_t i) { return i >= ps.nlines ? NTTW_DONE : (ps.nlines & 7u) == 0 ? (i & 7u) * (ps.nlines >> 3) + (i >> 3) : i; };
    uint32_t line = COUNTERS ? nttw_first(ps.nlines) : stride_line(stride_i);
    if (line != NTTW_DONE) fetch(line);
    if (COUNTERS && threadIdx.x == 0 && line != NTTW_DONE) pend = nttw_grab(ctr, ranges_left, ps.nlines);
    while (line != NTTW_DONE) {
#pragma unroll
        for (int k = 0; k < PRE; k++) {
            uint32_t q = threadIdx.x + k * blockDim.x;
            if (q < N) {
                uint32_t r = ps.log_len ? (__brev(q) >> (32 - ps.log_len)) : 0;
                L.store(r, ntt_unpack<F>(pre[k]));
            }
        }
        uint32_t line_next;
        if constexpr (COUNTERS) {
            if (threadIdx.x == 0) NEXT[par] = pend;
            __syncthreads();
            line_next = NEXT[par];
        } else {
            __syncthreads();
            stride_i += gridDim.x;
            line_next = stride_line(stride_i);
        }
        if (line_next != NTTW_DONE) fetch(line_next);
        if (COUNTERS && threadIdx.x == 0) pend = line_next != NTTW_DONE ? nttw_grab(ctr, ranges_left, ps.nlines) : NTTW_DONE;   // answers beside the prefetch
        for (uint32_t s = 0; s < ps.log_len; s++) {
            const uint32_t half = 1u << s;
            const double bound = 2.0 + 3.0 * s;
            for (uint32_t bf = threadIdx.x; bf < N / 2; bf += blockDim.x) {
                const uint32_t j = bf & (half - 1);
                const uint32_t i0 = ((bf >> s) << (s + 1)) + j, i1 = i0 + half;
                Fe29<F> u = L.load(i0, bound), v = L.load(i1, bound);
                if (s != 0) {
                    Fe29<F> tw = s < tw_layers ? T.load(j << (tw_layers - 1 - s), 1.0)
                                               : tw_load<F>(line_tw + (size_t)(j << (ps.log_len - 1 - s)) * TW_BYTES);
                    v = f29_mul(v, tw);
                }
                // uncarried sums (field29.cuh, as in k_ntt_wave): v is a multiplier result or an unpacked value, so
                // limbs grow by < 2^30 per layer; every second layer carries (an even layer leaves limbs
                // < 1.5 * 2^30, which the next layer's product and the post-twiddle product accept)
                Fe29<F> a2 = f29_add_nc(u, v), b2 = f29_sub_nc<3>(u, v);
                if (s & 1u) { a2 = f29_carry(a2); b2 = f29_carry(b2); }
                L.store(i0, a2);
                L.store(i1, b2);
            }
            __syncthreads();
        }
        unsigned char *out = dst + ((size_t)(line >> ps.split) * ps.out_hi + (size_t)(line & split_mask) * ps.out_lo) * 32;
        const double bound = 2.0 + 3.0 * ps.log_len;
        for (uint32_t k = threadIdx.x; k < N; k += blockDim.x) {
            Fe29<F> v = L.load(k, bound);
            if (ps.tw_shift != 0xFFFFFFFFu) {
                uint64_t e = ps.tw_single == 2 ? (((uint64_t)line << ps.log_len) | k) : (uint64_t)(line >> ps.tw_line_shift) * k;
                Fe29<F> tw = ps.tw_single ? tw_load<F>(t_lo + (size_t)e * TW_BYTES)
                                          : f29_mul(tw_load<F>(t_hi + (size_t)(e >> ps.tw_shift) * TW_BYTES), tw_load<F>(t_lo + (size_t)(e & lo_mask) * TW_BYTES));
                v = f29_mul(v, tw);
            } else if (ps.prescaled) {
                v = f29_redc(v);
            } else {
                v = f29_mul(v, tw_load<F>(scale));
            }
            fe_store(out + (size_t)k * ps.out_elem_stride * 32, ntt_canonical_product(v));   // v: a product or a reduction in every branch
        }
        __syncthreads();                                 // LDS is rewritten by the next line
        line = line_next; par ^= 1u;
    }
}

// ------------------------------------------------------------------------------------------
// k_ntt_wave: lines of up to 256 points, one WAVE per 256 points, no workgroup barrier.
//
// k_ntt_lines above walks one layer at a time through LDS: 12 barriers per 4096-point line, every
// value through LDS twice per layer (measured: 47 % of wave cycles waiting, 59 % of LDS cycles bank
// conflicts).  Here a wave keeps 4 points per lane in registers and runs TWO layers per trip: the
// two index bits being paired are register bits, so both layers are plain register arithmetic; then
// the wave transposes its 256 points through a private LDS region (9 limb planes of 256 dwords) to
// bring the next two index bits into register position.  A 256-point line costs 3 transposes and no
// barrier; waves are independent, so three per SIMD hide each other's memory latency without a
// software prefetch.  Lines shorter than 256 points share a wave (2^(8-m) lines per wave).
//
// Position algebra (DIT on bit-reversed input): p in [0, 256) = (line-in-wave << m) | position.
// Round t owns layers 2t and 2t+1; its layout puts bits 2t, 2t+1 of p in the register index r:
//     p_t(lane, r) = (lane >> 2t) << (2t + 2) | r << 2t | lane & (4^t - 1)
// The LDS slot of p is p with bits 5 and 6 folded into the low five bits (slot = p ^ 5 p5 ^ 26 p6):
// for each of the four layouts the 32 lanes of a half-wave then hit 32 different banks, both when
// a round's results are written and when the next round's operands are read.
static constexpr int NTTW_LOG = 8;                       // 256 points per wave
static constexpr int NTTW_WAVES = 4;                     // waves per workgroup; three workgroups per CU (41 KiB of LDS each)
static constexpr int NTTW_PLANE = 256;                   // dwords per limb plane
// Waves per SIMD and exchange planes per wave.  The transpose between two rounds is independent per limb, so a wave
// may send its nine limbs through fewer than nine planes, a group of planes at a time (the DS unit serves a wave's
// instructions in order): with NTTW_XPLANES <= 8 the planes of the four waves fit inside the 32 KiB tile they alias,
// a workgroup needs 36.5 KiB of LDS and FOUR of them fit a CU (NTTW_OCC = 4, 128 VGPRs).
#ifndef NTTW_OCC
#define NTTW_OCC 3
#endif
#ifndef NTTW_XPLANES
#define NTTW_XPLANES 9
#endif
static constexpr int NTTW_XP = NTTW_XPLANES;
static constexpr size_t NTTW_TILE_BYTES = 32768;
static constexpr size_t NTTW_X_BYTES = (size_t)NTTW_WAVES * NTTW_XP * NTTW_PLANE * 4 > NTTW_TILE_BYTES ? (size_t)NTTW_WAVES * NTTW_XP * NTTW_PLANE * 4 : NTTW_TILE_BYTES;
static constexpr size_t NTTW_LDS_BYTES = NTTW_X_BYTES + 128 * 9 * 4 + 16;   // exchange planes (under the tile) + N/2 twiddles + the next block-group (two words)

// Strided lines: element q of a line is in_elem_stride elements away from element q - 1, but the
// same element of the NEXT line is adjacent.  A lone wave reading its own line therefore issues 64
// separate 16-byte requests per instruction (measured: the loads and stores cost 0.17 ms of a 0.9 ms
// pass although the HBM traffic is ideal).  Where the 4 * 2^(8-m) lines of a workgroup are adjacent
// (NttPass::coop), the workgroup moves them as ONE tile: every 8 consecutive lanes read or write 128
// contiguous bytes, and the tile goes through LDS (aliasing the exchange planes, hence the
// barriers) to hand each wave its own lines.
// Tile layouts: 2048 slots of 16 bytes, slot(e, h) for half h of element e = row << m | position (10 bits).  The four
// accesses want different things of the low slot bits (= the bank group: ds_write_b128 serves 8 consecutive lanes per
// LDS cycle and is conflict-free when their slots differ mod 8, ds_read_b128 serves 16 lanes -- {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31} and the same + 32 -- and wants them different mod 16; MI355X_MICROARCH.md, LDS):
//   fill   8 lanes = both halves of rows i .. i + 3 at one position        -> low 3 bits from (h, row & 3)
//   take   16 lanes = one half of elements e0 + 4 L + r, L as above        -> low 4 bits from e bits 2 .. 5
//   give   8 lanes = one half of 8 consecutive elements                    -> low 3 bits from e bits 0 .. 2
//   drain  16 lanes = both halves of rows (0, 1) or (2, 3) at 4 positions  -> low 4 bits from (h, row & 3, position & 1)
// One layout cannot serve all four (the first version's, e ^ (pos >> 3 & 7), left fill 4-way and take / drain 2-way
// conflicted: 34 M conflict cycles of 88 M LDS cycles per pass, profiles/r03_a_ntt2p24_pmc.txt), but fill / take and
// give / drain never meet in one tile, so each pair gets its own.  Both are bijections for every m (the XOR mixes
// low bits with a function of high ones only); the conflict-free claims hold for m >= 6, the sizes tiles are used at.
HD uint32_t nttw_tslot_in(uint32_t e, uint32_t h, uint32_t m) {
    const uint32_t hb = m > 6u ? m : 6u;
    const uint32_t a = (((h << 6) | ((e & 3u) << 4) | (e >> 6)) << 4) | ((e >> 2) & 15u);
    return a ^ ((((e >> hb) & 3u) << 1) | h);
}
HD uint32_t nttw_tslot_out(uint32_t e, uint32_t h, uint32_t m) {
    const uint32_t hb = m > 3u ? m : 3u;
    const uint32_t b = ((e >> 3) << 4) | (h << 3) | (e & 7u), x = e >> hb;
    return b ^ (((x & 3u) << 1) | (m <= 6u ? (x >> 2) & 1u : 0u));    // 64-point lines: a drain group spans 8 rows of one position
}
HD uint32_t nttw_slot(uint32_t p) { return p ^ (((p >> 5) & 1u) * 0x05u) ^ (((p >> 6) & 1u) * 0x1Au); }
HD uint32_t nttw_pos(uint32_t lane, uint32_t r, uint32_t t) {
    const uint32_t sh = 2 * t;
    return ((lane >> sh) << (sh + 2)) | (r << sh) | (lane & ((1u << sh) - 1u));
}
// one DIT butterfly: (u, v) -> (u + w v, u - w v); bound(u), bound(v) <= b on entry, <= b + 3 on exit
// (round 0 ends one higher: nttw_bfly_one2).
// No carry pass (field29.cuh, f29_add_nc / f29_sub_nc): the limbs of u grow by < 2^30 per layer.  A
// round is two layers on values that enter carried (limbs < 2^29 + 8: from f29_unpack or f29_carry), so
// the second layer multiplies limbs < 1.5 * 2^30 and leaves limbs < 2.5 * 2^30 + 8 -- inside the 32-bit
// limb, and inside the multiplier's 2^31.5 for the post-twiddle product of the last round.  The wave
// carries once per round, when it parks the values in LDS.
template <class F> DEV void nttw_bfly(Fe29<F> &u, Fe29<F> &v, const uint32_t *tw9) {
    Fe29<F> w;
#pragma unroll
    for (int k = 0; k < 9; k++) w.l[k] = tw9[k];
    F29_SET(w, 1.0);
    const Fe29<F> t = f29_mul(v, w);
    v = f29_sub_nc<3>(u, t);
    u = f29_add_nc(u, t);
}
template <class F> DEV void nttw_bfly_one(Fe29<F> &u, Fe29<F> &v) {    // w = 1 (layer 0 of every line: v is an unpacked value < 2 P)
    const Fe29<F> t = v;
    v = f29_sub_nc<3>(u, t);
    u = f29_add_nc(u, t);
}
// w = 1 again: the butterflies of layer 1 whose position has p mod 2 = 0 (omega^0).  v = the uncarried sum of
// two unpacked values out of layer 0 (bound 4, limbs < 2^30): the bias is raised by 2^30 per limb.  Bounds on
// exit: u <= 8, v <= 9 (with the product: 6 and 7); limbs < 2^31, inside what f29_carry and the multiplier take.
template <class F> DEV void nttw_bfly_one2(Fe29<F> &u, Fe29<F> &v) {
    const Fe29<F> t = v;
    v = f29_sub_nc<5, F, 2>(u, t);
    u = f29_add_nc(u, t);
}

// the four points of a lane are four named values (never an indexed array: a dynamically indexed
// register array would be demoted to scratch memory)
template <class F, int K0, int K1> DEV void nttw_put(uint32_t *X, const Fe29<F> &v, uint32_t slot) {
#pragma unroll
    for (int k = K0; k < K1; k++) X[(k - K0) * NTTW_PLANE + slot] = v.l[k];
}
template <class F, int K0, int K1> DEV void nttw_get(const uint32_t *X, Fe29<F> &v, uint32_t slot, double bound) {
#pragma unroll
    for (int k = K0; k < K1; k++) v.l[k] = X[(k - K0) * NTTW_PLANE + slot];
    F29_SET(v, bound);
    (void)bound;
}
// limbs K0 .. K1 - 1 of the wave's 256 points change layout: slots sp* (this round's results) -> slots sg* (the next round's operands)
template <class F, int K0, int K1>
DEV void nttw_xchg(uint32_t *X, Fe29<F> &x0, Fe29<F> &x1, Fe29<F> &x2, Fe29<F> &x3, const uint32_t (&sp)[4], const uint32_t (&sg)[4], double bound) {
    nttw_put<F, K0, K1>(X, x0, sp[0]); nttw_put<F, K0, K1>(X, x1, sp[1]); nttw_put<F, K0, K1>(X, x2, sp[2]); nttw_put<F, K0, K1>(X, x3, sp[3]);
    WAVE_SYNC();
    nttw_get<F, K0, K1>(X, x0, sg[0], bound); nttw_get<F, K0, K1>(X, x1, sg[1], bound); nttw_get<F, K0, K1>(X, x2, sg[2], bound); nttw_get<F, K0, K1>(X, x3, sg[3], bound);
    WAVE_SYNC();
}
template <class F> struct NttwIo {
    const unsigned char *src;
    unsigned char *dst;
    const unsigned char *t_lo, *t_hi, *scale;
    NttPass ps;
    uint32_t lo_mask, split_mask, m, N;
    // DIT position p of this wave (round-0 layout) -> the canonical input element, or zero past the last line
    DEV Fe<typename F::Sat> fetch(uint32_t line0, uint32_t p) const {
        using S = typename F::Sat;
        const uint32_t pos = p & (N - 1), line = line0 + (p >> m);
        const uint32_t q = m ? (__brev(pos) >> (32 - m)) : 0;      // natural input index of DIT position pos
        if (line >= ps.nlines) return fe_zero<S>();
        return fe_load<S>(src + (in_start(line) + (size_t)q * ps.in_elem_stride) * 32);
    }
    DEV size_t in_start(uint32_t line) const { return (size_t)(line >> ps.split) * ps.in_hi + (size_t)(line & split_mask) * ps.in_lo; }
    DEV size_t out_start(uint32_t line) const { return (size_t)(line >> ps.split) * ps.out_hi + (size_t)(line & split_mask) * ps.out_lo; }
    // output k of its line: times the four-step twiddle -- left as the multiplier returns it, some
    // residue < 2 P < 2^255 that the next pass unpacks like any other value -- or, in the last pass,
    // times the final scale and canonical
    DEV Fe<typename F::Sat> finish(uint32_t line, uint32_t k, const Fe29<F> &x) const {
#if defined(NTTW_PROBE_NOMATH) || defined(NTTW_PROBE_NOFINISH)   // timing probes (tools/build_probe_variants.sh): results wrong on purpose
        return f29_pack(f29_carry(x));
#endif
        if (ps.tw_shift != 0xFFFFFFFFu) {
            const uint64_t e = ps.tw_single == 2 ? (((uint64_t)line << m) | k) : (uint64_t)(line >> ps.tw_line_shift) * k;
            const Fe29<F> tw = ps.tw_single ? tw_load<F>(t_lo + (size_t)e * TW_BYTES)
                                            : f29_mul(tw_load<F>(t_hi + (size_t)(e >> ps.tw_shift) * TW_BYTES), tw_load<F>(t_lo + (size_t)(e & lo_mask) * TW_BYTES));
            return f29_pack_product(f29_mul(x, tw));
        }
        if (ps.prescaled) return ntt_canonical_product(f29_redc(x));
        return ntt_canonical_product(f29_mul(x, tw_load<F>(scale)));
    }
    DEV void store(uint32_t line0, uint32_t p, const Fe29<F> &x) const {
        const uint32_t k = p & (N - 1), line = line0 + (p >> m);
        if (line >= ps.nlines) return;
        fe_store(dst + (out_start(line) + (size_t)k * ps.out_elem_stride) * 32, finish(line, k, x));
    }
    // ---- workgroup tiles (NttPass::coop): 1024 elements = TL adjacent lines x N points -------------
    // Both layouts are GF(2)-linear in (e, h): slot(e ^ d) = slot(e) ^ slot(d).  Every access below therefore computes
    // ONE slot per lane and phase with vector instructions and reaches the others (the other half, the other three
    // points of the lane, the other seven copy iterations) by an XOR with a wave-uniform constant -- the passes are
    // VALU-bound, and a slot formula evaluated per access cost more than the bank conflicts it removed.
    // global -> LDS tile, column q of every row stored at its DIT position brev(q); 8 x 16 bytes per lane.  Copy
    // iteration `it` handles element index e' = it * 128 + tid / 2: the same row, column q0 + it * 2^(m-3), whose DIT
    // position is brev(q0) | brev3(it).
    DEV void tile_fill(U4 *tile, uint32_t line_blk0) const {
        const uint32_t log_tl = 10 - m, e0 = threadIdx.x >> 1, q0 = e0 >> log_tl, i = e0 & ((1u << log_tl) - 1u), h = threadIdx.x & 1u;
        const unsigned char *base = src + (in_start(line_blk0) + (size_t)i + (size_t)q0 * ps.in_elem_stride) * 32 + h * 16;
        const size_t step = ((size_t)ps.in_elem_stride << (m - 3)) * 32;                   // m >= 3 wherever a pass has tiles (TL <= 128 rows)
        const uint32_t s0 = nttw_tslot_in((i << m) | (m ? (__brev(q0) >> (32 - m)) : 0u), h, m);
#pragma unroll
        for (int it = 0; it < 8; it++) {                           // (the compiler hoists the eight loads above the LDS stores)
            constexpr uint32_t B3[8] = {0, 4, 2, 6, 1, 5, 3, 7};
            const U4 v = *reinterpret_cast<const U4 *>(base + (size_t)it * step);
            tile[s0 ^ nttw_tslot_in(B3[it], 0u, m)] = v;
        }
    }
    // this wave's element (wave << 8 | p), p = 4 lane + r: four points of a lane differ in e bits 0, 1
    DEV uint32_t take_base(uint32_t wave, uint32_t lane) const { return nttw_tslot_in((wave << 8) | (lane << 2), 0u, m); }
    DEV Fe29<F> tile_take(const U4 *tile, uint32_t tb, uint32_t r) const {
        using S = typename F::Sat;
        const uint32_t sl = tb ^ nttw_tslot_in(r, 0u, m);
        const U4 a = tile[sl], b = tile[sl ^ nttw_tslot_in(0u, 1u, m)];
        Fe<S> s;
        s.l[0] = a.x; s.l[1] = a.y; s.l[2] = a.z; s.l[3] = a.w; s.l[4] = b.x; s.l[5] = b.y; s.l[6] = b.z; s.l[7] = b.w;
        return ntt_unpack<F>(s);
    }
    // results of the last round: p = nttw_pos(lane, r, tl) = nttw_pos(lane, 0, tl) | r << 2 tl
    DEV uint32_t give_base(uint32_t wave, uint32_t lane, uint32_t tl) const { return nttw_tslot_out((wave << 8) | nttw_pos(lane, 0, tl), 0u, m); }
    DEV void tile_give(U4 *tile, uint32_t gb, uint32_t line0, uint32_t p, uint32_t r, uint32_t tl, const Fe29<F> &x) const {
        const Fe<typename F::Sat> s = finish(line0 + (p >> m), p & (N - 1), x);
        const uint32_t sl = gb ^ nttw_tslot_out(r << (2 * tl), 0u, m);
        tile[sl] = U4{s.l[0], s.l[1], s.l[2], s.l[3]};
        tile[sl ^ nttw_tslot_out(0u, 1u, m)] = U4{s.l[4], s.l[5], s.l[6], s.l[7]};
    }
    // LDS tile -> global; copy iteration `it` handles output position k0 + it * 2^(m-3) of the same row
    DEV void tile_drain(const U4 *tile, uint32_t line_blk0) const {
        const uint32_t log_tl = 10 - m, e0 = threadIdx.x >> 1, k0 = e0 >> log_tl, i = e0 & ((1u << log_tl) - 1u), h = threadIdx.x & 1u;
        unsigned char *base = dst + (out_start(line_blk0) + (size_t)i + (size_t)k0 * ps.out_elem_stride) * 32 + h * 16;
        const size_t step = ((size_t)ps.out_elem_stride << (m - 3)) * 32;
        const uint32_t s0 = nttw_tslot_out((i << m) | k0, h, m);
#pragma unroll
        for (int it = 0; it < 8; it++)
            *reinterpret_cast<U4 *>(base + (size_t)it * step) = tile[s0 ^ nttw_tslot_out((uint32_t)it << (m - 3), 0u, m)];
    }
};

// Timing probe (tools/build_probe_variants.sh ... -DNTTW_PROBE_STAMPS; never in the product build): lane 0 of every workgroup
// writes s_memrealtime (100 MHz) at the phase boundaries of its first NTTW_STAMP_ITERS iterations to a buffer of its own.
#ifdef NTTW_PROBE_STAMPS
static constexpr uint32_t NTTW_STAMP_ITERS = 48, NTTW_STAMP_SLOTS = 8, NTTW_STAMP_WGS = 1024;
__device__ uint64_t g_nttw_stamps[NTTW_STAMP_WGS * (NTTW_STAMP_ITERS * NTTW_STAMP_SLOTS + 1)];
__device__ uint64_t g_nttw_clk[4];      // workgroup 0: (s_memtime, s_memrealtime) at its start and at its end: the shader clock of the launch
#define NTTW_STAMP(it, k) do { if (threadIdx.x == 0 && (it) < NTTW_STAMP_ITERS && blockIdx.x < NTTW_STAMP_WGS) { \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        g_nttw_stamps[(size_t)blockIdx.x * (NTTW_STAMP_ITERS * NTTW_STAMP_SLOTS + 1) + 1 + (it) * NTTW_STAMP_SLOTS + (k)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define NTTW_STAMP(it, k) ((void)0)
#endif

// COOP: bit 0 / bit 1 = the input / output lines of a workgroup move as one tile (NttPass::coop);
// a compile-time choice so that neither path's registers burden the other
template <class F, int COOP>
KERNEL void __launch_bounds__(64 * NTTW_WAVES) __attribute__((amdgpu_waves_per_eu(NTTW_OCC, NTTW_OCC))) k_ntt_wave(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst, NttPass ps,
                        const unsigned char *__restrict__ line_tw,   // omega_N^j, j < N/2, TW_BYTES each
                        const unsigned char *__restrict__ t_lo, const unsigned char *__restrict__ t_hi,
                        const unsigned char *__restrict__ scale,     // multiplier-form scale, or one, for the last pass
                        uint32_t *__restrict__ ctr) {                // NTTW_RANGES work counters of this launch, zero
    DYN_SHARED(uint32_t, lds);
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t *X = lds + (size_t)wv * NTTW_XP * NTTW_PLANE;        // this wave's exchange planes
    uint32_t *TW = lds + NTTW_X_BYTES / 4;                        // the line's twiddles, 9 dwords each, shared
    uint32_t *NEXT = TW + 128 * 9;                                // the block-group of the next iteration, by parity
    const uint32_t m = ps.log_len, N = 1u << m;
    for (uint32_t i = threadIdx.x; i < (N / 2) * 9; i += blockDim.x)
        TW[i] = reinterpret_cast<const uint32_t *>(line_tw + (size_t)(i / 9) * TW_BYTES)[i % 9];
    __syncthreads();
    const NttwIo<F> io{src, dst, t_lo, t_hi, scale, ps, ps.tw_shift == 0xFFFFFFFFu ? 0 : ((1u << ps.tw_shift) - 1u), (1u << ps.split) - 1u, m, N};
    const uint32_t log_lpw = NTTW_LOG - m, lpw = 1u << log_lpw;    // lines per wave
    const uint32_t rounds = (m + 1) / 2;
    // block-groups of NTTW_WAVES consecutive wave-groups (4 * lpw consecutive lines: their strided
    // elements share 128-byte lines), handed out by nttw_first / nttw_grab
    const uint32_t ngroups = (ps.nlines + lpw - 1) >> log_lpw, nbg = (ngroups + NTTW_WAVES - 1) / NTTW_WAVES;
    auto first_line = [&](uint32_t bg) { return (bg * NTTW_WAVES + wv) << log_lpw; };   // first line of this wave
    // (measured without gain, rounds 2 - 4: a register prefetch of the next group's elements, issued at the top or in front of the
    // drain; a fourth wave per SIMD, as eight-wave workgroups or with the exchange through five planes; Shoup products)
    // ALIASING INVARIANT: `tile` (32 KiB) lies over the exchange planes X of all four waves.  Every transition
    // between a tile phase (fill / take, give / drain) and an exchange phase (put / get of ANY wave) is therefore
    // separated by a workgroup barrier, including the one from the last round of group g to the fill of group g + 1.
    U4 *tile = reinterpret_cast<U4 *>(lds);
    const uint32_t row0 = wv << log_lpw;                           // this wave's first row of the tile
#ifdef NTTW_PROBE_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < NTTW_STAMP_WGS) {
        uint32_t hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_nttw_stamps[(size_t)blockIdx.x * (NTTW_STAMP_ITERS * NTTW_STAMP_SLOTS + 1)] = ((uint64_t)xcc << 32) | hwid;
    }
    uint32_t it_no = 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) { g_nttw_clk[0] = __builtin_amdgcn_s_memtime(); g_nttw_clk[1] = __builtin_amdgcn_s_memrealtime(); }
#endif
    uint32_t ranges_left = 0, par = 0;                             // (thread 0) ranges emptied; parity of the iteration
    if (threadIdx.x == 0) NEXT[0] = nttw_first(nbg);
    __syncthreads();
    uint32_t idx = NEXT[0];
    while (idx != NTTW_DONE) {
        const uint32_t line0 = first_line(idx), line_blk0 = line0 - row0;
        Fe29<F> x0, x1, x2, x3;
        NTTW_STAMP(it_no, 0);
#ifndef NTTW_PROBE_NOMEM                                           // timing probe: no tile loads, no tile stores (results wrong on purpose)
        if (COOP & 1) io.tile_fill(tile, line_blk0);
#endif
        // the next block-group: its counter answers behind this iteration's tile loads.  NEXT[par ^ 1] was last read
        // before the barrier below of the iteration before, which every wave has passed
        if (threadIdx.x == 0) NEXT[par ^ 1u] = nttw_grab(ctr, ranges_left, nbg);
        __syncthreads();
        const uint32_t idx_next = NEXT[par ^ 1u];
        if (COOP & 1) {
            NTTW_STAMP(it_no, 1);
            const uint32_t tb = io.take_base(wv, lane);
            x0 = io.tile_take(tile, tb, 0); x1 = io.tile_take(tile, tb, 1); x2 = io.tile_take(tile, tb, 2); x3 = io.tile_take(tile, tb, 3);
            __syncthreads();                                       // the exchange planes overwrite the tile
            NTTW_STAMP(it_no, 2);
        } else {
            x0 = ntt_unpack<F>(io.fetch(line0, nttw_pos(lane, 0, 0))); x1 = ntt_unpack<F>(io.fetch(line0, nttw_pos(lane, 1, 0)));
            x2 = ntt_unpack<F>(io.fetch(line0, nttw_pos(lane, 2, 0))); x3 = ntt_unpack<F>(io.fetch(line0, nttw_pos(lane, 3, 0)));
        }
#ifdef NTTW_PROBE_NOMATH
        for (uint32_t t = 0; t < 0; t++) {
#else
        for (uint32_t t = 0; t < rounds; t++) {                    // (the line length as a template parameter, with and without the rounds unrolled: +- 1 %, not kept)
#endif
            if (t) {                                               // transpose: layout t-1 -> layout t (carried: see nttw_bfly)
                // nttw_slot is GF(2)-linear and nttw_pos(lane, r, t) = nttw_pos(lane, 0, t) ^ r << 2 t: one slot per lane and layout
                // with vector instructions, the other three by an XOR with a wave-uniform constant
                const uint32_t sp0 = nttw_slot(nttw_pos(lane, 0, t - 1)), sg0 = nttw_slot(nttw_pos(lane, 0, t));
                const uint32_t sp[4] = {sp0, sp0 ^ nttw_slot(1u << (2 * t - 2)), sp0 ^ nttw_slot(2u << (2 * t - 2)), sp0 ^ nttw_slot(3u << (2 * t - 2))};
                const uint32_t sg[4] = {sg0, sg0 ^ nttw_slot(1u << (2 * t)), sg0 ^ nttw_slot(2u << (2 * t)), sg0 ^ nttw_slot(3u << (2 * t))};
                const double bound = 3.0 + 6.0 * t;                // + 3 per layer, + 1 for the product-free butterfly of layer 1
                x0 = f29_carry(x0); x1 = f29_carry(x1); x2 = f29_carry(x2); x3 = f29_carry(x3);
                constexpr int KS = NTTW_XP < 9 ? NTTW_XP : 9;      // the limbs go through the planes in groups of KS
                nttw_xchg<F, 0, KS>(X, x0, x1, x2, x3, sp, sg, bound);
                if constexpr (KS < 9) nttw_xchg<F, KS, (2 * KS < 9 ? 2 * KS : 9)>(X, x0, x1, x2, x3, sp, sg, bound);
                if constexpr (2 * KS < 9) nttw_xchg<F, 2 * KS, (3 * KS < 9 ? 3 * KS : 9)>(X, x0, x1, x2, x3, sp, sg, bound);
                static_assert(3 * KS >= 9, "at most three groups of exchange planes");
            }
            const uint32_t s = 2 * t, j = lane & ((1u << s) - 1u);  // j = p mod 2^s
            if (s == 0) {
                nttw_bfly_one(x0, x1);
                nttw_bfly_one(x2, x3);
            } else {
                const uint32_t *tw = TW + (size_t)(j << (m - 1 - s)) * 9;
                nttw_bfly(x0, x1, tw);
                nttw_bfly(x2, x3, tw);
            }
            if (s + 1 < m) {                                       // layer s + 1: p mod 2^(s+1) = j + (r & 1) 2^s
                if (s == 0) nttw_bfly_one2(x0, x2);                // j = 0 in every lane: omega^0, no product
                else nttw_bfly(x0, x2, TW + (size_t)(j << (m - 2 - s)) * 9);
                nttw_bfly(x1, x3, TW + (size_t)((j + (1u << s)) << (m - 2 - s)) * 9);
            }
        }
        const uint32_t tl = rounds ? rounds - 1 : 0;
        NTTW_STAMP(it_no, 3);
        if (COOP & 2) {
            __syncthreads();                                       // every wave is done with its exchange planes
            NTTW_STAMP(it_no, 4);
            const uint32_t gb = io.give_base(wv, lane, tl);
            io.tile_give(tile, gb, line0, nttw_pos(lane, 0, tl), 0, tl, x0); io.tile_give(tile, gb, line0, nttw_pos(lane, 1, tl), 1, tl, x1);
            io.tile_give(tile, gb, line0, nttw_pos(lane, 2, tl), 2, tl, x2); io.tile_give(tile, gb, line0, nttw_pos(lane, 3, tl), 3, tl, x3);
            NTTW_STAMP(it_no, 5);
            __syncthreads();
            NTTW_STAMP(it_no, 6);
#ifndef NTTW_PROBE_NOMEM
            io.tile_drain(tile, line_blk0);
#endif
            __syncthreads();                                       // the next group's tile or exchange planes reuse the space
            NTTW_STAMP(it_no, 7);
        } else {
            io.store(line0, nttw_pos(lane, 0, tl), x0); io.store(line0, nttw_pos(lane, 1, tl), x1);
            io.store(line0, nttw_pos(lane, 2, tl), x2); io.store(line0, nttw_pos(lane, 3, tl), x3);
            // `tile` aliases the exchange planes of ALL four waves: the next group's tile_fill must not start
            // while a slower wave is still between nttw_put and nttw_get (control flow is block-uniform)
            if (COOP & 1) __syncthreads();
            NTTW_STAMP(it_no, 7);
        }
#ifdef NTTW_PROBE_STAMPS
        it_no++;
#endif
        idx = idx_next;
        par ^= 1u;
    }
#ifdef NTTW_PROBE_STAMPS
    if (threadIdx.x == 0 && blockIdx.x == 0) { g_nttw_clk[2] = __builtin_amdgcn_s_memtime(); g_nttw_clk[3] = __builtin_amdgcn_s_memrealtime(); }
#endif
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
