// Carry-free 254-bit field arithmetic for the EC-heavy kernels: 9 limbs of 29 bits.
//
// Why: on gfx950 v_mad_u64_u32 issues every ~4.8 cycles per wave but has no carry-in, so a
// saturated 8 x 32-bit CIOS multiplier spends two thirds of its issue slots on zero-extending
// moves and 64-bit carry adds around its 128 multiply-adds (measured: 1805 cycles per wave
// multiplication).  With 29-bit limbs a 64-bit column holds nine 58-60-bit products plus the nine
// Montgomery reduction products without overflow, so every partial product is one in-place
// `acc += a_i * b_j` (v_mad_u64_u32 with the accumulator as its own addend), and carries are
// resolved once per column.
//
// Representation ("loose"): value = sum l[i] * 2^(29 i), every limb < 2^29 + 2^5, value < 11 P,
// congruent to x * 2^261 mod P (Montgomery form with R' = 2^261).  Multiplication accepts limbs
// up to 2^30 and any a * b <= 168 P^2, and returns a loose value < 2 P; subtraction adds a
// multiple K P of the modulus chosen per call site from the proven bounds of its operands
// (curve29.cuh lists them).  Nothing is reduced to canonical form until a result leaves the
// GPU (f29_to_r256).
//
// The reference's memory layout (4 x u64, R = 2^256, src/commitment.rs / halo2curves) is
// converted at the boundary only: x~ = x * 2^256 -> f29_from_r256() multiplies by 2^266 * 2^-261.
#pragma once
#include "field.cuh"

struct Fq29 {
    using Sat = FqP;
    static constexpr uint32_t P[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    static constexpr uint32_t N0 = 0x04866389u;   // -P^-1 mod 2^29
    static constexpr uint32_t ONE[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};   // 2^261 mod P
    static constexpr uint32_t R256_TO_R261[9] = {0x13349ca1u, 0x1a5d84a8u, 0x0a3e5cacu, 0x100249e0u, 0x12b951e8u, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};   // 2^266 mod P
};
struct Fr29 {
    using Sat = FrP;
    static constexpr uint32_t P[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    static constexpr uint32_t N0 = 0x0fffffffu;
    static constexpr uint32_t ONE[9] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
    static constexpr uint32_t R256_TO_R261[9] = {0x0fffead7u, 0x1d5444f4u, 0x04438aa5u, 0x03b4d096u, 0x134c84dau, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};
};

static constexpr uint32_t M29 = 0x1FFFFFFFu;

// F29_TRACK (host test builds only): every value carries a proven upper bound in units of P,
// and each operation asserts its preconditions, so a formula that could overflow a limb or a
// subtraction bias fails deterministically instead of for one input in 2^60.
#ifdef F29_TRACK
#include <cassert>
#define F29_BD(x) double bd = (x);
#define F29_SET(v, x) ((v).bd = (x))
#define F29_GET(v) ((v).bd)
#define F29_ASSERT(c) assert(c)
#else
#define F29_BD(x)
#define F29_SET(v, x) ((void)0)
#define F29_GET(v) (0.0)
#define F29_ASSERT(c) ((void)0)
#endif
static constexpr double F29_RP_OVER_P = 168.0;   // 2^261 / P = 168.9...: a*b <= 168 P^2 -> product < 2 P

template <class F> struct Fe29 {
    uint32_t l[9];
    F29_BD(0.0)
};

template <class F> HD Fe29<F> f29_zero() {
    Fe29<F> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = 0;
    return r;
}
template <class F> HD Fe29<F> f29_one() {
    Fe29<F> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = F::ONE[i];
    F29_SET(r, 1.0);
    return r;
}
// exact all-limbs-zero test (the identity marker ZZ = 0 is only ever written as literal zeros)
template <class F> HD bool f29_is_literal_zero(const Fe29<F> &a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= a.l[i];
    return o == 0;
}

// one parallel carry pass: limbs < 2^32 in, limbs < 2^29 + 8 out (top limb keeps its excess)
template <class F> HD Fe29<F> f29_carry(const Fe29<F> &a) {
    Fe29<F> r;
    r.l[0] = a.l[0] & M29;
#pragma unroll
    for (int i = 1; i < 8; i++) r.l[i] = (a.l[i] & M29) + (a.l[i - 1] >> 29);
    r.l[8] = a.l[8] + (a.l[7] >> 29);
    F29_SET(r, F29_GET(a));
    return r;
}
template <class F> HD Fe29<F> f29_add(const Fe29<F> &a, const Fe29<F> &b) {
    Fe29<F> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    F29_SET(r, F29_GET(a) + F29_GET(b));
    F29_ASSERT(F29_GET(r) <= 40.0);
    return f29_carry(r);
}
template <class F> HD Fe29<F> f29_dbl(const Fe29<F> &a) { return f29_add(a, a); }
// K * P with limbs 0..7 raised by 2^31 (borrowed as 4 from the next limb), so that every limb
// exceeds the matching limb of any loose subtrahend smaller than K * P.
template <class F, int K> struct F29Bias {
    uint32_t l[9];
    constexpr F29Bias() : l{} {
        uint64_t carry = 0;
        for (int i = 0; i < 9; i++) {
            uint64_t v = (uint64_t)K * F::P[i] + carry;
            l[i] = (i < 8) ? (uint32_t)(v & M29) : (uint32_t)v;
            carry = v >> 29;
        }
        for (int i = 0; i < 8; i++) {
            l[i] += 0x80000000u;
            l[i + 1] -= 4;
        }
    }
};
// a - b + K P.  Requires b < K P (loose limbs); result < a + K P.
template <int K, class F> HD Fe29<F> f29_sub(const Fe29<F> &a, const Fe29<F> &b) {
    constexpr F29Bias<F, K> bias{};
    Fe29<F> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + bias.l[i] - b.l[i];
    F29_ASSERT(F29_GET(b) <= (double)K - 0.01);     // subtrahend below the bias, limb by limb
    F29_SET(r, F29_GET(a) + (double)K);
    F29_ASSERT(F29_GET(r) <= 40.0);
    return f29_carry(r);
}
// a - b - 2 c + K P in one pass (one carry instead of three).  Requires b + 2 c < K P; the bias
// limbs (>= 2^31) still dominate b + 2 c limb by limb since carried limbs are < 2^29 + 8.
template <int K, class F> HD Fe29<F> f29_sub_b_2c(const Fe29<F> &a, const Fe29<F> &b, const Fe29<F> &c) {
    constexpr F29Bias<F, K> bias{};
    Fe29<F> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + bias.l[i] - b.l[i] - 2u * c.l[i];
    F29_ASSERT(F29_GET(b) + 2.0 * F29_GET(c) <= (double)K - 0.01);
    F29_SET(r, F29_GET(a) + (double)K);
    F29_ASSERT(F29_GET(r) <= 40.0);
    return f29_carry(r);
}
template <int K, class F> HD Fe29<F> f29_neg(const Fe29<F> &b) { return f29_sub<K>(f29_zero<F>(), b); }
// multiply by a small constant by repeated addition (3 x = 2 x + x)
template <class F> HD Fe29<F> f29_triple(const Fe29<F> &a) { return f29_add(f29_dbl(a), a); }

// Uncarried forms for chains of additions that end in a multiplication (the NTT butterflies): no
// carry pass, so limbs grow -- by at most 2^29 per f29_add_nc of a multiplier result and by at most
// 2^30 per f29_sub_nc.  The multiplier tolerates it: with one operand carried (limbs < 2^29 + 8) a
// column of nine products and nine reduction products stays below 2^64 for limbs of the other
// operand up to 2^31.5 (9 * 2^31.5 * 2^29 + 9 * 2^58 < 2^64).  f29_carry() restores limbs < 2^29 + 8
// from any limbs < 2^32.  Callers state their limb budget where they use these (ntt_kernels.cuh).
template <class F> HD Fe29<F> f29_add_nc(const Fe29<F> &a, const Fe29<F> &b) {
    Fe29<F> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    F29_SET(r, F29_GET(a) + F29_GET(b));
    F29_ASSERT(F29_GET(r) <= 40.0);
    return r;
}
// K * P with limbs 0..7 raised by RAISE * 2^29 (borrowed as RAISE from the next limb): dominates, limb by
// limb, any subtrahend < K P whose limbs 0..7 are below RAISE * 2^29 + 8 -- RAISE = 1: a multiplier result, an
// unpacked value or a carried one (limbs < 2^29 + 8); RAISE = 2: the uncarried sum of two of those.  `ok` states
// what that needs of the modulus: every limb of K P at least 8 (so that the borrow never wraps and a carried
// subtrahend's excess of up to 7 is covered); checked where the bias is used.
template <class F, int K, int RAISE = 1> struct F29BiasTight {
    uint32_t l[9];
    bool ok;
    constexpr F29BiasTight() : l{}, ok(true) {
        uint64_t carry = 0;
        for (int i = 0; i < 9; i++) {
            uint64_t v = (uint64_t)K * F::P[i] + carry;
            l[i] = (i < 8) ? (uint32_t)(v & M29) : (uint32_t)v;
            carry = v >> 29;
            if (l[i] < 8u) ok = false;
        }
        for (int i = 0; i < 8; i++) {
            l[i] += (uint32_t)RAISE * 0x20000000u;
            l[i + 1] -= (uint32_t)RAISE;
        }
    }
};
// a - b + K P without a carry pass.  Requires b < (K - 1) P with limbs 0..7 < RAISE * 2^29 + 8 (RAISE = 1:
// straight from f29_mul or f29_unpack, or carried); limbs of the result < limbs of a + (RAISE + 1) 2^29.
template <int K, class F, int RAISE = 1> HD Fe29<F> f29_sub_nc(const Fe29<F> &a, const Fe29<F> &b) {
    constexpr F29BiasTight<F, K, RAISE> bias{};
    static_assert(bias.ok, "a limb of K * P is too small for this bias");
    Fe29<F> r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        F29_ASSERT(b.l[i] <= bias.l[i]);
        r.l[i] = a.l[i] + bias.l[i] - b.l[i];
        F29_ASSERT(r.l[i] >= a.l[i]);                   // no wrap of the 32-bit limb
    }
    F29_ASSERT(F29_GET(b) <= (double)K - 1.0);
    F29_SET(r, F29_GET(a) + (double)K);
    F29_ASSERT(F29_GET(r) <= 40.0);
    return r;
}
#ifdef F29_TRACK
// test builds: the 64-bit columns of a 9 x 9 product plus its reduction cannot overflow for these limbs
template <class F> inline void f29_check_columns(const Fe29<F> &a, const Fe29<F> &b) {
    uint32_t ma = 0, mb = 0;
    for (int i = 0; i < 9; i++) { ma = a.l[i] > ma ? a.l[i] : ma; mb = b.l[i] > mb ? b.l[i] : mb; }
    const long double col = 9.0L * (long double)ma * (long double)mb + 9.0L * 536870912.0L * 536870912.0L + 68719476736.0L;
    assert(col < 18446744073709551616.0L);
}
template <class F> inline void f29_check_columns2(const Fe29<F> &a, const Fe29<F> &b, const Fe29<F> &c, const Fe29<F> &d) {
    uint32_t m[4] = {0, 0, 0, 0};
    const Fe29<F> *v[4] = {&a, &b, &c, &d};
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < 9; i++) m[k] = v[k]->l[i] > m[k] ? v[k]->l[i] : m[k];
    const long double col = 9.0L * ((long double)m[0] * m[1] + (long double)m[2] * m[3]) + 9.0L * 536870912.0L * 536870912.0L + 68719476736.0L;
    assert(col < 18446744073709551616.0L);
}
#define F29_CHECK_COLUMNS(a, b) f29_check_columns(a, b)
#define F29_CHECK_COLUMNS2(a, b, c, d) f29_check_columns2(a, b, c, d)
#else
#define F29_CHECK_COLUMNS(a, b) ((void)0)
#define F29_CHECK_COLUMNS2(a, b, c, d) ((void)0)
#endif

// Montgomery product a * b * 2^-261 mod P.  Limbs of a and b < 2^30 (or one of them carried and the
// other < 2^31.5, see above); result loose, < 1.5 P for a * b < 64 P^2.
template <class F> HD Fe29<F> f29_mul(const Fe29<F> &a, const Fe29<F> &b) {
    F29_ASSERT(F29_GET(a) * F29_GET(b) <= F29_RP_OVER_P);
    F29_CHECK_COLUMNS(a, b);
    uint64_t c[18];
#pragma unroll
    for (int k = 0; k < 18; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)a.l[i] * b.l[j];
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
    F29_SET(r, F29_GET(a) * F29_GET(b) / 168.9 + 1.0);
    return r;
}
// (a * b + c * d) * 2^-261 mod P with ONE Montgomery reduction: the two products share the 18
// column accumulators.  Operands are carried (limbs < 2^29 + 8, which every carrying f29 function
// returns): a column then holds at most 18 products < 2^58.01 plus 9 reduction products < 2^58 and a
// carry, < 2^62.8 -- which leaves room for ONE uncarried operand with limbs < 2^30 (f29_sub_nc of a
// zero minuend; the test build checks the columns of every call).  Result loose, < (a b + c d) / (2^261 P) + 1 in multiples of P.
template <class F> HD Fe29<F> f29_mul2_add(const Fe29<F> &a, const Fe29<F> &b, const Fe29<F> &c2, const Fe29<F> &d) {
    F29_ASSERT(F29_GET(a) * F29_GET(b) + F29_GET(c2) * F29_GET(d) <= F29_RP_OVER_P);
    F29_CHECK_COLUMNS2(a, b, c2, d);
    uint64_t c[18];
#pragma unroll
    for (int k = 0; k < 18; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)a.l[i] * b.l[j];
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)c2.l[i] * d.l[j];
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
    F29_SET(r, (F29_GET(a) * F29_GET(b) + F29_GET(c2) * F29_GET(d)) / 168.9 + 1.0);
    return r;
}
// Montgomery reduction alone: a * 2^-261 mod P -- half a multiplication (81 reduction products, no a * b).  Limbs of a
// up to 2^32 (a column holds one limb, nine reduction products < 2^58 and a carry); result loose, < a / (2^261 P) + 1.
// The NTT's last pass ends with it: the pass before has multiplied every element by 2^261 (and the ifft scale) inside
// its post-twiddle, so the value that leaves is exact and < 2 P.
template <class F> HD Fe29<F> f29_redc(const Fe29<F> &a) {
    uint64_t c[18];
#pragma unroll
    for (int k = 0; k < 9; k++) { c[k] = a.l[k]; c[k + 9] = 0; }
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
    F29_SET(r, F29_GET(a) / 168.9 + 1.0);
    return r;
}
template <class F> HD Fe29<F> f29_sqr(const Fe29<F> &a) {
    F29_ASSERT(F29_GET(a) * F29_GET(a) <= F29_RP_OVER_P);
    uint64_t c[18];
#pragma unroll
    for (int k = 0; k < 18; k++) c[k] = 0;
    uint32_t d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = a.l[i] << 1;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        c[2 * i] += (uint64_t)a.l[i] * a.l[i];
#pragma unroll
        for (int j = i + 1; j < 9; j++) c[i + j] += (uint64_t)a.l[i] * d[j];
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
    F29_SET(r, F29_GET(a) * F29_GET(a) / 168.9 + 1.0);
    return r;
}

// P[0]^-1 mod 2^29 (P is odd): Newton iteration, five doublings of the precision from 3 bits
constexpr uint32_t f29_inv_mod_2_29(uint32_t p0) {
    uint32_t x = p0;                                   // p0 * p0 = 1 mod 8
    for (int i = 0; i < 5; i++) x *= 2u - p0 * x;
    return x & M29;
}
// x == 0 mod P for a loose x < KMAX * P.  Fast reject on the low limb (carries only move upward, so
// the low 29 bits are already final): x = k P needs l[0] = k P[0] mod 2^29, i.e. l[0] * P[0]^-1 = k <= KMAX
// -- one multiplication instead of a comparison per multiple.  Full compare otherwise.
template <int KMAX, class F> HD bool f29_is_zero_mod_p(const Fe29<F> &a) {
    F29_ASSERT(F29_GET(a) <= (double)KMAX);
    constexpr uint32_t PINV = f29_inv_mod_2_29(F::P[0]);
    static_assert(((F::P[0] * PINV) & M29) == 1u, "P[0]^-1 mod 2^29");
    const bool maybe = ((a.l[0] * PINV) & M29) <= (uint32_t)KMAX;
    if (!maybe) return false;
    // exact: propagate carries, then compare with k * P
    uint32_t n[9];
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t v = a.l[i] + carry;
        n[i] = v & M29;
        carry = v >> 29;
    }
    n[8] = a.l[8] + carry;
    for (uint32_t k = 0; k <= KMAX; k++) {
        uint64_t cy = 0;
        bool eq = true;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            cy += (uint64_t)k * F::P[i];
            uint32_t want = (i < 8) ? (uint32_t)(cy & M29) : (uint32_t)cy;
            cy >>= 29;
            eq &= (n[i] == want);
        }
        if (eq) return true;
    }
    return false;
}

// ---- boundary conversions -----------------------------------------------------------------------
// 8 x 32 saturated limbs (any 256-bit value) -> 9 x 29 limbs, no arithmetic on the value
template <class F> HD Fe29<F> f29_unpack(const Fe<typename F::Sat> &s) {
    Fe29<F> r;
    r.l[0] = s.l[0] & M29;
    r.l[1] = ((s.l[0] >> 29) | (s.l[1] << 3)) & M29;
    r.l[2] = ((s.l[1] >> 26) | (s.l[2] << 6)) & M29;
    r.l[3] = ((s.l[2] >> 23) | (s.l[3] << 9)) & M29;
    r.l[4] = ((s.l[3] >> 20) | (s.l[4] << 12)) & M29;
    r.l[5] = ((s.l[4] >> 17) | (s.l[5] << 15)) & M29;
    r.l[6] = ((s.l[5] >> 14) | (s.l[6] << 18)) & M29;
    r.l[7] = ((s.l[6] >> 11) | (s.l[7] << 21)) & M29;
    r.l[8] = s.l[7] >> 8;
    F29_SET(r, 6.0);   // any 256-bit integer is < 6 P; callers with canonical data tighten this
    return r;
}
// canonical (< P) saturated value, no arithmetic
template <class F> HD Fe29<F> f29_unpack_canonical(const Fe<typename F::Sat> &s) {
    Fe29<F> r = f29_unpack<F>(s);
    F29_SET(r, 1.0);
    return r;
}
// loose value < 2^256 -> 8 x 32 saturated limbs of the same integer (no modular reduction)
template <class F> HD Fe<typename F::Sat> f29_pack(const Fe29<F> &a) {
    uint32_t n[9];
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t v = a.l[i] + carry;
        n[i] = v & M29;
        carry = v >> 29;
    }
    n[8] = a.l[8] + carry;
    Fe<typename F::Sat> s;
    s.l[0] = n[0] | (n[1] << 29);
    s.l[1] = (n[1] >> 3) | (n[2] << 26);
    s.l[2] = (n[2] >> 6) | (n[3] << 23);
    s.l[3] = (n[3] >> 9) | (n[4] << 20);
    s.l[4] = (n[4] >> 12) | (n[5] << 17);
    s.l[5] = (n[5] >> 15) | (n[6] << 14);
    s.l[6] = (n[6] >> 18) | (n[7] << 11);
    s.l[7] = (n[7] >> 21) | (n[8] << 8);
    return s;
}
// the same for a value straight out of f29_mul / f29_mul2_add / f29_redc / f29_sqr: their limbs 0 .. 7 are masked to 29 bits, so
// there is nothing to carry (24 instructions less per element)
template <class F> HD Fe<typename F::Sat> f29_pack_product(const Fe29<F> &a) {
#pragma unroll
    for (int i = 0; i < 8; i++) F29_ASSERT(a.l[i] <= M29);
    Fe<typename F::Sat> s;
    s.l[0] = a.l[0] | (a.l[1] << 29);
    s.l[1] = (a.l[1] >> 3) | (a.l[2] << 26);
    s.l[2] = (a.l[2] >> 6) | (a.l[3] << 23);
    s.l[3] = (a.l[3] >> 9) | (a.l[4] << 20);
    s.l[4] = (a.l[4] >> 12) | (a.l[5] << 17);
    s.l[5] = (a.l[5] >> 15) | (a.l[6] << 14);
    s.l[6] = (a.l[6] >> 18) | (a.l[7] << 11);
    s.l[7] = (a.l[7] >> 21) | (a.l[8] << 8);
    return s;
}
// reference layout (x * 2^256, canonical, saturated) -> loose x * 2^261
template <class F> HD Fe29<F> f29_from_r256(const Fe<typename F::Sat> &s) {
    Fe29<F> k;
#pragma unroll
    for (int i = 0; i < 9; i++) k.l[i] = F::R256_TO_R261[i];
    F29_SET(k, 1.0);
    return f29_mul(f29_unpack_canonical<F>(s), k);
}
// loose x * 2^261 -> reference layout x * 2^256, canonical: multiply by 2^256 * 2^-261 ... i.e.
// Montgomery-multiply by the plain integer 2^256 mod P?  Simpler: f29_mul(a, 2^256 mod P) gives
// a * 2^256 * 2^-261 = x * 2^256.  The constant is the saturated R1 of the field, unpacked.
template <class F> HD Fe<typename F::Sat> f29_to_r256(const Fe29<F> &a) {
    Fe<typename F::Sat> r1;
#pragma unroll
    for (int i = 0; i < 8; i++) r1.l[i] = F::Sat::R1[i];
    Fe29<F> v = f29_mul(a, f29_unpack_canonical<F>(r1));     // loose, < 2 P
    Fe<typename F::Sat> s = f29_pack(v);                     // < 2 P < 2^255
    return reduce_once(s);
}
