// 254-bit prime-field arithmetic for gfx950: 8 x 32-bit little-endian limbs, Montgomery form
// with R = 2^256 -- byte-identical to the 4 x u64 limbs halo2curves keeps in memory, so the
// reference's buffers (src/commitment.rs:80 `&self.ck[..]`, `v: &[C::Scalar]`; src/fft.rs:51
// `a: &mut [G]`) are consumed and produced without conversion.
//
// The multiplier is CIOS on v_mad_u64_u32 (32 x 32 + 64 -> 64), one row of the multiplicand and
// one Montgomery reduction row fused per outer step.  Both moduli are < 2^254, so the running
// value never needs a ninth limb.  No MFMA: this is carry-chained integer work.
#pragma once
#include "platform.h"

struct FqP {   // bn256::Fq  (coordinates of BN256 G1, scalars of Grumpkin)
    static constexpr uint32_t P[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t N0 = 0xe4866389u;   // -P^-1 mod 2^32
    static constexpr uint32_t R1[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
};
struct FrP {   // bn256::Fr  (scalars of BN256 G1, coordinates of Grumpkin, NTT field)
    static constexpr uint32_t P[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t N0 = 0xefffffffu;
    static constexpr uint32_t R1[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
};

template <class FP> struct Fe {
    uint32_t l[8];
};

// ---- raw 256-bit helpers ------------------------------------------------------------------
template <class FP> HD bool fe_is_zero(const Fe<FP> &a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.l[i];
    return o == 0;
}
template <class FP> HD bool fe_eq(const Fe<FP> &a, const Fe<FP> &b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.l[i] ^ b.l[i];
    return o == 0;
}
template <class FP> HD Fe<FP> fe_zero() {
    Fe<FP> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = 0;
    return r;
}
template <class FP> HD Fe<FP> fe_one() {   // Montgomery 1
    Fe<FP> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = FP::R1[i];
    return r;
}
// r = a - P, returns borrow
template <class FP> HD uint32_t sub_p(Fe<FP> &r, const Fe<FP> &a) {
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t d = (uint64_t)a.l[i] - FP::P[i] - br;
        r.l[i] = (uint32_t)d;
        br = (d >> 32) & 1;
    }
    return (uint32_t)br;
}
// canonical reduce of a value in [0, 2P)
template <class FP> HD Fe<FP> reduce_once(const Fe<FP> &a) {
    Fe<FP> t;
    uint32_t br = sub_p(t, a);
    return br ? a : t;
}

template <class FP> HD Fe<FP> fe_add(const Fe<FP> &a, const Fe<FP> &b) {
    Fe<FP> s;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.l[i] + b.l[i];
        s.l[i] = (uint32_t)c;
        c >>= 32;
    }
    return reduce_once(s);   // a + b < 2P < 2^255: no carry out of limb 7
}
template <class FP> HD Fe<FP> fe_sub(const Fe<FP> &a, const Fe<FP> &b) {
    Fe<FP> d;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t v = (uint64_t)a.l[i] - b.l[i] - br;
        d.l[i] = (uint32_t)v;
        br = (v >> 32) & 1;
    }
    uint32_t mask = (uint32_t)0 - (uint32_t)br;   // add P back when a < b
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)d.l[i] + (FP::P[i] & mask);
        d.l[i] = (uint32_t)c;
        c >>= 32;
    }
    return d;
}
template <class FP> HD Fe<FP> fe_neg(const Fe<FP> &a) {
    return fe_sub(fe_zero<FP>(), a);
}
template <class FP> HD Fe<FP> fe_dbl(const Fe<FP> &a) { return fe_add(a, a); }

// Montgomery product a * b * 2^-256 mod P, canonical output.
template <class FP> HD Fe<FP> fe_mul(const Fe<FP> &a, const Fe<FP> &b) {
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t v = (uint64_t)a.l[0] * b.l[i] + t[0];
        uint32_t m = (uint32_t)v * FP::N0;
        uint64_t w = (uint64_t)m * FP::P[0] + (uint32_t)v;
        uint64_t ca = v >> 32, cm = w >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            v = (uint64_t)a.l[j] * b.l[i] + t[j] + ca;
            ca = v >> 32;
            w = (uint64_t)m * FP::P[j] + (uint32_t)v + cm;
            t[j - 1] = (uint32_t)w;
            cm = w >> 32;
        }
        t[7] = (uint32_t)(ca + cm);
    }
    Fe<FP> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = t[i];
    return reduce_once(r);
}
template <class FP> HD Fe<FP> fe_sqr(const Fe<FP> &a) { return fe_mul(a, a); }

// Montgomery reduction only: a * 2^-256 mod P (leaves Montgomery form: canonical integer out)
template <class FP> HD Fe<FP> fe_from_mont(const Fe<FP> &a) {
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = a.l[i];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t m = t[0] * FP::N0;
        uint64_t w = (uint64_t)m * FP::P[0] + t[0];
        uint64_t cm = w >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            w = (uint64_t)m * FP::P[j] + t[j] + cm;
            t[j - 1] = (uint32_t)w;
            cm = w >> 32;
        }
        t[7] = (uint32_t)cm;
    }
    Fe<FP> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = t[i];
    return reduce_once(r);
}
template <class FP> HD Fe<FP> fe_to_mont(const Fe<FP> &a) {
    Fe<FP> r2;
#pragma unroll
    for (int i = 0; i < 8; i++) r2.l[i] = FP::R2[i];
    return fe_mul(a, r2);
}
// a^(P-2): Fermat inverse (0 -> 0).  Serial; used only by generators and tests on device.
template <class FP> HD Fe<FP> fe_inv(const Fe<FP> &a) {
    Fe<FP> acc = fe_one<FP>(), base = a;
    for (int i = 0; i < 8; i++) {
        uint32_t e = FP::P[i] - (i == 0 ? 2u : 0u);   // P[0] >= 2 for both moduli: no borrow
        for (int k = 0; k < 32; k++) {
            if ((e >> k) & 1) acc = fe_mul(acc, base);
            base = fe_sqr(base);
        }
    }
    return acc;
}
template <class FP> HD Fe<FP> fe_pow_u64(const Fe<FP> &a, uint64_t e) {
    Fe<FP> acc = fe_one<FP>(), base = a;
    while (e) {
        if (e & 1) acc = fe_mul(acc, base);
        base = fe_sqr(base);
        e >>= 1;
    }
    return acc;
}

// 16-byte vector type for coalesced 128-bit loads/stores of limb quads
struct alignas(16) U4 {
    uint32_t x, y, z, w;
};
struct alignas(8) U2 {
    uint32_t x, y;
};
template <class FP> HD Fe<FP> fe_load(const void *p) {
    const U4 *q = reinterpret_cast<const U4 *>(p);
    U4 a = q[0], b = q[1];
    Fe<FP> r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
template <class FP> HD void fe_store(void *p, const Fe<FP> &v) {
    U4 *q = reinterpret_cast<U4 *>(p);
    q[0] = U4{v.l[0], v.l[1], v.l[2], v.l[3]};
    q[1] = U4{v.l[4], v.l[5], v.l[6], v.l[7]};
}
