// Host-side (x86-64) field and XYZZ arithmetic used by the library's fixed-size epilogue:
// Horner over the W window sums that come back from the GPU (<= 32 points), the single
// inversion of Curve::to_affine() (reference src/commitment.rs:80), the combine of per-GPU
// partial window sums, and the derivation of NTT roots (reference src/fft.rs:12-27).
// O(1) work per call, independent of n; the MSM and NTT themselves run only on the GPU.
#pragma once
#include <cstdint>
#include <cstring>

#include "field.cuh"

namespace hostf {
typedef unsigned __int128 u128;

template <class FP> struct HFe {
    uint64_t l[4];
};
template <class FP> inline uint64_t P64(int i) { return (uint64_t)FP::P[2 * i] | ((uint64_t)FP::P[2 * i + 1] << 32); }
template <class FP> inline uint64_t N064() {
    uint64_t p0 = P64<FP>(0), inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - p0 * inv;
    return (uint64_t)0 - inv;
}
template <class FP> inline HFe<FP> zero() { HFe<FP> r; memset(&r, 0, sizeof r); return r; }
template <class FP> inline HFe<FP> one() {
    HFe<FP> r;
    for (int i = 0; i < 4; i++) r.l[i] = (uint64_t)FP::R1[2 * i] | ((uint64_t)FP::R1[2 * i + 1] << 32);
    return r;
}
template <class FP> inline HFe<FP> r2() {
    HFe<FP> r;
    for (int i = 0; i < 4; i++) r.l[i] = (uint64_t)FP::R2[2 * i] | ((uint64_t)FP::R2[2 * i + 1] << 32);
    return r;
}
template <class FP> inline bool is_zero(const HFe<FP> &a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
template <class FP> inline bool geq_p(const HFe<FP> &a) {
    for (int i = 3; i >= 0; i--) {
        uint64_t p = P64<FP>(i);
        if (a.l[i] > p) return true;
        if (a.l[i] < p) return false;
    }
    return true;
}
template <class FP> inline void sub_p(HFe<FP> &a) {
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a.l[i] - P64<FP>(i) - br;
        a.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1;
    }
}
template <class FP> inline HFe<FP> add(const HFe<FP> &a, const HFe<FP> &b) {
    HFe<FP> r; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (geq_p(r)) sub_p(r);
    return r;
}
template <class FP> inline HFe<FP> sub(const HFe<FP> &a, const HFe<FP> &b) {
    HFe<FP> r; uint64_t br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a.l[i] - b.l[i] - br;
        r.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1;
    }
    if (br) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) { c += (u128)r.l[i] + P64<FP>(i); r.l[i] = (uint64_t)c; c >>= 64; }
    }
    return r;
}
template <class FP> inline HFe<FP> dbl(const HFe<FP> &a) { return add(a, a); }
// Montgomery product, CIOS with the two carry chains of a row fused ("no-carry" form: valid because
// both moduli leave the top two bits of their top limb clear, so a row's running value fits 4 limbs
// plus one carry word that the next row absorbs).  ~1.6x the speed of the plain five-limb CIOS; the
// Horner epilogue of every commit is ~2 500 of these.
template <class FP> inline HFe<FP> mul(const HFe<FP> &a, const HFe<FP> &b) {
    static const uint64_t n0 = N064<FP>();
    static const uint64_t p0 = P64<FP>(0), p1 = P64<FP>(1), p2 = P64<FP>(2), p3 = P64<FP>(3);
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint64_t bi = b.l[i];
        u128 A = (u128)a.l[0] * bi + t0;
        const uint64_t lo = (uint64_t)A, m = lo * n0;
        u128 C = (u128)m * p0 + lo;                     // low word becomes zero
        A = (A >> 64) + (u128)a.l[1] * bi + t1;
        C = (C >> 64) + (u128)m * p1 + (uint64_t)A;
        t0 = (uint64_t)C;
        A = (A >> 64) + (u128)a.l[2] * bi + t2;
        C = (C >> 64) + (u128)m * p2 + (uint64_t)A;
        t1 = (uint64_t)C;
        A = (A >> 64) + (u128)a.l[3] * bi + t3;
        C = (C >> 64) + (u128)m * p3 + (uint64_t)A;
        t2 = (uint64_t)C;
        t3 = (uint64_t)(C >> 64) + (uint64_t)(A >> 64);
    }
    HFe<FP> r = {{t0, t1, t2, t3}};
    if (geq_p(r)) sub_p(r);
    return r;
}
template <class FP> inline HFe<FP> sqr(const HFe<FP> &a) { return mul(a, a); }   // (a dedicated square of 26 limb products measured no faster than this: 26.6 against 27.0 ns)
template <class FP> inline HFe<FP> to_mont(const HFe<FP> &a) { return mul(a, r2<FP>()); }
template <class FP> inline HFe<FP> from_u64(uint64_t v) { HFe<FP> r = {{v, 0, 0, 0}}; return to_mont(r); }
// a^e, e given as 4 plain u64 limbs: fixed four-bit windows from the top (256 squares, at most 64 + 14 products)
template <class FP> inline HFe<FP> pow(const HFe<FP> &a, const uint64_t e[4]) {
    HFe<FP> tab[16];
    tab[0] = one<FP>(); tab[1] = a;
    for (int i = 2; i < 16; i++) tab[i] = (i & 1) ? mul(tab[i - 1], a) : sqr(tab[i / 2]);
    HFe<FP> acc = one<FP>();
    bool started = false;
    for (int i = 63; i >= 0; i--) {
        const uint32_t nib = (uint32_t)(e[i / 16] >> (4 * (i % 16))) & 15u;
        if (started) { acc = sqr(acc); acc = sqr(acc); acc = sqr(acc); acc = sqr(acc); }
        if (nib) { acc = started ? mul(acc, tab[nib]) : tab[nib]; started = true; }
    }
    return acc;
}
template <class FP> inline HFe<FP> inv(const HFe<FP> &a) {
    uint64_t e[4];
    for (int i = 0; i < 4; i++) e[i] = P64<FP>(i);
    e[0] -= 2;   // P[0] >= 2, no borrow
    return pow(a, e);
}

template <class FP> struct HXyzz {
    HFe<FP> x, y, zz, zzz;
};
template <class FP> inline HXyzz<FP> identity() { HXyzz<FP> r; memset(&r, 0, sizeof r); return r; }
template <class FP> inline bool is_identity(const HXyzz<FP> &p) { return is_zero(p.zz); }
template <class FP> inline HXyzz<FP> dbl_pt(const HXyzz<FP> &p) {   // dbl-2008-s-1
    if (is_identity(p)) return p;
    HFe<FP> u = dbl(p.y);
    if (is_zero(u)) return identity<FP>();
    HFe<FP> v = sqr(u), w = mul(u, v), s = mul(p.x, v), xx = sqr(p.x), m = add(dbl(xx), xx);
    HXyzz<FP> r;
    r.x = sub(sqr(m), dbl(s));
    r.y = sub(mul(m, sub(s, r.x)), mul(w, p.y));
    r.zz = mul(v, p.zz); r.zzz = mul(w, p.zzz);
    return r;
}
template <class FP> inline HXyzz<FP> add_pt(const HXyzz<FP> &a, const HXyzz<FP> &b) {   // add-2008-s
    if (is_identity(b)) return a;
    if (is_identity(a)) return b;
    HFe<FP> u1 = mul(a.x, b.zz), u2 = mul(b.x, a.zz), s1 = mul(a.y, b.zzz), s2 = mul(b.y, a.zzz);
    HFe<FP> p = sub(u2, u1), r = sub(s2, s1);
    if (is_zero(p)) return is_zero(r) ? dbl_pt(a) : identity<FP>();
    HFe<FP> pp = sqr(p), ppp = mul(p, pp), q = mul(u1, pp);
    HXyzz<FP> o;
    o.x = sub(sub(sqr(r), ppp), dbl(q));
    o.y = sub(mul(r, sub(q, o.x)), mul(s1, ppp));
    o.zz = mul(mul(a.zz, b.zz), pp);
    o.zzz = mul(mul(a.zzz, b.zzz), ppp);
    return o;
}
// x || y as 8 u64 limbs, identity -> (0, 0)
template <class FP> inline void to_affine(const HXyzz<FP> &p, uint64_t out[8]) {
    if (is_identity(p)) { memset(out, 0, 64); return; }
    HFe<FP> zi = inv(p.zzz);
    HFe<FP> zzi = sqr(mul(zi, p.zz));
    HFe<FP> x = mul(p.x, zzi), y = mul(p.y, zi);
    memcpy(out, x.l, 32); memcpy(out + 4, y.l, 32);
}
}   // namespace hostf
