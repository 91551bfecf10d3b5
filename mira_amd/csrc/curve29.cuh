// XYZZ point arithmetic over the 9 x 29-bit field (field29.cuh) for the MSM kernels.
//
// Same formulas as curve.cuh (madd-2008-s, add-2008-s, dbl-2008-s-1; a = 0).  Values are "loose":
// each coordinate carries a proven bound in multiples of P, and every subtraction names the
// multiple K of P it adds (f29_sub<K>), chosen from those bounds:
//
//   invariant of every stored point:   X < 9 P,  Y < 5 P,  ZZ < 2 P,  ZZZ < 2 P
//   affine bases (canonical in memory): x, y < 1.01 P
//
// A product a * b stays below 2 P whenever bound(a) * bound(b) <= 168 (= 2^261 / P).  The host
// test build (F29_TRACK) re-derives every bound and asserts every precondition on each call.
#pragma once
#include "field29.cuh"

template <class F> struct Aff29 {
    Fe29<F> x, y;   // identity: both literal zero
};
template <class F> struct Xyzz29 {
    Fe29<F> x, y, zz, zzz;   // identity: zz literal zero
};
static constexpr int XYZZ29_BYTES = 144;   // 4 coordinates x 9 limbs x 4 B, 16-byte aligned
static constexpr int AFF29_BYTES = 64;     // bases are stored canonical and saturated (8 x 32) in R' = 2^261 form

template <class F> HD bool xyzz29_is_identity(const Xyzz29<F> &p) { return f29_is_literal_zero(p.zz); }
template <class F> HD Xyzz29<F> xyzz29_identity() {
    Xyzz29<F> r;
    r.x = f29_zero<F>(); r.y = f29_zero<F>(); r.zz = f29_zero<F>(); r.zzz = f29_zero<F>();
    return r;
}
template <class F> HD bool aff29_is_identity(const Aff29<F> &p) { return f29_is_literal_zero(p.x) && f29_is_literal_zero(p.y); }

// 2 * (affine, not identity): mdbl-2008-s-1
template <class F> HD Xyzz29<F> xyzz29_double_affine(const Aff29<F> &p) {
    Xyzz29<F> r;
    Fe29<F> u = f29_dbl(p.y);                                   // < 2.1
    Fe29<F> v = f29_sqr(u);                                     // < 2
    Fe29<F> w = f29_mul(u, v);
    Fe29<F> s = f29_mul(p.x, v);
    Fe29<F> m = f29_triple(f29_sqr(p.x));                       // < 6
    r.x = f29_sub<5>(f29_sqr(m), f29_dbl(s));                   // 2S < 4  -> X < 7
    r.y = f29_mul2_add(m, f29_sub<8>(s, r.x), w, f29_neg<3>(p.y));       // (S - X) < 10, * 6 = 60 ; 2 * 3 (a negated base has y < 2 P) -> Y < 2
    r.zz = v; r.zzz = w;
    return r;
}
// 2 * XYZZ: dbl-2008-s-1
template <class F> HD Xyzz29<F> xyzz29_double(const Xyzz29<F> &p) {
    if (xyzz29_is_identity(p)) return p;
    Xyzz29<F> r;
    Fe29<F> u = f29_dbl(p.y);                                   // < 10
    Fe29<F> v = f29_sqr(u);                                     // 100 <= 168
    Fe29<F> w = f29_mul(u, v);                                  // 20
    Fe29<F> s = f29_mul(p.x, v);                                // 18
    Fe29<F> m = f29_triple(f29_sqr(p.x));                       // 81 <= 168 ; m < 6
    r.x = f29_sub<5>(f29_sqr(m), f29_dbl(s));                   // < 7
    r.y = f29_mul2_add(m, f29_sub<8>(s, r.x), w, f29_neg<6>(p.y));       // 6 * 10 ; 2 * 6 -> < 2
    r.zz = f29_mul(v, p.zz); r.zzz = f29_mul(w, p.zzz);
    return r;
}

// acc += q (q affine, canonical): madd-2008-s, 8M + 2S on the common path
template <class F> HD void xyzz29_add_affine(Xyzz29<F> &acc, const Aff29<F> &q) {
    if (aff29_is_identity(q)) return;
    if (xyzz29_is_identity(acc)) {
        acc.x = q.x; acc.y = q.y; acc.zz = f29_one<F>(); acc.zzz = f29_one<F>();
        return;
    }
    Fe29<F> u2 = f29_mul(q.x, acc.zz);                          // < 2
    Fe29<F> s2 = f29_mul(q.y, acc.zzz);                         // < 2
    Fe29<F> p = f29_sub<10>(u2, acc.x);                         // X1 < 9  -> P < 12
    Fe29<F> r = f29_sub<6>(s2, acc.y);                          // Y1 < 5  -> R < 8
    if (f29_is_zero_mod_p<12>(p)) {
        if (f29_is_zero_mod_p<8>(r)) acc = xyzz29_double_affine(q);   // same point
        else acc = xyzz29_identity<F>();                            // opposite points
        return;
    }
    Fe29<F> pp = f29_sqr(p);                                    // 144 <= 168
    Fe29<F> ppp = f29_mul(p, pp);                               // 24
    Fe29<F> qq = f29_mul(acc.x, pp);                            // 18
    Fe29<F> x3 = f29_sub_b_2c<7>(f29_sqr(r), ppp, qq);                     // 64 ; PPP + 2Q < 6 -> X3 < 9
    // R (Q - X3) - Y1 PPP in one reduction: 8 * 12 + 7 * 2 = 110 -> Y3 < 2.  -Y1 without a carry pass
    // (Y1 is a multiplier result, an unpacked value or -- a bucket sum the fix-up wrote, k_accumulate<F, true> --
    // a carried one: limbs 0..7 < 2^29 + 8, which f29_sub_nc's bias covers, static_assert there): its limbs stay
    // below 2^30 + 8, and the shared columns hold 9 * 2^58 (R, Q - X3 carried) + 9 * (2^30 + 8) (2^29 + 8) + the
    // reduction's 9 * 2^58 < 2^64 (the test build checks the columns of every call)
    Fe29<F> y3 = f29_mul2_add(r, f29_sub<10>(qq, x3), f29_sub_nc<7>(f29_zero<F>(), acc.y), ppp);
    acc.x = x3; acc.y = y3;
    acc.zz = f29_mul(acc.zz, pp);
    acc.zzz = f29_mul(acc.zzz, ppp);
}

// acc += q (both XYZZ): add-2008-s, 12M + 2S
template <class F> HD void xyzz29_add(Xyzz29<F> &acc, const Xyzz29<F> &q) {
    if (xyzz29_is_identity(q)) return;
    if (xyzz29_is_identity(acc)) { acc = q; return; }
    Fe29<F> u1 = f29_mul(acc.x, q.zz);                          // 18
    Fe29<F> u2 = f29_mul(q.x, acc.zz);
    Fe29<F> s1 = f29_mul(acc.y, q.zzz);                         // 10
    Fe29<F> s2 = f29_mul(q.y, acc.zzz);
    Fe29<F> p = f29_sub<3>(u2, u1);                             // < 5
    Fe29<F> r = f29_sub<3>(s2, s1);                             // < 5
    if (f29_is_zero_mod_p<5>(p)) {
        if (f29_is_zero_mod_p<5>(r)) acc = xyzz29_double(acc);
        else acc = xyzz29_identity<F>();
        return;
    }
    Fe29<F> pp = f29_sqr(p);
    Fe29<F> ppp = f29_mul(p, pp);
    Fe29<F> qq = f29_mul(u1, pp);
    Fe29<F> x3 = f29_sub_b_2c<7>(f29_sqr(r), ppp, qq);                     // < 9
    Fe29<F> y3 = f29_mul2_add(r, f29_sub<10>(qq, x3), f29_sub_nc<3>(f29_zero<F>(), s1), ppp);   // 5 * 12 + 3 * 2 -> < 2 (-S1 uncarried: S1 is a multiplier result)
    acc.x = x3; acc.y = y3;
    acc.zz = f29_mul(f29_mul(acc.zz, q.zz), pp);
    acc.zzz = f29_mul(f29_mul(acc.zzz, q.zzz), ppp);
}

// ---- memory formats -------------------------------------------------------------------------------
// Bases: 64 B per point, canonical saturated limbs of x * 2^261 and y * 2^261 (converted once when
// the key is registered); identity = all zero.  `negate` folds the sign of a signed digit in.
template <class F> HD Aff29<F> aff29_load(const void *p, bool negate) {
    Aff29<F> r;
    r.x = f29_unpack_canonical<F>(fe_load<typename F::Sat>(p));
    r.y = f29_unpack_canonical<F>(fe_load<typename F::Sat>(reinterpret_cast<const unsigned char *>(p) + 32));
    if (negate && !f29_is_literal_zero(r.y)) r.y = f29_neg<2>(r.y);   // 2P - y < 2 P ... stays an exact negation mod P
    return r;
}
// (the negation of a base is taken on the saturated words, P - y by one borrow chain, before they are
// unpacked: 16 instructions against ~60 for a biased 29-bit-limb subtraction with its carry pass;
// y = 0 -- the identity (0, 0) -- stays literally zero)
template <class F> HD Aff29<F> aff29_from_raw(const U4 &a, const U4 &b, const U4 &c, const U4 &d, bool negate) {
    using S = typename F::Sat;
    Fe<S> x, y;
    x.l[0] = a.x; x.l[1] = a.y; x.l[2] = a.z; x.l[3] = a.w; x.l[4] = b.x; x.l[5] = b.y; x.l[6] = b.z; x.l[7] = b.w;
    y.l[0] = c.x; y.l[1] = c.y; y.l[2] = c.z; y.l[3] = c.w; y.l[4] = d.x; y.l[5] = d.y; y.l[6] = d.z; y.l[7] = d.w;
    Fe<S> ny;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t v = (uint64_t)S::P[i] - y.l[i] - br;
        ny.l[i] = (uint32_t)v;
        br = (v >> 32) & 1;
    }
    const bool flip = negate && !fe_is_zero(y);
#pragma unroll
    for (int i = 0; i < 8; i++) y.l[i] = flip ? ny.l[i] : y.l[i];
    Aff29<F> r;
    r.x = f29_unpack_canonical<F>(x);
    r.y = f29_unpack_canonical<F>(y);
    return r;
}
// XYZZ partial sums: 4 x 9 raw loose limbs = 144 B
template <class F> HD void f29_store_raw(void *p, const Fe29<F> &v) {
    uint32_t *q = reinterpret_cast<uint32_t *>(p);
#pragma unroll
    for (int i = 0; i < 9; i++) q[i] = v.l[i];
}
template <class F> HD Fe29<F> f29_load_raw(const void *p, double bound) {
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
    Fe29<F> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = q[i];
    F29_SET(r, bound);
    (void)bound;
    return r;
}
template <class F> HD void xyzz29_store(void *p, const Xyzz29<F> &v) {
    F29_ASSERT(F29_GET(v.x) <= 9.0 && F29_GET(v.y) <= 5.0 && F29_GET(v.zz) <= 2.0 && F29_GET(v.zzz) <= 2.0);
    // 36 words as 9 x 16-byte stores
    uint32_t w[36];
#pragma unroll
    for (int i = 0; i < 9; i++) { w[i] = v.x.l[i]; w[9 + i] = v.y.l[i]; w[18 + i] = v.zz.l[i]; w[27 + i] = v.zzz.l[i]; }
    U4 *q = reinterpret_cast<U4 *>(p);
#pragma unroll
    for (int i = 0; i < 9; i++) q[i] = U4{w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]};
}
template <class F> HD Xyzz29<F> xyzz29_load(const void *p) {
    const U4 *q = reinterpret_cast<const U4 *>(p);
    uint32_t w[36];
#pragma unroll
    for (int i = 0; i < 9; i++) { U4 t = q[i]; w[4 * i] = t.x; w[4 * i + 1] = t.y; w[4 * i + 2] = t.z; w[4 * i + 3] = t.w; }
    Xyzz29<F> r;
#pragma unroll
    for (int i = 0; i < 9; i++) { r.x.l[i] = w[i]; r.y.l[i] = w[9 + i]; r.zz.l[i] = w[18 + i]; r.zzz.l[i] = w[27 + i]; }
    F29_SET(r.x, 9.0); F29_SET(r.y, 5.0); F29_SET(r.zz, 2.0); F29_SET(r.zzz, 2.0);
    return r;
}
// Final export of a point: X, Y, ZZ, ZZZ as canonical saturated limbs in the reference's
// R = 2^256 Montgomery form (128 B), for the host epilogue.
template <class F> HD void xyzz29_export_r256(void *p, const Xyzz29<F> &v) {
    unsigned char *b = reinterpret_cast<unsigned char *>(p);
    if (xyzz29_is_identity(v)) {
#pragma unroll
        for (int i = 0; i < 8; i++) reinterpret_cast<U4 *>(b)[i] = U4{0, 0, 0, 0};
        return;
    }
    fe_store(b, f29_to_r256(v.x)); fe_store(b + 32, f29_to_r256(v.y));
    fe_store(b + 64, f29_to_r256(v.zz)); fe_store(b + 96, f29_to_r256(v.zzz));
}
