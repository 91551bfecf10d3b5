// Short-Weierstrass a = 0 curves (BN256 G1: y^2 = x^3 + 3 over Fq; Grumpkin: y^2 = x^3 - 17
// over Fr) in extended Jacobian "XYZZ" coordinates: x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2.
// Bucket accumulators are XYZZ (128 B); bases stay affine (64 B, identity = (0,0), the encoding
// the reference absorbs into its random oracle, src/poseidon/poseidon_hash.rs:137-140).
// The curve constant b never enters add/double for a = 0, so one template serves both curves.
#pragma once
#include "field.cuh"

template <class FP> struct Aff {
    Fe<FP> x, y;
};
template <class FP> struct Xyzz {
    Fe<FP> x, y, zz, zzz;   // identity: zz == 0
};

template <class FP> HD bool aff_is_identity(const Aff<FP> &p) { return fe_is_zero(p.x) && fe_is_zero(p.y); }
template <class FP> HD bool xyzz_is_identity(const Xyzz<FP> &p) { return fe_is_zero(p.zz); }
template <class FP> HD Xyzz<FP> xyzz_identity() {
    Xyzz<FP> r;
    r.x = fe_zero<FP>(); r.y = fe_zero<FP>(); r.zz = fe_zero<FP>(); r.zzz = fe_zero<FP>();
    return r;
}
template <class FP> HD Xyzz<FP> xyzz_from_affine(const Aff<FP> &p) {
    if (aff_is_identity(p)) return xyzz_identity<FP>();
    Xyzz<FP> r;
    r.x = p.x; r.y = p.y; r.zz = fe_one<FP>(); r.zzz = fe_one<FP>();
    return r;
}
template <class FP> HD Aff<FP> aff_neg(const Aff<FP> &p) {
    Aff<FP> r;
    r.x = p.x; r.y = fe_neg(p.y);
    return r;
}

// 2 * (affine, not identity): mdbl-2008-s-1
template <class FP> HD Xyzz<FP> xyzz_double_affine(const Aff<FP> &p) {
    Xyzz<FP> r;
    Fe<FP> u = fe_dbl(p.y);
    if (fe_is_zero(u)) return xyzz_identity<FP>();   // order-2 point (none on these curves)
    Fe<FP> v = fe_sqr(u);
    Fe<FP> w = fe_mul(u, v);
    Fe<FP> s = fe_mul(p.x, v);
    Fe<FP> xx = fe_sqr(p.x);
    Fe<FP> m = fe_add(fe_dbl(xx), xx);
    r.x = fe_sub(fe_sqr(m), fe_dbl(s));
    r.y = fe_sub(fe_mul(m, fe_sub(s, r.x)), fe_mul(w, p.y));
    r.zz = v; r.zzz = w;
    return r;
}
// 2 * XYZZ: dbl-2008-s-1
template <class FP> HD Xyzz<FP> xyzz_double(const Xyzz<FP> &p) {
    if (xyzz_is_identity(p)) return p;
    Xyzz<FP> r;
    Fe<FP> u = fe_dbl(p.y);
    if (fe_is_zero(u)) return xyzz_identity<FP>();
    Fe<FP> v = fe_sqr(u);
    Fe<FP> w = fe_mul(u, v);
    Fe<FP> s = fe_mul(p.x, v);
    Fe<FP> xx = fe_sqr(p.x);
    Fe<FP> m = fe_add(fe_dbl(xx), xx);
    r.x = fe_sub(fe_sqr(m), fe_dbl(s));
    r.y = fe_sub(fe_mul(m, fe_sub(s, r.x)), fe_mul(w, p.y));
    r.zz = fe_mul(v, p.zz); r.zzz = fe_mul(w, p.zzz);
    return r;
}

// acc += q (q affine): madd-2008-s, 8M + 2S on the common path
template <class FP> HD void xyzz_add_affine(Xyzz<FP> &acc, const Aff<FP> &q) {
    if (aff_is_identity(q)) return;
    if (xyzz_is_identity(acc)) {
        acc.x = q.x; acc.y = q.y; acc.zz = fe_one<FP>(); acc.zzz = fe_one<FP>();
        return;
    }
    Fe<FP> u2 = fe_mul(q.x, acc.zz);
    Fe<FP> s2 = fe_mul(q.y, acc.zzz);
    Fe<FP> p = fe_sub(u2, acc.x);
    Fe<FP> r = fe_sub(s2, acc.y);
    if (fe_is_zero(p)) {
        if (fe_is_zero(r)) acc = xyzz_double_affine(q);   // same point
        else acc = xyzz_identity<FP>();                    // opposite points
        return;
    }
    Fe<FP> pp = fe_sqr(p);
    Fe<FP> ppp = fe_mul(p, pp);
    Fe<FP> qq = fe_mul(acc.x, pp);
    Fe<FP> x3 = fe_sub(fe_sub(fe_sqr(r), ppp), fe_dbl(qq));
    Fe<FP> y3 = fe_sub(fe_mul(r, fe_sub(qq, x3)), fe_mul(acc.y, ppp));
    acc.x = x3; acc.y = y3;
    acc.zz = fe_mul(acc.zz, pp);
    acc.zzz = fe_mul(acc.zzz, ppp);
}

// acc += q (both XYZZ): add-2008-s, 12M + 2S
template <class FP> HD void xyzz_add(Xyzz<FP> &acc, const Xyzz<FP> &q) {
    if (xyzz_is_identity(q)) return;
    if (xyzz_is_identity(acc)) { acc = q; return; }
    Fe<FP> u1 = fe_mul(acc.x, q.zz);
    Fe<FP> u2 = fe_mul(q.x, acc.zz);
    Fe<FP> s1 = fe_mul(acc.y, q.zzz);
    Fe<FP> s2 = fe_mul(q.y, acc.zzz);
    Fe<FP> p = fe_sub(u2, u1);
    Fe<FP> r = fe_sub(s2, s1);
    if (fe_is_zero(p)) {
        if (fe_is_zero(r)) acc = xyzz_double(acc);
        else acc = xyzz_identity<FP>();
        return;
    }
    Fe<FP> pp = fe_sqr(p);
    Fe<FP> ppp = fe_mul(p, pp);
    Fe<FP> qq = fe_mul(u1, pp);
    Fe<FP> x3 = fe_sub(fe_sub(fe_sqr(r), ppp), fe_dbl(qq));
    Fe<FP> y3 = fe_sub(fe_mul(r, fe_sub(qq, x3)), fe_mul(s1, ppp));
    acc.x = x3; acc.y = y3;
    acc.zz = fe_mul(fe_mul(acc.zz, q.zz), pp);
    acc.zzz = fe_mul(fe_mul(acc.zzz, q.zzz), ppp);
}

template <class FP> HD Aff<FP> xyzz_to_affine(const Xyzz<FP> &p) {   // one inversion: serial use only
    Aff<FP> r;
    if (xyzz_is_identity(p)) { r.x = fe_zero<FP>(); r.y = fe_zero<FP>(); return r; }
    Fe<FP> zi = fe_inv(p.zzz);                   // 1/ZZZ
    Fe<FP> zz_inv = fe_sqr(fe_mul(zi, p.zz));    // (ZZ/ZZZ)^2 = 1/ZZ  (ZZ = z^2, ZZZ = z^3)
    r.x = fe_mul(p.x, zz_inv);
    r.y = fe_mul(p.y, zi);
    return r;
}

template <class FP> HD Aff<FP> aff_load(const void *p) {
    Aff<FP> r;
    r.x = fe_load<FP>(p);
    r.y = fe_load<FP>(reinterpret_cast<const unsigned char *>(p) + 32);
    return r;
}
template <class FP> HD void aff_store(void *p, const Aff<FP> &v) {
    fe_store(p, v.x);
    fe_store(reinterpret_cast<unsigned char *>(p) + 32, v.y);
}
template <class FP> HD Xyzz<FP> xyzz_load(const void *p) {
    const unsigned char *b = reinterpret_cast<const unsigned char *>(p);
    Xyzz<FP> r;
    r.x = fe_load<FP>(b); r.y = fe_load<FP>(b + 32); r.zz = fe_load<FP>(b + 64); r.zzz = fe_load<FP>(b + 96);
    return r;
}
template <class FP> HD void xyzz_store(void *p, const Xyzz<FP> &v) {
    unsigned char *b = reinterpret_cast<unsigned char *>(p);
    fe_store(b, v.x); fe_store(b + 32, v.y); fe_store(b + 64, v.zz); fe_store(b + 96, v.zzz);
}
