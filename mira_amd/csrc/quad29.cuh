// Quad-cooperative XYZZ arithmetic for the latency-bound tails of the MSM (fix-up trees, bucket
// reduction, window sums).
//
// Those kernels are chains of dependent general additions on a handful of waves: a lone wave
// issues one VALU instruction every ~4 cycles, so the 14 multiplications of add-2008-s take ~6.8 us
// however idle the GPU is (measured: 37 chained operations = 0.25 ms of a 0.78 ms commit of 2^17
// pairs).  The formula is only FOUR multiplications deep, though.  Here the four lanes of a DPP
// quad hold the same two points and each computes ONE of the (up to) four independent products of
// a level; the products are broadcast back with quad_perm moves (no LDS, no barrier).  An addition
// costs 4 multiplication latencies + ~150 moves instead of 14, a doubling 3 instead of 9.
//
// Contract of every function here: all four lanes of a quad pass bit-identical arguments and get
// bit-identical results (special cases -- identity, P = Q, P = -Q -- therefore branch quad-uniformly).
// QUAD_COOPERATIVE is the platform's statement that quad_perm exists (platform.h); the test-only
// host emulation (tests/emu/emu.h) sets it false and every lane computes all four products itself --
// the same values by the contract above.
#pragma once
#include "curve29.cuh"

template <class F> DEV Fe29<F> quad_pick(const Fe29<F> &x0, const Fe29<F> &x1, const Fe29<F> &x2, const Fe29<F> &x3) {
    const uint32_t q = quad_lane();
    Fe29<F> r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint32_t lo = (q & 1u) ? x1.l[i] : x0.l[i], hi = (q & 1u) ? x3.l[i] : x2.l[i];
        r.l[i] = (q & 2u) ? hi : lo;
    }
    F29_SET(r, F29_GET(x0) > F29_GET(x1) ? F29_GET(x0) : F29_GET(x1));
    if (F29_GET(x2) > F29_GET(r)) F29_SET(r, F29_GET(x2));
    if (F29_GET(x3) > F29_GET(r)) F29_SET(r, F29_GET(x3));
    return r;
}
// four independent products a_k * b_k, one per lane of the quad, all four results in every lane
template <class F>
DEV void quad_mul4(const Fe29<F> &a0, const Fe29<F> &b0, const Fe29<F> &a1, const Fe29<F> &b1, const Fe29<F> &a2, const Fe29<F> &b2,
                   const Fe29<F> &a3, const Fe29<F> &b3, Fe29<F> &r0, Fe29<F> &r1, Fe29<F> &r2, Fe29<F> &r3) {
    if constexpr (QUAD_COOPERATIVE) {
        F29_ASSERT(F29_GET(a0) * F29_GET(b0) <= F29_RP_OVER_P && F29_GET(a1) * F29_GET(b1) <= F29_RP_OVER_P);
        F29_ASSERT(F29_GET(a2) * F29_GET(b2) <= F29_RP_OVER_P && F29_GET(a3) * F29_GET(b3) <= F29_RP_OVER_P);
        Fe29<F> a = quad_pick(a0, a1, a2, a3), b = quad_pick(b0, b1, b2, b3);
        F29_SET(a, 1.0); F29_SET(b, 1.0);                          // checked per product above
        const Fe29<F> m = f29_mul(a, b);
#pragma unroll
        for (int i = 0; i < 9; i++) {
            r0.l[i] = quad_bcast<0>(m.l[i]); r1.l[i] = quad_bcast<1>(m.l[i]);
            r2.l[i] = quad_bcast<2>(m.l[i]); r3.l[i] = quad_bcast<3>(m.l[i]);
        }
    } else {
        r0 = f29_mul(a0, b0); r1 = f29_mul(a1, b1); r2 = f29_mul(a2, b2); r3 = f29_mul(a3, b3);
    }
    F29_SET(r0, F29_GET(a0) * F29_GET(b0) / 168.9 + 1.0); F29_SET(r1, F29_GET(a1) * F29_GET(b1) / 168.9 + 1.0);
    F29_SET(r2, F29_GET(a2) * F29_GET(b2) / 168.9 + 1.0); F29_SET(r3, F29_GET(a3) * F29_GET(b3) / 168.9 + 1.0);
}

// 2 * XYZZ: dbl-2008-s-1 in three levels (curve29.cuh's xyzz29_double lists the bounds)
template <class F> DEV Xyzz29<F> xyzz29_double_quad(const Xyzz29<F> &p) {
    if (xyzz29_is_identity(p)) return p;
    Xyzz29<F> r;
    const Fe29<F> u = f29_dbl(p.y);                               // < 10
    Fe29<F> v, xx, t0, t1;
    quad_mul4(u, u, p.x, p.x, u, u, p.x, p.x, v, xx, t0, t1);     // 100, 81
    const Fe29<F> m = f29_triple(xx);                             // < 6
    Fe29<F> w, s, mm;
    quad_mul4(u, v, p.x, v, v, p.zz, m, m, w, s, r.zz, mm);       // 20, 18, 4, 36
    r.x = f29_sub<5>(mm, f29_dbl(s));                             // < 7
    Fe29<F> ya, yb;
    quad_mul4(m, f29_sub<8>(s, r.x), w, f29_neg<6>(p.y), w, p.zzz, w, p.zzz, ya, yb, r.zzz, t0);   // 60, 12, 4
    r.y = f29_add(ya, yb);                                        // < 4
    return r;
}

// acc += q (both XYZZ): add-2008-s in four levels
template <class F> DEV void xyzz29_add_quad(Xyzz29<F> &acc, const Xyzz29<F> &q) {
    if (xyzz29_is_identity(q)) return;
    if (xyzz29_is_identity(acc)) { acc = q; return; }
    Fe29<F> u1, u2, s1, s2;
    quad_mul4(acc.x, q.zz, q.x, acc.zz, acc.y, q.zzz, q.y, acc.zzz, u1, u2, s1, s2);   // 18, 18, 10, 10
    const Fe29<F> p = f29_sub<3>(u2, u1);                         // < 5
    const Fe29<F> r = f29_sub<3>(s2, s1);                         // < 5
    if (f29_is_zero_mod_p<5>(p)) {
        if (f29_is_zero_mod_p<5>(r)) acc = xyzz29_double_quad(acc);
        else acc = xyzz29_identity<F>();
        return;
    }
    Fe29<F> pp, rr, zz12, zzz12;
    quad_mul4(p, p, r, r, acc.zz, q.zz, acc.zzz, q.zzz, pp, rr, zz12, zzz12);          // 25, 25, 4, 4
    Fe29<F> ppp, qq, t0;
    quad_mul4(p, pp, u1, pp, zz12, pp, zz12, pp, ppp, qq, acc.zz, t0);                 // 10, 4, 4
    acc.x = f29_sub_b_2c<7>(rr, ppp, qq);                         // < 9
    Fe29<F> ya, yb;
    quad_mul4(r, f29_sub<10>(qq, acc.x), f29_sub_nc<3>(f29_zero<F>(), s1), ppp, zzz12, ppp, zzz12, ppp, ya, yb, acc.zzz, t0);   // 60, 6, 4 (-S1 uncarried: limbs < 2^30 beside a carried operand)
    acc.y = f29_add(ya, yb);                                      // < 4
}
