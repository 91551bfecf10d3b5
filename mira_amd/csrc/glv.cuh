// The GLV split (opt-in per key: mira_msm_precompute_ex(handle, MIRA_TABLE_GLV)).
//
// Both curves are y^2 = x^3 + b, so phi(x, y) = (beta x, y) with beta^3 = 1 in the base field is an endomorphism, and on the
// prime-order group it is multiplication by a cube root of unity lambda of the scalar field.  A scalar k is rewritten as
// k = k1 + k2 lambda with k1, k2 < 2^128, and  k P = k1 P + k2 phi(P):  a commit of n pairs becomes one of 2 n pairs with
// half-length scalars over the interleaved key [P_0, phi(P_0), P_1, phi(P_1), ...] (a second resident copy, 128 B per
// point).  The bucket additions stay what they were (2 n x 129 / c against n x 256 / c); the windows -- and with them the
// bucket reduction, the window sums and the host's chain of doublings -- halve.
//
// Decomposition (Gallant-Lambert-Vanstone; constants derived and checked against the oracle by tools/glv_constants.py):
// (a1, b1), (a2, b2) a reduced basis of {(x, y): x + y lambda = 0 mod r}, c1 = round(k b2 / r), c2 = round(-k b1 / r) taken as
// the high halves of k * G1, k * G2 (G_i = round(2^256 b / r)), (k1, k2) = (k, 0) - c1 (a1, b1) - c2 (a2, b2), |k_i| < 2^126.
// The halves are signed: the digits of |k_i| are written with the sign of k_i (the pipeline's digits are signed anyway: a negative
// digit subtracts the point).  c_i differs from the exact quotient by < 1/8 (G_i is rounded), so |k_i| < 0.625 (|v1| + |v2|)
// < 2^126.13: 127 bits, 128 with the carry of the signed digits -- and under c W = 128 (c = 8, 16) the last window then stays
// below 2^(c-1), so no carry leaves it.  A scalar below 2^126 is left as (k, 0): the zeros and small values of a witness vector
// keep their zero digits.
#pragma once
#include "field.cuh"

#include "glv_consts.h"

// limbs [LO, LO + NOUT) of a (NA limbs) * b (NB limbs), + 2^(32 LO - 1) first where ROUND (the rounding of c1, c2)
template <int NA, int NB, int LO, int NOUT, bool ROUND, class TB> HD void glv_mul_window(const uint32_t *a, const TB &b, uint32_t *out) {
    uint32_t t[NA + NB + 1];
#pragma unroll
    for (int i = 0; i < NA + NB + 1; i++) t[i] = 0;
    if (ROUND) t[LO - 1] = 0x80000000u;
#pragma unroll
    for (int i = 0; i < NA; i++) {
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const uint64_t x = (uint64_t)a[i] * b[j] + t[i + j] + carry;
            t[i + j] = (uint32_t)x;
            carry = x >> 32;
        }
#pragma unroll
        for (int j = i + NB; j < NA + NB + 1; j++) {           // the carry runs on (the rounding bit may already sit there)
            const uint64_t x = (uint64_t)t[j] + carry;
            t[j] = (uint32_t)x;
            carry = x >> 32;
        }
    }
#pragma unroll
    for (int i = 0; i < NOUT; i++) out[i] = LO + i < NA + NB + 1 ? t[LO + i] : 0u;
}
// acc (5 limbs, mod 2^160) +/-= a * b
template <bool SUB, class TB> HD void glv_muladd5(uint32_t *acc, const uint32_t *a, const TB &b) {
    uint32_t p[5];
    glv_mul_window<5, 5, 0, 5, false>(a, b, p);
    uint64_t c = SUB ? 1 : 0;                                   // acc - p = acc + ~p + 1
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const uint64_t x = (uint64_t)acc[i] + (SUB ? ~p[i] : p[i]) + c;
        acc[i] = (uint32_t)x;
        c = x >> 32;
    }
}
// canonical k (8 limbs) -> magnitudes h1, h2 (5 limbs each, < 2^126.13) and signs with k = s1 h1 + s2 h2 lambda mod r
template <class FS> HD void glv_split(const uint32_t *k, uint32_t *h1, uint32_t *h2, bool &neg1, bool &neg2) {
    using G = Glv<FS>;
    neg1 = neg2 = false;
    if ((k[4] | k[5] | k[6] | k[7]) == 0 && (k[3] >> 30) == 0) {  // k < 2^126: (k, 0)
#pragma unroll
        for (int i = 0; i < 4; i++) { h1[i] = k[i]; h2[i] = 0; }
        h1[4] = h2[4] = 0;
        return;
    }
    uint32_t c1[5], c2[5];
    glv_mul_window<8, 3, 8, 5, true>(k, G::G1, c1);             // round(k G1 / 2^256) < 2^65
    glv_mul_window<8, 5, 8, 5, true>(k, G::G2, c2);             // round(k G2 / 2^256) < 2^129
#pragma unroll
    for (int i = 0; i < 5; i++) { h1[i] = k[i]; h2[i] = 0; }     // two's complement mod 2^160 from here
    glv_muladd5<true>(h1, c1, G::A1);
    glv_muladd5<true>(h1, c2, G::A2);
    glv_muladd5<false>(h2, c1, G::NB1);
    glv_muladd5<true>(h2, c2, G::B2);
    auto magnitude = [](uint32_t *h) {
        const bool neg = (h[4] >> 31) != 0;
        if (neg) {
            uint64_t c = 1;
#pragma unroll
            for (int i = 0; i < 5; i++) { const uint64_t x = (uint64_t)(~h[i]) + c; h[i] = (uint32_t)x; c = x >> 32; }
        }
        return neg;
    };
    neg1 = magnitude(h1);
    neg2 = magnitude(h2);
}

// The interleaved key: out[2 i] = P_i, out[2 i + 1] = phi(P_i) = (beta x_i, y_i); resident layout in and out (canonical
// x * 2^261, y * 2^261; the identity (0, 0) stays the identity).  beta_r261: beta * 2^261, canonical saturated.
template <class F>
KERNEL void k_glv_bases(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst, uint64_t n, const unsigned char *__restrict__ beta_r261) {
    using S = typename F::Sat;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fe<S> x = fe_load<S>(src + i * 64), y = fe_load<S>(src + i * 64 + 32);
    const Fe<S> bx = reduce_once(f29_pack(f29_mul(f29_unpack_canonical<F>(x), f29_unpack_canonical<F>(fe_load<S>(beta_r261)))));
    fe_store(dst + i * 128, x);
    fe_store(dst + i * 128 + 32, y);
    fe_store(dst + i * 128 + 64, bx);
    fe_store(dst + i * 128 + 96, y);
}
