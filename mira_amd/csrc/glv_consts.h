// Constants of the GLV split (glv.cuh), also read by the host side (capi.hip: beta for the endomorphism copy of a key, the
// number of scalar bits of a half).  Derived and checked against the oracle by tools/glv_constants.py.
#pragma once
#include "field.cuh"

template <class FS> struct Glv;
template <> struct Glv<FrP> {                                  // curve 0: scalars in bn256::Fr, lambda = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd
    static constexpr uint32_t G1[3] = {0xc7e0b3d7u, 0xd91d232eu, 0x00000002u}, G2[5] = {0x391eb18eu, 0x7a7bd9d4u, 0xa773d2cfu, 0x4ccef014u, 0x00000002u};
    static constexpr uint32_t A1[5] = {0x94d213e3u, 0x89d32568u, 0x00000000u, 0x00000000u, 0x00000000u}, A2[5] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u, 0x00000000u};
    static constexpr uint32_t NB1[5] = {0x7d4f1128u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u, 0x00000000u}, B2[5] = {0x94d213e3u, 0x89d32568u, 0x00000000u, 0x00000000u, 0x00000000u};   // -b1, b2
    static constexpr uint64_t BETA[4] = {0x5763473177fffffeull, 0xd4f263f1acdb5c4full, 0x59e26bcea0d48bacull, 0x0000000000000000ull};                // beta, a plain integer of the base field
};
template <> struct Glv<FqP> {                                  // curve 1: scalars in bn256::Fq, lambda = 0x59e26bcea0d48bacd4f263f1acdb5c4f5763473177fffffe
    static constexpr uint32_t G1[3] = {0xc7e0b3d2u, 0xd91d232eu, 0x00000002u}, G2[5] = {0x391eb18eu, 0x7a7bd9d4u, 0xa773d2cfu, 0x4ccef014u, 0x00000002u};
    static constexpr uint32_t A1[5] = {0x94d213e2u, 0x89d32568u, 0x00000000u, 0x00000000u, 0x00000000u}, A2[5] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u, 0x00000000u};
    static constexpr uint32_t NB1[5] = {0x7d4f1129u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u, 0x00000000u}, B2[5] = {0x94d213e2u, 0x89d32568u, 0x00000000u, 0x00000000u, 0x00000000u};   // -b1, b2
    static constexpr uint64_t BETA[4] = {0x8b17ea66b99c90ddull, 0x5bfc41088d8daaa7ull, 0xb3c4d79d41a91758ull, 0x0000000000000000ull};                // beta, a plain integer of the base field
};
static constexpr uint32_t GLV_BITS = 128;                       // |k1|, |k2| < 2^127, and the carry of the signed digits

