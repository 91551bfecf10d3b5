// Host side of the witness-folding kernels (fold_kernels.cuh) and of key export.
#include "ctx.h"
#include "fold_kernels.cuh"
#include "host_field.hpp"

// reference form (x * 2^256) -> multiplier form (x * 2^261 = (32 x) * 2^256) as 9 x 29-bit limbs in 48 bytes
template <class FP> static void to_mult48(const uint64_t v_r256[4], uint32_t out[12]) {
    hostf::HFe<FP> s;
    memcpy(s.l, v_r256, 32);
    s = hostf::mul(s, hostf::from_u64<FP>(32));
    memset(out, 0, 48);
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit / 64, sh = bit % 64;
        uint64_t v = s.l[w] >> sh;
        if (sh > 35 && w + 1 < 4) v |= s.l[w + 1] << (64 - sh);
        out[i] = (uint32_t)(v & 0x1FFFFFFFu);
    }
}
static uint32_t fold_grid(uint64_t n) { return (uint32_t)std::min<uint64_t>(256 * 8, (n + 255) / 256); }

template <class F, class FP> static int fold_witness_t(void *d_out, const void *d_w1, const void *d_w2, const uint64_t r[4], size_t n) {
    int rc;
    if ((rc = g.fold_consts.ensure(48 * (FOLD_MAX_TERMS + 1)))) return rc;
    uint32_t rm[12];
    to_mult48<FP>(r, rm);
    RT_CHECK(rt_h2d(g.fold_consts.p, rm, 48, g.stream));
    tm_begin();
    LAUNCH(k_fold_axpy<F>, fold_grid(n), 256, 0, g.stream, reinterpret_cast<const unsigned char *>(d_w1), reinterpret_cast<const unsigned char *>(d_w2),
           reinterpret_cast<const unsigned char *>(g.fold_consts.p), (uint64_t)n, reinterpret_cast<unsigned char *>(d_out));
    tm_mark("fold_witness");
    RT_CHECK(rt_last());
    RT_CHECK(rt_sync(g.stream));
    tm_end();
    return MIRA_OK;
}
template <class F, class FP> static int fold_error_t(void *d_e, const void *const *d_terms, size_t K, const uint64_t r[4], size_t n) {
    int rc;
    if ((rc = g.fold_consts.ensure(48 * (FOLD_MAX_TERMS + 1)))) return rc;
    uint32_t pw[12 * FOLD_MAX_TERMS];
    hostf::HFe<FP> rr, p;
    memcpy(rr.l, r, 32);
    p = rr;                                                // r^1, r^2, ... (src/plonk/mod.rs:1119-1121)
    FoldTerms terms;
    memset(&terms, 0, sizeof terms);
    for (size_t k = 0; k < K; k++) {
        to_mult48<FP>(p.l, pw + 12 * k);
        p = hostf::mul(p, rr);
        terms.t[k] = reinterpret_cast<const unsigned char *>(d_terms[k]);
    }
    RT_CHECK(rt_h2d(reinterpret_cast<unsigned char *>(g.fold_consts.p) + 48, pw, 48 * K, g.stream));
    tm_begin();
    LAUNCH(k_fold_error<F>, fold_grid(n), 256, 0, g.stream, reinterpret_cast<unsigned char *>(d_e), terms, (uint32_t)K,
           (const unsigned char *)(reinterpret_cast<unsigned char *>(g.fold_consts.p) + 48), (uint64_t)n);
    tm_mark("fold_error");
    RT_CHECK(rt_last());
    RT_CHECK(rt_sync(g.stream));
    tm_end();
    return MIRA_OK;
}
int fold_witness_device(int field, void *d_out, const void *d_w1, const void *d_w2, const uint64_t r[4], size_t n) {
    return field == 1 ? fold_witness_t<Fr29, FrP>(d_out, d_w1, d_w2, r, n) : fold_witness_t<Fq29, FqP>(d_out, d_w1, d_w2, r, n);
}
int fold_error_device(int field, void *d_e, const void *const *d_terms, size_t K, const uint64_t r[4], size_t n) {
    return field == 1 ? fold_error_t<Fr29, FrP>(d_e, d_terms, K, r, n) : fold_error_t<Fq29, FqP>(d_e, d_terms, K, r, n);
}
