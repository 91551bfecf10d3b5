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
    LAUNCH(k_fold_error<F>, fold_grid(n), 256, 0, g.stream, reinterpret_cast<unsigned char *>(d_e), (const unsigned char *)d_e, terms, (uint32_t)K,
           (const unsigned char *)(reinterpret_cast<unsigned char *>(g.fold_consts.p) + 48), (uint64_t)n);
    tm_mark("fold_error");
    RT_CHECK(rt_last());
    RT_CHECK(rt_sync(g.stream));
    tm_end();
    return MIRA_OK;
}
// RelaxedPlonkWitness::fold in one submission (src/plonk/mod.rs:1097-1134): W' = W1 + r W2 over n_w elements and
// E' = E + sum_k r^(k+1) T_k over n, constants in one copy, one synchronisation.  d_e_out may be d_e.
template <class F, class FP>
static int fold_relaxed_t(void *d_w_out, const void *d_w1, const void *d_w2, size_t n_w, void *d_e_out, const void *d_e, const void *const *d_terms, size_t K,
                          const uint64_t r[4], size_t n) {
    int rc;
    if ((rc = g.fold_consts.ensure(48 * (FOLD_MAX_TERMS + 1)))) return rc;
    uint32_t cm[12 * (FOLD_MAX_TERMS + 1)];
    to_mult48<FP>(r, cm);
    hostf::HFe<FP> rr, p;
    memcpy(rr.l, r, 32);
    p = rr;
    FoldTerms terms;
    memset(&terms, 0, sizeof terms);
    for (size_t k = 0; k < K; k++) {
        to_mult48<FP>(p.l, cm + 12 * (k + 1));
        p = hostf::mul(p, rr);
        terms.t[k] = reinterpret_cast<const unsigned char *>(d_terms[k]);
    }
    RT_CHECK(rt_h2d(g.fold_consts.p, cm, 48 * (K + 1), g.stream));
    tm_begin();
    if (n_w)
        LAUNCH(k_fold_axpy<F>, fold_grid(n_w), 256, 0, g.stream, reinterpret_cast<const unsigned char *>(d_w1), reinterpret_cast<const unsigned char *>(d_w2),
               reinterpret_cast<const unsigned char *>(g.fold_consts.p), (uint64_t)n_w, reinterpret_cast<unsigned char *>(d_w_out));
    tm_mark("fold_witness");
    if (n && K)
        LAUNCH(k_fold_error<F>, fold_grid(n), 256, 0, g.stream, reinterpret_cast<unsigned char *>(d_e_out), reinterpret_cast<const unsigned char *>(d_e), terms, (uint32_t)K,
               (const unsigned char *)(reinterpret_cast<unsigned char *>(g.fold_consts.p) + 48), (uint64_t)n);
    else if (n && d_e_out != d_e)
        RT_CHECK(rt_d2d(d_e_out, d_e, n * 32, g.stream));
    tm_mark("fold_error");
    RT_CHECK(rt_last());
    RT_CHECK(rt_sync(g.stream));
    tm_end();
    return MIRA_OK;
}
template <class F, class FP> static int lincomb_t(void *d_out, const void *const *d_vecs, const uint64_t *coeffs, size_t K, size_t n) {
    int rc;
    if ((rc = g.fold_consts.ensure(48 * (FOLD_MAX_TERMS + 1)))) return rc;
    uint32_t cm[12 * FOLD_MAX_TERMS];
    FoldTerms vecs;
    memset(&vecs, 0, sizeof vecs);
    for (size_t k = 0; k < K; k++) {
        to_mult48<FP>(coeffs + 4 * k, cm + 12 * k);
        vecs.t[k] = reinterpret_cast<const unsigned char *>(d_vecs[k]);
    }
    RT_CHECK(rt_h2d(reinterpret_cast<unsigned char *>(g.fold_consts.p) + 48, cm, 48 * K, g.stream));
    tm_begin();
    LAUNCH(k_lincomb<F>, fold_grid(n), 256, 0, g.stream, reinterpret_cast<unsigned char *>(d_out), vecs, (uint32_t)K,
           (const unsigned char *)(reinterpret_cast<unsigned char *>(g.fold_consts.p) + 48), (uint64_t)n);
    tm_mark("lincomb");
    RT_CHECK(rt_last());
    RT_CHECK(rt_sync(g.stream));
    tm_end();
    return MIRA_OK;
}
template <class F, class FP> static int lincomb_multi_t(void *const *d_outs, size_t M, const void *const *d_vecs, size_t J, const uint64_t *coeffs, size_t n) {
    int rc;
    if ((rc = g.fold_consts.ensure(48 * (FOLD_MAX_TERMS + 1 + LINCOMB_MAX_OUTS * FOLD_MAX_TERMS)))) return rc;
    std::vector<uint32_t> cm(12 * M * J);
    FoldTerms vecs;
    FoldOuts outs;
    memset(&vecs, 0, sizeof vecs); memset(&outs, 0, sizeof outs);
    for (size_t j = 0; j < J; j++) vecs.t[j] = reinterpret_cast<const unsigned char *>(d_vecs[j]);
    for (size_t m = 0; m < M; m++) {
        outs.t[m] = reinterpret_cast<unsigned char *>(d_outs[m]);
        for (size_t j = 0; j < J; j++) to_mult48<FP>(coeffs + 4 * (m * J + j), cm.data() + 12 * (m * J + j));
    }
    unsigned char *d_c = reinterpret_cast<unsigned char *>(g.fold_consts.p) + 48 * (FOLD_MAX_TERMS + 1);
    RT_CHECK(rt_h2d(d_c, cm.data(), 48 * M * J, g.stream));
    tm_begin();
    LAUNCH(k_lincomb_multi<F>, fold_grid(n), 256, 0, g.stream, outs, (uint32_t)M, vecs, (uint32_t)J, (const unsigned char *)d_c, (uint64_t)n);
    tm_mark("lincomb_multi");
    RT_CHECK(rt_last());
    RT_CHECK(rt_sync(g.stream));                                 // (the staged coefficients live on this function's stack until here)
    tm_end();
    return MIRA_OK;
}
int lincomb_multi_device(int field, void *const *d_outs, size_t M, const void *const *d_vecs, size_t J, const uint64_t *coeffs, size_t n) {
    return field == 1 ? lincomb_multi_t<Fr29, FrP>(d_outs, M, d_vecs, J, coeffs, n) : lincomb_multi_t<Fq29, FqP>(d_outs, M, d_vecs, J, coeffs, n);
}
int lincomb_device(int field, void *d_out, const void *const *d_vecs, const uint64_t *coeffs, size_t K, size_t n) {
    return field == 1 ? lincomb_t<Fr29, FrP>(d_out, d_vecs, coeffs, K, n) : lincomb_t<Fq29, FqP>(d_out, d_vecs, coeffs, K, n);
}

// Rounds of k_pow_tree until one value per point is left: each round folds up to 11 levels
// (8 leaves per lane, 256 lanes per workgroup).
template <class FP> static int pow_tree_t(const void *d_leaves, uint32_t levels_total, size_t leaf_point_stride, const uint64_t *weights, uint32_t P, uint64_t *out) {
    int rc;
    const size_t n = (size_t)1 << levels_total;
    if (levels_total == 0) {                                  // a single leaf per point is its own root
        for (uint32_t p = 0; p < P; p++) RT_CHECK(rt_d2h(out + 4 * p, reinterpret_cast<const unsigned char *>(d_leaves) + (size_t)p * leaf_point_stride * 32, 32, g.stream));
        RT_CHECK(rt_sync(g.stream));
        return MIRA_OK;
    }
    const size_t wbytes = (size_t)P * levels_total * 32;
    const size_t first_out = n >> std::min<uint32_t>(levels_total, 11);
    if ((rc = g.tree_w.ensure(wbytes))) return rc;
    if ((rc = g.tree_a.ensure(std::max<size_t>(1, first_out) * P * 32))) return rc;
    if ((rc = g.tree_b.ensure(std::max<size_t>(1, first_out >> std::min<size_t>(11, levels_total > 11 ? levels_total - 11 : 0)) * P * 32 + 32))) return rc;
    RT_CHECK(rt_h2d(g.tree_w.p, weights, wbytes, g.stream));
    tm_begin();
    const unsigned char *in = reinterpret_cast<const unsigned char *>(d_leaves);
    uint64_t in_stride = leaf_point_stride;
    uint32_t level0 = 0;
    size_t n_cur = n;
    unsigned char *bufs[2] = {reinterpret_cast<unsigned char *>(g.tree_a.p), reinterpret_cast<unsigned char *>(g.tree_b.p)};
    int which = 0;
    while (level0 < levels_total) {
        const uint32_t levels = std::min<uint32_t>(11, levels_total - level0);
        const uint32_t serial = levels > 8 ? levels - 8 : 0;
        const uint32_t block = 1u << (levels - serial);
        const size_t n_out = n_cur >> levels;
        LAUNCH_BARRIER(k_pow_tree<FP>, dim3((uint32_t)n_out, P), block, (size_t)block * 32, g.stream, in, in_stride, serial, levels, level0,
                       (const unsigned char *)g.tree_w.p, levels_total, bufs[which], (uint64_t)n_out);
        in = bufs[which]; in_stride = n_out; n_cur = n_out; level0 += levels; which ^= 1;
    }
    tm_mark("pow_tree");
    RT_CHECK(rt_last());
    RT_CHECK(rt_d2h(out, in, (size_t)P * 32, g.stream));     // n_cur == 1: point p's root at element p
    RT_CHECK(rt_sync(g.stream));
    tm_end();
    return MIRA_OK;
}
int pow_tree_reduce_device(int field, const void *d_leaves, uint32_t levels, size_t leaf_point_stride, const uint64_t *weights, uint32_t P, uint64_t *out) {
    return field == 1 ? pow_tree_t<FrP>(d_leaves, levels, leaf_point_stride, weights, P, out) : pow_tree_t<FqP>(d_leaves, levels, leaf_point_stride, weights, P, out);
}

int fold_witness_device(int field, void *d_out, const void *d_w1, const void *d_w2, const uint64_t r[4], size_t n) {
    return field == 1 ? fold_witness_t<Fr29, FrP>(d_out, d_w1, d_w2, r, n) : fold_witness_t<Fq29, FqP>(d_out, d_w1, d_w2, r, n);
}
int fold_relaxed_device(int field, void *d_w_out, const void *d_w1, const void *d_w2, size_t n_w, void *d_e_out, const void *d_e, const void *const *d_terms, size_t K,
                        const uint64_t r[4], size_t n) {
    return field == 1 ? fold_relaxed_t<Fr29, FrP>(d_w_out, d_w1, d_w2, n_w, d_e_out, d_e, d_terms, K, r, n)
                      : fold_relaxed_t<Fq29, FqP>(d_w_out, d_w1, d_w2, n_w, d_e_out, d_e, d_terms, K, r, n);
}
int fold_error_device(int field, void *d_e, const void *const *d_terms, size_t K, const uint64_t r[4], size_t n) {
    return field == 1 ? fold_error_t<Fr29, FrP>(d_e, d_terms, K, r, n) : fold_error_t<Fq29, FqP>(d_e, d_terms, K, r, n);
}
