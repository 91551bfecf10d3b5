// NTT host side: root derivation (reference src/fft.rs:12-27), twiddle tables, pass schedule.
#include "ctx.h"
#include "host_field.hpp"
#include "ntt_kernels.cuh"

// ------------------------------------------------------------------------------------------
// NTT host side
using HFr = hostf::HFe<FrP>;
static constexpr uint32_t NTT_PERSISTENT_GRID = 256;   // one 1024-lane workgroup per CU (144 KiB of LDS each)
static HFr fr_root_of_unity(bool inverse) {
    // ROOT_OF_UNITY = 7^((r-1) >> 28); multiplicative generator 7, S = 28 (halo2curves bn256::Fr)
    uint64_t e[4];
    for (int i = 0; i < 4; i++) e[i] = hostf::P64<FrP>(i);
    e[0] -= 1;
    for (int s = 0; s < 28; s++) {
        for (int j = 0; j < 3; j++) e[j] = (e[j] >> 1) | (e[j + 1] << 63);
        e[3] >>= 1;
    }
    HFr w = hostf::pow(hostf::from_u64<FrP>(7), e);
    return inverse ? hostf::inv(w) : w;
}
static HFr get_omega_or_inv_h(uint32_t k, bool inverse) {   // src/fft.rs:12-23
    HFr w = fr_root_of_unity(inverse);
    for (uint32_t i = k; i < 28; i++) w = hostf::sqr(w);
    return w;
}

// device tables for (log_n, omega): [line_tw(m1) | line_tw(m2) | t_lo | t_hi]
struct NttTables {
    uint32_t m1, m2, h;
    size_t off_tw1, off_tw2, off_lo, off_hi;
};
static int ntt_prepare_tables(uint32_t log_n, const uint64_t omega[4], NttTables &t) {
    t.m1 = log_n <= NTT_MAX_LOG_LINE ? log_n : (log_n + 1) / 2;
    t.m2 = log_n - t.m1;
    t.h = t.m1;   // exponent split for omega^(i2 * k1): low h bits / rest
    const size_t n_tw1 = t.m1 ? (size_t)1 << (t.m1 - 1) : 1, n_tw2 = t.m2 ? (size_t)1 << (t.m2 - 1) : 1;
    const size_t n_lo = (size_t)1 << t.h, n_hi = (size_t)1 << (log_n - t.h);
    t.off_tw1 = 0; t.off_tw2 = t.off_tw1 + n_tw1 * TW_BYTES; t.off_lo = t.off_tw2 + n_tw2 * TW_BYTES; t.off_hi = t.off_lo + n_lo * TW_BYTES;
    std::string key((const char *)omega, 32);
    key += std::to_string(log_n);
    if (key == g.ntt_tables_key) return MIRA_OK;
    int rc;
    if ((rc = g.ntt_tables.ensure(t.off_hi + n_hi * TW_BYTES))) return rc;
    if ((rc = g.ntt_consts.ensure(256))) return rc;
    RT_CHECK(rt_h2d(g.ntt_consts.p, omega, 32, g.stream));
    unsigned char *tab = reinterpret_cast<unsigned char *>(g.ntt_tables.p);
    const unsigned char *w = reinterpret_cast<const unsigned char *>(g.ntt_consts.p);
    const unsigned char *none = nullptr;
    const uint64_t n = (uint64_t)1 << log_n;
    LAUNCH(k_pow_table<Fr29>, ceil_div(n_tw1, 256), 256, 0, g.stream, w, n >> t.m1, (uint32_t)n_tw1, none, tab + t.off_tw1);
    if (t.m2) {
        LAUNCH(k_pow_table<Fr29>, ceil_div(n_tw2, 256), 256, 0, g.stream, w, n >> t.m2, (uint32_t)n_tw2, none, tab + t.off_tw2);
        LAUNCH(k_pow_table<Fr29>, ceil_div(n_lo, 256), 256, 0, g.stream, w, (uint64_t)1, (uint32_t)n_lo, none, tab + t.off_lo);
        LAUNCH(k_pow_table<Fr29>, ceil_div(n_hi, 256), 256, 0, g.stream, w, (uint64_t)1 << t.h, (uint32_t)n_hi, none, tab + t.off_hi);
    }
    RT_CHECK(rt_last());
    g.ntt_tables_key = key;
    return MIRA_OK;
}

// best_fft on device memory, optional final scale (Montgomery, host limbs) for ifft
static int ntt_device_locked(void *d_a, uint32_t log_n, const uint64_t omega[4], const uint64_t *scale) {
    int rc;
    if (!d_a || !omega) { set_error("null argument"); return MIRA_E_BAD_ARG; }
    if (log_n > 28) { set_error("k=" + std::to_string(log_n) + " should no larger than F::S=28"); return MIRA_E_BAD_ARG; }
    if (log_n > 2 * NTT_MAX_LOG_LINE) { set_error("log_n > 24 is not supported by this build"); return MIRA_E_UNSUPPORTED; }
    NttTables t;
    tm_begin();
    if ((rc = ntt_prepare_tables(log_n, omega, t))) return rc;
    tm_mark("twiddle_tables");
    const unsigned char *tab = reinterpret_cast<const unsigned char *>(g.ntt_tables.p);
    unsigned char *scale_d = nullptr;
    uint32_t scale36[12] = {0};
    {
        // the last pass multiplies by the ifft scale, or by one: reference form (x * 2^256) ->
        // multiplier-operand form (x * 2^261 = (32 x) * 2^256), 9 x 29-bit limbs
        HFr s = hostf::one<FrP>();
        if (scale) memcpy(s.l, scale, 32);
        s = hostf::mul(s, hostf::from_u64<FrP>(32));
        for (int i = 0; i < 9; i++) {
            const int bit = 29 * i, w = bit / 64, sh = bit % 64;
            uint64_t v = s.l[w] >> sh;
            if (sh > 35 && w + 1 < 4) v |= s.l[w + 1] << (64 - sh);
            scale36[i] = (uint32_t)(v & 0x1FFFFFFFu);
        }
        RT_CHECK(rt_h2d(reinterpret_cast<unsigned char *>(g.ntt_consts.p) + 64, scale36, 48, g.stream));
        scale_d = reinterpret_cast<unsigned char *>(g.ntt_consts.p) + 64;
    }
    unsigned char *a = reinterpret_cast<unsigned char *>(d_a);
    const unsigned char *cnull = nullptr;
    auto threads_for = [](uint32_t m) { return std::min<uint32_t>(1024, std::max<uint32_t>(64, (1u << m) / 2)); };
    auto lds_for = [](uint32_t m) { return ((size_t)NTT_LDS_BYTES_PER_ELEM << m) + 16 + ((size_t)NTT_LDS_BYTES_PER_ELEM << NTT_LDS_TW_LOG); };
    if (t.m2 == 0) {
        NttPass ps{t.m1, 1, 0, 1, 0, 1, 0xFFFFFFFFu, 0u};
        LAUNCH_BARRIER(k_ntt_lines<Fr29>, 1, threads_for(t.m1), lds_for(t.m1), g.stream, (const unsigned char *)a, a, ps,
                       tab + t.off_tw1, cnull, cnull, (const unsigned char *)scale_d);
        tm_mark("ntt_single");
    } else {
        const uint64_t n1 = (uint64_t)1 << t.m1, n2 = (uint64_t)1 << t.m2;
        if ((rc = g.ntt_tmp.ensure(((size_t)32) << log_n))) return rc;
        unsigned char *tmp = reinterpret_cast<unsigned char *>(g.ntt_tmp.p);
        // pass 1: columns i2 of the n1 x n2 view; B[k1][i2] * omega^(i2 k1) -> tmp[i2 * n1 + k1]
        NttPass p1{t.m1, (uint32_t)n2, 1, n2, n1, 1, t.h, 0u};
        LAUNCH_BARRIER(k_ntt_lines<Fr29>, std::min<uint32_t>((uint32_t)n2, NTT_PERSISTENT_GRID), threads_for(t.m1), lds_for(t.m1), g.stream, (const unsigned char *)a, tmp, p1,
                       tab + t.off_tw1, tab + t.off_lo, tab + t.off_hi, cnull);
        tm_mark("ntt_pass1");
        // pass 2: for each k1 the length-n2 transform over i2; X[k1 + n1 k2] -> a
        NttPass p2{t.m2, (uint32_t)n1, 1, n1, 1, n1, 0xFFFFFFFFu, 0u};
        LAUNCH_BARRIER(k_ntt_lines<Fr29>, std::min<uint32_t>((uint32_t)n1, NTT_PERSISTENT_GRID), threads_for(t.m2), lds_for(t.m2), g.stream, (const unsigned char *)tmp, a, p2,
                       tab + t.off_tw2, cnull, cnull, (const unsigned char *)scale_d);
        tm_mark("ntt_pass2");
    }
    RT_CHECK(rt_last());
    RT_CHECK(rt_sync(g.stream));
    tm_end();
    return MIRA_OK;
}

static int distribute_powers_locked(void *d_a, uint32_t log_n, bool into_coset) {   // src/fft.rs:205-226
    // Fr::ZETA (halo2curves bn256::Fr, WithSmallOrderMulGroup<3>)
    HFr z_plain = {{0xb8ca0b2d36636f23ULL, 0xcc37a73fec2bc5e9ULL, 0x048b6e193fd84104ULL, 0x30644e72e131a029ULL}};
    HFr z = hostf::to_mont(z_plain), zi = hostf::sqr(z);
    uint64_t pw[8];
    memcpy(pw, into_coset ? z.l : zi.l, 32);
    memcpy(pw + 4, into_coset ? zi.l : z.l, 32);
    int rc;
    if ((rc = g.ntt_consts.ensure(256))) return rc;
    unsigned char *d_pw = reinterpret_cast<unsigned char *>(g.ntt_consts.p) + 128;
    RT_CHECK(rt_h2d(d_pw, pw, 64, g.stream));
    const uint64_t n = (uint64_t)1 << log_n;
    LAUNCH(k_distribute_powers<Fr29>, ceil_div(n, 256), 256, 0, g.stream, reinterpret_cast<unsigned char *>(d_a), n, (const unsigned char *)d_pw);
    RT_CHECK(rt_last());
    RT_CHECK(rt_sync(g.stream));
    return MIRA_OK;
}

int ntt_kind_device(void *d_a, uint32_t log_n, NttKind kind, const uint64_t *omega_in) {
    int rc;
    if (log_n > 28) { set_error("k=" + std::to_string(log_n) + " should no larger than F::S=28"); return MIRA_E_BAD_ARG; }
    if (kind == NTT_BEST) return ntt_device_locked(d_a, log_n, omega_in, nullptr);
    const bool inverse = (kind == NTT_IFFT || kind == NTT_COSET_IFFT);
    HFr w = get_omega_or_inv_h(log_n, inverse);
    if (kind == NTT_COSET_FFT && (rc = distribute_powers_locked(d_a, log_n, true))) return rc;
    if (inverse) {
        uint64_t e[4] = {log_n, 0, 0, 0};
        HFr divisor = hostf::pow(hostf::inv(hostf::from_u64<FrP>(2)), e);   // TWO_INV^log_n, src/fft.rs:25-27
        rc = ntt_device_locked(d_a, log_n, w.l, divisor.l);
    } else {
        rc = ntt_device_locked(d_a, log_n, w.l, nullptr);
    }
    if (rc) return rc;
    if (kind == NTT_COSET_IFFT) return distribute_powers_locked(d_a, log_n, false);
    return MIRA_OK;
}

int ntt_init() {
#ifndef MIRA_CPU_EMU
    // a 4096-point line is 128 KiB of LDS, above the 64 KiB default
    RT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ntt_lines<Fr29>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#endif
    return MIRA_OK;
}
int ntt_get_omega_or_inv(uint32_t k, bool inverse, uint64_t out[4]) {
    HFr w = get_omega_or_inv_h(k, inverse);
    memcpy(out, w.l, 32);
    return MIRA_OK;
}
