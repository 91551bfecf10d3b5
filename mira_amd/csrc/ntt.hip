// NTT host side: root derivation (reference src/fft.rs:12-27), twiddle tables, pass schedule.
#include "ctx.h"
#include "host_field.hpp"
#include "ntt_kernels.cuh"

// ------------------------------------------------------------------------------------------
// NTT host side
using HFr = hostf::HFe<FrP>;
static constexpr uint32_t NTT_PERSISTENT_GRID = 256;
// g.ntt_consts: [0, 256) uploaded constants (omega, scales, zeta powers); [1024, 4096) the work counters of k_ntt_wave, one
// 128-byte line per (pass, range): zeroed at the head of every transform
static constexpr size_t NTT_CONSTS_BYTES = 4096, NTT_CTR_OFFSET = 1024, NTT_CTR_PASS_BYTES = NTTW_RANGES * NTTW_CTR_STRIDE * 4;
static constexpr uint32_t NTT_SINGLE_TW_LOG = 16;      // post-twiddle exponents below 2^16: one 3 MiB table instead of a two-table product   // workgroups per resident round: one per CU times what fits a CU
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

// Schedule of one transform: up to three passes of lines of at most 2^max_line points
// (n = n1 n2 n3; index algebra in ntt_device_locked).  Device tables for (log_n, omega):
//   line_tw[p]   omega_N^j, j < N/2, for the line length N of pass p (shared when lengths repeat)
//   lo[s], hi[s] the two halves of the post-twiddle exponent of pass s < passes - 1
struct NttTables {
    uint32_t passes, m[3], h[2], single[2];
    size_t off_tw[3], off_lo[2], off_hi[2], off_full;
};
// A table of n entries for the first post-twiddle (single[0] = 2) instead of the two-table product:
// one multiplication per element less in pass 1 (2^24: 0.96 -> 0.85 ms), for 48 n bytes per cached
// (omega, log_n) -- 805 MB at 2^24, streamed once per transform beside an ALU-bound kernel.
static constexpr uint32_t NTT_FULL_TW_MAX_LOG = 24;
// Which kernel, and the longest line of a transform of 2^log_n points.  Lines of up to 256 points
// can run on the wave-level kernel (k_ntt_wave: no barriers, two layers per LDS trip); it wins from
// 2^20 up (three passes of 2^7 .. 2^8-point lines: 2^22 0.65 vs 0.74 ms, 2^24 2.55 vs 2.82) and at
// <= 2^8 it is the only sensible shape.  2^9 .. 2^19 stay on the workgroup-level kernel: one launch up to
// 2^12, two passes of 2^7 .. 2^10-point lines up to 2^19, where a third launch would cost more than the
// wave kernel saves (2^17: 0.046 vs 0.063 ms); 2^25 .. 2^28 need its longer lines to fit three passes.
// MIRA_TUNE_NTT_WAVE: 0 = never, 1 = wherever three passes of 256-point lines reach (tests);
// MIRA_TUNE_NTT_MAX_LOG_LINE: a smaller maximum line makes the two- and three-pass schedules
// reachable at sizes the CPU emulation can run.
static bool ntt_use_wave(uint32_t log_n) {
    const int64_t mode = g.tune[MIRA_TUNE_NTT_WAVE];
    if (mode == 0) return false;
    if (mode > 0) return log_n <= 3 * NTTW_LOG;
    return log_n <= NTTW_LOG || (log_n >= 20 && log_n <= 3 * NTTW_LOG);
}
// One workgroup can hold a line of 2^12 points, but it is ONE workgroup on one CU walking twelve layers behind barriers: 2^12 points
// took 92 us as a single line and take 40 us as two passes of 64-point lines on 64 CUs (2^11: 57 -> 39, 2^10: 42 -> 38; 2^9 stays a
// single line, 35 against 37: tools/ntt_small_probe.py).  The single line is for 2^9 points and fewer.
static constexpr uint32_t NTT_SINGLE_LINE_MAX_LOG = 9;
static uint32_t ntt_max_log_line(uint32_t log_n) {
    if (g.tune[MIRA_TUNE_NTT_MAX_LOG_LINE] < 0)
        return ntt_use_wave(log_n) ? NTTW_LOG : (log_n > NTT_SINGLE_LINE_MAX_LOG && log_n <= NTT_MAX_LOG_LINE) ? NTT_SINGLE_LINE_MAX_LOG : NTT_MAX_LOG_LINE;
    const int v = (int)g.tune[MIRA_TUNE_NTT_MAX_LOG_LINE];
    return (uint32_t)std::min(std::max(v, 1), NTT_MAX_LOG_LINE);
}
// scale261 (reference Montgomery form): what the post-twiddle feeding the LAST pass is multiplied by -- 2^261 times the
// transform's final scale -- so that the last pass ends with a Montgomery reduction instead of a multiplication.
static int ntt_prepare_tables(uint32_t log_n, const uint64_t omega[4], const HFr &scale261, NttTables &t, bool want_full = true) {
    const uint32_t max_line = ntt_max_log_line(log_n);
    t.passes = log_n <= max_line ? 1 : log_n <= 2 * max_line ? 2 : 3;
    uint32_t rest = log_n;
    for (uint32_t p = 0; p < 3; p++) {                        // balanced split, larger factors first
        t.m[p] = p < t.passes ? (rest + (t.passes - p) - 1) / (t.passes - p) : 0;
        rest -= t.m[p];
    }
    // exponent ranges of the post-twiddles: pass 0 uses powers of omega_n below n, pass 1 (of three)
    // powers of omega_n^(n1) below n / n1
    const uint32_t range[2] = {log_n, log_n - t.m[0]};
    size_t off = 0;
    size_t n_tw[3], n_lo[2] = {0, 0}, n_hi[2] = {0, 0};
    for (uint32_t p = 0; p < 3; p++) {
        n_tw[p] = (p < t.passes) ? (t.m[p] ? (size_t)1 << (t.m[p] - 1) : 1) : 0;
        t.off_tw[p] = off; off += n_tw[p] * TW_BYTES;
    }
    const uint32_t single_log = (uint32_t)tuned(MIRA_TUNE_NTT_SINGLE_TW_LOG, NTT_SINGLE_TW_LOG);
    const bool full0 = want_full && t.passes > 1 && range[0] > single_log && log_n <= (uint32_t)tuned(MIRA_TUNE_NTT_FULL_TW_MAX_LOG, NTT_FULL_TW_MAX_LOG);
    for (uint32_t q = 0; q + 1 < t.passes; q++) {
        t.single[q] = range[q] <= single_log;                                      // small ranges: one table, no product per element
        t.h[q] = t.single[q] ? range[q] : (range[q] + 1) / 2;
        n_lo[q] = (size_t)1 << t.h[q]; n_hi[q] = (size_t)1 << (range[q] - t.h[q]);
        t.off_lo[q] = off; off += n_lo[q] * TW_BYTES;
        t.off_hi[q] = off; off += n_hi[q] * TW_BYTES;
    }
    t.off_full = off;
    if (full0) off += ((size_t)TW_BYTES) << log_n;
    std::string key((const char *)omega, 32);
    key.append((const char *)scale261.l, 32);
    key += std::to_string(log_n) + "/" + std::to_string(max_line) + "/" + std::to_string(single_log) + (full0 ? "/full" : "");
    // four cached sets, least recently used replaced: fft and ifft of two sizes alternate without rebuilding
    int slot = -1, lru = 0;
    for (int i = 0; i < Ctx::NTT_SETS; i++) {
        if (g.ntt_set_key[i] == key) slot = i;
        if (g.ntt_set_stamp[i] < g.ntt_set_stamp[lru]) lru = i;
    }
    const bool hit = slot >= 0;
    if (!hit) slot = lru;
    g.ntt_set_stamp[slot] = ++g.ntt_stamp;
    g.ntt_set_cur = slot;
    if (full0) t.single[0] = 2;
    if (hit) return MIRA_OK;
    int rc;
    g.ntt_set_key[slot].clear();
    // an evicted set that is far larger than the new one goes back to the allocator (a 2^24 set is 0.8 GB)
    if (g.ntt_set[slot].cap > 4 * off + ((size_t)1 << 20)) g.ntt_set[slot].release();
    if ((rc = g.ntt_set[slot].ensure(off))) {
        // the n-entry table is an optimisation: without the memory for it the two-table product still works.  The slot that
        // could not be grown is empty now (DevBuf::ensure released it): it goes back to the FRONT of the replacement order so that
        // the retry takes it again instead of evicting a second cached set, and the failed attempt's message does not outlive it.
        g.ntt_set_stamp[slot] = 0;
        if (full0) { set_error(""); return ntt_prepare_tables(log_n, omega, scale261, t, false); }
        return rc;
    }
    if ((rc = g.ntt_consts.ensure(NTT_CONSTS_BYTES))) return rc;
    RT_CHECK(rt_h2d(g.ntt_consts.p, omega, 32, g.stream));
    RT_CHECK(rt_h2d(reinterpret_cast<unsigned char *>(g.ntt_consts.p) + 32, scale261.l, 32, g.stream));
    unsigned char *tab = reinterpret_cast<unsigned char *>(g.ntt_set[slot].p);
    const unsigned char *w = reinterpret_cast<const unsigned char *>(g.ntt_consts.p), *w_scale = w + 32;
    const unsigned char *none = nullptr;
    const uint64_t n = (uint64_t)1 << log_n;
    for (uint32_t p = 0; p < t.passes; p++)
        LAUNCH(k_pow_table<Fr29>, ceil_div(n_tw[p], 256), 256, 0, g.stream, w, n >> t.m[p], (uint32_t)n_tw[p], none, tab + t.off_tw[p]);
    for (uint32_t q = 0; q + 1 < t.passes; q++) {
        const uint64_t base_stride = q == 0 ? 1 : (uint64_t)1 << t.m[0];   // omega_n, then omega_n^(n1)
        // the boundary in front of the last pass carries the final scale: on its only table, or on the high one of two
        const bool last = q + 2 == t.passes, lo_only = t.single[q] == 1;
        LAUNCH(k_pow_table<Fr29>, ceil_div(n_lo[q], 256), 256, 0, g.stream, w, base_stride, (uint32_t)n_lo[q], last && lo_only ? w_scale : none, tab + t.off_lo[q]);
        LAUNCH(k_pow_table<Fr29>, ceil_div(n_hi[q], 256), 256, 0, g.stream, w, base_stride << t.h[q], (uint32_t)n_hi[q], last && !lo_only ? w_scale : none, tab + t.off_hi[q]);
    }
    if (full0)
        LAUNCH(k_tw_full<Fr29>, ceil_div(n, 256), 256, 0, g.stream, (const unsigned char *)(tab + t.off_lo[0]), (const unsigned char *)(tab + t.off_hi[0]), t.h[0], t.m[0], n,
               tab + t.off_full);
    RT_CHECK(rt_last());
    g.ntt_set_key[slot] = key;
    return MIRA_OK;
}

// best_fft on device memory, optional final scale (Montgomery, host limbs) for ifft.
//
// One pass (n <= 4096): the line is the transform.  Two passes, n = n1 n2 (four-step), input index
// i = i1 n2 + i2, output k = k1 + n1 k2:
//   pass 0  for every i2: length-n1 transform over i1 (stride n2), times w^(i2 k1)  -> tmp[i2 n1 + k1]
//   pass 1  for every k1: length-n2 transform over i2 (stride n1)                    -> a[k1 + n1 k2]
// Three passes, n = n1 n2 n3 (log_n 25..28), m = n2 n3, i = i1 m + j, j = j2 n3 + j3,
// k = k1 + n1 (k2 + n2 k3) -- the four-step split applied to n = n1 m and again to m = n2 n3:
//   pass 0  for every j: length-n1 over i1 (stride m), times w^(j k1)               -> tmp[j n1 + k1]
//   pass 1  for every (j3, k1): length-n2 over j2 (stride n3 n1), times w_m^(j3 k2)  -> a[j3 n1 n2 + k2 n1 + k1]
//   pass 2  for every (k2, k1): length-n3 over j3 (stride n1 n2), in place           -> a[k1 + n1 k2 + n1 n2 k3]
// (pass 2 reads and writes one and the same set of addresses per line).
static int ntt_device_locked(void *d_a, uint32_t log_n, const uint64_t omega[4], const uint64_t *scale) {
    int rc;
    if (!d_a || !omega) { set_error("null argument"); return MIRA_E_BAD_ARG; }
    if (log_n > 28) { set_error("k=" + std::to_string(log_n) + " should no larger than F::S=28"); return MIRA_E_BAD_ARG; }
    if (log_n > 3 * ntt_max_log_line(log_n)) { set_error("log_n exceeds three passes of the configured line length"); return MIRA_E_UNSUPPORTED; }
    NttTables t;
    tm_begin();
    // the final scale (ifft: TWO_INV^log_n, else one) rides on the post-twiddle in front of the last pass, times 2^261:
    // mont(s * 2^261) = s_mont * 32 * R^2 / R^2 ... in this library's terms mul(mul(s, 32), R^2)
    HFr s_final = hostf::one<FrP>();
    if (scale) memcpy(s_final.l, scale, 32);
    const HFr scale261 = hostf::mul(hostf::mul(s_final, hostf::from_u64<FrP>(32)), hostf::r2<FrP>());
    if ((rc = ntt_prepare_tables(log_n, omega, scale261, t))) return rc;
    if ((rc = g.ntt_consts.ensure(NTT_CONSTS_BYTES))) return rc;
    tm_mark("twiddle_tables");
    const unsigned char *tab = reinterpret_cast<const unsigned char *>(g.ntt_set[g.ntt_set_cur].p);
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
        // (uploaded when it changes: a forward transform after a forward transform has the scale of one already)
        static uint32_t uploaded36[12];
        static const void *uploaded_to = nullptr;
        if (uploaded_to != g.ntt_consts.p || memcmp(uploaded36, scale36, 48) != 0) {
            RT_CHECK(rt_h2d(reinterpret_cast<unsigned char *>(g.ntt_consts.p) + 64, scale36, 48, g.stream));
            memcpy(uploaded36, scale36, 48); uploaded_to = g.ntt_consts.p;
        }
        scale_d = reinterpret_cast<unsigned char *>(g.ntt_consts.p) + 64;
    }
    unsigned char *a = reinterpret_cast<unsigned char *>(d_a);
    const unsigned char *cnull = nullptr;
    auto threads_for = [](uint32_t m) { return std::min<uint32_t>(1024, std::max<uint32_t>(64, (1u << m) / 2)); };
    auto lds_for = [](uint32_t m) { return ((size_t)NTT_LDS_BYTES_PER_ELEM << m) + 16 + ((size_t)NTT_LDS_BYTES_PER_ELEM << NTT_LDS_TW_LOG) + 16; };   // line, twiddles, the next lines' numbers
    const bool use_wave = ntt_use_wave(log_n);
    auto lo_ptr = [&](int tw) { return tw < 0 ? (const unsigned char *)nullptr : (tw == 0 && t.single[0] == 2) ? tab + t.off_full : tab + t.off_lo[tw]; };
    auto run = [&](NttPass ps, const unsigned char *src, unsigned char *dst, uint32_t p, int tw, const char *name) {
        // persistent grid: as many workgroups per CU as LDS and the 2048-lane limit allow (one for
        // 4096-point lines, eight for the 512-point lines of the three-pass schedule)
        if (use_wave && ps.log_len <= (uint32_t)NTTW_LOG) {
            // one wave per 256 points, four waves per workgroup, NTTW_OCC workgroups per CU
            const uint32_t lpw = 1u << (NTTW_LOG - ps.log_len);
            const uint32_t nbg = ceil_div(ceil_div(ps.nlines, lpw), NTTW_WAVES);
            // the NTTW_WAVES * lpw lines of a workgroup form a tile when they are adjacent in memory
            // (line l starts at (l >> split) * hi + (l & mask) * lo) and the pass has only whole tiles
            const uint64_t tl = (uint64_t)NTTW_WAVES * lpw;
            auto adjacent = [&](uint64_t hi, uint64_t lo) { return ps.split == 0 ? hi == 1 : (lo == 1 && tl <= ((uint64_t)1 << ps.split)); };
            if (NTTW_WAVES == 4 && ps.nlines % tl == 0 && ps.log_len >= 3)   // (the tile copy loops walk 8 columns per lane: lines of 8 points and more)
                ps.coop = (adjacent(ps.in_hi, ps.in_lo) ? 1u : 0u) | (adjacent(ps.out_hi, ps.out_lo) ? 2u : 0u);
            const uint32_t wgrid = std::min<uint32_t>(nbg, (uint32_t)std::max<size_t>(1, tuned(MIRA_TUNE_NTT_GRID, NTT_PERSISTENT_GRID * NTTW_OCC)));
            // the pass's work counters, zero -- only where the kernel will draw from them (nttw_grab: four block-groups per workgroup and
            // more): a small transform is 25 us of kernels, a memset in front of it a fifth of that
            if (nbg >= NTTW_DYNAMIC_MIN * wgrid)
                (void)rt_memset(reinterpret_cast<unsigned char *>(g.ntt_consts.p) + NTT_CTR_OFFSET + p * NTT_CTR_PASS_BYTES, 0, NTT_CTR_PASS_BYTES, g.stream);   // (a failure surfaces in rt_last below)
#define NTTW_LAUNCH(COOP)                                                                                                                                  \
    LAUNCH_BARRIER((k_ntt_wave<Fr29, COOP>), wgrid, 64 * NTTW_WAVES, NTTW_LDS_BYTES, g.stream, src, dst, ps,                                                \
                   tab + t.off_tw[p], lo_ptr(tw), tw >= 0 ? tab + t.off_hi[tw] : cnull, tw >= 0 ? cnull : (const unsigned char *)scale_d,                 \
                   reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(g.ntt_consts.p) + NTT_CTR_OFFSET + p * NTT_CTR_PASS_BYTES))
            switch (ps.coop) {
                case 0: NTTW_LAUNCH(0); break;
                case 1: NTTW_LAUNCH(1); break;
                case 2: NTTW_LAUNCH(2); break;
                default: NTTW_LAUNCH(3); break;
            }
#undef NTTW_LAUNCH
            tm_mark(name);
            return;
        }
        const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(8, (160 * 1024) / lds_for(ps.log_len)), 2048 / threads_for(ps.log_len)));
        const uint32_t grid = std::min<uint32_t>(ps.nlines, (uint32_t)std::max<size_t>(1, tuned(MIRA_TUNE_NTT_GRID, NTT_PERSISTENT_GRID * per_cu)));
#define NTTL_LAUNCH(COUNTERS)                                                                                                                              \
    LAUNCH_BARRIER((k_ntt_lines<Fr29, COUNTERS>), grid, threads_for(ps.log_len), lds_for(ps.log_len), g.stream, src, dst, ps,                                \
                   tab + t.off_tw[p], lo_ptr(tw), tw >= 0 ? tab + t.off_hi[tw] : cnull, tw >= 0 ? cnull : (const unsigned char *)scale_d,                   \
                   reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(g.ntt_consts.p) + NTT_CTR_OFFSET + p * NTT_CTR_PASS_BYTES))
        if (ps.nlines >= NTTW_DYNAMIC_MIN * grid) {
            (void)rt_memset(reinterpret_cast<unsigned char *>(g.ntt_consts.p) + NTT_CTR_OFFSET + p * NTT_CTR_PASS_BYTES, 0, NTT_CTR_PASS_BYTES, g.stream);
            NTTL_LAUNCH(true);
        } else NTTL_LAUNCH(false);
#undef NTTL_LAUNCH
        tm_mark(name);
    };
    const uint64_t n1 = (uint64_t)1 << t.m[0], n2 = (uint64_t)1 << t.m[1], n3 = (uint64_t)1 << t.m[2];
    const uint32_t NONE = 0xFFFFFFFFu;
    if (t.passes == 1) {
        run(NttPass{t.m[0], 1, 0, 0, 0, 0, 1, 0, 0, 1, NONE, 0u, 0u, 0u}, a, a, 0, -1, "ntt_single");
    } else {
        if ((rc = g.ntt_tmp.ensure(((size_t)32) << log_n))) return rc;
        unsigned char *tmp = reinterpret_cast<unsigned char *>(g.ntt_tmp.p);
        if (t.passes == 2) {
            run(NttPass{t.m[0], (uint32_t)n2, 0, 0, 1, 0, n2, n1, 0, 1, t.h[0], t.single[0], 0u, 0u}, a, tmp, 0, 0, "ntt_pass1");
            run(NttPass{t.m[1], (uint32_t)n1, 0, 0, 1, 0, n1, 1, 0, n1, NONE, 0u, 0u, 1u}, tmp, a, 1, -1, "ntt_pass2");
        } else {
            const uint64_t m = n2 * n3;
            run(NttPass{t.m[0], (uint32_t)m, 0, 0, 1, 0, m, n1, 0, 1, t.h[0], t.single[0], 0u, 0u}, a, tmp, 0, 0, "ntt_pass1");
            run(NttPass{t.m[1], (uint32_t)(n1 * n3), t.m[0], t.m[0], n1, 1, n3 * n1, n1 * n2, 1, n1, t.h[1], t.single[1], 0u, 0u}, tmp, a, 1, 1, "ntt_pass2");
            run(NttPass{t.m[2], (uint32_t)(n1 * n2), 0, 0, 1, 0, n1 * n2, 1, 0, n1 * n2, NONE, 0u, 0u, 1u}, a, a, 2, -1, "ntt_pass3");
        }
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
    if ((rc = g.ntt_consts.ensure(NTT_CONSTS_BYTES))) return rc;
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
    RT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ntt_lines<Fr29, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ntt_lines<Fr29, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ntt_wave<Fr29, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTTW_LDS_BYTES));
    RT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ntt_wave<Fr29, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTTW_LDS_BYTES));
    RT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ntt_wave<Fr29, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTTW_LDS_BYTES));
    RT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ntt_wave<Fr29, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTTW_LDS_BYTES));
#endif
    return MIRA_OK;
}
int ntt_get_omega_or_inv(uint32_t k, bool inverse, uint64_t out[4]) {
    HFr w = get_omega_or_inv_h(k, inverse);
    memcpy(out, w.l, 32);
    return MIRA_OK;
}

#ifdef NTTW_PROBE_STAMPS
// timing probe only (ntt_kernels.cuh: NTTW_STAMP): the stamp buffer of the LAST k_ntt_wave launches
extern "C" int mira_debug_ntt_stamps(uint64_t *out, size_t n_words) {
    const size_t total = (size_t)NTTW_STAMP_WGS * (NTTW_STAMP_ITERS * NTTW_STAMP_SLOTS + 1);
    if (n_words > total) n_words = total;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nttw_stamps), n_words * 8, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
extern "C" int mira_debug_ntt_clk(uint64_t *out4) { return hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_nttw_clk), 32, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1; }
extern "C" int mira_debug_ntt_stamps_clear() {
    static std::vector<uint64_t> z((size_t)NTTW_STAMP_WGS * (NTTW_STAMP_ITERS * NTTW_STAMP_SLOTS + 1), 0);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_nttw_stamps), z.data(), z.size() * 8, 0, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}
#endif
