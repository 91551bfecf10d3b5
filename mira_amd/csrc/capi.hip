// libmira_gpu.so: process-wide context + the C ABI of include/mira_gpu.h.  The host side
// mirrors the reference's compiled-language surface for this path: CommitmentKey::commit
// (src/commitment.rs:78-87) and fft / ifft / coset_fft / coset_ifft / best_fft
// (src/fft.rs:51-196).  Kernels live in the per-curve units and ntt.hip.
#include "ctx.h"
#include "host_field.hpp"
#include "glv_consts.h"
#include <cerrno>
#include <chrono>
#include <fcntl.h>
#include <sys/stat.h>
#include <thread>
#include <condition_variable>
#include <functional>
#include <unistd.h>

#ifdef MIRA_CPU_EMU
thread_local dim3 threadIdx, blockIdx;
dim3 blockDim, gridDim;
pthread_barrier_t *emu_barrier = nullptr;
unsigned char *emu_dyn_shared = nullptr;
#endif

static thread_local std::string g_err;
void set_error(const std::string &s) { g_err = s; }
static std::mutex g_lock;
Ctx g;
static std::map<uint64_t, Bases> g_bases;
static constexpr size_t PLAN_HIST_MIN_N = (size_t)1 << 15;  // below this the pre-pass (one extra sync) costs more than it can save
static constexpr size_t TABLE_MIN_N = (size_t)1 << 18;   // below this an MSM is latency-bound and the per-window path is as fast
static constexpr size_t TABLE16_MIN_N = (size_t)1 << 12;

// constants block on device: [0] gen bn256 (64 B, R form) [64] gen grumpkin (64 B, R form)
// [128] b bn256 (32 B, R' form) [160] b grumpkin (32 B, R' form)
static int upload_consts();

static int ensure_ctx() {
    if (g.ready) return MIRA_OK;
#ifndef MIRA_CPU_EMU
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) {
        set_error("no HIP device visible (libmira_gpu has no CPU fallback)");
        return MIRA_E_NO_DEVICE;
    }
    RT_CHECK(hipSetDevice(g.device));
    if (!g.stream) {
        RT_CHECK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
        g.own_stream = true;
    }
#endif
    int rc;
    if ((rc = curve_init_bn256())) return rc;
    if ((rc = curve_init_grumpkin())) return rc;
    if ((rc = ntt_init())) return rc;
    g.ready = true;
    rc = upload_consts();
    if (rc != MIRA_OK) g.ready = false;
    return rc;
}

// ------------------------------------------------------------------------------------------
// timing
void tm_begin() {
    g.tm.names.clear(); g.tm.ms.clear();
#ifndef MIRA_CPU_EMU
    if (g.tm.enabled) {
        if (g.tm.ev.empty()) {
            g.tm.ev.resize(128);
            for (auto &e : g.tm.ev) (void)hipEventCreate(&e);
        }
        (void)hipEventRecord(g.tm.ev[0], g.stream);
    }
#endif
}
void tm_mark(const char *name) {
    if (!g.tm.enabled) return;
    g.tm.names.push_back(name);
#ifndef MIRA_CPU_EMU
    if (g.tm.names.size() < g.tm.ev.size()) (void)hipEventRecord(g.tm.ev[g.tm.names.size()], g.stream);
#endif
}
void tm_end() {   // after stream sync
    if (!g.tm.enabled) return;
    g.tm.ms.assign(g.tm.names.size(), 0.f);
#ifndef MIRA_CPU_EMU
    for (size_t i = 0; i < g.tm.names.size() && i + 1 < g.tm.ev.size(); i++)
        (void)hipEventElapsedTime(&g.tm.ms[i], g.tm.ev[i], g.tm.ev[i + 1]);
#endif
}

// ------------------------------------------------------------------------------------------
// curve dispatch helpers
template <class FP> static hostf::HFe<FP> small_const(long v) {
    hostf::HFe<FP> x = hostf::from_u64<FP>((uint64_t)(v < 0 ? -v : v));
    return v < 0 ? hostf::sub(hostf::zero<FP>(), x) : x;
}
static int upload_consts() {
    uint64_t blk[32];
    memset(blk, 0, sizeof blk);
    auto gx = small_const<FqP>(1), gy = small_const<FqP>(2);
    memcpy(blk + 0, gx.l, 32); memcpy(blk + 4, gy.l, 32);
    auto hx = small_const<FrP>(1);
    // sqrt(-16) mod r = 17631683881184975370165255887551781615748388533673675138860
    hostf::HFe<FrP> hy_plain = {{0x833fc48d823f272cULL, 0x2d270d45f1181294ULL, 0xcf135e7506a45d63ULL, 0x2ULL}};
    auto hy = hostf::to_mont(hy_plain);
    memcpy(blk + 8, hx.l, 32); memcpy(blk + 12, hy.l, 32);
    // curve constants in the resident R' = 2^261 form: b * 2^261 = (32 b) * 2^256
    auto b0 = small_const<FqP>(3 * 32);
    auto b1 = small_const<FrP>(-17 * 32);
    memcpy(blk + 16, b0.l, 32); memcpy(blk + 20, b1.l, 32);
    // beta * 2^261 of each curve's endomorphism (glv.cuh): (32 beta) * 2^256, beta < 2^192
    auto beta_r261 = [](const uint64_t b[4], auto tag) {
        using FP = decltype(tag);
        hostf::HFe<FP> v = {{b[0] << 5, (b[1] << 5) | (b[0] >> 59), (b[2] << 5) | (b[1] >> 59), (b[3] << 5) | (b[2] >> 59)}};
        return hostf::to_mont(v);
    };
    auto be0 = beta_r261(Glv<FrP>::BETA, FqP{});
    auto be1 = beta_r261(Glv<FqP>::BETA, FrP{});
    memcpy(blk + 24, be0.l, 32); memcpy(blk + 28, be1.l, 32);
    int rc = g.consts.ensure(sizeof blk);
    if (rc) return rc;
    RT_CHECK(rt_h2d(g.consts.p, blk, sizeof blk, g.stream));
    RT_CHECK(rt_sync(g.stream));
    return MIRA_OK;
}

// ------------------------------------------------------------------------------------------
// MSM plan

// Estimated time of one submission in microseconds for window width c.
//
// Dense vectors: measured.  plan_wall_us[r][c] is the wall time of one commit of 2^plan_log_n[r]
// uniform scalars under width c on MI355X (tools/plan_calibrate.py, one box, one run, profiles/r03_d_plan_calibrate.txt; boxes differ
// by 5 - 10 %, the ORDER of the widths within a row is what is used).  Between rows: linear in
// log2 n; beyond the last row: proportional to n.
//
// Other vectors (witnesses: mostly zeros and short values, src/util.rs:189-193) are looked up as the
// dense vector with the same number of bucket additions: n_eff = additions(c) / W(c), from the bit
// lengths of the actual scalars (bitlen_hist, summed over the batch; null = uniform field elements).
// On top, the one effect the dense table cannot know: a bucket made heavy by the length
// distribution -- the scalars of length len share the 2^((len - 1) mod c) values their top digit can
// take, and a length that is a multiple of c always carries a 1 into the next window -- costs two
// LDS trees of general additions (5 us per level here).  A batch is count * W windows of one launch
// sequence: the additions scale, the latency does not; its W * (count - 1) * B extra counters are
// scanned at 5 800 per microsecond.
static const int plan_log_n[10] = {6, 10, 13, 15, 16, 17, 18, 19, 20, 21};   // (all rows re-measured at the END of round 4, per-window path alone -- PLAIN=1 tools/plan_calibrate.py, one box: profiles/r04_o_plan_calibrate.txt; the table of the middle of the round still had c = 8 level with 13 at 2^17, where the later tail work had moved 12 and 13 by 8 %: planned plain commits of 2^17 pairs took c = 8, 0.58 ms against 0.49)
static const double plan_wall_us[10][17] = {
    //            c = 4     5     6     7     8     9    10    11    12    13    14    15    16
    {0, 0, 0, 0,   193,   194,   229,   233,   265,   253,   294,   317,   281,   308,   466,   396,   413},
    {0, 0, 0, 0,   276,   235,   234,   241,   222,   259,   280,   287,   336,   363,   549,   475,   495},
    {0, 0, 0, 0,   300,   298,   305,   298,   275,   265,   271,   289,   298,   340,   468,   450,   483},
    {0, 0, 0, 0,   373,   389,   420,   362,   362,   387,   352,   337,   354,   352,   493,   461,   514},
    {0, 0, 0, 0,   496,   479,   499,   551,   418,   466,   485,   424,   406,   385,   488,   483,   547},
    {0, 0, 0, 0,   780,   699,   688,   718,   569,   599,   619,   565,   493,   495,   602,   579,   609},
    {0, 0, 0, 0,  1333,  1170,  1075,  1045,   854,   865,   862,   773,   720,   698,   802,   757,   775},
    {0, 0, 0, 0,  2484,  2089,  1895,  1758,  1442,  1535,  1416,  1232,  1146,  1061,  1152,  1081,  1055},
    {0, 0, 0, 0,  4911,  4127,  3654,  3255,  2757,  2781,  2492,  2218,  1987,  1860,  1927,  1747,  1743},
    {0, 0, 0, 0,  9931,  8222,  7233,  6390,  5410,  5416,  4749,  4231,  3851,  3523,  3506,  3160,  3013},
};
static double plan_table_us(uint32_t c, double n_eff) {
    const double x = std::log2(std::max(n_eff, 1.0));
    if (x <= plan_log_n[0]) return plan_wall_us[0][c];
    for (int r = 1; r < 10; r++)
        if (x <= plan_log_n[r]) {
            const double t = (x - plan_log_n[r - 1]) / (plan_log_n[r] - plan_log_n[r - 1]);
            return plan_wall_us[r - 1][c] * (1.0 - t) + plan_wall_us[r][c] * t;
        }
    return plan_wall_us[9][c] * n_eff / std::exp2((double)plan_log_n[9]);
}
// additions per MSM and the heaviest bucket load of a length distribution (h[len] scalars of bit length
// len).  Integer arithmetic only: this runs 13 times per commit on the host (with exp2 / ceil on
// doubles it cost 40 us, more than the choice of width gains at 2^16 pairs).
static void plan_len_stats(uint32_t c, const double *h, double *adds_out, double *load_out) {
    double adds = 0, load = 0;
    for (uint32_t len = 1; len < 256; len++) {
        if (h[len] == 0) continue;
        adds += h[len] * (double)((len + c - 1) / c);
        load = std::max(load, h[len] / (double)(1u << ((len - 1) % c)));
        if (len % c == 0) load = std::max(load, h[len]);
    }
    *adds_out = adds; *load_out = load;
}
// lengths of UNIFORM field elements as fractions: r = 0.756 * 2^254 -> 254: 0.339, 253: 0.331, 252: 0.165, ...
static const double *plan_uniform_fractions() {
    static double f[256];
    static bool ready = false;
    if (!ready) {
        for (int len = 0; len < 256; len++) f[len] = len > 254 ? 0.0 : len == 254 ? 0.3386 : len < 200 ? 0.0 : 0.6614 * std::exp2((double)len - 253.0);
        ready = true;
    }
    return f;
}
static double plan_heavy_us(double adds, double load, uint32_t count) {
    const double seg = std::max(16.0, adds * count / (256.0 * 4 * 3 * 64));
    const double partials = load / seg;
    return partials > 6.0 ? 5.0 * (std::ceil(std::log2(partials)) + 3.0) : 0.0;
}
static double plan_cost_us(uint32_t c, double n, uint32_t count, const double *bitlen_hist /* per MSM, or null = uniform */) {
    const double W = std::ceil(256.0 / c), B = (double)(1u << (c - 1));
    double heavy = 0, n_eff = n;
    if (bitlen_hist) {
        double adds, load, u_adds, u_load;
        plan_len_stats(c, bitlen_hist, &adds, &load);        // scalars of each length, per MSM
        // the dense vector with as many additions: the table was measured on UNIFORM field elements, which have u_adds non-zero
        // digits each -- not W (254 bits under 15-bit windows: 17 digits in 18 windows; dividing by W made a uniform 2^22-pair
        // vector look 6 % shorter under c = 15 than the vector the table was measured on, and 15 won over the faster 16)
        plan_len_stats(c, plan_uniform_fractions(), &u_adds, &u_load);
        n_eff = adds / u_adds;
        // the measured table already holds what the top window of UNIFORM field elements costs: only the
        // excess over a uniform vector with as many additions counts
        heavy = std::max(0.0, plan_heavy_us(adds, load, count) - plan_heavy_us(u_adds * n_eff, u_load * n_eff, count));
    }
    return plan_table_us(c, n_eff * count) + heavy + W * (count - 1) * B / 5800.0;
}

// The GLV split (glv.cuh) has a table of its own: wall time in microseconds of one commit of 2^glv_log_n[r] uniform pairs -- twice
// as many half-length scalars -- under width c (tools/glv_probe.py --calibrate; re-measured at the end of round 4, profiles/r04_o_plan_calibrate.txt).  The widths that cut
// 128 bits evenly stand out (9 at 2^17, 13 at 2^18 - 2^19, 16 beyond): a last window that holds only a few bits of every half is a
// handful of very heavy buckets.
static const int glv_log_n[11] = {10, 12, 14, 15, 16, 17, 18, 19, 20, 21, 22};
static const double glv_wall_us[11][17] = {
    //                c = 5      6      7      8      9     10     11     12     13     14     15     16
    {    0,     0,     0,     0,     0,   258,   231,   209,   201,   229,   222,   256,   258,   287,   383,   497,   484},
    {    0,     0,     0,     0,     0,   247,   245,   250,   225,   243,   226,   247,   273,   261,   351,   487,   491},
    {    0,     0,     0,     0,     0,   297,   312,   301,   285,   342,   305,   295,   299,   290,   364,   450,   431},
    {    0,     0,     0,     0,     0,   360,   364,   349,   319,   339,   365,   330,   306,   309,   387,   470,   487},
    {    0,     0,     0,     0,     0,   483,   451,   428,   437,   397,   473,   470,   384,   332,   403,   479,   537},
    {    0,     0,     0,     0,     0,   731,   640,   601,   589,   533,   594,   599,   496,   441,   509,   571,   623},
    {    0,     0,     0,     0,     0,  1169,  1025,   939,   877,   862,   868,   835,   693,   627,   703,   745,   759},
    {    0,     0,     0,     0,     0,  2116,  1857,  1663,  1467,  1465,  1387,  1308,  1117,  1017,  1078,  1108,  1079},
    {    0,     0,     0,     0,     0,  4043,  3548,  3177,  2803,  2799,  2505,  2297,  1998,  1807,  1829,  1841,  1741},   // (c < 9: not measured, scaled from the 2^19 row)
    {    0,     0,     0,     0,     0,  8188,  7186,  6435,  5677,  5669,  5023,  4617,  4092,  3643,  3666,  3548,  3264},   // (c < 9: not measured, scaled from the 2^19 row)
    {    0,     0,     0,     0,     0, 16300, 14305, 12810, 11300, 11285,  9870,  8977,  8082,  7357,  7152,  6959,  6222},   // (c < 9: not measured, scaled from the 2^19 row)
};
static double glv_table_us(uint32_t c, double pairs) {
    const double x = std::log2(std::max(pairs, 1.0));
    if (x <= glv_log_n[0]) return glv_wall_us[0][c];
    for (int r = 1; r < 11; r++)
        if (x <= glv_log_n[r]) {
            const double t = (x - glv_log_n[r - 1]) / (glv_log_n[r] - glv_log_n[r - 1]);
            return glv_wall_us[r - 1][c] * (1.0 - t) + glv_wall_us[r][c] * t;
        }
    return glv_wall_us[10][c] * pairs / std::exp2((double)glv_log_n[10]);
}
// n halves (2 x the pairs); with the bit lengths of the previous commit's halves: the dense commit with as many bucket additions,
// plus what its heavy buckets cost beyond a uniform vector's (as plan_cost_us and pick_shared do)
static double glv_cost_us(uint32_t c, double n_halves, const double *hist, uint32_t count = 1) {
    const uint32_t W = (GLV_BITS + c - 1) / c;
    double pairs = n_halves / 2, heavy = 0;
    if (hist) {
        double adds, load;
        plan_len_stats(c, hist, &adds, &load);
        const double dense_halves = std::max(1.0, adds / W);
        pairs = dense_halves / 2;
        static double uniform[256];                          // bit lengths of a magnitude uniform below 2^126 (what the table was measured on)
        if (uniform[126] == 0)
            for (int len = 1; len <= 126; len++) uniform[len] = std::exp2((double)len - 127.0);
        double u_adds, u_load;
        plan_len_stats(c, uniform, &u_adds, &u_load);
        heavy = std::max(0.0, plan_heavy_us(adds, load, 1) - plan_heavy_us(u_adds * dense_halves, u_load * dense_halves, 1));
    }
    // a batch: the commits' additions in one launch, and W 2^(c-1) more buckets to reduce per further commit (as plan_cost_us)
    return glv_table_us(c, pairs * count) + heavy + (double)W * (count - 1) * (double)(1u << (c - 1)) / 5800.0;
}

static MsmPlan make_plan(size_t n, int32_t forced_c, uint32_t count = 1, uint64_t stride = 0, const uint32_t *bitlen_hist = nullptr, uint32_t bits = 256) {
    MsmPlan p;
    uint32_t best_c = 13;
    double best = 1e300;
    double per_msm[256];
    if (bitlen_hist)
        for (int len = 0; len < 256; len++) per_msm[len] = (double)bitlen_hist[len] / count;
    for (uint32_t c = (bits == 256 ? 4 : 5); c <= 16 && !forced_c; c++) {
        const double cost = bits == 256 ? plan_cost_us(c, (double)n, count, bitlen_hist ? per_msm : nullptr)
                                        : glv_cost_us(c, (double)n, bitlen_hist ? per_msm : nullptr, count);   // the halves of the GLV split: their own table
        if (cost < best * 0.99) { best = cost; best_c = c; }    // ties go to the narrower window (fewer buckets: less that skewed data can upset)
    }
    p.c = forced_c ? (uint32_t)forced_c : best_c;
    p.est_us = forced_c ? 0.0 : best;
    p.W = (bits + p.c - 1) / p.c;
    p.B = 1u << (p.c - 1);
    p.count = count; p.stride = stride; p.Wt = p.W * count;
    p.NB = p.Wt * p.B;
    // histogram / scatter tiling: about one workgroup per CU, at least 1024 points per tile (msm_host.cuh tiles each point chunk the same way)
    uint32_t want_tiles = std::max<uint32_t>(1, MSM_HIST_WGS / p.Wt);
    p.tile = std::max<uint32_t>(1024, ceil_div(n, want_tiles));
    p.tile = (p.tile + 1023) / 1024 * 1024;
    p.ntiles = ceil_div(n, p.tile);
    // accumulate: one segment of consecutive sorted entries per resident lane (k_plan fixes the
    // segment length on the device from the number of non-zero digits); 142 VGPRs -> 3 waves/SIMD
    uint64_t entries = (uint64_t)n * p.Wt;
    p.lanes = 256u * 4u * 3u * 64u;
    p.L = (uint32_t)tuned(MIRA_TUNE_MIN_SEGMENT, 10);   // minimum segment length: more, shorter segments keep the lanes of a small commit busy (16 -> 10: 2^15 pairs 0.40 -> 0.35 ms), below 10 the cut runs cost the fix-up more than the additions gain (tools/min_segment_probe.py)
    p.T = (uint32_t)std::min<uint64_t>(p.lanes, ceil_div(entries, p.L));   // upper bound of segments
    plan_reduction(p, 1);                                    // callers that can take several pieces per window ask again
    return p;
}
// Pieces per bucket set for a commit whose points the library's own epilogue combines (capi.hip: horner_pieces): the device's
// Horner chain over the bits of a bucket index is cut into P parts and the host's chain of doublings, which passes every bit
// position anyway, adds P points per window instead of one (0.25 us each).
static uint32_t default_pieces(const MsmPlan &p, uint32_t max_points) {
    uint32_t P = (uint32_t)tuned(MIRA_TUNE_REDUCE_PIECES, 3);
    const uint32_t sets_per_result = p.shared ? 1u : p.W;
    while (P > 1 && sets_per_result * P > max_points) P--;
    return std::max(1u, P);
}

// Shared-bucket fixed-base tables (mira_msm_precompute_ex(handle, c), c = 8 .. 16): W = ceil(256 / c) signed c-bit
// digits per scalar against the tables 2^(c w) P_i, ONE set of 2^(c-1) buckets for all windows, `sums` partial
// sums back (no chain of doublings on the host).  A commit then pays ceil(256 / c) additions per pair and the
// fix-up / bucket reduction of ONE window of 2^(c-1) buckets: narrow widths for the small commits of a fold
// step (few buckets: a short tail), 16 bits for the large ones (fewest additions).
static MsmPlan make_plan_shared(size_t n, const Bases::SharedSet &set, uint64_t table_n, uint32_t count = 1, uint64_t stride = 0) {
    MsmPlan p = make_plan(n, (int32_t)set.c, count, stride);
    p.shared = true; p.shared_tables = set.p; p.table_n = table_n;
    p.NB = count * p.B;                                      // one bucket set per MSM
    plan_reduction(p, 1);
    return p;
}
// Which of a key's shared-bucket sets serves a commit of n pairs (count of them in one submission).  Measured
// (tools/shared_width_probe.py, profiles/r03_d_shared_widths.txt): wall time of one commit in microseconds under width c at
// 2^12 .. 2^21 pairs, interpolated in log2 n like the per-window planner's table; beyond the last row proportional to
// n.  MIRA_TUNE_TABLE_WIDTH names a width outright (calibration, tests).
static const int shared_log_n[6] = {12, 14, 16, 17, 19, 21};
static const double shared_wall_us[6][17] = {
    //                          c = 8     9    10    11    12    13    14    15    16      (re-measured at the end of round 4: profiles/r04_o_plan_calibrate.txt)
    {0, 0, 0, 0, 0, 0, 0, 0,   199,   216,   208,   233,   223,   212,   239,   244,   296},
    {0, 0, 0, 0, 0, 0, 0, 0,   261,   270,   283,   294,   315,   308,   278,   269,   293},
    {0, 0, 0, 0, 0, 0, 0, 0,   355,   337,   348,   381,   448,   472,   434,   368,   423},
    {0, 0, 0, 0, 0, 0, 0, 0,   520,   501,   491,   498,   564,   534,   557,   468,   497},
    {0, 0, 0, 0, 0, 0, 0, 0,  1522,  1544,  1404,  1318,  1306,  1196,  1189,  1026,  1021},
    {0, 0, 0, 0, 0, 0, 0, 0,  5978,  5853,  5072,  4661,  4298,  3998,  3751,  3341,  3243},
};
static double shared_cost_us(uint32_t c, double n) {
    const double x = std::log2(std::max(n, 1.0));
    if (x <= shared_log_n[0]) return shared_wall_us[0][c];
    for (int r = 1; r < 6; r++)
        if (x <= shared_log_n[r]) {
            const double t = (x - shared_log_n[r - 1]) / (shared_log_n[r] - shared_log_n[r - 1]);
            return shared_wall_us[r - 1][c] * (1.0 - t) + shared_wall_us[r][c] * t;
        }
    return shared_wall_us[5][c] * n / std::exp2((double)shared_log_n[5]);
}
// sharded: every rank must pick the same set whatever its chunk length -> the widest.  bitlen_hist (or null): the bit
// lengths of the scalars of the previous commit of this shape -- a witness vector (mostly zeros and short values) is looked
// up as the dense vector with as many bucket additions, as the per-window planner does (1.8 M witness scalars are 0.23 M
// dense ones under 16-bit windows: a narrow set serves them, not the 16-bit one their length suggests).
static const Bases::SharedSet *pick_shared(const Bases &bs, size_t n, uint32_t count, bool sharded, const uint32_t *bitlen_hist = nullptr) {
    if (bs.shared.empty()) return nullptr;
    const size_t forced = tuned(MIRA_TUNE_TABLE_WIDTH, 0);
    const Bases::SharedSet *best = nullptr;
    double best_us = 1e300, h[256];
    if (bitlen_hist)
        for (int len = 0; len < 256; len++) h[len] = (double)bitlen_hist[len];
    for (const auto &set : bs.shared) {
        if (forced) { if (set.c == forced) return &set; continue; }
        double n_eff = (double)n * count, heavy = 0;
        if (bitlen_hist) {
            double adds, load, u_adds, u_load;
            plan_len_stats(set.c, h, &adds, &load);
            // the dense vector with as many additions: the table was measured on UNIFORM field elements, which have u_adds non-zero
            // digits each, not W (a 15-bit set has 18 tables for the 17 digits of a uniform scalar) -- as plan_cost_us counts
            plan_len_stats(set.c, plan_uniform_fractions(), &u_adds, &u_load);
            n_eff = std::max(1.0, adds / u_adds);
            // ... plus what the length distribution makes heavy beyond a uniform vector with as many additions (the measured table
            // holds the latter): 32-bit witness values under 15-bit windows share TWO top-digit values, under 16-bit windows all
            // carry a one into the third window, under 13-bit windows they spread over 32
            heavy = std::max(0.0, plan_heavy_us(adds, load, 1) - plan_heavy_us(u_adds * n_eff, u_load * n_eff, 1));
        }
        // a batch is count bucket sets to reduce: ~1 ns per bucket of every further set (6 x 2^15 buckets: 0.19 ms of k_reduce_chunks
        // against 0.03 for one set; profiles/r03_d_batch_tables.txt)
        const double us = sharded ? -(double)set.c : shared_cost_us(set.c, n_eff) + heavy + (count - 1) * (double)(1u << (set.c - 1)) / 1000.0;
        if (us < best_us) { best_us = us; best = &set; }
    }
    return best;
}

// A few resident host threads for the independent epilogues of a batch (creating and joining five threads per batch cost more
// than the 45 us chain each of them ran).  The pool is created on first use and never destroyed: its threads sleep on a
// condition variable until the process ends.  Callers hold the ABI lock, so there is one parallel_for at a time.
#ifndef MIRA_CPU_EMU
namespace {
struct HostPool {
    std::mutex m;
    std::condition_variable wake, done_cv;
    const std::function<void(size_t)> *fn = nullptr;
    size_t next = 0, count = 0, running = 0;
    uint64_t epoch = 0;
    std::vector<std::thread> threads;
    explicit HostPool(size_t n) {
        for (size_t t = 0; t < n; t++)
            threads.emplace_back([this] {
                uint64_t seen = 0;
                std::unique_lock<std::mutex> lk(m);
                for (;;) {
                    wake.wait(lk, [&] { return epoch != seen && next < count; });
                    seen = epoch;
                    while (next < count) {
                        const size_t i = next++;
                        running++;
                        lk.unlock();
                        (*fn)(i);
                        lk.lock();
                        running--;
                    }
                    if (running == 0) done_cv.notify_all();
                }
            });
        for (auto &t : threads) t.detach();
    }
};
}   // namespace
#endif
#ifndef MIRA_CPU_EMU
static HostPool &host_pool() {
    static HostPool *pool = new HostPool(std::max<size_t>(1, std::min<size_t>(7, std::thread::hardware_concurrency() > 1 ? std::thread::hardware_concurrency() - 1 : 1)));
    return *pool;                                            // never destroyed: its threads wait on it until the process ends
}
#endif
// threads a parallel region can count on, the caller included
static size_t host_parallel_width() {
#ifdef MIRA_CPU_EMU
    return 1;
#else
    return host_pool().threads.size() + 1;
#endif
}
static void host_parallel_for(size_t count, const std::function<void(size_t)> &fn) {
#ifdef MIRA_CPU_EMU
    for (size_t i = 0; i < count; i++) fn(i);
#else
    if (count <= 1) { if (count) fn(0); return; }
    HostPool *pool = &host_pool();
    static std::mutex one_region;                           // callers outside the library lock (mira_g1_*) take turns
    std::lock_guard<std::mutex> region(one_region);
    {
        std::lock_guard<std::mutex> lk(pool->m);
        pool->fn = &fn; pool->next = 0; pool->count = count; pool->epoch++;
    }
    pool->wake.notify_all();
    std::unique_lock<std::mutex> lk(pool->m);
    while (pool->next < pool->count) {                      // the caller works too
        const size_t i = pool->next++;
        pool->running++;
        lk.unlock();
        fn(i);
        lk.lock();
        pool->running--;
    }
    pool->done_cv.wait(lk, [&] { return pool->running == 0; });
    pool->count = 0;
#endif
}

// The epilogue of a commit: sum_(w < W) sum_(p < P) 2^(c w + piece_start(cb, P, p)) pts[w P + p] by ONE chain of doublings from the
// highest bit position down (Horner), then to_affine.  P = 1 is the plain sum over window sums; c = 0, P = 1 a plain sum of
// partial sums (wide tables).
template <class FB>
static void horner_pieces(const uint64_t *pts, const PartialShape &sh, uint64_t out[8]) {
    using namespace hostf;
    HXyzz<FB> acc = identity<FB>();
    uint32_t at = 0;                                         // bit position the accumulator currently stands at
    bool first = true;
    for (int w = (int)sh.W - 1; w >= 0; w--)
        for (int p = (int)sh.P - 1; p >= 0; p--) {
            const uint32_t pos = sh.c * (uint32_t)w + piece_start(sh.cb, sh.P, (uint32_t)p);
            if (!first) for (uint32_t k = pos; k < at; k++) acc = dbl_pt(acc);
            HXyzz<FB> t;
            memcpy(&t, pts + ((size_t)w * sh.P + (size_t)p) * 16, 128);
            acc = add_pt(acc, t);
            at = pos; first = false;
        }
    for (uint32_t k = 0; k < at; k++) acc = dbl_pt(acc);     // (the lowest piece of window 0 stands at bit 0: nothing to do)
    to_affine(acc, out);
}
template <class FB>
static void sum_partials(const uint64_t *partials, size_t nparts, uint32_t W, uint64_t *out_windows) {
    using namespace hostf;
    for (uint32_t w = 0; w < W; w++) {
        HXyzz<FB> acc = identity<FB>();
        for (size_t k = 0; k < nparts; k++) {
            HXyzz<FB> t;
            memcpy(&t, partials + k * MIRA_PARTIAL_U64 + (size_t)w * 16, 128);
            acc = add_pt(acc, t);
        }
        memcpy(out_windows + (size_t)w * 16, &acc, 128);
    }
}

// sharded: the caller is one rank of a point-chunk sharded MSM.  All ranks must produce the same
// kind of partial, so the choice between table and per-window mode then depends only on whether
// the handle has tables (and on the forced width), never on this rank's chunk length.
// The endomorphism copy of a key, built the first time a commit takes the GLV split (MIRA_TUNE_GLV_AUTO_MAX_LOG): the split
// halves the windows -- half the bucket reduction, half the host's chain of doublings -- for 2 x the key's memory and one
// streaming kernel.  Keys shorter than 2^12 points are not worth a copy.  A failed allocation leaves the key as it is.
static constexpr size_t GLV_AUTO_MIN_KEY = (size_t)1 << 12;
static bool glv_possible(const Bases &bs) {                  // a copy exists, or the library may build one for this key
    if (tuned(MIRA_TUNE_GLV, 1) == 0) return false;
    if (bs.glv) return true;
    const size_t max_log = tuned(MIRA_TUNE_GLV_AUTO_MAX_LOG, 26);
    return !bs.glv_auto_failed && max_log != 0 && bs.n >= GLV_AUTO_MIN_KEY && bs.n <= ((size_t)1 << std::min<size_t>(max_log, 30));
}
static bool glv_ready(const Bases &bs) {                     // ... and it is there now
    if (!glv_possible(bs)) return false;
    if (bs.glv) return true;
    Bases &mut = const_cast<Bases &>(bs);
    const unsigned char *consts = reinterpret_cast<const unsigned char *>(g.consts.p);
    const int rc = bs.curve == MIRA_CURVE_BN256 ? build_glv_bn256(mut, consts + 192) : build_glv_grumpkin(mut, consts + 224);
    if (rc != MIRA_OK) { bs.glv_auto_failed = true; (void)rt_last(); return false; }
    return bs.glv != nullptr;
}
// Plain path or GLV split for this commit?  With a forced width: the split wherever the key has (or may get) its copy, as
// before.  Planned: both planners are asked -- their tables are measured walls of the two paths (tools/plan_calibrate.py,
// tools/glv_probe.py --calibrate) -- and the split must be ahead by 2 %: it wins up to ~2^19 pairs (2^17: 0.49 against 0.53 ms)
// and for the batches of a fold step, and loses from 2^20 on, where the decomposition in k_digits and the doubled point
// stream cost more than the halved bucket reduction saves (profiles/r04_c_glv.txt).
static bool choose_glv(const Bases &bs, const MsmPlan &plain, const MsmPlan &split, size_t pairs) {
    if (!glv_possible(bs)) return false;
    if (plain.est_us > 0 && split.est_us > 0) {
        if (split.est_us >= 0.98 * plain.est_us) return false;
    } else if (!bs.glv && pairs > ((size_t)1 << 19)) return false;     // forced width, no estimates: a copy the caller asked for is used; none is built for sizes the split loses at
    return glv_ready(bs);
}
// ---- width trials (ctx.h: Bases::WidthTrial) ------------------------------------------------------------------------------------
static constexpr size_t TRIAL_MIN_N = (size_t)1 << 12;
static constexpr int TRIAL_RUNS = 2;
static constexpr int TRIAL_OFFSETS[5] = {0, +1, -1, +2, -2};    // the model's width, then its neighbours: the landscape has bumps (a width that leaves a two-bit top window), so all five are measured rather than walked
static Bases::WidthTrial *trial_for(const Bases &bs, size_t n, uint32_t count, uint32_t kind, uint32_t c_model) {
    if (tuned(MIRA_TUNE_WIDTH_TRIALS, 1) == 0 || n * count < TRIAL_MIN_N) return nullptr;
    for (auto &t : bs.trials)
        if (t.n == n && t.count == count && t.kind == kind) { t.stamp = ++bs.trial_stamp; return &t; }
    if (bs.trials.size() >= 12) {                            // a key sees a handful of shapes; the least recently used one goes
        size_t lru = 0;
        for (size_t i = 1; i < bs.trials.size(); i++) if (bs.trials[i].stamp < bs.trials[lru].stamp) lru = i;
        bs.trials.erase(bs.trials.begin() + (long)lru);
    }
    Bases::WidthTrial t;
    t.n = n; t.count = count; t.kind = kind; t.c0 = t.best_c = t.cur_c = c_model; t.stamp = ++bs.trial_stamp;
    bs.trials.push_back(t);
    return &bs.trials.back();
}
static uint32_t trial_width(const Bases::WidthTrial &t) { return t.done ? t.best_c : t.cur_c; }
// the wall time of the commit that ran under trial_width(t)
static void trial_report(Bases::WidthTrial &t, double us, uint32_t c_min, uint32_t c_max) {
    if (t.done) return;
    t.cur_us = t.cur_runs == 0 ? us : std::min(t.cur_us, us);
    if (++t.cur_runs < TRIAL_RUNS) return;
    if (t.best_us == 0 || t.cur_us < 0.98 * t.best_us) { t.best_us = t.cur_us; t.best_c = t.cur_c; }   // a neighbour must be ahead by more than the noise
    for (t.steps++; t.steps < 5; t.steps++) {
        const int c = (int)t.c0 + TRIAL_OFFSETS[t.steps];
        if (c >= (int)c_min && c <= (int)c_max) { t.cur_c = (uint32_t)c; t.cur_runs = 0; return; }
    }
    t.done = true;
}

// ... and the same among a key's shared-bucket table sets: the model (pick_shared) ranks them for dense vectors; for the witness
// vectors of a fold step it was 12 % off (14 x 2^17 scalars: the 11-bit set, 0.82 ms, where the 15-bit one takes 0.73).  A shape's first
// commits go through every set the key has, twice each, and the fastest is kept (kind bit 2 marks these records; steps = the set's
// position in the key's list).
static const Bases::SharedSet *trial_set(const Bases &bs, const Bases::WidthTrial &t, const Bases::SharedSet *model) {
    const uint32_t c = t.done ? t.best_c : t.cur_c;
    for (const auto &set : bs.shared) if (set.c == c) return &set;
    return model;
}
static void trial_report_sets(Bases::WidthTrial &t, double us, const Bases &bs) {
    if (t.done) return;
    t.cur_us = t.cur_runs == 0 ? us : std::min(t.cur_us, us);
    if (++t.cur_runs < TRIAL_RUNS) return;
    if (t.best_us == 0 || t.cur_us < 0.98 * t.best_us) { t.best_us = t.cur_us; t.best_c = t.cur_c; }
    for (; (size_t)t.steps < bs.shared.size(); t.steps++)
        if (bs.shared[(size_t)t.steps].c != t.c0) { t.cur_c = bs.shared[(size_t)t.steps].c; t.cur_runs = 0; t.steps++; return; }   // (the model's set went first)
    t.done = true;
}

// allow_pieces: the caller combines the points itself with horner_pieces (a commit of this process); else the public partial
// format, one point per window (*shape then has P = 1).
static int msm_partial_locked(uint64_t handle, size_t first, const void *d_scalars, size_t n, uint64_t *out_partial,
                              PartialShape *shape, bool sharded = false, int32_t requested_c = 0, const void *h_scalars = nullptr, bool allow_pieces = false) {
    int rc = ensure_ctx();
    if (rc) return rc;
    auto it = g_bases.find(handle);
    if (it == g_bases.end()) { set_error("unknown bases handle"); return MIRA_E_BAD_ARG; }
    const Bases &bs = it->second;
    if (first > bs.n || n > bs.n - first) {
        set_error("Can't commit too long input: input len: " + std::to_string(first + n) + ", but limit is " + std::to_string(bs.n));
        return MIRA_E_TOO_LONG;
    }
    // window width: the call's own (sharded partials), else this key's (mira_msm_set_handle_window_bits), else the process default
    const int32_t forced_c = bs.forced_c ? bs.forced_c : g.forced_c;
    // 20- / 22-bit tables pay from 2^18 pairs (2^19 buckets to reduce whatever n is); the shared-bucket sets have the
    // bucket count of ONE window of the per-window path, so they win from a few thousand pairs.  A key with both
    // uses the wide tables where they pay and a shared set below.
    const bool tables_ok = forced_c == 0 && requested_c == 0;
    const bool table_mode = bs.tables && tables_ok && (sharded || n >= tuned(MIRA_TUNE_TABLE_MIN_N, TABLE_MIN_N));
    // Data-dependent planning for single (unsharded) commits (ranks of a sharded MSM must agree on
    // the window width, so they keep the dense estimate).  The statistics are those of the previous
    // commit of the same length over this key -- successive fold steps commit witnesses of one
    // shape -- so no call waits for a pre-pass: this call's histogram is enqueued ahead of its MSM
    // kernels and read after the synchronisation that ends it.
    const size_t hist_min_n = tuned(MIRA_TUNE_PLAN_HIST_MIN_N, PLAN_HIST_MIN_N);
    const bool can_hist = !table_mode && !sharded && forced_c == 0 && requested_c == 0 && n >= hist_min_n && d_scalars;
    // statistics are consumed only by the kind of path that collected them: the halves of the GLV split have other lengths than
    // the scalars they come from
    const uint32_t *stat_any = (can_hist && bs.stat_n == n) ? bs.stat_hist : nullptr;
    const uint32_t *stat_full = bs.stat_kind == 0 ? stat_any : nullptr;
    const Bases::SharedSet *set = (tables_ok && !table_mode && (sharded || n >= tuned(MIRA_TUNE_SHARED_MIN_N, TABLE16_MIN_N))) ? pick_shared(bs, n, 1, sharded, stat_full) : nullptr;
    const bool use_hist = can_hist && !set;
    // a rank of a sharded MSM that was not given a width takes 16, whatever its chunk length: partials
    // of different widths cannot be combined, and chunk lengths differ between ranks
    const int32_t width = requested_c ? requested_c : (sharded && forced_c == 0) ? 16 : forced_c;
    // the GLV split (glv.cuh): 2 n half-length scalars over the interleaved key; not for ranks of a sharded MSM (their partials
    // must have one shape whatever each rank's key holds) nor beside a table set
    const bool glv_ok = !set && !table_mode && !sharded && n != 0 && n < (1ull << 30) && glv_possible(bs);
    const MsmPlan p_plain = make_plan(n, width, 1, 0, (use_hist && bs.stat_kind == 0) ? stat_any : nullptr);
    const MsmPlan p_split = glv_ok ? make_plan(2 * n, width, 1, 0, (use_hist && bs.stat_kind == 1) ? stat_any : nullptr, GLV_BITS) : p_plain;
    const bool glv = glv_ok && choose_glv(bs, p_plain, p_split, n);
    MsmPlan p = glv ? p_split : p_plain;
    // the planner's width for this shape, checked against its neighbours on the first commits of the shape (trial_*)
    // (not before the scalar statistics of the shape exist where they are collected: the model's width for a witness vector
    // without them is the dense vector's, too far from the best one for its neighbourhood to hold it)
    const bool stats_pending = use_hist && !stat_any;
    Bases::WidthTrial *trial = (width == 0 && !sharded && !set && !table_mode && n && !stats_pending) ? trial_for(bs, n, 1, (glv ? 1u : 0u) | (h_scalars ? 2u : 0u), p.c) : nullptr;
    if (trial && trial_width(*trial) != p.c)
        p = glv ? make_plan(2 * n, (int32_t)trial_width(*trial), 1, 0, nullptr, GLV_BITS) : make_plan(n, (int32_t)trial_width(*trial), 1, 0, nullptr);
    p.glv = glv; p.glv_bases = glv ? bs.glv : nullptr;
    // (a commit of n W >= 2^32 entries is cut into point chunks inside the launch sequence, msm_host.cuh; the 31-bit limit is the
    // point index of a sorted entry, the sign in bit 31)
    if (n >= (1ull << 31)) { set_error("n too large for 31-bit point indices"); return MIRA_E_UNSUPPORTED; }
    if (p.W > MIRA_MAX_WINDOWS) { set_error("window configuration exceeds MIRA_MAX_WINDOWS"); return MIRA_E_UNSUPPORTED; }
    if (allow_pieces && !g.windows_dst) plan_reduction(p, default_pieces(p, MIRA_MAX_WINDOWS));
    *shape = PartialShape{p.c, p.W, p.cb, p.pieces};
    g.last_c = (int32_t)p.c; g.last_w = (int32_t)p.W; g.last_table_c = 0;
    memset(out_partial, 0, MIRA_PARTIAL_U64 * 8);
    if (n == 0) {
        // an empty chunk (a rank beyond the prefix being committed) answers with the identity in the SHAPE its mode has for any
        // length: the ranks of a sharded MSM exchange and combine partials of one shape (mira_amd/dist.py checks it)
        if (set) { *shape = PartialShape{0, 1, set->c - 1, 1}; g.last_c = 0; g.last_w = 1; g.last_table_c = (int32_t)set->c; }
        else if (table_mode) { *shape = PartialShape{0, 64, 0, 1}; g.last_c = 0; g.last_w = 64; g.last_table_c = (int32_t)bs.table_c; }
        return MIRA_OK;
    }
    if (!d_scalars) { set_error("null scalars"); return MIRA_E_BAD_ARG; }
    // fixed-base mode: window tables present, MSM large enough to be throughput-bound, no forced width
    if (set) {                                               // shared buckets through the per-window launch sequence
        Bases::WidthTrial *strial = (!sharded && bs.shared.size() > 1 && tuned(MIRA_TUNE_TABLE_WIDTH, 0) == 0 && !(can_hist && !stat_any))
                                        ? trial_for(bs, n, 1, 4u | (h_scalars ? 2u : 0u), set->c) : nullptr;
        if (strial) set = trial_set(bs, *strial, set);
        const auto t_set = std::chrono::steady_clock::now();
        MsmPlan ps = make_plan_shared(n, *set, bs.n);
        if (allow_pieces && !g.windows_dst) plan_reduction(ps, default_pieces(ps, MIRA_MAX_WINDOWS));
        *shape = PartialShape{0, 1, ps.cb, ps.pieces};       // the pieces of ONE bucket set (P = 1: its sum)
        g.last_c = 0; g.last_w = (int32_t)ps.pieces; g.last_table_c = (int32_t)set->c;
        ps.stats = can_hist;
        rc = bs.curve == MIRA_CURVE_BN256 ? msm_launch_bn256(bs, first, d_scalars, h_scalars, n, ps, out_partial)
                                          : msm_launch_grumpkin(bs, first, d_scalars, h_scalars, n, ps, out_partial);
        if (rc == MIRA_OK && strial) trial_report_sets(*strial, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_set).count(), bs);
        if (rc == MIRA_OK && can_hist) {                     // msm_launch ends with a stream synchronisation
            memcpy(bs.stat_hist, g.hist_host, sizeof bs.stat_hist);
            bs.stat_n = n; bs.stat_kind = 0;
        }
        return rc;
    }
    if (table_mode) {
        if (h_scalars) RT_CHECK(rt_h2d(const_cast<void *>(d_scalars), h_scalars, n * 32, g.stream));
        *shape = PartialShape{0, 64, 0, 1};                 // 64 partial sums, combined by a plain sum
        g.last_c = 0; g.last_w = 64; g.last_table_c = (int32_t)bs.table_c;
        return bs.curve == MIRA_CURVE_BN256 ? msm_launch_table_bn256(bs, first, d_scalars, n, out_partial)
                                             : msm_launch_table_grumpkin(bs, first, d_scalars, n, out_partial);
    }
    p.stats = use_hist;
    const auto t_launch = std::chrono::steady_clock::now();
    rc = bs.curve == MIRA_CURVE_BN256 ? msm_launch_bn256(bs, first, d_scalars, h_scalars, n, p, out_partial) : msm_launch_grumpkin(bs, first, d_scalars, h_scalars, n, p, out_partial);
    if (rc == MIRA_OK && trial)
        trial_report(*trial, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_launch).count(), glv ? 5 : 4, 16);
    if (rc == MIRA_OK && use_hist) {                         // msm_launch ends with a stream synchronisation
        memcpy(bs.stat_hist, g.hist_host, sizeof bs.stat_hist);
        bs.stat_n = n; bs.stat_kind = glv ? 1 : 0;
    }
    return rc;
}

static int combine_locked(int curve, const uint64_t *partials, size_t nparts, uint32_t c, uint32_t W, uint64_t out[8]) {
    if (curve != MIRA_CURVE_BN256 && curve != MIRA_CURVE_GRUMPKIN) { set_error("unknown curve"); return MIRA_E_BAD_ARG; }
    if (!partials || !out || nparts == 0 || c > 16 || W < 1 || W > MIRA_MAX_WINDOWS) { set_error("bad combine arguments"); return MIRA_E_BAD_ARG; }
    std::vector<uint64_t> win((size_t)W * 16);
    const PartialShape sh{c, W, c ? c - 1 : 0, 1};
    if (curve == MIRA_CURVE_BN256) { sum_partials<FqP>(partials, nparts, W, win.data()); horner_pieces<FqP>(win.data(), sh, out); }
    else { sum_partials<FrP>(partials, nparts, W, win.data()); horner_pieces<FrP>(win.data(), sh, out); }
    return MIRA_OK;
}


// count MSMs over the prefix of one key in a single pass of the pipeline (a batch is count * W
// windows).  Chunked so that the window-counter scan and the 32-bit entry offsets stay in range.
static int msm_batch_device_locked(uint64_t handle, const void *d_scalars, size_t n, size_t count, size_t stride, uint64_t *out_affine,
                                   const uint64_t *const *h_batch = nullptr /* the vectors are still in host memory: d_scalars is their staging buffer */) {
    int rc = ensure_ctx();
    if (rc) return rc;
    auto it = g_bases.find(handle);
    if (it == g_bases.end()) { set_error("unknown bases handle"); return MIRA_E_BAD_ARG; }
    const Bases &bs = it->second;
    if (!out_affine || (count && n && !d_scalars) || (count > 1 && stride < n)) { set_error("bad batch arguments"); return MIRA_E_BAD_ARG; }
    if (n > bs.n) {
        set_error("Can't commit too long input: input len: " + std::to_string(n) + ", but limit is " + std::to_string(bs.n));
        return MIRA_E_TOO_LONG;
    }
    if (n == 0) { memset(out_affine, 0, count * 64); return MIRA_OK; }
    if (n >= (1ull << 31)) { set_error("n too large for 32-bit entry offsets"); return MIRA_E_UNSUPPORTED; }
    // shared-bucket tables: every commitment of the batch gets ONE bucket set for its W windows
    // (ceil(256 / c) additions per pair, 2^(c-1) buckets per commitment instead of W 2^(c-1)),
    // and its partial sums come back to be added -- no chain of doublings
    const int32_t forced_c = bs.forced_c ? bs.forced_c : g.forced_c;
    const Bases::SharedSet *set = (forced_c == 0 && n >= tuned(MIRA_TUNE_SHARED_MIN_N, TABLE16_MIN_N)) ? pick_shared(bs, n, (uint32_t)std::min<size_t>(count, 64), false) : nullptr;
    if (set) {
        // which of the key's sets: the model's choice, checked against the others on the first batches of the shape (trial_report_sets)
        Bases::WidthTrial *strial = (bs.shared.size() > 1 && tuned(MIRA_TUNE_TABLE_WIDTH, 0) == 0 && count <= 64)
                                        ? trial_for(bs, n, (uint32_t)count, 4u | (h_batch ? 2u : 0u), set->c) : nullptr;
        if (strial) set = trial_set(bs, *strial, set);
        const auto t_set = std::chrono::steady_clock::now();
        const uint32_t Ws = (256 + set->c - 1) / set->c;
        const size_t per = std::max<size_t>(1, std::min<size_t>(64, (size_t)(((1ull << 32) - 1) / ((uint64_t)n * Ws))));
        std::vector<uint64_t> sums;
        for (size_t done = 0; done < count; done += per) {
            const size_t cnt = std::min(per, count - done);
            MsmPlan p = make_plan_shared(n, *set, bs.n, (uint32_t)cnt, stride);
            plan_reduction(p, default_pieces(p, MIRA_MAX_WINDOWS));
            p.h_batch = h_batch ? h_batch + done : nullptr;
            g.last_c = 0; g.last_w = (int32_t)p.pieces; g.last_table_c = (int32_t)set->c;
            sums.assign(cnt * p.pieces * 16, 0);
            const unsigned char *sc = reinterpret_cast<const unsigned char *>(d_scalars) + done * stride * 32;
            rc = bs.curve == MIRA_CURVE_BN256 ? msm_launch_bn256(bs, 0, sc, nullptr, n, p, sums.data()) : msm_launch_grumpkin(bs, 0, sc, nullptr, n, p, sums.data());
            if (rc) return rc;
            const PartialShape sh{0, 1, p.cb, p.pieces};
            for (size_t b = 0; b < cnt; b++) {
                const uint64_t *w = sums.data() + b * p.pieces * 16;
                if (bs.curve == MIRA_CURVE_BN256) horner_pieces<FqP>(w, sh, out_affine + (done + b) * 8);
                else horner_pieces<FrP>(w, sh, out_affine + (done + b) * 8);
            }
        }
        if (strial) trial_report_sets(*strial, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_set).count(), bs);
        return MIRA_OK;
    }
    // the GLV split (glv.cuh) where the key has its endomorphism copy: 2 n half-length scalars per commitment, half the windows
    bool glv = n < (1ull << 30) && glv_possible(bs);
    if (glv) {                                               // the planners' estimates for a batch of this shape decide (choose_glv)
        const uint32_t shape = (uint32_t)std::min<size_t>(count, 8);
        glv = choose_glv(bs, make_plan(n, forced_c, shape, stride), make_plan(2 * n, forced_c, shape, stride, nullptr, GLV_BITS), n * shape);
    }
    const size_t nv = glv ? 2 * n : n;
    const uint32_t bits = glv ? GLV_BITS : 256;
    MsmPlan p1 = make_plan(nv, forced_c, 1, 0, nullptr, bits);
    // per launch: W_total * B counters <= 2^21 (three-launch scan) and n * W_total entries < 2^32
    size_t per = std::max<size_t>(1, std::min<size_t>((size_t)((1ull << 21) / ((uint64_t)p1.W * p1.B)),
                                                      (size_t)(((1ull << 32) - 1) / ((uint64_t)nv * p1.W))));
    std::vector<uint64_t> win;
    for (size_t done = 0; done < count; done += per) {
        const size_t cnt = std::min(per, count - done);
        MsmPlan p = make_plan(nv, forced_c, (uint32_t)cnt, stride, nullptr, bits);
        Bases::WidthTrial *trial = forced_c == 0 ? trial_for(bs, n, (uint32_t)cnt, (glv ? 1u : 0u) | (h_batch ? 2u : 0u), p.c) : nullptr;
        if (trial && trial_width(*trial) != p.c && (uint64_t)((bits + trial_width(*trial) - 1) / trial_width(*trial)) * cnt * (1ull << (trial_width(*trial) - 1)) <= (1ull << 21))
            p = make_plan(nv, (int32_t)trial_width(*trial), (uint32_t)cnt, stride, nullptr, bits);      // (a width whose counters one scan takes)
        p.glv = glv; p.glv_bases = glv ? bs.glv : nullptr;
        p.h_batch = h_batch ? h_batch + done : nullptr;
        plan_reduction(p, default_pieces(p, 1u << 20));
        g.last_c = (int32_t)p.c; g.last_w = (int32_t)p.W; g.last_table_c = 0;
        win.assign((size_t)p.Wt * p.pieces * 16, 0);
        const unsigned char *sc = reinterpret_cast<const unsigned char *>(d_scalars) + done * stride * 32;
        const auto t_launch = std::chrono::steady_clock::now();
        rc = bs.curve == MIRA_CURVE_BN256 ? msm_launch_bn256(bs, 0, sc, nullptr, n, p, win.data()) : msm_launch_grumpkin(bs, 0, sc, nullptr, n, p, win.data());
        if (rc) return rc;
        if (trial && trial_width(*trial) == p.c)
            trial_report(*trial, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_launch).count(), glv ? 5 : 4, 16);
        else if (trial) trial->done = true;                  // the candidate does not fit one scan: the walk ends where it is
        // the epilogues of a batch are independent chains of ~250 doublings (60 us each): one host thread per commitment
        const PartialShape sh{p.c, p.W, p.cb, p.pieces};
        auto epilogue = [&](size_t b) {
            const uint64_t *w = win.data() + b * p.W * p.pieces * 16;
            if (bs.curve == MIRA_CURVE_BN256) horner_pieces<FqP>(w, sh, out_affine + (done + b) * 8);
            else horner_pieces<FrP>(w, sh, out_affine + (done + b) * 8);
        };
        host_parallel_for(cnt, epilogue);
    }
    return MIRA_OK;
}

static int ntt_kind_device_ctx(void *d_a, uint32_t log_n, NttKind kind, const uint64_t *omega_in) {
    int rc = ensure_ctx();
    if (rc) return rc;
    return ntt_kind_device(d_a, log_n, kind, omega_in);
}
static int ntt_kind_host_locked(uint64_t *a, uint32_t log_n, NttKind kind, const uint64_t *omega_in) {
    int rc = ensure_ctx();
    if (rc) return rc;
    if (!a) { set_error("null argument"); return MIRA_E_BAD_ARG; }
    if (log_n > 28) { set_error("k=" + std::to_string(log_n) + " should no larger than F::S=28"); return MIRA_E_BAD_ARG; }
    const size_t bytes = (size_t)32 << log_n;
    if ((rc = g.ntt_stage.ensure(bytes))) return rc;
    RT_CHECK(rt_h2d(g.ntt_stage.p, a, bytes, g.stream));
    if ((rc = ntt_kind_device(g.ntt_stage.p, log_n, kind, omega_in))) return rc;
    RT_CHECK(rt_d2h(a, g.ntt_stage.p, bytes, g.stream));
    RT_CHECK(rt_sync(g.stream));
    return MIRA_OK;
}

// ---- the instance side of a fold: a handful of single-scalar multiplications on the host ----------------------
// k * P for a canonical integer k < 2^255 by a width-5 NAF: 256 doublings and on average 43 additions of
// +- (1, 3, .. 15) P, against 128 additions bit by bit.
template <class FB> static hostf::HXyzz<FB> lift_affine(const uint64_t p[8]) {
    using namespace hostf;
    HXyzz<FB> r = identity<FB>();
    bool zero = true;
    for (int i = 0; i < 8; i++) zero &= (p[i] == 0);
    if (!zero) { memcpy(r.x.l, p, 32); memcpy(r.y.l, p + 4, 32); r.zz = one<FB>(); r.zzz = one<FB>(); }
    return r;
}
struct Wnaf5 {
    int8_t d[260];
    int len = 0;
    explicit Wnaf5(const uint64_t k_in[4]) {
        uint64_t k[5] = {k_in[0], k_in[1], k_in[2], k_in[3], 0};
        while (k[0] | k[1] | k[2] | k[3] | k[4]) {
            int digit = 0;
            if (k[0] & 1) {
                digit = (int)(k[0] & 31u);
                if (digit >= 16) {                               // k += 32 - digit: the low five bits become zero
                    digit -= 32;
                    uint64_t add = (uint64_t)(-digit);
                    for (int i = 0; i < 5 && add; i++) { const uint64_t t = k[i] + add; add = t < add; k[i] = t; }
                } else {
                    k[0] -= (uint64_t)digit;                     // clears the low bits, no borrow
                }
            }
            d[len++] = (int8_t)digit;
            for (int i = 0; i < 4; i++) k[i] = (k[i] >> 1) | (k[i + 1] << 63);
            k[4] >>= 1;
        }
    }
};
template <class FB> struct OddMultiples {                        // (2 i + 1) P, i < 8
    hostf::HXyzz<FB> t[8];
    explicit OddMultiples(const hostf::HXyzz<FB> &P) {
        const hostf::HXyzz<FB> P2 = hostf::dbl_pt(P);
        t[0] = P;
        for (int i = 1; i < 8; i++) t[i] = hostf::add_pt(t[i - 1], P2);
    }
    hostf::HXyzz<FB> signed_multiple(int digit) const {          // digit odd, |digit| <= 15
        hostf::HXyzz<FB> r = t[(digit < 0 ? -digit : digit) >> 1];
        if (digit < 0) r.y = hostf::sub(hostf::zero<FB>(), r.y);
        return r;
    }
};
// sum_i k_i P_i, one shared chain of doublings (Straus over the NAFs); scalars in Montgomery form, points affine
template <class FB, class FS> static hostf::HXyzz<FB> g1_straus(const uint64_t *scalars, const uint64_t *points, size_t count) {
    using namespace hostf;
    std::vector<Wnaf5> naf;
    std::vector<OddMultiples<FB>> table;
    naf.reserve(count); table.reserve(count);
    const HFe<FS> one_plain = {{1, 0, 0, 0}};
    int top = 0;
    for (size_t i = 0; i < count; i++) {
        HFe<FS> s;
        memcpy(s.l, scalars + 4 * i, 32);
        s = mul(s, one_plain);                                   // leave Montgomery form: canonical integer
        naf.emplace_back(s.l);
        table.emplace_back(lift_affine<FB>(points + 8 * i));
        top = std::max(top, naf.back().len);
    }
    HXyzz<FB> R = identity<FB>();
    for (int b = top - 1; b >= 0; b--) {
        R = dbl_pt(R);
        for (size_t i = 0; i < count; i++)
            if (b < naf[i].len && naf[i].d[b]) R = add_pt(R, table[i].signed_multiple(naf[i].d[b]));
    }
    return R;
}
// acc + scalar * point on affine points: the single-scalar best_multiexp calls of
// RelaxedPlonkInstance::fold (src/plonk/mod.rs:986-999, 1049-1053).
template <class FB, class FS> static void g1_mul_add_t(const uint64_t acc[8], const uint64_t scalar[4], const uint64_t point[8], uint64_t out[8]) {
    hostf::to_affine(hostf::add_pt(g1_straus<FB, FS>(scalar, point, 1), lift_affine<FB>(acc)), out);
}
// acc + sum_i scalars[i] * points[i] on affine points: the instance side of a fold, E_commit + sum_k r^(k+1) T_k over the
// d - 1 cross-term commitments (src/plonk/mod.rs:1049-1053).  The terms are dealt to the resident host threads, every
// thread walks ONE chain of doublings for its terms.
template <class FB, class FS> static hostf::HXyzz<FB> g1_lincomb_xyzz(const uint64_t *scalars, const uint64_t *points, size_t count) {
    using namespace hostf;
    const size_t groups = std::min<size_t>(count, host_parallel_width());
    if (groups <= 1) return g1_straus<FB, FS>(scalars, points, count);
    std::vector<HXyzz<FB>> part(groups);
    host_parallel_for(groups, [&](size_t gi) {
        const size_t lo = count * gi / groups, hi = count * (gi + 1) / groups;
        part[gi] = g1_straus<FB, FS>(scalars + 4 * lo, points + 8 * lo, hi - lo);
    });
    HXyzz<FB> R = part[0];
    for (size_t gi = 1; gi < groups; gi++) R = add_pt(R, part[gi]);
    return R;
}
template <class FB, class FS>
static void g1_lincomb_t(const uint64_t acc[8], const uint64_t *scalars, const uint64_t *points, size_t count, uint64_t out[8]) {
    hostf::to_affine(hostf::add_pt(g1_lincomb_xyzz<FB, FS>(scalars, points, count), lift_affine<FB>(acc)), out);
}
// RelaxedPlonkInstance::fold, the commitments (src/plonk/mod.rs:986-999: W1_i + r W2_i; :1049-1053: E + sum_k r^(k+1) T_k),
// all of it one parallel region: every W commitment and every group of cross-term commitments is one task.
template <class FB, class FS>
static void g1_fold_commitments_t(const uint64_t r[4], const uint64_t *w1, const uint64_t *w2, size_t nw, const uint64_t e[8], const uint64_t *t_commits,
                                  size_t count, uint64_t *w_out, uint64_t e_out[8]) {
    using namespace hostf;
    std::vector<uint64_t> powers(4 * count);
    HFe<FS> rr, p;
    memcpy(rr.l, r, 32);
    p = rr;
    for (size_t k = 0; k < count; k++) { memcpy(&powers[4 * k], p.l, 32); p = mul(p, rr); }    // r^1, r^2, ... (iter::successors)
    const size_t width = host_parallel_width();
    const size_t egroups = count ? std::min<size_t>(count, width > nw ? width - nw : 1) : 0;
    std::vector<HXyzz<FB>> part(egroups);
    host_parallel_for(nw + egroups, [&](size_t i) {
        if (i < nw) { g1_mul_add_t<FB, FS>(w1 + 8 * i, r, w2 + 8 * i, w_out + 8 * i); return; }
        const size_t gi = i - nw, lo = count * gi / egroups, hi = count * (gi + 1) / egroups;
        part[gi] = g1_straus<FB, FS>(&powers[4 * lo], t_commits + 8 * lo, hi - lo);
    });
    HXyzz<FB> R = lift_affine<FB>(e);
    for (size_t gi = 0; gi < egroups; gi++) R = add_pt(R, part[gi]);
    to_affine(R, e_out);
}

// ------------------------------------------------------------------------------------------
// C ABI
extern "C" {

int mira_device_count(void) {
#ifndef MIRA_CPU_EMU
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
#else
    return 1;
#endif
}
int mira_init(int device) {
    std::lock_guard<std::mutex> lk(g_lock);
    if (g.ready && device != g.device) { set_error("library already bound to device " + std::to_string(g.device)); return MIRA_E_BAD_ARG; }
    g.device = device;
    return ensure_ctx();
}
int mira_set_stream(void *hip_stream) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
#ifndef MIRA_CPU_EMU
    if (hip_stream) {
        if (g.own_stream && g.stream) { (void)hipStreamSynchronize(g.stream); (void)hipStreamDestroy(g.stream); }
        g.stream = reinterpret_cast<hipStream_t>(hip_stream); g.own_stream = false;
    } else if (!g.own_stream) {
        RT_CHECK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking)); g.own_stream = true;
    }
#else
    (void)hip_stream;
#endif
    return MIRA_OK;
}
const char *mira_last_error(void) { return g_err.c_str(); }

// Both register calls leave a library-owned resident copy of the key in the engine's layout
// (canonical x * 2^261, y * 2^261; k_convert_bases), so the caller's buffer -- host or device --
// is free again when the call returns.
static int register_common(int curve, const void *src, bool src_on_device, size_t n, uint64_t *handle_out) {
    int rc = ensure_ctx();
    if (rc) return rc;
    if ((curve != MIRA_CURVE_BN256 && curve != MIRA_CURVE_GRUMPKIN) || !handle_out || (n && !src)) { set_error("bad register arguments"); return MIRA_E_BAD_ARG; }
    Bases b; b.curve = curve; b.n = n; b.owned = true;
    if (rt_malloc(&b.d, std::max<size_t>(64, n * 64)) != hipSuccess || !b.d) { set_error("device allocation for bases failed"); return MIRA_E_ALLOC; }
    if (n) {
        const void *conv_src = src;
        if (!src_on_device) {
            RT_CHECK(rt_h2d(b.d, src, n * 64, g.stream));
            conv_src = b.d;   // convert in place
        }
        rc = curve == MIRA_CURVE_BN256 ? convert_bases_bn256(conv_src, b.d, n) : convert_bases_grumpkin(conv_src, b.d, n);
        if (rc) { (void)rt_free(b.d); return rc; }
    }
    *handle_out = g.next_handle++;
    g_bases[*handle_out] = b;
    return MIRA_OK;
}
int mira_msm_register_bases(int curve, const uint64_t *bases, size_t n, uint64_t *handle_out) {
    std::lock_guard<std::mutex> lk(g_lock);
    return register_common(curve, bases, false, n, handle_out);
}
int mira_msm_register_bases_device(int curve, const void *d_bases, size_t n, uint64_t *handle_out) {
    std::lock_guard<std::mutex> lk(g_lock);
    return register_common(curve, d_bases, true, n, handle_out);
}
int mira_msm_unregister(uint64_t handle) {
    std::lock_guard<std::mutex> lk(g_lock);
    auto it = g_bases.find(handle);
    if (it == g_bases.end()) { set_error("unknown bases handle"); return MIRA_E_BAD_ARG; }
    if (it->second.owned && it->second.d) (void)rt_free(it->second.d);
    if (it->second.tables) (void)rt_free(it->second.tables);
    for (auto &set : it->second.shared) (void)rt_free(set.p);
    if (it->second.glv) (void)rt_free(it->second.glv);
    g_bases.erase(it);
    return MIRA_OK;
}
static int precompute_locked(uint64_t handle, int32_t window_bits) {
    int rc = ensure_ctx();
    if (rc) return rc;
    auto it = g_bases.find(handle);
    if (it == g_bases.end()) { set_error("unknown bases handle"); return MIRA_E_BAD_ARG; }
    Bases &bs = it->second;
    if (window_bits >= 8 && window_bits <= 16) {             // a shared-bucket set: any number of widths beside each other
        for (const auto &set : bs.shared) if (set.c == (uint32_t)window_bits) return MIRA_OK;
        const uint32_t W = (256 + (uint32_t)window_bits - 1) / (uint32_t)window_bits;
        if ((uint64_t)bs.n * W >= (1ull << 31)) { set_error("key too long for 31-bit table indices"); return MIRA_E_UNSUPPORTED; }
        Bases tmp = bs;                                      // build_tables fills tables / table_c / table_w of what it is given
        tmp.tables = nullptr; tmp.table_c = tmp.table_w = 0;
        rc = bs.curve == MIRA_CURVE_BN256 ? build_tables_bn256(tmp, (uint32_t)window_bits, W) : build_tables_grumpkin(tmp, (uint32_t)window_bits, W);
        if (rc) return rc;
        if (tmp.tables) { bs.shared.push_back({tmp.tables, (uint32_t)window_bits, W}); bs.trials.clear(); }   // the trials among the sets start again
        return MIRA_OK;
    }
    if (window_bits == MIRA_TABLE_GLV) {                       // the interleaved key [P_i, phi(P_i)] of the GLV split
        if (bs.glv || bs.n == 0) return MIRA_OK;
        const unsigned char *consts = reinterpret_cast<const unsigned char *>(g.consts.p);
        return bs.curve == MIRA_CURVE_BN256 ? build_glv_bn256(bs, consts + 192) : build_glv_grumpkin(bs, consts + 224);
    }
    if (window_bits != 20 && window_bits != 22) { set_error("window tables are built for 8- to 16-bit (shared buckets), 20- or 22-bit windows"); return MIRA_E_BAD_ARG; }
    if (bs.tables && bs.table_c != (uint32_t)window_bits) { set_error("this key already has wide tables of another width"); return MIRA_E_BAD_ARG; }
    const uint32_t W = window_bits == 22 ? 12 : 13;          // ceil(256 / c); 12 x 22 = 264 covers a signed 254-bit scalar
    if ((uint64_t)bs.n * W >= (1ull << 31)) { set_error("key too long for 31-bit table indices"); return MIRA_E_UNSUPPORTED; }
    return bs.curve == MIRA_CURVE_BN256 ? build_tables_bn256(bs, (uint32_t)window_bits, W) : build_tables_grumpkin(bs, (uint32_t)window_bits, W);
}
int mira_msm_precompute(uint64_t handle) {
    std::lock_guard<std::mutex> lk(g_lock);
    return precompute_locked(handle, 20);
}
int mira_msm_precompute_ex(uint64_t handle, int32_t window_bits) {
    std::lock_guard<std::mutex> lk(g_lock);
    return precompute_locked(handle, window_bits);
}
int mira_msm_check_bases(uint64_t handle) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    auto it = g_bases.find(handle);
    if (it == g_bases.end()) { set_error("unknown bases handle"); return MIRA_E_BAD_ARG; }
    const Bases &bs = it->second;
    if (bs.n == 0) return MIRA_OK;
    if ((rc = g.heavy.ensure(64))) return rc;
    uint32_t *bad = reinterpret_cast<uint32_t *>(g.heavy.p);
    RT_CHECK(rt_memset(bad, 0, 4, g.stream));
    if ((rc = bs.curve == MIRA_CURVE_BN256 ? check_bases_bn256(bs, bad) : check_bases_grumpkin(bs, bad))) return rc;
    uint32_t h = 0;
    RT_CHECK(rt_d2h(&h, bad, 4, g.stream));
    RT_CHECK(rt_sync(g.stream));
    if (h) { set_error("Wrong key, " + std::to_string(h) + " points out of curve"); return MIRA_E_INVALID_POINT; }
    return MIRA_OK;
}

static int msm_device_locked(uint64_t handle, const void *d_scalars, size_t n, uint64_t out_affine[8], const void *h_scalars = nullptr) {
    if (!out_affine) { set_error("null output"); return MIRA_E_BAD_ARG; }
    uint64_t part[MIRA_PARTIAL_U64];
    PartialShape sh;
    int rc = msm_partial_locked(handle, 0, d_scalars, n, part, &sh, false, 0, h_scalars, true);
    if (rc) return rc;
    if (g_bases[handle].curve == MIRA_CURVE_BN256) horner_pieces<FqP>(part, sh, out_affine);
    else horner_pieces<FrP>(part, sh, out_affine);
    return MIRA_OK;
}
int mira_msm_device(uint64_t handle, const void *d_scalars, size_t n, uint64_t out_affine[8]) {
    std::lock_guard<std::mutex> lk(g_lock);
    return msm_device_locked(handle, d_scalars, n, out_affine);
}
int mira_msm(uint64_t handle, const uint64_t *scalars, size_t n, uint64_t out_affine[8]) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    if (n && !scalars) { set_error("null scalars"); return MIRA_E_BAD_ARG; }
    auto it = g_bases.find(handle);
    if (it == g_bases.end()) { set_error("unknown bases handle"); return MIRA_E_BAD_ARG; }
    if (n > it->second.n) {   // length check before any copy, as commit does (src/commitment.rs:79)
        set_error("Can't commit too long input: input len: " + std::to_string(n) + ", but limit is " + std::to_string(it->second.n));
        return MIRA_E_TOO_LONG;
    }
    // the scalars cross PCIe inside the launch sequence, chunk by chunk beside the kernels (msm_host.cuh)
    if (n && (rc = g.scalars_stage.ensure(n * 32))) return rc;
    return msm_device_locked(handle, g.scalars_stage.p, n, out_affine, n ? scalars : nullptr);
}
int mira_msm_batch_device(uint64_t handle, const void *d_scalars, size_t n, size_t count, size_t stride_elems, uint64_t *out_affine) {
    std::lock_guard<std::mutex> lk(g_lock);
    return msm_batch_device_locked(handle, d_scalars, n, count, stride_elems, out_affine);
}
int mira_msm_batch(uint64_t handle, const uint64_t *const *scalars, size_t n, size_t count, uint64_t *out_affine) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    if (count && n && !scalars) { set_error("null scalars"); return MIRA_E_BAD_ARG; }
    auto it = g_bases.find(handle);
    if (it == g_bases.end()) { set_error("unknown bases handle"); return MIRA_E_BAD_ARG; }
    if (n > it->second.n) {
        set_error("Can't commit too long input: input len: " + std::to_string(n) + ", but limit is " + std::to_string(it->second.n));
        return MIRA_E_TOO_LONG;
    }
    if (n && count) {
        if ((rc = g.scalars_stage.ensure(n * count * 32))) return rc;
        for (size_t b = 0; b < count; b++)
            if (!scalars[b]) { set_error("null scalars"); return MIRA_E_BAD_ARG; }
    }
    // the vectors cross PCIe inside the launch sequence, in point chunks beside the kernels (msm_host.cuh)
    return msm_batch_device_locked(handle, g.scalars_stage.p, n, count, n, out_affine, (n && count) ? scalars : nullptr);
}
int mira_msm_partial_device(uint64_t handle, size_t first, const void *d_scalars, size_t n, uint64_t out_partial[MIRA_PARTIAL_U64],
                            int32_t *window_bits, int32_t *num_windows) {
    std::lock_guard<std::mutex> lk(g_lock);
    if (!out_partial || !window_bits || !num_windows) { set_error("null output"); return MIRA_E_BAD_ARG; }
    if (*window_bits != 0 && (*window_bits < 4 || *window_bits > 16)) { set_error("window_bits must be 0 or 4..16"); return MIRA_E_BAD_ARG; }
    PartialShape sh;
    int rc = msm_partial_locked(handle, first, d_scalars, n, out_partial, &sh, true, *window_bits);
    if (rc) return rc;
    *window_bits = (int32_t)sh.c; *num_windows = (int32_t)sh.W;
    return MIRA_OK;
}
int mira_msm_combine(int curve, const uint64_t *partials, size_t nparts, int32_t window_bits, int32_t num_windows, uint64_t out_affine[8]) {
    return combine_locked(curve, partials, nparts, (uint32_t)window_bits, (uint32_t)num_windows, out_affine);
}
int mira_set_tuning(int knob, int64_t value) {
    std::lock_guard<std::mutex> lk(g_lock);
    if (knob < 0 || knob > MIRA_TUNE_NTT_GRID || (knob == MIRA_TUNE_PASS_ENTRIES_LOG && value > 32)) { set_error("unknown tuning knob"); return MIRA_E_BAD_ARG; }
    g.tune[knob] = value;
    return MIRA_OK;
}
int mira_msm_plan_window_bits(size_t n, int32_t *window_bits) {
    std::lock_guard<std::mutex> lk(g_lock);                  // make_plan reads the tuning knobs mira_set_tuning writes under this lock
    if (!window_bits) { set_error("null output"); return MIRA_E_BAD_ARG; }
    *window_bits = (int32_t)make_plan(std::max<size_t>(n, 1), 0).c;
    return MIRA_OK;
}
int mira_msm_last_plan(int32_t *window_bits, int32_t *num_windows) {
    std::lock_guard<std::mutex> lk(g_lock);
    if (!window_bits || !num_windows) { set_error("null output"); return MIRA_E_BAD_ARG; }
    *window_bits = g.last_c; *num_windows = g.last_w;
    return MIRA_OK;
}
int mira_msm_last_table_bits(int32_t *table_bits) {
    std::lock_guard<std::mutex> lk(g_lock);
    if (!table_bits) { set_error("null output"); return MIRA_E_BAD_ARG; }
    *table_bits = g.last_table_c;
    return MIRA_OK;
}
int mira_msm_set_window_bits(int32_t c) {
    std::lock_guard<std::mutex> lk(g_lock);
    if (c != 0 && (c < 4 || c > 16)) { set_error("window bits must be 0 or in [4,16]"); return MIRA_E_BAD_ARG; }
    g.forced_c = c;
    return MIRA_OK;
}

int mira_msm_set_handle_window_bits(uint64_t handle, int32_t c) {
    std::lock_guard<std::mutex> lk(g_lock);
    if (c != 0 && (c < 4 || c > 16)) { set_error("window bits must be 0 or in [4,16]"); return MIRA_E_BAD_ARG; }
    auto it = g_bases.find(handle);
    if (it == g_bases.end()) { set_error("unknown bases handle"); return MIRA_E_BAD_ARG; }
    it->second.forced_c = c;
    return MIRA_OK;
}
int mira_msm_partial_to_device(uint64_t handle, size_t first, const void *d_scalars, size_t n, void *d_out_partial,
                               int32_t *window_bits, int32_t *num_windows) {
    std::lock_guard<std::mutex> lk(g_lock);
    if (!d_out_partial || !window_bits || !num_windows) { set_error("null output"); return MIRA_E_BAD_ARG; }
    if (*window_bits != 0 && (*window_bits < 4 || *window_bits > 16)) { set_error("window_bits must be 0 or 4..16"); return MIRA_E_BAD_ARG; }
    int rc = ensure_ctx();
    if (rc) return rc;
    RT_CHECK(rt_memset(d_out_partial, 0, MIRA_PARTIAL_U64 * 8, g.stream));     // words beyond the partial's windows (and an empty chunk) read as the identity
    RT_CHECK(rt_sync(g.stream));
    uint64_t unused[MIRA_PARTIAL_U64];
    PartialShape sh;
    g.windows_dst = d_out_partial;
    rc = msm_partial_locked(handle, first, d_scalars, n, unused, &sh, true, *window_bits);
    g.windows_dst = nullptr;
    if (rc) return rc;
    *window_bits = (int32_t)sh.c; *num_windows = (int32_t)sh.W;
    return MIRA_OK;
}

// CommitmentKey::load_from_file + the is_on_curve pass of load_or_setup_cache (src/commitment.rs:110-127, 145-154)
int mira_msm_register_bases_file(int curve, const char *path, uint32_t k, int validate, uint64_t *handle_out) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    if ((curve != MIRA_CURVE_BN256 && curve != MIRA_CURVE_GRUMPKIN) || !path || !handle_out || k > 31) { set_error("bad register arguments"); return MIRA_E_BAD_ARG; }
    const size_t n = (size_t)1 << k;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { set_error(std::string(path) + ": " + strerror(errno)); return MIRA_E_IO; }
    struct stat sb;
    if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < n * 64) {                  // read_exact of vec_len * size_of::<C>() bytes
        set_error("failed to fill whole buffer");
        close(fd);
        return MIRA_E_IO;
    }
    Bases b; b.curve = curve; b.n = n; b.owned = true;
    if (rt_malloc(&b.d, n * 64) != hipSuccess || !b.d) { close(fd); set_error("device allocation for bases failed"); return MIRA_E_ALLOC; }
    uint32_t *bad = nullptr;
    if ((rc = g.heavy.ensure(64)) == MIRA_OK) {
        bad = reinterpret_cast<uint32_t *>(g.heavy.p);
        rc = curve == MIRA_CURVE_BN256 ? load_bases_file_bn256(b, fd, validate != 0, bad) : load_bases_file_grumpkin(b, fd, validate != 0, bad);
    }
    close(fd);
    if (rc == MIRA_OK && validate) {
        uint32_t h = 0;
        if (rt_d2h(&h, bad, 4, g.stream) != hipSuccess || rt_sync(g.stream) != hipSuccess) { set_error("device to host copy failed"); rc = MIRA_E_NO_DEVICE; }
        else if (h) { set_error("Wrong file in cache, some ptr out of curve"); rc = MIRA_E_INVALID_POINT; }
    }
    if (rc != MIRA_OK) { (void)rt_free(b.d); return rc; }
    *handle_out = g.next_handle++;
    g_bases[*handle_out] = b;
    return MIRA_OK;
}
// save_to_file (src/commitment.rs:96-101): the key as the raw slice of reference-layout points
int mira_msm_save_bases_file(uint64_t handle, const char *path) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    auto it = g_bases.find(handle);
    if (it == g_bases.end() || !path) { set_error("unknown bases handle"); return MIRA_E_BAD_ARG; }
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { set_error(std::string(path) + ": " + strerror(errno)); return MIRA_E_IO; }
    rc = it->second.curve == MIRA_CURVE_BN256 ? save_bases_file_bn256(it->second, fd) : save_bases_file_grumpkin(it->second, fd);
    if (close(fd) != 0 && rc == MIRA_OK) { set_error(std::string("close failed: ") + strerror(errno)); rc = MIRA_E_IO; }
    return rc;
}

// Release grow-only workspaces, largest first, until at most keep_bytes remain.
int mira_trim(size_t keep_bytes, size_t *released_out) {
    std::lock_guard<std::mutex> lk(g_lock);
    if (released_out) *released_out = 0;
    if (!g.ready) return MIRA_OK;
    RT_CHECK(rt_sync(g.stream));
    if (g.copy_stream) RT_CHECK(rt_sync(g.copy_stream));
    // `consts`, `ntt_consts` and `fold_consts` hold uploaded constants (a few hundred bytes): never released
    std::vector<DevBuf *> bufs = {&g.digits, &g.counts, &g.offsets, &g.cursor, &g.block_sums, &g.sorted_idx, &g.bucket_sums, &g.part, &g.coarse_offsets,
                                  &g.fine_counts, &g.fine_cursor, &g.head_part, &g.tail_part, &g.tail_key, &g.heavy, &g.heavy_out, &g.chunks, &g.window_sums,
                                  &g.scalars_stage, &g.ntt_tmp, &g.ntt_stage, &g.graph_ws, &g.tree_a, &g.tree_b, &g.hist_dev};
    for (int i = 0; i < Ctx::NTT_SETS; i++) bufs.push_back(&g.ntt_set[i]);
    size_t total = 0;
    for (DevBuf *b : bufs) total += b->cap;
    std::sort(bufs.begin(), bufs.end(), [](const DevBuf *a, const DevBuf *b) { return a->cap > b->cap; });
    size_t released = 0;
    for (DevBuf *b : bufs) {
        if (total - released <= keep_bytes || b->cap == 0) break;
        for (int i = 0; i < Ctx::NTT_SETS; i++)
            if (b == &g.ntt_set[i]) { g.ntt_set_key[i].clear(); g.ntt_set_stamp[i] = 0; }
        if (b == &g.hist_dev) g.hist_sel = 0;
        released += b->cap;
        b->release();
    }
    if (released_out) *released_out = released;
    return MIRA_OK;
}
int mira_dev_mem_info(size_t *free_bytes, size_t *total_bytes) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    size_t f = 0, t = 0;
#ifndef MIRA_CPU_EMU
    RT_CHECK(hipMemGetInfo(&f, &t));
#endif
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return MIRA_OK;
}

int mira_ntt_bn256_fr(uint64_t *a, uint32_t log_n, const uint64_t omega[4]) {
    std::lock_guard<std::mutex> lk(g_lock);
    if (!omega) { set_error("null omega"); return MIRA_E_BAD_ARG; }
    return ntt_kind_host_locked(a, log_n, NTT_BEST, omega);
}
int mira_ntt_bn256_fr_device(void *d_a, uint32_t log_n, const uint64_t omega[4]) {
    std::lock_guard<std::mutex> lk(g_lock);
    if (!omega) { set_error("null omega"); return MIRA_E_BAD_ARG; }
    return ntt_kind_device_ctx(d_a, log_n, NTT_BEST, omega);
}
int mira_fft_bn256_fr(uint64_t *a, uint32_t log_n) { std::lock_guard<std::mutex> lk(g_lock); return ntt_kind_host_locked(a, log_n, NTT_FFT, nullptr); }
int mira_ifft_bn256_fr(uint64_t *a, uint32_t log_n) { std::lock_guard<std::mutex> lk(g_lock); return ntt_kind_host_locked(a, log_n, NTT_IFFT, nullptr); }
int mira_fft_bn256_fr_device(void *d_a, uint32_t log_n) { std::lock_guard<std::mutex> lk(g_lock); return ntt_kind_device_ctx(d_a, log_n, NTT_FFT, nullptr); }
int mira_ifft_bn256_fr_device(void *d_a, uint32_t log_n) { std::lock_guard<std::mutex> lk(g_lock); return ntt_kind_device_ctx(d_a, log_n, NTT_IFFT, nullptr); }
int mira_coset_fft_bn256_fr(uint64_t *a, uint32_t log_n) { std::lock_guard<std::mutex> lk(g_lock); return ntt_kind_host_locked(a, log_n, NTT_COSET_FFT, nullptr); }
int mira_coset_ifft_bn256_fr(uint64_t *a, uint32_t log_n) { std::lock_guard<std::mutex> lk(g_lock); return ntt_kind_host_locked(a, log_n, NTT_COSET_IFFT, nullptr); }
int mira_get_omega_or_inv(uint32_t k, int is_inverse, uint64_t out[4]) {
    if (!out || k > 28) { set_error("k=" + std::to_string(k) + " should no larger than F::S=28"); return MIRA_E_BAD_ARG; }
    return ntt_get_omega_or_inv(k, is_inverse != 0, out);
}

int mira_fold_witness_device(int field, void *d_out, const void *d_w1, const void *d_w2, const uint64_t r[4], size_t n) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    if ((field != 0 && field != 1) || !r || (n && (!d_out || !d_w1 || !d_w2))) { set_error("bad fold arguments"); return MIRA_E_BAD_ARG; }
    if (!n) return MIRA_OK;
    return fold_witness_device(field, d_out, d_w1, d_w2, r, n);
}
int mira_fold_error_device(int field, void *d_e, const void *const *d_cross_terms, size_t num_terms, const uint64_t r[4], size_t n) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    if ((field != 0 && field != 1) || !r || num_terms > 16 || (num_terms && !d_cross_terms) || (n && !d_e)) { set_error("bad fold arguments"); return MIRA_E_BAD_ARG; }
    for (size_t k = 0; k < num_terms; k++)
        if (n && !d_cross_terms[k]) { set_error("null cross term"); return MIRA_E_BAD_ARG; }
    if (!n || !num_terms) return MIRA_OK;
    return fold_error_device(field, d_e, d_cross_terms, num_terms, r, n);
}
int mira_fold_relaxed_witness_device(int field, void *d_w_out, const void *d_w1, const void *d_w2, size_t n_w, void *d_e_out, const void *d_e,
                                     const void *const *d_cross_terms, size_t num_terms, const uint64_t r[4], size_t n) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    if ((field != 0 && field != 1) || !r || num_terms > 16 || (num_terms && !d_cross_terms) || (n_w && (!d_w_out || !d_w1 || !d_w2)) || (n && (!d_e_out || !d_e))) {
        set_error("bad fold arguments");
        return MIRA_E_BAD_ARG;
    }
    for (size_t k = 0; k < num_terms; k++)
        if (n && !d_cross_terms[k]) { set_error("null cross term"); return MIRA_E_BAD_ARG; }
    if (!n_w && !n) return MIRA_OK;
    return fold_relaxed_device(field, d_w_out, d_w1, d_w2, n_w, d_e_out, d_e, d_cross_terms, num_terms, r, n);
}
int mira_lincomb_device(int field, void *d_out, const void *const *d_vecs, const uint64_t *coeffs, size_t num_vecs, size_t n) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    if ((field != MIRA_FIELD_FQ && field != MIRA_FIELD_FR) || num_vecs == 0 || num_vecs > 16 || !d_vecs || !coeffs || (n && !d_out)) { set_error("bad linear-combination arguments"); return MIRA_E_BAD_ARG; }
    for (size_t k = 0; k < num_vecs; k++)
        if (n && !d_vecs[k]) { set_error("null vector"); return MIRA_E_BAD_ARG; }
    if (!n) return MIRA_OK;
    return lincomb_device(field, d_out, d_vecs, coeffs, num_vecs, n);
}
int mira_lincomb_multi_device(int field, void *const *d_outs, size_t num_outs, const void *const *d_vecs, size_t num_vecs, const uint64_t *coeffs, size_t n) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    if ((field != MIRA_FIELD_FQ && field != MIRA_FIELD_FR) || num_outs == 0 || num_outs > 8 || num_vecs == 0 || num_vecs > 16 || !d_outs || !d_vecs || !coeffs) {
        set_error("bad linear-combination arguments");
        return MIRA_E_BAD_ARG;
    }
    for (size_t k = 0; k < num_vecs; k++)
        if (n && !d_vecs[k]) { set_error("null vector"); return MIRA_E_BAD_ARG; }
    for (size_t m = 0; m < num_outs; m++) {
        if (n && !d_outs[m]) { set_error("null output"); return MIRA_E_BAD_ARG; }
        for (size_t q = 0; q < m; q++)
            if (n && d_outs[m] == d_outs[q]) { set_error("two outputs share one buffer"); return MIRA_E_BAD_ARG; }
        for (size_t k = 0; k < num_vecs; k++)
            if (n && d_outs[m] == d_vecs[k]) { set_error("an output aliases an input vector"); return MIRA_E_BAD_ARG; }
    }
    if (!n) return MIRA_OK;
    return lincomb_multi_device(field, d_outs, num_outs, d_vecs, num_vecs, coeffs, n);
}
int mira_pow_tree_reduce_device(int field, const void *d_leaves, size_t n_leaves, size_t leaf_point_stride, const uint64_t *weights, uint32_t num_points, uint64_t *out) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    if ((field != MIRA_FIELD_FQ && field != MIRA_FIELD_FR) || !d_leaves || !out || num_points == 0 || num_points > 65535 || n_leaves == 0) { set_error("bad tree-reduction arguments"); return MIRA_E_BAD_ARG; }
    if (n_leaves & (n_leaves - 1)) {
        // itertools::tree_reduce would pair nodes of different heights: `unreachable!` in the reference
        set_error("tree reduction needs a power-of-two number of leaves, got " + std::to_string(n_leaves));
        return MIRA_E_UNSUPPORTED;
    }
    uint32_t levels = 0;
    while (((size_t)1 << levels) < n_leaves) levels++;
    if (levels && !weights) { set_error("null weights"); return MIRA_E_BAD_ARG; }
    if (leaf_point_stride != 0 && leaf_point_stride < n_leaves) { set_error("leaf_point_stride shorter than the leaves"); return MIRA_E_BAD_ARG; }
    return pow_tree_reduce_device(field, d_leaves, levels, leaf_point_stride, weights, num_points, out);
}
int mira_graph_eval_device(int field, const mira_graph *graph, const mira_eval_column *columns, uint32_t num_columns, const uint64_t *challenges,
                           uint32_t num_challenges, size_t num_rows, void *d_out) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    if ((field != MIRA_FIELD_FQ && field != MIRA_FIELD_FR) || !graph || (graph->code_words && !graph->code) || (graph->num_constants && !graph->constants) ||
        (graph->num_rotations && !graph->rotations) || (num_columns && !columns) || (num_challenges && !challenges) || (num_rows && !d_out) ||
        num_columns > 0xFFFFFu || graph->num_rotations > 512u) {
        set_error("bad graph evaluation arguments");
        return MIRA_E_BAD_ARG;
    }
    return graph_eval_device(field, graph, columns, num_columns, challenges, num_challenges, num_rows, d_out);
}
int mira_graph_compile(int field, const mira_graph *graph, uint32_t num_challenges, uint32_t num_columns, uint64_t *handle_out) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    if ((field != MIRA_FIELD_FQ && field != MIRA_FIELD_FR) || !graph || !handle_out || (graph->code_words && !graph->code) || (graph->num_constants && !graph->constants) ||
        (graph->num_rotations && !graph->rotations) || num_columns > 0xFFFFFu || graph->num_rotations > 512u) {
        set_error("bad graph compilation arguments");
        return MIRA_E_BAD_ARG;
    }
    return graph_compile(field, graph, num_challenges, num_columns, handle_out);
}
int mira_graph_eval_compiled(uint64_t handle, const mira_eval_column *columns, uint32_t num_columns, const uint64_t *challenges, uint32_t num_challenges,
                             size_t num_rows, void *d_out) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    if ((num_columns && !columns) || (num_challenges && !challenges) || (num_rows && !d_out)) { set_error("bad graph evaluation arguments"); return MIRA_E_BAD_ARG; }
    return graph_eval_compiled(handle, columns, num_columns, challenges, num_challenges, num_rows, d_out);
}
int mira_graph_eval_batch(const uint64_t *handles, uint32_t count, const mira_eval_column *columns, uint32_t num_columns, const uint64_t *challenges,
                          uint32_t num_challenges, size_t num_rows, void *const *d_outs) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    if ((count && (!handles || !d_outs)) || (num_columns && !columns) || (num_challenges && !challenges)) { set_error("bad graph evaluation arguments"); return MIRA_E_BAD_ARG; }
    return graph_eval_batch(handles, count, columns, num_columns, challenges, num_challenges, num_rows, d_outs);
}
int mira_graph_free(uint64_t handle) {
    std::lock_guard<std::mutex> lk(g_lock);
    return graph_free(handle);
}
int mira_graph_specialize(const uint64_t *handles, uint32_t count, const mira_eval_column *columns, uint32_t num_columns) {
    std::unique_lock<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    if ((count && !handles) || (num_columns && !columns)) { set_error("null argument"); return MIRA_E_BAD_ARG; }
    return graph_specialize(handles, count, columns, num_columns, &lk);
}
int mira_graph_is_specialized(uint64_t handle, int32_t *out) {
    std::lock_guard<std::mutex> lk(g_lock);
    return graph_is_specialized(handle, out);
}
int mira_graph_set_cache_dir(const char *dir) {
    std::lock_guard<std::mutex> lk(g_lock);
    return graph_set_cache_dir(dir);
}
int mira_graph_jit_compile_check(const char *source, size_t *code_size_out) {
    return graph_jit_compile_check(source, code_size_out);       // touches neither the device nor the library's state (rtc() is loaded once)
}
int mira_graph_jit_stats(uint32_t *compiled_out, uint32_t *from_disk_out) {
    std::lock_guard<std::mutex> lk(g_lock);
    return graph_jit_stats(compiled_out, from_disk_out);
}
int mira_graph_jit_source(uint64_t handle, const mira_eval_column *columns, uint32_t num_columns, char *buf, size_t cap, size_t *len_out) {
    std::lock_guard<std::mutex> lk(g_lock);
    if (num_columns && !columns) { set_error("null argument"); return MIRA_E_BAD_ARG; }
    return graph_jit_source(handle, columns, num_columns, buf, cap, len_out);
}
int mira_g1_mul_add(int curve, const uint64_t acc[8], const uint64_t scalar[4], const uint64_t point[8], uint64_t out[8]) {
    if ((curve != MIRA_CURVE_BN256 && curve != MIRA_CURVE_GRUMPKIN) || !acc || !scalar || !point || !out) { set_error("bad arguments"); return MIRA_E_BAD_ARG; }
    if (curve == MIRA_CURVE_BN256) g1_mul_add_t<FqP, FrP>(acc, scalar, point, out);
    else g1_mul_add_t<FrP, FqP>(acc, scalar, point, out);
    return MIRA_OK;
}
int mira_g1_lincomb(int curve, const uint64_t acc[8], const uint64_t *scalars, const uint64_t *points, size_t count, uint64_t out[8]) {
    if ((curve != MIRA_CURVE_BN256 && curve != MIRA_CURVE_GRUMPKIN) || !acc || !out || (count && (!scalars || !points)) || count > 64) { set_error("bad arguments"); return MIRA_E_BAD_ARG; }
    if (curve == MIRA_CURVE_BN256) g1_lincomb_t<FqP, FrP>(acc, scalars, points, count, out);
    else g1_lincomb_t<FrP, FqP>(acc, scalars, points, count, out);
    return MIRA_OK;
}
int mira_g1_fold_commitments(int curve, const uint64_t r[4], const uint64_t *w1, const uint64_t *w2, size_t nw, const uint64_t e[8], const uint64_t *t_commits,
                             size_t count, uint64_t *w_out, uint64_t e_out[8]) {
    if ((curve != MIRA_CURVE_BN256 && curve != MIRA_CURVE_GRUMPKIN) || !r || !e || !e_out || (nw && (!w1 || !w2 || !w_out)) || (count && !t_commits) || nw > 64 || count > 64) {
        set_error("bad arguments");
        return MIRA_E_BAD_ARG;
    }
    if (curve == MIRA_CURVE_BN256) g1_fold_commitments_t<FqP, FrP>(r, w1, w2, nw, e, t_commits, count, w_out, e_out);
    else g1_fold_commitments_t<FrP, FqP>(r, w1, w2, nw, e, t_commits, count, w_out, e_out);
    return MIRA_OK;
}
int mira_msm_download_bases(uint64_t handle, size_t first, size_t n, uint64_t *bases_out) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    auto it = g_bases.find(handle);
    if (it == g_bases.end()) { set_error("unknown bases handle"); return MIRA_E_BAD_ARG; }
    const Bases &bs = it->second;
    if (first > bs.n || n > bs.n - first || (n && !bases_out)) { set_error("range outside the registered key"); return MIRA_E_BAD_ARG; }
    if (!n) return MIRA_OK;
    if ((rc = g.scalars_stage.ensure(n * 64))) return rc;
    if ((rc = bs.curve == MIRA_CURVE_BN256 ? export_bases_bn256(bs, first, n, g.scalars_stage.p) : export_bases_grumpkin(bs, first, n, g.scalars_stage.p))) return rc;
    RT_CHECK(rt_d2h(bases_out, g.scalars_stage.p, n * 64, g.stream));
    RT_CHECK(rt_sync(g.stream));
    return MIRA_OK;
}

int mira_synth_scalars_device(int curve, size_t n, uint64_t index0, uint64_t seed, int kind, void *d_out) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    if ((curve != MIRA_CURVE_BN256 && curve != MIRA_CURVE_GRUMPKIN) || (n && !d_out)) { set_error("bad synth arguments"); return MIRA_E_BAD_ARG; }
    if (!n) return MIRA_OK;
    return curve == MIRA_CURVE_BN256 ? synth_scalars_bn256(n, index0, seed, kind, d_out) : synth_scalars_grumpkin(n, index0, seed, kind, d_out);
}
int mira_synth_bases_device(int curve, size_t n, uint64_t index0, uint64_t seed, void *d_out) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    if ((curve != MIRA_CURVE_BN256 && curve != MIRA_CURVE_GRUMPKIN) || (n && !d_out)) { set_error("bad synth arguments"); return MIRA_E_BAD_ARG; }
    if (!n) return MIRA_OK;
    return curve == MIRA_CURVE_BN256 ? synth_bases_bn256(n, index0, seed, d_out) : synth_bases_grumpkin(n, index0, seed, d_out);
}

int mira_dev_alloc(size_t bytes, void **d_out) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    if (!d_out) { set_error("null output"); return MIRA_E_BAD_ARG; }
    if (rt_malloc(d_out, std::max<size_t>(bytes, 64)) != hipSuccess || !*d_out) { set_error("device allocation failed"); return MIRA_E_ALLOC; }
    return MIRA_OK;
}
int mira_dev_free(void *d) { std::lock_guard<std::mutex> lk(g_lock); if (d) (void)rt_free(d); return MIRA_OK; }
int mira_dev_upload(void *d_dst, const void *h_src, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    RT_CHECK(rt_h2d(d_dst, h_src, bytes, g.stream));
    RT_CHECK(rt_sync(g.stream));
    return MIRA_OK;
}
int mira_dev_download(void *h_dst, const void *d_src, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    RT_CHECK(rt_d2h(h_dst, d_src, bytes, g.stream));
    RT_CHECK(rt_sync(g.stream));
    return MIRA_OK;
}
int mira_dev_copy(void *d_dst, const void *d_src, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    if (bytes && (!d_dst || !d_src)) { set_error("null argument"); return MIRA_E_BAD_ARG; }
    RT_CHECK(rt_d2d(d_dst, d_src, bytes, g.stream));
    RT_CHECK(rt_sync(g.stream));
    return MIRA_OK;
}
int mira_dev_sync(void) {
    std::lock_guard<std::mutex> lk(g_lock);
    int rc = ensure_ctx();
    if (rc) return rc;
    RT_CHECK(rt_sync(g.stream));
    return MIRA_OK;
}

int mira_set_timing(int enabled) { std::lock_guard<std::mutex> lk(g_lock); g.tm.enabled = enabled != 0; return MIRA_OK; }
int mira_get_timings(const char **names, float *ms, int capacity) {
    std::lock_guard<std::mutex> lk(g_lock);
    int cnt = (int)std::min(g.tm.names.size(), g.tm.ms.size());
    for (int i = 0; i < cnt && i < capacity; i++) {
        if (names) names[i] = g.tm.names[i];
        if (ms) ms[i] = g.tm.ms[i];
    }
    return cnt;
}

}   // extern "C"
