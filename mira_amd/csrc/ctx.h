// Shared host-side state of libmira_gpu.so: the bound device, its work stream, grow-only
// device workspaces and the stage timers.  One translation unit per curve keeps hipcc builds
// parallel; they all meet here.
#pragma once
#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mira_gpu.h"
#include "platform.h"

void set_error(const std::string &s);

#ifndef MIRA_CPU_EMU
#define RT_CHECK(expr)                                                                     \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                  \
            return MIRA_E_NO_DEVICE;                                                       \
        }                                                                                  \
    } while (0)
static inline hipError_t rt_malloc(void **p, size_t n) { return hipMalloc(p, n); }
static inline hipError_t rt_free(void *p) { return hipFree(p); }
static inline hipError_t rt_memset(void *p, int v, size_t n, hipStream_t s) { return hipMemsetAsync(p, v, n, s); }
static inline hipError_t rt_h2d(void *d, const void *h, size_t n, hipStream_t s) { return hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s); }
static inline hipError_t rt_d2h(void *h, const void *d, size_t n, hipStream_t s) { return hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s); }
static inline hipError_t rt_d2d(void *d, const void *s_, size_t n, hipStream_t s) { return hipMemcpyAsync(d, s_, n, hipMemcpyDeviceToDevice, s); }
static inline hipError_t rt_sync(hipStream_t s) { return hipStreamSynchronize(s); }
static inline hipError_t rt_last() { return hipGetLastError(); }
static inline hipError_t rt_host_alloc(void **p, size_t n) { return hipHostMalloc(p, n, hipHostMallocDefault); }
static inline hipError_t rt_host_free(void *p) { return hipHostFree(p); }
// pinned host memory a kernel writes directly (fine-grained, coherent): *dev is the address the device uses for *host
static inline hipError_t rt_host_alloc_mapped(void **host, void **dev, size_t n) {
    hipError_t e = hipHostMalloc(host, n, hipHostMallocMapped | hipHostMallocCoherent);
    return e != hipSuccess ? e : hipHostGetDevicePointer(dev, *host, 0);
}
// Wait until a kernel on `s` has stored `stamp` into the mapped word `flag` (its results lie in mapped memory in front of
// that store).  Spins on the word -- hipStreamSynchronize costs 4 - 7 us more (tools/host_epilogue_probe.hip) -- and asks
// the stream now and then, so that a kernel that died ends the wait with the runtime's error.
static inline hipError_t rt_wait_flag(const uint64_t *flag, uint64_t stamp, hipStream_t s) {
    for (uint64_t spins = 1;; spins++) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == stamp) return hipSuccess;
        if ((spins & 0xFFFF) == 0) {
            const hipError_t q = hipStreamQuery(s);
            if (q == hipSuccess) return __atomic_load_n(flag, __ATOMIC_ACQUIRE) == stamp ? hipSuccess : hipErrorUnknown;
            if (q != hipErrorNotReady) return q;
        }
        __builtin_ia32_pause();
    }
}
static inline hipError_t rt_event_sync(hipEvent_t e) { return hipEventSynchronize(e); }
// `waiter` does not start work enqueued after this call before everything enqueued on `done` so far has finished
static inline hipError_t rt_stream_wait(hipStream_t waiter, hipStream_t done, hipEvent_t ev) {
    hipError_t e = hipEventRecord(ev, done);
    return e != hipSuccess ? e : hipStreamWaitEvent(waiter, ev, 0);
}
#else
typedef int hipEvent_t;
static inline int rt_stream_wait(hipStream_t, hipStream_t, hipEvent_t) { return 0; }
static inline int rt_event_sync(hipEvent_t) { return 0; }
#define RT_CHECK(expr) do { (void)(expr); } while (0)
static inline int rt_malloc(void **p, size_t n) { *p = aligned_alloc(64, (n + 63) / 64 * 64); return *p ? 0 : 1; }
static inline int rt_free(void *p) { free(p); return 0; }
static inline int rt_memset(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline int rt_h2d(void *d, const void *h, size_t n, hipStream_t) { memcpy(d, h, n); return 0; }
static inline int rt_d2h(void *h, const void *d, size_t n, hipStream_t) { memcpy(h, d, n); return 0; }
static inline int rt_d2d(void *d, const void *s_, size_t n, hipStream_t) { memcpy(d, s_, n); return 0; }
static inline int rt_sync(hipStream_t) { return 0; }
static inline int rt_last() { return 0; }
static inline int rt_host_alloc(void **p, size_t n) { *p = malloc(n); return *p ? 0 : 1; }
static inline int rt_host_free(void *p) { free(p); return 0; }
static inline int rt_host_alloc_mapped(void **host, void **dev, size_t n) { *host = malloc(n); *dev = *host; return *host ? 0 : 1; }
static inline int rt_wait_flag(const uint64_t *flag, uint64_t stamp, hipStream_t) { return *flag == stamp ? 0 : 1; }   // kernels have run when the launch returns
#endif

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    void release() { if (p) (void)rt_free(p); p = nullptr; cap = 0; }
    int ensure(size_t bytes) {
        if (bytes <= cap) return MIRA_OK;
        if (p) (void)rt_free(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        if (rt_malloc(&p, want) != hipSuccess || !p) {
            p = nullptr;
            set_error("device allocation of " + std::to_string(want) + " bytes failed");
            return MIRA_E_ALLOC;
        }
        cap = want;
        return MIRA_OK;
    }
};

struct Timing {
    bool enabled = false;
    std::vector<const char *> names;
    std::vector<float> ms;
#ifndef MIRA_CPU_EMU
    std::vector<hipEvent_t> ev;
#endif
};

struct Ctx {
    bool ready = false;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t copy_stream = nullptr;   // host scalars travel here, chunk by chunk, beside the kernels of earlier chunks
    std::vector<hipEvent_t> copy_events;
    int32_t forced_c = 0;
    int64_t tune[24] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};   // mira_set_tuning overrides, < 0 = default
    Timing tm;
    // MSM workspace (grow-only, shared by all handles: calls are serialised by the ABI lock)
    DevBuf digits, counts, offsets, cursor, block_sums, sorted_idx, bucket_sums, part, coarse_offsets, fine_counts, fine_cursor;
    DevBuf head_part, tail_part, tail_key, heavy, heavy_out, chunks, window_sums, scalars_stage, consts;
    // NTT workspace
    DevBuf ntt_tmp, ntt_stage, ntt_consts;
    static constexpr int NTT_SETS = 4;           // cached twiddle-table sets (ntt.hip)
    DevBuf ntt_set[NTT_SETS];
    std::string ntt_set_key[NTT_SETS];
    uint64_t ntt_set_stamp[NTT_SETS] = {0, 0, 0, 0}, ntt_stamp = 0;
    int ntt_set_cur = 0;
    DevBuf fold_consts;
    DevBuf graph_consts, graph_ws;
    DevBuf hist_dev;
    DevBuf tree_w, tree_a, tree_b;   // weighted tree reduction: level weights, ping-pong partial results
    uint32_t hist_host[256] = {};    // the statistics of the last commit that collected them
    int32_t last_c = 0, last_w = 0;  // mira_msm_last_plan
    int32_t last_table_c = 0;        // mira_msm_last_table_bits: width of the table set the last commit went through, 0 = none
    uint32_t hist_sel = 0;           // which of the two device histograms the next commit adds into
    unsigned char *out_host = nullptr, *out_host_dev = nullptr; size_t out_host_cap = 0;   // mapped pinned memory: k_set_finish writes the pieces + statistics + a flag word there
    uint64_t out_stamp = 0; DevBuf finish_ctr;   // cross-term evaluator: staged program, intermediates[slot][lane]
    void *windows_dst = nullptr;     // mira_msm_partial_to_device: device destination of the window sums of the call in flight
    uint64_t next_handle = 1;
};
extern Ctx g;

struct Bases {
    int curve;
    size_t n;
    void *d = nullptr;
    bool owned = false;
    int32_t forced_c = 0;     // mira_msm_set_handle_window_bits: this key's window width, 0 = planner / process default
    void *tables = nullptr;   // fixed-base window tables 2^(table_c w) P_i, w < table_w (table_kernels.cuh: 20 or 22 bits), or null
    uint32_t table_c = 0, table_w = 0;
    // shared-bucket table sets (8 .. 16 bits, msm_host.cuh): any number of widths beside each other, each W x the key
    struct SharedSet { void *p; uint32_t c, W; };
    std::vector<SharedSet> shared;
    mutable void *glv = nullptr;   // the interleaved key of the GLV split, 2 n points (mira_msm_precompute_ex(handle, MIRA_TABLE_GLV), or built at the first commit that can use it), or null
    mutable bool glv_auto_failed = false;   // the automatic build could not allocate: not tried again
    // bit-length histogram of the scalars of the previous commit of stat_n elements over this key
    // (planning input for the next one of the same length; never affects a result)
    mutable uint32_t stat_hist[256] = {0};
    mutable size_t stat_n = 0;
    mutable int stat_kind = 0;     // what was histogrammed: 0 = whole scalars (per-window path, shared-bucket sets), 1 = the halves of the GLV split
    // Width trials (capi.hip: trial_*): successive fold steps commit vectors of one shape over one key thousands of times, so the
    // planner's choice for a shape is CHECKED against its neighbours on the first few commits -- the model's width and the four
    // around it, each timed twice -- and the fastest one measured is kept.  Never changes a result.
    struct WidthTrial {
        size_t n = 0; uint32_t count = 0, kind = 0;               // the shape: pairs, commitments per submission, bit 0 = GLV split, bit 1 = host scalars
        uint32_t c0 = 0, best_c = 0, cur_c = 0;
        double best_us = 0, cur_us = 0;
        int cur_runs = 0, steps = 0;
        bool done = false;
        uint64_t stamp = 0;
    };
    mutable std::vector<WidthTrial> trials;
    mutable uint64_t trial_stamp = 0;
};

void tm_begin();
void tm_mark(const char *name);
void tm_end();

static inline size_t tuned(int knob, size_t dflt) { return g.tune[knob] < 0 ? dflt : (size_t)g.tune[knob]; }
static inline uint32_t ceil_div(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

#ifndef MSM_HIST_WGS
#define MSM_HIST_WGS 256      // workgroups of k_hist / k_scatter over one chunk: one per CU -- every workgroup zeroes and flushes a whole bucket set
                              // (B global atomics), so fewer of them win until a CU is left idle (profiles/r04_x_hist_workgroups.txt; probe builds override it)
#endif
// window width c, W windows, B = 2^(c-1) buckets per window, histogram tiling; accumulate:
// `lanes` resident lanes, minimum segment length L, at most T segments; reduction chunk m
struct MsmPlan {
    uint32_t c, W, B, NB, tile, ntiles, L, T, m, nchunks, lanes;
    uint32_t count, Wt;   // MSMs in this submission, total windows count * W
    uint64_t stride;      // scalars of MSM b start at element b * stride
    // fixed-base tables with 8 .. 16-bit windows: the W windows share ONE set of B buckets (NB = B), entries name
    // table points w * table_n + first + i, and the window sums come back as `sums` plain partial sums
    bool shared = false;
    const void *shared_tables = nullptr;   // the W tables of this width, table_n points each
    uint64_t table_n = 0;
    uint32_t sums = 0;
    // GLV (glv.cuh): the n scalars become 2 n half-length ones over the interleaved key [P_i, phi(P_i)]; W = ceil(129 / c)
    bool glv = false;
    const void *glv_bases = nullptr;
    bool stats = false;   // also histogram the bit lengths of the scalars (planning input of the next commit of this shape)
    // a batch whose vectors are still in HOST memory (mira_msm_batch): h_batch[b] = vector b; they cross PCIe in point chunks beside
    // the kernels, chunk k staged as count consecutive slices of its length (the stride of the digit kernel is then the chunk's)
    const uint64_t *const *h_batch = nullptr;
    // bucket reduction (reduce_kernels.cuh, plan_reduction below): nsets bucket sets of 2^cb buckets each, chunks of 2^lambda
    // buckets, 2^kappa chunks per workgroup, 2^gamma workgroup nodes per set, `pieces` results per set; rquad: phase A by quads
    uint32_t nsets = 0, cb = 0, lambda = 0, kappa = 0, gamma = 0, pieces = 1;
    bool rquad = false;
    double est_us = 0;    // the planner's estimate for this plan (0: width forced, nothing estimated)
};

// first bit position of piece p of P over the cb bits of a bucket index (p = P: cb)
static inline uint32_t piece_start(uint32_t cb, uint32_t P, uint32_t p) { return (uint32_t)(((uint64_t)p * cb + P - 1) / P); }
// What a launch sequence hands back: W * P points whose weighted sum is the commitment,
//     sum_(w < W) sum_(p < P) 2^(c w + piece_start(cb, P, p)) point[w P + p]
// per-window buckets: c = window width, cb = c - 1; a shared-bucket table set: c = 0, W = 1, cb = its width - 1;
// wide tables: c = 0, W = 64 plain partial sums, P = 1.  The public partial format (mira_msm_partial_*) is P = 1.
struct PartialShape { uint32_t c = 0, W = 0, cb = 0, P = 1; };

// Shape of the bucket reduction of a plan.  Quads for phase A while the chunks are few (the chain of dependent additions is
// what takes the time: chunks of four buckets), single lanes in chunks of eight once they fill the SIMDs.
static inline void plan_reduction(MsmPlan &p, uint32_t want_pieces) {
    p.nsets = p.shared ? p.count : p.Wt;
    p.cb = p.c - 1;
    const uint64_t chunks_q = (uint64_t)p.nsets << (p.cb - std::min<uint32_t>(2, p.cb));
    p.rquad = g.tune[MIRA_TUNE_REDUCE_QUAD] >= 0 ? g.tune[MIRA_TUNE_REDUCE_QUAD] != 0 : chunks_q * 4 <= 98304;
    p.lambda = std::min<uint32_t>(p.rquad ? 2 : 3, p.cb);
    if (g.tune[MIRA_TUNE_REDUCE_LAMBDA] >= 1) p.lambda = std::min<uint32_t>((uint32_t)g.tune[MIRA_TUNE_REDUCE_LAMBDA], p.cb);
    // k_set_finish holds the 2^gamma nodes of a set in LDS, (kappa + 2) points each: with 128 quads per workgroup (kappa = 7) that is
    // gamma <= 6, i.e. eta <= 13 -- a 16-bit set in chunks of two buckets would need more: wider chunks there (tuning knobs included)
    while (p.rquad && p.cb - p.lambda > 13) p.lambda++;
    const uint32_t eta = p.cb - p.lambda;
#ifdef MIRA_CPU_EMU
    p.kappa = std::min<uint32_t>(eta, 3);                                      // emulated lanes are OS threads: small workgroups, a taller second tree
#else
    p.kappa = std::min<uint32_t>(eta, p.rquad ? (eta > 12 ? 7 : 6) : 8);      // 64 (128) quads or 256 lanes per workgroup
#endif
    p.gamma = eta - p.kappa;
    p.pieces = std::max<uint32_t>(1, std::min<uint32_t>(std::min<uint32_t>(want_pieces, 8), std::max<uint32_t>(1, p.cb / 2)));
    p.m = 1u << p.lambda;
    p.nchunks = p.B >> p.lambda;
}

// per-curve translation units (msm_bn256.hip / msm_grumpkin.hip)
// h_scalars != null: the scalars are still in host memory; d_scalars is then the device staging buffer they are copied to
int msm_launch_bn256(const Bases &bs, size_t first, const void *d_scalars, const void *h_scalars, size_t n, const MsmPlan &p, uint64_t *host_windows);
int msm_launch_grumpkin(const Bases &bs, size_t first, const void *d_scalars, const void *h_scalars, size_t n, const MsmPlan &p, uint64_t *host_windows);
int msm_launch_table_bn256(const Bases &bs, size_t first, const void *d_scalars, size_t n, uint64_t *host_sums);
int msm_launch_table_grumpkin(const Bases &bs, size_t first, const void *d_scalars, size_t n, uint64_t *host_sums);
int build_tables_bn256(Bases &bs, uint32_t c, uint32_t W);
int build_tables_grumpkin(Bases &bs, uint32_t c, uint32_t W);
int build_glv_bn256(Bases &bs, const void *d_beta_r261);
int build_glv_grumpkin(Bases &bs, const void *d_beta_r261);
int curve_init_bn256();
int convert_bases_bn256(const void *d_src, void *d_dst, size_t n);
int convert_bases_grumpkin(const void *d_src, void *d_dst, size_t n);
int curve_init_grumpkin();
int synth_scalars_bn256(size_t n, uint64_t index0, uint64_t seed, int kind, void *d_out);
int synth_scalars_grumpkin(size_t n, uint64_t index0, uint64_t seed, int kind, void *d_out);
int synth_bases_bn256(size_t n, uint64_t index0, uint64_t seed, void *d_out);
int synth_bases_grumpkin(size_t n, uint64_t index0, uint64_t seed, void *d_out);
int check_bases_bn256(const Bases &bs, uint32_t *d_bad);
int check_bases_grumpkin(const Bases &bs, uint32_t *d_bad);
// key cache file <-> HBM (msm_host.cuh): read 2^k points in chunks beside their conversion / validation; write
int load_bases_file_bn256(Bases &b, int fd, bool validate, uint32_t *d_bad);
int load_bases_file_grumpkin(Bases &b, int fd, bool validate, uint32_t *d_bad);
int save_bases_file_bn256(const Bases &b, int fd);
int save_bases_file_grumpkin(const Bases &b, int fd);

// fold.hip
int fold_witness_device(int field, void *d_out, const void *d_w1, const void *d_w2, const uint64_t r[4], size_t n);
int fold_error_device(int field, void *d_e, const void *const *d_terms, size_t K, const uint64_t r[4], size_t n);
int fold_relaxed_device(int field, void *d_w_out, const void *d_w1, const void *d_w2, size_t n_w, void *d_e_out, const void *d_e, const void *const *d_terms, size_t K,
                        const uint64_t r[4], size_t n);
int lincomb_device(int field, void *d_out, const void *const *d_vecs, const uint64_t *coeffs, size_t K, size_t n);
int lincomb_multi_device(int field, void *const *d_outs, size_t M, const void *const *d_vecs, size_t J, const uint64_t *coeffs, size_t n);
int pow_tree_reduce_device(int field, const void *d_leaves, uint32_t levels, size_t leaf_point_stride, const uint64_t *weights, uint32_t P, uint64_t *out);
int export_bases_bn256(const Bases &bs, size_t first, size_t n, void *d_out);
int export_bases_grumpkin(const Bases &bs, size_t first, size_t n, void *d_out);

// graph.hip
int graph_compile(int field, const mira_graph *gr, uint32_t num_challenges, uint32_t num_columns, uint64_t *handle_out);
int graph_eval_compiled(uint64_t handle, const mira_eval_column *columns, uint32_t num_columns, const uint64_t *challenges, uint32_t num_challenges,
                        size_t num_rows, void *d_out);
int graph_free(uint64_t handle);
int graph_eval_batch(const uint64_t *handles, uint32_t count, const mira_eval_column *columns, uint32_t num_columns, const uint64_t *challenges,
                     uint32_t num_challenges, size_t num_rows, void *const *d_outs);
int graph_specialize(const uint64_t *handles, uint32_t count, const mira_eval_column *columns, uint32_t num_columns,
                     std::unique_lock<std::mutex> *library_lock);   // held on entry and on return; released while the compiler runs
int graph_is_specialized(uint64_t handle, int32_t *out);
int graph_set_cache_dir(const char *dir);
int graph_jit_compile_check(const char *src, size_t *code_size_out);
int graph_jit_stats(uint32_t *compiled_out, uint32_t *from_disk_out);
int graph_jit_source(uint64_t handle, const mira_eval_column *columns, uint32_t num_columns, char *buf, size_t cap, size_t *len_out);
int graph_eval_device(int field, const mira_graph *gr, const mira_eval_column *columns, uint32_t num_columns, const uint64_t *challenges,
                      uint32_t num_challenges, size_t num_rows, void *d_out);

// ntt.hip
enum NttKind { NTT_BEST, NTT_FFT, NTT_IFFT, NTT_COSET_FFT, NTT_COSET_IFFT };
int ntt_init();
int ntt_kind_device(void *d_a, uint32_t log_n, NttKind kind, const uint64_t *omega_in);
int ntt_get_omega_or_inv(uint32_t k, bool inverse, uint64_t out[4]);
