// Host-side launch sequence of one MSM (see msm_kernels.cuh for the pipeline).  Included by
// the per-curve translation units.
#pragma once
#include "ctx.h"
#ifndef MSM_DIGITS_BLOCK
#define MSM_DIGITS_BLOCK 256   // lanes per workgroup of k_digits (probe builds override it)
#endif
#include "msm_kernels.cuh"
#include "table_kernels.cuh"
#include "sort_kernels.cuh"
#include "reduce_kernels.cuh"
#include "host_field.hpp"
#include <cerrno>
#include <thread>
#include <unistd.h>

#ifdef MIRA_CPU_EMU
static constexpr uint32_t FIXUP_HEAVY_GRID = 4;     // emulated lanes are OS threads: keep the idle grid small (stage A: a quarter of it)
static constexpr uint32_t FIXUP_HEAVY_BLOCKS = 1;
#else
#ifndef MIRA_FIXUP_HEAVY_BLOCKS
#define MIRA_FIXUP_HEAVY_BLOCKS 256
#endif
static constexpr uint32_t FIXUP_HEAVY_BLOCKS = MIRA_FIXUP_HEAVY_BLOCKS; // workgroups of k_fixup_all that take the heavy sub-jobs: one per CU (they leave at once when there are none)
static constexpr uint32_t FIXUP_HEAVY_GRID = 1024;   // waves of either heavy stage = 256 workgroups of 256 lanes, one per CU (the sub-jobs of stage A are sized to just fill them; dispatching 1024 mostly idle workgroups alone took 15 us)
#endif

// quads for the bucket reduction while its work items number less than ~1.5 waves per SIMD (the
// chain of dependent additions is what takes the time there); single lanes beyond (throughput)
static constexpr uint32_t SCAN_OWN_PREFIX_TILES = 32;   // up to 2^16 bucket counters: k_scan_c<SCAN_OWN_PREFIX>, one launch instead of three
static constexpr uint32_t SCAN_SOLO_TILES = 2;      // up to 4096 bucket counters: k_scan_c<SCAN_SOLO> (measured: 14 -> 7 us for one tile, 15 -> 13 for two, 16 -> 30 for seven)
static inline bool reduce_with_quads(uint64_t work_items) { return work_items * 4 <= 98304; }
#ifdef MIRA_CPU_EMU
static constexpr uint32_t SET_FINISH_BLOCK = 32;      // emulated lanes are OS threads
#else
static constexpr uint32_t SET_FINISH_BLOCK = 256;
#endif

// Host scalars (commit(&self, v: &[C::Scalar]) hands over host memory, src/commitment.rs:78) are
// cut into point chunks: chunk k + 1 crosses PCIe on the copy stream while the kernels of chunk k
// run.  Every chunk goes through digits .. fix-up on its own, its bucket runs starting from the sums
// of the chunks before (k_accumulate<F, true>); bucket reduction and window sums run once.  The first chunk is small (its
// copy is the only one nothing hides), the following ones double up to 2^20 pairs.
static inline std::vector<size_t> host_chunks(size_t n, size_t min_n) {
    std::vector<size_t> ends;
    if (n < min_n) { ends.push_back(n); return ends; }
    size_t done = 0, step = std::max<size_t>(1, min_n / 2);         // 2^18 by default
    while (done < n) {
        size_t len = std::min(step, n - done);
        if (n - done - len < step / 2) len = n - done;             // a short tail joins the last chunk
        done += len;
        ends.push_back(done);
        if (step < 4 * std::max<size_t>(1, min_n / 2)) step <<= 1;   // up to 2^20
    }
    return ends;
}

template <class F, class FS>
static int msm_launch_body(const Bases &bs, size_t first, const void *d_scalars, const void *h_scalars, size_t n, const MsmPlan &p,
                           uint64_t *host_windows /* count * W * 16 u64 */);
// A failure inside the launch sequence (an allocation, a runtime error) must not leave work behind that the next
// call would trip over: the copy stream may still be reading the caller's host buffer, the work stream may still
// be using the workspaces, and the planning histogram this commit was to fill is half written.
template <class F, class FS>
static int msm_launch(const Bases &bs, size_t first, const void *d_scalars, const void *h_scalars, size_t n, const MsmPlan &p,
                      uint64_t *host_windows) {
    const uint32_t hist_sel_before = g.hist_sel;
    const int rc = msm_launch_body<F, FS>(bs, first, d_scalars, h_scalars, n, p, host_windows);
    if (rc != MIRA_OK) {
        if (g.copy_stream) (void)rt_sync(g.copy_stream);
        (void)rt_sync(g.stream);
        (void)rt_last();
        if (p.stats && g.hist_dev.p) {                       // both histograms back to zero, selector as before
            (void)rt_memset(g.hist_dev.p, 0, 2048, g.stream);
            (void)rt_sync(g.stream);
            g.hist_sel = hist_sel_before;
        }
    }
    return rc;
}
template <class F, class FS>
static int msm_launch_body(const Bases &bs, size_t first, const void *d_scalars, const void *h_scalars, size_t n, const MsmPlan &p,
                           uint64_t *host_windows /* count * W * 16 u64 */) {
    int rc;
    // chunks: [0, ends[0]), [ends[0], ends[1]), ...  One chunk unless the scalars are in host memory (a single vector, or the vectors
    // of a batch: the chunking is over POINTS, a chunk of a batch holds that range of every vector) or the commit is too long for
    // the 32-bit entry offsets of one pass (n W >= 2^32: the reference's 2^27 .. 2^28-point keys, examples/groth16/main.rs:47-75).
    const size_t mult = p.glv ? 2 : 1;                        // columns of the digit matrix per scalar
    const size_t host_min = tuned(MIRA_TUNE_HOST_CHUNK_MIN_N, (size_t)1 << 19);
    std::vector<size_t> ends;
    if (h_scalars && p.count == 1) ends = host_chunks(n, host_min);
    else if (p.h_batch) {
        for (size_t e : host_chunks(n * p.count, host_min)) ends.push_back(std::min(n, (e + p.count - 1) / p.count));
        ends.back() = n;
        ends.erase(std::unique(ends.begin(), ends.end()), ends.end());
    } else ends = {n};
    {
        const uint64_t pass_entries = (1ull << tuned(MIRA_TUNE_PASS_ENTRIES_LOG, 32)) - 1;          // (tests cut small commits this way)
        const size_t max_chunk = std::max<size_t>(64, (size_t)(pass_entries / ((uint64_t)mult * p.Wt)) / 64 * 64);
        std::vector<size_t> cut;
        for (size_t k = 0, lo = 0; k < ends.size(); lo = ends[k++])
            for (size_t at = lo; at < ends[k];) { at = std::min(ends[k], at + max_chunk); cut.push_back(at); }
        ends.swap(cut);
    }
    size_t nmax = 0;
    for (size_t k = 0, lo = 0; k < ends.size(); lo = ends[k++]) nmax = std::max(nmax, ends[k] - lo);
    nmax *= mult;
    const size_t entries_max = nmax * p.Wt;
    if ((rc = g.digits.ensure(entries_max * 2))) return rc;
    if ((rc = g.counts.ensure(((size_t)p.NB + 1) * 4))) return rc;
    if ((rc = g.offsets.ensure(((size_t)p.NB + 1) * 4))) return rc;
    if ((rc = g.cursor.ensure(((size_t)p.NB + 1) * 4))) return rc;
    const uint32_t scan_blocks = ceil_div(p.NB, SCAN_TILE);
    if (scan_blocks > 1024) { set_error("window configuration exceeds the scan capacity"); return MIRA_E_UNSUPPORTED; }
    if ((rc = g.block_sums.ensure(1024 * 4))) return rc;
    if ((rc = g.sorted_idx.ensure(entries_max * 4 + 8))) return rc;
    const size_t staged_min_n = tuned(MIRA_TUNE_STAGED_MIN_N, (size_t)1 << 19);
    if (nmax * p.count >= staged_min_n && p.c >= 9) {
        if ((rc = g.part.ensure(entries_max * 8 + 8))) return rc;
        if ((rc = g.coarse_offsets.ensure(((size_t)p.Wt * 512 + 1) * 4))) return rc;
    }
    if ((rc = g.bucket_sums.ensure((size_t)p.NB * XYZZ29_BYTES))) return rc;
    if ((rc = g.head_part.ensure((size_t)p.T * XYZZ29_BYTES))) return rc;
    if ((rc = g.tail_part.ensure((size_t)p.T * XYZZ29_BYTES))) return rc;
    if ((rc = g.tail_key.ensure((size_t)p.T * 4))) return rc;
    if ((rc = g.heavy.ensure(64 + ((size_t)p.T / 2 + 8) * 32 + ((size_t)p.T / 4 + 8) * 16))) return rc;   // counters, runs, sub-job descriptors, medium runs
    if ((rc = g.heavy_out.ensure(((size_t)p.T / 2 + 8) * XYZZ29_BYTES))) return rc;
    const size_t node_points = ((size_t)p.nsets << p.gamma) * (p.kappa + 2);               // one node of kappa + 2 points per workgroup of k_bucket_tree
    if ((rc = g.chunks.ensure(node_points * XYZZ29_BYTES))) return rc;
    if ((rc = g.window_sums.ensure((size_t)p.nsets * p.pieces * 128 + 1024))) return rc;   // pieces + the planning statistics
#ifndef MIRA_CPU_EMU
    if (h_scalars || p.h_batch) {
        if (!g.copy_stream) RT_CHECK(hipStreamCreateWithFlags(&g.copy_stream, hipStreamNonBlocking));
        while (g.copy_events.size() < ends.size()) {
            hipEvent_t e;
            RT_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            g.copy_events.push_back(e);
        }
    }
#endif

    hipStream_t st = g.stream;
    uint32_t *heavy_count = reinterpret_cast<uint32_t *>(g.heavy.p);          // [0] runs, [1] sub-jobs, [2..3] plan
    U4 *heavy_runs = reinterpret_cast<U4 *>(heavy_count + 16);
    U4 *heavy_subs = heavy_runs + ((size_t)p.T / 4 + 4);
    U4 *heavy_meds = reinterpret_cast<U4 *>(reinterpret_cast<unsigned char *>(g.heavy.p) + 64 + ((size_t)p.T / 2 + 8) * 32);
    uint32_t *plan = heavy_count + 2;
    uint32_t *no_u32 = nullptr;
    unsigned char *no_u8 = nullptr;
    // coarse bin of the staged sort = top 8 bits of the bucket id (6..9 measured equal)
    const uint32_t fine_bits = (p.c - 1) - std::min<uint32_t>(p.c - 1, 8);
    const uint32_t CB = p.B >> fine_bits;                                       // <= 256 coarse bins per window

    // No clearing passes: k_digits zeroes the bucket counters, k_scan_c the heavy-run counters and (first
    // chunk) the identity marker of every bucket without entries, every segment of k_accumulate writes its tail key.
    tm_begin();
    // planning statistics: two device histograms, this commit's (zero since the previous one cleared it) and the next one's
    uint32_t *hist = nullptr, *hist_clear = nullptr;
    if (p.stats) {
        if (!g.hist_dev.p) {
            if ((rc = g.hist_dev.ensure(2048))) return rc;
            RT_CHECK(rt_memset(g.hist_dev.p, 0, 2048, st));
            g.hist_sel = 0;
        }
        hist = reinterpret_cast<uint32_t *>(g.hist_dev.p) + 256 * g.hist_sel;
        hist_clear = reinterpret_cast<uint32_t *>(g.hist_dev.p) + 256 * (g.hist_sel ^ 1u);
        g.hist_sel ^= 1u;
    }
    for (size_t k = 0, lo = 0; k < ends.size(); lo = ends[k++]) {
        const size_t ns = ends[k] - lo, nc = ns * mult, entries = nc * p.Wt;   // ns scalars, nc columns of digits
        // a chunk of a host batch is staged as `count` consecutive slices of ns scalars, behind the chunks before it
        const unsigned char *sc = reinterpret_cast<const unsigned char *>(d_scalars) + (p.h_batch ? lo * p.count : lo) * 32;
        const uint64_t stride_k = p.h_batch ? (uint64_t)ns : (uint64_t)p.stride;
        // per-window buckets: entries are chunk-local point indices; shared buckets: entries name table points
        const unsigned char *bases = p.shared ? reinterpret_cast<const unsigned char *>(p.shared_tables)
                                   : p.glv  ? reinterpret_cast<const unsigned char *>(p.glv_bases) + (first + lo) * 128
                                            : reinterpret_cast<const unsigned char *>(bs.d) + (first + lo) * 64;
        const uint32_t wgroup = p.shared ? p.W : 1u, idx_stride = p.shared ? (uint32_t)p.table_n : 0u, idx_first = p.shared ? (uint32_t)(first + lo) : 0u;
        const uint32_t add = k ? 1u : 0u;
        if (h_scalars || p.h_batch) {
            // pageable or pinned, the copy engine moves it beside the kernels of the previous chunk
            hipStream_t cs = g.copy_stream ? g.copy_stream : st;
            if (p.h_batch)
                for (uint32_t b = 0; b < p.count; b++)
                    RT_CHECK(rt_h2d(const_cast<unsigned char *>(sc) + (size_t)b * ns * 32, reinterpret_cast<const unsigned char *>(p.h_batch[b]) + lo * 32, ns * 32, cs));
            else
                RT_CHECK(rt_h2d(const_cast<unsigned char *>(sc), reinterpret_cast<const unsigned char *>(h_scalars) + lo * 32, ns * 32, cs));
            if (cs != st) RT_CHECK(rt_stream_wait(st, cs, g.copy_events[k]));
        }
        // histogram / scatter tiling of this chunk: about one workgroup per CU, at least 1024 points per tile
        const uint32_t tile = (std::max<uint32_t>(1024, ceil_div(nc, std::max<uint32_t>(1, MSM_HIST_WGS / p.Wt))) + 1023) / 1024 * 1024, ntiles = ceil_div(nc, tile);
        const bool staged = nc * p.count >= staged_min_n && p.c >= 9;   // a batch is count MSMs' worth of entries
        if (p.glv)
            LAUNCH((k_digits<FS, true>), dim3(ceil_div(ns, MSM_DIGITS_BLOCK), p.count), MSM_DIGITS_BLOCK, 0, st, sc, (uint32_t)ns, stride_k, p.c, p.W,
                   reinterpret_cast<int16_t *>(g.digits.p), reinterpret_cast<uint32_t *>(g.counts.p), p.NB + 1, hist, hist_clear);
        else
        LAUNCH(k_digits<FS>, dim3(ceil_div(nc, MSM_DIGITS_BLOCK), p.count), MSM_DIGITS_BLOCK, 0, st, sc, (uint32_t)nc, stride_k, p.c, p.W,
               reinterpret_cast<int16_t *>(g.digits.p), reinterpret_cast<uint32_t *>(g.counts.p), p.NB + 1, hist, hist_clear);
        tm_mark("digits");
        LAUNCH_BARRIER_FLEX(k_hist, dim3(ntiles, p.Wt), 1024, (size_t)p.B * 4, st, reinterpret_cast<const int16_t *>(g.digits.p), (uint32_t)nc,
                       p.B, tile, reinterpret_cast<uint32_t *>(g.counts.p), wgroup);
        tm_mark("hist");
        if (scan_blocks <= SCAN_SOLO_TILES) {                 // few buckets: one workgroup scans them all, one launch instead of three
            LAUNCH_BARRIER(k_scan_c<SCAN_SOLO>, 1, SCAN_BLOCK, 0, st, reinterpret_cast<const uint32_t *>(g.counts.p), p.NB,
                           reinterpret_cast<const uint32_t *>(g.block_sums.p), reinterpret_cast<uint32_t *>(g.offsets.p),
                           reinterpret_cast<uint32_t *>(g.cursor.p), staged ? reinterpret_cast<uint32_t *>(g.coarse_offsets.p) : no_u32, fine_bits,
                           plan, p.lanes, p.L, heavy_count, add ? no_u8 : reinterpret_cast<unsigned char *>(g.bucket_sums.p));
        } else if (scan_blocks <= SCAN_OWN_PREFIX_TILES) {    // the bucket counts of the small commits: every workgroup sums what lies in front of it
            LAUNCH_BARRIER(k_scan_c<SCAN_OWN_PREFIX>, scan_blocks, SCAN_BLOCK, 0, st, reinterpret_cast<const uint32_t *>(g.counts.p), p.NB,
                           reinterpret_cast<const uint32_t *>(g.block_sums.p), reinterpret_cast<uint32_t *>(g.offsets.p),
                           reinterpret_cast<uint32_t *>(g.cursor.p), staged ? reinterpret_cast<uint32_t *>(g.coarse_offsets.p) : no_u32, fine_bits,
                           plan, p.lanes, p.L, heavy_count, add ? no_u8 : reinterpret_cast<unsigned char *>(g.bucket_sums.p));
        } else {
        LAUNCH_BARRIER(k_scan_a, scan_blocks, SCAN_BLOCK, 0, st, reinterpret_cast<const uint32_t *>(g.counts.p), p.NB,
                       reinterpret_cast<uint32_t *>(g.block_sums.p));
        LAUNCH_BARRIER(k_scan_b, 1, 1024, 0, st, reinterpret_cast<uint32_t *>(g.block_sums.p), scan_blocks);
        LAUNCH_BARRIER(k_scan_c<SCAN_BLOCK_SUMS>, scan_blocks, SCAN_BLOCK, 0, st, reinterpret_cast<const uint32_t *>(g.counts.p), p.NB,
                       reinterpret_cast<const uint32_t *>(g.block_sums.p), reinterpret_cast<uint32_t *>(g.offsets.p),
                       reinterpret_cast<uint32_t *>(g.cursor.p), staged ? reinterpret_cast<uint32_t *>(g.coarse_offsets.p) : no_u32, fine_bits,
                       plan, p.lanes, p.L, heavy_count, add ? no_u8 : reinterpret_cast<unsigned char *>(g.bucket_sums.p));
        }
        tm_mark("scan");
        // sort: LDS-staged two-level partition for large inputs (bursts of consecutive entries), the
        // single-level scatter otherwise (small inputs: the tile structure buys nothing there)
        if (staged) {
            if (p.shared)
                LAUNCH_BARRIER((k_stage1<int16_t, true, STAGE_TILE_PW>), dim3(ceil_div(nc, STAGE_TILE_PW), p.Wt), 1024, (size_t)STAGE_TILE_PW * 8, st, reinterpret_cast<const int16_t *>(g.digits.p),
                                    (uint32_t)nc, p.B, fine_bits, CB, idx_stride, idx_first, wgroup, reinterpret_cast<uint32_t *>(g.coarse_offsets.p), reinterpret_cast<U2 *>(g.part.p));
            else
                LAUNCH_BARRIER((k_stage1<int16_t, false, STAGE_TILE_PW>), dim3(ceil_div(nc, STAGE_TILE_PW), p.Wt), 1024, (size_t)STAGE_TILE_PW * 8, st, reinterpret_cast<const int16_t *>(g.digits.p),
                                    (uint32_t)nc, p.B, fine_bits, CB, 0u, 0u, 1u, reinterpret_cast<uint32_t *>(g.coarse_offsets.p), reinterpret_cast<U2 *>(g.part.p));
            tm_mark("sort_level1");
            LAUNCH_BARRIER((k_stage2<STAGE_TILE_PW, STAGE_MAX_KEYS2_PW>), ceil_div(entries, STAGE_TILE_PW), 1024, (size_t)STAGE_TILE_PW * 6, st, reinterpret_cast<const U2 *>(g.part.p),
                                reinterpret_cast<const uint32_t *>(g.offsets.p) + p.NB, fine_bits, reinterpret_cast<uint32_t *>(g.cursor.p),
                                reinterpret_cast<uint32_t *>(g.sorted_idx.p));
        } else
            LAUNCH_BARRIER_FLEX(k_scatter, dim3(ntiles, p.Wt), 1024, (size_t)p.B * 4, st, reinterpret_cast<const int16_t *>(g.digits.p),
                       (uint32_t)nc, p.B, tile, reinterpret_cast<uint32_t *>(g.cursor.p), reinterpret_cast<uint32_t *>(g.sorted_idx.p), wgroup, idx_stride, idx_first);
        tm_mark("scatter");
        if (add)
            LAUNCH((k_accumulate<F, true>), ceil_div(p.T, 128), 128, 0, st, reinterpret_cast<const uint32_t *>(g.sorted_idx.p),
                   reinterpret_cast<const uint32_t *>(g.offsets.p), p.NB, bases, (const uint32_t *)plan,
                   reinterpret_cast<unsigned char *>(g.bucket_sums.p), reinterpret_cast<unsigned char *>(g.head_part.p),
                   reinterpret_cast<unsigned char *>(g.tail_part.p), reinterpret_cast<uint32_t *>(g.tail_key.p), heavy_count, heavy_runs, heavy_subs, heavy_meds);
        else
            LAUNCH((k_accumulate<F, false>), ceil_div(p.T, 128), 128, 0, st, reinterpret_cast<const uint32_t *>(g.sorted_idx.p),
                   reinterpret_cast<const uint32_t *>(g.offsets.p), p.NB, bases, (const uint32_t *)plan,
                   reinterpret_cast<unsigned char *>(g.bucket_sums.p), reinterpret_cast<unsigned char *>(g.head_part.p),
                   reinterpret_cast<unsigned char *>(g.tail_part.p), reinterpret_cast<uint32_t *>(g.tail_key.p), heavy_count, heavy_runs, heavy_subs, heavy_meds);
        tm_mark("accumulate");
        const uint32_t fix_by_bucket = p.NB < p.T ? p.NB : 0u;   // fewer buckets than segments: index the fix-up by bucket
        const uint32_t fix_items = fix_by_bucket ? fix_by_bucket : p.T;
        // the short chains and the sub-jobs of the heavy runs (listed by k_accumulate) side by side in one launch
        LAUNCH_BARRIER(k_fixup_all<F>, FIXUP_HEAVY_BLOCKS + ceil_div(fix_items, HEAVY_BLOCK_A / 4), HEAVY_BLOCK_A, 0, st, FIXUP_HEAVY_BLOCKS, (const uint32_t *)plan,
                       reinterpret_cast<const uint32_t *>(g.offsets.p), reinterpret_cast<const unsigned char *>(g.head_part.p), reinterpret_cast<const unsigned char *>(g.tail_part.p),
                       reinterpret_cast<const uint32_t *>(g.tail_key.p), reinterpret_cast<unsigned char *>(g.bucket_sums.p), fix_by_bucket,
                       (const uint32_t *)heavy_count, (const U4 *)heavy_subs, (const U4 *)heavy_runs, reinterpret_cast<unsigned char *>(g.heavy_out.p), (const U4 *)heavy_meds);
        LAUNCH_BARRIER(k_fixup_heavy_b<F>, (FIXUP_HEAVY_GRID + 3) / 4, HEAVY_BLOCK_B, 0, st, (const uint32_t *)heavy_count, (const U4 *)heavy_runs,
                       reinterpret_cast<const unsigned char *>(g.heavy_out.p), reinterpret_cast<const unsigned char *>(g.tail_part.p),
                       reinterpret_cast<unsigned char *>(g.bucket_sums.p));
        tm_mark("fixup");
    }
    // bucket reduction (reduce_kernels.cuh): running sums over chunks of 2^lambda buckets and the first kappa levels of the
    // tree over the chunks in one launch, the remaining gamma levels and the pieces of every set in a second
    PieceCfg pc;
    pc.P = p.pieces;
    for (uint32_t i = 0; i <= 8; i++) pc.start[i] = piece_start(p.cb, p.pieces, std::min(i, p.pieces));
    const uint32_t tree_groups = p.nsets << p.gamma, npts = p.nsets * p.pieces;
    const size_t lds_a = ((size_t)2 << p.kappa) * XYZZ29_BYTES, lds_b = ((size_t)(p.kappa + 2) << p.gamma) * XYZZ29_BYTES;
    if (p.rquad)
        LAUNCH_BARRIER((k_bucket_tree<F, true>), tree_groups, std::max(4u, 4u << p.kappa), lds_a, st, reinterpret_cast<const unsigned char *>(g.bucket_sums.p),
                       p.lambda, p.kappa, reinterpret_cast<unsigned char *>(g.chunks.p));
    else
        LAUNCH_BARRIER((k_bucket_tree<F, false>), tree_groups, std::max(4u, 1u << p.kappa), lds_a, st, reinterpret_cast<const unsigned char *>(g.bucket_sums.p),
                       p.lambda, p.kappa, reinterpret_cast<unsigned char *>(g.chunks.p));
    tm_mark("reduce_chunks");
    uint32_t *no_ctr = nullptr;
    uint64_t *no_flag = nullptr;
    if (g.windows_dst) {                                     // mira_msm_partial_to_device: the sums stay in device memory
        LAUNCH_BARRIER(k_set_finish<F>, p.nsets, SET_FINISH_BLOCK, lds_b, st, reinterpret_cast<const unsigned char *>(g.chunks.p), p.kappa + 2, p.gamma, p.lambda, pc,
                       reinterpret_cast<unsigned char *>(g.window_sums.p), (const uint32_t *)hist, no_ctr, no_flag, (uint64_t)0);
        tm_mark("window_sum");
        RT_CHECK(rt_last());
        RT_CHECK(rt_d2d(g.windows_dst, g.window_sums.p, (size_t)npts * 128, st));
        RT_CHECK(rt_sync(st));
    } else {
        // pieces (and statistics) straight into mapped pinned memory, a flag word behind them: no copy command, no
        // hipStreamSynchronize (7 us of a commit, tools/host_epilogue_probe.hip)
        const size_t bytes = (size_t)npts * 128 + 1024, flag_at = (bytes + 63) / 64 * 64;
        if (g.out_host_cap < flag_at + 64) {
            RT_CHECK(rt_sync(st));
            if (g.out_host) (void)rt_host_free(g.out_host);
            g.out_host = g.out_host_dev = nullptr; g.out_host_cap = 0;
            RT_CHECK(rt_host_alloc_mapped(reinterpret_cast<void **>(&g.out_host), reinterpret_cast<void **>(&g.out_host_dev), flag_at + 64 + 4096));
            g.out_host_cap = flag_at + 64 + 4096;
        }
        if (!g.finish_ctr.p) {
            if ((rc = g.finish_ctr.ensure(64))) return rc;
            RT_CHECK(rt_memset(g.finish_ctr.p, 0, 64, st));
        }
        uint64_t *flag_host = reinterpret_cast<uint64_t *>(g.out_host + flag_at);
        const uint64_t stamp = ++g.out_stamp;
        __atomic_store_n(flag_host, (uint64_t)0, __ATOMIC_RELEASE);
        LAUNCH_BARRIER(k_set_finish<F>, p.nsets, SET_FINISH_BLOCK, lds_b, st, reinterpret_cast<const unsigned char *>(g.chunks.p), p.kappa + 2, p.gamma, p.lambda, pc,
                       g.out_host_dev, (const uint32_t *)hist, reinterpret_cast<uint32_t *>(g.finish_ctr.p), reinterpret_cast<uint64_t *>(g.out_host_dev + flag_at), stamp);
        tm_mark("window_sum");
        RT_CHECK(rt_last());
        // a long commit waits in hipStreamSynchronize as before (nothing to gain there, and the spin asks the stream now and then)
        if (g.tm.enabled || (size_t)n * p.count > ((size_t)1 << 21)) RT_CHECK(rt_sync(st));
        RT_CHECK(rt_wait_flag(flag_host, stamp, st));
        memcpy(host_windows, g.out_host, (size_t)npts * 128);
        if (p.stats) memcpy(g.hist_host, g.out_host + (size_t)npts * 128, 1024);
    }
    tm_end();
    return MIRA_OK;
}


template <class F, class FS> static int curve_init() {
#ifndef MIRA_CPU_EMU
    RT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_window_sum<F, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bucket_tree<F, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bucket_tree<F, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_set_finish<F>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // the LDS-staged histogram needs more than the 64 KiB default (128 KiB at c = 16)
    RT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_hist), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_stage1<int16_t, false, STAGE_TILE_PW>), hipFuncAttributeMaxDynamicSharedMemorySize, STAGE_TILE_PW * 8));   // + 10 KiB static
    RT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_stage1<int32_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, STAGE_TILE * 8));
    RT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_stage1<int16_t, true, STAGE_TILE_PW>), hipFuncAttributeMaxDynamicSharedMemorySize, STAGE_TILE_PW * 8));
    RT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_stage2<STAGE_TILE, STAGE_MAX_KEYS2>), hipFuncAttributeMaxDynamicSharedMemorySize, STAGE_TILE * 6));   // + 52 KiB static
    RT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_stage2<STAGE_TILE_PW, STAGE_MAX_KEYS2_PW>), hipFuncAttributeMaxDynamicSharedMemorySize, STAGE_TILE_PW * 6));
#endif
    return MIRA_OK;
}
template <class FS> static int synth_scalars(size_t n, uint64_t index0, uint64_t seed, int kind, void *d_out) {
    LAUNCH(k_synth_scalars<FS>, ceil_div(n, 256), 256, 0, g.stream, (uint64_t)n, index0, seed, kind, reinterpret_cast<unsigned char *>(d_out));
    RT_CHECK(rt_last());
    RT_CHECK(rt_sync(g.stream));
    return MIRA_OK;
}
template <class FB> static int synth_bases(size_t n, uint64_t index0, uint64_t seed, const unsigned char *d_gen, void *d_out) {
    LAUNCH(k_synth_bases<FB>, ceil_div(n, 64), 64, 0, g.stream, (uint64_t)n, index0, seed, d_gen, reinterpret_cast<unsigned char *>(d_out));
    RT_CHECK(rt_last());
    RT_CHECK(rt_sync(g.stream));
    return MIRA_OK;
}
template <class F> static int check_bases(const Bases &bs, const unsigned char *d_b, uint32_t *d_bad) {
    LAUNCH(k_check_on_curve<F>, ceil_div(bs.n, 256), 256, 0, g.stream, (const unsigned char *)bs.d, (uint64_t)bs.n, d_b, d_bad);
    RT_CHECK(rt_last());
    return MIRA_OK;
}
template <class F> static int convert_bases(const void *d_src, void *d_dst, size_t n) {
    if (!n) return MIRA_OK;
    LAUNCH(k_convert_bases<F>, ceil_div(n, 256), 256, 0, g.stream, reinterpret_cast<const unsigned char *>(d_src),
           reinterpret_cast<unsigned char *>(d_dst), (uint64_t)n);
    RT_CHECK(rt_last());
    RT_CHECK(rt_sync(g.stream));
    return MIRA_OK;
}
template <class F> static int export_bases(const Bases &bs, size_t first, size_t n, void *d_out) {
    if (!n) return MIRA_OK;
    LAUNCH(k_export_bases<F>, ceil_div(n, 256), 256, 0, g.stream, reinterpret_cast<const unsigned char *>(bs.d) + first * 64,
           reinterpret_cast<unsigned char *>(d_out), (uint64_t)n);
    RT_CHECK(rt_last());
    RT_CHECK(rt_sync(g.stream));
    return MIRA_OK;
}

// ---- the interleaved key of the GLV split (glv.cuh) -----------------------------------------------
template <class F> static int build_glv(Bases &bs, const void *d_beta_r261) {
    const size_t bytes = (size_t)bs.n * 128;
    void *t = nullptr;
    if (rt_malloc(&t, bytes) != hipSuccess || !t) {
        set_error("device allocation of " + std::to_string(bytes) + " bytes for the endomorphism copy of the key failed");
        return MIRA_E_ALLOC;
    }
    LAUNCH(k_glv_bases<F>, ceil_div(bs.n, 256), 256, 0, g.stream, reinterpret_cast<const unsigned char *>(bs.d), reinterpret_cast<unsigned char *>(t), (uint64_t)bs.n,
           reinterpret_cast<const unsigned char *>(d_beta_r261));
    if (rt_last() != hipSuccess || rt_sync(g.stream) != hipSuccess) { (void)rt_free(t); set_error("building the endomorphism copy of the key failed"); return MIRA_E_NO_DEVICE; }
    bs.glv = t;
    return MIRA_OK;
}

// ---- fixed-base window tables (table_kernels.cuh) -------------------------------------------------
template <class F> static int build_tables(Bases &bs, uint32_t c, uint32_t W) {
    if (bs.tables || bs.n == 0) return MIRA_OK;
    const size_t bytes = (size_t)W * bs.n * 64;
    void *t = nullptr;
    if (rt_malloc(&t, bytes) != hipSuccess || !t) {
        set_error("device allocation of " + std::to_string(bytes) + " bytes for the window tables failed");
        return MIRA_E_ALLOC;
    }
    unsigned char *tb = reinterpret_cast<unsigned char *>(t);
    RT_CHECK(rt_d2d(tb, bs.d, bs.n * 64, g.stream));                        // T_0 = the key itself
    for (uint32_t w = 1; w < W; w++)
        LAUNCH(k_table_step<F>, ceil_div(bs.n, 64), 64, 0, g.stream, (const unsigned char *)(tb + (size_t)(w - 1) * bs.n * 64),
               tb + (size_t)w * bs.n * 64, (uint64_t)bs.n, c);
    RT_CHECK(rt_last());
    RT_CHECK(rt_sync(g.stream));
    bs.tables = t; bs.table_c = c; bs.table_w = W;
    return MIRA_OK;
}

// One MSM over the tables: W signed c-bit digits per scalar (c = 20: 13, c = 22: 12), one set of 2^(c-1)
// buckets.  Returns TABLE_SUMS partial sums (XYZZ, canonical, reference form) whose plain sum is the result.
template <class F, class FS>
static int msm_launch_table(const Bases &bs, size_t first, const void *d_scalars, size_t n, uint64_t *host_sums) {
    int rc;
    const TableCfg tc = table_cfg(bs.table_c);
    const uint32_t TABLE_W = tc.W, TABLE_B = tc.B, TABLE_FINE_BITS = tc.fine_bits;
    const size_t entries = (size_t)n * TABLE_W;
    const uint32_t lanes = 256u * 4u * 3u * 64u, Lmin = 16;
    const uint32_t T = (uint32_t)std::min<uint64_t>(lanes, ceil_div(entries, Lmin));
    const uint32_t m = 8, nchunks = TABLE_B / m;
    if ((rc = g.digits.ensure(entries * 4))) return rc;
    if ((rc = g.counts.ensure(((size_t)TABLE_CB + 1) * 4))) return rc;
    if ((rc = g.coarse_offsets.ensure(((size_t)TABLE_CB + 1) * 4))) return rc;
    if ((rc = g.cursor.ensure(((size_t)TABLE_CB + 1) * 4))) return rc;
    if ((rc = g.block_sums.ensure(1024 * 4))) return rc;
    if ((rc = g.part.ensure(entries * 8 + 8))) return rc;
    if ((rc = g.sorted_idx.ensure(entries * 4 + 8))) return rc;
    if ((rc = g.offsets.ensure(((size_t)TABLE_B + 1) * 4))) return rc;
    if ((rc = g.fine_counts.ensure(((size_t)TABLE_B + 1) * 4))) return rc;
    if ((rc = g.fine_cursor.ensure(((size_t)TABLE_B + 1) * 4))) return rc;
    if ((rc = g.bucket_sums.ensure((size_t)TABLE_B * XYZZ29_BYTES))) return rc;
    if ((rc = g.head_part.ensure((size_t)T * XYZZ29_BYTES))) return rc;
    if ((rc = g.tail_part.ensure((size_t)T * XYZZ29_BYTES))) return rc;
    if ((rc = g.tail_key.ensure((size_t)T * 4))) return rc;
    if ((rc = g.heavy.ensure(64 + ((size_t)T / 2 + 8) * 32 + ((size_t)T / 4 + 8) * 16))) return rc;
    if ((rc = g.heavy_out.ensure(((size_t)T / 2 + 8) * XYZZ29_BYTES))) return rc;
    if ((rc = g.chunks.ensure((size_t)nchunks * XYZZ29_BYTES))) return rc;
    if ((rc = g.window_sums.ensure((size_t)TABLE_SUMS * 128))) return rc;

    hipStream_t st = g.stream;
    uint32_t *heavy_count = reinterpret_cast<uint32_t *>(g.heavy.p);
    uint32_t *plan = heavy_count + 2;
    U4 *heavy_runs = reinterpret_cast<U4 *>(heavy_count + 16);
    U4 *heavy_subs = heavy_runs + ((size_t)T / 4 + 4);
    U4 *heavy_meds = reinterpret_cast<U4 *>(reinterpret_cast<unsigned char *>(g.heavy.p) + 64 + ((size_t)T / 2 + 8) * 32);
    // level-1 tiles of 32k points: runs of ~64 entries per (workgroup, coarse bin)
    const uint32_t tile = 32768, ntiles = ceil_div(n, tile);
    tm_begin();
    uint32_t *no_u32 = nullptr;
    unsigned char *no_u8 = nullptr;
    RT_CHECK(rt_memset(g.counts.p, 0, ((size_t)TABLE_CB + 1) * 4, st));
    tm_mark("memset");
    LAUNCH(k_digits32<FS>, ceil_div(n, 256), 256, 0, st, reinterpret_cast<const unsigned char *>(d_scalars), (uint32_t)n, tc.c, tc.W,
           reinterpret_cast<int32_t *>(g.digits.p));
    tm_mark("digits");
    LAUNCH_BARRIER_FLEX(k_thist_coarse, dim3(ntiles, TABLE_W), 512, 0, st, reinterpret_cast<const int32_t *>(g.digits.p), (uint32_t)n, tile, TABLE_FINE_BITS,
                        reinterpret_cast<uint32_t *>(g.counts.p));
    tm_mark("hist");
    LAUNCH_BARRIER(k_scan_a, 1, SCAN_BLOCK, 0, st, reinterpret_cast<const uint32_t *>(g.counts.p), TABLE_CB, reinterpret_cast<uint32_t *>(g.block_sums.p));
    LAUNCH_BARRIER(k_scan_b, 1, 1024, 0, st, reinterpret_cast<uint32_t *>(g.block_sums.p), 1u);
    LAUNCH_BARRIER(k_scan_c<SCAN_BLOCK_SUMS>, 1, SCAN_BLOCK, 0, st, reinterpret_cast<const uint32_t *>(g.counts.p), TABLE_CB,
                   reinterpret_cast<const uint32_t *>(g.block_sums.p), reinterpret_cast<uint32_t *>(g.coarse_offsets.p),
                   reinterpret_cast<uint32_t *>(g.cursor.p), no_u32, 0u, no_u32, 0u, 0u, no_u32, no_u8);
    tm_mark("scan");
    // level 1 by coarse bin (top 9 bits), bucket counts from its output, scan, level 2 by bucket
    const uint32_t *coarse_total = reinterpret_cast<const uint32_t *>(g.coarse_offsets.p) + TABLE_CB;
    LAUNCH_BARRIER((k_stage1<int32_t, true>), dim3(ceil_div(n, STAGE_TILE), TABLE_W), 1024, (size_t)STAGE_TILE * 8, st,
                        reinterpret_cast<const int32_t *>(g.digits.p), (uint32_t)n, TABLE_B, TABLE_FINE_BITS, TABLE_CB, (uint32_t)bs.n, (uint32_t)first, TABLE_W,
                        reinterpret_cast<uint32_t *>(g.cursor.p), reinterpret_cast<U2 *>(g.part.p));
    tm_mark("sort_level1");
    RT_CHECK(rt_memset(g.fine_counts.p, 0, ((size_t)TABLE_B + 1) * 4, st));
    LAUNCH_BARRIER_FLEX(k_stage2_count, ceil_div(entries, STAGE_TILE), 1024, 0, st, reinterpret_cast<const U2 *>(g.part.p), coarse_total, TABLE_FINE_BITS,
                        reinterpret_cast<uint32_t *>(g.fine_counts.p));
    const uint32_t fscan = ceil_div(TABLE_B, SCAN_TILE);
    LAUNCH_BARRIER(k_scan_a, fscan, SCAN_BLOCK, 0, st, reinterpret_cast<const uint32_t *>(g.fine_counts.p), TABLE_B, reinterpret_cast<uint32_t *>(g.block_sums.p));
    LAUNCH_BARRIER(k_scan_b, 1, 1024, 0, st, reinterpret_cast<uint32_t *>(g.block_sums.p), fscan);
    LAUNCH_BARRIER(k_scan_c<SCAN_BLOCK_SUMS>, fscan, SCAN_BLOCK, 0, st, reinterpret_cast<const uint32_t *>(g.fine_counts.p), TABLE_B,
                   reinterpret_cast<const uint32_t *>(g.block_sums.p), reinterpret_cast<uint32_t *>(g.offsets.p), reinterpret_cast<uint32_t *>(g.fine_cursor.p),
                   no_u32, 0u, plan, lanes, Lmin, heavy_count, reinterpret_cast<unsigned char *>(g.bucket_sums.p));
    tm_mark("bucket_count_scan");
    LAUNCH_BARRIER((k_stage2<STAGE_TILE, STAGE_MAX_KEYS2>), ceil_div(entries, STAGE_TILE), 1024, (size_t)STAGE_TILE * 6, st, reinterpret_cast<const U2 *>(g.part.p), coarse_total,
                        TABLE_FINE_BITS, reinterpret_cast<uint32_t *>(g.fine_cursor.p), reinterpret_cast<uint32_t *>(g.sorted_idx.p));
    tm_mark("sort_level2");
    LAUNCH((k_accumulate<F, false>), ceil_div(T, 128), 128, 0, st, reinterpret_cast<const uint32_t *>(g.sorted_idx.p),
           reinterpret_cast<const uint32_t *>(g.offsets.p), TABLE_B, reinterpret_cast<const unsigned char *>(bs.tables), (const uint32_t *)plan,
           reinterpret_cast<unsigned char *>(g.bucket_sums.p), reinterpret_cast<unsigned char *>(g.head_part.p),
           reinterpret_cast<unsigned char *>(g.tail_part.p), reinterpret_cast<uint32_t *>(g.tail_key.p), heavy_count, heavy_runs, heavy_subs, heavy_meds);
    tm_mark("accumulate");
    LAUNCH_BARRIER(k_fixup_all<F>, FIXUP_HEAVY_BLOCKS + ceil_div(T, HEAVY_BLOCK_A / 4), HEAVY_BLOCK_A, 0, st, FIXUP_HEAVY_BLOCKS, (const uint32_t *)plan,
                   reinterpret_cast<const uint32_t *>(g.offsets.p), reinterpret_cast<const unsigned char *>(g.head_part.p), reinterpret_cast<const unsigned char *>(g.tail_part.p),
                   reinterpret_cast<const uint32_t *>(g.tail_key.p), reinterpret_cast<unsigned char *>(g.bucket_sums.p), 0u,
                   (const uint32_t *)heavy_count, (const U4 *)heavy_subs, (const U4 *)heavy_runs, reinterpret_cast<unsigned char *>(g.heavy_out.p), (const U4 *)heavy_meds);
    LAUNCH_BARRIER(k_fixup_heavy_b<F>, (FIXUP_HEAVY_GRID + 3) / 4, HEAVY_BLOCK_B, 0, st, (const uint32_t *)heavy_count, (const U4 *)heavy_runs,
                   reinterpret_cast<const unsigned char *>(g.heavy_out.p), reinterpret_cast<const unsigned char *>(g.tail_part.p),
                   reinterpret_cast<unsigned char *>(g.bucket_sums.p));
    tm_mark("fixup");
    LAUNCH((k_reduce_chunks<F, false>), ceil_div(nchunks, 256), 256, 0, st, reinterpret_cast<const unsigned char *>(g.bucket_sums.p),
           TABLE_B, m, 1u, reinterpret_cast<unsigned char *>(g.chunks.p));
    tm_mark("reduce_chunks");
    LAUNCH_BARRIER((k_window_sum<F, false>), TABLE_SUMS, WSUM_BLOCK, (size_t)WSUM_BLOCK * XYZZ29_BYTES, st,
                   reinterpret_cast<const unsigned char *>(g.chunks.p), nchunks / TABLE_SUMS, reinterpret_cast<unsigned char *>(g.window_sums.p), (const uint32_t *)nullptr);
    tm_mark("window_sum");
    RT_CHECK(rt_last());
    if (g.windows_dst) RT_CHECK(rt_d2d(g.windows_dst, g.window_sums.p, (size_t)TABLE_SUMS * 128, st));
    else RT_CHECK(rt_d2h(host_sums, g.window_sums.p, (size_t)TABLE_SUMS * 128, st));
    RT_CHECK(rt_sync(st));
    tm_end();
    return MIRA_OK;
}

// ---- commitment-key cache file <-> HBM ----------------------------------------------------
// The file is the raw `[C]` slice save_to_file writes (src/commitment.rs:96-101): 64 bytes per point, reference
// layout.  load_from_file (:110-127) reads it whole into host memory and load_or_setup_cache (:145-154) then walks
// it with is_on_curve; here it goes straight to HBM in chunks: chunk i + 1 is read from the file (several threads,
// pread) into the other of two pinned buffers and copied while chunk i is converted to the resident layout and
// checked against the curve equation on the work stream.
static constexpr size_t KEYFILE_CHUNK_POINTS = (size_t)1 << 20;     // 64 MiB
static inline int read_exact_parallel(int fd, unsigned char *dst, size_t bytes, size_t file_off) {
    const size_t nthreads = std::max<size_t>(1, std::min<size_t>(8, std::thread::hardware_concurrency()));
    const size_t slice = (bytes + nthreads - 1) / nthreads;
    std::vector<int> fail(nthreads, 0);
    std::vector<std::thread> th;
    auto work = [&](size_t t) {
        size_t lo = std::min(bytes, t * slice), hi = std::min(bytes, lo + slice);
        while (lo < hi) {
            const ssize_t r = pread(fd, dst + lo, hi - lo, (off_t)(file_off + lo));
            if (r <= 0) { fail[t] = 1; return; }                  // error or end of file: read_exact's "failed to fill whole buffer"
            lo += (size_t)r;
        }
    };
    for (size_t t = 1; t < nthreads; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    for (int f : fail) if (f) return 1;
    return 0;
}
template <class F> static int load_bases_file(Bases &b, int fd, bool validate, const unsigned char *d_curve_b, uint32_t *d_bad) {
    unsigned char *pinned[2] = {nullptr, nullptr};
    const size_t chunk_bytes = std::min(b.n, KEYFILE_CHUNK_POINTS) * 64;
    auto body = [&]() -> int {
        for (int k = 0; k < 2; k++) RT_CHECK(rt_host_alloc(reinterpret_cast<void **>(&pinned[k]), chunk_bytes));
#ifndef MIRA_CPU_EMU
        if (!g.copy_stream) RT_CHECK(hipStreamCreateWithFlags(&g.copy_stream, hipStreamNonBlocking));
        while (g.copy_events.size() < 2) {
            hipEvent_t e;
            RT_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            g.copy_events.push_back(e);
        }
#endif
        hipStream_t st = g.stream, cs = g.copy_stream ? g.copy_stream : st;
        if (validate) RT_CHECK(rt_memset(d_bad, 0, 4, st));
        unsigned char *dst = reinterpret_cast<unsigned char *>(b.d);
        size_t i = 0;
        for (size_t off = 0; off < b.n; off += KEYFILE_CHUNK_POINTS, i++) {
            const size_t cnt = std::min(KEYFILE_CHUNK_POINTS, b.n - off);
            unsigned char *buf = pinned[i & 1];
#ifndef MIRA_CPU_EMU
            if (i >= 2) RT_CHECK(rt_event_sync(g.copy_events[i & 1]));     // the copy out of this buffer two chunks ago is done
#endif
            if (read_exact_parallel(fd, buf, cnt * 64, off * 64)) { set_error("failed to fill whole buffer"); return MIRA_E_IO; }
            RT_CHECK(rt_h2d(dst + off * 64, buf, cnt * 64, cs));
#ifndef MIRA_CPU_EMU
            if (cs != st) RT_CHECK(rt_stream_wait(st, cs, g.copy_events[i & 1]));
            else RT_CHECK(hipEventRecord(g.copy_events[i & 1], cs));
#endif
            LAUNCH(k_convert_bases<F>, ceil_div(cnt, 256), 256, 0, st, (const unsigned char *)(dst + off * 64), dst + off * 64, (uint64_t)cnt);
            if (validate) LAUNCH(k_check_on_curve<F>, ceil_div(cnt, 256), 256, 0, st, (const unsigned char *)(dst + off * 64), (uint64_t)cnt, d_curve_b, d_bad);
        }
        RT_CHECK(rt_last());
        RT_CHECK(rt_sync(cs));
        RT_CHECK(rt_sync(st));
        return MIRA_OK;
    };
    const int rc = body();
    if (rc != MIRA_OK) { if (g.copy_stream) (void)rt_sync(g.copy_stream); (void)rt_sync(g.stream); }
    for (int k = 0; k < 2; k++) if (pinned[k]) (void)rt_host_free(pinned[k]);
    return rc;
}
template <class F> static int save_bases_file(const Bases &b, int fd) {
    int rc;
    const size_t chunk = std::min(b.n, KEYFILE_CHUNK_POINTS);
    if (!chunk) return MIRA_OK;
    if ((rc = g.scalars_stage.ensure(chunk * 64))) return rc;
    unsigned char *pinned = nullptr;
    RT_CHECK(rt_host_alloc(reinterpret_cast<void **>(&pinned), chunk * 64));
    rc = MIRA_OK;
    for (size_t off = 0; off < b.n && rc == MIRA_OK; off += chunk) {
        const size_t cnt = std::min(chunk, b.n - off);
        LAUNCH(k_export_bases<F>, ceil_div(cnt, 256), 256, 0, g.stream, reinterpret_cast<const unsigned char *>(b.d) + off * 64,
               reinterpret_cast<unsigned char *>(g.scalars_stage.p), (uint64_t)cnt);
        if (rt_d2h(pinned, g.scalars_stage.p, cnt * 64, g.stream) != hipSuccess || rt_sync(g.stream) != hipSuccess) { set_error("device to host copy failed"); rc = MIRA_E_NO_DEVICE; break; }
        for (size_t done = 0; done < cnt * 64;) {
            const ssize_t w = write(fd, pinned + done, cnt * 64 - done);
            if (w <= 0) { set_error(std::string("write failed: ") + strerror(errno)); rc = MIRA_E_IO; break; }
            done += (size_t)w;
        }
    }
    (void)rt_host_free(pinned);
    return rc;
}

