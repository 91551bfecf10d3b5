// Pippenger MSM kernels for gfx950.  Replaces halo2_proofs::arithmetic::best_multiexp as
// called by CommitmentKey::commit (reference src/commitment.rs:78-87).
//
// Pipeline (all on one HIP stream, no host round trip until the W window sums come back):
//   k_digits      scalar (Montgomery) -> canonical integer -> W signed c-bit digits (int16)
//   k_hist        per (window, point-tile) workgroup: bucket histogram staged in LDS
//                 (2^(c-1) counters = 128 KiB at c = 16), flushed with coalesced atomics
//   k_scan_*      exclusive scan of the W * 2^(c-1) counters
//   k_scatter     same tiling; the workgroup claims a contiguous range per bucket (one global
//                 atomic per non-empty LDS bin) and places (point index | sign), 4 bytes per entry.
//                 (A two-level coarse/fine sort was built and measured: 1.59 ms against 1.30 ms
//                 for this single level at 2^22 -- it moves 3x the bytes and L2 write-combining
//                 does not make up for it.)
//   k_accumulate  (9 x 29-bit limb field, curve29.cuh) every lane owns exactly L consecutive sorted entries (perfect balance however
//                 skewed the scalars are); complete bucket runs go straight to bucket_sums, runs
//                 cut by a lane boundary leave a head/tail partial
//   k_fixup*      joins the partials: short chains by one lane, long chains (heavy buckets such as
//                 "all witness cells equal 1") by two levels of workgroup LDS trees
//   k_reduce_chunks / k_window_sum   sum_b (b+1) * S_b per window by chunked running sums
// Host: Horner over the W window sums and the single inversion of to_affine().
#pragma once
#include "curve.cuh"
#include "curve29.cuh"
#include "quad29.cuh"
#include "glv.cuh"

static constexpr uint32_t KEY_NONE = 0xFFFFFFFFu;
// Chains longer than this go to k_fixup_heavy (a quad adds a link in ~2.2 us plus the load of the partial).  Raising it
// to 16 was measured: commits whose longest chains are 7 .. 16 links gain, but wherever the heavy stages run anyway the
// medium chains then sit in k_fixup for 100 us (2^13 pairs under 5-bit windows: 57 -> 158 us) -- the threshold stays.
static constexpr int HEAVY_SPAN = 6;
// MEDIUM runs (HEAVY_SPAN < partials <= MEDIUM_SPAN) go where they are cheaper, decided on the device when their number is
// known: a few of them are sub-jobs of the heavy section (several quads and a tree each: 0.04 against 0.07 ms for the fix-up of
// a witness-like 2^17-pair commit), thousands of them -- the 32-bit values of a 1.8 M-scalar witness vector fill 4 096 buckets
// with 7 - 8 partials each -- would flood it with sub-jobs that share workgroups with the really heavy ones, and are summed as
// plain chains by the quads of the light section instead (that commit's fix-up 0.18 -> 0.10 ms; profiles/r04_f_heavy_span.txt).
static constexpr int MEDIUM_SPAN = 12;
#ifdef MIRA_CPU_EMU
static constexpr uint32_t MEDIUM_AS_CHAINS_FROM = 8;     // (the emulation's sizes reach both placements with this)
#else
static constexpr uint32_t MEDIUM_AS_CHAINS_FROM = 1024;
#endif
static constexpr int WSUM_BLOCK = 512;       // k_window_sum: 128 quads
static constexpr uint32_t HEAVY_SUB = 64;    // partials per stage-A sub-job of a heavy run: one wave (16 quads x 4 partials, then a 4-level tree)

// ------------------------------------------------------------------------------------------
// (also clears the ncounts bucket counters the histogram that follows adds into: one launch less)
// Planning statistics (hist != null): the histogram of the bit lengths of the canonical scalars
// (hist[0] = zeros, hist[b] = scalars with top set bit b - 1), summed over the batch.  Witness vectors
// are mostly zeros and small values (SURVEY.md 7 "bucket contention"; src/util.rs:189-193 zero-pads
// every column), so the number of non-zero digits -- and with it the best window width -- depends on
// the data.  Counted here, where every scalar is already canonical in registers (a pre-pass of its
// own read the scalars a second time and cost a launch, a memset and a copy: 30-50 us per commit).
// hist_clear = the OTHER of the two histogram buffers, zeroed for the next commit.
// GLV (glv.cuh): every scalar becomes TWO half-length ones, columns 2 i and 2 i + 1 of a digit matrix of 2 n columns.
template <class FS, bool GLV = false>
KERNEL void k_digits(const unsigned char *__restrict__ scalars, uint32_t n, uint64_t stride, uint32_t c, uint32_t W,
                     int16_t *__restrict__ digits, uint32_t *__restrict__ counts, uint32_t ncounts,
                     uint32_t *__restrict__ hist, uint32_t *__restrict__ hist_clear) {
    __shared__ uint32_t bins[256];
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t k = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; k < ncounts; k += gridDim.x * gridDim.y * blockDim.x) counts[k] = 0;
    // a SAMPLE: every eighth block counts, eight-fold (planning wants proportions; every block flushing
    // its two or three hot bins to the same global counters serialises in L2 -- 50 us at 2^17 scalars)
    const bool sampled = hist && (gridDim.x < 8 || (blockIdx.x & 7u) == 0);
    const uint32_t weight = gridDim.x < 8 ? 1u : 8u;
    if (hist) {                                          // uniform across the grid
        for (uint32_t b = threadIdx.x; b < 256; b += blockDim.x) bins[b] = 0;
        if (blockIdx.x == 0 && blockIdx.y == 0)
            for (uint32_t b = threadIdx.x; b < 256; b += blockDim.x) hist_clear[b] = 0;
        __syncthreads();
    }
    if (i < n) {
        const uint32_t b = blockIdx.y;                   // MSM of the batch: its windows are b*W .. b*W + W-1
        Fe<FS> s = fe_from_mont(fe_load<FS>(scalars + ((size_t)b * stride + i) * 32));
        if constexpr (GLV) {
            uint32_t h[2][5];
            bool neg[2];
            glv_split<FS>(s.l, h[0], h[1], neg[0], neg[1]);
            int16_t *dg = digits + (size_t)b * W * (2 * (size_t)n);
            const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
#pragma unroll
            for (int e = 0; e < 2; e++) {
                if (sampled) {
                    uint32_t len = 0;
#pragma unroll
                    for (int k = 0; k < 5; k++)
                        if (h[e][k]) len = 32u * k + (32u - (uint32_t)__builtin_clz(h[e][k]));
                    atomicAdd(&bins[len], 1u);
                }
                uint32_t carry = 0;
                for (uint32_t w = 0; w < W; w++) {
                    const uint32_t raw = (h[e][0] & mask) + carry;
#pragma unroll
                    for (int k = 0; k < 4; k++) h[e][k] = (h[e][k] >> c) | (h[e][k + 1] << (32 - c));
                    h[e][4] >>= c;
                    int32_t d;
                    // raw = 2^(c-1) may be written as -2^(c-1) with a carry or as +2^(c-1) without: whichever leaves the STORED digit
                    // (negated for a negative half) at -2^(c-1), the one of the two an int16 holds at c = 16.  The last window keeps
                    // what it holds (glv.cuh: magnitudes < 2^126.13 leave it below 2^(c-1) even under c W = 128).
                    if ((raw > half || (raw == half && !neg[e])) && w + 1 < W) { d = (int32_t)raw - (int32_t)(1u << c); carry = 1; }
                    else { d = (int32_t)raw; carry = 0; }
                    dg[(size_t)w * (2 * (size_t)n) + 2 * (size_t)i + e] = (int16_t)(neg[e] ? -d : d);
                }
            }
        } else {
        if (sampled) {
            uint32_t len = 0;
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (s.l[k]) len = 32u * k + (32u - (uint32_t)__builtin_clz(s.l[k]));
            atomicAdd(&bins[len > 255 ? 255 : len], 1u);
        }
        int16_t *dg = digits + (size_t)b * W * n;
        const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
        uint32_t carry = 0;
        for (uint32_t w = 0; w < W; w++) {
            uint32_t raw = (s.l[0] & mask) + carry;
#pragma unroll
            for (int k = 0; k < 7; k++) s.l[k] = (s.l[k] >> c) | (s.l[k + 1] << (32 - c));   // c in [1,16]
            s.l[7] >>= c;
            int32_t d;
            if (raw >= half) { d = (int32_t)raw - (int32_t)(1u << c); carry = 1; }
            else { d = (int32_t)raw; carry = 0; }
            dg[(size_t)w * n + i] = (int16_t)d;
        }
        }
    }
    if (sampled) {                                       // uniform across the block
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < 256; b += blockDim.x)
            if (bins[b]) atomicAdd(&hist[b], bins[b] * weight);
    }
}

// ------------------------------------------------------------------------------------------
// grid = (ntiles, W_total), dynamic LDS = B * 4 bytes.  Window w counts into bucket set w / wgroup:
// wgroup = 1, every window has its own buckets; wgroup = W, the W windows of one MSM share a set
// (fixed-base tables, msm_host.cuh).
KERNEL void k_hist(const int16_t *__restrict__ digits, uint32_t n, uint32_t B, uint32_t tile,
                   uint32_t *__restrict__ counts, uint32_t wgroup) {
    DYN_SHARED(uint32_t, bins);
    const uint32_t w = blockIdx.y;
    for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) bins[b] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * tile, end = (base + tile < n) ? base + tile : n;
    const int16_t *dw = digits + (size_t)w * n;
    auto count = [&](int32_t d) { if (d != 0) atomicAdd(&bins[(uint32_t)(d < 0 ? -d : d) - 1], 1u); };
    // four digits per 8-byte load where the window's digits are 8-byte aligned (n a multiple of 4; tiles are
    // multiples of 1024): a quarter of the load instructions (2-byte loads: 1.4 TB/s)
    uint32_t i0 = base;
    if (((reinterpret_cast<uintptr_t>(dw + base)) & 7u) == 0) {
        const uint32_t groups = (end - base) / 4;
        const U2 *dv = reinterpret_cast<const U2 *>(dw + base);
        for (uint32_t q = threadIdx.x; q < groups; q += blockDim.x) {
            const U2 v = dv[q];
            count((int16_t)(v.x & 0xFFFFu)); count((int16_t)(v.x >> 16)); count((int16_t)(v.y & 0xFFFFu)); count((int16_t)(v.y >> 16));
        }
        i0 = base + groups * 4;
    }
    for (uint32_t i = i0 + threadIdx.x; i < end; i += blockDim.x) count(dw[i]);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) {
        uint32_t cnt = bins[b];
        if (cnt) atomicAdd(&counts[(size_t)(w / wgroup) * B + b], cnt);
    }
}

// ------------------------------------------------------------------------------------------
// Exclusive scan of counts[0..NB) in three launches.  SCAN_ITEMS consecutive counters per lane.
static constexpr int SCAN_BLOCK = 256, SCAN_ITEMS = 8, SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

KERNEL void k_scan_a(const uint32_t *__restrict__ counts, uint32_t NB, uint32_t *__restrict__ block_sums) {
    __shared__ uint32_t red[SCAN_BLOCK];
    uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS, s = 0;
    for (int k = 0; k < SCAN_ITEMS; k++)
        if (base + k < NB) s += counts[base + k];
    red[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t st = SCAN_BLOCK / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = red[0];
}
// one workgroup of 1024 lanes; nblocks <= 1024
KERNEL void k_scan_b(uint32_t *__restrict__ block_sums, uint32_t nblocks) {
    __shared__ uint32_t buf[1024];
    uint32_t v = threadIdx.x < nblocks ? block_sums[threadIdx.x] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t add = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
        __syncthreads();
        buf[threadIdx.x] += add;
        __syncthreads();
    }
    if (threadIdx.x < nblocks) block_sums[threadIdx.x] = buf[threadIdx.x] - v;   // exclusive
}
// offsets[NB] = total number of sorted entries; cursor = copy of offsets for k_scatter.
// Optional by-products (null = skip), each of which used to be a launch of its own:
//   cursor1  coarse cursors of the staged sort, cursor1[g] = offsets[g << fine_bits]
//   bucket_sums  the identity marker (ZZ = 0) of every bucket without entries: k_accumulate and the
//            fix-up only write buckets that have some, and the bucket reduction reads them all
//   plan     {L, T} of k_accumulate -- segment length chosen once the number of non-zero digits is
//            known: exactly one segment per resident lane (every SIMD slot busy for the whole kernel
//            and all lanes finishing together), never shorter than min_L -- and heavy_ctr cleared
// SCAN_SOLO: ONE workgroup walks all the tiles itself, carrying the running total -- for small bucket counts
// (a commit of 2^17 pairs under 8-bit windows has 4096 buckets) one launch instead of three.
// SCAN_OWN_PREFIX: every workgroup first sums the counters in front of its tile itself (16-byte loads, all from L2) -- quadratic
// in the number of tiles, so for the bucket counts of the small commits only (<= SCAN_OWN_PREFIX_TILES tiles: 2^16 counters, 128 KiB
// read by the last workgroup); again one launch instead of three, and no block_sums.
enum : int { SCAN_BLOCK_SUMS = 0, SCAN_SOLO = 1, SCAN_OWN_PREFIX = 2 };
template <int MODE = SCAN_BLOCK_SUMS>
KERNEL void k_scan_c(const uint32_t *__restrict__ counts, uint32_t NB, const uint32_t *__restrict__ block_sums,
                     uint32_t *__restrict__ offsets, uint32_t *__restrict__ cursor, uint32_t *__restrict__ cursor1, uint32_t fine_bits,
                     uint32_t *__restrict__ plan, uint32_t resident_lanes, uint32_t min_L, uint32_t *__restrict__ heavy_ctr,
                     unsigned char *__restrict__ bucket_sums) {
    __shared__ uint32_t buf[SCAN_BLOCK], carry_s;
    const uint32_t ntiles = (NB + SCAN_TILE - 1) / SCAN_TILE;
    constexpr bool SOLO = MODE == SCAN_SOLO;
    if (SOLO) {
        if (threadIdx.x == 0) carry_s = 0;
        __syncthreads();
    }
    if (MODE == SCAN_OWN_PREFIX) {
        const U4 *cv = reinterpret_cast<const U4 *>(counts);
        uint32_t pre = 0;
        for (uint32_t q = threadIdx.x; q < blockIdx.x * (SCAN_TILE / 4); q += SCAN_BLOCK) {
            const U4 v = cv[q];
            pre += v.x + v.y + v.z + v.w;
        }
        buf[threadIdx.x] = pre;
        __syncthreads();
        for (uint32_t st = SCAN_BLOCK / 2; st > 0; st >>= 1) {
            if (threadIdx.x < st) buf[threadIdx.x] += buf[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) carry_s = buf[0];
        __syncthreads();
    }
    for (uint32_t tile = SOLO ? 0 : blockIdx.x; tile < (SOLO ? ntiles : blockIdx.x + 1); tile++) {
        uint32_t base = tile * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
        uint32_t loc[SCAN_ITEMS], s = 0;
        for (int k = 0; k < SCAN_ITEMS; k++) {
            loc[k] = (base + k < NB) ? counts[base + k] : 0;
            s += loc[k];
        }
        buf[threadIdx.x] = s;
        __syncthreads();
        for (uint32_t off = 1; off < SCAN_BLOCK; off <<= 1) {
            uint32_t add = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
            __syncthreads();
            buf[threadIdx.x] += add;
            __syncthreads();
        }
        uint32_t run = (MODE != SCAN_BLOCK_SUMS ? carry_s : block_sums[tile]) + buf[threadIdx.x] - s;
        for (int k = 0; k < SCAN_ITEMS; k++) {
            if (base + k < NB) {
                offsets[base + k] = run; cursor[base + k] = run;
                if (cursor1 && ((base + k) & ((1u << fine_bits) - 1u)) == 0) cursor1[(base + k) >> fine_bits] = run;
                if (bucket_sums && loc[k] == 0) {
                    uint32_t *zz = reinterpret_cast<uint32_t *>(bucket_sums + (size_t)(base + k) * XYZZ29_BYTES) + 18;
#pragma unroll
                    for (int q = 0; q < 9; q++) zz[q] = 0;
                }
            }
            run += loc[k];
            if (base + k == NB - 1) {
                offsets[NB] = run;
                if (plan) {
                    uint32_t L = (uint32_t)(((uint64_t)run + resident_lanes - 1) / resident_lanes);
                    if (L < min_L) L = min_L;
                    plan[0] = L;
                    plan[1] = (uint32_t)(((uint64_t)run + L - 1) / L);
                    heavy_ctr[0] = 0; heavy_ctr[1] = 0; heavy_ctr[4] = 0; heavy_ctr[5] = 0;
                }
            }
        }
        if (SOLO) {
            __syncthreads();
            if (threadIdx.x == SCAN_BLOCK - 1) carry_s += buf[threadIdx.x];
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------
// grid = (ntiles, W_total), dynamic LDS = B * 4 bytes.  Same tiling and bucket sets as k_hist.  Entries name point
// (w % wgroup) * idx_stride + idx_first + i: the point itself (0, 0) or its copy in window table w % wgroup.
KERNEL void k_scatter(const int16_t *__restrict__ digits, uint32_t n, uint32_t B, uint32_t tile,
                      uint32_t *__restrict__ cursor, uint32_t *__restrict__ sorted, uint32_t wgroup, uint32_t idx_stride, uint32_t idx_first) {
    DYN_SHARED(uint32_t, bins);
    const uint32_t w = blockIdx.y;
    for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) bins[b] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * tile, end = (base + tile < n) ? base + tile : n;
    const int16_t *dw = digits + (size_t)w * n;
    for (uint32_t i = base + threadIdx.x; i < end; i += blockDim.x) {
        int32_t d = dw[i];
        if (d != 0) atomicAdd(&bins[(uint32_t)(d < 0 ? -d : d) - 1], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) {
        uint32_t cnt = bins[b];
        if (cnt) bins[b] = atomicAdd(&cursor[(size_t)(w / wgroup) * B + b], cnt);   // claimed range start
    }
    __syncthreads();
    for (uint32_t i = base + threadIdx.x; i < end; i += blockDim.x) {
        int32_t d = dw[i];
        if (d != 0) {
            uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1;
            uint32_t pos = atomicAdd(&bins[b], 1u);
            sorted[pos] = ((w % wgroup) * idx_stride + idx_first + i) | (d < 0 ? 0x80000000u : 0u);   // 4 bytes per entry; the bucket is implied by the position
        }
    }
}

// ------------------------------------------------------------------------------------------
// Key registration: reference layout (x * 2^256, y * 2^256; src/commitment.rs:26-29) -> the
// engine's resident layout: canonical saturated limbs of x * 2^261, y * 2^261 (field29.cuh).
// In place when src == dst.
template <class F>
KERNEL void k_convert_bases(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst, uint64_t n) {
    using S = typename F::Sat;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<S> x = fe_load<S>(src + i * 64), y = fe_load<S>(src + i * 64 + 32);
    if (!(fe_is_zero(x) && fe_is_zero(y))) {
        x = reduce_once(f29_pack(f29_from_r256<F>(x)));
        y = reduce_once(f29_pack(f29_from_r256<F>(y)));
    }
    fe_store(dst + i * 64, x);
    fe_store(dst + i * 64 + 32, y);
}

// The inverse of k_convert_bases: resident layout -> reference layout (key export / cache file).
template <class F>
KERNEL void k_export_bases(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst, uint64_t n) {
    using S = typename F::Sat;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<S> x = fe_load<S>(src + i * 64), y = fe_load<S>(src + i * 64 + 32);
    if (!(fe_is_zero(x) && fe_is_zero(y))) {
        x = f29_to_r256(f29_unpack_canonical<F>(x));
        y = f29_to_r256(f29_unpack_canonical<F>(y));
    }
    fe_store(dst + i * 64, x);
    fe_store(dst + i * 64 + 32, y);
}

// bad_count += 1 for every resident base off the curve y^2 = x^3 + b (b given as b * 2^261,
// canonical saturated).  The reference validates a cached key this way (src/commitment.rs:145-154).
template <class F>
KERNEL void k_check_on_curve(const unsigned char *__restrict__ bases, uint64_t n, const unsigned char *__restrict__ b_r261,
                             uint32_t *__restrict__ bad_count) {
    using S = typename F::Sat;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Aff29<F> p = aff29_load<F>(bases + i * 64, false);
    if (aff29_is_identity(p)) return;
    Fe29<F> lhs = f29_sqr(p.y);
    Fe29<F> rhs = f29_add(f29_mul(f29_sqr(p.x), p.x), f29_unpack_canonical<F>(fe_load<S>(b_r261)));
    if (!f29_is_zero_mod_p<6>(f29_sub<4>(lhs, rhs))) atomicAdd(bad_count, 1u);
}

// ------------------------------------------------------------------------------------------
// largest b in [lo, NB) with offsets[b] <= pos  (offsets non-decreasing, offsets[NB] = total > pos)
DEV uint32_t bucket_of(const uint32_t *__restrict__ offsets, uint32_t lo, uint32_t NB, uint32_t pos) {
    uint32_t hi = NB;           // invariant: offsets[lo] <= pos < offsets[hi]
    while (hi - lo > 1) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (offsets[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}

// ADD: the bucket sums of the earlier point chunks of the same MSM are already in bucket_sums (host
// scalars arrive chunk by chunk, msm_host.cuh).  The lane in which a run STARTS takes the bucket's
// current value as the run's initial accumulator instead of the identity -- exactly one lane per
// run does, so every earlier sum is counted once, at the price of one 144-byte load per run and no
// extra registers; complete runs, partials and the fix-up then work as for a single chunk.
template <bool ADD, class F> DEV Xyzz29<F> run_start(const unsigned char *bucket_sums, uint32_t bucket) {
    if constexpr (ADD) return xyzz29_load<F>(bucket_sums + (size_t)bucket * XYZZ29_BYTES);
    else return xyzz29_identity<F>();
}
// Timing probe (tools/build_probe_variants.sh ... -DMSM_PROBE_STAMPS; never in the product build): lane 0 of every wave of
// k_accumulate writes its hardware id and s_memrealtime (100 MHz) at its start and at its end.
#ifdef MSM_PROBE_STAMPS
static __device__ uint64_t g_acc_stamps[4096 * 3];
#endif
template <class F, bool ADD>
KERNEL void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(3, 3))) k_accumulate(const uint32_t *__restrict__ sorted, const uint32_t *__restrict__ offsets, uint32_t NB,
                         const unsigned char *__restrict__ bases,
                         const uint32_t *__restrict__ plan, unsigned char *__restrict__ bucket_sums,
                         unsigned char *__restrict__ head_part, unsigned char *__restrict__ tail_part, uint32_t *__restrict__ tail_key,
                         uint32_t *__restrict__ heavy_ctr, U4 *__restrict__ runs, U4 *__restrict__ subs, U4 *__restrict__ meds) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
#ifdef MSM_PROBE_STAMPS
    const uint64_t probe_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const uint32_t total = offsets[NB];
    const uint32_t L = plan[0];
    const uint64_t start64 = (uint64_t)t * L;
    if (start64 >= total) return;
    const uint32_t start = (uint32_t)start64;
    const uint32_t end = (total - start > L) ? start + L : total;
    // The bucket of an entry is implied by its position: bucket b owns sorted[offsets[b] ..
    // offsets[b+1]).  Find the bucket holding `start` once, then walk the boundaries; the boundary
    // after next is always already loaded.  (offsets[NB] = total, so indices are clamped to NB.  The
    // boundary load of the run-end branch is waited for at the end of that branch, together with the
    // base gathers just issued; fetching it one run ahead in every iteration instead was measured
    // slower, 5.55 against 5.32 ms.)
    uint32_t cur = bucket_of(offsets, 0, NB, start);
    uint32_t run_end = offsets[cur + 1];
    uint32_t next_end = offsets[cur + 2 <= NB ? cur + 2 : NB];
    const bool cont_prev = offsets[cur] < start;
    bool first = true;
    Xyzz29<F> acc = cont_prev ? xyzz29_identity<F>() : run_start<ADD, F>(bucket_sums, cur);
    // two-deep software pipeline: the entry two ahead and the base one ahead are in flight while
    // the current mixed add (about 9k issue cycles per wave) runs
    uint32_t ent0 = sorted[start];
    uint32_t ent1 = (start + 1 < end) ? sorted[start + 1] : ent0;
    const U4 *bp = reinterpret_cast<const U4 *>(bases + (size_t)(ent0 & 0x7FFFFFFFu) * 64);
    U4 r0 = bp[0], r1 = bp[1], r2 = bp[2], r3 = bp[3];
    for (uint32_t j = start; j < end; j++) {
        const uint32_t e = ent0;
        const U4 c0 = r0, c1 = r1, c2 = r2, c3 = r3;
        // Unconditional loads (ent1 and the clamped index are always valid; the last iterations fetch a
        // point again that nobody uses): inside `if (j + 1 < end)` the compiler merged the loaded
        // registers with the old ones right behind the loads -- s_waitcnt vmcnt at the top of every
        // iteration, the whole gather latency exposed (SQ_WAIT_ANY 15 % of the wave cycles).
        bp = reinterpret_cast<const U4 *>(bases + (size_t)(ent1 & 0x7FFFFFFFu) * 64);
        r0 = bp[0]; r1 = bp[1]; r2 = bp[2]; r3 = bp[3];
        ent0 = ent1;
        ent1 = sorted[(j + 2 < end) ? j + 2 : end - 1];
        if (j == run_end) {                                  // the run of `cur` is complete
            if (first && cont_prev) xyzz29_store(head_part + (size_t)t * XYZZ29_BYTES, acc);
            else xyzz29_store(bucket_sums + (size_t)cur * XYZZ29_BYTES, acc);
            first = false;
            cur++;
            run_end = next_end;
            if (run_end <= j) {                              // empty buckets in between: search
                cur = bucket_of(offsets, cur, NB, j);
                run_end = offsets[cur + 1];
            }
            next_end = offsets[cur + 2 <= NB ? cur + 2 : NB];
            acc = run_start<ADD, F>(bucket_sums, cur);
        }
        xyzz29_add_affine(acc, aff29_from_raw<F>(c0, c1, c2, c3, (e >> 31) != 0));
    }
    const bool cont_next = end < run_end;                    // the run continues in the next lane
    const bool is_head = first && cont_prev;
    if (is_head) xyzz29_store(head_part + (size_t)t * XYZZ29_BYTES, acc);
    else if (cont_next) xyzz29_store(tail_part + (size_t)t * XYZZ29_BYTES, acc);
    else xyzz29_store(bucket_sums + (size_t)cur * XYZZ29_BYTES, acc);
    const bool holds_tail = !is_head && cont_next;
    tail_key[t] = holds_tail ? cur : KEY_NONE;               // every segment of the plan writes its key: no clearing pass
    // A run cut into more than HEAVY_SPAN partials (a heavy bucket) is entered into the heavy list HERE, by the one lane that holds
    // its tail partial: the list is complete when this kernel ends, so the sub-jobs of the heavy runs and the short chains of all
    // the others are summed side by side in ONE launch (k_fixup_all) instead of one after the other (round 3: k_fixup wrote the
    // list, k_fixup_heavy_a followed -- 49 + 42 us of a 0.54 ms commit of 2^17 pairs, both latency-bound on mostly idle SIMDs).
#ifdef MSM_PROBE_STAMPS
    if ((threadIdx.x & 63u) == 0 && (t >> 6) < 4096u) {
        uint32_t hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        g_acc_stamps[(t >> 6) * 3] = hwid; g_acc_stamps[(t >> 6) * 3 + 1] = probe_t0; g_acc_stamps[(t >> 6) * 3 + 2] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    if (holds_tail) {
        const uint32_t span = (run_end - 1) / L - t;          // lanes t + 1 .. t + span hold head partials of this run
        if (span > (uint32_t)HEAVY_SPAN && span <= (uint32_t)MEDIUM_SPAN) {
            meds[atomicAdd(&heavy_ctr[5], 1u)] = U4{t, span, cur, 0};          // a medium run: placed when their number is known (k_fixup_all)
        } else if (span > (uint32_t)MEDIUM_SPAN) {
            const uint32_t nsub = (span + HEAVY_SUB - 1) / HEAVY_SUB;
            const uint32_t h = atomicAdd(&heavy_ctr[0], 1u);
            const uint32_t base = atomicAdd(&heavy_ctr[1], nsub);
            if (nsub > 1) atomicAdd(&heavy_ctr[4], 1u);
            runs[h] = U4{t, span, cur, base};
            for (uint32_t q = 0; q < nsub; q++) {
                const uint32_t first_p = t + 1 + q * HEAVY_SUB;
                const uint32_t cnt = (span - q * HEAVY_SUB < HEAVY_SUB) ? span - q * HEAVY_SUB : HEAVY_SUB;
                subs[base + q] = U4{first_p, cnt, h, 0};
            }
        }
    }
}

template <bool QUAD, class F> DEV void xyzz29_add_sel(Xyzz29<F> &acc, const Xyzz29<F> &q) {
    if constexpr (QUAD) xyzz29_add_quad(acc, q); else xyzz29_add(acc, q);
}
template <bool QUAD, class F> DEV Xyzz29<F> xyzz29_double_sel(const Xyzz29<F> &p) {
    if constexpr (QUAD) return xyzz29_double_quad(p); else return xyzz29_double(p);
}
// The tail kernels below are chains of dependent general additions on few waves.  Every work item
// is owned by a QUAD of lanes (quad29.cuh): the four lanes load the same operands and share the
// multiplications of each addition, which cuts the latency of a link from ~6.8 us to ~2 us.
DEV uint32_t quad_gid() { return (blockIdx.x * blockDim.x + threadIdx.x) >> 2; }

// The short chains: one quad per cut run sums its tail partial and its <= HEAVY_SPAN head partials.  Longer runs (heavy buckets: the
// carry bucket of small witness values, a column of ones, the sparse top window) were entered into the heavy list by k_accumulate
// and are summed by the sub-jobs of the same launch (k_fixup_all below), two LDS trees whatever their length.
// heavy_ctr = {runs, sub-jobs, -, -, runs of more than one sub-job}; runs[h] = {lane, span, bucket, first sub-job};
// subs[s] = {first partial, count, run, -}.
// (One LANE per cut run instead of a quad, for the sizes at which the cut runs fill the SIMDs, was built and measured in round 4:
// 2^17 pairs c = 13 fix-up 104 -> 117 us, c = 12 123 -> 151, c = 16 51 -> 48; 2^22 pairs 59 -> 54 -- the kernel is bound by its
// dependent loads and a lane's 14 serial multiplications are no better hidden than a quad's 4.)
// Many medium runs BESIDE really heavy ones are summed as chains by the light section; alone (8 192 pairs under 8-bit windows:
// thousands of runs of seven partials and nothing heavier) they are quicker as sub-jobs of their own -- 0.040 against 0.065 ms.
DEV bool medium_as_chains(const uint32_t *__restrict__ heavy_ctr) { return heavy_ctr[5] >= MEDIUM_AS_CHAINS_FROM && heavy_ctr[1] != 0; }
// tail partial + its `span` head partials, by one quad; the next partial is requested before the current one is added
template <class F>
DEV void fixup_chain(uint32_t t, uint32_t span, uint32_t key, const unsigned char *__restrict__ head_part, const unsigned char *__restrict__ tail_part,
                     unsigned char *__restrict__ bucket_sums) {
    Xyzz29<F> acc = xyzz29_load<F>(tail_part + (size_t)t * XYZZ29_BYTES);
    Xyzz29<F> nxt = xyzz29_load<F>(head_part + (size_t)(t + 1) * XYZZ29_BYTES);
    for (uint32_t q = 1; q <= span; q++) {
        const Xyzz29<F> cur = nxt;
        nxt = xyzz29_load<F>(head_part + (size_t)(t + (q < span ? q + 1 : q)) * XYZZ29_BYTES);     // (index clamped: the last fetch is unused)
        xyzz29_add_quad(acc, cur);
    }
    if (quad_lane() == 0) xyzz29_store(bucket_sums + (size_t)key * XYZZ29_BYTES, acc);
}
template <class F>
DEV void fixup_light(uint32_t gid, uint32_t nquads, const uint32_t *__restrict__ plan, const uint32_t *__restrict__ offsets,
                     const unsigned char *__restrict__ head_part, const unsigned char *__restrict__ tail_part,
                     const uint32_t *__restrict__ tail_key, unsigned char *__restrict__ bucket_sums, uint32_t num_buckets,
                     const uint32_t *__restrict__ heavy_ctr, const U4 *__restrict__ meds) {
    // the medium runs, when there are many: dealt to the quads of the light section
    const uint32_t n_medium = heavy_ctr[5];
    if (medium_as_chains(heavy_ctr))
        for (uint32_t m = gid; m < n_medium; m += nquads) {
            const U4 r = meds[m];                                  // {lane, span, bucket, -}
            fixup_chain<F>(r.x, r.y, r.z, head_part, tail_part, bucket_sums);
        }
    const uint32_t L = plan[0], T = plan[1];
    // Indexed by segment (the lane that holds the run's tail partial) when segments are fewer than buckets; by bucket
    // (num_buckets != 0) when buckets are fewer -- small MSMs, where every bucket is cut several times and only one segment in
    // four holds a tail.
    uint32_t t, key, run_end;
    if (num_buckets) {
        key = gid;
        if (key >= num_buckets) return;
        const uint32_t run_start = offsets[key];
        run_end = offsets[key + 1];
        if (run_end == run_start) return;
        t = run_start / L;
        if ((run_end - 1) / L == t) return;        // inside one segment: k_accumulate wrote the bucket itself
    } else {
        t = gid;
        if (t >= T) return;
        key = tail_key[t];
        if (key == KEY_NONE) return;
        run_end = offsets[key + 1];
    }
    const uint32_t span = (run_end - 1) / L - t;   // lanes t+1 .. t+span hold head partials of this run
    if (span > (uint32_t)HEAVY_SPAN) return;       // a medium or heavy run: listed by k_accumulate
    fixup_chain<F>(t, span, key, head_part, tail_part, bucket_sums);
}

// LDS tree over the values of the first `cnt` quads of the workgroup (cnt <= blockDim.x / 4);
// result in quad 0.  red: blockDim.x / 4 entries of XYZZ29_BYTES.
template <class F> DEV void block_tree_sum_quad(Xyzz29<F> &acc, unsigned char *red, uint32_t cnt) {
    const uint32_t qi = threadIdx.x >> 2;
    if (quad_lane() == 0) xyzz29_store(red + qi * XYZZ29_BYTES, acc);
    __syncthreads();
    uint32_t width = 1;
    while (width < cnt) width <<= 1;
    for (uint32_t st = width >> 1; st > 0; st >>= 1) {
        if (qi < st && qi + st < cnt) {
            xyzz29_add_quad(acc, xyzz29_load<F>(red + (qi + st) * XYZZ29_BYTES));
            if (quad_lane() == 0) xyzz29_store(red + qi * XYZZ29_BYTES, acc);
        }
        __syncthreads();
    }
}
// LDS tree inside groups of `gq` consecutive quads (gq a power of two <= blockDim.x / 4): the sum of the values of the
// first `cnt` quads of a group ends in the group's quad 0.  cnt may differ between groups; the loop bounds do not.
template <class F> DEV void group_tree_sum_quad(Xyzz29<F> &acc, unsigned char *red, uint32_t gq, uint32_t cnt) {
    const uint32_t qi = threadIdx.x >> 2, ql = qi & (gq - 1u);
    if (quad_lane() == 0) xyzz29_store(red + qi * XYZZ29_BYTES, acc);
    __syncthreads();
    for (uint32_t st = gq >> 1; st > 0; st >>= 1) {
        if (ql < st && ql + st < cnt) {
            xyzz29_add_quad(acc, xyzz29_load<F>(red + (qi + st) * XYZZ29_BYTES));
            if (quad_lane() == 0) xyzz29_store(red + qi * XYZZ29_BYTES, acc);
        }
        __syncthreads();
    }
}
// stage A: sub_out[s] = sum of the <= 64 head partials of sub-job s.  One wave = 16 quads works on 16 / gq sub-jobs at
// a time, gq quads each: every quad first adds its share of the partials, then an LDS tree over the gq quads.  A
// quad addition costs ~4 100 issue cycles whatever the number of busy quads, so the kernel is bound by
// max(chain of one wave, waves per SIMD x chain): with few sub-jobs a whole wave per sub-job (gq = 16: chain 4 + 4)
// is fastest; when EVERY bucket is a heavy run (dense vectors under narrow windows: 4 096 sub-jobs at 2^17 pairs,
// c = 8) that puts four waves on every SIMD, half of whose additions are idle tree levels -- 89 us.  gq is
// therefore chosen so that the waves just cover the 1 024 SIMDs: 16 384 / sub-jobs, between 2 and 16.
// Workgroups of 256 lanes (64 quads, four waves on the four SIMDs of a CU): 64-lane workgroups were seen packed onto a
// fraction of the CUs, several chains sharing a SIMD while others idle.  heavy_ctr[4] counts the runs that stage B
// still has to finish (more than one sub-job).
#ifdef MIRA_CPU_EMU
static constexpr uint32_t HEAVY_BLOCK_A = 32, HEAVY_BLOCK_B = 256;   // emulated lanes are OS threads, and k_fixup_all is launched for every commit: eight quads per workgroup there
#else
static constexpr uint32_t HEAVY_BLOCK_A = 256, HEAVY_BLOCK_B = 256;
#endif
// A run that is one sub-job (<= HEAVY_SUB partials: the usual case) is finished here -- tail partial
// added, bucket written -- and stage B skips it.
template <class F>
DEV void fixup_heavy_a(uint32_t block, uint32_t nblocks, unsigned char *red, const uint32_t *__restrict__ heavy_ctr, const U4 *__restrict__ subs, const U4 *__restrict__ runs,
                       const unsigned char *__restrict__ head_part, const unsigned char *__restrict__ tail_part,
                       unsigned char *__restrict__ sub_out, unsigned char *__restrict__ bucket_sums, const U4 *__restrict__ meds) {
    // sub-jobs [0, nheavy) come from the heavy runs; a FEW medium runs follow them as sub-jobs of their own (one each)
    const uint32_t nheavy = heavy_ctr[1], n_medium = heavy_ctr[5], qi = threadIdx.x >> 2;
    const uint32_t nsubs = nheavy + (medium_as_chains(heavy_ctr) ? 0u : n_medium);
    uint32_t lg = 4;                                                                            // quads per sub-job = 2^lg: as many as the stage's workgroups hold, 2 .. 16
    while (lg > 1 && (nsubs << lg) > nblocks * (HEAVY_BLOCK_A / 4)) lg--;
    while ((1u << lg) > HEAVY_BLOCK_A / 4) lg--;                                                // (never more than a workgroup has)
    const uint32_t gq = 1u << lg, spb = (HEAVY_BLOCK_A / 4) >> lg, ql = qi & (gq - 1u);       // sub-jobs per workgroup
    for (uint32_t s0 = block * spb; s0 < nsubs; s0 += nblocks * spb) {
        const uint32_t s = s0 + (qi >> lg);
        const bool active = s < nsubs;
        U4 d = U4{0, 0, 0, 0}, rm = U4{0, 0, 0, 0};
        const bool medium = active && s >= nheavy;
        if (medium) { rm = meds[s - nheavy]; d = U4{rm.x + 1, rm.y, 0, 0}; }          // {first partial, count, -, -} of run {lane, span, bucket}
        else if (active) d = subs[s];
        Xyzz29<F> acc = xyzz29_identity<F>();
        if (ql < d.y) {
            // the next partial is fetched before the current one is added (its index clamped: the last fetch is unused)
            Xyzz29<F> nxt = xyzz29_load<F>(head_part + (size_t)(d.x + ql) * XYZZ29_BYTES);
            for (uint32_t q = ql; q < d.y; q += gq) {
                const Xyzz29<F> cur = nxt;
                const uint32_t qn = q + gq < d.y ? q + gq : q;
                nxt = xyzz29_load<F>(head_part + (size_t)(d.x + qn) * XYZZ29_BYTES);
                xyzz29_add_quad(acc, cur);
            }
        }
        group_tree_sum_quad(acc, red, gq, d.y < gq ? d.y : gq);
        if (active && ql == 0) {
            const U4 r = medium ? rm : runs[d.z];                  // {lane, span, bucket, first sub-job}
            if (r.y <= HEAVY_SUB) {
                xyzz29_add_quad(acc, xyzz29_load<F>(tail_part + (size_t)r.x * XYZZ29_BYTES));
                if (quad_lane() == 0) xyzz29_store(bucket_sums + (size_t)r.z * XYZZ29_BYTES, acc);
            } else if (quad_lane() == 0) {
                xyzz29_store(sub_out + (size_t)s * XYZZ29_BYTES, acc);
            }
        }
        __syncthreads();
    }
}
// ONE launch after k_accumulate: workgroups [0, heavy_blocks) sum the sub-jobs of the heavy runs (stage A; they come first in the
// grid so that the long chains start first), the others the short chains, 64 quads each.  Both are latency-bound on mostly idle
// SIMDs: side by side they take what the longer one takes.
template <class F>
KERNEL void __launch_bounds__(256) k_fixup_all(uint32_t heavy_blocks, const uint32_t *__restrict__ plan, const uint32_t *__restrict__ offsets,
                        const unsigned char *__restrict__ head_part, const unsigned char *__restrict__ tail_part,
                        const uint32_t *__restrict__ tail_key, unsigned char *__restrict__ bucket_sums, uint32_t num_buckets,
                        const uint32_t *__restrict__ heavy_ctr, const U4 *__restrict__ subs, const U4 *__restrict__ runs,
                        unsigned char *__restrict__ sub_out, const U4 *__restrict__ meds) {
    __shared__ __attribute__((aligned(16))) unsigned char red[(HEAVY_BLOCK_A / 4) * XYZZ29_BYTES];
    if (blockIdx.x < heavy_blocks) {                                 // block-uniform
        fixup_heavy_a<F>(blockIdx.x, heavy_blocks, red, heavy_ctr, subs, runs, head_part, tail_part, sub_out, bucket_sums, meds);
        return;
    }
    fixup_light<F>((blockIdx.x - heavy_blocks) * (blockDim.x >> 2) + (threadIdx.x >> 2), (gridDim.x - heavy_blocks) * (blockDim.x >> 2), plan, offsets, head_part, tail_part,
                   tail_key, bucket_sums, num_buckets, heavy_ctr, meds);
}
// stage B: one workgroup per heavy run: bucket = tail partial + sum of its sub-job results
template <class F>
KERNEL void __launch_bounds__(256) k_fixup_heavy_b(const uint32_t *__restrict__ heavy_ctr, const U4 *__restrict__ runs,
                          const unsigned char *__restrict__ sub_out, const unsigned char *__restrict__ tail_part,
                          unsigned char *__restrict__ bucket_sums) {
    __shared__ __attribute__((aligned(16))) unsigned char red[(HEAVY_BLOCK_B / 4) * XYZZ29_BYTES];
    if (heavy_ctr[4] == 0) return;                                 // every heavy run was one sub-job: stage A finished them all
    const uint32_t nruns = heavy_ctr[0], qi = threadIdx.x >> 2, nq = blockDim.x >> 2;
    for (uint32_t h = blockIdx.x; h < nruns; h += gridDim.x) {
        const U4 r = runs[h];                                      // {lane, span, bucket, first sub-job}
        const uint32_t nsub = (r.y + HEAVY_SUB - 1) / HEAVY_SUB;
        if (nsub == 1) continue;                                   // finished by stage A
        Xyzz29<F> acc = xyzz29_identity<F>();
        for (uint32_t q = qi; q < nsub; q += nq)
            xyzz29_add_quad(acc, xyzz29_load<F>(sub_out + (size_t)(r.w + q) * XYZZ29_BYTES));
        if (qi == 0) xyzz29_add_quad(acc, xyzz29_load<F>(tail_part + (size_t)r.x * XYZZ29_BYTES));
        block_tree_sum_quad(acc, red, nsub < nq ? nsub : nq);
        if (threadIdx.x == 0) xyzz29_store(bucket_sums + (size_t)r.z * XYZZ29_BYTES, acc);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// Bucket reduction.  QUAD = true: every work item is a quad of lanes (latency-bound sizes: few
// buckets, the chain of dependent additions is what takes the time); QUAD = false: one lane per work
// item (2^19 buckets at c = 16: throughput-bound, four times fewer lanes do the same work).
// Work item (w, j) folds buckets [j*m, (j+1)*m) of window w:
//   R[w][j] = sum_i (j*m + i + 1) * S[w][j*m + i]
template <class F, bool QUAD>
KERNEL void __launch_bounds__(256) k_reduce_chunks(const unsigned char *__restrict__ bucket_sums, uint32_t B, uint32_t m, uint32_t W,
                            unsigned char *__restrict__ R) {
    const uint32_t nchunks = B / m;
    const uint32_t g = QUAD ? quad_gid() : blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nchunks * W) return;
    const uint32_t w = g / nchunks, j = g % nchunks;
    const size_t b0 = (size_t)w * B + (size_t)j * m;
    const unsigned char *S = bucket_sums + b0 * XYZZ29_BYTES;
    Xyzz29<F> running = xyzz29_identity<F>(), ws = xyzz29_identity<F>();
    for (int i = (int)m - 1; i >= 0; i--) {
        xyzz29_add_sel<QUAD>(running, xyzz29_load<F>(S + (size_t)i * XYZZ29_BYTES));
        xyzz29_add_sel<QUAD>(ws, running);
    }
    // + (j*m) * running, MSB-first double-and-add on the (<= 15-bit) chunk offset
    const uint32_t k = j * m;
    if (k != 0 && !xyzz29_is_identity(running)) {
        Xyzz29<F> acc = xyzz29_identity<F>();
        for (int bit = 31 - __builtin_clz(k); bit >= 0; bit--) {
            acc = xyzz29_double_sel<QUAD>(acc);
            if ((k >> bit) & 1) xyzz29_add_sel<QUAD>(acc, running);
        }
        xyzz29_add_sel<QUAD>(ws, acc);
    }
    if (!QUAD || quad_lane() == 0) xyzz29_store(R + (size_t)g * XYZZ29_BYTES, ws);
}

// Workgroup per window: window_sums[w] = sum_j R[w][j], exported as X, Y, ZZ, ZZZ in the
// reference's canonical R = 2^256 form (128 B) for the host epilogue.  blockDim.x == WSUM_BLOCK:
// 128 quads, or 512 single lanes (72 KiB of LDS).
template <class F, bool QUAD>
KERNEL void __launch_bounds__(512) k_window_sum(const unsigned char *__restrict__ R, uint32_t nchunks, unsigned char *__restrict__ window_sums,
                                                const uint32_t *__restrict__ hist) {   // planning statistics (or null): copied behind the sums, one copy to the host for both
    DYN_SHARED(unsigned char, red);
    const uint32_t w = blockIdx.x;
    if (hist && w == 0 && threadIdx.x < 256) reinterpret_cast<uint32_t *>(window_sums + (size_t)gridDim.x * 128)[threadIdx.x] = hist[threadIdx.x];
    const uint32_t qi = QUAD ? threadIdx.x >> 2 : threadIdx.x, nq = QUAD ? blockDim.x >> 2 : blockDim.x;
    Xyzz29<F> acc = xyzz29_identity<F>();
    for (uint32_t q = qi; q < nchunks; q += nq)
        xyzz29_add_sel<QUAD>(acc, xyzz29_load<F>(R + ((size_t)w * nchunks + q) * XYZZ29_BYTES));
    if constexpr (QUAD) {
        block_tree_sum_quad(acc, red, nchunks < nq ? nchunks : nq);
    } else {
        xyzz29_store(red + threadIdx.x * XYZZ29_BYTES, acc);
        __syncthreads();
        for (uint32_t st = WSUM_BLOCK / 2; st > 0; st >>= 1) {
            if (threadIdx.x < st && threadIdx.x + st < nchunks) {      // lanes beyond nchunks hold the identity
                xyzz29_add(acc, xyzz29_load<F>(red + (threadIdx.x + st) * XYZZ29_BYTES));
                xyzz29_store(red + threadIdx.x * XYZZ29_BYTES, acc);
            }
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) xyzz29_export_r256(window_sums + (size_t)w * 128, acc);
}

// ------------------------------------------------------------------------------------------
// Synthetic inputs; same definition as oracle/pyref.py synth_* so that tests
// can cross-check them.  One splitmix64 stream per index.
HD uint64_t sm_next(uint64_t &s) {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static constexpr uint64_t STREAM_MUL = 0xD6E8FEB86659FD93ull;

// kind 0 uniform, 1 witness-like (70 % zero, 20 % < 2^32, 10 % uniform).  Montgomery out.
template <class FS>
KERNEL void k_synth_scalars(uint64_t n, uint64_t index0, uint64_t seed, int kind, unsigned char *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s = seed + (index0 + i) * STREAM_MUL;
    Fe<FS> v;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint64_t x = sm_next(s);
        v.l[2 * k] = (uint32_t)x; v.l[2 * k + 1] = (uint32_t)(x >> 32);
    }
    for (int it = 0; it < 6; it++) {   // 2^256 < 6 P
        Fe<FS> t;
        if (!sub_p(t, v)) v = t;
    }
    if (kind == 1) {
        uint64_t sel = sm_next(s) % 10;
        if (sel < 7) v = fe_zero<FS>();
        else if (sel < 9) {
#pragma unroll
            for (int k = 1; k < 8; k++) v.l[k] = 0;
        }
    }
    fe_store(out + i * 32, fe_to_mont(v));
}

// P_i = k_i * G, k_i = odd 128-bit integer from the stream; affine Montgomery out.
template <class FB>
KERNEL void __launch_bounds__(64) k_synth_bases(uint64_t n, uint64_t index0, uint64_t seed, const unsigned char *__restrict__ gen,
                          unsigned char *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s = seed + (index0 + i) * STREAM_MUL;
    uint64_t k0 = sm_next(s) | 1ull, k1 = sm_next(s);
    Aff<FB> g = aff_load<FB>(gen);
    Xyzz<FB> acc = xyzz_identity<FB>();
    for (int bit = 127; bit >= 0; bit--) {
        acc = xyzz_double(acc);
        uint64_t word = bit >= 64 ? k1 : k0;
        if ((word >> (bit & 63)) & 1) xyzz_add_affine(acc, g);
    }
    aff_store(out + i * 64, xyzz_to_affine(acc));
}
