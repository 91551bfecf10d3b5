// LDS-staged two-level bucket sort (replaces the single-level k_scatter for large MSMs).
//
// k_scatter places every entry with its own 4-byte store at a random address: 67 M partial-line
// writes per 2^22 MSM reach HBM one by one (measured 58 G stores/s, 1.15 ms).  Here every global
// write is a burst of consecutive entries:
//   level 1 (k_stage1)  a workgroup takes a tile of 16384 points of one window, ranks them by
//                       coarse bin (bucket >> fine_bits) with LDS atomics, stages them in LDS in
//                       bin order and copies each bin's run (tile / bins ~ 128 entries) to the
//                       range it claimed with ONE global atomic per bin;
//   level 2 (k_stage2)  a workgroup takes 16384 consecutive level-1 entries (one or two coarse
//                       bins), ranks them by bucket, stages, and copies each bucket's run to the
//                       range claimed in the final array.
// The bucket histogram (k_hist) and its scan already give every cursor: fine cursors are the bucket
// offsets, coarse cursors the offsets at coarse-bin boundaries.
#pragma once
#include "field.cuh"

static constexpr uint32_t STAGE_TILE = 16384;      // entries per workgroup tile (fixed-base tables: 2^19 .. 2^21 buckets)
// The per-window path (<= 2^15 buckets per window) takes half-size tiles: 74 KiB of LDS instead of 148,
// so TWO level-2 workgroups share a CU and one's loads overlap the other's ranking (level 1 + level 2
// 0.49 -> 0.42 ms at 2^22 pairs, 0.14 -> 0.11 at 2^20).  Its tiles span ~64 buckets; 2048 keys are plenty.
#ifndef MSM_STAGE_TILE_PW
#define MSM_STAGE_TILE_PW 8192      // (probe builds override these two)
#endif
#ifndef MSM_STAGE_KEYS_PW
#define MSM_STAGE_KEYS_PW 2048
#endif
static constexpr uint32_t STAGE_TILE_PW = MSM_STAGE_TILE_PW, STAGE_MAX_KEYS2_PW = MSM_STAGE_KEYS_PW;
static constexpr uint32_t STAGE_MAX_BINS1 = 512;   // coarse bins per window handled in LDS
static constexpr uint32_t STAGE_MAX_KEYS2 = 4096;  // bucket range one level-2 tile may span in LDS

// exclusive scan of cnt[0..nb) into ofs[0..nb) by the whole workgroup (any blockDim); returns total in *total
DEV void block_excl_scan(const uint32_t *cnt, uint32_t *ofs, uint32_t *tmp /* blockDim entries */, uint32_t nb, uint32_t *total) {
    const uint32_t per = (nb + blockDim.x - 1) / blockDim.x;
    const uint32_t lo = (threadIdx.x * per < nb) ? threadIdx.x * per : nb, hi = (lo + per < nb) ? lo + per : nb;
    uint32_t s = 0;
    for (uint32_t b = lo; b < hi; b++) s += cnt[b];
    tmp[threadIdx.x] = s;
    __syncthreads();
    // Hillis-Steele over the per-lane sums
    for (uint32_t off = 1; off < blockDim.x; off <<= 1) {
        uint32_t add = (threadIdx.x >= off) ? tmp[threadIdx.x - off] : 0;
        __syncthreads();
        tmp[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = tmp[threadIdx.x] - s;
    for (uint32_t b = lo; b < hi; b++) { ofs[b] = run; run += cnt[b]; }
    if (threadIdx.x == blockDim.x - 1) *total = tmp[threadIdx.x];
    __syncthreads();
}

// (the coarse cursors, cursor1[g] = offsets[g << fine_bits], are a by-product of k_scan_c)

// level 1.  grid = (ntiles, W_total); digits int16 window-major; B buckets per window, CB = B >> fine_bits
// coarse bins per window (<= STAGE_MAX_BINS1).  part entries: x = index | sign << 31, y = w * B + bucket.
// dynamic LDS: STAGE_TILE * 8 bytes staging.
// TABLE = true (fixed-base mode, table_kernels.cuh): the `wgroup` windows of one MSM share ONE bucket
// set (MSM j = w / wgroup of a batch: y = j * B + bucket, cursors indexed by j and the coarse bin) and
// x = (w % wgroup) * N + first + i, the index into the window tables (N = registered key length).
template <class DIGIT, bool TABLE, uint32_t TILE = STAGE_TILE>
KERNEL void __launch_bounds__(1024) k_stage1(const DIGIT *__restrict__ digits, uint32_t n, uint32_t B, uint32_t fine_bits, uint32_t CB,
                                              uint32_t N, uint32_t first, uint32_t wgroup, uint32_t *__restrict__ cursor1, U2 *__restrict__ part) {
    DYN_SHARED(U2, stage);
    __shared__ uint32_t cnt[STAGE_MAX_BINS1], lofs[STAGE_MAX_BINS1], gbase[STAGE_MAX_BINS1], tmp[1024], total_s;
    const uint32_t w = blockIdx.y;
    const uint32_t base = blockIdx.x * TILE, end = (base + TILE < n) ? base + TILE : n;
    const DIGIT *dw = digits + (size_t)w * n;
    const uint32_t set = TABLE ? w / wgroup : w;
    const uint32_t key_base = set * B, cur_base = set * CB, idx_base = TABLE ? (w % wgroup) * N + first : 0u;
    for (uint32_t b = threadIdx.x; b < CB; b += blockDim.x) cnt[b] = 0;
    __syncthreads();
    constexpr int PER = TILE / 1024;               // 16 points per lane: blockDim.x must be 1024 (the test emulation runs these kernels with all 1024 lanes too)
    DIGIT dreg[PER];                                     // the tile's digits stay in registers between the two passes
#pragma unroll
    for (int k = 0; k < PER; k++) {
        uint32_t i = base + k * 1024 + threadIdx.x;
        dreg[k] = (i < end) ? dw[i] : (DIGIT)0;
    }
#pragma unroll
    for (int k = 0; k < PER; k++) {
        int32_t d = dreg[k];
        if (d != 0) atomicAdd(&cnt[((uint32_t)(d < 0 ? -d : d) - 1) >> fine_bits], 1u);
    }
    __syncthreads();
    block_excl_scan(cnt, lofs, tmp, CB, &total_s);
    for (uint32_t b = threadIdx.x; b < CB; b += blockDim.x) {
        uint32_t c = cnt[b];
        gbase[b] = c ? atomicAdd(&cursor1[(size_t)cur_base + b], c) : 0;
        cnt[b] = lofs[b];                                // cnt[] becomes the running LDS cursor
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; k++) {
        int32_t d = dreg[k];
        if (d != 0) {
            uint32_t i = base + k * 1024 + threadIdx.x;
            uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1;
            uint32_t p = atomicAdd(&cnt[b >> fine_bits], 1u);
            stage[p] = U2{(idx_base + i) | (d < 0 ? 0x80000000u : 0u), key_base + b};
        }
    }
    __syncthreads();
    const uint32_t total = total_s;
    for (uint32_t p = threadIdx.x; p < total; p += blockDim.x) {     // consecutive p: consecutive addresses within a bin's run
        U2 e = stage[p];
        uint32_t cb = (e.y - key_base) >> fine_bits;
        part[gbase[cb] + (p - lofs[cb])] = e;
    }
}

// Bucket counts from the level-1 output (table mode: 2^19 buckets do not fit an LDS histogram over
// the raw digits, but a tile of coarse-sorted entries spans only ~2000 of them).  Same tiling as k_stage2.
KERNEL void __launch_bounds__(1024) k_stage2_count(const U2 *__restrict__ part, const uint32_t *__restrict__ total_ptr, uint32_t fine_bits,
                                                    uint32_t *__restrict__ counts) {
    __shared__ uint32_t cnt[STAGE_MAX_KEYS2];
    const uint32_t total = *total_ptr;
    const uint32_t base = blockIdx.x * STAGE_TILE;
    if (base >= total) return;
    const uint32_t end = (total - base > STAGE_TILE) ? base + STAGE_TILE : total;
    const uint32_t key_lo = (part[base].y >> fine_bits) << fine_bits;
    for (uint32_t b = threadIdx.x; b < STAGE_MAX_KEYS2; b += blockDim.x) cnt[b] = 0;
    __syncthreads();
    for (uint32_t p = base + threadIdx.x; p < end; p += blockDim.x) {
        uint32_t y = part[p].y, k = y - key_lo;
        if (k < STAGE_MAX_KEYS2) atomicAdd(&cnt[k], 1u);
        else atomicAdd(&counts[y], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < STAGE_MAX_KEYS2; b += blockDim.x) {
        uint32_t c = cnt[b];
        if (c) atomicAdd(&counts[(size_t)key_lo + b], c);
    }
}

// level 2.  grid = ceil(total / STAGE_TILE) upper bound (tiles beyond `total` exit).  Tile entries
// have bucket ids in [key_lo, key_lo + STAGE_MAX_KEYS2) in the common case; entries outside that
// range (sparse inputs spanning many coarse bins) are placed directly.
// dynamic LDS: STAGE_TILE * 4 (staged x) + STAGE_TILE * 2 (staged local key).
template <uint32_t TILE = STAGE_TILE, uint32_t KEYS = STAGE_MAX_KEYS2>
KERNEL void __launch_bounds__(1024) k_stage2(const U2 *__restrict__ part, const uint32_t *__restrict__ total_ptr, uint32_t fine_bits,
                                              uint32_t *__restrict__ cursor2, uint32_t *__restrict__ sorted) {
    DYN_SHARED(uint32_t, stage_x);
    uint16_t *stage_k = reinterpret_cast<uint16_t *>(stage_x + TILE);
    __shared__ uint32_t cnt[KEYS], lofs[KEYS], gbase[KEYS], tmp[1024], total_s;
    const uint32_t total = *total_ptr;
    const uint32_t base = blockIdx.x * TILE;
    if (base >= total) return;
    const uint32_t end = (total - base > TILE) ? base + TILE : total;
    const uint32_t key_lo = (part[base].y >> fine_bits) << fine_bits;
    for (uint32_t b = threadIdx.x; b < KEYS; b += blockDim.x) cnt[b] = 0;
    __syncthreads();
    constexpr int PER = TILE / 1024;
    U2 ereg[PER];                                        // the tile's entries stay in registers between the two passes
#pragma unroll
    for (int j = 0; j < PER; j++) {
        uint32_t p = base + j * 1024 + threadIdx.x;
        ereg[j] = (p < end) ? part[p] : U2{0, 0xFFFFFFFFu};
    }
#pragma unroll
    for (int j = 0; j < PER; j++) {
        uint32_t k = ereg[j].y - key_lo;
        if (ereg[j].y != 0xFFFFFFFFu && k < KEYS) atomicAdd(&cnt[k], 1u);
    }
    __syncthreads();
    block_excl_scan(cnt, lofs, tmp, KEYS, &total_s);
    for (uint32_t b = threadIdx.x; b < KEYS; b += blockDim.x) {
        uint32_t c = cnt[b];
        gbase[b] = c ? atomicAdd(&cursor2[(size_t)key_lo + b], c) : 0;
        cnt[b] = lofs[b];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; j++) {
        U2 e = ereg[j];
        if (e.y == 0xFFFFFFFFu) continue;
        uint32_t k = e.y - key_lo;
        if (k < KEYS) {
            uint32_t q = atomicAdd(&cnt[k], 1u);
            stage_x[q] = e.x;
            stage_k[q] = (uint16_t)k;
        } else {
            sorted[atomicAdd(&cursor2[e.y], 1u)] = e.x;             // out-of-range key: direct placement
        }
    }
    __syncthreads();
    const uint32_t staged = total_s;
    for (uint32_t q = threadIdx.x; q < staged; q += blockDim.x) {
        uint32_t k = stage_k[q];
        sorted[gbase[k] + (q - lofs[k])] = stage_x[q];
    }
}
