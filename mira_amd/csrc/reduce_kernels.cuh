// Bucket reduction of the MSM: T = sum_b (b + 1) S_b over the B = 2^cb buckets of a set, for many sets at once
// (one set per window, per commitment of a batch over a shared-bucket table set, ...).  The last stage of
// best_multiexp's bucket method (halo2_proofs, called from src/commitment.rs:80) -- there a serial running sum per
// window; here no lane runs a chain longer than 2 m + O(log) additions whatever B is.
//
// A bucket index is split b = h m + l, m = 2^lambda buckets per chunk, eta = cb - lambda high bits:
//     T = sum_h V_h + m sum_h h A_h,      A_h = sum_l S_(h,l),   V_h = sum_l (l + 1) S_(h,l)     (phase A)
//       = V + m sum_(t < eta) 2^t D_t,    V = sum_h V_h,         D_t = sum of the A_h with bit t of h set
// Phase A is the classic running sum over the m buckets of a chunk: 2 m dependent additions per work item, all chunks
// in parallel -- two general additions per bucket, which is what the whole reduction costs in arithmetic.  The per-chunk
// scalar multiplication by h m of round 3 (a 15-bit double-and-add: as many field products again, and a chain of 22
// operations) is gone: the high bits are resolved by a TREE over the chunks whose nodes carry the vector
//     [A, V, D_0 .. D_(k-1)]                                    (a node that covers 2^k chunks)
// and combine as  parent[j] = left[j] + right[j]  for j < k + 2,  parent.D_k = right.A  -- k + 2 independent additions
// per combine, ~3 per chunk over the whole tree, depth eta.  k_bucket_tree runs phase A and the first kappa levels of
// the tree for 2^kappa chunks per workgroup in LDS; k_set_finish (one workgroup per set) runs the remaining levels over
// the workgroups' nodes and leaves the set's result as P PIECES
//     piece_p = sum_(pos in [s_p, s_(p+1))) 2^(pos - s_p) X_pos,    X_0 = V,  X_(lambda + t) = D_t,
// so that T = sum_p 2^(s_p) piece_p: the host's chain of doublings over the windows (capi.hip: horner_pieces) stops at
// every piece instead of every window -- the doublings it does anyway, P - 1 more additions per window -- and the
// device's own Horner chains are (cb / P) long instead of cb.  P = 1 gives the plain window sum (sharded partials).
//
// Every tree addition is done by a DPP quad (quad29.cuh).  Phase A by quads while the work is latency-bound, by single
// lanes when there are enough chunks to fill the SIMDs (2^19 buckets under 16-bit windows).
#pragma once
#include "quad29.cuh"

struct PieceCfg {
    uint32_t P;             // pieces per set, 1 .. 8
    uint32_t start[9];      // start[p] = first bit position of piece p; start[P] = cb
};

// Nodes of `S0` components each at slots [i * S0, i * S0 + S0), i < 2^levels: combine them pairwise, level by level,
// into ONE node of S0 + levels components at slot 0.  A node of level k owns the S0 2^k slots of its leaves and
// holds S0 + k components at their start.  Block-uniform control flow (barriers inside).
template <class F> DEV void node_tree_quad(unsigned char *slots, uint32_t S0, uint32_t levels) {
    const uint32_t qi = threadIdx.x >> 2, nq = blockDim.x >> 2;
    for (uint32_t k = 0; k < levels; k++) {
        const uint32_t parents = 1u << (levels - 1 - k), comps = S0 + k, span = S0 << k;
        for (uint32_t q = qi; q < parents * comps; q += nq) {
            const uint32_t p = q / comps, j = q - p * comps;
            unsigned char *L = slots + ((size_t)(2 * p) * span + j) * XYZZ29_BYTES;
            const unsigned char *R = L + (size_t)span * XYZZ29_BYTES;
            Xyzz29<F> a = xyzz29_load<F>(L);
            const Xyzz29<F> b = xyzz29_load<F>(R);
            xyzz29_add_quad(a, b);
            if (quad_lane() == 0) {
                xyzz29_store(L, a);
                // D_k = right.A.  At k = 0 the slot behind the left node's S0 components IS the right node's A; from
                // level 1 on it is a free slot inside the left child's region (S0 + k < S0 2^k), and right.A (component 0,
                // which this quad has just read) is copied there.
                if (j == 0 && k > 0) xyzz29_store(slots + ((size_t)(2 * p) * span + comps) * XYZZ29_BYTES, b);
            }
        }
        __syncthreads();
    }
}

// grid = sets * (B >> lambda >> kappa) workgroups of (QUAD ? 4 : 1) << kappa lanes; dynamic LDS = 2 XYZZ29_BYTES << kappa.
// Buckets of all sets are contiguous (set s at bucket s B), chunks and workgroups therefore too: workgroup g covers chunks
// [g 2^kappa, (g + 1) 2^kappa) of the global chunk sequence and never straddles two sets (2^kappa divides B / m).
// nodes_out[g] = kappa + 2 points: A, V, D_0 .. D_(kappa-1) of the workgroup's chunks (chunk indices relative to its first).
template <class F, bool QUAD>
KERNEL void __launch_bounds__(512) k_bucket_tree(const unsigned char *__restrict__ bucket_sums, uint32_t lambda, uint32_t kappa,
                                                 unsigned char *__restrict__ nodes_out) {
    DYN_SHARED(unsigned char, slots);
    const uint32_t m = 1u << lambda;
    const uint32_t item = QUAD ? threadIdx.x >> 2 : threadIdx.x;
    if (item < (1u << kappa)) {                              // (a workgroup has at least one quad: more lanes than chunks when kappa < 2)
        const size_t chunk = ((size_t)blockIdx.x << kappa) + item;
        const unsigned char *S = bucket_sums + (chunk << lambda) * XYZZ29_BYTES;
        Xyzz29<F> running = xyzz29_identity<F>(), ws = xyzz29_identity<F>();
        // the next bucket sum is requested before the current one is added (its index clamped: the last fetch is unused)
        Xyzz29<F> nxt = xyzz29_load<F>(S + (size_t)(m - 1) * XYZZ29_BYTES);
#ifdef MIRA_PROBE_NO_PHASE_A                                  // timing probes only (tools/build_probe_variants.sh): wrong results
        for (int i = 0; i >= 1; i--) {
#else
        for (int i = (int)m - 1; i >= 0; i--) {
#endif
            const Xyzz29<F> cur = nxt;
            nxt = xyzz29_load<F>(S + (size_t)(i > 0 ? i - 1 : 0) * XYZZ29_BYTES);
            if constexpr (QUAD) { xyzz29_add_quad(running, cur); xyzz29_add_quad(ws, running); }
            else { xyzz29_add(running, cur); xyzz29_add(ws, running); }
        }
        if (!QUAD || quad_lane() == 0) {
            xyzz29_store(slots + (size_t)(2 * item) * XYZZ29_BYTES, running);
            xyzz29_store(slots + (size_t)(2 * item + 1) * XYZZ29_BYTES, ws);
        }
    }
    __syncthreads();
#ifndef MIRA_PROBE_NO_TREE
    node_tree_quad<F>(slots, 2, kappa);
#endif
    const uint32_t words = (kappa + 2) * (XYZZ29_BYTES / 4);
    uint32_t *dst = reinterpret_cast<uint32_t *>(nodes_out + (size_t)blockIdx.x * (kappa + 2) * XYZZ29_BYTES);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(slots);
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
}

// One workgroup per set: the 2^gamma nodes of its workgroups (S0 = kappa + 2 components each, contiguous) -> the set's
// node [A, V, D_0 .. D_(eta-1)], eta = kappa + gamma -> P pieces, exported as X, Y, ZZ, ZZZ in the reference's canonical
// R = 2^256 form (128 B each) at out[(set P + p) 128] for the host epilogue.  dynamic LDS = (S0 << gamma) XYZZ29_BYTES.
template <class F>
KERNEL void __launch_bounds__(512) k_set_finish(const unsigned char *__restrict__ nodes, uint32_t S0, uint32_t gamma, uint32_t lambda, PieceCfg pc,
                                                unsigned char *__restrict__ out,
                                                const uint32_t *__restrict__ hist,     // planning statistics (or null): copied behind the pieces, one copy to the host for both
                                                uint32_t *__restrict__ done_ctr, uint64_t *__restrict__ flag, uint64_t stamp) {
    // flag (or null): `out` is mapped host memory; the workgroup that finishes last stores `stamp` there, behind everybody's
    // results (system-scope fences on both sides of the device-scope counter), and leaves the counter at zero for the next launch
    DYN_SHARED(unsigned char, slots);
    const uint32_t set = blockIdx.x;
    if (hist && set == 0)
        for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) reinterpret_cast<uint32_t *>(out + (size_t)gridDim.x * pc.P * 128)[i] = hist[i];
    const uint32_t words = (S0 << gamma) * (XYZZ29_BYTES / 16);      // 16 bytes per lane and trip (a point is nine of them)
    const U4 *src = reinterpret_cast<const U4 *>(nodes + (size_t)set * (S0 << gamma) * XYZZ29_BYTES);
    U4 *dst = reinterpret_cast<U4 *>(slots);
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
    node_tree_quad<F>(slots, S0, gamma);
    const uint32_t p = threadIdx.x >> 2;
    if (p < pc.P) {
        // Horner over the bit positions of the piece, highest first; positions 1 .. lambda - 1 hold nothing
        Xyzz29<F> acc = xyzz29_identity<F>();
        for (int pos = (int)pc.start[p + 1] - 1; pos >= (int)pc.start[p]; pos--) {
            acc = xyzz29_double_quad(acc);
            if (pos == 0) xyzz29_add_quad(acc, xyzz29_load<F>(slots + XYZZ29_BYTES));
            if (pos >= (int)lambda) xyzz29_add_quad(acc, xyzz29_load<F>(slots + (size_t)(2 + pos - (int)lambda) * XYZZ29_BYTES));
        }
        if (quad_lane() == 0) xyzz29_export_r256(out + ((size_t)set * pc.P + p) * 128, acc);
    }
    if (!flag) return;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(done_ctr, 1u) == gridDim.x - 1) {
        *done_ctr = 0;
        __threadfence_system();
        store_release_system(flag, stamp);
    }
}
