// Cross-term evaluation on the GPU: the calculation list of a GraphEvaluator (reference
// src/polynomial/graph_evaluator.rs:70-162, 361-390) run for every row at once.
//
// The reference walks the list once per row, rows in parallel over rayon threads
// (src/nifs/vanilla/mod.rs:104-113).  Here a lane owns a row and the whole wave walks the list in
// lockstep: instruction words, constants and challenges are wave-uniform (scalar loads), column
// reads are 32-byte loads at consecutive rows (rotations shift the whole wave), and the
// intermediates live in a slot-major workspace ws[slot][limb][lane] so that every intermediate
// access is nine fully coalesced dword bursts per wave.  Slots are the host's register allocation of
// the intermediates (graph.hip): a handful, reused, so the workspace stays cache resident.
//
// Arithmetic: 9 x 29-bit limbs (field29.cuh: half the instructions of the 8 x 32-bit CIOS per
// multiplication).  Values are "loose" between calculations; the host COMPILES the program
// (graph.hip) with a proven bound for every value -- invariant: whatever is stored or forwarded is
// < 12 P, so any two values may be multiplied (144 <= 168) -- and inserts a normalising
// multiplication by one where a chain of additions would leave the budget.  Columns are read as they lie
// in memory (x * 2^256, no lifting product): the compiler tracks for every value the power of two it
// carries (its "form", graph.hip), hands constants and challenges over in the form each use wants,
// and converts where forms cannot be made to agree; the program's last instruction leaves the result
// as x * 2^256 below 2 P and the kernel stores it canonical.  Field results are exact, so every value
// equals the reference's bit for bit whatever the order of rows.
#pragma once
#include "field29.cuh"
#include "../../include/mira_gpu.h"

struct GraphCol {
    const unsigned char *p;
    uint32_t kind, pad;
};

// Compiled instruction stream: per instruction
//   [op | K << 8]  [dst slot]  [bounds of a and b in 1/256 P: lo 16 | hi 16 bits]  [source a]  ([source b])
// and for GOP_MAC (a * b + c: an addition that absorbed the product feeding it, graph.hip) two more: [source c] [bound of c]
// INTERMEDIATE payloads are slots; GRAPH_SRC_PREV = the value of the instruction just before
// (still in registers -- most results of a post-order expression walk are consumed by the very next
// instruction and never touch the workspace).  K = the multiple of P a subtraction adds.  The
// bounds word only feeds the test build's bound bookkeeping (F29_TRACK).
static constexpr uint32_t GRAPH_SRC_PREV = 4u;
static constexpr uint32_t GRAPH_NO_SLOT = 0xFFFFFFFFu;
static constexpr uint32_t GOP_ADD = 0, GOP_SUB = 1, GOP_MUL = 2, GOP_SQR = 3, GOP_DBL = 4, GOP_NEG = 5, GOP_COPY = 6, GOP_NORM = 7, GOP_MAC = 8;
static constexpr double GRAPH_MAX_BOUND = 12.0;          // of every stored or forwarded value, in multiples of P

template <class F> DEV Fe29<F> graph_sub(const Fe29<F> &a, const Fe29<F> &b, uint32_t K) {
    switch (K) {                                         // smallest multiple of P above the subtrahend (chosen by the host)
        case 2: return f29_sub<2>(a, b);
        case 4: return f29_sub<4>(a, b);
        case 8: return f29_sub<8>(a, b);
        default: return f29_sub<16>(a, b);
    }
}

// one compiled graph of a batch: the graphs of a batch (the d - 1 cross-term expressions of a fold
// step, src/nifs/vanilla/mod.rs:100-121) read the same columns and challenges
struct GraphJob {
    const uint32_t *code, *consts29, *challenges29;      // constants and this evaluation's challenges, each in the forms the program reads them
    const int32_t *rotations;
    unsigned char *out;
    uint32_t ninstr, nlds;                               // slots below nlds live in LDS (the compiler numbers the most used ones lowest)
};

// grid = (row blocks, graphs of the batch): a fold step's handful of graphs over 2^17 rows are two waves
// per SIMD each -- launched together they fill the wave slots (82 VGPRs: six per SIMD)
template <class F>
KERNEL void __launch_bounds__(256) k_graph_eval(const GraphJob *__restrict__ jobs,
                         const GraphCol *__restrict__ cols, uint64_t nrows, uint32_t *__restrict__ ws_all, uint64_t ws_stride) {
    using S = typename F::Sat;
    DYN_SHARED(uint32_t, lds);                           // lds[slot][limb][lane of the workgroup]
    const GraphJob job = jobs[blockIdx.y];
    const uint32_t nlds = job.nlds;
    const uint32_t *__restrict__ code = job.code, *__restrict__ consts29 = job.consts29, *__restrict__ challenges29 = job.challenges29;
    const int32_t *__restrict__ rotations = job.rotations;
    unsigned char *__restrict__ out = job.out;
    const uint32_t ninstr = job.ninstr;
    uint32_t *__restrict__ ws = ws_all + (size_t)blockIdx.y * ws_stride;
    const uint64_t T = (uint64_t)gridDim.x * blockDim.x, lane = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t row = lane; row < nrows; row += T) {
        Fe29<F> v = f29_zero<F>(), prev;
        auto fetch = [&](uint32_t s, uint32_t bound) -> Fe29<F> {
            const uint32_t kind = s >> 29, payload = s & 0x1FFFFFFFu;
            Fe29<F> r;
            if (kind == GRAPH_SRC_PREV) {
                r = prev;
            } else if (kind == MIRA_SRC_CONSTANT || kind == MIRA_SRC_CHALLENGE) {
                const uint32_t *p = (kind == MIRA_SRC_CONSTANT ? consts29 : challenges29) + (size_t)payload * 9;
#pragma unroll
                for (int k = 0; k < 9; k++) r.l[k] = p[k];
            } else if (kind == MIRA_SRC_INTERMEDIATE) {
                if (payload < nlds) {                        // wave-uniform
#pragma unroll
                    for (int k = 0; k < 9; k++) r.l[k] = lds[(payload * 9 + k) * blockDim.x + threadIdx.x];
                } else {
#pragma unroll
                    for (int k = 0; k < 9; k++) r.l[k] = ws[((size_t)payload * 9 + k) * T + lane];
                }
            } else {
                const GraphCol c = cols[payload & 0xFFFFFu];
                int64_t rr = ((int64_t)row + rotations[payload >> 20]) % (int64_t)nrows;   // rem_euclid, graph_evaluator.rs:51-53
                if (rr < 0) rr += (int64_t)nrows;
                // a column enters as it lies in memory -- x * 2^256, "form 1" of the compiler (graph.hip), no lifting
                // product -- and a selector as the number one in that form (src/plonk/eval.rs:62)
                if (c.kind == MIRA_COL_BOOL) {
                    Fe<S> one_r;
#pragma unroll
                    for (int k = 0; k < 8; k++) one_r.l[k] = c.p[rr] ? S::R1[k] : 0u;
                    r = f29_unpack_canonical<F>(one_r);
                } else {
                    r = f29_unpack_canonical<F>(fe_load<S>(c.p + (size_t)rr * 32));
                }
            }
            F29_SET(r, (double)bound / 256.0);
            (void)bound;
            return r;
        };
        const uint32_t *pc = code;
        for (uint32_t i = 0; i < ninstr; i++) {
            const uint32_t head = pc[0], dst = pc[1], bounds = pc[2];
            prev = v;
            const uint32_t op = head & 0xFFu, K = head >> 8;
            const Fe29<F> a = fetch(pc[3], bounds & 0xFFFFu);
            if (op == GOP_ADD) { v = f29_add(a, fetch(pc[4], bounds >> 16)); pc += 5; }
            else if (op == GOP_SUB) { v = graph_sub(a, fetch(pc[4], bounds >> 16), K); pc += 5; }
            else if (op == GOP_MUL) { v = f29_mul(a, fetch(pc[4], bounds >> 16)); pc += 5; }
            else if (op == GOP_MAC) { const Fe29<F> b = fetch(pc[4], bounds >> 16), c = fetch(pc[5], pc[6]); v = f29_add(f29_mul(a, b), c); pc += 7; }
            else {
                if (op == GOP_SQR) v = f29_sqr(a);
                else if (op == GOP_DBL) v = f29_dbl(a);
                else if (op == GOP_NEG) v = graph_sub(f29_zero<F>(), a, K);
                else if (op == GOP_NORM) v = f29_mul(a, f29_one<F>());
                else v = a;                                  // GOP_COPY
                pc += 4;
            }
            if (dst != GRAPH_NO_SLOT) {                      // no slot: nobody reads it again
                if (dst < nlds) {
#pragma unroll
                    for (int k = 0; k < 9; k++) lds[(dst * 9 + k) * blockDim.x + threadIdx.x] = v.l[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 9; k++) ws[((size_t)dst * 9 + k) * T + lane] = v.l[k];
                }
            }
        }
        fe_store(out + row * 32, reduce_once(f29_pack(v)));     // the program's last instruction left x * 2^256 below 2 P
    }
}
