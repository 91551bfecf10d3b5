// Cross-term evaluation on the GPU: the calculation list of a GraphEvaluator (reference
// src/polynomial/graph_evaluator.rs:70-162, 361-390) run for every row at once.
//
// The reference walks the list once per row, rows in parallel over rayon threads
// (src/nifs/vanilla/mod.rs:104-113).  Here a lane owns a row and the whole wave walks the list in
// lockstep: instruction words, constants and challenges are wave-uniform (scalar loads), column
// reads are 32-byte loads at consecutive rows (rotations shift the whole wave), and the
// intermediates live in a slot-major workspace ws[slot][lane] so that every intermediate access is
// a fully coalesced 2 KiB burst per wave.  Slots are the host's register allocation of the
// intermediates (graph.hip): a handful, reused, so the workspace stays cache resident.
//
// All arithmetic is canonical Montgomery (field.cuh): every value equals the reference's bit for
// bit whatever the order of rows.
#pragma once
#include "field.cuh"
#include "../../include/mira_gpu.h"

struct GraphCol {
    const unsigned char *p;
    uint32_t kind, pad;
};

// resolved stream: per calculation  [op | nparts << 8] [dst slot] [sources...]; INTERMEDIATE payloads are
// slots; GRAPH_SRC_PREV = the value of the calculation just before (still in registers -- most
// results of a post-order expression walk are consumed by the very next calculation and never
// touch the workspace)
static constexpr uint32_t GRAPH_SRC_PREV = 4u;
static constexpr uint32_t GRAPH_NO_SLOT = 0xFFFFFFFFu;
template <class FP>
KERNEL void k_graph_eval(const uint32_t *__restrict__ code, uint32_t ncalc, const unsigned char *__restrict__ consts,
                         const unsigned char *__restrict__ challenges, const int32_t *__restrict__ rotations,
                         const GraphCol *__restrict__ cols, uint64_t nrows, unsigned char *__restrict__ ws, unsigned char *__restrict__ out) {
    const uint64_t T = (uint64_t)gridDim.x * blockDim.x, lane = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t row = lane; row < nrows; row += T) {
        Fe<FP> v = fe_zero<FP>(), prev;
        auto fetch = [&](uint32_t s) -> Fe<FP> {
            const uint32_t kind = s >> 29, payload = s & 0x1FFFFFFFu;
            if (kind == GRAPH_SRC_PREV) return prev;
            if (kind == MIRA_SRC_CONSTANT) return fe_load<FP>(consts + (size_t)payload * 32);
            if (kind == MIRA_SRC_INTERMEDIATE) return fe_load<FP>(ws + ((size_t)payload * T + lane) * 32);
            if (kind == MIRA_SRC_CHALLENGE) return fe_load<FP>(challenges + (size_t)payload * 32);
            const GraphCol c = cols[payload & 0xFFFFFu];
            int64_t r = ((int64_t)row + rotations[payload >> 20]) % (int64_t)nrows;   // rem_euclid, graph_evaluator.rs:51-53
            if (r < 0) r += (int64_t)nrows;
            if (c.kind == MIRA_COL_BOOL) return c.p[r] ? fe_one<FP>() : fe_zero<FP>();   // selector, src/plonk/eval.rs:62
            return fe_load<FP>(c.p + (size_t)r * 32);
        };
        const uint32_t *pc = code;
        for (uint32_t i = 0; i < ncalc; i++) {
            const uint32_t head = pc[0], dst = pc[1];
            prev = v;
            const uint32_t op = head & 0xFFu, nparts = head >> 8;
            pc += 2;
            if (op == MIRA_OP_ADD) { v = fe_add(fetch(pc[0]), fetch(pc[1])); pc += 2; }
            else if (op == MIRA_OP_SUB) { v = fe_sub(fetch(pc[0]), fetch(pc[1])); pc += 2; }
            else if (op == MIRA_OP_MUL) { v = fe_mul(fetch(pc[0]), fetch(pc[1])); pc += 2; }
            else if (op == MIRA_OP_SQUARE) { v = fe_sqr(fetch(pc[0])); pc += 1; }
            else if (op == MIRA_OP_DOUBLE) { v = fe_dbl(fetch(pc[0])); pc += 1; }
            else if (op == MIRA_OP_NEGATE) { v = fe_neg(fetch(pc[0])); pc += 1; }
            else if (op == MIRA_OP_STORE) { v = fetch(pc[0]); pc += 1; }
            else {                                           // HORNER: start, factor, parts[] (graph_evaluator.rs:148-155)
                v = fetch(pc[0]);
                const Fe<FP> f = fetch(pc[1]);
                for (uint32_t k = 0; k < nparts; k++) v = fe_add(fe_mul(v, f), fetch(pc[2 + k]));
                pc += 2 + nparts;
            }
            if (dst != GRAPH_NO_SLOT) fe_store(ws + ((size_t)dst * T + lane) * 32, v);   // no slot: nobody reads it again
        }
        fe_store(out + row * 32, v);
    }
}
