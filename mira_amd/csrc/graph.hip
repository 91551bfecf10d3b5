// Host side of the cross-term evaluator (graph_kernels.cuh): validates the flattened
// GraphEvaluator (include/mira_gpu.h), allocates workspace slots for its intermediates and
// launches one lane per row.
#include "ctx.h"
#include "graph_kernels.cuh"

namespace {

struct Calc {
    uint32_t op, nparts;
    size_t first_src, nsrc;   // into the flat source list
};

// number of operand words of a calculation, or -1 for an unknown opcode
int operand_count(uint32_t op, uint32_t nparts) {
    switch (op) {
        case MIRA_OP_ADD: case MIRA_OP_SUB: case MIRA_OP_MUL: return 2;
        case MIRA_OP_SQUARE: case MIRA_OP_DOUBLE: case MIRA_OP_NEGATE: case MIRA_OP_STORE: return 1;
        case MIRA_OP_HORNER: return 2 + (int)nparts;
        default: return -1;
    }
}

}   // namespace

// The reference keeps one intermediate per calculation (graph_evaluator.rs:354-359).  Most die
// young: slots are handed out by last use, so a 300-calculation gate needs ~10-20 of them.
int graph_eval_device(int field, const mira_graph *gr, const mira_eval_column *columns, uint32_t num_columns, const uint64_t *challenges,
                      uint32_t num_challenges, size_t num_rows, void *d_out) {
    int rc;
    const uint32_t n = gr->num_calculations;
    std::vector<Calc> calcs;
    std::vector<uint32_t> srcs;
    calcs.reserve(n);
    size_t pos = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (pos >= gr->code_words) { set_error("graph code ends inside calculation " + std::to_string(i)); return MIRA_E_BAD_ARG; }
        const uint32_t head = gr->code[pos++];
        const uint32_t op = head & 0xFFu, nparts = head >> 8;
        const int cnt = operand_count(op, nparts);
        if (cnt < 0 || (op != MIRA_OP_HORNER && nparts != 0)) { set_error("unknown calculation " + std::to_string(head) + " at index " + std::to_string(i)); return MIRA_E_BAD_ARG; }
        if (pos + (size_t)cnt > gr->code_words) { set_error("graph code ends inside calculation " + std::to_string(i)); return MIRA_E_BAD_ARG; }
        calcs.push_back(Calc{op, nparts, srcs.size(), (size_t)cnt});
        for (int k = 0; k < cnt; k++) {
            const uint32_t s = gr->code[pos++];
            const uint32_t kind = s >> 29, payload = s & 0x1FFFFFFFu;
            if (kind == MIRA_SRC_CONSTANT) {
                if (payload >= gr->num_constants) { set_error("constant index out of boundary: " + std::to_string(payload)); return MIRA_E_BAD_ARG; }
            } else if (kind == MIRA_SRC_INTERMEDIATE) {
                if (payload >= i) { set_error("calculation " + std::to_string(i) + " reads intermediate " + std::to_string(payload) + " before it is written"); return MIRA_E_BAD_ARG; }
            } else if (kind == MIRA_SRC_CHALLENGE) {
                if (payload >= num_challenges) {
                    set_error("challenge index out of boundary: " + std::to_string(payload));   // EvalError::ChallengeIndexOutOfBoundary
                    return MIRA_E_BAD_ARG;
                }
            } else if (kind == MIRA_SRC_COLUMN) {
                const uint32_t col = payload & 0xFFFFFu, rot = payload >> 20;
                if (col >= num_columns || !columns[col].d_data) {
                    set_error("column variable index out of boundary: " + std::to_string(col));   // EvalError::ColumnVariableIndexOutOfBoundary
                    return MIRA_E_BAD_ARG;
                }
                if (columns[col].kind != MIRA_COL_FIELD && columns[col].kind != MIRA_COL_BOOL) { set_error("unknown column kind"); return MIRA_E_BAD_ARG; }
                if (rot >= gr->num_rotations) { set_error("rotation index out of boundary: " + std::to_string(rot)); return MIRA_E_BAD_ARG; }
            } else {
                set_error("unknown value source kind " + std::to_string(kind));
                return MIRA_E_BAD_ARG;
            }
            srcs.push_back(s);
        }
    }
    if (pos != gr->code_words) { set_error("graph code has trailing words"); return MIRA_E_BAD_ARG; }
    if (num_rows == 0) return MIRA_OK;
    if (num_rows > ((size_t)1 << 31)) { set_error("num_rows > 2^31"); return MIRA_E_UNSUPPORTED; }
    if (n == 0) {                                            // Ok(F::ZERO), graph_evaluator.rs:386-389
        RT_CHECK(rt_memset(d_out, 0, num_rows * 32, g.stream));
        RT_CHECK(rt_sync(g.stream));
        return MIRA_OK;
    }

    // readers of every intermediate; the final calculation's value leaves through `out`
    std::vector<uint32_t> last_use(n, 0), first_use(n, 0xFFFFFFFFu);
    for (uint32_t i = 0; i < n; i++)
        for (size_t k = 0; k < calcs[i].nsrc; k++) {
            const uint32_t s = srcs[calcs[i].first_src + k];
            if ((s >> 29) != MIRA_SRC_INTERMEDIATE) continue;
            const uint32_t t = s & 0x1FFFFFFFu;
            last_use[t] = i;
            if (first_use[t] == 0xFFFFFFFFu) first_use[t] = i;
        }
    // a value read only by the next calculation is forwarded in registers; the rest get a slot
    // from their definition to their last reader
    auto used = [&](uint32_t t) { return first_use[t] != 0xFFFFFFFFu; };
    auto forwarded = [&](uint32_t t) { return used(t) && first_use[t] == t + 1 && last_use[t] == t + 1; };
    std::vector<uint32_t> slot_of(n, GRAPH_NO_SLOT), free_slots, stream;
    std::vector<std::vector<uint32_t>> dying(n);
    for (uint32_t t = 0; t < n; t++)
        if (used(t) && !forwarded(t)) dying[last_use[t]].push_back(t);
    uint32_t nslots = 0;
    for (uint32_t i = 0; i < n; i++) {
        stream.push_back(calcs[i].op | calcs[i].nparts << 8);
        const size_t dst_at = stream.size();
        stream.push_back(GRAPH_NO_SLOT);
        for (size_t k = 0; k < calcs[i].nsrc; k++) {
            uint32_t s = srcs[calcs[i].first_src + k];
            if ((s >> 29) == MIRA_SRC_INTERMEDIATE) {
                const uint32_t t = s & 0x1FFFFFFFu;
                s = forwarded(t) ? (GRAPH_SRC_PREV << 29) : ((MIRA_SRC_INTERMEDIATE << 29) | slot_of[t]);
            }
            stream.push_back(s);
        }
        // operands are in registers before the result is written: a slot that dies here can take it
        for (uint32_t t : dying[i]) free_slots.push_back(slot_of[t]);
        if (used(i) && !forwarded(i)) {
            if (free_slots.empty()) free_slots.push_back(nslots++);
            slot_of[i] = free_slots.back();
            free_slots.pop_back();
            stream[dst_at] = slot_of[i];
        }
    }

    const uint32_t block = 256;
    const uint32_t grid = (uint32_t)std::min<size_t>((num_rows + block - 1) / block, 256 * 4);
    const size_t T = (size_t)grid * block;
    // one staging area: code | constants | challenges | rotations | columns
    auto align16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_code = 0, o_const = align16(o_code + stream.size() * 4), o_chal = align16(o_const + (size_t)gr->num_constants * 32),
                 o_rot = align16(o_chal + (size_t)num_challenges * 32), o_cols = align16(o_rot + (size_t)gr->num_rotations * 4),
                 total = align16(o_cols + (size_t)num_columns * sizeof(GraphCol));
    std::vector<unsigned char> stage(total, 0);
    memcpy(stage.data() + o_code, stream.data(), stream.size() * 4);
    if (gr->num_constants) memcpy(stage.data() + o_const, gr->constants, (size_t)gr->num_constants * 32);
    if (num_challenges) memcpy(stage.data() + o_chal, challenges, (size_t)num_challenges * 32);
    if (gr->num_rotations) memcpy(stage.data() + o_rot, gr->rotations, (size_t)gr->num_rotations * 4);
    for (uint32_t c = 0; c < num_columns; c++) {
        GraphCol gc{reinterpret_cast<const unsigned char *>(columns[c].d_data), columns[c].kind, 0};
        memcpy(stage.data() + o_cols + (size_t)c * sizeof(GraphCol), &gc, sizeof gc);
    }
    if ((rc = g.graph_consts.ensure(total))) return rc;
    if ((rc = g.graph_ws.ensure(std::max<size_t>(1, nslots) * T * 32))) return rc;
    RT_CHECK(rt_h2d(g.graph_consts.p, stage.data(), total, g.stream));
    RT_CHECK(rt_sync(g.stream));                             // `stage` is pageable host memory about to go out of scope
    const unsigned char *base = reinterpret_cast<const unsigned char *>(g.graph_consts.p);
    tm_begin();
    if (field == MIRA_FIELD_FQ)
        LAUNCH(k_graph_eval<FqP>, grid, block, 0, g.stream, reinterpret_cast<const uint32_t *>(base + o_code), n, base + o_const, base + o_chal,
               reinterpret_cast<const int32_t *>(base + o_rot), reinterpret_cast<const GraphCol *>(base + o_cols), (uint64_t)num_rows,
               reinterpret_cast<unsigned char *>(g.graph_ws.p), reinterpret_cast<unsigned char *>(d_out));
    else
        LAUNCH(k_graph_eval<FrP>, grid, block, 0, g.stream, reinterpret_cast<const uint32_t *>(base + o_code), n, base + o_const, base + o_chal,
               reinterpret_cast<const int32_t *>(base + o_rot), reinterpret_cast<const GraphCol *>(base + o_cols), (uint64_t)num_rows,
               reinterpret_cast<unsigned char *>(g.graph_ws.p), reinterpret_cast<unsigned char *>(d_out));
    tm_mark("graph_eval");
    RT_CHECK(rt_last());
    RT_CHECK(rt_sync(g.stream));
    tm_end();
    return MIRA_OK;
}
