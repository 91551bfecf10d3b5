// Host side of the cross-term evaluator (graph_kernels.cuh).  A flattened GraphEvaluator
// (include/mira_gpu.h) is COMPILED once per circuit -- validated; constants read at their uses; products
// folded into the sums that take them; the power of two every value carries chosen so that no column
// has to be lifted into the multiplier's form; intermediates given workspace slots, the most used ones
// in LDS; every value given a proven bound; the instruction stream and constants uploaded -- and the
// handle is then evaluated for any number of (columns, challenges) pairs, alone or with the other graphs
// of a fold step, with one small upload and one launch.
#include "ctx.h"
#include "graph_kernels.cuh"
#include "graph_jit.hpp"
#include "host_field.hpp"
#ifndef MIRA_CPU_EMU
#include <dlfcn.h>
#include <unistd.h>
#include <thread>
#endif

namespace {

struct Calc {
    uint32_t op, nparts;
    size_t first_src, nsrc;   // into the flat source list
};

// compiler-internal calculation: addend + p * q (an ADD that absorbed the single-use MUL feeding it); never accepted from a caller
constexpr uint32_t OP_MAC_INTERNAL = 0xFEu;

// number of operand words of a calculation, or -1 for an unknown opcode
int operand_count(uint32_t op, uint32_t nparts) {
    switch (op) {
        case MIRA_OP_ADD: case MIRA_OP_SUB: case MIRA_OP_MUL: return 2;
        case MIRA_OP_SQUARE: case MIRA_OP_DOUBLE: case MIRA_OP_NEGATE: case MIRA_OP_STORE: return 1;
        case MIRA_OP_HORNER: return 2 + (int)nparts;
        case OP_MAC_INTERNAL: return 3;
        default: return -1;
    }
}

// Forms.  The kernel's multiplier divides by 2^261, so a value x is carried as x * 2^(261 - 5 f) for some
// integer f, its FORM: f = 0 is the multiplier's own Montgomery form (closed under multiplication),
// f = 1 is the reference's memory layout x * 2^256 -- a column as it is read, no lifting product --
// and the product of forms f1 and f2 has form f1 + f2; sums need equal forms.  Constants and
// challenges are converted on the host to whatever form their use wants.
// reference form (x * 2^256, 4 x u64) -> x * 2^(261 - 5 form) as 9 x 29-bit limbs, canonical
template <class FP> void to_limbs29(const uint64_t in[4], int form, uint32_t out[9]) {
    hostf::HFe<FP> s;
    memcpy(s.l, in, 32);
    static hostf::HFe<FP> p32 = hostf::from_u64<FP>(32), i32 = hostf::inv(hostf::from_u64<FP>(32));
    for (int k = form; k < 1; k++) s = hostf::mul(s, p32);       // 2^256 -> 2^(256 + 5 (1 - form))
    for (int k = 1; k < form; k++) s = hostf::mul(s, i32);
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit / 64, sh = bit % 64;
        uint64_t v = s.l[w] >> sh;
        if (sh > 35 && w + 1 < 4) v |= s.l[w + 1] << (64 - sh);
        out[i] = (uint32_t)(v & 0x1FFFFFFFu);
    }
}
void to_limbs29(int field, const uint64_t in[4], int form, uint32_t out[9]) {
    if (field == MIRA_FIELD_FQ) to_limbs29<FqP>(in, form, out); else to_limbs29<FrP>(in, form, out);
}
void one_raw(int field, uint64_t out[4]) {                      // 1 in the reference form
    if (field == MIRA_FIELD_FQ) { auto o = hostf::one<FqP>(); memcpy(out, o.l, 32); } else { auto o = hostf::one<FrP>(); memcpy(out, o.l, 32); }
}

struct Program {
    int field = 0;
    uint32_t ninstr = 0, nslots = 0, num_challenges = 0, num_columns = 0, num_rotations = 0, num_calculations = 0;
    std::vector<uint32_t> used_columns;        // column indices the code reads
    std::vector<std::pair<uint32_t, int>> chal_vars;   // (challenge, form) of every challenge operand: converted per evaluation
    void *d_static = nullptr;                  // code | constants | rotations
    size_t o_code = 0, o_const = 0, o_rot = 0;
    std::vector<uint32_t> h_stream;            // the instruction stream and rotations again on the host: what graph_jit.hpp writes out as a kernel
    std::vector<int32_t> h_rot;
#ifndef MIRA_CPU_EMU
    hipModule_t jit_mod = nullptr;             // the specialised kernel of this program (mira_graph_specialize), or null: interpreted
    hipFunction_t jit_fn = nullptr;
#endif
    std::vector<uint32_t> jit_kinds;           // the column kinds that kernel was built for (an evaluation with others is interpreted)
    DevBuf dyn;                                // challenges | column table of the current evaluation
    unsigned char *h_dyn = nullptr;            // pinned staging of the same
    size_t o_chal = 0, o_cols = 0, o_jobs = 0, dyn_bytes = 0;   // | job table of the batch this program leads
};
std::map<uint64_t, Program> g_programs;
uint32_t g_jit_last_compiled = 0, g_jit_last_from_disk = 0;   // of the last mira_graph_specialize: kernels compiled / read from the cache directory
constexpr uint32_t GRAPH_MAX_BATCH = 16;           // graphs per launch
// Intermediates kept in LDS per workgroup (9 KiB each at 256 lanes).  Measured at k = 17: with a batch
// that fills the wave slots two slots are best (eleven graphs 1.98 -> 1.87 ms; eight cost occupancy, 2.14),
// a lone graph of two workgroups per CU takes eight (0.280 -> 0.247 ms).
constexpr uint32_t GRAPH_LDS_SLOTS_BATCH = 2, GRAPH_LDS_SLOTS_LONE = 8;

size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }
int ensure_dyn(Program &pg, size_t bytes) {
    if (bytes <= pg.dyn_bytes && pg.h_dyn) return MIRA_OK;
    if (pg.h_dyn) (void)rt_host_free(pg.h_dyn);
    pg.h_dyn = nullptr; pg.dyn_bytes = 0;
    int rc = pg.dyn.ensure(bytes);
    if (rc) return rc;
    if (rt_host_alloc(reinterpret_cast<void **>(&pg.h_dyn), bytes) != hipSuccess) { pg.h_dyn = nullptr; set_error("pinned allocation for the compiled graph failed"); return MIRA_E_ALLOC; }
    pg.dyn_bytes = bytes;
    return MIRA_OK;
}

}   // namespace

// The reference keeps one intermediate per calculation (graph_evaluator.rs:354-359).  Most die
// young: slots are handed out by last use, so a 300-calculation gate needs ~10-20 of them.
int graph_compile(int field, const mira_graph *gr, uint32_t num_challenges, uint32_t num_columns, uint64_t *handle_out) {
    const uint32_t n_in = gr->num_calculations;
    std::vector<Calc> calcs;
    std::vector<uint32_t> srcs;
    std::vector<bool> col_used(num_columns, false);
    calcs.reserve(n_in);
    size_t pos = 0;
    for (uint32_t i = 0; i < n_in; i++) {
        if (pos >= gr->code_words) { set_error("graph code ends inside calculation " + std::to_string(i)); return MIRA_E_BAD_ARG; }
        const uint32_t head = gr->code[pos++];
        const uint32_t op = head & 0xFFu, nparts = head >> 8;
        const int cnt = op == OP_MAC_INTERNAL ? -1 : operand_count(op, nparts);
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
                if (col >= num_columns) {
                    set_error("column variable index out of boundary: " + std::to_string(col));   // EvalError::ColumnVariableIndexOutOfBoundary
                    return MIRA_E_BAD_ARG;
                }
                if (rot >= gr->num_rotations) { set_error("rotation index out of boundary: " + std::to_string(rot)); return MIRA_E_BAD_ARG; }
                col_used[col] = true;
            } else {
                set_error("unknown value source kind " + std::to_string(kind));
                return MIRA_E_BAD_ARG;
            }
            srcs.push_back(s);
        }
    }
    if (pos != gr->code_words) { set_error("graph code has trailing words"); return MIRA_E_BAD_ARG; }

    // Constants and challenges that the reference copies into intermediates (Store, graph_evaluator.rs:261-279)
    // are read at their uses instead: as direct operands the host can hand each use the form it wants.
    {
        std::vector<uint32_t> alias(n_in, 0xFFFFFFFFu), new_index(n_in, 0);
        for (uint32_t i = 0; i + 1 < n_in; i++) {                // the last calculation is the result: it stays
            const uint32_t w = srcs[calcs[i].first_src];
            if (calcs[i].op == MIRA_OP_STORE && ((w >> 29) == MIRA_SRC_CONSTANT || (w >> 29) == MIRA_SRC_CHALLENGE)) alias[i] = w;
        }
        std::vector<Calc> calcs2;
        std::vector<uint32_t> srcs2;
        for (uint32_t i = 0; i < n_in; i++) {
            if (alias[i] != 0xFFFFFFFFu) continue;
            new_index[i] = (uint32_t)calcs2.size();
            calcs2.push_back(Calc{calcs[i].op, calcs[i].nparts, srcs2.size(), calcs[i].nsrc});
            for (size_t k = 0; k < calcs[i].nsrc; k++) {
                uint32_t w = srcs[calcs[i].first_src + k];
                if ((w >> 29) == MIRA_SRC_INTERMEDIATE) {
                    const uint32_t t = w & 0x1FFFFFFFu;
                    w = alias[t] != 0xFFFFFFFFu ? alias[t] : ((MIRA_SRC_INTERMEDIATE << 29) | new_index[t]);
                }
                srcs2.push_back(w);
            }
        }
        calcs.swap(calcs2);
        srcs.swap(srcs2);
    }
    const uint32_t n_mid = (uint32_t)calcs.size();

    // Multiply-accumulate fusion.  Gates are sums of products: `acc = acc + c_i * x_i` flattens to MUL, ADD
    // pairs whose product is read once, by the ADD.  Folding the MUL into the ADD (one instruction
    // addend + p * q) halves the instruction count of such chains, and the running sum then stays in the
    // forwarding register from link to link instead of going through a workspace slot while the
    // product is computed.  Not when an operand of the MUL is itself a forwarded value (it would need
    // a slot instead); values are exact field elements, so regrouping changes no result.
    {
        std::vector<uint32_t> nuses(n_mid, 0);
        for (uint32_t i = 0; i < n_mid; i++)
            for (size_t k = 0; k < calcs[i].nsrc; k++) {
                const uint32_t w = srcs[calcs[i].first_src + k];
                if ((w >> 29) == MIRA_SRC_INTERMEDIATE) nuses[w & 0x1FFFFFFFu]++;
            }
        std::vector<uint32_t> absorbed(n_mid, 0xFFFFFFFFu);          // MUL j -> the ADD that takes it
        std::vector<int> takes(n_mid, -1);                            // ADD i -> which of its operands is the absorbed MUL
        for (uint32_t i = 0; i < n_mid; i++) {
            if (calcs[i].op != MIRA_OP_ADD) continue;
            int best = -1;
            uint32_t best_j = 0;
            for (int k = 0; k < 2; k++) {
                const uint32_t w = srcs[calcs[i].first_src + k];
                if ((w >> 29) != MIRA_SRC_INTERMEDIATE) continue;
                const uint32_t j = w & 0x1FFFFFFFu;
                if (calcs[j].op != MIRA_OP_MUL || nuses[j] != 1 || absorbed[j] != 0xFFFFFFFFu) continue;
                bool ok = true;
                for (size_t q = 0; q < 2; q++) {
                    const uint32_t o = srcs[calcs[j].first_src + q];
                    if ((o >> 29) == MIRA_SRC_INTERMEDIATE && (o & 0x1FFFFFFFu) + 1 == j && nuses[o & 0x1FFFFFFFu] == 1) ok = false;
                }
                if (ok && (best < 0 || j > best_j)) { best = k; best_j = j; }
            }
            if (best >= 0) { takes[i] = best; absorbed[best_j] = i; }
        }
        std::vector<uint32_t> new_index(n_mid, 0);
        std::vector<Calc> calcs2;
        std::vector<uint32_t> srcs2;
        auto remap = [&](uint32_t w) { return (w >> 29) == MIRA_SRC_INTERMEDIATE ? ((MIRA_SRC_INTERMEDIATE << 29) | new_index[w & 0x1FFFFFFFu]) : w; };
        for (uint32_t i = 0; i < n_mid; i++) {
            if (absorbed[i] != 0xFFFFFFFFu) continue;
            new_index[i] = (uint32_t)calcs2.size();
            if (takes[i] >= 0) {
                const uint32_t j = srcs[calcs[i].first_src + takes[i]] & 0x1FFFFFFFu;
                calcs2.push_back(Calc{OP_MAC_INTERNAL, 0, srcs2.size(), 3});
                srcs2.push_back(remap(srcs[calcs[i].first_src + 1 - takes[i]]));
                srcs2.push_back(remap(srcs[calcs[j].first_src]));
                srcs2.push_back(remap(srcs[calcs[j].first_src + 1]));
            } else {
                calcs2.push_back(Calc{calcs[i].op, calcs[i].nparts, srcs2.size(), calcs[i].nsrc});
                for (size_t k = 0; k < calcs[i].nsrc; k++) srcs2.push_back(remap(srcs[calcs[i].first_src + k]));
            }
        }
        calcs.swap(calcs2);
        srcs.swap(srcs2);
    }
    const uint32_t n = (uint32_t)calcs.size();

    // ---- forms (see to_limbs29): which form should every calculation's value have, so that as few
    // conversions (one multiplication each) as possible are needed?  Columns are form 1, constants and
    // challenges are free, a product's form is the sum of its factors' forms, a sum's operands must agree,
    // and the result may leave in any form (the last instruction converts and reduces it anyway).
    // Values whose form can still slide -- products with a free factor, and sums of such -- are elements
    // of a union-find with potentials: form = value(element) + offset; meeting a fixed form pins a whole
    // group.  What cannot be reconciled (a shared subexpression wanted in two forms, ...) is converted
    // where it is used.
    struct Desc { int kind, f; uint32_t e; };                // kind 0: free (constant / challenge), 1: fixed form f, 2: sliding, form = val(e) + f
    std::vector<uint32_t> uf_parent;
    std::vector<int> uf_pot, uf_var;                         // pot = val(x) - val(parent); var of a resolved root
    std::vector<char> uf_res;
    auto uf_new = [&]() { uf_parent.push_back((uint32_t)uf_parent.size()); uf_pot.push_back(0); uf_var.push_back(0); uf_res.push_back(0); return (uint32_t)uf_parent.size() - 1; };
    auto uf_find = [&](uint32_t x, int &pot) {                 // root of x, pot = val(x) - val(root)
        pot = 0;
        uint32_t r = x;
        while (uf_parent[r] != r) { pot += uf_pot[r]; r = uf_parent[r]; }
        uint32_t y = x; int acc = pot;                       // path compression
        while (uf_parent[y] != y) { const uint32_t nx = uf_parent[y]; const int py = uf_pot[y]; uf_parent[y] = r; uf_pot[y] = acc; acc -= py; y = nx; }
        return r;
    };
    auto uf_pin = [&](uint32_t e, int value) {                 // val(e) := value unless the group is pinned already
        int pot; const uint32_t r = uf_find(e, pot);
        if (!uf_res[r]) { uf_res[r] = 1; uf_var[r] = value - pot; }
    };
    auto uf_pinned = [&](uint32_t e, int &value) { int pot; const uint32_t r = uf_find(e, pot); value = uf_var[r] + pot; return (bool)uf_res[r]; };
    auto uf_unite = [&](uint32_t a2, uint32_t b2, int d) {      // val(b2) = val(a2) + d, if both groups can still move
        int pa, pb; const uint32_t ra = uf_find(a2, pa), rb = uf_find(b2, pb);
        if (ra == rb) return;
        if (uf_res[ra] && uf_res[rb]) return;
        if (uf_res[rb]) { uf_parent[ra] = rb; uf_pot[ra] = pb - d - pa; }       // val(ra) = val(rb) + pb - d - pa
        else { uf_parent[rb] = ra; uf_pot[rb] = pa + d - pb; }
    };
    auto final_form = [&](const Desc &x) { if (x.kind == 1) return x.f; int v; uf_pinned(x.e, v); return v + x.f; };   // unpinned groups sit at var 0
    auto as_fixed = [&](Desc &x, int want) {                  // pin a sliding value so that its form is `want` (if still possible)
        if (x.kind == 2) { uf_pin(x.e, want - x.f); x = Desc{1, final_form(x), 0}; }
    };
    auto add_rule = [&](Desc x, Desc y) -> Desc {
        if (x.kind == 0 && y.kind == 0) return Desc{2, 0, uf_new()};
        if (x.kind == 0) return y;
        if (y.kind == 0) return x;
        int v;
        if (x.kind == 2 && uf_pinned(x.e, v)) x = Desc{1, v + x.f, 0};
        if (y.kind == 2 && uf_pinned(y.e, v)) y = Desc{1, v + y.f, 0};
        if (x.kind == 1 && y.kind == 1) return x;
        if (x.kind == 1) { as_fixed(y, x.f); return x; }
        if (y.kind == 1) { as_fixed(x, y.f); return y; }
        uf_unite(x.e, y.e, x.f - y.f);                         // val(y.e) + y.f = val(x.e) + x.f
        return x;
    };
    auto mul_rule = [&](Desc x, Desc y) -> Desc {
        if (x.kind == 0 || y.kind == 0) return Desc{2, 0, uf_new()};           // a free factor: the product can have any form
        int v;
        if (x.kind == 2 && uf_pinned(x.e, v)) x = Desc{1, v + x.f, 0};
        if (y.kind == 2 && uf_pinned(y.e, v)) y = Desc{1, v + y.f, 0};
        if (x.kind == 2 && y.kind == 2) as_fixed(y, 0);
        if (x.kind == 1 && y.kind == 1) return Desc{1, x.f + y.f, 0};
        if (x.kind == 1) return Desc{2, y.f + x.f, y.e};
        return Desc{2, x.f + y.f, x.e};
    };
    std::vector<Desc> desc(n);
    auto operand_desc = [&](uint32_t w) -> Desc {
        const uint32_t kind = w >> 29;
        if (kind == MIRA_SRC_INTERMEDIATE) return desc[w & 0x1FFFFFFFu];
        if (kind == MIRA_SRC_COLUMN) return Desc{1, 1, 0};
        return Desc{0, 0, 0};
    };
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t *w = srcs.data() + calcs[i].first_src;
        Desc r;
        switch (calcs[i].op) {
            case MIRA_OP_ADD: case MIRA_OP_SUB: r = add_rule(operand_desc(w[0]), operand_desc(w[1])); break;
            case MIRA_OP_MUL: r = mul_rule(operand_desc(w[0]), operand_desc(w[1])); break;
            case MIRA_OP_SQUARE: { Desc x = operand_desc(w[0]); if (x.kind == 0) r = Desc{2, 0, uf_new()}; else { as_fixed(x, 0); r = Desc{1, 2 * final_form(x), 0}; } break; }
            case MIRA_OP_DOUBLE: case MIRA_OP_NEGATE: case MIRA_OP_STORE: { Desc x = operand_desc(w[0]); r = x.kind == 0 ? Desc{2, 0, uf_new()} : x; break; }
            case OP_MAC_INTERNAL: r = add_rule(operand_desc(w[0]), mul_rule(operand_desc(w[1]), operand_desc(w[2]))); break;
            default:                                             // HORNER: value = value * factor + part
                r = operand_desc(w[0]);
                if (r.kind == 0) r = Desc{2, 0, uf_new()};
                for (uint32_t k = 0; k < calcs[i].nparts; k++) r = add_rule(mul_rule(r, operand_desc(w[1])), operand_desc(w[2 + k]));
                break;
        }
        desc[i] = r;
    }
    std::vector<int> form_of(n);
    for (uint32_t i = 0; i < n; i++) form_of[i] = final_form(desc[i]);

    // readers of every intermediate; the final calculation's value leaves through `out`
    std::vector<uint32_t> last_use(n, 0), first_use(n, 0xFFFFFFFFu);
    for (uint32_t i = 0; i < n; i++)
        for (size_t k = 0; k < calcs[i].nsrc; k++) {
            const uint32_t s2 = srcs[calcs[i].first_src + k];
            if ((s2 >> 29) != MIRA_SRC_INTERMEDIATE) continue;
            const uint32_t t = s2 & 0x1FFFFFFFu;
            last_use[t] = i;
            if (first_use[t] == 0xFFFFFFFFu) first_use[t] = i;
        }
    // a value read only by the next calculation is forwarded in registers; the rest get a slot
    // from their definition to their last reader.  (HORNER expands to several instructions, each of
    // which moves the forwarding register on: its operands always come from slots.)
    auto used = [&](uint32_t t) { return first_use[t] != 0xFFFFFFFFu; };
    auto reads_of = [&](uint32_t i, uint32_t t) {              // how often calculation i reads intermediate t
        uint32_t c = 0;
        for (size_t k = 0; k < calcs[i].nsrc; k++) c += srcs[calcs[i].first_src + k] == ((MIRA_SRC_INTERMEDIATE << 29) | t);
        return c;
    };
    // (read ONCE: a form conversion of one operand moves the forwarding register on, a second read would see the converted value)
    auto forwarded = [&](uint32_t t) {
        return used(t) && first_use[t] == t + 1 && last_use[t] == t + 1 && calcs[t + 1].op != MIRA_OP_HORNER && reads_of(t + 1, t) == 1;
    };
    std::vector<uint32_t> slot_of(n, GRAPH_NO_SLOT), free_slots, stream;
    std::vector<double> bound_of(n, 0.0);                    // proven bound of every calculation's value, in multiples of P
    std::vector<std::vector<uint32_t>> dying(n);
    for (uint32_t t = 0; t < n; t++)
        if (used(t) && !forwarded(t)) dying[last_use[t]].push_back(t);
    uint32_t nslots = 0, ninstr = 0;
    size_t last_head = 0;                                    // stream index of the most recent instruction
    auto bcode = [](double b) { return (uint32_t)std::min(65535.0, std::ceil(b * 256.0)); };
    // one compiled instruction on resolved sources sa, sb with proven bounds ba, bb; returns the bound of the result
    auto emit = [&](uint32_t op, uint32_t sa, double ba, uint32_t sb, double bb) -> double {
        uint32_t K = 0;
        double rb = 0;
        const bool binary = op == GOP_ADD || op == GOP_SUB || op == GOP_MUL;
        auto bias = [](double b2) { return b2 < 1.99 ? 2u : b2 < 3.99 ? 4u : b2 < 7.99 ? 8u : 16u; };   // f29_sub<K> needs the subtrahend below K P
        switch (op) {
            case GOP_ADD: rb = ba + bb; break;
            case GOP_SUB: K = bias(bb); rb = ba + K; break;
            case GOP_NEG: K = bias(ba); rb = K; break;
            case GOP_MUL: rb = ba * bb / 168.9 + 1.0; break;
            case GOP_SQR: rb = ba * ba / 168.9 + 1.0; break;
            case GOP_DBL: rb = 2 * ba; break;
            case GOP_NORM: rb = ba / 168.9 + 1.0; break;
            default: rb = ba; break;
        }
        last_head = stream.size();
        stream.push_back(op | K << 8);
        stream.push_back(GRAPH_NO_SLOT);
        stream.push_back(bcode(ba) | bcode(binary ? bb : 0.0) << 16);
        stream.push_back(sa);
        if (binary) stream.push_back(sb);
        ninstr++;
        return rb;
    };
    // addend + p * q
    auto emit_mac = [&](uint32_t sc, double bc, uint32_t sp, double bp, uint32_t sq, double bq) -> double {
        last_head = stream.size();
        stream.push_back(GOP_MAC);
        stream.push_back(GRAPH_NO_SLOT);
        stream.push_back(bcode(bp) | bcode(bq) << 16);
        stream.push_back(sp);
        stream.push_back(sq);
        stream.push_back(sc);
        stream.push_back(bcode(bc));
        ninstr++;
        return bp * bq / 168.9 + 1.0 + bc;
    };
    const uint32_t PREV = GRAPH_SRC_PREV << 29;
    // constants in the forms their uses want: pool entry = (constant index, or -1 for the number one; form)
    std::vector<std::pair<int, int>> pool;
    auto pool_word = [&](int ci, int form) {
        for (size_t k = 0; k < pool.size(); k++)
            if (pool[k].first == ci && pool[k].second == form) return (uint32_t)((MIRA_SRC_CONSTANT << 29) | k);
        pool.push_back({ci, form});
        return (uint32_t)((MIRA_SRC_CONSTANT << 29) | (pool.size() - 1));
    };
    std::vector<std::pair<uint32_t, int>> chal_vars;
    auto chal_word = [&](uint32_t ch, int form) {
        for (size_t k = 0; k < chal_vars.size(); k++)
            if (chal_vars[k].first == ch && chal_vars[k].second == form) return (uint32_t)((MIRA_SRC_CHALLENGE << 29) | k);
        chal_vars.push_back({ch, form});
        return (uint32_t)((MIRA_SRC_CHALLENGE << 29) | (chal_vars.size() - 1));
    };
    // an operand as the emitter sees it: the word the kernel fetches, a proven bound, and its form (free: any)
    struct Opnd { uint32_t w; double b; int f; bool free, prev; };
    auto materialise = [&](Opnd &x, int form) {               // a free operand in the given form
        if (!x.free) return;
        const uint32_t kind = x.w >> 29, id = x.w & 0x1FFFFFFFu;
        x.w = kind == MIRA_SRC_CONSTANT ? pool_word((int)id, form) : chal_word(id, form);
        x.f = form; x.free = false;
    };
    auto convert_prev = [&](double b, int from, int to) -> double {   // the forwarded value into another form: times the number one in form to - from
        return from == to ? b : emit(GOP_MUL, PREV, b, pool_word(-1, to - from), 1.0);
    };
    auto convert = [&](Opnd &x, int to) {                     // any fixed operand into form `to`: the result is the forwarded value
        x.b = emit(GOP_MUL, x.w, x.b, pool_word(-1, to - x.f), 1.0);
        x.w = PREV; x.prev = true; x.f = to;
    };
    // x (+ / -) y in form `target`; returns the bound, the value is the forwarded one
    auto emit_addsub = [&](uint32_t op, Opnd x, Opnd y, int target) -> double {
        int F;
        if (x.free && y.free) F = target;
        else if (x.free) F = y.f;
        else if (y.free) F = x.f;
        else if (x.f == y.f) F = x.f;
        else if (x.prev) { convert(x, y.f); F = y.f; }        // never convert the OTHER operand while one sits in the forwarding register
        else if (y.prev) { convert(y, x.f); F = x.f; }
        else { convert(y, x.f); F = x.f; }
        materialise(x, F); materialise(y, F);
        return convert_prev(emit(op, x.w, x.b, y.w, y.b), F, target);
    };
    auto emit_mul = [&](Opnd x, Opnd y, int target) -> double {
        if (x.free && y.free) { materialise(x, target); materialise(y, 0); }
        else if (x.free) materialise(x, target - y.f);
        else if (y.free) materialise(y, target - x.f);
        return convert_prev(emit(GOP_MUL, x.w, x.b, y.w, y.b), x.f + y.f, target);
    };
    for (uint32_t i = 0; i < n; i++) {
        // resolve the operands: intermediates become slots or the forwarded register
        std::vector<Opnd> o(calcs[i].nsrc);
        for (size_t k = 0; k < calcs[i].nsrc; k++) {
            const uint32_t w = srcs[calcs[i].first_src + k];
            const uint32_t kind = w >> 29;
            if (kind == MIRA_SRC_INTERMEDIATE) {
                const uint32_t t = w & 0x1FFFFFFFu;
                o[k] = Opnd{forwarded(t) ? PREV : ((MIRA_SRC_INTERMEDIATE << 29) | slot_of[t]), bound_of[t], form_of[t], false, forwarded(t)};
            } else if (kind == MIRA_SRC_COLUMN) {
                o[k] = Opnd{w, 1.0, 1, false, false};         // canonical, in the reference's form
            } else {
                o[k] = Opnd{w, 1.0, 0, true, false};          // constants and challenges are canonical in whatever form they are asked for
            }
        }
        const int T = form_of[i];
        double rb;
        switch (calcs[i].op) {
            case MIRA_OP_ADD: rb = emit_addsub(GOP_ADD, o[0], o[1], T); break;
            case MIRA_OP_SUB: rb = emit_addsub(GOP_SUB, o[0], o[1], T); break;
            case MIRA_OP_MUL: rb = emit_mul(o[0], o[1], T); break;
            case MIRA_OP_SQUARE:
                if (o[0].free) materialise(o[0], T % 2 == 0 ? T / 2 : 0);
                rb = convert_prev(emit(GOP_SQR, o[0].w, o[0].b, 0, 0), 2 * o[0].f, T);
                break;
            case MIRA_OP_DOUBLE: case MIRA_OP_NEGATE: case MIRA_OP_STORE: {
                materialise(o[0], T);
                const uint32_t gop = calcs[i].op == MIRA_OP_DOUBLE ? GOP_DBL : calcs[i].op == MIRA_OP_NEGATE ? GOP_NEG : GOP_COPY;
                rb = convert_prev(emit(gop, o[0].w, o[0].b, 0, 0), o[0].f, T);
                break;
            }
            case OP_MAC_INTERNAL: {                              // o[0] + o[1] * o[2]
                Opnd c = o[0], pq = o[1], q = o[2];
                const bool free_factor = pq.free || q.free;
                const int fp = free_factor ? 0 : pq.f + q.f;     // the product's form, if it is not ours to choose
                if (c.free) {                                    // a constant addend takes the product's form
                    const int F = free_factor ? T : fp;
                    if (pq.free && q.free) { materialise(pq, F); materialise(q, 0); }
                    else if (pq.free) materialise(pq, F - q.f);
                    else if (q.free) materialise(q, F - pq.f);
                    materialise(c, F);
                    rb = convert_prev(emit_mac(c.w, c.b, pq.w, pq.b, q.w, q.b), F, T);
                } else if (free_factor) {                        // the free factor makes the product meet the addend
                    if (pq.free && q.free) { materialise(pq, c.f); materialise(q, 0); }
                    else if (pq.free) materialise(pq, c.f - q.f);
                    else materialise(q, c.f - pq.f);
                    rb = convert_prev(emit_mac(c.w, c.b, pq.w, pq.b, q.w, q.b), c.f, T);
                } else if (fp == c.f) {
                    rb = convert_prev(emit_mac(c.w, c.b, pq.w, pq.b, q.w, q.b), fp, T);
                } else if (c.prev) {                             // the addend is the forwarded value: bring IT to the product's form
                    convert(c, fp);
                    rb = convert_prev(emit_mac(c.w, c.b, pq.w, pq.b, q.w, q.b), fp, T);
                } else {                                         // product first (a factor may be the forwarded value), then the sum
                    const double bp = emit(GOP_MUL, pq.w, pq.b, q.w, q.b);
                    rb = emit_addsub(GOP_ADD, Opnd{PREV, bp, fp, false, true}, c, T);
                }
                break;
            }
            default: {                                           // HORNER: start, factor, parts[] (graph_evaluator.rs:148-155): value = value * factor + part
                materialise(o[0], T);
                rb = emit(GOP_COPY, o[0].w, o[0].b, 0, 0);
                int fv = o[0].f;
                for (uint32_t k = 0; k < calcs[i].nparts; k++) {
                    Opnd part = o[2 + k];
                    const int want = part.free ? T : part.f;     // the product in the form of the part it meets
                    rb = emit_mul(Opnd{PREV, rb, fv, false, true}, o[1], o[1].free ? want : fv + o[1].f);
                    fv = o[1].free ? want : fv + o[1].f;
                    rb = emit_addsub(GOP_ADD, Opnd{PREV, rb, fv, false, true}, part, part.free ? fv : part.f);
                    fv = part.free ? fv : part.f;
                    if (rb > GRAPH_MAX_BOUND) rb = emit(GOP_NORM, PREV, rb, 0, 0);
                }
                rb = convert_prev(rb, fv, T);
                break;
            }
        }
        if (rb > GRAPH_MAX_BOUND) rb = emit(GOP_NORM, PREV, rb, 0, 0);   // keep the invariant: stored and forwarded values < 12 P
        if (i + 1 == n) rb = emit(GOP_MUL, PREV, rb, pool_word(-1, 1 - T), 1.0);   // the result: into the reference's form, below 2 P (the kernel stores it canonical)
        bound_of[i] = rb;
        // operands are in registers before the result is written: a slot that dies here can take it
        for (uint32_t t : dying[i]) free_slots.push_back(slot_of[t]);
        if (used(i) && !forwarded(i)) {
            if (free_slots.empty()) free_slots.push_back(nslots++);
            slot_of[i] = free_slots.back();
            free_slots.pop_back();
            stream[last_head + 1] = slot_of[i];              // the calculation's last instruction writes the slot
        }
    }

    // Renumber the slots by how often the program touches them, most used first: the kernel keeps the
    // lowest-numbered ones in LDS (GRAPH_LDS_SLOTS_*) and the rest in the global workspace.
    {
        auto words_of = [](uint32_t head) { const uint32_t op = head & 0xFFu; return op == GOP_MAC ? 7u : (op == GOP_ADD || op == GOP_SUB || op == GOP_MUL) ? 5u : 4u; };
        auto nsrc_of = [](uint32_t head) { const uint32_t op = head & 0xFFu; return op == GOP_MAC ? 3u : (op == GOP_ADD || op == GOP_SUB || op == GOP_MUL) ? 2u : 1u; };
        std::vector<uint64_t> uses(nslots, 0);
        for (size_t pos = 0; pos < stream.size(); pos += words_of(stream[pos])) {
            if (stream[pos + 1] != GRAPH_NO_SLOT) uses[stream[pos + 1]]++;
            for (uint32_t k = 0; k < nsrc_of(stream[pos]); k++) {
                const uint32_t w = stream[pos + 3 + k];
                if ((w >> 29) == MIRA_SRC_INTERMEDIATE) uses[w & 0x1FFFFFFFu]++;
            }
        }
        std::vector<uint32_t> order(nslots), rank(nslots);
        for (uint32_t i = 0; i < nslots; i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return uses[a] > uses[b]; });
        for (uint32_t i = 0; i < nslots; i++) rank[order[i]] = i;
        for (size_t pos = 0; pos < stream.size(); pos += words_of(stream[pos])) {
            if (stream[pos + 1] != GRAPH_NO_SLOT) stream[pos + 1] = rank[stream[pos + 1]];
            for (uint32_t k = 0; k < nsrc_of(stream[pos]); k++) {
                uint32_t &w = stream[pos + 3 + k];
                if ((w >> 29) == MIRA_SRC_INTERMEDIATE) w = (MIRA_SRC_INTERMEDIATE << 29) | rank[w & 0x1FFFFFFFu];
            }
        }
    }

    Program pg;
    pg.field = field; pg.ninstr = ninstr; pg.nslots = nslots; pg.num_challenges = num_challenges; pg.num_columns = num_columns;
    pg.num_rotations = gr->num_rotations; pg.num_calculations = n_in;
    for (uint32_t c = 0; c < num_columns; c++)
        if (col_used[c]) pg.used_columns.push_back(c);
    // static part on the device: code | constants (9 x 29-bit limbs, multiplier form) | rotations
    pg.o_code = 0;
    pg.o_const = align16(stream.size() * 4);
    pg.o_rot = align16(pg.o_const + pool.size() * 36);
    pg.chal_vars = chal_vars;
    pg.h_stream = stream;
    pg.h_rot.assign(gr->rotations, gr->rotations + gr->num_rotations);
    const size_t total = align16(pg.o_rot + (size_t)gr->num_rotations * 4) + 16;
    std::vector<unsigned char> stage(total, 0);
    memcpy(stage.data() + pg.o_code, stream.data(), stream.size() * 4);
    uint64_t one_r[4];
    one_raw(field, one_r);
    for (size_t k = 0; k < pool.size(); k++)
        to_limbs29(field, pool[k].first < 0 ? one_r : gr->constants + (size_t)pool[k].first * 4, pool[k].second,
                   reinterpret_cast<uint32_t *>(stage.data() + pg.o_const) + k * 9);
    if (gr->num_rotations) memcpy(stage.data() + pg.o_rot, gr->rotations, (size_t)gr->num_rotations * 4);
    if (rt_malloc(&pg.d_static, total) != hipSuccess || !pg.d_static) { set_error("device allocation for the compiled graph failed"); return MIRA_E_ALLOC; }
    RT_CHECK(rt_h2d(pg.d_static, stage.data(), total, g.stream));
    RT_CHECK(rt_sync(g.stream));                             // `stage` is pageable host memory about to go out of scope
    // dynamic part of an evaluation: column table | job table | the challenges of every program of the batch in the forms
    // it reads them, staged in pinned host memory (grown by the evaluation that needs more)
    pg.o_cols = 0;
    pg.o_jobs = align16((size_t)num_columns * sizeof(GraphCol));
    pg.o_chal = align16(pg.o_jobs + (size_t)GRAPH_MAX_BATCH * sizeof(GraphJob));
    int rc = ensure_dyn(pg, pg.o_chal + (chal_vars.size() + 8) * 36 * 4);
    if (rc) { (void)rt_free(pg.d_static); return rc; }
    *handle_out = g.next_handle++;
    g_programs[*handle_out] = pg;
    return MIRA_OK;
}

int graph_free(uint64_t handle) {
    auto it = g_programs.find(handle);
    if (it == g_programs.end()) { set_error("unknown graph handle"); return MIRA_E_BAD_ARG; }
    Program &pg = it->second;
#ifndef MIRA_CPU_EMU
    if (pg.jit_mod) (void)hipModuleUnload(pg.jit_mod);
#endif
    if (pg.d_static) (void)rt_free(pg.d_static);
    if (pg.dyn.p) (void)rt_free(pg.dyn.p);
    if (pg.h_dyn) (void)rt_host_free(pg.h_dyn);
    g_programs.erase(it);
    return MIRA_OK;
}

// count compiled graphs over the same columns and challenges, results to d_outs[k]; one launch per
// GRAPH_MAX_BATCH graphs
int graph_eval_batch(const uint64_t *handles, uint32_t count, const mira_eval_column *columns, uint32_t num_columns, const uint64_t *challenges,
                     uint32_t num_challenges, size_t num_rows, void *const *d_outs) {
    int rc;
    if (count == 0) return MIRA_OK;
    std::vector<Program *> pgs(count);
    for (uint32_t k = 0; k < count; k++) {
        auto it = g_programs.find(handles[k]);
        if (it == g_programs.end()) { set_error("unknown graph handle"); return MIRA_E_BAD_ARG; }
        Program &pg = *(pgs[k] = &it->second);
        if (num_challenges != pg.num_challenges || num_columns != pg.num_columns) {
            set_error("the graph was compiled for " + std::to_string(pg.num_challenges) + " challenges and " + std::to_string(pg.num_columns) + " columns");
            return MIRA_E_BAD_ARG;
        }
        if (pg.field != pgs[0]->field) { set_error("the graphs of a batch must be over one field"); return MIRA_E_BAD_ARG; }
        for (uint32_t col : pg.used_columns) {
            if (!columns[col].d_data) {
                set_error("column variable index out of boundary: " + std::to_string(col));   // EvalError::ColumnVariableIndexOutOfBoundary / InvalidWitnessIndex
                return MIRA_E_BAD_ARG;
            }
            if (columns[col].kind != MIRA_COL_FIELD && columns[col].kind != MIRA_COL_BOOL) { set_error("unknown column kind"); return MIRA_E_BAD_ARG; }
        }
        if (num_rows && !d_outs[k]) { set_error("null output"); return MIRA_E_BAD_ARG; }
    }
    if (num_rows == 0) return MIRA_OK;
    if (num_rows > ((size_t)1 << 31)) { set_error("num_rows > 2^31"); return MIRA_E_UNSUPPORTED; }
    const uint32_t block = 256;
    const uint32_t grid = (uint32_t)std::min<size_t>((num_rows + block - 1) / block, 256 * 4);
    const size_t T = (size_t)grid * block;
    Program &p0 = *pgs[0];                                   // its staging carries the batch's column table, job table and challenges
    {
        size_t worst = 0;                                    // challenge entries of the largest launch
        for (uint32_t done = 0; done < count; done += GRAPH_MAX_BATCH) {
            size_t w = 0;
            for (uint32_t k = done; k < std::min<uint32_t>(count, done + GRAPH_MAX_BATCH); k++) w += pgs[k]->chal_vars.size();
            worst = std::max(worst, w);
        }
        if ((rc = ensure_dyn(p0, p0.o_chal + (worst + 1) * 36))) return rc;
    }
    // the call's column pointers and challenges (each in the forms its program reads it): one small copy from pinned memory
    for (uint32_t c = 0; c < num_columns; c++) {
        GraphCol gc{reinterpret_cast<const unsigned char *>(columns[c].d_data), columns[c].kind, 0};
        memcpy(p0.h_dyn + p0.o_cols + (size_t)c * sizeof(GraphCol), &gc, sizeof gc);
    }
#ifndef MIRA_CPU_EMU
    static bool lds_ready = false;                           // eight LDS slots are 72 KiB, above the 64 KiB default
    if (!lds_ready) {
        RT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_graph_eval<Fq29>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GRAPH_LDS_SLOTS_LONE * 9 * block * 4)));
        RT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_graph_eval<Fr29>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GRAPH_LDS_SLOTS_LONE * 9 * block * 4)));
        lds_ready = true;
    }
#endif
    tm_begin();
#ifndef MIRA_CPU_EMU
    std::vector<std::pair<hipFunction_t, GraphJob>> jit_jobs;
#endif
    for (uint32_t done = 0; done < count; done += GRAPH_MAX_BATCH) {
        const uint32_t cnt = std::min<uint32_t>(GRAPH_MAX_BATCH, count - done);
        if (done) RT_CHECK(rt_sync(g.stream));               // the previous launch's copy still reads the pinned staging
        uint32_t live = 0, max_slots = 1, live_guess = 0;
        size_t chal_at = 0;
        for (uint32_t k = 0; k < cnt; k++) live_guess += pgs[done + k]->num_calculations != 0;
        const uint32_t lds_slots = (uint64_t)grid * live_guess <= 512 ? GRAPH_LDS_SLOTS_LONE : GRAPH_LDS_SLOTS_BATCH;
        for (uint32_t k = 0; k < cnt; k++) {
            Program &pg = *pgs[done + k];
            if (pg.num_calculations == 0) {                  // Ok(F::ZERO), graph_evaluator.rs:386-389
                RT_CHECK(rt_memset(d_outs[done + k], 0, num_rows * 32, g.stream));
                continue;
            }
            const unsigned char *st = reinterpret_cast<const unsigned char *>(pg.d_static);
            for (size_t v = 0; v < pg.chal_vars.size(); v++)
                to_limbs29(p0.field, challenges + (size_t)pg.chal_vars[v].first * 4, pg.chal_vars[v].second, reinterpret_cast<uint32_t *>(p0.h_dyn + p0.o_chal) + (chal_at + v) * 9);
            GraphJob job{reinterpret_cast<const uint32_t *>(st + pg.o_code), reinterpret_cast<const uint32_t *>(st + pg.o_const),
                         reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(p0.dyn.p) + p0.o_chal) + chal_at * 9,
                         reinterpret_cast<const int32_t *>(st + pg.o_rot), reinterpret_cast<unsigned char *>(d_outs[done + k]), pg.ninstr,
                         std::min<uint32_t>(pg.nslots, lds_slots)};
            chal_at += pg.chal_vars.size();
#ifndef MIRA_CPU_EMU
            if (pg.jit_fn) {
                bool same = true;
                for (uint32_t col : pg.used_columns) same &= columns[col].kind == pg.jit_kinds[col];
                if (same) { jit_jobs.push_back({pg.jit_fn, job}); continue; }
            }
#endif
            memcpy(p0.h_dyn + p0.o_jobs + (size_t)live * sizeof(GraphJob), &job, sizeof job);
            max_slots = std::max(max_slots, pg.nslots);
            live++;
        }
#ifdef MIRA_CPU_EMU
        if (live) RT_CHECK(rt_h2d(p0.dyn.p, p0.h_dyn, p0.dyn_bytes, g.stream));
#else
        if (live || !jit_jobs.empty()) RT_CHECK(rt_h2d(p0.dyn.p, p0.h_dyn, p0.dyn_bytes, g.stream));
        // specialised programs: one launch each (a launch of 2^17 rows is two waves per SIMD, what their 256 VGPRs allow)
        for (auto &jj : jit_jobs) {
            struct { const uint32_t *consts29, *chal29; const GraphCol *cols; unsigned char *out; uint64_t nrows; } args{
                jj.second.consts29, jj.second.challenges29, reinterpret_cast<const GraphCol *>(reinterpret_cast<const unsigned char *>(p0.dyn.p) + p0.o_cols), jj.second.out, (uint64_t)num_rows};
            size_t arg_bytes = sizeof args;
            void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &arg_bytes, HIP_LAUNCH_PARAM_END};
            const uint32_t jgrid = (uint32_t)std::min<size_t>((num_rows + graphjit::BLOCK - 1) / graphjit::BLOCK, 256 * 4 * 2 * 4);
            RT_CHECK(hipModuleLaunchKernel(jj.first, jgrid, 1, 1, graphjit::BLOCK, 1, 1, 0, g.stream, nullptr, cfg));
        }
        jit_jobs.clear();
#endif
        if (live) {
            const size_t ws_stride = (size_t)max_slots * 9 * T;
            if ((rc = g.graph_ws.ensure(ws_stride * live * 4))) return rc;
            const unsigned char *dy = reinterpret_cast<const unsigned char *>(p0.dyn.p);
            if (p0.field == MIRA_FIELD_FQ)
                LAUNCH(k_graph_eval<Fq29>, dim3(grid, live), block, (size_t)lds_slots * 9 * block * 4, g.stream, reinterpret_cast<const GraphJob *>(dy + p0.o_jobs),
                       reinterpret_cast<const GraphCol *>(dy + p0.o_cols), (uint64_t)num_rows, reinterpret_cast<uint32_t *>(g.graph_ws.p), (uint64_t)ws_stride);
            else
                LAUNCH(k_graph_eval<Fr29>, dim3(grid, live), block, (size_t)lds_slots * 9 * block * 4, g.stream, reinterpret_cast<const GraphJob *>(dy + p0.o_jobs),
                       reinterpret_cast<const GraphCol *>(dy + p0.o_cols), (uint64_t)num_rows, reinterpret_cast<uint32_t *>(g.graph_ws.p), (uint64_t)ws_stride);
        }
    }
    tm_mark("graph_eval");
    RT_CHECK(rt_last());
    RT_CHECK(rt_sync(g.stream));
    tm_end();
    return MIRA_OK;
}

int graph_eval_compiled(uint64_t handle, const mira_eval_column *columns, uint32_t num_columns, const uint64_t *challenges, uint32_t num_challenges,
                        size_t num_rows, void *d_out) {
    return graph_eval_batch(&handle, 1, columns, num_columns, challenges, num_challenges, num_rows, &d_out);
}

// one-shot form: compile, evaluate, free
int graph_eval_device(int field, const mira_graph *gr, const mira_eval_column *columns, uint32_t num_columns, const uint64_t *challenges,
                      uint32_t num_challenges, size_t num_rows, void *d_out) {
    uint64_t h = 0;
    int rc = graph_compile(field, gr, num_challenges, num_columns, &h);
    if (rc) return rc;
    rc = graph_eval_compiled(h, columns, num_columns, challenges, num_challenges, num_rows, d_out);
    const std::string err = rc ? std::string(mira_last_error()) : std::string();
    graph_free(h);
    if (rc) set_error(err);
    return rc;
}

// ---- specialised kernels (graph_jit.hpp) ----------------------------------------------------------------------------
static std::vector<uint32_t> kinds_of(const mira_eval_column *columns, uint32_t num_columns) {
    std::vector<uint32_t> k(num_columns, MIRA_COL_FIELD);
    for (uint32_t c = 0; c < num_columns; c++) if (columns) k[c] = columns[c].kind;
    return k;
}
int graph_jit_source(uint64_t handle, const mira_eval_column *columns, uint32_t num_columns, char *buf, size_t cap, size_t *len_out) {
    auto it = g_programs.find(handle);
    if (it == g_programs.end()) { set_error("unknown graph handle"); return MIRA_E_BAD_ARG; }
    const Program &pg = it->second;
    if (!len_out) { set_error("null output"); return MIRA_E_BAD_ARG; }
    if (pg.num_calculations == 0 || pg.ninstr == 0) { *len_out = 0; return MIRA_OK; }
    if (num_columns != pg.num_columns) { set_error("the graph was compiled for " + std::to_string(pg.num_columns) + " columns"); return MIRA_E_BAD_ARG; }
    const std::string src = graphjit::source(pg.field, pg.h_stream, pg.ninstr, pg.h_rot, kinds_of(columns, num_columns), tuned(MIRA_TUNE_JIT_LOADS_AHEAD, graphjit::LOADS_AHEAD_DEFAULT));
    *len_out = src.size();
    if (buf && cap) {
        const size_t ncopy = std::min(cap - 1, src.size());
        memcpy(buf, src.data(), ncopy);
        buf[ncopy] = 0;
    }
    return MIRA_OK;
}

#ifndef MIRA_CPU_EMU
#include "jit_headers.inc"
#include <fcntl.h>
#include <sys/stat.h>
namespace graphjit {
Rtc &rtc() {
    static Rtc r;
    if (r.tried) return r;
    r.tried = true;
    for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) { r.error = "libhiprtc.so not found: graphs stay interpreted"; return r; }
    auto sym = [&](const char *n) { void *p = dlsym(r.lib, n); if (!p) r.error = std::string("libhiprtc.so lacks ") + n; return p; };
    r.create = reinterpret_cast<decltype(r.create)>(sym("hiprtcCreateProgram"));
    r.compile = reinterpret_cast<decltype(r.compile)>(sym("hiprtcCompileProgram"));
    r.log_size = reinterpret_cast<decltype(r.log_size)>(sym("hiprtcGetProgramLogSize"));
    r.log = reinterpret_cast<decltype(r.log)>(sym("hiprtcGetProgramLog"));
    r.code_size = reinterpret_cast<decltype(r.code_size)>(sym("hiprtcGetCodeSize"));
    r.code = reinterpret_cast<decltype(r.code)>(sym("hiprtcGetCode"));
    r.destroy = reinterpret_cast<decltype(r.destroy)>(sym("hiprtcDestroyProgram"));
    return r;
}
static const char *const JIT_OPTIONS[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
std::vector<char> compile(const std::string &src, std::string &err) {
    Rtc &r = rtc();
    std::vector<char> out;
    if (!r.error.empty()) { err = r.error; return out; }
    void *prog = nullptr;
    // the kernel headers travel inside the library: `#include "field29.cuh"` (and its own includes) resolve to these texts
    if (r.create(&prog, src.c_str(), "mira_jit.hip", JIT_HDR_COUNT, const_cast<const char **>(JIT_HDR_TEXT), const_cast<const char **>(JIT_HDR_NAME)) != 0) {
        err = "hiprtcCreateProgram failed";
        return out;
    }
    const int rc = r.compile(prog, 3, const_cast<const char **>(JIT_OPTIONS));
    if (rc != 0) {
        size_t n = 0;
        (void)r.log_size(prog, &n);
        std::string log(n, 0);
        if (n > 1) (void)r.log(prog, &log[0]);
        err = "hiprtcCompileProgram failed (" + std::to_string(rc) + "): " + log.substr(0, 2000);
        (void)r.destroy(&prog);
        return out;
    }
    size_t n = 0;
    if (r.code_size(prog, &n) == 0 && n) { out.resize(n); if (r.code(prog, out.data()) != 0) out.clear(); }
    if (out.empty()) err = "hiprtcGetCode failed";
    (void)r.destroy(&prog);
    return out;
}

// ---- code objects on disk (mira_graph_set_cache_dir) --------------------------------------------------------------
// A file per kernel: magic | key of the build environment | source length | source | code length | hash of the code | code.
// The file name is a hash of source and environment; a hit must match both byte for byte and the code must hash to what the
// header says, so a colliding, stale or damaged file (other kernel headers, another ROCm, another GPU architecture, a
// truncated write) is a miss, never a wrong kernel.  The hash guards against damage, not against an adversary: a code object
// is executed on the GPU as it is, so the directory must belong to the user and be writable by nobody else -- checked when it
// is set and again for every file that is taken.
static std::string g_cache_dir;
static uint64_t fnv1a(const void *p, size_t n, uint64_t h = 0xcbf29ce484222325ull) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}
static bool read_file(const std::string &path, std::vector<char> &out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    struct stat sb;
    if (fstat(fileno(f), &sb) != 0 || sb.st_uid != geteuid() || (sb.st_mode & (S_IWGRP | S_IWOTH))) { fclose(f); return false; }   // somebody else's file, or one others may write
    out.clear();
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
    const bool ok = !ferror(f);
    fclose(f);
    return ok;
}
// everything besides the source text that decides the code object: the embedded headers, the compiler options, the GPU
// architecture the process runs on, the HIP runtime and the hiprtc that compiles
static const std::string &environment_key() {
    static std::string key;
    if (!key.empty()) return key;
    Rtc &r = rtc();
    uint64_t h = 0xcbf29ce484222325ull;
    for (int i = 0; i < JIT_HDR_COUNT; i++) {
        h = fnv1a(JIT_HDR_TEXT[i], strlen(JIT_HDR_TEXT[i]), h);
        h = fnv1a(JIT_HDR_NAME[i], strlen(JIT_HDR_NAME[i]), h);
    }
    std::string opts;
    for (const char *o : JIT_OPTIONS) { opts += o; opts += ' '; }
    int major = 0, minor = 0, runtime = 0;
    if (r.lib) {
        auto version = reinterpret_cast<int (*)(int *, int *)>(dlsym(r.lib, "hiprtcVersion"));
        if (version) (void)version(&major, &minor);
    }
    (void)hipRuntimeGetVersion(&runtime);
    hipDeviceProp_t prop;
    std::string arch = "unknown";
    if (hipGetDeviceProperties(&prop, g.device) == hipSuccess) arch = prop.gcnArchName;
    char buf[96];
    snprintf(buf, sizeof buf, "%016llx", (unsigned long long)h);
    key = "mira-jit-2 arch " + arch + " hip " + std::to_string(runtime) + " hiprtc " + std::to_string(major) + "." + std::to_string(minor) + " headers " + buf + " options " + opts;
    return key;
}
static std::string cache_path(const std::string &src) {
    const std::string &env = environment_key();
    const uint64_t a = fnv1a(src.data(), src.size()), b = fnv1a(env.data(), env.size(), a ^ 0x9e3779b97f4a7c15ull);
    char name[64];
    snprintf(name, sizeof name, "/mira_jit_%016llx%016llx.bin", (unsigned long long)a, (unsigned long long)b);
    return g_cache_dir + name;
}
static constexpr char CACHE_MAGIC[8] = {'M', 'I', 'R', 'A', 'J', 'I', 'T', '2'};
static void code_hash(const std::vector<char> &code, uint64_t out[2]) {
    out[0] = fnv1a(code.data(), code.size());
    out[1] = fnv1a(code.data(), code.size(), 0x84222325cbf29ce4ull ^ code.size());
}
static std::vector<char> cache_load(const std::string &src) {
    std::vector<char> file, code;
    if (g_cache_dir.empty() || !read_file(cache_path(src), file)) return code;
    const std::string &env = environment_key();
    size_t pos = 0;
    auto take = [&](const void *want, size_t n) { const bool ok = pos + n <= file.size() && memcmp(file.data() + pos, want, n) == 0; pos += n; return ok; };
    auto take_len = [&](uint64_t &v) { if (pos + 8 > file.size()) return false; memcpy(&v, file.data() + pos, 8); pos += 8; return true; };
    uint64_t n_env = 0, n_src = 0, n_code = 0, want[2] = {0, 0}, have[2];
    if (!take(CACHE_MAGIC, 8) || !take_len(n_env) || n_env != env.size() || !take(env.data(), env.size())) return code;
    if (!take_len(n_src) || n_src != src.size() || !take(src.data(), src.size())) return code;
    if (!take_len(n_code) || !take_len(want[0]) || !take_len(want[1]) || n_code == 0 || pos + n_code != file.size()) return code;
    code.assign(file.begin() + (long)pos, file.end());
    code_hash(code, have);
    if (have[0] != want[0] || have[1] != want[1]) code.clear();
    return code;
}
static void cache_store(const std::string &src, const std::vector<char> &code) {   // best effort: a failure costs the next process a compilation
    if (g_cache_dir.empty() || code.empty()) return;
    const std::string path = cache_path(src), tmp = path + ".tmp" + std::to_string((unsigned long long)getpid());
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL, 0600);
    FILE *f = fd >= 0 ? fdopen(fd, "wb") : nullptr;
    if (!f) { if (fd >= 0) close(fd); return; }
    const std::string &env = environment_key();
    uint64_t hash[2];
    code_hash(code, hash);
    const uint64_t n_env = env.size(), n_src = src.size(), n_code = code.size();
    bool ok = fwrite(CACHE_MAGIC, 1, 8, f) == 8 && fwrite(&n_env, 8, 1, f) == 1 && fwrite(env.data(), 1, env.size(), f) == env.size();
    ok = ok && fwrite(&n_src, 8, 1, f) == 1 && fwrite(src.data(), 1, src.size(), f) == src.size();
    ok = ok && fwrite(&n_code, 8, 1, f) == 1 && fwrite(hash, 8, 2, f) == 2 && fwrite(code.data(), 1, code.size(), f) == code.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());   // rename: readers see a whole file or none
}
}   // namespace graphjit
#endif

// Directory for the code objects of specialised kernels, or null / "" for none (the default): a later process -- or this
// one after mira_graph_free -- that specialises the same graph loads the kernel instead of compiling it.
int graph_set_cache_dir(const char *dir) {
#ifdef MIRA_CPU_EMU
    (void)dir;
    return MIRA_OK;
#else
    std::string d = dir ? dir : "";
    while (d.size() > 1 && d.back() == '/') d.pop_back();
    if (!d.empty()) {
        // code objects found there are executed on the GPU: the directory must be the user's own, writable by nobody else
        struct stat sb;
        if (stat(d.c_str(), &sb) != 0 || !S_ISDIR(sb.st_mode)) { set_error(d + " is not a directory"); return MIRA_E_IO; }
        if (sb.st_uid != geteuid() || (sb.st_mode & (S_IWGRP | S_IWOTH))) {
            set_error(d + " must belong to the calling user and be writable by nobody else (mode 0700 or 0755): kernels found there are executed");
            return MIRA_E_BAD_ARG;
        }
    }
    graphjit::g_cache_dir = d;
    return MIRA_OK;
#endif
}
// source text -> code object size, through the library's own hiprtc path and embedded headers; needs no device
int graph_jit_compile_check(const char *src, size_t *code_size_out) {
#ifdef MIRA_CPU_EMU
    (void)src; (void)code_size_out;
    set_error("the host emulation has no run-time compiler");
    return MIRA_E_JIT_UNAVAILABLE;
#else
    if (!src) { set_error("null source"); return MIRA_E_BAD_ARG; }
    if (!graphjit::rtc().error.empty()) { set_error(graphjit::rtc().error); return MIRA_E_JIT_UNAVAILABLE; }
    std::string err;
    const std::vector<char> code = graphjit::compile(src, err);
    if (code.empty()) { set_error(err); return MIRA_E_JIT_FAILED; }
    if (code_size_out) *code_size_out = code.size();
    return MIRA_OK;
#endif
}

// Every handle gets its own kernel; the compilations run on one host thread each (a MainGate<5> evaluation point takes
// ~5 s).  A handle that is specialised already, or has no calculations, is left as it is.  On failure nothing changes:
// the graphs stay interpreted.
int graph_specialize(const uint64_t *handles, uint32_t count, const mira_eval_column *columns, uint32_t num_columns, std::unique_lock<std::mutex> *library_lock) {
#ifdef MIRA_CPU_EMU
    (void)handles; (void)count; (void)columns; (void)num_columns; (void)library_lock;
    set_error("the host emulation has no run-time compiler: graphs stay interpreted");
    return MIRA_E_JIT_UNAVAILABLE;
#else
    std::vector<uint64_t> todo;                              // handles, not pointers: the lock is released while the compiler runs
    for (uint32_t k = 0; k < count; k++) {
        auto it = g_programs.find(handles[k]);
        if (it == g_programs.end()) { set_error("unknown graph handle"); return MIRA_E_BAD_ARG; }
        const Program &pg = it->second;
        if (num_columns != pg.num_columns) { set_error("the graph was compiled for " + std::to_string(pg.num_columns) + " columns"); return MIRA_E_BAD_ARG; }
        if (pg.jit_fn || pg.num_calculations == 0 || pg.ninstr == 0) continue;
        if (pg.ninstr > graphjit::MAX_INSTR) { set_error("graph of " + std::to_string(pg.ninstr) + " instructions is too long to specialise"); return MIRA_E_UNSUPPORTED; }
        if (std::find(todo.begin(), todo.end(), handles[k]) == todo.end()) todo.push_back(handles[k]);
    }
    g_jit_last_compiled = g_jit_last_from_disk = 0;
    if (todo.empty()) return MIRA_OK;
    if (!graphjit::rtc().error.empty()) { set_error(graphjit::rtc().error); return MIRA_E_JIT_UNAVAILABLE; }
    std::vector<std::vector<char>> code(todo.size());
    std::vector<std::string> errs(todo.size());
    std::vector<std::thread> workers;
    const std::vector<uint32_t> kinds = kinds_of(columns, num_columns);
    const size_t ahead = tuned(MIRA_TUNE_JIT_LOADS_AHEAD, graphjit::LOADS_AHEAD_DEFAULT);
    // code objects of this process by source text: a second evaluator of the same graph (another PlonkStructure of the same
    // circuit, the other leg of a benchmark) costs a module load, not a compilation
    static std::map<std::string, std::vector<char>> compiled;
    std::vector<std::string> src(todo.size());
    std::vector<size_t> fresh;
    size_t from_disk = 0;
    for (size_t k = 0; k < todo.size(); k++) {
        const Program &pg = g_programs.find(todo[k])->second;
        src[k] = graphjit::source(pg.field, pg.h_stream, pg.ninstr, pg.h_rot, kinds, ahead);
        auto hit = compiled.find(src[k]);
        if (hit != compiled.end()) { code[k] = hit->second; continue; }
        code[k] = graphjit::cache_load(src[k]);              // a file of an earlier process (mira_graph_set_cache_dir)
        if (code[k].empty()) fresh.push_back(k); else { compiled[src[k]] = code[k]; from_disk++; }
    }
    // The compiler runs for seconds and touches nothing of the library's: other threads may commit, transform and evaluate
    // (interpreted) meanwhile.  They may also free one of these handles -- looked up again below.
    if (library_lock && !fresh.empty()) library_lock->unlock();
    auto work = [&](size_t k) { code[k] = graphjit::compile(src[k], errs[k]); };
    for (size_t q = 1; q < fresh.size(); q++) workers.emplace_back(work, fresh[q]);
    if (!fresh.empty()) work(fresh[0]);
    for (auto &t : workers) t.join();
    if (library_lock && !fresh.empty()) library_lock->lock();
    for (size_t k = 0; k < todo.size(); k++)
        if (code[k].empty()) { set_error(errs[k]); return MIRA_E_JIT_FAILED; }
    if (compiled.size() + fresh.size() > 256) compiled.clear();   // a bound on what a long-lived process keeps (a code object is ~200 KiB)
    for (size_t k : fresh) { compiled[src[k]] = code[k]; graphjit::cache_store(src[k], code[k]); }
    g_jit_last_compiled = (uint32_t)fresh.size(); g_jit_last_from_disk = (uint32_t)from_disk;
    std::vector<Program *> live(todo.size(), nullptr);       // freed meanwhile, or specialised by another thread: nothing to do
    for (size_t k = 0; k < todo.size(); k++) {
        auto it = g_programs.find(todo[k]);
        if (it != g_programs.end() && !it->second.jit_fn) live[k] = &it->second;
    }
    std::vector<hipModule_t> mods(todo.size(), nullptr);
    std::vector<hipFunction_t> fns(todo.size(), nullptr);
    for (size_t k = 0; k < todo.size(); k++) {
        if (!live[k]) continue;
        hipError_t e = hipModuleLoadData(&mods[k], code[k].data());
        if (e == hipSuccess) e = hipModuleGetFunction(&fns[k], mods[k], "mira_jit_eval");
        if (e != hipSuccess) {
            for (size_t q = 0; q <= k; q++) if (mods[q]) (void)hipModuleUnload(mods[q]);
            set_error(std::string("loading a specialised kernel failed: ") + hipGetErrorString(e));
            return MIRA_E_JIT_FAILED;
        }
    }
    for (size_t k = 0; k < todo.size(); k++)
        if (live[k]) { live[k]->jit_mod = mods[k]; live[k]->jit_fn = fns[k]; live[k]->jit_kinds = kinds; }
    return MIRA_OK;
#endif
}
int graph_jit_stats(uint32_t *compiled_out, uint32_t *from_disk_out) {
    if (compiled_out) *compiled_out = g_jit_last_compiled;
    if (from_disk_out) *from_disk_out = g_jit_last_from_disk;
    return MIRA_OK;
}
int graph_is_specialized(uint64_t handle, int32_t *out) {
    auto it = g_programs.find(handle);
    if (it == g_programs.end() || !out) { set_error("unknown graph handle"); return MIRA_E_BAD_ARG; }
#ifdef MIRA_CPU_EMU
    *out = 0;
#else
    *out = it->second.jit_fn ? 1 : 0;
#endif
    return MIRA_OK;
}
