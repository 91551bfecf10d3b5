// Specialised cross-term kernels: the compiled instruction stream of one graph (graph.hip) written out as
// straight-line HIP and compiled for gfx950 at run time (hiprtc).
//
// k_graph_eval (graph_kernels.cuh) INTERPRETS the stream: every instruction is decoded with wave-uniform branches,
// its operands come out of LDS or workspace slots (nine dword accesses each), a column is waited for where it is
// read, and a rotated row costs a 64-bit division per column read.  A gate polynomial is fixed for the whole IVC
// run, so the same stream can be a kernel of its own: intermediates are SSA values the register allocator places
// (two waves per SIMD, 256 VGPRs), loads are hoisted by the compiler, constants are scalar loads at literal
// offsets, a rotation is reduced once per row, and there is no decode.  Measured on the MainGate<5> cross terms at
// 2^17 rows: 38 % -> ~75 % of the multiplier's rate (profiles/r03_d_graph_jit.txt).
//
// The arithmetic is the interpreter's, instruction for instruction (same field29.cuh calls in the same order on
// the same operands), so every value -- not just the canonical result -- is the one k_graph_eval computes.
//
// hiprtc is loaded with dlopen at first use: a machine without it keeps the interpreter (mira_graph_specialize
// reports MIRA_E_JIT_UNAVAILABLE).  The headers the source includes (field29.cuh, field.cuh, platform.h) are part of
// the library: the build embeds their text (Makefile: jit_headers.inc) and hands it to hiprtc as named headers, so
// nothing has to lie beside libmira_gpu.so.
#pragma once
#include <string>
#include <vector>

#include "graph_kernels.cuh"

namespace graphjit {

constexpr uint32_t MAX_INSTR = 1536;            // longer streams stay with the interpreter (compile time grows with the square of a basic block)
constexpr size_t LOADS_AHEAD_DEFAULT = 4;       // column reads in flight ahead of their use (MIRA_TUNE_JIT_LOADS_AHEAD)
constexpr uint32_t BLOCK = 64;                  // lanes per workgroup of a specialised kernel (measured: 64 ahead of 128 and 256 by 3 - 6 %)

inline uint32_t words_of(uint32_t head) { const uint32_t op = head & 0xFFu; return op == GOP_MAC ? 7u : (op == GOP_ADD || op == GOP_SUB || op == GOP_MUL) ? 5u : 4u; }

// HIP source of one program.  `rotations`: the graph's rotation table (baked in: a rotation of zero reads the lane's own row).
// `kinds`: MIRA_COL_FIELD / MIRA_COL_BOOL of every column index the stream reads (others: anything).
inline std::string source(int field, const std::vector<uint32_t> &stream, uint32_t ninstr, const std::vector<int32_t> &rotations, const std::vector<uint32_t> &kinds,
                          size_t loads_ahead = LOADS_AHEAD_DEFAULT) {
    std::string s;
    s.reserve(64 * 1024);
    s += "#include \"field29.cuh\"\n";
    s += field == MIRA_FIELD_FQ ? "using F = Fq29;\n" : "using F = Fr29;\n";
    s += "using S = F::Sat;\n"
         "struct GraphCol { const unsigned char *p; uint32_t kind, pad; };\n"
         "DEV Fe29<F> jit_const(const uint32_t *__restrict__ p, uint32_t i) {\n"
         "    Fe29<F> r;\n"
         "#pragma unroll\n"
         "    for (int k = 0; k < 9; k++) r.l[k] = p[i * 9 + k];\n"
         "    return r;\n"
         "}\n"
         "// a column as it lies in memory (x * 2^256); a selector as the number one in that form (graph_kernels.cuh).  The KIND of\n"
         "// every column is fixed when the kernel is built: a branch on it in front of every read cost 880 spilled registers\n"
         "DEV Fe29<F> jit_bool(uint32_t byte) {\n"
         "    Fe<S> o;\n"
         "#pragma unroll\n"
         "    for (int k = 0; k < 8; k++) o.l[k] = byte ? S::R1[k] : 0u;\n"
         "    return f29_unpack_canonical<F>(o);\n"
         "}\n"
         "extern \"C\" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))\n"
         "mira_jit_eval(const uint32_t *__restrict__ consts29, const uint32_t *__restrict__ chal29, const GraphCol *__restrict__ cols,\n"
         "              unsigned char *__restrict__ out, uint64_t nrows) {\n"
         "    const uint64_t T = (uint64_t)gridDim.x * blockDim.x;\n"
         "    for (uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; row < nrows; row += T) {\n";
    // rows of the rotations the stream uses, once per row (rem_euclid, graph_evaluator.rs:51-53)
    std::vector<char> rot_used(rotations.size(), 0);
    {
        size_t pos = 0;
        for (uint32_t i = 0; i < ninstr; i++) {
            const uint32_t head = stream[pos], op = head & 0xFFu;
            const uint32_t nsrc = op == GOP_MAC ? 3u : (op == GOP_ADD || op == GOP_SUB || op == GOP_MUL) ? 2u : 1u;
            for (uint32_t k = 0; k < nsrc; k++) {
                const uint32_t w = stream[pos + 3 + k];
                if ((w >> 29) == MIRA_SRC_COLUMN) rot_used[(w & 0x1FFFFFFFu) >> 20] = 1;
            }
            pos += words_of(head);
        }
    }
    for (size_t r = 0; r < rotations.size(); r++) {
        if (!rot_used[r] || rotations[r] == 0) continue;
        s += "        uint64_t row_r" + std::to_string(r) + ";\n"
             "        { int64_t rr = ((int64_t)row + (" + std::to_string(rotations[r]) + ")) % (int64_t)nrows; if (rr < 0) rr += (int64_t)nrows; row_r" + std::to_string(r) + " = (uint64_t)rr; }\n";
    }
    // Column reads, in the order the stream consumes them.  Two waves per SIMD do not hide a load that is waited for where
    // it is issued (and the scheduling barriers keep the compiler from hoisting it), so the generator does the software
    // pipelining: the packed value of read k + LOADS_AHEAD is requested where read k is consumed, and unpacked at its use.
    struct ColRead { uint32_t col, rot; };
    std::vector<ColRead> reads;
    {
        size_t pos = 0;
        for (uint32_t i = 0; i < ninstr; i++) {
            const uint32_t head = stream[pos], op = head & 0xFFu;
            const uint32_t nsrc = op == GOP_MAC ? 3u : (op == GOP_ADD || op == GOP_SUB || op == GOP_MUL) ? 2u : 1u;
            for (uint32_t k = 0; k < nsrc; k++) {
                const uint32_t w = stream[pos + 3 + k];
                if ((w >> 29) == MIRA_SRC_COLUMN) reads.push_back({(w & 0x1FFFFFFFu) & 0xFFFFFu, (w & 0x1FFFFFFFu) >> 20});
            }
            pos += words_of(head);
        }
    }
    const size_t LOADS_AHEAD = loads_ahead;
    size_t issued = 0, consumed = 0;
    auto is_bool = [&](uint32_t col) { return col < kinds.size() && kinds[col] == MIRA_COL_BOOL; };
    auto issue_until = [&](size_t upto) {                     // request reads [issued, upto)
        for (; issued < upto && issued < reads.size(); issued++) {
            const ColRead &r = reads[issued];
            const std::string row = rotations[r.rot] == 0 ? std::string("row") : "row_r" + std::to_string(r.rot), c = "c" + std::to_string(issued);
            if (is_bool(r.col)) s += "        const uint32_t " + c + " = cols[" + std::to_string(r.col) + "].p[" + row + "];\n";
            else s += "        const Fe<S> " + c + " = fe_load<S>(cols[" + std::to_string(r.col) + "].p + " + row + " * 32);\n";
        }
    };
    issue_until(LOADS_AHEAD);
    std::vector<std::string> slot_val;
    auto operand = [&](uint32_t w, uint32_t i) -> std::string {
        const uint32_t kind = w >> 29, payload = w & 0x1FFFFFFFu;
        if (kind == GRAPH_SRC_PREV) return "t" + std::to_string(i - 1);
        if (kind == MIRA_SRC_CONSTANT) return "jit_const(consts29, " + std::to_string(payload) + "u)";
        if (kind == MIRA_SRC_CHALLENGE) return "jit_const(chal29, " + std::to_string(payload) + "u)";
        if (kind == MIRA_SRC_INTERMEDIATE) return payload < slot_val.size() && !slot_val[payload].empty() ? slot_val[payload] : std::string("f29_zero<F>()");
        const size_t k = consumed++;                          // operands are resolved in stream order: this is read k
        issue_until(k + 1 + LOADS_AHEAD);
        return is_bool(payload & 0xFFFFFu) ? "jit_bool(c" + std::to_string(k) + ")" : "f29_unpack_canonical<F>(c" + std::to_string(k) + ")";
    };
    auto bias = [](uint32_t K) { return K == 2 ? "2" : K == 4 ? "4" : K == 8 ? "8" : "16"; };   // graph_sub's cases
    size_t pos = 0;
    for (uint32_t i = 0; i < ninstr; i++) {
        const uint32_t head = stream[pos], dst = stream[pos + 1], op = head & 0xFFu, K = head >> 8;
        const uint32_t nsrc = op == GOP_MAC ? 3u : (op == GOP_ADD || op == GOP_SUB || op == GOP_MUL) ? 2u : 1u;
        const std::string a = operand(stream[pos + 3], i), b = nsrc > 1 ? operand(stream[pos + 4], i) : std::string(), c3 = nsrc > 2 ? operand(stream[pos + 5], i) : std::string();
        const std::string t = "t" + std::to_string(i);
        std::string e;
        bool product = false;
        switch (op) {
            case GOP_ADD: e = "f29_add(" + a + ", " + b + ")"; break;
            case GOP_SUB: e = std::string("f29_sub<") + bias(K) + ">(" + a + ", " + b + ")"; break;
            case GOP_MUL: e = "f29_mul(" + a + ", " + b + ")"; product = true; break;
            case GOP_MAC: e = "f29_add(f29_mul(" + a + ", " + b + "), " + c3 + ")"; product = true; break;
            case GOP_SQR: e = "f29_sqr(" + a + ")"; product = true; break;
            case GOP_DBL: e = "f29_dbl(" + a + ")"; break;
            case GOP_NEG: e = std::string("f29_sub<") + bias(K) + ">(f29_zero<F>(), " + a + ")"; break;
            case GOP_NORM: e = "f29_mul(" + a + ", f29_one<F>())"; product = true; break;
            default: e = a; break;                                   // GOP_COPY
        }
        s += "        const Fe29<F> " + t + " = " + e + ";\n";
        // one scheduling region per product: the machine scheduler is quadratic in the length of a region (16 s -> 5 s
        // for 110 products) and interleaving products only raises the register pressure (90 -> 2 spilled registers)
        if (product) s += "        __builtin_amdgcn_sched_barrier(0);\n";
        if (dst != GRAPH_NO_SLOT) {
            if (dst >= slot_val.size()) slot_val.resize(dst + 1);
            slot_val[dst] = t;
        }
        pos += words_of(head);
    }
    s += "        fe_store(out + row * 32, reduce_once(f29_pack(t" + std::to_string(ninstr - 1) + ")));\n"
         "    }\n"
         "}\n";
    return s;
}

#ifndef MIRA_CPU_EMU
// ---- hiprtc, loaded on demand ------------------------------------------------------------------------------------
struct Rtc {
    void *lib = nullptr;
    int (*create)(void **, const char *, const char *, int, const char **, const char **) = nullptr;
    int (*compile)(void *, int, const char **) = nullptr;
    int (*log_size)(void *, size_t *) = nullptr;
    int (*log)(void *, char *) = nullptr;
    int (*code_size)(void *, size_t *) = nullptr;
    int (*code)(void *, char *) = nullptr;
    int (*destroy)(void **) = nullptr;
    std::string error;
    bool tried = false;
};
Rtc &rtc();                                                      // graph.hip
// source -> code object; empty on failure (message in `err`).  Thread-safe once rtc() is loaded.
std::vector<char> compile(const std::string &src, std::string &err);
#endif

}   // namespace graphjit
