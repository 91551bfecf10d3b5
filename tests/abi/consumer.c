/* A compiled-C consumer of include/mira_gpu.h -- what a Rust `extern "C"` block binds to is THIS header, not the
 * ctypes table of mira_amd/_lib.py.  Built by the system C compiler (-std=c11 -Wall -Werror) against the header and
 * linked to libmira_gpu.so (tests/test_gpu_abi_consumer.py, -m gpu) or, for the CPU suite, to the test-only
 * emulation build of the same sources (tests/test_abi.py).  It checks struct layout at compile time and calls the
 * entry points a maintainer's shim would call first:
 *   register -> mira_msm on the reference's (r - 1) * G == -G vector (src/digest.rs:98-113) -> unregister;
 *   MIRA_E_TOO_LONG with Error::TooLongInput's text (src/commitment.rs:21-24);
 *   the 8-point FFT known-answer vector (src/fft.rs:240-249) and ifft(fft(x)) == x (:265-279);
 *   mira_graph_compile / mira_graph_eval_compiled on a three-calculation graph (src/polynomial/graph_evaluator.rs);
 *   the key-file entry point's error path, mira_trim, the per-handle window width.
 * Exit code 0 and "consumer ok" on success. */
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "mira_gpu.h"
#include "vectors.h"

/* the layout a Rust #[repr(C)] mirror of these structs must have (LP64) */
_Static_assert(sizeof(mira_graph) == 48, "mira_graph size");
_Static_assert(offsetof(mira_graph, code) == 0 && offsetof(mira_graph, code_words) == 8, "mira_graph.code / code_words");
_Static_assert(offsetof(mira_graph, num_calculations) == 16 && offsetof(mira_graph, num_constants) == 20, "mira_graph counts");
_Static_assert(offsetof(mira_graph, constants) == 24 && offsetof(mira_graph, rotations) == 32, "mira_graph pointers");
_Static_assert(offsetof(mira_graph, num_rotations) == 40 && offsetof(mira_graph, reserved) == 44, "mira_graph tail");
_Static_assert(sizeof(mira_eval_column) == 16 && offsetof(mira_eval_column, kind) == 8 && offsetof(mira_eval_column, reserved) == 12, "mira_eval_column");
_Static_assert(sizeof(size_t) == 8 && sizeof(void *) == 8, "LP64");
_Static_assert(MIRA_PARTIAL_U64 == 1024, "partial buffer words");

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "consumer FAILED %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); \
                                             fprintf(stderr, " (last error: %s)\n", mira_last_error()); return 1; } } while (0)

int main(void) {
    CHECK(mira_device_count() >= 1, "no device");
    CHECK(mira_init(0) == MIRA_OK, "mira_init");

    /* ---- MSM: (r - 1) * G == -G ------------------------------------------------------------------ */
    uint64_t handle = 0, out[8];
    CHECK(mira_msm_register_bases(MIRA_CURVE_BN256, KAT_G, 1, &handle) == MIRA_OK && handle != 0, "register");
    CHECK(mira_msm(handle, KAT_R_MINUS_1, 1, out) == MIRA_OK, "mira_msm");
    CHECK(memcmp(out, KAT_NEG_G, 64) == 0, "(r - 1) * G != -G");
    CHECK(mira_msm_set_handle_window_bits(handle, 9) == MIRA_OK, "per-handle width");
    CHECK(mira_msm(handle, KAT_R_MINUS_1, 1, out) == MIRA_OK && memcmp(out, KAT_NEG_G, 64) == 0, "(r - 1) * G under 9-bit windows");
    int32_t c = 0, w = 0;
    CHECK(mira_msm_last_plan(&c, &w) == MIRA_OK && c == 9 && w == 29, "last plan %d %d", (int)c, (int)w);
    CHECK(mira_msm(handle, KAT_R_MINUS_1, 0, out) == MIRA_OK, "empty commit");
    for (int i = 0; i < 8; i++) CHECK(out[i] == 0, "empty commit is the identity (0, 0)");
    uint64_t two[8];
    memcpy(two, KAT_R_MINUS_1, 32); memcpy(two + 4, KAT_R_MINUS_1, 32);
    CHECK(mira_msm(handle, two, 2, out) == MIRA_E_TOO_LONG, "TooLongInput code");
    CHECK(strcmp(mira_last_error(), "Can't commit too long input: input len: 2, but limit is 1") == 0, "TooLongInput text: %s", mira_last_error());
    CHECK(mira_msm_unregister(handle) == MIRA_OK, "unregister");
    CHECK(mira_msm(handle, KAT_R_MINUS_1, 1, out) == MIRA_E_BAD_ARG, "a released handle is unknown");

    /* ---- NTT: the reference's 8-point vector, then a round trip ---------------------------------- */
    uint64_t a[8][4];
    memcpy(a, KAT_FFT_IN, sizeof a);
    CHECK(mira_fft_bn256_fr(&a[0][0], 3) == MIRA_OK, "fft");
    CHECK(memcmp(a, KAT_FFT_OUT, sizeof a) == 0, "fft known-answer vector");
    CHECK(mira_ifft_bn256_fr(&a[0][0], 3) == MIRA_OK && memcmp(a, KAT_FFT_IN, sizeof a) == 0, "ifft(fft(x)) == x");
    uint64_t omega[4];
    CHECK(mira_get_omega_or_inv(3, 0, omega) == MIRA_OK, "omega");
    CHECK(mira_ntt_bn256_fr(&a[0][0], 3, omega) == MIRA_OK && memcmp(a, KAT_FFT_OUT, sizeof a) == 0, "best_fft with get_omega_or_inv(3)");
    CHECK(mira_fft_bn256_fr(&a[0][0], 29) == MIRA_E_BAD_ARG, "k > S is refused");

    /* ---- cross-term evaluator: out[i] = col[i] * col[i + 1] + const ------------------------------- */
    const uint32_t code[] = {
        MIRA_OP_STORE, (MIRA_SRC_COLUMN << 29) | 0u | (0u << 20),
        MIRA_OP_STORE, (MIRA_SRC_COLUMN << 29) | 0u | (1u << 20),
        MIRA_OP_HORNER | (1u << 8), (MIRA_SRC_INTERMEDIATE << 29) | 0u, (MIRA_SRC_INTERMEDIATE << 29) | 1u, (MIRA_SRC_CONSTANT << 29) | 0u,
    };
    const int32_t rotations[] = {0, 1};
    mira_graph g;
    memset(&g, 0, sizeof g);
    g.code = code; g.code_words = sizeof code / sizeof code[0]; g.num_calculations = 3;
    g.constants = GRAPH_CONST; g.num_constants = 1; g.rotations = rotations; g.num_rotations = 2;
    void *d_col = NULL, *d_out = NULL;
    CHECK(mira_dev_alloc(4 * 32, &d_col) == MIRA_OK && mira_dev_alloc(4 * 32, &d_out) == MIRA_OK, "alloc");
    CHECK(mira_dev_upload(d_col, GRAPH_COL, 4 * 32) == MIRA_OK, "upload");
    mira_eval_column col;
    memset(&col, 0, sizeof col);
    col.d_data = d_col; col.kind = MIRA_COL_FIELD;
    uint64_t gh = 0, got[4][4];
    CHECK(mira_graph_compile(MIRA_FIELD_FR, &g, 0, 1, &gh) == MIRA_OK && gh != 0, "graph compile");
    CHECK(mira_graph_eval_compiled(gh, &col, 1, NULL, 0, 4, d_out) == MIRA_OK, "graph eval");
    CHECK(mira_dev_download(got, d_out, sizeof got) == MIRA_OK && memcmp(got, GRAPH_WANT, sizeof got) == 0, "graph values");
    g.num_calculations = 4;                                          /* malformed: fewer calculations in the code than announced */
    uint64_t bad = 0;
    CHECK(mira_graph_compile(MIRA_FIELD_FR, &g, 0, 1, &bad) == MIRA_E_BAD_ARG, "malformed graph is refused");
    CHECK(mira_graph_free(gh) == MIRA_OK, "graph free");
    CHECK(mira_dev_free(d_col) == MIRA_OK && mira_dev_free(d_out) == MIRA_OK, "free");

    /* ---- key file error path, memory ------------------------------------------------------------- */
    CHECK(mira_msm_register_bases_file(MIRA_CURVE_BN256, "/nonexistent/mira/key.bin", 4, 1, &handle) == MIRA_E_IO, "missing key file");
    size_t released = 0, free_b = 0, total_b = 0;
    CHECK(mira_trim(0, &released) == MIRA_OK, "trim");
    CHECK(mira_dev_mem_info(&free_b, &total_b) == MIRA_OK && free_b <= total_b, "mem info");
    CHECK(mira_fft_bn256_fr(&a[0][0], 3) == MIRA_OK, "a call after mira_trim re-allocates what it needs");
    printf("consumer ok\n");
    return 0;
}
