/*
 * mira_gpu.h -- C ABI of libmira_gpu.so, the MI355X (gfx950) MSM + NTT engine that drops in
 * behind joshbeal/mira's commitment and FFT functions.
 *
 * The reference has no FFI for this path; these entry points are what a `cfg(feature="gpu")`
 * shim in the reference would bind (INTEGRATION.md shows the Rust side).  Each one cites the
 * reference interface it replaces (paths relative to the reference repository root).
 *
 * Data layout (the in-memory representation of halo2curves types, passed by pointer cast):
 *   field element  = 4 x uint64_t little-endian limbs, Montgomery form, R = 2^256   (32 bytes)
 *   affine point   = x || y                                                        (64 bytes)
 *   identity point = all-zero bytes (src/poseidon/poseidon_hash.rs:137-140)
 *   curve ids      : MIRA_CURVE_BN256 (G1, y^2 = x^3 + 3 over Fq, scalars Fr)
 *                    MIRA_CURVE_GRUMPKIN (y^2 = x^3 - 17 over Fr, scalars Fq)
 *
 * Ownership: the caller owns every buffer; the library borrows it for the duration of a call.
 * Registered bases are copied into a library-owned device buffer; the caller's copy is never
 * retained.  Errors: every function returns MIRA_OK (0) or a negative MIRA_E_* code and
 * never throws or aborts across the ABI; mira_last_error() describes the last failure on the
 * calling thread.  Threading: all entry points are re-entrant (one process-wide lock).
 * There is no CPU fallback: without a usable gfx950 device every compute call fails with
 * MIRA_E_NO_DEVICE.
 */
#ifndef MIRA_GPU_H
#define MIRA_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIRA_CURVE_BN256 0
#define MIRA_CURVE_GRUMPKIN 1

#define MIRA_OK 0
#define MIRA_E_NO_DEVICE (-1)     /* no HIP device / HIP runtime error */
#define MIRA_E_BAD_ARG (-2)       /* null pointer, unknown curve or handle, size out of range */
#define MIRA_E_TOO_LONG (-3)      /* scalars longer than the registered key: Error::TooLongInput */
#define MIRA_E_ALLOC (-4)         /* device allocation failed */
#define MIRA_E_UNSUPPORTED (-5)   /* size not supported by this build */
#define MIRA_E_INVALID_POINT (-6) /* a base is not on the curve */
#define MIRA_E_IO (-7)            /* key file missing, unreadable or shorter than 2^k points */
#define MIRA_E_JIT_UNAVAILABLE (-8) /* mira_graph_specialize: no run-time compiler on this machine (no libhiprtc.so); the graphs stay interpreted */
#define MIRA_E_JIT_FAILED (-9)    /* mira_graph_specialize: the compiler or the module loader failed (mira_last_error has its log); the graphs stay interpreted */

/* Largest number of windows any MSM configuration uses; sizes mira_msm_partial buffers. */
#define MIRA_MAX_WINDOWS 64
/* One partial = W XYZZ points of 16 u64 each (X, Y, ZZ, ZZZ), W <= MIRA_MAX_WINDOWS. */
#define MIRA_PARTIAL_U64 (MIRA_MAX_WINDOWS * 16)

/* ---- device / stream ------------------------------------------------------------------- */
int mira_device_count(void);
/* Bind the library to a HIP device (default 0) -- one process per GPU. */
int mira_init(int device);
/* Run all work on the caller's hipStream_t (e.g. torch's current stream); NULL = own stream. */
int mira_set_stream(void *hip_stream);
const char *mira_last_error(void);

/* ---- MSM: CommitmentKey::commit (src/commitment.rs:78-87) ------------------------------
 * register = upload the key once (`CommitmentKey { ck: Box<[C]> }`, src/commitment.rs:26-29;
 * it is immutable and outlives every fold step, src/ivc/public_params.rs:50).               */
int mira_msm_register_bases(int curve, const uint64_t *bases /* n * 8 limbs */, size_t n, uint64_t *handle_out);
/* Same, for bases already in device memory.  Either way the library keeps its own resident copy
 * in the engine's layout, so the caller's buffer is free again when the call returns. */
int mira_msm_register_bases_device(int curve, const void *d_bases, size_t n, uint64_t *handle_out);
int mira_msm_unregister(uint64_t handle);
/* Optional, once per key: build the fixed-base window tables 2^(20 w) * P_i, w < 13, in HBM
 * (13 x the key size: 3.3 GiB at 2^22 points).  Later MSMs of >= 2^18 pairs on this handle then use
 * one shared set of 2^19 buckets and 13 instead of 16 additions per pair.  Results are unchanged
 * (bit-identical); mira_msm_partial_device then reports window_bits = 0, num_windows = 64 (64
 * partial sums to be added).  MIRA_E_ALLOC if the tables do not fit. */
int mira_msm_precompute(uint64_t handle);
/* Same with the window width named: 20 (as above), 22 (12 tables, 2^21 buckets, 12 additions per
 * pair: for commits of 2^24 pairs and more) -- one of the two per key -- or any of 8 .. 16: a SHARED-BUCKET
 * set of W = ceil(256 / c) tables 2^(c w) * P_i (W x the key size: 16 bits = 1.9 GB, 12 bits = 2.6 GB for the
 * 1.8 M-point key of a k = 17 fold step).  All W windows of a commit then share ONE set of 2^(c-1) buckets:
 * W additions per pair, the fix-up and bucket reduction of one window instead of W, and no Horner
 * epilogue on the host -- the latency-bound part of the small commits of a fold step.  Several shared
 * widths may be built beside each other (and beside the wide tables): every commit of >= 2^12 pairs
 * picks the set that is fastest for its length (narrow for 2^17 pairs, 16 bits for 2^21) and -- from the second commit of
 * a shape on -- for the bit lengths of its scalars (a witness vector of mostly zeros is served like the short dense one it amounts to), commits of
 * >= 2^18 pairs the wide tables when the key has them.  mira_msm_partial_device reports window_bits = 0,
 * num_windows = 16; a rank of a sharded MSM takes the wide tables if
 * present, else the widest shared set, whatever its chunk length.  Results are bit-identical whichever
 * set serves a commit.  Building a width twice is a no-op.
 * window_bits = MIRA_TABLE_GLV: not a table but the ENDOMORPHISM COPY of the key, [P_0, phi(P_0), P_1, phi(P_1), ...] with
 * phi(x, y) = (beta x, y) (2 x the key's memory).  Both curves have j = 0, so phi(P) = lambda P for a cube root of unity
 * lambda of the scalar field: a single commit through the per-window path then splits every scalar into two 128-bit halves
 * (k = k1 + k2 lambda, k P = k1 P + k2 phi(P)) -- the same bucket additions over half the windows: half the bucket reduction,
 * half the chain of doublings on the host.  The same points bit for bit; mira_msm_last_plan reports ceil(129 / c) windows.
 * Single commits and batches through the per-window path use it; commits served by a table set and ranks of a sharded MSM
 * (whose partials must have one shape on every rank) do not.  Since round 4 the library builds the copy by itself for keys of
 * up to 2^26 points the first time a commit of at most 2^21 pairs can use it (MIRA_TUNE_GLV_AUTO_MAX_LOG; an allocation
 * failure just leaves the key without one): the call is only needed for longer keys or to pay the cost up front. */
#define MIRA_TABLE_GLV 2
int mira_msm_precompute_ex(uint64_t handle, int32_t window_bits);
/* Validate every registered base against the curve equation on the GPU, as
 * load_or_setup_cache does with is_on_curve (src/commitment.rs:145-154). */
int mira_msm_check_bases(uint64_t handle);
/* The commitment-key cache file straight into HBM: CommitmentKey::load_from_file (src/commitment.rs:110-127)
 * and, with validate != 0, the is_on_curve pass of load_or_setup_cache (:145-154) -- the raw in-memory
 * `[C]` dump save_to_file writes (:96-101), 2^k points of 64 bytes.  The file is read in 64 MiB chunks
 * by several threads into two pinned buffers; chunk i + 1 is read and copied while chunk i is converted to
 * the resident layout and checked on the GPU.  A file shorter than 2^k points -> MIRA_E_IO ("failed to fill
 * whole buffer", read_exact); a point off the curve -> MIRA_E_INVALID_POINT ("Wrong file in cache, some
 * ptr out of curve"), no handle.  mira_msm_save_bases_file writes that file from a registered key. */
int mira_msm_register_bases_file(int curve, const char *path, uint32_t k, int validate, uint64_t *handle_out);
int mira_msm_save_bases_file(uint64_t handle, const char *path);

/* out_affine = sum_i scalars[i] * bases[i], i < n <= registered length (the key's PREFIX is
 * used, src/commitment.rs:80).  n > registered length -> MIRA_E_TOO_LONG (src/commitment.rs:21-24,
 * 81-86).  Replaces best_multiexp(v, &ck[..v.len()]).to_affine().                            */
int mira_msm(uint64_t handle, const uint64_t *scalars /* n * 4 limbs */, size_t n, uint64_t out_affine[8]);
int mira_msm_device(uint64_t handle, const void *d_scalars, size_t n, uint64_t out_affine[8]);

/* `count` commitments over the same key in one submission -- the cross-term commits of one
 * fold step (`cross_terms.iter().map(|v| ck.commit(v))`, src/nifs/vanilla/mod.rs:124-127) are
 * count vectors of equal length n.  Each result is bit-identical to a separate mira_msm call.
 * scalars[b] = host pointer of vector b; device form: vector b starts at element b * stride_elems.
 * out_affine = count * 8 limbs. */
int mira_msm_batch(uint64_t handle, const uint64_t *const *scalars, size_t n, size_t count, uint64_t *out_affine);
int mira_msm_batch_device(uint64_t handle, const void *d_scalars, size_t n, size_t count, size_t stride_elems, uint64_t *out_affine);

/* Point-chunk sharding across GPUs (one process per GPU): each rank runs mira_msm_partial on
 * its chunk, the ranks all-gather the MIRA_PARTIAL_U64 words, and every rank combines.
 * `first` = index of the chunk's first base inside the registered key.  Every rank must make the
 * same choices: the same window width -- *window_bits on entry (4..16), or, if that is 0, the
 * mira_msm_set_window_bits value, or 16: a partial's width never depends on the chunk length,
 * which differs between ranks -- and either all or none with mira_msm_precompute'd keys (a handle
 * with tables answers with table-mode partials here whatever its chunk length, unless a width is
 * requested).  On return *window_bits / *num_windows describe the partial (0 / 64 for tables). */
int mira_msm_partial_device(uint64_t handle, size_t first, const void *d_scalars, size_t n,
                            uint64_t out_partial[MIRA_PARTIAL_U64], int32_t *window_bits, int32_t *num_windows);
int mira_msm_combine(int curve, const uint64_t *partials /* nparts * MIRA_PARTIAL_U64 */, size_t nparts,
                     int32_t window_bits, int32_t num_windows, uint64_t out_affine[8]);
/* Same, with the partial left in DEVICE memory (d_out_partial: MIRA_PARTIAL_U64 * 8 bytes, e.g. a tensor an
 * RCCL all-gather reads): nothing crosses PCIe before the exchange.  Words beyond the partial's windows are zero. */
int mira_msm_partial_to_device(uint64_t handle, size_t first, const void *d_scalars, size_t n,
                               void *d_out_partial, int32_t *window_bits, int32_t *num_windows);
/* Window width c (4..16) of every later commit over THIS key; 0 = let the planner choose from n and the
 * scalar statistics.  Per handle, so two caller threads working on two keys never see each other's choice. */
int mira_msm_set_handle_window_bits(uint64_t handle, int32_t c);
/* Process-wide default width for keys without one of their own (tests, benchmarks; 0 = planner).  A process
 * whose threads want different widths uses the per-handle call above.  All ranks of a sharded MSM must use the
 * same c. */
int mira_msm_set_window_bits(int32_t c);
/* The width the planner picks for a commit of n uniform scalars when it has no statistics of the data (a pure
 * function of n: every rank of a sharded MSM derives the same width from the same global length). */
int mira_msm_plan_window_bits(size_t n, int32_t *window_bits);
/* Diagnostics: the window width and count the planner used for the most recent commit of this
 * process (0 / the number of partial sums in fixed-base table mode). */
int mira_msm_last_plan(int32_t *window_bits, int32_t *num_windows);
/* ... and the width of the table set that commit went through (8 .. 16 shared buckets, 20 or 22 wide tables), 0 = none. */
int mira_msm_last_table_bits(int32_t *table_bits);

/* Thresholds of the engine's internal choices, for tests and tuning runs (they never change a
 * result): the smallest MSM that takes the LDS-staged two-level sort, the smallest MSM that uses a
 * handle's window tables, the smallest commit whose scalar-length statistics plan the next one, and
 * the longest NTT line (log2) -- shorter lines make the two- and three-pass schedules reachable at
 * small sizes -- and whether lines of up to 256 points take the wave-level kernel (1: wherever it
 * can; 0: never; default: by size); the smallest commit of HOST scalars that is cut into point chunks so
 * that the copy of one chunk overlaps the kernels of the previous one; the largest post-twiddle exponent
 * range (log2) an NTT serves from one table -- beyond it pass 1 reads a table of n entries, or, above
 * 2^24 points, multiplies two table entries.  value < 0 restores the default. */
#define MIRA_TUNE_STAGED_MIN_N 0
#define MIRA_TUNE_TABLE_MIN_N 1
#define MIRA_TUNE_PLAN_HIST_MIN_N 2
#define MIRA_TUNE_NTT_MAX_LOG_LINE 3
#define MIRA_TUNE_NTT_WAVE 4
#define MIRA_TUNE_HOST_CHUNK_MIN_N 5
#define MIRA_TUNE_NTT_SINGLE_TW_LOG 6
/* largest transform (log2) whose first post-twiddle is served from a table of n entries (48 n bytes per cached
 * table set); default 24, 0 = never */
#define MIRA_TUNE_NTT_FULL_TW_MAX_LOG 7
/* serve commits from the shared-bucket table set of exactly this width (calibration, tests); 0 / default = choose by length */
#define MIRA_TUNE_TABLE_WIDTH 8
/* column reads a specialised cross-term kernel keeps in flight ahead of their use (mira_graph_specialize); default 4 */
#define MIRA_TUNE_JIT_LOADS_AHEAD 9
/* smallest number of sorted entries one lane of k_accumulate adds (shorter segments: more lanes busy on a small commit,
 * more cut runs for the fix-up); default 10 */
#define MIRA_TUNE_MIN_SEGMENT 10
/* 0: do not use a key's endomorphism copy (MIRA_TABLE_GLV) -- same-key A/B runs and tests; default 1 */
#define MIRA_TUNE_GLV 11
/* bucket reduction (csrc/reduce_kernels.cuh): results per bucket set handed to the host's chain of doublings (default: by
 * shape; 1 = plain window sums), log2 of the buckets per running-sum chunk (default 2 by quads / 3 by lanes), and whether
 * the running sums are taken by quads of lanes (1) or single lanes (0; default: quads while the chunks are few) */
#define MIRA_TUNE_REDUCE_PIECES 12
#define MIRA_TUNE_REDUCE_LAMBDA 13
#define MIRA_TUNE_REDUCE_QUAD 14
/* smallest commit served from a key's shared-bucket table sets (default 2^12); MIRA_TUNE_TABLE_MIN_N is the wide tables' */
#define MIRA_TUNE_SHARED_MIN_N 15
/* log2 of the sorted entries (pairs x windows) one pass of the pipeline takes; a longer commit is cut into point chunks that
 * add into one set of buckets (default and maximum 32: the entry offsets are 32-bit; tests cut small commits with it) */
#define MIRA_TUNE_PASS_ENTRIES_LOG 16
/* log2 of the longest key whose endomorphism copy (MIRA_TABLE_GLV, 2 x the key's memory) the library builds BY ITSELF, at the
 * first single commit or batch that can use it (commits of at most 2^21 pairs over keys of at least 2^12 points); default 26,
 * 0 = only where mira_msm_precompute_ex(handle, MIRA_TABLE_GLV) was called */
#define MIRA_TUNE_GLV_AUTO_MAX_LOG 17
/* 1 (default): the window width the planner picks for a shape of commit (pairs, commitments per submission, path) over a key is
 * checked against its four neighbours on the first ten commits of that shape -- each candidate timed twice -- and the fastest
 * measured is kept for the rest of the key's life (commits of >= 2^12 pairs, unforced widths, not the ranks of a sharded MSM); 0: the
 * planner's tables alone decide */
#define MIRA_TUNE_WIDTH_TRIALS 18
/* workgroups in the persistent grid of the NTT kernels (default: 256 CUs x the workgroups that fit one).  A pass with at least
 * four block-groups (wave-level kernel) or lines (workgroup-level kernel) per workgroup hands them out through per-XCD counters;
 * tests set a small grid so that this path, its ranges without a home workgroup included, runs at sizes the CPU emulation reaches */
#define MIRA_TUNE_NTT_GRID 19
int mira_set_tuning(int knob, int64_t value);

/* Read a range of the registered key back in the reference layout (cache file writing,
 * save_to_file, src/commitment.rs:96-101). */
int mira_msm_download_bases(uint64_t handle, size_t first, size_t n, uint64_t *bases_out /* n * 8 limbs */);

/* ---- the step after the MSM in one fold: RelaxedPlonkWitness::fold (src/plonk/mod.rs:1097-1134)
 * field: MIRA_FIELD_FQ / MIRA_FIELD_FR.  Vectors stay in HBM between the commits and the fold.
 *   mira_fold_witness_device:  out[i] = w1[i] + r * w2[i]
 *   mira_fold_error_device:    e[i]  += sum_k r^(k+1) * cross_terms[k][i],  k < num_terms <= 16
 * mira_g1_mul_add: out = acc + scalar * point on affine points -- the single-scalar best_multiexp
 * calls of RelaxedPlonkInstance::fold (src/plonk/mod.rs:986-999, 1049-1053); host, O(256).   */
#define MIRA_FIELD_FQ 0
#define MIRA_FIELD_FR 1
int mira_fold_witness_device(int field, void *d_out, const void *d_w1, const void *d_w2, const uint64_t r[4], size_t n);
int mira_fold_error_device(int field, void *d_e, const void *const *d_cross_terms, size_t num_terms, const uint64_t r[4], size_t n);
/* Both halves of RelaxedPlonkWitness::fold in one submission (src/plonk/mod.rs:1097-1134): w_out[i] = w1[i] + r * w2[i], i < n_w,
 * and e_out[i] = e[i] + sum_k r^(k+1) * cross_terms[k][i], i < n, k < num_terms <= 16 (d_e_out may be d_e: in place) -- the values
 * of the two calls above, with one copy of constants and one synchronisation instead of two each. */
int mira_fold_relaxed_witness_device(int field, void *d_w_out, const void *d_w1, const void *d_w2, size_t n_w, void *d_e_out, const void *d_e,
                                     const void *const *d_cross_terms, size_t num_terms, const uint64_t r[4], size_t n);
int mira_g1_mul_add(int curve, const uint64_t acc[8], const uint64_t scalar[4], const uint64_t point[8], uint64_t out[8]);
/* out = acc + sum_i scalars[i] * points[i], count <= 64: E_commit + sum_k r^(k+1) T_k over the cross-term
 * commitments (src/plonk/mod.rs:1049-1053); host: width-5 NAFs, the terms dealt to the library's resident host
 * threads, one chain of doublings per thread. */
int mira_g1_lincomb(int curve, const uint64_t acc[8], const uint64_t *scalars /* count * 4 */, const uint64_t *points /* count * 8 */,
                    size_t count, uint64_t out[8]);
/* The commitment side of RelaxedPlonkInstance::fold in one call (src/plonk/mod.rs:986-999, 1049-1053):
 *   w_out[i] = w1[i] + r * w2[i], i < nw <= 64;   e_out = e + sum_k r^(k+1) * t_commits[k], k < count <= 64
 * -- the same points as nw calls of mira_g1_mul_add and one of mira_g1_lincomb over the powers of r, as ONE parallel
 * region on the host threads (every W commitment and every group of cross-term commitments is a task). */
int mira_g1_fold_commitments(int curve, const uint64_t r[4], const uint64_t *w1 /* nw * 8 */, const uint64_t *w2 /* nw * 8 */, size_t nw,
                             const uint64_t e[8], const uint64_t *t_commits /* count * 8 */, size_t count, uint64_t *w_out /* nw * 8 */, uint64_t e_out[8]);

/* ---- the step before the MSM in one fold: cross-term evaluation ----------------------------
 * GraphEvaluator::evaluate over all rows (src/polynomial/graph_evaluator.rs:361-390, called per
 * row from commit_cross_terms, src/nifs/vanilla/mod.rs:100-121): out[row] = value of the last
 * calculation, for row < num_rows.  The output stays in HBM, ready for mira_msm_batch_device and
 * mira_fold_error_device.
 *
 * mira_graph is the reference structure flattened (graph_evaluator.rs:165-185):
 *   constants   num_constants field elements (Montgomery, as the reference holds them)
 *   rotations   num_rotations i32; a column read at rotation index k uses row
 *               (row + rotations[k]).rem_euclid(num_rows)                  (:51-53)
 *   code        the calculations in order; calculation i writes intermediate i (its `target`).
 *               One header word  op | nparts << 8,  then the operands as value sources:
 *                 ADD, SUB, MUL: a, b.   SQUARE, DOUBLE, NEGATE, STORE: a.
 *                 HORNER: start, factor, parts[nparts]   (value = start; value = value*factor + part)
 *   value source  kind << 29 | payload                                   (ValueSource, :56-68)
 *                 CONSTANT: index into constants.  INTERMEDIATE: index of an earlier calculation.
 *                 CHALLENGE: index into challenges.
 *                 COLUMN: column | rotation_index << 20 -- Fixed{index, rotation} and
 *                 Poly{index, rotation} after the caller has resolved the column (selector, fixed
 *                 or witness slice: eval_column_var, src/plonk/eval.rs:57-69 and :136-229) to an
 *                 entry of `columns`.
 * columns[k]: device pointer to num_rows values; kind MIRA_COL_FIELD = field elements,
 * MIRA_COL_BOOL = bytes (a selector: non-zero -> ONE, zero -> ZERO).  A null pointer marks a column
 * index that does not resolve; using it is ColumnVariableIndexOutOfBoundary / InvalidWitnessIndex
 * -> MIRA_E_BAD_ARG, as are a challenge or constant index out of range
 * (ChallengeIndexOutOfBoundary) and malformed code.  No calculations -> out = 0 (:386-389).    */
#define MIRA_OP_ADD 0u
#define MIRA_OP_SUB 1u
#define MIRA_OP_MUL 2u
#define MIRA_OP_SQUARE 3u
#define MIRA_OP_DOUBLE 4u
#define MIRA_OP_NEGATE 5u
#define MIRA_OP_HORNER 6u
#define MIRA_OP_STORE 7u
#define MIRA_SRC_CONSTANT 0u
#define MIRA_SRC_INTERMEDIATE 1u
#define MIRA_SRC_COLUMN 2u
#define MIRA_SRC_CHALLENGE 3u
#define MIRA_COL_FIELD 0u
#define MIRA_COL_BOOL 1u
typedef struct mira_graph {
    const uint32_t *code;
    size_t code_words;
    uint32_t num_calculations;
    uint32_t num_constants;
    const uint64_t *constants;   /* num_constants * 4 limbs */
    const int32_t *rotations;
    uint32_t num_rotations;
    uint32_t reserved;
} mira_graph;
typedef struct mira_eval_column {
    const void *d_data;
    uint32_t kind;
    uint32_t reserved;
} mira_eval_column;
int mira_graph_eval_device(int field, const mira_graph *graph, const mira_eval_column *columns, uint32_t num_columns,
                           const uint64_t *challenges /* num_challenges * 4 limbs, host */, uint32_t num_challenges,
                           size_t num_rows, void *d_out /* num_rows field elements */);
/* The same in two steps.  A GraphEvaluator is built once per circuit (GraphEvaluator::new,
 * graph_evaluator.rs:196-206) and evaluated at every fold step with new witnesses and challenges:
 * mira_graph_compile validates the structure (everything above except the columns' pointers),
 * allocates the intermediates and uploads the program once; mira_graph_eval_compiled then costs one
 * small copy (challenges, column pointers) and one launch.  The counts of challenges and columns are
 * fixed at compile time; a null column that the code reads is the error described above. */
int mira_graph_compile(int field, const mira_graph *graph, uint32_t num_challenges, uint32_t num_columns, uint64_t *handle_out);
int mira_graph_eval_compiled(uint64_t handle, const mira_eval_column *columns, uint32_t num_columns,
                             const uint64_t *challenges, uint32_t num_challenges, size_t num_rows, void *d_out);
int mira_graph_free(uint64_t handle);
/* The d - 1 cross-term graphs of one fold step (src/nifs/vanilla/mod.rs:100-121 evaluates them one
 * after the other over the same witnesses and challenges) in one submission: count compiled graphs
 * of one field, compiled for the same numbers of challenges and columns, over the same columns;
 * graph k writes d_outs[k].  Row for row the values of count mira_graph_eval_compiled calls. */
int mira_graph_eval_batch(const uint64_t *handles, uint32_t count, const mira_eval_column *columns, uint32_t num_columns,
                          const uint64_t *challenges, uint32_t num_challenges, size_t num_rows, void *const *d_outs);
/* Optional, once per circuit: give every one of these compiled graphs a kernel of its own.  The engine writes the
 * graph's instruction stream out as straight-line HIP and compiles it for the device at run time (hiprtc, one host
 * thread per graph, ~5 s for one evaluation point of a MainGate<5> gate): intermediates live in registers instead of
 * LDS / workspace slots, there is no decoding, a rotated row is reduced once per row instead of once per column read.
 * Evaluation then runs at about twice the interpreter's rate; every value is the interpreter's (the same field
 * operations in the same order).  The gate polynomial of a circuit is fixed for the whole IVC run
 * (src/ivc/public_params.rs: the PlonkStructure is built once), so this belongs where the GraphEvaluators are built.
 * The call blocks for the compilation but releases the library's lock meanwhile: other threads keep committing and
 * evaluating (interpreted), and a handle freed before the compiler is done is skipped.
 * Soft failures, each with its own code and the reason in mira_last_error(): MIRA_E_JIT_UNAVAILABLE (no libhiprtc.so on this
 * machine), MIRA_E_UNSUPPORTED (a graph of more than 1536 instructions), MIRA_E_JIT_FAILED (the compiler or the module loader
 * failed: the text carries the compiler's log).  In all three nothing has changed and the graphs keep being interpreted --
 * on the GPU, there is no host path.  The three kernel headers the generated source includes are embedded in the library.
 * `columns`: the column table the graphs will be evaluated over -- only the KINDS are read (which columns are selector
 * bytes is a property of the circuit); an evaluation over columns of other kinds than the kernel was built for is
 * interpreted.
 * mira_graph_is_specialized reports 1 / 0; mira_graph_jit_source copies the generated source (NUL-terminated, at most
 * cap bytes; *len_out = its full length) for inspection and for tests. */
int mira_graph_specialize(const uint64_t *handles, uint32_t count, const mira_eval_column *columns, uint32_t num_columns);
int mira_graph_is_specialized(uint64_t handle, int32_t *out);
int mira_graph_jit_source(uint64_t handle, const mira_eval_column *columns, uint32_t num_columns, char *buf, size_t cap, size_t *len_out);
/* Code objects on disk.  mira_graph_set_cache_dir(dir): every kernel mira_graph_specialize compiles from now on is also
 * written to `dir` (which must exist), and a graph whose kernel lies there is loaded instead of compiled -- the second
 * process of an IVC run over the same circuit specialises in milliseconds.  A file is taken only if the generated source
 * and the build environment -- the kernel headers embedded in libmira_gpu.so, the compiler options, the GPU architecture
 * of the bound device, the HIP runtime and hiprtc versions -- are the ones it was built from, byte for byte, and its code
 * hashes to what its header says; writing is best effort (an unwritable directory costs compilations, not errors).  A code
 * object is executed as found, so the directory must belong to the calling user and be writable by nobody else (else
 * MIRA_E_BAD_ARG; not a directory: MIRA_E_IO), and a file there that is somebody else's or group / world-writable is
 * ignored.  NULL or "" (the default): no files.  The reference keeps its commitment keys the same way
 * (src/commitment.rs:96-167, `.cache/`).
 * mira_graph_jit_stats: how many kernels the last mira_graph_specialize compiled / read from the directory.
 * mira_graph_jit_compile_check: compile a source text through the library's own path (hiprtc + the embedded headers) and
 * report the size of the code object; needs no GPU (build checks, tests). */
int mira_graph_set_cache_dir(const char *dir);
int mira_graph_jit_stats(uint32_t *compiled_out, uint32_t *from_disk_out);
int mira_graph_jit_compile_check(const char *source, size_t *code_size_out);


/* ---- ProtoGalaxy's polynomial pipeline around the NTT (src/nifs/protogalaxy/poly/mod.rs) -------
 * mira_pow_tree_reduce_device: the weighted tree reduction of compute_F (:131-166) and compute_G
 * (:263-290), itertools::tree_reduce with node(left, right) = left + right * weights[p][height]:
 *     out[p] = sum_i leaves[p][i] * prod_{j : bit j of i} weights[p][j],   i < n_leaves = 2^levels
 * leaves: device; point p's leaves start at element p * leaf_point_stride, 0 = every point reads
 * the same leaves (compute_F: the challenge sits on the edges; compute_G: in the leaves).
 * weights: host, num_points x levels elements.  out: host, num_points elements -- the input of
 * the small ifft that yields the coefficients.  n_leaves must be a power of two (the reference's
 * tree is `unreachable!` otherwise) -> MIRA_E_UNSUPPORTED.
 * mira_lincomb_device: out[i] = sum_k coeffs[k] * vecs[k][i], k < num_vecs <= 16 -- the witness of
 * FoldedTrace at one challenge, L_0(X) * acc + sum_j L_j(X) * trace_j (folded_trace.rs:54-131). */
int mira_pow_tree_reduce_device(int field, const void *d_leaves, size_t n_leaves, size_t leaf_point_stride,
                                const uint64_t *weights, uint32_t num_points, uint64_t *out);
int mira_lincomb_device(int field, void *d_out, const void *const *d_vecs, const uint64_t *coeffs /* num_vecs * 4 limbs, host */,
                        size_t num_vecs, size_t n);
/* num_outs <= 8 linear combinations of the same num_vecs <= 16 vectors in one sweep: outs[m][i] = sum_k coeffs[m][k] * vecs[k][i]
 * (coeffs: num_outs x num_vecs elements, row-major, host).  An output must not alias an input.  The interpolation step of the
 * cross terms: commit_cross_terms' d vectors (src/nifs/vanilla/mod.rs:100-121) are the coefficients of
 * f(W1 + X W2, c1 + X c2) in X, i.e. fixed linear combinations of d + 1 values of that polynomial per row. */
int mira_lincomb_multi_device(int field, void *const *d_outs, size_t num_outs, const void *const *d_vecs, size_t num_vecs,
                              const uint64_t *coeffs, size_t n);

/* ---- NTT over bn256::Fr (src/fft.rs) -----------------------------------------------------
 * In place, natural order in and out.  `a` = 2^log_n elements, log_n <= 28 = Fr::S as in the
 * reference (src/fft.rs:13); the device needs a second buffer of the same size above 2^12.      */
/* best_fft(a, omega, log_n), src/fft.rs:51 */
int mira_ntt_bn256_fr(uint64_t *a, uint32_t log_n, const uint64_t omega[4]);
int mira_ntt_bn256_fr_device(void *d_a, uint32_t log_n, const uint64_t omega[4]);
/* fft / ifft, src/fft.rs:160-174 (omega from get_omega_or_inv, ifft scales by TWO_INV^log_n) */
int mira_fft_bn256_fr(uint64_t *a, uint32_t log_n);
int mira_ifft_bn256_fr(uint64_t *a, uint32_t log_n);
int mira_fft_bn256_fr_device(void *d_a, uint32_t log_n);
int mira_ifft_bn256_fr_device(void *d_a, uint32_t log_n);
/* coset_fft / coset_ifft, src/fft.rs:178-196 (ZETA pattern of distribute_powers_zeta, :205-226) */
int mira_coset_fft_bn256_fr(uint64_t *a, uint32_t log_n);
int mira_coset_ifft_bn256_fr(uint64_t *a, uint32_t log_n);
/* get_omega_or_inv(k, is_inverse), src/fft.rs:12-23; Montgomery form out */
int mira_get_omega_or_inv(uint32_t k, int is_inverse, uint64_t out[4]);

/* ---- synthetic inputs for benchmarks and tests -------------------------
 * kind 0 = uniform field elements, 1 = witness-like (70 % zero, 20 % < 2^32, 10 % uniform).
 * index0 = global index of the first element (lets ranks generate their own chunk).          */
int mira_synth_scalars_device(int curve, size_t n, uint64_t index0, uint64_t seed, int kind, void *d_out);
int mira_synth_bases_device(int curve, size_t n, uint64_t index0, uint64_t seed, void *d_out);

/* ---- library-held device memory ------------------------------------------------------------
 * Workspaces (digit, sort and bucket buffers; NTT temporaries and the four cached twiddle-table sets,
 * up to 805 MB each at 2^24 points) grow on demand and are reused by later calls: after one commit of
 * 2^26 pairs the library holds about 6 GiB.  mira_trim releases the largest of them until at most
 * keep_bytes remain (0 = everything; registered keys, their window tables and compiled graphs are never
 * touched) and reports the bytes released; the next call re-allocates what it needs.
 * mira_dev_mem_info: hipMemGetInfo of the bound device. */
int mira_trim(size_t keep_bytes, size_t *released_out /* may be NULL */);
int mira_dev_mem_info(size_t *free_bytes, size_t *total_bytes);

/* ---- device memory helpers (thin hipMalloc/hipMemcpy wrappers for non-torch callers) ---- */
int mira_dev_alloc(size_t bytes, void **d_out);
int mira_dev_free(void *d);
int mira_dev_upload(void *d_dst, const void *h_src, size_t bytes);
int mira_dev_download(void *h_dst, const void *d_src, size_t bytes);
int mira_dev_copy(void *d_dst, const void *d_src, size_t bytes);   /* device to device */
int mira_dev_sync(void);

/* ---- measurement ------------------------------------------------------------------------
 * With timing on, every stage of the next MSM / NTT is bracketed by HIP events on the work
 * stream.  mira_get_timings returns the stage count and fills names/ms (ms[i] = device time of
 * stage i of the most recent call).  Stage names are static strings.                         */
int mira_set_timing(int enabled);
int mira_get_timings(const char **names, float *ms, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* MIRA_GPU_H */
