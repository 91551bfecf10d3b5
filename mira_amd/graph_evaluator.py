"""Bindings of the cross-term evaluator, the step before the MSM in one fold:

* `GraphEvaluator`: a calculation graph in the flattened form of `include/mira_gpu.h` (constants, rotations, the
  calculation list of src/polynomial/graph_evaluator.rs:93-176) and its life on the device -- compiled once per
  circuit (`mira_graph_compile`), evaluated for every row at once (`mira_graph_eval_compiled` /
  `mira_graph_eval_batch`), optionally through a kernel of its own (`mira_graph_specialize`).  It replaces the
  per-row `evaluate` loop of `commit_cross_terms` (src/nifs/vanilla/mod.rs:100-121).  Building the graph from an
  `Expression` (GraphEvaluator::new, graph_evaluator.rs:196-352) is symbolic host work of the reference that this
  engine does not replace: the test / bench harness restates it in `harness/graph_evaluator.py`.
* `PlonkEvalDomain`: the column index space of `eval_column_var` / `eval_advice_var`
  (src/plonk/eval.rs:57-69, 136-229) resolved to device pointers;
* `commit_cross_terms`: evaluate every cross term into HBM and commit them in one batched MSM;
* `CrossTermPlan`: the device side of "d cross terms from d + 1 evaluations" (graphs and interpolation coefficients
  are handed in; `harness/graph_evaluator.py` derives them from the gate polynomial).

Constants and challenges are plain Python integers below the field modulus on this side; device
data is in the reference's Montgomery layout like everywhere else in this package."""
import ctypes
import os

import numpy as np

from . import _lib

FIELD_FQ, FIELD_FR = 0, 1
MODULUS = {
    FIELD_FQ: 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47,   # bn256::Fq
    FIELD_FR: 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001,   # bn256::Fr
}
OP_ADD, OP_SUB, OP_MUL, OP_SQUARE, OP_DOUBLE, OP_NEGATE, OP_HORNER, OP_STORE = range(8)
SRC_CONSTANT, SRC_INTERMEDIATE, SRC_COLUMN, SRC_CHALLENGE = range(4)
COL_FIELD, COL_BOOL = 0, 1


def to_montgomery(values, field):
    """ints -> (n, 4) uint64 limbs of v * 2^256 mod p, the layout the device reads."""
    mod = MODULUS[field]
    out = np.zeros((len(values), 4), dtype=np.uint64)
    for i, v in enumerate(values):
        m = (v % mod) * (1 << 256) % mod
        for k in range(4):
            out[i, k] = (m >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


# ---------------------------------------------------------------- GraphEvaluator
# value sources are tuples ordered like the reference's derived PartialOrd (graph_evaluator.rs:55-68):
# (SRC_CONSTANT, id) < (SRC_INTERMEDIATE, id); columns and challenges only appear inside Store
_ZERO, _ONE, _TWO = (SRC_CONSTANT, 0), (SRC_CONSTANT, 1), (SRC_CONSTANT, 2)


class GraphEvaluator:
    """constants: ints (the reference's table starts 0, 1, 2, graph_evaluator.rs:183-192); rotations: ints;
    calculations: tuples (op, source...) with sources (SRC_*, id) or (SRC_COLUMN, index, rotation id);
    calculation i writes intermediate i."""

    def __init__(self, field=FIELD_FR, constants=(0, 1, 2), rotations=(), calculations=()):
        self.field = field
        self.mod = MODULUS[field]
        self.constants = list(constants)
        self.rotations = list(rotations)
        self.calculations = list(calculations)

    @property
    def num_intermediates(self):
        return len(self.calculations)

    # ---- the flattened form of include/mira_gpu.h ------------------------------------------
    @staticmethod
    def _source_word(src):
        if src[0] == SRC_COLUMN:
            return (SRC_COLUMN << 29) | src[1] | (src[2] << 20)
        return (src[0] << 29) | src[1]

    def flatten(self):
        """-> (code uint32[], constants (n, 4) uint64, rotations int32[])"""
        words = []
        for calc in self.calculations:
            op, srcs = calc[0], calc[1:]
            words.append(op | ((len(srcs) - 2) << 8 if op == OP_HORNER else 0))
            words.extend(self._source_word(s) for s in srcs)
        return (np.array(words, dtype=np.uint32), to_montgomery(self.constants, self.field), np.array(self.rotations, dtype=np.int32))

    def compiled(self, num_challenges, num_columns, lib):
        """mira_graph_compile once per (evaluator, library, shape): the graph is built once per circuit
        (GraphEvaluator::new) and evaluated at every fold step."""
        cache = self.__dict__.setdefault("_compiled", {})
        key = (id(lib), num_challenges, num_columns)
        if key not in cache:
            code, consts, rots = self.flatten()
            g = _lib.MiraGraph(code.ctypes.data_as(ctypes.c_void_p), len(code), len(self.calculations), len(consts),
                               consts.ctypes.data_as(ctypes.c_void_p), rots.ctypes.data_as(ctypes.c_void_p), len(rots), 0)
            h = ctypes.c_uint64()
            lib.check(lib.c.mira_graph_compile(self.field, ctypes.byref(g), num_challenges, num_columns, ctypes.byref(h)))
            cache[key] = (h.value, lib)
        return cache[key][0]

    def close(self):
        for h, lib in self.__dict__.pop("_compiled", {}).values():
            lib.c.mira_graph_free(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def evaluate_device(self, columns, challenges, num_rows, d_out=None, lib=None):
        """Every row at once.  columns: list of (device pointer, COL_FIELD | COL_BOOL) or None for an
        index that does not resolve; challenges: ints.  Returns the device pointer of the
        num_rows results (allocated here unless `d_out` is given)."""
        lib = lib or _lib.load()
        handle = self.compiled(len(challenges), len(columns), lib)
        cols = (_lib.MiraEvalColumn * max(1, len(columns)))()
        for k, c in enumerate(columns):
            cols[k].d_data, cols[k].kind = (None, 0) if c is None else (c[0], c[1])
        ch = to_montgomery(list(challenges), self.field)
        out = d_out if d_out is not None else lib.alloc(max(1, num_rows) * 32)
        try:
            lib.check(lib.c.mira_graph_eval_compiled(handle, cols, len(columns), ch.ctypes.data_as(ctypes.c_void_p), len(ch), num_rows, ctypes.c_void_p(out)))
        except Exception:
            if d_out is None:
                lib.free(out)
            raise
        return out

    @staticmethod
    def evaluate_batch_device(evaluators, columns, challenges, num_rows, d_outs, lib=None):
        """The cross-term graphs of one fold step (src/nifs/vanilla/mod.rs:100-121 walks them one after
        the other over the same data) in one submission; evaluator k writes d_outs[k]."""
        lib = lib or _lib.load()
        if not evaluators:
            return
        handles = (ctypes.c_uint64 * len(evaluators))(*[ev.compiled(len(challenges), len(columns), lib) for ev in evaluators])
        cols = (_lib.MiraEvalColumn * max(1, len(columns)))()
        for k, c in enumerate(columns):
            cols[k].d_data, cols[k].kind = (None, 0) if c is None else (c[0], c[1])
        ch = to_montgomery(list(challenges), evaluators[0].field)
        outs = (ctypes.c_void_p * len(evaluators))(*d_outs)
        lib.check(lib.c.mira_graph_eval_batch(handles, len(evaluators), cols, len(columns), ch.ctypes.data_as(ctypes.c_void_p), len(ch), num_rows, outs))

    @staticmethod
    def _column_table(columns):
        cols = (_lib.MiraEvalColumn * max(1, len(columns)))()
        for k, c in enumerate(columns):
            cols[k].d_data, cols[k].kind = (None, 0) if c is None else (c[0], c[1])
        return cols

    @staticmethod
    def specialize(evaluators, columns, num_challenges, lib=None):
        """mira_graph_specialize: every one of these graphs gets a kernel of its own, compiled for the device at run
        time from its instruction stream (once per circuit: the gate polynomial is fixed for the whole IVC run).  Only
        the KINDS of `columns` matter here.  Returns True when the graphs are specialised; False -- with a warning that
        carries the library's reason -- when this machine has no run-time compiler or a graph is too long to specialise
        (they stay interpreted: same values, about half the rate).  A compiler or loader FAILURE raises MiraError with
        the compiler's log."""
        lib = lib or _lib.load()
        evaluators = [ev for ev in evaluators if ev is not None]
        if not evaluators:
            return True
        handles = (ctypes.c_uint64 * len(evaluators))(*[ev.compiled(num_challenges, len(columns), lib) for ev in evaluators])
        rc = lib.c.mira_graph_specialize(handles, len(evaluators), GraphEvaluator._column_table(columns), len(columns))
        if rc in (_lib.MIRA_E_JIT_UNAVAILABLE, _lib.MIRA_E_UNSUPPORTED):
            import warnings
            warnings.warn("mira_graph_specialize: " + (lib.c.mira_last_error() or b"").decode(errors="replace") + " -- graphs stay interpreted", RuntimeWarning, stacklevel=2)
            return False
        lib.check(rc)
        return True

    @staticmethod
    def set_jit_cache_dir(path, lib=None):
        """mira_graph_set_cache_dir: keep the code objects of specialised kernels in this directory (None: no files)."""
        lib = lib or _lib.load()
        lib.check(lib.c.mira_graph_set_cache_dir(None if path is None else os.fsencode(path)))

    @staticmethod
    def jit_stats(lib=None):
        """mira_graph_jit_stats: (kernels compiled, kernels read from the cache directory) by the last specialize call."""
        lib = lib or _lib.load()
        compiled, from_disk = ctypes.c_uint32(), ctypes.c_uint32()
        lib.check(lib.c.mira_graph_jit_stats(ctypes.byref(compiled), ctypes.byref(from_disk)))
        return compiled.value, from_disk.value

    def is_specialized(self, num_challenges, num_columns, lib=None):
        lib = lib or _lib.load()
        out = ctypes.c_int32()
        lib.check(lib.c.mira_graph_is_specialized(self.compiled(num_challenges, num_columns, lib), ctypes.byref(out)))
        return bool(out.value)

    def jit_source(self, columns, num_challenges, lib=None):
        """mira_graph_jit_source: the HIP source mira_graph_specialize would compile for this graph over columns of these kinds."""
        lib = lib or _lib.load()
        h = self.compiled(num_challenges, len(columns), lib)
        cols, n = GraphEvaluator._column_table(columns), ctypes.c_size_t()
        lib.check(lib.c.mira_graph_jit_source(h, cols, len(columns), None, 0, ctypes.byref(n)))
        buf = ctypes.create_string_buffer(n.value + 1)
        lib.check(lib.c.mira_graph_jit_source(h, cols, len(columns), buf, n.value + 1, ctypes.byref(n)))
        return buf.value.decode()

    def evaluate(self, getter, lib=None):
        """Host-array convenience: getter = dict(selectors=[bool arrays], fixed=[(n, 4) uint64],
        advice=[(n, 4) uint64], challenges=[ints]) as the reference's test mock
        (graph_evaluator.rs:405-443); returns the (num_rows, 4) results."""
        lib = lib or _lib.load()
        sel, fix, adv = getter.get("selectors", []), getter.get("fixed", []), getter.get("advice", [])
        num_rows = len(fix[0]) if fix else len(sel[0]) if sel else (len(adv[0]) if adv else 1)   # row_size, src/plonk/eval.rs:47-54
        ptrs, cols = [], []
        try:
            for s in sel:
                a = np.ascontiguousarray(np.asarray(s), dtype=np.uint8)
                ptrs.append(lib.alloc(max(1, a.nbytes))); lib.upload(ptrs[-1], a); cols.append((ptrs[-1], COL_BOOL))
            for f in list(fix) + list(adv):
                a = np.ascontiguousarray(f, dtype=np.uint64).reshape(-1, 4)
                ptrs.append(lib.alloc(max(1, a.nbytes))); lib.upload(ptrs[-1], a); cols.append((ptrs[-1], COL_FIELD))
            d = self.evaluate_device(cols, getter.get("challenges", []), num_rows, lib=lib)
            ptrs.append(d)
            return lib.download(d, (num_rows, 4))
        finally:
            for p in ptrs:
                lib.free(p)


# ---------------------------------------------------------------- PlonkEvalDomain (src/plonk/eval.rs:93-229)
class PlonkEvalDomain:
    """Device-resident evaluation data of one fold step.  selectors: device byte columns, fixed:
    device field columns, W1s / W2s: lists of (device pointer, length in elements) -- the witness
    vectors of the two instances, each a concatenation of row_size-long columns."""

    def __init__(self, num_advice, num_lookup, challenges, selectors, fixed, W1s, W2s, row_size):
        self.num_advice, self.num_lookup = num_advice, num_lookup
        self.challenges, self.selectors, self.fixed = list(challenges), list(selectors), list(fixed)
        self.W1s, self.W2s, self.row_size = list(W1s), list(W2s), row_size

    def _advice(self, index):
        """eval_advice_var (src/plonk/eval.rs:152-228): advice index -> (device pointer) or None."""
        max_width = self.num_advice + self.num_lookup * 5
        first = index < max_width
        if not first:
            index -= max_width
        ws = self.W1s if first else self.W2s
        if index < self.num_advice:
            i, j = 0, index
        else:
            lookup, sub = divmod(index - self.num_advice, 5)
            first_round = sub < 3
            if not first_round:
                sub -= 3
            if len(ws) == 2:
                i, j = (0, self.num_advice + lookup * 3 + sub) if first_round else (1, lookup * 2 + sub)
            elif len(ws) == 3:
                i, j = (1, lookup * 3 + sub) if first_round else (2, lookup * 2 + sub)
            else:
                return None                                     # Error::InvalidWitnessIndex
        if i >= len(ws) or ws[i][1] < (j + 1) * self.row_size:
            return None
        return ws[i][0] + j * self.row_size * 32

    def columns(self):
        """The column table for `evaluate_device`: selectors, fixed, then both instances' advice."""
        cols = [(p, COL_BOOL) for p in self.selectors] + [(p, COL_FIELD) for p in self.fixed]
        for a in range(2 * (self.num_advice + self.num_lookup * 5)):
            p = self._advice(a)
            cols.append(None if p is None else (p, COL_FIELD))
        return cols


def commit_cross_terms(key, evaluators, domain, lib=None):
    """The evaluation and commit spans of commit_cross_terms (src/nifs/vanilla/mod.rs:100-127):
    evaluators[k] is the GraphEvaluator of cross term k, or None for an absent term (a zero vector,
    :115) -- or `evaluators` is a CrossTermPlan, which produces the same d vectors from d + 1 evaluations of the gate
    polynomial.  Returns (device pointer of the len(evaluators) x row_size cross terms, commitments
    (len, 8) uint64); the caller frees the pointer (mira_fold_error_device consumes it first)."""
    lib = lib or _lib.load()
    if isinstance(evaluators, CrossTermPlan):
        n, count = domain.row_size, evaluators.degree
        d = lib.alloc(max(1, n * count) * 32)
        try:
            evaluators.evaluate_device(domain.columns(), domain.challenges, n, d, lib=lib)
            return d, key.commit_batch_device(d, n, count)
        except Exception:
            lib.free(d)
            raise
    n, count = domain.row_size, len(evaluators)
    d = lib.alloc(max(1, n * count) * 32)
    try:
        cols = domain.columns()
        zero = None
        for k, ev in enumerate(evaluators):
            if ev is None:
                zero = zero if zero is not None else np.zeros((n, 4), dtype=np.uint64)
                lib.upload(d + k * n * 32, zero)
        live = [k for k, ev in enumerate(evaluators) if ev is not None]
        GraphEvaluator.evaluate_batch_device([evaluators[k] for k in live], cols, domain.challenges, n, [d + k * n * 32 for k in live], lib=lib)
        commits = key.commit_batch_device(d, n, count) if count else np.zeros((0, 8), dtype=np.uint64)
    except Exception:
        lib.free(d)
        raise
    return d, commits


# ---------------------------------------------------------------- cross terms by evaluation + interpolation
class CrossTermPlan:
    """All d cross terms of one fold step from d + 1 evaluations of the homogeneous gate polynomial f itself.

    The reference multiplies f(W1 + X W2, c1 + X c2) out symbolically (GroupedPoly::new, src/polynomial/
    grouped_poly.rs:88-140) and evaluates every coefficient T_k as a graph of its own (src/nifs/vanilla/mod.rs:
    100-121): for the MainGate<5> circuits 815 (one gate) and 2 448 (two gates) calculations per row, against 81
    and 162 for f.  The coefficients of a degree-d polynomial in X are also fixed by d + 1 of its values, so here
        p_0 = f(W1, c1) = T_0,    p_inf = f(W2, c2) = T_d,    p_x = f(W1 + x W2, c1 + x c2),  x = 1, -1, 2, -2, ... (d - 1 points)
    are evaluated row by row (d + 1 graphs over the same columns, one mira_graph_eval_batch) and
        T_k = sum_x Ainv[k][x] (p_x - T_0 - x^d T_d),  A[x][k] = x^k,  1 <= k <= d - 1
    is one linear combination of those d + 1 vectors per cross term (mira_lincomb_multi_device).  Field arithmetic is
    exact, so every T_k is the reference's value bit for bit (tests compare with the grouped graphs and the oracle);
    the work is (d + 1) (|f| + one fold per queried column) + d - 1 streamed passes instead of sum_k |T_k|.

    This class is the device side: `evaluators` are the d + 1 graphs in the order p_0, p_inf, p_x ..., `coeffs[k - 1]` the
    integer coefficients of T_k on those vectors.  harness/graph_evaluator.py derives both from the gate polynomial."""

    def __init__(self, degree, field, evaluators, coeffs):
        self.degree, self.field, self.mod = degree, field, MODULUS[field]
        self.evaluators = list(evaluators)
        self.coeffs = [list(row) for row in coeffs]
        self._coeffs_mont = {}                                             # the same in the library's form, converted once

    @property
    def num_calculations(self):
        return [ev.num_intermediates for ev in self.evaluators]

    def specialize(self, columns, num_challenges, lib=None):
        """One run-time compiled kernel per evaluation point (GraphEvaluator.specialize); once per circuit."""
        return GraphEvaluator.specialize(self.evaluators, columns, num_challenges, lib=lib)

    def evaluate_device(self, columns, challenges, num_rows, d_terms, lib=None):
        """d_terms: device buffer of degree x num_rows elements; term k (1 .. d) lands at d_terms + (k - 1) * num_rows * 32,
        the layout commit_cross_terms hands to mira_msm_batch_device and mira_fold_error_device."""
        lib = lib or _lib.load()
        d, n = self.degree, num_rows
        if d == 1:                                                         # one cross term: the leading coefficient
            self.evaluators[1].evaluate_device(columns, challenges, n, d_out=d_terms, lib=lib)
            return
        d_p = self._scratch(lib, (d + 1) * max(1, n) * 32)                 # kept between fold steps: an allocation and a release of 29 MiB cost 0.3 ms per call
        outs = [d_p + j * n * 32 for j in range(d + 1)]
        outs[1] = d_terms + (d - 1) * n * 32                               # p_inf IS T_d
        GraphEvaluator.evaluate_batch_device(self.evaluators, columns, challenges, n, outs, lib=lib)
        vecs = (ctypes.c_void_p * (d + 1))(*outs)
        for k0 in range(0, d - 1, 8):                                      # mira_lincomb_multi_device: up to 8 terms per sweep over the d + 1 vectors
            ks = range(k0, min(d - 1, k0 + 8))
            c = self._coeffs_mont.get(k0)
            if c is None:
                c = self._coeffs_mont[k0] = to_montgomery([v for k in ks for v in self.coeffs[k]], self.field)
            dst = (ctypes.c_void_p * len(ks))(*[d_terms + k * n * 32 for k in ks])
            lib.check(lib.c.mira_lincomb_multi_device(self.field, dst, len(ks), vecs, d + 1, c.ctypes.data_as(ctypes.c_void_p), n))

    def _scratch(self, lib, nbytes):
        have = self.__dict__.get("_scratch_buf")
        if have and have[0] is lib and have[2] >= nbytes:
            return have[1]
        if have:
            have[0].free(have[1])
        self._scratch_buf = (lib, lib.alloc(nbytes), nbytes)
        return self._scratch_buf[1]

    def close(self):
        """Release the evaluation scratch and the compiled graphs."""
        have = self.__dict__.pop("_scratch_buf", None)
        if have:
            have[0].free(have[1])
        for ev in self.evaluators:
            ev.close()

    def __del__(self):
        try:
            have = self.__dict__.pop("_scratch_buf", None)
            if have:
                have[0].free(have[1])
        except Exception:
            pass
