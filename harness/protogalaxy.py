"""TEST / BENCH HARNESS (not the engine).  Host-side mirror of ProtoGalaxy's polynomial pipeline, the only caller of
the reference's FFT: `compute_F`, `compute_G`, `compute_K` (reference
src/nifs/protogalaxy/poly/mod.rs:66-179, 218-303, 339-382), the Lagrange helpers they use
(src/polynomial/lagrange.rs) and `FoldedTrace` (src/nifs/protogalaxy/poly/folded_trace.rs).

The heavy parts run on the GPU through the C ABI: every gate over every row
(`mira_graph_eval_device`), the folded witnesses (`mira_lincomb_device`), the weighted tree
reduction over all gate evaluations (`mira_pow_tree_reduce_device`) and the small transforms
(`mira_ifft_bn256_fr`, `mira_coset_*`).  What stays here is O(points) scalar arithmetic on Python
integers.  The field is bn256::Fr (the only one of the two with an FFT domain).

Data model: `Structure` = the parts of `PlonkStructure` this path reads; a trace = `Trace`
(challenges as integers, witness vectors resident in HBM)."""
import ctypes

import numpy as np

from mira_amd import _lib

from . import graph_evaluator as G

FIELD = G.FIELD_FR
R_MOD = G.MODULUS[FIELD]
ZETA = 0x30644E72E131A029048B6E193FD84104CC37A73FEC2BC5E9B8CA0B2D36636F23   # Fr::ZETA, as in csrc/ntt.hip


def _to_int(limbs):
    """(4,) uint64 Montgomery limbs -> integer"""
    v = sum(int(x) << (64 * i) for i, x in enumerate(limbs))
    return v * pow(1 << 256, -1, R_MOD) % R_MOD


def _ints(arr):
    return [_to_int(row) for row in np.asarray(arr, dtype=np.uint64).reshape(-1, 4)]


def get_omega(log_n, lib):
    out = np.zeros(4, dtype=np.uint64)
    lib.check(lib.c.mira_get_omega_or_inv(log_n, 0, out.ctypes.data_as(ctypes.c_void_p)))
    return _to_int(out)


# ---------------------------------------------------------------- src/polynomial/lagrange.rs
def iter_cyclic_subgroup(log_n, lib):
    """[1, w, w^2, ...] of length 2^log_n (:22-26)"""
    w, out, v = get_omega(log_n, lib), [], 1
    for _ in range(1 << log_n):
        out.append(v)
        v = v * w % R_MOD
    return out


def eval_vanish_polynomial(log_n, point):
    """X^(2^log_n) - 1 (:81-83)"""
    return (pow(point, 1 << log_n, R_MOD) - 1) % R_MOD


def eval_lagrange_poly_for_cyclic_group(X, log_n, lib):
    """[L_0(X), ..., L_(n-1)(X)] over the cyclic subgroup (:50-74)"""
    n = 1 << log_n
    inv_n = pow(n, -1, R_MOD)
    z = eval_vanish_polynomial(log_n, X)
    out = []
    for value in iter_cyclic_subgroup(log_n, lib):
        d = (X - value) % R_MOD
        if z == 0 and d == 0:
            out.append(1)                                       # the 0/0 case, :64-66
        else:
            out.append(value * inv_n % R_MOD * (z * pow(d, -1, R_MOD) % R_MOD) % R_MOD)
    return out


# ---------------------------------------------------------------- data
class Structure:
    """k, gates (Expressions), selectors (device byte columns), fixed (device field columns),
    num_advice_columns, num_lookups -- what compute_F/G/K read of PlonkStructure."""

    def __init__(self, k, gates, selectors, fixed, num_advice, num_lookup=0):
        self.k, self.gates = k, list(gates)
        self.selectors, self.fixed = list(selectors), list(fixed)
        self.num_advice, self.num_lookup = num_advice, num_lookup

    def max_degree(self):
        """max over gates of Expression::degree (src/polynomial/expression.rs:431-447): advice and
        lookup queries and challenges count 1, selectors / fixed / constants 0."""
        first_advice = len(self.selectors) + len(self.fixed)

        def deg(e):
            if isinstance(e, G.Constant):
                return 0
            if isinstance(e, G.Polynomial):
                return 1 if e.index >= first_advice else 0
            if isinstance(e, G.Challenge):
                return 1
            if isinstance(e, (G.Negated, G.Scaled)):
                return deg(e.a)
            if isinstance(e, G.Sum):
                return max(deg(e.a), deg(e.b))
            if isinstance(e, G.Product):
                return deg(e.a) + deg(e.b)
            raise TypeError(e)
        return max((deg(g) for g in self.gates), default=0)


class Trace:
    """GetChallenges + GetWitness: challenges (ints), W = [(device pointer, length in elements)]"""

    def __init__(self, challenges, W):
        self.challenges, self.W = list(challenges), list(W)


def _evaluate_gates(S, trace, d_out, lib):
    """plonk::iter_evaluate_witness (src/plonk/mod.rs:1158-1178): [gate1(row0..), gate2(row0..), ...]
    written to d_out, 2^k values per gate."""
    rows = 1 << S.k
    dom = G.PlonkEvalDomain(S.num_advice, S.num_lookup, trace.challenges, S.selectors, S.fixed, trace.W, [], rows)
    cols = dom.columns()
    for g, gate in enumerate(S.gates):
        G.GraphEvaluator.new(gate, FIELD).evaluate_device(cols, trace.challenges, rows, d_out=d_out + g * rows * 32, lib=lib)


def _tree_reduce(d_leaves, n_leaves, point_stride, weights, lib):
    """-> ints, one per point"""
    w = G.to_montgomery([x for row in weights for x in row], FIELD)
    out = np.zeros((len(weights), 4), dtype=np.uint64)
    lib.check(lib.c.mira_pow_tree_reduce_device(FIELD, ctypes.c_void_p(d_leaves), n_leaves, point_stride, w.ctypes.data_as(ctypes.c_void_p),
                                                len(weights), out.ctypes.data_as(ctypes.c_void_p)))
    return out


def _ifft_small(points_mont, log_n, lib):
    a = np.ascontiguousarray(points_mont, dtype=np.uint64).reshape(-1, 4).copy()
    lib.check(lib.c.mira_ifft_bn256_fr(a.ctypes.data_as(ctypes.c_void_p), log_n))
    return a


# ---------------------------------------------------------------- compute_F (:66-179)
def compute_F(betas, delta, S, trace, lib=None):
    """F(X) = sum_i pow_i(beta + X delta) f_i(w); returns the coefficients as integers."""
    lib = lib or _lib.load()
    rows, gates = 1 << S.k, len(S.gates)
    count = rows * gates
    if count == 0:
        return []
    levels = (count - 1).bit_length()                          # count.next_power_of_two().ilog2()
    points_count = 1 << (levels - 1).bit_length() if levels > 1 else 1   # .next_power_of_two()
    log_points = points_count.bit_length() - 1
    if log_points == 0:
        raise ValueError("fft_domain_size must be non-zero (NonZeroU32 in the reference)")
    betas = list(betas)[:levels]
    weights = [[(b + X * delta) % R_MOD for b in betas] for X in iter_cyclic_subgroup(log_points, lib)]
    d = lib.alloc(count * 32)
    try:
        _evaluate_gates(S, trace, d, lib)
        points = _tree_reduce(d, count, 0, weights, lib)       # every challenge reads the same leaves
    finally:
        lib.free(d)
    return _ints(_ifft_small(points, log_points, lib))


# ---------------------------------------------------------------- FoldedTrace (folded_trace.rs)
def fold_traces(points, accumulator, traces, lib):
    """For every X in `points`: the trace L_0(X) acc + sum_j L_j(X) trace_j (witness vectors on the
    device, challenges on the host).  Returns (list of Trace, device pointers to free)."""
    log_n = (len(traces)).bit_length()                         # (len + 1).next_power_of_two().ilog2()
    if (1 << log_n) < len(traces) + 1:
        log_n += 1
    all_traces = [accumulator] + list(traces)
    folded, owned = [], []
    for X in points:
        L = eval_lagrange_poly_for_cyclic_group(X, log_n, lib)[: len(all_traces)]
        coeffs = G.to_montgomery(L, FIELD)
        W = []
        for col, (_, length) in enumerate(accumulator.W):
            d = lib.alloc(max(1, length) * 32)
            owned.append(d)
            vecs = (ctypes.c_void_p * len(all_traces))(*[t.W[col][0] for t in all_traces])
            lib.check(lib.c.mira_lincomb_device(FIELD, ctypes.c_void_p(d), vecs, coeffs.ctypes.data_as(ctypes.c_void_p), len(all_traces), length))
            W.append((d, length))
        ch = [sum(l * t.challenges[i] for l, t in zip(L, all_traces)) % R_MOD for i in range(len(accumulator.challenges))]
        folded.append(Trace(ch, W))
    return folded, owned


# ---------------------------------------------------------------- compute_G (:218-303)
def compute_G(S, betas_stroke, accumulator, traces, lib=None):
    lib = lib or _lib.load()
    if not traces:
        raise ValueError("You can't fold 0 traces")              # Error::EmptyTracesNotAllowed
    rows, gates = 1 << S.k, len(S.gates)
    count = rows * gates
    if count == 0:
        return []
    points_count = 1 << (len(traces) * S.max_degree()).bit_length()     # (t d + 1).next_power_of_two()
    log_points = points_count.bit_length() - 1
    levels = (count - 1).bit_length()
    betas_stroke = list(betas_stroke)[:levels]
    points = iter_cyclic_subgroup(log_points, lib)
    folded, owned = fold_traces(points, accumulator, traces, lib)
    d = lib.alloc(points_count * count * 32)
    try:
        for p, ft in enumerate(folded):
            _evaluate_gates(S, ft, d + p * count * 32, lib)
        values = _tree_reduce(d, count, count, [betas_stroke] * points_count, lib)   # the challenge sits in the leaves
    finally:
        lib.free(d)
        for ptr in owned:
            lib.free(ptr)
    if log_points == 0:
        return _ints(values)
    return _ints(_ifft_small(values, log_points, lib))


def beta_stroke(betas, alpha, delta):
    """BetaStrokeIter (:318-337): beta_i + alpha * delta^(2^i), delta doubling as the reference does"""
    out = []
    for b in betas:
        out.append((b + alpha * delta) % R_MOD)
        delta = 2 * delta % R_MOD
    return out


# ---------------------------------------------------------------- compute_K (:339-382)
def compute_K(S, f_alpha, betas_stroke, accumulator, traces, lib=None):
    lib = lib or _lib.load()
    g_poly = compute_G(S, betas_stroke, accumulator, traces, lib)
    points_count = 1 << (len(traces) * S.max_degree()).bit_length()
    log_n = points_count.bit_length() - 1
    from mira_amd import fft as F
    g_evals = _ints(F.coset_fft(G.to_montgomery(g_poly, FIELD), lib=lib))
    k_evals = []
    for w, g_y in zip(iter_cyclic_subgroup(log_n, lib), g_evals):
        pt = ZETA * w % R_MOD
        l_y = f_alpha * eval_lagrange_poly_for_cyclic_group(pt, log_n, lib)[0] % R_MOD
        z_y = eval_vanish_polynomial(log_n, pt)
        k_evals.append((g_y - l_y) * pow(z_y, -1, R_MOD) % R_MOD)    # on the coset z_y != 0
    return _ints(F.coset_ifft(G.to_montgomery(k_evals, FIELD), lib=lib))
