"""where commit_cross_terms' graphs come from
(src/nifs/vanilla/mod.rs:100-104: S.custom_gates_lookup_compressed.grouped().iter_from_first()).

* the reference's own `Display` known-answer tests (src/polynomial/expression.rs:528-606,
  src/polynomial/grouped_poly.rs:287-461; data in tests/golden/ref_kats.json) on the product-side mirror
  (mira_amd/expression.py, grouped_poly.py) AND on the oracle's independent restatement (oracle/pyref.py);
* mirror == oracle, node for node, on the MainGate<5> circuits of an IVC step (one gate: secondary circuit,
  two gates compressed with a challenge: primary circuit);
* the identity the folding scheme rests on (src/nifs/vanilla/tests.rs:189,228 compare folded instances that
  only agree if it holds):  f(W1 + X W2, c1 + X c2) = sum_k X^k T_k  for the homogeneous f and its grouped
  terms T_k, at random X, with Python integers -- independent of every graph and kernel;
* the cross-term graphs on the emulated kernel against the C oracle and the direct evaluation.
"""
import random

import numpy as np
import pytest

from helpers import ints_to_mont, load_golden, mont_to_ints
from harness import expression as E
from harness import graph_evaluator as G
from harness import main_gate as MG
from harness.grouped_poly import GroupedPoly
from oracle import cref as C
from oracle import pyref as P

KATS = load_golden("ref_kats.json")
MODS = {0: P.P_MOD, 1: P.R_MOD}


def from_json(t):
    """golden input (nested lists, constants in hex) -> oracle tuple"""
    k = t[0]
    if k == "const":
        return ("const", int(t[1], 16))
    if k == "poly":
        return ("poly", t[1], t[2])
    if k == "chal":
        return ("chal", t[1])
    if k == "neg":
        return ("neg", from_json(t[1]))
    if k == "scaled":
        return ("scaled", from_json(t[1]), int(t[2], 16))
    return (k, from_json(t[1]), from_json(t[2]))


def to_mirror(t):
    """oracle tuple -> product-side Expression"""
    k = t[0]
    if k == "const":
        return E.Constant(t[1])
    if k == "poly":
        return E.Polynomial(t[1], t[2])
    if k == "chal":
        return E.Challenge(t[1])
    if k == "neg":
        return E.Negated(to_mirror(t[1]))
    if k == "scaled":
        return E.Scaled(to_mirror(t[1]), t[2])
    return {"sum": E.Sum, "prod": E.Product}[k](to_mirror(t[1]), to_mirror(t[2]))


# ---- the reference's Display KATs --------------------------------------------------------------------------------
def test_expression_display_kat():
    k = KATS["expression_display"]
    t = from_json(k["input"])
    assert P.expr_to_string(t) == k["string"]
    assert str(to_mirror(t)) == k["string"]
    a = E.Polynomial(0)
    e1 = a - E.Constant(1)
    assert str(e1 * e1 + a * 2) == k["string"]                       # the same through the operators, as the reference writes it


@pytest.mark.parametrize("name", ["homogeneous_simple", "homogeneous"])
def test_homogeneous_kats(name):
    k = KATS[name]
    t = from_json(k["input"])
    assert P.expr_to_string(P.expr_homogeneous(t, k["ctx"])[0]) == k["string"]
    assert str(to_mirror(t).homogeneous(E.QueryIndexContext(**k["ctx"]))[0]) == k["string"]


@pytest.mark.parametrize("name", ["grouped_simple_add", "grouped_simple_sub", "grouped_simple_mul", "grouped_mul"])
def test_grouped_poly_op_kats(name):
    k = KATS[name]
    lhs = {int(d): from_json(t) for d, t in k["lhs"].items()}
    rhs = {int(d): from_json(t) for d, t in k["rhs"].items()}
    got_o = {"add": P.grouped_add, "sub": P.grouped_sub, "mul": P.grouped_mul}[k["op"]](P.grouped_from(lhs), P.grouped_from(rhs))
    assert [f"{d};{P.expr_to_string(t)}" for d, t in enumerate(got_o) if t is not None] == k["strings"]
    a = GroupedPoly.from_terms({d: to_mirror(t) for d, t in lhs.items()})
    b = GroupedPoly.from_terms({d: to_mirror(t) for d, t in rhs.items()})
    got_m = {"add": a.__add__, "sub": a.__sub__, "mul": a.__mul__}[k["op"]](b)
    assert [f"{d};{t}" for d, t in got_m.iter_with_degree()] == k["strings"]


def test_grouped_poly_creation_kat():
    k = KATS["grouped_creation"]
    t = from_json(k["input"])
    got_o = P.grouped_new(t, k["ctx"])
    assert [f"{d};{P.expr_to_string(x)}" for d, x in enumerate(got_o) if x is not None] == k["strings"]
    got_m = GroupedPoly.new(to_mirror(t), E.QueryIndexContext(**k["ctx"]))
    assert [f"{d};{x}" for d, x in got_m.iter_with_degree()] == k["strings"]


def test_grouped_poly_scale_and_neg():
    g = GroupedPoly.from_terms({0: E.Polynomial(1), 2: E.Challenge(0)})
    assert [f"{d};{t}" for d, t in (g * 7).iter_with_degree()] == ["0;0x7 * Z_1", "2;0x7 * r_0"]     # Mul<&F>: Constant(k) * term
    assert [f"{d};{t}" for d, t in (-g).iter_with_degree()] == ["0;-Z_1", "2;-r_0"]
    assert (g * 7).iter() == [to_mirror(x) if x else None for x in P.grouped_scale(P.grouped_from({0: ("poly", 1, 0), 2: ("chal", 0)}), 7)]
    assert g.get(1) is None and g.get(5) is None and len(g) == 3 and g.iter_from_first() == g.iter()[1:]


# ---- the MainGate<5> circuits ------------------------------------------------------------------------------------
def oracle_circuit(T, count):
    nf, na = count * (2 * T + 5), count * (T + 2)
    gates = [P.main_gate_polynomial(T, 0, nf, g * (2 * T + 5), g * (T + 2)) for g in range(count)]
    return P.compressed_gates(gates, dict(num_selectors=0, num_fixed=nf, num_advice=na, num_challenges=0, num_lookups=0))


@pytest.mark.parametrize("T,count", [(5, 1), (5, 2), (2, 1), (4, 3)])
def test_main_gate_mirror_equals_oracle(T, count):
    cg, ctx = MG.compressed_circuit(T, count)
    o = oracle_circuit(T, count)
    assert cg.compressed.to_tuple() == o["compressed"] and cg.homogeneous.to_tuple() == o["homogeneous"] and cg.degree == o["degree"]
    assert [None if t is None else t.to_tuple() for t in cg.grouped.iter()] == o["grouped"]
    assert vars(ctx) == o["ctx"]
    # the shape SURVEY.md 3(A) derives: one gate -> degree 5, 5 cross terms, [u]; g gates -> the compression challenge y to the power g - 1 on top
    assert cg.degree == 5 + count - 1 and len(cg.grouped) == cg.degree + 1
    assert ctx.num_challenges == (1 if count == 1 else 2)
    assert cg.homogeneous.degree(ctx) == cg.degree
    assert str(cg.grouped.get(1)) == P.expr_to_string(o["grouped"][1])


def test_main_gate_shape_T5():
    """MainGate<5> (src/main_gate.rs:543-589): 7 advice, 15 fixed columns, no selector; the gate as its name spells it"""
    gates, ctx = MG.circuit_gates(5, 1)
    assert (ctx.num_selectors, ctx.num_fixed, ctx.num_advice) == (0, 15, 7)
    s = str(gates[0])
    assert s.startswith("Z_11 * Z_17 * Z_18 + Z_10 * Z_15 * Z_16 + Z_12 * Z_20 + Z_14 + Z_13 * Z_21 + Z_0 * Z_15 + Z_5 * Z_15 * Z_15 * Z_15 * Z_15 * Z_15")
    assert gates[0].degree(ctx) == 5 and gates[0].num_challenges() == 0


def random_instance(rng, ctx, mod, rows):
    nfix = ctx["num_selectors"] + ctx["num_fixed"]
    return dict(selectors=[], fixed=[[rng.randrange(mod) for _ in range(rows)] for _ in range(nfix)],
                W1=[[rng.randrange(mod) for _ in range(rows)] for _ in range(ctx["num_advice"])],
                W2=[[rng.choice([0, 1, rng.randrange(mod)]) for _ in range(rows)] for _ in range(ctx["num_advice"])],
                c1=[rng.randrange(mod) for _ in range(ctx["num_challenges"])], c2=[rng.randrange(mod) for _ in range(ctx["num_challenges"])])


@pytest.mark.parametrize("field,T,count", [(1, 5, 1), (0, 5, 1), (1, 5, 2), (0, 3, 2)])
def test_folding_identity_python_integers(field, T, count):
    """f(W1 + X W2, c1 + X c2) = sum_k X^k T_k, and fold_transform builds the left side symbolically"""
    mod, rows = MODS[field], 3
    rng = random.Random(1000 + 10 * T + count + field)
    o = oracle_circuit(T, count)
    ctx, f, terms = o["ctx"], o["homogeneous"], o["grouped"]
    inst = random_instance(rng, ctx, mod, rows)
    both = dict(selectors=[], fixed=inst["fixed"], advice=inst["W1"] + inst["W2"], challenges=inst["c1"] + inst["c2"])
    mm, nn = ctx["num_selectors"] + ctx["num_fixed"], ctx["num_advice"]
    folded_sym = P.expr_fold_transform(f, mm, nn)                  # challenge 2 * num_challenges is the folding variable
    assert to_mirror(f).fold_transform(mm, nn).to_tuple() == folded_sym
    for row in range(rows):
        X = rng.randrange(mod)
        folded = dict(selectors=[], fixed=inst["fixed"], challenges=[(a + X * b) % mod for a, b in zip(inst["c1"], inst["c2"])],
                      advice=[[(a + X * b) % mod for a, b in zip(w1, w2)] for w1, w2 in zip(inst["W1"], inst["W2"])])
        lhs = P.eval_expression(f, folded, row, rows, mod)
        rhs = sum(pow(X, k, mod) * P.eval_expression(t, both, row, rows, mod) for k, t in enumerate(terms)) % mod
        assert lhs == rhs
        sym = dict(both, challenges=both["challenges"] + [X])
        assert P.eval_expression(folded_sym, sym, row, rows, mod) == lhs
    # T_0 = f on instance 1 alone, T_d = f on instance 2 alone
    only1 = dict(selectors=[], fixed=inst["fixed"], advice=inst["W1"], challenges=inst["c1"])
    only2 = dict(selectors=[], fixed=inst["fixed"], advice=inst["W2"], challenges=inst["c2"])
    assert P.eval_expression(terms[0], both, 1, rows, mod) == P.eval_expression(f, only1, 1, rows, mod)
    assert P.eval_expression(terms[-1], both, 1, rows, mod) == P.eval_expression(f, only2, 1, rows, mod)


@pytest.mark.parametrize("field,count", [(1, 2), (0, 1)])
def test_cross_term_graphs_on_emulated_kernel(emu_lib, field, count):
    """the d cross-term graphs of the MainGate<5> circuits, one batched submission, against the C oracle's walk of the
    same calculation lists and the direct evaluation of the grouped expressions"""
    mod, rows = MODS[field], 24
    rng = random.Random(77 + field)
    cg, ctx = MG.compressed_circuit(5, count)
    inst = random_instance(rng, vars(ctx), mod, rows)
    both = dict(selectors=[], fixed=inst["fixed"], advice=inst["W1"] + inst["W2"], challenges=inst["c1"] + inst["c2"])
    evs = [G.GraphEvaluator.new(t, field) for t in cg.grouped.iter_from_first()]
    arrs = [ints_to_mont(c, mod) for c in both["fixed"] + both["advice"]]
    ptrs, cols = [], []
    for a in arrs:
        p = emu_lib.alloc(a.nbytes); emu_lib.upload(p, a); ptrs.append(p); cols.append((p, G.COL_FIELD))
    d = emu_lib.alloc(len(evs) * rows * 32)
    G.GraphEvaluator.evaluate_batch_device(evs, cols, both["challenges"], rows, [d + k * rows * 32 for k in range(len(evs))], lib=emu_lib)
    got = emu_lib.download(d, (len(evs), rows, 4))
    chal_m = ints_to_mont(both["challenges"], mod)
    for k, (ev, t) in enumerate(zip(evs, cg.grouped.iter_from_first())):
        want = [P.eval_expression(t.to_tuple(), both, r, rows, mod) for r in range(rows)]
        assert mont_to_ints(got[k], mod) == want
        code, consts, rots = ev.flatten()
        assert mont_to_ints(C.graph_eval(field, code, ev.num_intermediates, consts, rots, arrs, chal_m, rows), mod) == want
    for p in ptrs + [d]:
        emu_lib.free(p)


@pytest.mark.parametrize("field,T,count", [(1, 5, 2), (0, 5, 1), (1, 2, 1), (0, 3, 3)])
def test_cross_term_plan_equals_grouped_terms(emu_lib, field, T, count):
    """CrossTermPlan: the d cross terms from d + 1 evaluations of f and one linear combination each -- the same field
    elements as the reference's grouped graphs (direct Python-integer evaluation of every grouped term)."""
    mod, rows = MODS[field], 19
    rng = random.Random(900 + 10 * T + count + field)
    cg, ctx = MG.compressed_circuit(T, count)
    plan = G.CrossTermPlan.from_compressed_gates(cg, ctx, field)
    assert plan.degree == cg.degree and len(plan.evaluators) == cg.degree + 1
    inst = random_instance(rng, vars(ctx), mod, rows)
    both = dict(selectors=[], fixed=inst["fixed"], advice=inst["W1"] + inst["W2"], challenges=inst["c1"] + inst["c2"])
    arrs = [ints_to_mont(c, mod) for c in both["fixed"] + both["advice"]]
    ptrs, cols = [], []
    for a in arrs:
        p = emu_lib.alloc(a.nbytes); emu_lib.upload(p, a); ptrs.append(p); cols.append((p, G.COL_FIELD))
    d = emu_lib.alloc(plan.degree * rows * 32)
    plan.evaluate_device(cols, both["challenges"], rows, d, lib=emu_lib)
    got = emu_lib.download(d, (plan.degree, rows, 4))
    for k, t in enumerate(cg.grouped.iter_from_first()):
        assert mont_to_ints(got[k], mod) == [P.eval_expression(t.to_tuple(), both, r, rows, mod) for r in range(rows)], k
    for p in ptrs + [d]:
        emu_lib.free(p)


def test_fold_step_schedule_is_derived():
    """The k = 17 fold-step schedule -- 14 / 7 advice columns, 30 / 15 fixed, one combining challenge in the primary circuit,
    6 / 5 cross terms, 13 MSM calls -- follows from the reference's `configure` functions (harness/main_gate.py:
    ivc_circuit_shape cites them), not from a table copied by hand."""
    primary, secondary = MG.ivc_circuit_shape(5, 1, 17), MG.ivc_circuit_shape(5, 0, 17)
    assert (primary["num_advice"], primary["num_fixed"], primary["main_gates"]) == (14, 30, 2)
    assert (secondary["num_advice"], secondary["num_fixed"], secondary["main_gates"]) == (7, 15, 1)
    assert (primary["num_challenges"], secondary["num_challenges"]) == (1, 0)
    assert (primary["eval_challenges"], secondary["eval_challenges"]) == (2, 1)                 # [y, u] and [u] per instance
    assert (primary["folding_degree"], primary["cross_terms"]) == (7, 6) and (secondary["folding_degree"], secondary["cross_terms"]) == (6, 5)
    assert primary["round_sizes"] == [14 << 17] and secondary["round_sizes"] == [7 << 17]
    sched = MG.fold_step_msm_schedule(17)
    assert sched == {0: (1835008, 6), 1: (917504, 5)}
    assert sum(1 + terms for _, terms in sched.values()) == 13
    assert sum(n + terms * (1 << 17) for n, terms in sched.values()) == (14 + 7 + 6 + 5) << 17   # 4.19 M scalar-point pairs per fold step
