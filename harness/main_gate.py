"""The custom gate of `MainGate<T>` as the folding scheme sees it (reference src/main_gate.rs:543-589):

    q_m[0] s[0] s[1] + q_m[1] s[2] s[3] + sum_i q_1[i] s[i] + sum_i q_5[i] s[i]^5 + rc + q_i input + q_o out = 0

`MainGate::configure` allocates T + 2 advice columns (state[T], input, out) and 2 T + 5 fixed columns in the
order q_1[T], q_5[T], q_m[2], q_i, q_o, rc, and no selector; `Expression::from_halo2_expr`
(src/polynomial/expression.rs:303-341) maps a fixed query to index num_selectors + column and an advice query to
num_selectors + num_fixed + column.  The tree below has the reference's association order (halo2's `+` and `*`
build Sum and Product nodes exactly as written in main_gate.rs:577-588), so the compressed / homogeneous /
grouped forms derived from it are the ones `PlonkStructure::custom_gates_lookup_compressed` holds.

`circuit_gates(T, count)`: `count` MainGate<T> configurations in one constraint system (the primary circuit of an
IVC step configures two): gate g uses advice columns g (T + 2) .. and fixed columns g (2 T + 5) ..
"""
from .expression import CompressedGates, Polynomial, QueryIndexContext

MULTIPLICATION_COUNT = 2


def main_gate_expression(T, num_selectors, num_fixed, first_fixed=0, first_advice=0):
    """The gate polynomial of one MainGate<T> whose columns start at fixed column `first_fixed` and advice
    column `first_advice` of a constraint system with `num_selectors` selectors and `num_fixed` fixed columns."""
    assert T >= 2
    fixed = lambda c: Polynomial(num_selectors + first_fixed + c)
    advice = lambda c: Polynomial(num_selectors + num_fixed + first_advice + c)
    state = [advice(i) for i in range(T)]
    inp, out = advice(T), advice(T + 1)
    q_1 = [fixed(i) for i in range(T)]
    q_5 = [fixed(T + i) for i in range(T)]
    q_m = [fixed(2 * T + i) for i in range(MULTIPLICATION_COUNT)]
    q_i, q_o, rc = fixed(2 * T + 2), fixed(2 * T + 3), fixed(2 * T + 4)

    def pow_5(v):
        v2 = v * v
        return v2 * v2 * v
    init_term = q_m[0] * state[0] * state[1] + q_i * inp + rc + q_o * out
    if T >= 4:
        init_term = q_m[1] * state[2] * state[3] + init_term
    acc = init_term
    for s, q1, q5 in zip(state, q_1, q_5):
        acc = acc + (q1 * s + q5 * pow_5(s))
    return acc


def circuit_gates(T=5, count=1):
    """-> (gate expressions, QueryIndexContext) of a constraint system holding `count` MainGate<T> configurations,
    no selectors, no lookups (ConstraintSystemMetainfo::build, src/table/constraint_system_metainfo.rs:53-102)."""
    num_fixed, num_advice = count * (2 * T + 5), count * (T + 2)
    gates = [main_gate_expression(T, 0, num_fixed, g * (2 * T + 5), g * (T + 2)) for g in range(count)]
    return gates, QueryIndexContext(num_selectors=0, num_fixed=num_fixed, num_advice=num_advice, num_challenges=0, num_lookups=0)


def compressed_circuit(T=5, count=1):
    """-> (CompressedGates, ctx after CompressedGates::new): what PlonkStructure carries for that circuit.
    One gate: degree 5, 5 cross terms, challenges [u]; two gates: one compression challenge y, degree 6,
    6 cross terms, challenges [y, u] -- per instance, so evaluation sees [c1.., u1, c2.., u2]
    (src/nifs/vanilla/mod.rs:87-96)."""
    gates, ctx = circuit_gates(T, count)
    return CompressedGates.new(gates, ctx), ctx


def ivc_circuit_shape(T=5, step_circuit_main_gates=1, k=17):
    """The constraint system of one half of an IVC step, derived from the reference's `configure` functions instead of read off
    by hand (SURVEY.md 3(A) asked to confirm its table "on first real run"; no Rust toolchain exists here, so the derivation is
    made executable and asserted in tests/test_cross_term_expressions.py::test_fold_step_schedule_is_derived).

    `StepFoldingCircuit::configure` (src/ivc/step_folding_circuit.rs:275-292) calls `MainGate::configure(cs)` for the folding
    chip, then the step circuit's own `configure`, then allocates ONE instance column; it panics if the step circuit added an
    instance column.  `MainGate::<T>::configure` (src/main_gate.rs:543-552) allocates T state + input + out = T + 2 advice columns
    and q_1[T] + q_5[T] + q_m[2] + q_i + q_o + rc = 2 T + 5 fixed columns, no selector, and ONE gate.  The step circuits:
        examples/groth16/circuit.rs:125-127, examples/merkle, examples/poseidon   `MainGate::configure(cs)` again   -> 1 more
        examples/trivial (src/ivc/step_circuit.rs:161, `fn configure(_cs) {}`)                                      -> 0
    with T = 5 everywhere (src/gadgets/merkle_tree_gadget/mod.rs:1, examples/trivial/main.rs:24).  Everything else follows from
    `ConstraintSystemMetainfo::build` (src/table/constraint_system_metainfo.rs:34-118): no lookups -> one prover round of
    num_advice * 2^k witness elements and, with more than one gate, one challenge to combine them; the folding degree is the
    number of grouped terms, the cross terms one fewer (src/nifs/vanilla/mod.rs:100-104 `iter_from_first`)."""
    count = 1 + step_circuit_main_gates
    cg, ctx = compressed_circuit(T, count)
    assert ctx.num_selectors == 0 and ctx.num_lookups == 0
    return {
        "main_gates": count, "num_advice": ctx.num_advice, "num_fixed": ctx.num_fixed, "num_selectors": 0, "num_lookups": 0, "num_instance_columns": 1,
        "num_challenges": cg.compressed.num_challenges(),               # PlonkStructure::num_challenges: 0 (one gate) or 1 (the combining challenge)
        "eval_challenges": ctx.num_challenges,                          # per instance in the evaluation domain: those + u
        "round_sizes": [ctx.num_advice << k],                           # ONE witness vector: W1 = concatenate_with_padding(advice, 2^k)
        "folding_degree": len(cg.grouped), "cross_terms": len(cg.grouped) - 1,
        "witness_commit_len": ctx.num_advice << k, "cross_term_len": 1 << k,
    }


def fold_step_msm_schedule(k=17, T=5):
    """The MSM calls of one `IVC::fold_step` (src/ivc/incrementally_verifiable_computation.rs:384-562) in the groth16 / merkle /
    poseidon examples: the PRIMARY circuit (BN256) is StepFoldingCircuit over a step circuit that configures its own MainGate,
    the SECONDARY (Grumpkin) over the trivial step circuit.  Per curve: one witness commit (run_sps_protocol_0 / _1,
    src/plonk/mod.rs:680-688) and `cross_terms` commits of 2^k scalars (src/nifs/vanilla/mod.rs:123-127).
    -> {curve id: (witness length, cross terms)}, the shape bench.py and tests/test_gpu_fold_step.py replay."""
    primary, secondary = ivc_circuit_shape(T, 1, k), ivc_circuit_shape(T, 0, k)
    return {0: (primary["witness_commit_len"], primary["cross_terms"]), 1: (secondary["witness_commit_len"], secondary["cross_terms"])}
