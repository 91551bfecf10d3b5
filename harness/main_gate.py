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
