"""The reference's three_rounds_test (src/nifs/vanilla/tests.rs:313-340) on the CPU: K = 5, the circuits of the test itself
(`get_sequence(1, 3, 2, 7)` and `get_sequence(3, 2, 2, 7)`), folded one after the other into the default accumulator through the
C ABI of the test-only emulation build.  The body is tests/lookup_fold_case.py; the GPU suite runs the same at K = 5 and K = 13."""
import random

from harness import lookup as LK
from lookup_fold_case import MOD, run_lookup_fold


def test_lookup_structure_shape():
    """FiboCircuitWithLookup as ConstraintSystemMetainfo::build sees it: one gate + four lookup expressions, three challenges
    (r1 compresses the vector lookup, r2 is the log-derivative point, r3 combines the five expressions) and u; folding degree
    = the number of grouped terms; round sizes 3, 3 and 2 columns (src/table/constraint_system_metainfo.rs:64-91)."""
    gates, ctx, L, T = LK.fibo_lookup_gates()
    assert len(gates) == 5 and ctx.num_challenges == 2 and ctx.num_fold_vars() == 3 + 5
    cg, ctx, _, _ = LK.compressed_fibo_lookup()
    assert cg.compressed.num_challenges() == 3 and ctx.num_challenges == 4
    # degrees count folded variables (advice, lookup variables, challenges): the vanishing polynomial L - l has s_xor * out * r1^2 (3)
    # and enters the combination under r3^3 -> 6, the highest of the five; six cross terms per fold
    assert cg.degree == 6 and len(cg.grouped) == cg.degree + 1
    seq = LK.get_sequence(1, 3, 2, 7)
    assert seq == [1, 3, 2, 2, 3, 3, 2] and max(seq) < 5                # every XOR operand is in the 5 x 5 table


def test_three_rounds_fold_on_emulation(emu_lib):
    rng = random.Random(0x33)
    traces = []
    for a, b, c in ((1, 3, 2), (3, 2, 2)):
        seq = LK.get_sequence(a, b, c, 7)
        traces.append(LK.LookupTrace(5, MOD, [rng.randrange(MOD) for _ in range(3)], seq=(seq[0], seq[1], seq[2], 7)))
    assert run_lookup_fold(emu_lib, 5, traces, seed=0x34) == 6
