"""-m gpu: a fold step WITH a lookup argument on the device (VERDICT r03 missing item 2) -- the reference's only three-round
folding test, src/nifs/vanilla/tests.rs:256-340, at its own size and at 2^13 rows.  Body: tests/lookup_fold_case.py."""
import random

import pytest

from harness import lookup as LK
from lookup_fold_case import MOD, run_lookup_fold

pytestmark = pytest.mark.gpu


def test_three_rounds_fold_reference_circuits(gpu_lib):
    rng = random.Random(0x35)
    traces = []
    for a, b, c in ((1, 3, 2), (3, 2, 2)):
        seq = LK.get_sequence(a, b, c, 7)
        traces.append(LK.LookupTrace(5, MOD, [rng.randrange(MOD) for _ in range(3)], seq=(seq[0], seq[1], seq[2], 7)))
    assert run_lookup_fold(gpu_lib, 5, traces, seed=0x36) == 6


def test_three_rounds_fold_2p13_rows(gpu_lib):
    """the same circuit over 8 192 rows (a seeded mix of XOR rows, addition rows and empty rows), three traces folded in turn"""
    rng = random.Random(0x37)
    traces = [LK.LookupTrace(13, MOD, [rng.randrange(MOD) for _ in range(3)], seed=0x100 + i) for i in range(3)]
    assert run_lookup_fold(gpu_lib, 13, traces, seed=0x38, sample_rows=4) == 6
