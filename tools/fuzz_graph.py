"""Development tool: randomised parity of the cross-term evaluator and the NTT on the GPU.
usage: python tools/fuzz_graph.py [seconds] [seed]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from graph_cases import MODS, gate_like_expression, mock_data, oracle_columns, random_expression
from helpers import ints_to_mont
from mira_amd import _lib, fft as F
from harness import graph_evaluator as G
from oracle import cref as C
lib = _lib.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end, graphs, ntts, biggest, jitted = time.time() + budget, 0, 0, 0, 0
while time.time() < t_end:
    field = rng.randrange(2)
    mod = MODS[field]
    n = rng.choice([1, 2, 3, 7, 64, 100, 777, 2048, 5000])
    nsel, nfix, nadv, nchal = rng.randrange(3), rng.randrange(1, 3), rng.randrange(1, 6), rng.randrange(1, 4)
    ints, arrs = mock_data(field, n, nsel, nfix, nadv, nchal, seed=rng.getrandbits(30))
    ncols = nsel + nfix + nadv
    e = gate_like_expression(rng, rng.choice([1, 2, 6, 20]), rng.choice([3, 5, 7]), ncols, nchal) if rng.random() < 0.7 else random_expression(rng, 8, ncols, nchal)
    ge = G.GraphEvaluator.new(e, field)
    code, consts, rots = ge.flatten()
    want = C.graph_eval(field, code, ge.num_intermediates, consts, rots, oracle_columns(arrs), ints_to_mont(ints["challenges"], mod), n)
    got = ge.evaluate(arrs, lib=lib)
    assert (got == want).all(), (field, n, ge.num_intermediates)
    graphs += 1; biggest = max(biggest, ge.num_intermediates)
    if graphs % 100 == 0:
        print(f"{graphs} graphs ok ({jitted} specialised)", flush=True)     # (a run that is silent for seven minutes is taken to be hung)
    if graphs % 6 == 0:                                   # the same graph through a run-time compiled kernel of its own
        kinds = [(1, G.COL_BOOL)] * nsel + [(1, G.COL_FIELD)] * (nfix + nadv)
        if G.GraphEvaluator.specialize([ge], kinds, len(ints["challenges"]), lib=lib):
            assert ge.is_specialized(len(ints["challenges"]), ncols, lib=lib)
            assert (ge.evaluate(arrs, lib=lib) == want).all(), ("specialised", field, n, ge.num_intermediates)
            jitted += 1
    ge.close()
    if graphs % 10 == 0:
        k = rng.randrange(0, 19)
        a = C.synth_scalars(0, 1 << k, seed=rng.getrandbits(30), kind=rng.randrange(2))
        op = rng.choice(["fft", "ifft", "coset_fft", "coset_ifft"])
        want = getattr(C, op)(a, k)
        got = getattr(F, op)(a, k) if op in ("fft", "ifft") else getattr(F, op)(a)
        assert (got == want).all(), (op, k)
        ntts += 1
print(f"fuzz: {graphs} graphs (largest {biggest} calculations; {jitted} of them also through a specialised kernel) and {ntts} transforms, all bit-exact", flush=True)
