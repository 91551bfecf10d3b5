"""Shared builders for the cross-term evaluation tests: seeded random Expression trees and the
mock evaluation data of the reference's own tests (src/polynomial/graph_evaluator.rs:405-443)."""
import random

import numpy as np

from helpers import ints_to_mont
from harness import graph_evaluator as G
from oracle import pyref as P

MODS = {0: P.P_MOD, 1: P.R_MOD}


def random_expression(rng, depth, ncols, nchal, pool=None):
    """A tree of about 2^depth nodes; `pool` collects sub-trees so some are repeated verbatim
    (the graph must share them) and special constants 0, 1, 2 show up (the graph simplifies them)."""
    pool = pool if pool is not None else []
    if depth == 0 or rng.random() < 0.08:
        k = rng.random()
        if k < 0.55:
            return G.Polynomial(rng.randrange(ncols), rng.choice([0, 0, 0, 1, -1, 2]))
        if k < 0.70 and nchal:
            return G.Challenge(rng.randrange(nchal))
        if k < 0.85:
            return G.Constant(0 if rng.random() < 0.03 else rng.choice([1, 2, 3, 5]))
        return G.Constant(rng.getrandbits(250))
    if pool and rng.random() < 0.15:
        return rng.choice(pool)
    k = rng.random()
    if k < 0.12:
        e = G.Negated(random_expression(rng, depth - 1, ncols, nchal, pool))
    elif k < 0.50:
        e = G.Sum(random_expression(rng, depth - 1, ncols, nchal, pool), random_expression(rng, depth - 1, ncols, nchal, pool))
    elif k < 0.88:
        e = G.Product(random_expression(rng, depth - 1, ncols, nchal, pool), random_expression(rng, depth - 1, ncols, nchal, pool))
    else:
        e = G.Scaled(random_expression(rng, depth - 1, ncols, nchal, pool), 0 if rng.random() < 0.03 else rng.choice([1, 2, 7, rng.getrandbits(200)]))
    pool.append(e)
    return e


def gate_like_expression(rng, nterms, depth, ncols, nchal):
    """sum of `nterms` random products-of-sums, the size of a real compressed gate (hundreds of calculations)"""
    pool, e = [], None
    for _ in range(nterms):
        t = random_expression(rng, depth, ncols, nchal, pool)
        e = t if e is None else G.Sum(e, t)
    return e


def mock_data(field, num_rows, nsel, nfix, nadv, nchal, seed):
    """-> (int getter for the Python oracle, array getter for the device / C oracle)"""
    rng = random.Random(seed)
    mod = MODS[field]
    sel = [[rng.random() < 0.5 for _ in range(num_rows)] for _ in range(nsel)]
    fix = [[rng.getrandbits(256) % mod for _ in range(num_rows)] for _ in range(nfix)]
    adv = [[rng.choice([0, 1, rng.getrandbits(256) % mod]) for _ in range(num_rows)] for _ in range(nadv)]
    chal = [rng.getrandbits(256) % mod for _ in range(nchal)]
    ints = dict(selectors=sel, fixed=fix, advice=adv, challenges=chal)
    arrs = dict(selectors=[np.array(s, dtype=np.uint8) for s in sel], fixed=[ints_to_mont(f, mod) for f in fix],
                advice=[ints_to_mont(a, mod) for a in adv], challenges=chal)
    return ints, arrs


def oracle_columns(arrs):
    """column list for oracle.cref.graph_eval in the index order of eval_column_var"""
    return list(arrs["selectors"]) + list(arrs["fixed"]) + list(arrs["advice"])
