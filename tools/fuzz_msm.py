"""Development tool: randomised MSM parity on the GPU against the oracle -- sizes, curves, scalar
distributions, forced and planned window widths, window tables, the GLV split, batches, chunk partials.
usage: python tools/fuzz_msm.py [seconds] [seed]"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mira_amd import _lib, commitment as cm
from oracle import cref as C
from oracle import pyref as P
lib = _lib.load()
lib.tune(_lib.TUNE_TABLE_MIN_N, 1)
lib.tune(_lib.TUNE_SHARED_MIN_N, 1)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
MOD = {0: P.R_MOD, 1: P.P_MOD}                      # scalar field of each curve


def to_mont(vals, mod):
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        m = v * (1 << 256) % mod
        for k in range(4):
            out[i, k] = (m >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


def scalars(cid, n, kind):
    mod = MOD[cid]
    if kind in (0, 1):
        return C.synth_scalars(cid, n, seed=rng.getrandbits(32), kind=kind)
    if kind == 2:                                    # all equal to a small constant
        return np.tile(to_mont([rng.choice([1, 2, 3, mod - 1])], mod), (n, 1))
    if kind == 3:                                    # single-bit scalars
        return to_mont([1 << rng.randrange(254) for _ in range(n)], mod) if n <= 4096 else C.synth_scalars(cid, n, seed=rng.getrandbits(32), kind=1)
    bits = rng.choice([1, 7, 13, 16, 17, 32, 64, 128])    # values below 2^bits
    base = C.synth_scalars(cid, n, seed=rng.getrandbits(32))
    ints = [int(x) for x in base[:, 0]]
    return to_mont([v & ((1 << min(bits, 64)) - 1) for v in ints], mod) if n <= 20000 else base


t_end, cases, keys = time.time() + budget, 0, {}
while time.time() < t_end:
    cid = rng.randrange(2)
    n = rng.choice([1, 2, 3, 17, 64, 255, 1000, 4096, 5000, 1 << 14, 40000, 1 << 16, 100000, 1 << 17, 200000, 1 << 18])
    kind = rng.randrange(5)
    forced = rng.choice([0, 0, 0, 4, 5, 7, 9, 11, 12, 13, 14, 15, 16])
    mode = rng.choice(["commit", "commit", "tables", "batch", "partials", "tables_batch"])
    staged_small, chunks_small = rng.random() < 0.3, rng.random() < 0.3      # the LDS-staged sort / host point chunks at small sizes too
    lib.tune(_lib.TUNE_STAGED_MIN_N, 1 if staged_small else -1)
    lib.tune(_lib.TUNE_HOST_CHUNK_MIN_N, 2048 if chunks_small else -1)
    bases = C.synth_bases(cid, n, seed=rng.getrandbits(16)) if n <= 4096 else None
    key = cm.CommitmentKey(cid, bases) if bases is not None else cm.CommitmentKey.synthetic(cid, n, seed=rng.getrandbits(16))
    if bases is None:
        bases = key.download()
    glv = rng.random() < 0.35
    if glv:
        key.precompute(_lib.TABLE_GLV)                     # the endomorphism copy: single commits split every scalar in two
    sc = scalars(cid, n, kind)
    want = C.msm_pippenger(cid, sc, bases)
    lib.check(lib.c.mira_msm_set_window_bits(forced))
    desc = f"curve {cid} n {n} kind {kind} c {forced} {mode} staged {staged_small} chunks {chunks_small} glv {glv}"
    try:
        if mode == "commit":
            got = key.commit(sc)
            got2 = key.commit(sc)                        # second call: planned from the first one's statistics
            assert (got == got2).all(), desc
        elif mode == "tables" and n <= (1 << 16):
            lib.check(lib.c.mira_msm_set_window_bits(0))
            for width in rng.sample([8, 9, 10, 11, 12, 13, 14, 15, 16, 16, 20, 22], rng.randrange(1, 4)):   # several sets beside each other (one wide one at most)
                if width < 20 or not getattr(key, "_wide", False):
                    key.precompute(width)
                    key._wide = getattr(key, "_wide", False) or width >= 20
            got = key.commit(sc)
            assert (got == key.commit(sc)).all(), desc      # second call: the set chosen from the first one's statistics
        elif mode == "tables_batch" and n <= (1 << 16):     # 16-bit shared-bucket tables, one bucket set per commitment
            lib.check(lib.c.mira_msm_set_window_bits(0))
            key.precompute(rng.choice([8, 10, 12, 13, 15, 16]))
            m = max(1, n // 2)
            vs = [sc[:m], scalars(cid, m, rng.randrange(3)), np.zeros((m, 4), dtype=np.uint64)]
            res = key.commit_batch(vs)
            assert (res[1] == C.msm_pippenger(cid, vs[1], bases[:m])).all() and not res[2].any(), desc
            got, want = res[0], C.msm_pippenger(cid, vs[0], bases[:m])
        elif mode == "batch":
            m = max(1, n // 2)
            vs = [sc[:m], scalars(cid, m, rng.randrange(3))]
            res = key.commit_batch(vs)
            assert (res[1] == C.msm_pippenger(cid, vs[1], bases[:m])).all(), desc
            got, want = res[0], C.msm_pippenger(cid, vs[0], bases[:m])
        elif mode == "partials" and n >= 2:
            d = lib.alloc(n * 32); lib.upload(d, sc)
            cut = rng.randrange(1, n)
            pa, ca, wa = key.commit_partial_device(0, d, cut)
            pb, cb, wb = key.commit_partial_device(cut, d + cut * 32, n - cut)
            assert (ca, wa) == (cb, wb), desc
            got = cm.combine_partials(cid, np.stack([pa, pb]), ca, wa)
            lib.free(d)
        else:
            got = key.commit(sc)
        assert (got == want).all(), desc
    finally:
        lib.check(lib.c.mira_msm_set_window_bits(0))
        lib.tune(_lib.TUNE_STAGED_MIN_N, -1); lib.tune(_lib.TUNE_HOST_CHUNK_MIN_N, -1)
        key.close()
    cases += 1
    if cases % 25 == 0:
        print(f"{cases} cases ok, last: {desc}", flush=True)
print(f"fuzz: {cases} cases, all bit-exact", flush=True)
