"""Development soak: random NTT sizes, directions, kernels and grid sizes on the GPU against the oracle (the work distribution of the
NTT kernels -- counters, ranges without a home workgroup, the static stride -- under real concurrency).

usage: python tools/fuzz_ntt.py [seconds] [seed]
"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mira_amd import _lib, fft as F
from oracle import cref as C

lib = _lib.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20261005
rng = random.Random(seed)
t_end = time.time() + budget
cases = fails = 0
by_k = {}
while time.time() < t_end:
    k = rng.choice([rng.randint(0, 12), rng.randint(13, 19), rng.randint(20, 22), rng.randint(20, 24), 25])
    if k >= 24 and rng.random() < 0.6:
        k = rng.randint(16, 23)
    wave = rng.choice([-1, -1, 1, 0])
    grid = rng.choice([-1, -1, 1, 2, 3, 5, 7, 8, 13, 37, 64, 100, 255, 256, 700])
    max_line = rng.choice([-1, -1, -1, 8, 7, 6]) if wave != 0 else -1
    if max_line > 0 and k > 3 * max_line:
        max_line = -1
    kind = rng.choice(["fft", "ifft", "coset_fft", "coset_ifft", "best_fft"])
    a = C.synth_scalars(0, 1 << k, seed=rng.randrange(1 << 30))
    lib.tune(_lib.TUNE_NTT_WAVE, wave); lib.tune(_lib.TUNE_NTT_GRID, grid); lib.tune(_lib.TUNE_NTT_MAX_LOG_LINE, max_line)
    try:
        if kind == "fft": got, want = F.fft(a, k), C.fft(a, k)
        elif kind == "ifft": got, want = F.ifft(a, k), C.ifft(a, k)
        elif kind == "coset_fft": got, want = F.coset_fft(a), C.coset_fft(a, k)
        elif kind == "coset_ifft": got, want = F.coset_ifft(a), C.coset_ifft(a, k)
        else:
            w = C.get_omega_or_inv(k, bool(rng.getrandbits(1)))
            got, want = F.best_fft(a, w, k), C.best_fft(a, w, k)
        ok = bool((got == want).all())
    except _lib.MiraError as e:
        ok = "three passes" in str(e) or "exceeds" in str(e)      # a forced short line cannot reach the size: refused, not wrong
        if not ok:
            print("ERROR", k, kind, wave, grid, max_line, e, flush=True)
    cases += 1
    by_k[k] = by_k.get(k, 0) + 1
    if not ok:
        fails += 1
        print("MISMATCH k=%d %s wave=%d grid=%d max_line=%d" % (k, kind, wave, grid, max_line), flush=True)
for knob in (_lib.TUNE_NTT_WAVE, _lib.TUNE_NTT_GRID, _lib.TUNE_NTT_MAX_LOG_LINE):
    lib.tune(knob, -1)
print("ntt fuzz: %d cases, %d failures, seed %d, sizes %s" % (cases, fails, seed, dict(sorted(by_k.items()))))
sys.exit(1 if fails else 0)
