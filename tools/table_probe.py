"""Development probe: fixed-base (window table) mode vs per-window mode at one size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mira_amd import _lib, commitment as cm
lib = _lib.load()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << log_n
key = cm.CommitmentKey.synthetic(0, n)
d = cm.synth_scalars_device(0, n)
ref = key.commit_device(d, n)
def run(tag):
    key.commit_device(d, n)
    lib.check(lib.c.mira_set_timing(1)); acc = {}; reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        out = key.commit_device(d, n)
        for name, ms in lib.timings(): acc[name] = acc.get(name, 0) + ms / reps
    wall = (time.perf_counter() - t0) / reps * 1e3
    lib.check(lib.c.mira_set_timing(0))
    print(tag, f"wall {wall:.3f} ms = {n / wall / 1e3:.1f} M pairs/s", {a: round(b, 3) for a, b in acc.items()}, "same" if (out == ref).all() else "DIFFERENT")
run("per-window c=16")
t0 = time.perf_counter(); key.precompute(); print(f"precompute {time.perf_counter() - t0:.3f} s")
run("fixed-base c=20")
