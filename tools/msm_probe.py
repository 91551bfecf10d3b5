"""Development probe: per-stage timings of MSMs of the fold-step sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib
if os.environ.get("MIRA_PROBE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIRA_PROBE_LIB"])   # development: a variant build
from mira_amd import commitment as cm
lib = _lib.load()
cases = [(0, 131072, 0), (1, 131072, 0), (0, 14 << 17, 1), (1, 7 << 17, 1), (0, 1 << 20, 0), (0, 1 << 22, 0)]
if len(sys.argv) > 1:
    cases = [(0, 1 << int(sys.argv[1]), 0)]
for cid, n, kind in cases:
    key = cm.CommitmentKey.synthetic(cid, n)
    d = cm.synth_scalars_device(cid, n, kind=kind)
    key.commit_device(d, n)
    lib.check(lib.c.mira_set_timing(1))
    acc = {}; reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        key.commit_device(d, n)
        for name, ms in lib.timings():
            acc[name] = acc.get(name, 0) + ms / reps
    wall = (time.perf_counter() - t0) / reps * 1e3
    lib.check(lib.c.mira_set_timing(0))
    t0 = time.perf_counter()
    for _ in range(reps):
        key.commit_device(d, n)
    wall_nt = (time.perf_counter() - t0) / reps * 1e3
    print(f"curve {cid} n {n} kind {kind}: wall {wall:.3f} ms (no timers {wall_nt:.3f}) sum_stages {sum(acc.values()):.3f}", {a: round(b, 3) for a, b in acc.items()})
    key.close(); lib.free(d)
