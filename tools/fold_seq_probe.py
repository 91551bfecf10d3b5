"""Development probe: the 13 commits of the k = 17 fold step one call at a time, per-call wall times."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm
lib = _lib.load()
k = 17; n = 1 << k
plan = {cm.CURVE_BN256: (14 << k, 6), cm.CURVE_GRUMPKIN: (7 << k, 5)}
keys, wit, cross = {}, {}, {}
for c, (nw, cnt) in plan.items():
    keys[c] = cm.CommitmentKey.synthetic(c, nw, seed=0x464F4C44 + c)
    wit[c] = cm.synth_scalars_device(c, nw, seed=0x1000 + c, kind=1)
    cross[c] = lib.alloc(cnt * n * 32)
    for i in range(cnt):
        lib.check(lib.c.mira_synth_scalars_device(c, n, 0, 0x2000 + 16 * c + i, 0, ctypes.c_void_p(cross[c] + i * n * 32)))
def run(log=False):
    ts = []
    for c, (nw, cnt) in plan.items():
        t0 = time.perf_counter(); keys[c].commit_device(wit[c], nw); ts.append((time.perf_counter() - t0) * 1e3)
        for i in range(cnt):
            t0 = time.perf_counter(); keys[c].commit_device(cross[c] + i * n * 32, n); ts.append((time.perf_counter() - t0) * 1e3)
    return ts
run(); run()
for _ in range(3):
    ts = run()
    print("total %.3f ms:" % sum(ts), " ".join(f"{t:.3f}" for t in ts), flush=True)
lib.check(lib.c.mira_set_timing(1))
keys[0].commit_device(cross[0], n)
print("stages 131072:", {a: round(b, 3) for a, b in lib.timings()})
