"""Development probe: the planner's window width against every forced width, small to large MSMs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm
lib = _lib.load()
cases = [(0, 1 << 15, 0), (0, 131072, 0), (1, 131072, 0), (0, 14 << 17, 1), (1, 7 << 17, 1), (0, 1 << 18, 0), (0, 1 << 19, 0), (0, 1 << 20, 0), (0, 1 << 21, 0), (0, 1 << 22, 0)]
for cid, n, kind in cases:
    key = cm.CommitmentKey.synthetic(cid, n); d = cm.synth_scalars_device(cid, n, kind=kind)
    row = []
    for c in (0, 11, 12, 13, 14, 15, 16):
        lib.check(lib.c.mira_msm_set_window_bits(c))
        key.commit_device(d, n)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); key.commit_device(d, n); ts.append((time.perf_counter() - t0) * 1e3)
        row.append(f"c={c}: {sorted(ts)[3]:.3f}")
    lib.check(lib.c.mira_msm_set_window_bits(0))
    print(f"curve {cid} n {n} kind {kind}  " + "  ".join(row), flush=True)
    key.close(); lib.free(d)
