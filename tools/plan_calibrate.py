"""Development probe: wall time of one commit for every window width, across the sizes the planner
(plan_cost_us in capi.hip) has to decide for.  Output feeds its base_us table and candidate list."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm
lib = _lib.load()
if os.environ.get("PLAIN"):                    # the per-window path alone (the rows of plan_wall_us): no endomorphism copy is built or used
    lib.tune(_lib.TUNE_GLV_AUTO_MAX_LOG, 0)
    lib.tune(_lib.TUNE_GLV, 0)
cases = [(64, 0), (1024, 0), (8192, 0), (32768, 0), (65536, 0), (131072, 0), (1 << 18, 0), (1 << 19, 0), (1 << 20, 0), (1 << 21, 0), (14 << 17, 1), (7 << 17, 1), (131072, 1)]
widths = [0] + list(range(4, 17))
print("n kind " + " ".join(f"c={c}" for c in widths), flush=True)
for n, kind in cases:
    key = cm.CommitmentKey.synthetic(0, n); d = cm.synth_scalars_device(0, n, kind=kind)
    row = []
    for c in widths:
        lib.check(lib.c.mira_msm_set_window_bits(c))
        for _ in range(16 if c == 0 else 6): key.commit_device(d, n)      # planned: incl. the width trials of the shape
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); key.commit_device(d, n); ts.append((time.perf_counter() - t0) * 1e3)
        row.append(f"{sorted(ts)[7]:.3f}")
        if c == 0:
            import ctypes
            pc, pw = ctypes.c_int32(), ctypes.c_int32()
            lib.check(lib.c.mira_msm_last_plan(ctypes.byref(pc), ctypes.byref(pw)))
            row[-1] += f"(c={pc.value})"
    lib.check(lib.c.mira_msm_set_window_bits(0))
    print(n, kind, " ".join(row), flush=True)
    key.close(); lib.free(d)
