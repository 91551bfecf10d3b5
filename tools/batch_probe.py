"""Development probe: stage timings of the batched cross-term commits of a k = 17 fold step."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm
lib = _lib.load()
for kv in filter(None, os.environ.get("TUNE", "").split(",")):     # TUNE=12=3,13=2: mira_set_tuning(knob, value)
    k_, v_ = kv.split("="); lib.tune(int(k_), int(v_))
n = 1 << 17
for cid, cnt in ((0, 6), (1, 5)):
    key = cm.CommitmentKey.synthetic(cid, n)
    d = cm.synth_scalars_device(cid, cnt * n, seed=77)
    for c in (0, 12, 13, 14, 15, 16):
        lib.check(lib.c.mira_msm_set_window_bits(c))
        for _ in range(16 if c == 0 else 2): key.commit_batch_device(d, n, cnt)      # planned: incl. the width trials of the shape
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); key.commit_batch_device(d, n, cnt); ts.append((time.perf_counter() - t0) * 1e3)
        lib.check(lib.c.mira_set_timing(1)); key.commit_batch_device(d, n, cnt); st = {a: round(b, 3) for a, b in lib.timings()}; lib.check(lib.c.mira_set_timing(0))
        pc, pw = ctypes.c_int32(), ctypes.c_int32()
        lib.check(lib.c.mira_msm_last_plan(ctypes.byref(pc), ctypes.byref(pw)))
        print(f"curve {cid} batch {cnt} x 2^17 c={c} (plan {pc.value} x {pw.value}): {sorted(ts)[2]:.3f} ms {st}", flush=True)
    lib.check(lib.c.mira_msm_set_window_bits(0))
    key.close(); lib.free(d)
