"""Development probe: wall time of one commit over the shared-bucket table set of every width 8 .. 16
(mira_msm_precompute_ex, MIRA_TUNE_TABLE_WIDTH), beside the per-window path, across the sizes a fold
step commits.  Output feeds shared_wall_us in capi.hip.  With `stages` as first argument it also prints
the stage timings of every width at 131 072 pairs."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm
lib = _lib.load()
stages = len(sys.argv) > 1 and sys.argv[1] == "stages"
cases = [(1 << 12, 0), (1 << 14, 0), (1 << 16, 0), (131072, 0), (1 << 19, 0), (1 << 21, 0), (14 << 17, 1), (7 << 17, 1)]
widths = list(range(8, 17))
print("n kind per-window " + " ".join(f"c={c}" for c in widths) + " chosen", flush=True)
for n, kind in cases:
    key = cm.CommitmentKey.synthetic(0, n); d = cm.synth_scalars_device(0, n, kind=kind)
    def med(reps=15):
        [key.commit_device(d, n) for _ in range(6)]
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); out = key.commit_device(d, n); ts.append((time.perf_counter() - t0) * 1e3)
        return sorted(ts)[reps // 2], out
    t0, want = med()
    row = [f"{t0:.3f}"]
    for c in widths:
        key.precompute(c)
        lib.check(lib.c.mira_set_tuning(_lib.TUNE_TABLE_WIDTH, c))
        t, got = med()
        tb = ctypes.c_int32()
        lib.check(lib.c.mira_msm_last_table_bits(ctypes.byref(tb)))
        assert tb.value == c, (tb.value, c)
        assert (got == want).all(), (n, kind, c)
        row.append(f"{t:.3f}")
        if stages and n == 131072 and kind == 0:
            lib.check(lib.c.mira_set_timing(1)); key.commit_device(d, n)
            print("   c", c, {a: round(b, 3) for a, b in lib.timings()}, flush=True)
            lib.check(lib.c.mira_set_timing(0))
    lib.check(lib.c.mira_set_tuning(_lib.TUNE_TABLE_WIDTH, -1))
    for _ in range(24): key.commit_device(d, n)          # the trials among the nine sets (two commits each) settle first
    t, got = med()
    tb = ctypes.c_int32()
    lib.check(lib.c.mira_msm_last_table_bits(ctypes.byref(tb)))
    assert (got == want).all()
    row.append(f"{t:.3f}(c={tb.value})")
    print(n, kind, " ".join(row), flush=True)
    key.close(); lib.free(d)
