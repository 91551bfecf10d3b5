"""The GLV split (mira_msm_precompute_ex(handle, MIRA_TABLE_GLV)) against the plain per-window path on one key, same box:
wall time of one commit per size, planner's width and every forced width, the points compared.
usage: python tools/glv_probe.py [log_n ...]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm
lib = _lib.load()
logs = [int(a) for a in sys.argv[1:] if not a.startswith("-")] or [12, 15, 17, 19, 22]


def med(f, reps):
    f(); f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


def plan():
    c, w = ctypes.c_int32(), ctypes.c_int32()
    lib.check(lib.c.mira_msm_last_plan(ctypes.byref(c), ctypes.byref(w)))
    return c.value, w.value


def calibrate():
    """rows of glv_wall_us (capi.hip): wall time in microseconds of one GLV commit of 2^k uniform pairs under every width"""
    sizes = [10, 12, 14, 15, 16, 17, 18, 19, 20, 21, 22]
    print("static const int glv_log_n[%d] = {%s};" % (len(sizes), ", ".join(map(str, sizes))))
    for k in sizes:
        n = 1 << k
        key = cm.CommitmentKey.synthetic(0, n); d = cm.synth_scalars_device(0, n)
        key.precompute(_lib.TABLE_GLV)
        row = [0] * 17
        for c in range(5, 17):
            if k >= 20 and c < 9:
                row[c] = 0
                continue
            lib.check(lib.c.mira_msm_set_window_bits(c))
            row[c] = round(med(lambda: key.commit_device(d, n), 15 if k <= 19 else 7) * 1e3)
        lib.check(lib.c.mira_msm_set_window_bits(0))
        best = min((v, c) for c, v in enumerate(row) if v)
        print("    {%s},   // 2^%d: best c = %d" % (", ".join("%5d" % v for v in row), k, best[1]), flush=True)
        key.close(); lib.free(d)


def compare():
    for cid in (0, 1):
        for k in logs:
            n = 1 << k
            key = cm.CommitmentKey.synthetic(cid, n); d = cm.synth_scalars_device(cid, n)
            key.precompute(_lib.TABLE_GLV)
            reps = 21 if k <= 19 else 7
            lib.tune(_lib.TUNE_GLV, 0)
            want = key.commit_device(d, n); t_plain = med(lambda: key.commit_device(d, n), reps); p_plain = plan()
            lib.tune(_lib.TUNE_GLV, 1)
            got = key.commit_device(d, n); t_glv = med(lambda: key.commit_device(d, n), reps); p_glv = plan()
            row = []
            for c in range(max(6, p_glv[0] - 2), min(16, p_glv[0] + 3) + 1):
                lib.check(lib.c.mira_msm_set_window_bits(c))
                row.append("%d: %.3f" % (c, med(lambda: key.commit_device(d, n), max(5, reps // 3))))
            lib.check(lib.c.mira_msm_set_window_bits(0))
            print("curve %d 2^%d  plain %.3f ms %s | glv %.3f ms %s | same point %s | glv forced  %s" % (cid, k, t_plain, p_plain, t_glv, p_glv, bool((got == want).all()), "  ".join(row)), flush=True)
            key.close(); lib.free(d)



if __name__ == "__main__":
    if "--calibrate" in sys.argv:
        calibrate()
    else:
        compare()
