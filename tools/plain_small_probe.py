import os, sys, time
sys.path.insert(0, "/root/repo")
from mira_amd import _lib, commitment as cm
lib = _lib.load()
lib.tune(_lib.TUNE_GLV_AUTO_MAX_LOG, 0)
import ctypes
for n in (131072,):
    for c in (0, 8, 12, 13, 14, 16):
        lib.check(lib.c.mira_msm_set_window_bits(c))
        key = cm.CommitmentKey.synthetic(0, n); d = cm.synth_scalars_device(0, n)
        for _ in range(30): key.commit_device(d, n)
        t0 = time.perf_counter()
        for _ in range(200): key.commit_device(d, n)
        wall = (time.perf_counter() - t0) / 200 * 1e3
        cc, ww = ctypes.c_int32(), ctypes.c_int32()
        lib.check(lib.c.mira_msm_last_plan(ctypes.byref(cc), ctypes.byref(ww)))
        lib.check(lib.c.mira_set_timing(1))
        acc = {}
        for _ in range(20):
            key.commit_device(d, n)
            for name, ms in lib.timings():
                acc[name] = acc.get(name, 0) + ms / 20
        lib.check(lib.c.mira_set_timing(0))
        print("n=%d forced c=%d -> c=%d W=%d wall %.4f ms, kernels %.4f" % (n, c, cc.value, ww.value, wall, sum(acc.values())), {a: round(b, 4) for a, b in acc.items()}, flush=True)
        key.close()
