import os, sys, time
sys.path.insert(0, "/root/repo")
from mira_amd import _lib, commitment as cm
lib = _lib.load()
for n in (64, 1024, 8192, 32768, 65536):
    key = cm.CommitmentKey.synthetic(0, n); d = cm.synth_scalars_device(0, n)
    row = []
    for c in (0, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13):
        lib.check(lib.c.mira_msm_set_window_bits(c))
        key.commit_device(d, n)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); key.commit_device(d, n); ts.append((time.perf_counter() - t0) * 1e3)
        row.append(f"{c}:{sorted(ts)[3]:.3f}")
    lib.check(lib.c.mira_msm_set_window_bits(0))
    print(n, " ".join(row), flush=True)
    key.close(); lib.free(d)
