"""Development probe: fixed-base table mode, 20-bit against 22-bit windows, 2^22 ... 2^26 pairs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm
lib = _lib.load()
for log_n in (22, 24, 26):
    n = 1 << log_n
    d = cm.synth_scalars_device(0, n)
    row = []
    ref = None
    for width in (0, 20, 22):
        key = cm.CommitmentKey.synthetic(0, n)
        t0 = time.perf_counter()
        if width:
            key.precompute(width)
        tb = time.perf_counter() - t0
        key.commit_device(d, n)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); p = key.commit_device(d, n); ts.append(time.perf_counter() - t0)
        t = sorted(ts)[1]
        ref = p if ref is None else ref
        lib.check(lib.c.mira_set_timing(1)); key.commit_device(d, n); st = {a: round(b, 2) for a, b in lib.timings()}; lib.check(lib.c.mira_set_timing(0))
        row.append(f"width {width}: {t * 1e3:.2f} ms {n / t / 1e6:.0f} M/s (tables {tb:.2f} s) same {bool((p == ref).all())} {st}")
        key.close()
    print(f"2^{log_n}: " + "\n        ".join(row), flush=True)
    lib.free(d)
