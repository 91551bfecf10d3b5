"""Development probe: MSM at 2^24 and 2^26 on one GPU -- timing, size-independent checks (window
widths agree, chunk partials combine to the whole), the fixed-base table mode, and oracle parity
at both sizes (the 2^26 oracle run takes about 75 s on the box's 256 host threads)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mira_amd import _lib, commitment as cm
lib = _lib.load()
for log_n in (24, 26):
    n = 1 << log_n
    t0 = time.perf_counter(); key = cm.CommitmentKey.synthetic(0, n); d = cm.synth_scalars_device(0, n)
    print(f"2^{log_n}: inputs {time.perf_counter() - t0:.2f} s", flush=True)
    key.commit_device(d, n)
    t0 = time.perf_counter(); whole = key.commit_device(d, n); dt = time.perf_counter() - t0
    print(f"2^{log_n}: {dt * 1e3:.2f} ms = {n / dt / 1e6:.1f} M pairs/s", flush=True)
    lib.check(lib.c.mira_msm_set_window_bits(13))
    alt = key.commit_device(d, n)
    lib.check(lib.c.mira_msm_set_window_bits(16))
    parts = []
    for g in range(8):
        part, c, w = key.commit_partial_device(g * (n // 8), d + g * (n // 8) * 32, n // 8)
        parts.append(part)
    comb = cm.combine_partials(0, np.stack(parts), c, w)
    lib.check(lib.c.mira_msm_set_window_bits(0))
    print(f"2^{log_n}: c=13 equals c=16: {(alt == whole).all()}  8 chunk partials combine to whole: {(comb == whole).all()}", flush=True)
    from oracle import cref as C
    t0 = time.perf_counter(); want = C.commit(0, key.bases(), lib.download(d, (n, 4)))
    print(f"2^{log_n}: oracle {time.perf_counter() - t0:.1f} s bit-exact: {(want == whole).all()}", flush=True)
    t0 = time.perf_counter(); key.precompute(); tb = time.perf_counter() - t0
    key.commit_device(d, n)
    t0 = time.perf_counter(); tab = key.commit_device(d, n); dt = time.perf_counter() - t0
    print(f"2^{log_n}: fixed-base tables built in {tb:.2f} s; {dt * 1e3:.2f} ms = {n / dt / 1e6:.1f} M pairs/s; same point: {(tab == whole).all()}", flush=True)
    key.close(); lib.free(d)
