"""Development probe: one commit of 2^17 .. 2^19 pairs under minimum segment lengths 6 .. 16 (MIRA_TUNE_MIN_SEGMENT), per-window path and table sets."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm
lib = _lib.load()
for n in (1 << 15, 1 << 17, 1 << 18, 1 << 19):
    key = cm.CommitmentKey.synthetic(0, n); d = cm.synth_scalars_device(0, n)
    want = key.commit_device(d, n)
    for tables in (0, 11, 13, 16):
        if tables:
            key.precompute(tables); lib.tune(_lib.TUNE_TABLE_WIDTH, tables)
        row = []
        for L in (16, 12, 10, 8, 6):
            lib.tune(_lib.TUNE_MIN_SEGMENT, L)
            key.commit_device(d, n); key.commit_device(d, n)
            ts = []
            for _ in range(9):
                t0 = time.perf_counter(); got = key.commit_device(d, n); ts.append((time.perf_counter() - t0) * 1e3)
            assert (got == want).all()
            lib.check(lib.c.mira_set_timing(1)); key.commit_device(d, n); st = dict(lib.timings()); lib.check(lib.c.mira_set_timing(0))
            row.append(f"L={L}: {sorted(ts)[4]:.3f} (acc {st.get('accumulate', 0):.3f} fix {st.get('fixup', 0):.3f})")
        print(f"n={n} tables={tables}: " + "  ".join(row), flush=True)
    lib.tune(_lib.TUNE_TABLE_WIDTH, -1); lib.tune(_lib.TUNE_MIN_SEGMENT, -1)
    key.close(); lib.free(d)
