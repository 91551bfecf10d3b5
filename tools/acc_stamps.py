"""Development probe: when the waves of k_accumulate finish, by the wave slot they hold on their SIMD (a -DMSM_PROBE_STAMPS build).

usage: MIRA_PROBE_LIB=tools/_variants/msm_stamps.so python tools/acc_stamps.py [log_n] [window_bits]
"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ["MIRA_PROBE_LIB"])
from mira_amd import commitment as cm
lib = _lib.load()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
c = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = 1 << k
lib.check(lib.c.mira_msm_set_window_bits(c))
lib.tune(_lib.TUNE_GLV_AUTO_MAX_LOG, 0)
key = cm.CommitmentKey.synthetic(0, n); d = cm.synth_scalars_device(0, n)
for _ in range(20):
    key.commit_device(d, n)
buf = np.zeros(4096 * 3, dtype=np.uint64)
assert lib.c.mira_debug_acc_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size)) == 0
b = buf.reshape(4096, 3).astype(np.int64)
b = b[b[:, 1] > 0]
t0 = b[:, 1].min()
print("waves:", len(b), " kernel span %.1f us; starts within %.1f us" % ((b[:, 2].max() - t0) / 100.0, (b[:, 1].max() - t0) / 100.0))
slot = b[:, 0] & 15
for s_ in sorted(set(slot)):
    e = (b[slot == s_, 2] - t0) / 100.0
    st = (b[slot == s_, 1] - t0) / 100.0
    print("  wave slot %d: %4d waves, start %.1f us, finished at mean %.1f (min %.1f, max %.1f) us" % (s_, len(e), st.mean(), e.mean(), e.min(), e.max()))
