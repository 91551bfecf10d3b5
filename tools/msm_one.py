"""Development probe: repeat one MSM size (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm
lib = _lib.load()
cid, n, kind, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
key = cm.CommitmentKey.synthetic(cid, n)
d = cm.synth_scalars_device(cid, n, kind=kind)
for _ in range(reps):
    key.commit_device(d, n)
