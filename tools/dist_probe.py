import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch, torch.distributed as dist
from mira_amd import _lib
lib = _lib.load()
torch.cuda.set_device(0)
dist.init_process_group("nccl")
from mira_amd import commitment as cm
from mira_amd.dist import ShardedCommitmentKey
n = 1 << 22
skey = ShardedCommitmentKey.synthetic(0, n, window_bits=16)
d = cm.synth_scalars_device(0, n)
for _ in range(3): skey.commit_device(d, n)
T = {}
def tick(name, t0): T[name] = T.get(name, 0) + (time.perf_counter() - t0) * 1e3
reps = 10
for _ in range(reps):
    t0 = time.perf_counter(); lib.check(lib.c.mira_msm_set_window_bits(16)); part, c, w = skey.key.commit_partial_device(0, d, n); tick("partial", t0)
    t0 = time.perf_counter(); mine = torch.from_numpy(np.ascontiguousarray(part[: w * 16]).view(np.int64)).to("cuda"); tick("h2d", t0)
    t0 = time.perf_counter(); out = torch.empty(mine.numel(), dtype=torch.int64, device="cuda"); dist.all_gather_into_tensor(out, mine); tick("all_gather", t0)
    t0 = time.perf_counter(); g = out.cpu().numpy().view(np.uint64).reshape(1, -1); tick("d2h", t0)
    t0 = time.perf_counter(); parts = np.zeros((1, _lib.MIRA_PARTIAL_U64), dtype=np.uint64); parts[:, : w * 16] = g; r = cm.combine_partials(0, parts, c, w); tick("combine", t0)
    t0 = time.perf_counter(); skey.commit_device(d, n); tick("whole", t0)
    t0 = time.perf_counter(); skey.key.commit_device(d, n); tick("single_gpu_call", t0)
print({k: round(v / reps, 3) for k, v in T.items()})
dist.destroy_process_group()
