"""Child process of tests/test_gpu_dist.py: one rank, backend nccl (= RCCL), started fresh so that the process group
exists before anything touches the GPU.  Commits through ShardedCommitmentKey (whole key, a prefix, TooLongInput)
and prints one JSON line with the points."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n, cid, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)                                              # RCCL's banner goes to stderr
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import ctypes
    from mira_amd import _lib, commitment as cm
    from mira_amd.dist import ShardedCommitmentKey
    lib = _lib.load()
    lib.check(lib.c.mira_init(0))
    key = ShardedCommitmentKey.synthetic(cid, n)
    d = cm.synth_scalars_device(cid, n)
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "points": {}}
    for m in (n, n - 1234, 7):
        out["points"][str(m)] = [int(v) for v in key.commit_device(d, m)]
        c, w = ctypes.c_int32(), ctypes.c_int32()
        lib.check(lib.c.mira_msm_last_plan(ctypes.byref(c), ctypes.byref(w)))
        out.setdefault("widths", {})[str(m)] = [key._agreed_window_bits(m), c.value]
    try:
        key.commit_device(d, n + 1)
        out["too_long"] = None
    except cm.TooLongInput as e:
        out["too_long"] = [e.input_len, e.limit]
    # the single-GPU per-window planner on a fresh key (no statistics yet) picks the width the sharded key derives
    # (the endomorphism split, which a lone GPU takes by default at this size, is not a sharded partial's business: off)
    lib.tune(_lib.TUNE_GLV_AUTO_MAX_LOG, 0)
    plain = cm.CommitmentKey.synthetic(cid, n, seed=0x1234)
    plain.commit_device(d, n)
    c, w = ctypes.c_int32(), ctypes.c_int32()
    lib.check(lib.c.mira_msm_last_plan(ctypes.byref(c), ctypes.byref(w)))
    out["single_gpu_planner_width"] = c.value
    dist.barrier()
    dist.destroy_process_group()
    real_stdout.write(json.dumps(out) + "\n")
    real_stdout.flush()


if __name__ == "__main__":
    main()
