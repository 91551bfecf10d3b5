#!/usr/bin/env python3
"""Headline benchmark: BN256 G1 MSM throughput through CommitmentKey::commit on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one MSM (the whole commit hot path: digit recoding, bucket sort, bucket accumulation,
bucket reduction, window combine, to_affine) over one batch of synthetic scalars, with scalars and
bases already resident in HBM.  N = 1: BASELINE.json configs[1] (2^22 pairs, 16-bit windows).
N > 1: one process per GPU; the (scalar, base) pairs are sharded by point chunk, 2^22 pairs per
GPU (weak scaling); each rank reduces its chunk to W window sums, one RCCL all-gather exchanges
them (W * 128 bytes per rank) and every rank combines -- EC addition is not an RCCL reduction
operator, so all-gather + local add is the exchange.

Prints ONE JSON line on rank 0.  Extra keys: `roofline` (dominant kernel k_accumulate, algorithmic
bytes = 96 B per pair), `cpu_baseline` (the C restatement of the reference's
best_multiexp on the host cores, bounded sample), `stages_ms`, and at N = 1 `extras` with the
2^24 NTT and the k = 17 fold-step MSM schedule beside their CPU timings.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MSM_BYTES_PER_PAIR = 96        # 64 B affine base + 32 B scalar, each read once
NTT_BYTES_PER_ELEM = 64        # 32 B read + 32 B written


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def host_cpu_info():
    """What the host actually gives this process: logical CPUs, the affinity mask and the cgroup CPU
    quota.  The CPU baseline runs on `effective` threads -- what rayon's default pool would use
    (std::thread::available_parallelism honours both the mask and the quota)."""
    logical = os.cpu_count() or 1
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else logical
    quota = None
    try:                                                   # cgroup v2
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()[:2]
            if q != "max":
                quota = int(q) / int(period)
    except Exception:
        try:                                               # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if q > 0:
                quota = q / period
        except Exception:
            pass
    effective = max(1, min(affinity, int(quota + 0.999) if quota else affinity))
    return {"logical_cpus": logical, "sched_affinity": affinity, "cgroup_cpu_quota": quota, "effective": effective}


def launch_ranks(n_gpus, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks through torch.distributed.run
    as a CHILD process (this process has not touched the GPU and never will) and pass its output
    and exit code on."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    log(f"[bench] --gpus {n_gpus} without WORLD_SIZE: launching {' '.join(cmd)}")
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log-n", type=int, default=0, help="weak scaling: log2 of pairs per GPU (default 22 at --gpus 1)")
    ap.add_argument("--total-log-n", type=int, default=0,
                    help="strong scaling (BASELINE configs[4]): one MSM of 2^N pairs cut into --gpus point chunks (default 26 at --gpus > 1)")
    ap.add_argument("--window-bits", type=int, default=16)
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-key-load", action="store_true", help="skip the 8 GiB commitment-key file leg of the extras")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="REHEARSAL ONLY: all ranks share GPU 0 and exchange over gloo -- the whole multi-rank path with the product library on a one-GPU box; never a measurement")
    ap.add_argument("--emulate", action="store_true",
                    help="REHEARSAL ONLY: gloo + the test-only host emulation of the kernels, to exercise the multi-rank launch on a box without GPUs; never a measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    # Libraries (RCCL prints its version banner) write to stdout; the contract is ONE JSON line
    # there.  Keep the real stdout aside and point fd 1 at stderr for the duration of the run.
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("MIRA_BENCH_FORCE_DIST"):      # FORCE_DIST: exercise the RCCL path with one rank
        import torch
        import torch.distributed as dist
        if args.emulate or args.rehearse_one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    ranks_seen = dist.get_world_size() if dist is not None else 1
    if ranks_seen != args.gpus:
        log(f"[bench] --gpus {args.gpus} but {ranks_seen} rank(s) are running")
        sys.exit(3)
    n_gpus = ranks_seen

    from mira_amd import _lib
    from mira_amd import commitment as cm
    if args.emulate:
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "mira_amd", "csrc"), "emu"])
        lib = _lib.MiraLib(os.path.join(ROOT, "tests", "emu", "libmira_emu.so"))
        _lib._lib = lib                                            # the host mirror's default library for this rehearsal
    else:
        lib = _lib.load()
    lib.check(lib.c.mira_init(0 if (args.emulate or args.rehearse_one_gpu) else local_rank))
    lib.check(lib.c.mira_msm_set_window_bits(args.window_bits))

    cid = cm.CURVE_BN256
    # N = 1: BASELINE configs[1] (2^22).  N > 1: BASELINE configs[4], ONE 2^26 MSM cut into N point
    # chunks (strong scaling) unless --log-n asks for the weak-scaling variant.
    strong = args.total_log_n > 0 or (n_gpus > 1 and args.log_n == 0)
    total_log_n = args.total_log_n or 26
    log_n = args.log_n or 22
    if strong:
        total = 1 << total_log_n
        from mira_amd.dist import chunk_bounds
        lo, hi = chunk_bounds(total, n_gpus, rank)
        n, index0 = hi - lo, lo
    else:
        n = 1 << log_n
        total, index0 = n * n_gpus, rank * n
    t0 = time.time()
    if dist is None:
        key = cm.CommitmentKey.synthetic(cid, n)
    else:
        from mira_amd.dist import ShardedCommitmentKey
        skey = ShardedCommitmentKey.synthetic(cid, total, window_bits=args.window_bits)        # rank r holds bases [lo_r, hi_r)
        key = skey.key
    d_scalars = cm.synth_scalars_device(cid, n, index0=index0)
    log(f"[rank {rank}] inputs generated on GPU in {time.time() - t0:.1f}s (n = {n} on this GPU)")

    def sync_all():
        lib.check(lib.c.mira_dev_sync())
        if dist is not None:
            if not (args.emulate or args.rehearse_one_gpu):
                import torch
                torch.cuda.synchronize()
            dist.barrier()

    def step():
        if dist is None:
            return key.commit_device(d_scalars, n)
        return skey.commit_device(d_scalars, total)               # partial MSM + RCCL all-gather + combine

    for _ in range(args.warmup):
        result = step()
    lib.check(lib.c.mira_set_timing(1))
    stage_acc = {}
    sync_all()
    t0 = time.perf_counter()
    step_walls, rank_diag = [], []
    for _ in range(args.steps):
        ts = time.perf_counter()
        result = step()
        step_walls.append((time.perf_counter() - ts) * 1e3)     # a commit is synchronous: this is the step's own wall time
        if dist is not None:
            rank_diag.append(dict(skey.last))
        for name, ms in lib.timings():
            stage_acc[name] = stage_acc.get(name, 0.0) + ms
    sync_all()
    elapsed = time.perf_counter() - t0
    lib.check(lib.c.mira_set_timing(0))
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if (args.emulate or args.rehearse_one_gpu) else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    value = total / (elapsed / args.steps) / 1e6
    stages = {k: v / args.steps for k, v in stage_acc.items()}
    t_acc = stages.get("accumulate", 0.0)
    achieved = (MSM_BYTES_PER_PAIR * n / (t_acc * 1e-3) / 1e9) if t_acc > 0 else None
    used_c, used_w = ctypes.c_int32(), ctypes.c_int32()
    lib.check(lib.c.mira_msm_last_plan(ctypes.byref(used_c), ctypes.byref(used_w)))
    window_bits = args.window_bits or used_c.value or 16       # --window-bits 0: whatever the planner (or the sharded key) chose
    adds_per_pair = (256 + window_bits - 1) // window_bits

    out = {
        "metric": "bn256_g1_msm_throughput", "value": round(value, 3), "unit": "M scalar-point pairs/s",
        "n_gpus": n_gpus, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "step_ms": {"median": round(sorted(step_walls)[len(step_walls) // 2], 4), "min": round(min(step_walls), 4), "max": round(max(step_walls), 4)},   # this rank's steps; `value` is from the whole timed region
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u256 mod p (9 x 29-bit limbs in u32/u64, Montgomery)",
        "data": ("synthetic -- CPU EMULATION REHEARSAL of the launch path, not a measurement" if args.emulate else
                 "synthetic -- REHEARSAL: every rank on GPU 0, gloo exchange; not a measurement" if args.rehearse_one_gpu else "synthetic"),
        "config": {"workload": (f"BN256 G1 MSM 2^{total_log_n} pairs point-sharded over {n_gpus} GPU(s)" if strong else f"BN256 G1 MSM 2^{log_n} pairs per GPU")
                               + f" via CommitmentKey::commit, {window_bits}-bit signed windows",
                   "pairs_per_gpu": n, "total_pairs": total, "window_bits": window_bits,
                   "parallelism": f"point-chunk x{n_gpus}" if n_gpus > 1 else "single GPU",
                   "inputs": "uniform Fr scalars, bases k_i*G, resident in HBM"},
        "roofline": {"bound": "hbm", "kernel": "k_accumulate", "achieved": None if achieved is None else round(achieved, 2),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 5),
                     "traffic": load_traffic((n.bit_length() - 1) if n & (n - 1) == 0 else -1),
                     "traffic_source": "profiles/pmc_traffic.json: a separate rocprofv3 --pmc pass of this command on the builder's box, not this run",
                     "algorithmic_bytes_per_launch": MSM_BYTES_PER_PAIR * n,
                     "avg_launch_ms": round(t_acc, 4),
                     "alu": None if t_acc <= 0 else {
                         "achieved_G_modmul_per_s": round(10 * adds_per_pair * n / (t_acc * 1e-3) / 1e9, 1), "microbench_peak_G_modmul_per_s": 171.0,
                         "frac": round(10 * adds_per_pair * n / (t_acc * 1e-3) / 1e9 / 171.0, 3),
                         "peak_issue_derived": {"G_modmul_per_s_mads_only": round(PEAK_ISSUE_MAD_ONLY, 1), "G_modmul_per_s_all_valu": round(PEAK_ISSUE_ALL, 1),
                                                "frac_of_all_valu": round(10 * adds_per_pair * n / (t_acc * 1e-3) / 1e9 / PEAK_ISSUE_ALL, 3),
                                                "how": "1024 SIMDs x 64 lanes x 2.4 GHz / (162 v_mad_u64_u32 x 4.76 cyc [+ 85 other VALU x 2.08 cyc]) per f29_mul"},
                         # the instruction mix of the shipped mixed addition itself (hipcc -S: 1 467 v_mad_u64_u32, 82 v_mul_lo_u32, 152 v_lshl_add_u64,
                         # 144 v_lshrrev_b64, ~530 simple VALU on the common path) at the measured issue costs: 9 855 cycles per wave-addition
                         "mix_issue_bound": {"cycles_per_wave_addition": MADD_ISSUE_CYCLES, "G_additions_per_s_at_max_clock": round(PEAK_MADD, 2),
                                             "achieved_G_additions_per_s": round(adds_per_pair * n / (t_acc * 1e-3) / 1e9, 2),
                                             "frac_at_max_clock": round(adds_per_pair * n / (t_acc * 1e-3) / 1e9 / PEAK_MADD, 3),
                                             "note": "the chip holds ~2.06 GHz under this load (GRBM_GUI_ACTIVE / 8 / duration, profiles/r03_c_msm2p22_pmc_valu.txt): "
                                                     "0.99 of the bound at the sustained clock (profiles/r03_c_accumulate_variants.txt)"},
                         "note": f"{adds_per_pair} mixed XYZZ additions x 10 field multiplications per pair; peak = f29_mul microbenchmark "
                                 "(profiles/r01_b_microbench_f29.txt)"},
                     "note": "integer-ALU bound (about 160 modular multiplications per pair); see DESIGN.md section 4"},
        "stages_ms": {k: round(v, 4) for k, v in stages.items()},
        # the commitment of the last timed step, x || y as 8 little-endian u64 limbs (Montgomery form): lets a test
        # compare a multi-rank run with the oracle's point for the same synthetic inputs
        "result_affine_u64": [hex(int(v)) for v in np.asarray(result, dtype=np.uint64).reshape(-1)],
    }

    if dist is not None:
        # Self-diagnosis of an N-rank run: what every rank saw, medians over the timed steps -- its pairs, the agreed window shape,
        # its partial MSM, the exchange (which includes waiting for the slowest rank) and the host combine.  A rank that is slow,
        # holds the wrong chunk or planned another width shows here, in the one line the driver keeps.
        def med(key_):
            vals = sorted(d[key_] for d in rank_diag)
            return vals[len(vals) // 2] if vals else None
        mine = {"rank": rank, "local_rank": local_rank, "pairs": n, "window_bits": med("window_bits"), "num_windows": med("num_windows"),
                "partial_ms": med("partial_ms"), "exchange_us": med("exchange_us"), "combine_ms": med("combine_ms"),
                "step_ms_median": round(sorted(step_walls)[len(step_walls) // 2], 4), "step_ms_max": round(max(step_walls), 4)}
        gathered = [None] * n_gpus
        dist.all_gather_object(gathered, mine)
        out["ranks"] = gathered
        out["agreed_window"] = {"window_bits": mine["window_bits"], "num_windows": mine["num_windows"],
                                "same_on_all_ranks": len({(g_["window_bits"], g_["num_windows"]) for g_ in gathered}) == 1}
    if strong and n_gpus > 1 and dist is not None and not args.emulate and not args.no_extras:
        out["extras"] = multi_gpu_extras(lib, cm, dist, rank, n_gpus, total_log_n, args)
    if rank == 0 and n_gpus == 1 and not args.no_cpu:
        out["cpu_baseline"], parity = cpu_baseline_msm(lib, cm, key, d_scalars, n, window_bits)
        out["parity"] = parity
    if rank == 0 and n_gpus == 1 and not args.no_extras:
        out["extras"] = extras(lib, cm, not args.no_cpu, args.no_key_load)
        # BASELINE.json's metric names two figures: the MSM throughput above and "NIFS fold-step ms (k=17)" (configs[3]).  The
        # second one is measured in the same run (extras.nifs_fold_step_k17 has the spans and the variants) and repeated here, at
        # the top level of the line, beside the CPU port's time for the same chain on this box's host cores.
        fs = out["extras"].get("nifs_fold_step_k17", {})
        if "ms" in fs:
            out["secondary_metric"] = {"metric": "nifs_fold_step_ms_k17", "value": fs["ms"], "unit": "ms", "higher_is_better": False,
                                       "spans_ms": fs.get("spans_ms"), "cpu_baseline": {"value": fs.get("cpu_ms"), "unit": "ms", "cores": fs.get("cpu_cores"), "kind": "port"},
                                       "bit_exact_vs_oracle": fs.get("bit_exact_commits_terms_folds"),
                                       "config": {"workload": "one IVC fold step at k = 17, both curves: witness commits, cross-term evaluation, batched cross-term commits, "
                                                              "W / E and instance folding (the witness commits of src/plonk/mod.rs:680-688 and the evaluation + commit + fold chain of src/nifs/vanilla/mod.rs:80-140, 220-251; "
                                                              "circuit synthesis, Poseidon challenges and the step circuit itself stay on the caller's CPU and are not in it)",
                                                  "rows": fs.get("rows"), "advice_columns": fs.get("advice_columns"), "cross_terms": fs.get("cross_terms")}}

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()


def multi_gpu_extras(lib, cm, dist, rank, n_gpus, total_log_n, args):
    """Beside the strong-scaling headline of an N-GPU run: (a) the weak-scaling variant (2^22 pairs
    per GPU, all ranks), (b) on rank 0 alone the SAME 2^total MSM on one GPU -- the N = 1 point of the
    strong-scaling curve, measured in the same job."""
    import torch
    from mira_amd.dist import ShardedCommitmentKey
    ex = {}
    cid = cm.CURVE_BN256

    on_gpu = not args.rehearse_one_gpu                     # tensors of the timing exchange live where the backend works

    def barrier():
        lib.check(lib.c.mira_dev_sync())
        if on_gpu:
            torch.cuda.synchronize()
        dist.barrier()

    def timed(fn, reps):
        fn(); barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) / reps
    try:
        n = 1 << 22
        wkey = ShardedCommitmentKey.synthetic(cid, n * n_gpus, seed=0x5745414B, window_bits=args.window_bits)
        d = cm.synth_scalars_device(cid, n, index0=rank * n, seed=0x5745414C)
        dt = timed(lambda: wkey.commit_device(d, n * n_gpus), 10)
        ex["weak_2p22_per_gpu"] = {"ms_per_step": round(dt * 1e3, 4), "M_pairs_per_s": round(n * n_gpus / dt / 1e6, 2), "scaling": "weak"}
        wkey.key.close(); lib.free(d)
    except Exception as e:
        ex["weak_2p22_per_gpu"] = {"error": repr(e)}
    try:
        res = None
        if rank == 0:
            n = 1 << total_log_n
            key1 = cm.CommitmentKey.synthetic(cid, n)
            d1 = cm.synth_scalars_device(cid, n)
            key1.commit_device(d1, n)
            t0 = time.perf_counter()
            for _ in range(3):
                p1 = key1.commit_device(d1, n)
            dt = (time.perf_counter() - t0) / 3
            res = {"n_gpus": 1, "ms_per_step": round(dt * 1e3, 3), "M_pairs_per_s": round(n / dt / 1e6, 2),
                   "note": f"the same 2^{total_log_n} MSM on rank 0's GPU alone, same job"}
            key1.close(); lib.free(d1)
        dist.barrier()
        ex["strong_scaling_reference_1gpu"] = res
    except Exception as e:
        ex["strong_scaling_reference_1gpu"] = {"error": repr(e)}
        dist.barrier()
    return ex


def load_traffic_key(key):
    """HBM bytes from the committed PMC profile (collected in separate rocprofv3 --pmc passes,
    corrected as MI355X_MICROARCH.md prescribes), or null."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(key)
    except Exception:
        return None


def load_traffic(log_n):
    return load_traffic_key(f"k_accumulate_2p{log_n}")


# Issue-derived ceiling of the field multiplier: one f29_mul is 162 v_mad_u64_u32 (81 products + 81 reduction
# products) and about 85 other VALU instructions; v_mad_u64_u32 issues once per 4.76 cycles per SIMD with all wave
# slots full, a simple VALU instruction once per 2.08 (profiles/r01_a_microbench.txt); 1024 SIMDs x 64 lanes at
# the 2.4 GHz maximum clock (the chip sustains about 2.1 GHz under this load: the kernel can never reach it).
MODMUL_MADS, MODMUL_OTHER, MAD_CYC, VALU_CYC, MAX_CLOCK_GHZ = 162, 85, 4.76, 2.08, 2.4
PEAK_ISSUE_MAD_ONLY = 1024 * 64 * MAX_CLOCK_GHZ / (MODMUL_MADS * MAD_CYC)
PEAK_ISSUE_ALL = 1024 * 64 * MAX_CLOCK_GHZ / (MODMUL_MADS * MAD_CYC + MODMUL_OTHER * VALU_CYC)
MADD_ISSUE_CYCLES = round(1467 * 4.76 + 82 * 5.07 + 152 * 4.36 + 144 * 4.8 + 530 * 2.08)
PEAK_MADD = 1024 * 64 * MAX_CLOCK_GHZ / MADD_ISSUE_CYCLES


def cpu_baseline_msm(lib, cm, key, d_scalars, n, window_bits):
    """The oracle's restatement of best_multiexp (kind "port") on a bounded sample of the same
    workload, all host cores; the same sample re-run on the GPU gives the parity flag."""
    from oracle import cref as C
    cpu = host_cpu_info()
    threads = cpu["effective"]                             # rayon's default: available_parallelism (mask and cgroup quota)
    C.set_threads(threads)                                 # ... for every later CPU leg of this run as well
    sample = min(n, 1 << 22 if threads >= 12 else 1 << 20)  # ~1.5 M bucket additions per second per core
    bases = key.download(0, sample)
    sc = lib.download(d_scalars, (sample, 4))
    t0 = time.perf_counter()
    want = C.msm_pippenger(cm.CURVE_BN256, sc, bases, threads)
    dt = time.perf_counter() - t0
    got = key.commit_device(d_scalars, sample)
    one = min(sample, 1 << 17)                             # the same algorithm on ONE thread (c = ceil(ln n) = 12 there)
    t0 = time.perf_counter()
    C.msm_pippenger(cm.CURVE_BN256, sc[:one], bases[:one], 1)
    dt1 = time.perf_counter() - t0
    base = {"value": round(sample / dt / 1e6, 4), "unit": "M scalar-point pairs/s", "cores": threads, "kind": "port",
            "sample": f"first 2^{sample.bit_length() - 1} of the 2^{n.bit_length() - 1} pairs, {dt:.2f} s, C restatement of halo2 best_multiexp "
                      f"(chunk per thread, c = ceil(ln chunk) windows)",
            "host": cpu, "one_thread": {"value": round(one / dt1 / 1e6, 4), "sample": f"first 2^{one.bit_length() - 1} pairs, {dt1:.2f} s"}}
    return base, {"sample_pairs": sample, "bit_exact_vs_oracle": bool((got == want).all())}


def extras(lib, cm, with_cpu, skip_key_load=False):
    ex = {}
    lib.check(lib.c.mira_msm_set_window_bits(0))
    # ---- NTT 2^24 over bn256::Fr (BASELINE configs[2]) ---------------------------------------
    try:
        k = 24
        n = 1 << k
        d = cm.synth_scalars_device(cm.CURVE_BN256, n, seed=0x4E5454)
        from mira_amd import fft as F
        F.fft_device(d, k)                       # warm-up: builds the twiddle tables
        lib.check(lib.c.mira_set_timing(1))

        def timed(reps):
            walls, acc = [], {}
            for _ in range(reps):
                t0 = time.perf_counter()
                F.fft_device(d, k)
                walls.append(time.perf_counter() - t0)
                for name, ms in lib.timings():
                    acc.setdefault(name, []).append(ms)
            return sorted(walls)[reps // 2], {a: sorted(b)[len(b) // 2] for a, b in acc.items()}

        # Two figures, because the chip's clock decides a VALU-bound 2 ms kernel: `first_transforms` = the median of the first
        # seven transforms after this process has kept the GPU idle (what rounds 1-3 reported as "ms": the clock is still
        # ramping); "ms" = the median of 40 transforms behind 20 untimed ones, i.e. under a sustained load, which is how a
        # prover that keeps the GPU busy meets the transform (measured round 4: the same kernel reads 2.33 and 2.06 ms)
        dt_first, acc_first = timed(7)
        for _ in range(20):
            F.fft_device(d, k)
        dt, acc = timed(40)
        lib.check(lib.c.mira_set_timing(0))
        passes = sorted(a for a in acc if a.startswith("ntt_pass") or a == "ntt_single")
        kern = sum(acc[a] for a in passes)                         # EVERY butterfly pass of the transform (2^24: three)
        ex["ntt_2p24"] = {"ms": round(dt * 1e3, 3), "M_elements_per_s": round(n / dt / 1e6, 2), "stages_ms": {a: round(b, 4) for a, b in acc.items()},
                          "timing": "median of 40 transforms behind 20 untimed ones (sustained clock)",
                          "first_transforms": {"ms": round(dt_first * 1e3, 3), "stages_ms": {a: round(b, 4) for a, b in acc_first.items()},
                                               "timing": "median of the first 7 transforms after an idle GPU (the figure of rounds 1-3)"},
                          "roofline": {"bound": "hbm", "kernel": "k_ntt_wave", "passes": len(passes), "kernel_ms_all_passes": round(kern, 4),
                                       "achieved": round(NTT_BYTES_PER_ELEM * n / (kern * 1e-3) / 1e9, 2) if kern else None,
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(NTT_BYTES_PER_ELEM * n / (kern * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if kern else None,
                                       "algorithmic_bytes": NTT_BYTES_PER_ELEM * n, "traffic": load_traffic_key("ntt_2p24"),
                                       "traffic_source": "profiles/pmc_traffic.json (separate rocprofv3 --pmc passes on the builder's box, all passes of one transform), not this run"}}
        if with_cpu:
            from oracle import cref as C
            ks = 20
            a = lib.download(d, (1 << ks, 4))
            t0 = time.perf_counter()
            want = C.fft(a, ks)
            dtc = time.perf_counter() - t0
            got = F.fft(a, ks)
            ex["ntt_2p24"]["cpu_baseline"] = {"value": round((1 << ks) / dtc / 1e6, 3), "unit": "M elements/s", "cores": C.num_threads(), "kind": "port",
                                              "sample": f"2^{ks}-point fft, {dtc:.2f} s, restatement of src/fft.rs best_fft"}
            ex["ntt_2p24"]["bit_exact_vs_oracle_2p20"] = bool((got == want).all())
        lib.free(d)
    except Exception as e:   # extras never take the headline down
        ex["ntt_2p24"] = {"error": repr(e)}

    # ---- a measured ceiling beside the nominal 8 TB/s: a device-to-device copy of 2 GiB, far beyond the
    # 256 MiB Infinity Cache, bytes read + bytes written over the wall time of mira_dev_copy (which ends synchronised)
    try:
        nbytes = 2 << 30
        d_a, d_b = lib.alloc(nbytes), lib.alloc(nbytes)
        lib.copy(d_b, d_a, nbytes)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); lib.copy(d_b, d_a, nbytes); ts.append(time.perf_counter() - t0)
        lib.free(d_a); lib.free(d_b)
        dt = sorted(ts)[2]
        ex["hbm_copy_measured"] = {"GB_per_s": round(2 * nbytes / dt / 1e9, 1), "bytes_copied": nbytes, "ms": round(dt * 1e3, 3),
                                   "frac_of_nominal_peak": round(2 * nbytes / dt / 1e9 / HBM_PEAK_GBS, 3),
                                   "note": "hipMemcpyAsync device to device + synchronise; read + write traffic counted"}
    except Exception as e:
        ex["hbm_copy_measured"] = {"error": repr(e)}

    # ---- the rest of the metric's range, 2^20 / 2^24 / 2^26 pairs, same path as the headline (16-bit
    # windows, scalars and key resident in HBM), timed here so that the driver's clock is around them
    try:
        lib.check(lib.c.mira_msm_set_window_bits(16))
        sweep = {}
        for log_n in (20, 24, 26):
            n = 1 << log_n
            key = cm.CommitmentKey.synthetic(cm.CURVE_BN256, n)
            d = cm.synth_scalars_device(cm.CURVE_BN256, n)
            for _ in range(max(2, (1 << 24) // n)):             # ~25 ms of the same commit first: the clock a sustained load runs at (see ntt_2p24)
                key.commit_device(d, n)
            reps = 5 if log_n <= 24 else 3
            t0 = time.perf_counter()
            for _ in range(reps):
                key.commit_device(d, n)
            dt = (time.perf_counter() - t0) / reps
            sweep[f"2p{log_n}"] = {"ms": round(dt * 1e3, 3), "M_pairs_per_s": round(n / dt / 1e6, 1), "reps": reps}
            key.close(); lib.free(d)
        ex["msm_sweep_16bit_windows"] = sweep
    except Exception as e:
        ex["msm_sweep_16bit_windows"] = {"error": repr(e)}
    finally:
        lib.check(lib.c.mira_msm_set_window_bits(0))

    # ---- the same range over fixed-base window tables (opt-in, mira_msm_precompute_ex: the key of a prover never changes, so
    # 2^(c w) P_i can sit in HBM -- 13 x the key at 20-bit windows, 12 x at 22): one bucket set, 13 / 12 additions per pair
    # instead of 16.  The tables are built once per key, outside the timing; the point is the per-window path's, bit for bit.
    try:
        tab = {}
        for log_n, width in ((24, 20), (26, 22)):
            n = 1 << log_n
            key = cm.CommitmentKey.synthetic(cm.CURVE_BN256, n)
            d = cm.synth_scalars_device(cm.CURVE_BN256, n)
            ref = key.commit_device(d, n)
            t0 = time.perf_counter(); key.precompute(width); pre_s = time.perf_counter() - t0
            for _ in range(2):
                out = key.commit_device(d, n)
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                out = key.commit_device(d, n)
            dt = (time.perf_counter() - t0) / reps
            windows = -(-254 // width)
            tab[f"2p{log_n}"] = {"ms": round(dt * 1e3, 3), "M_pairs_per_s": round(n / dt / 1e6, 1), "window_bits": width, "additions_per_pair": windows,
                                 "table_bytes": windows * n * 64, "precompute_s": round(pre_s, 2), "same_point_as_per_window_path": bool((out == ref).all())}
            key.close(); lib.free(d)
        ex["msm_sweep_fixed_base"] = tab
    except Exception as e:
        ex["msm_sweep_fixed_base"] = {"error": repr(e)}

    # ---- the reference's largest real commits (examples/groth16/main.rs:47-75: k = 24 tables over 2^27 .. 2^28-point keys):
    # a 14 x 2^24-pair witness commit and a 2^28-pair commit -- the latter is 2^32 sorted entries under 16-bit windows, more than
    # the 32-bit entry offsets of one pass: cut into point chunks inside the launch sequence (msm_host.cuh)
    try:
        big = {}
        n28 = 1 << 28
        key = cm.CommitmentKey.synthetic(cm.CURVE_BN256, n28, seed=0x3238)
        d = cm.synth_scalars_device(cm.CURVE_BN256, n28, seed=0x3239)
        for name, n in (("14x2p24", 14 << 24), ("2p28", n28)):
            key.commit_device(d, n)
            t0 = time.perf_counter(); key.commit_device(d, n); dt = time.perf_counter() - t0
            c_, w_ = ctypes.c_int32(), ctypes.c_int32()
            lib.check(lib.c.mira_msm_last_plan(ctypes.byref(c_), ctypes.byref(w_)))
            big[name] = {"pairs": n, "ms": round(dt * 1e3, 2), "M_pairs_per_s": round(n / dt / 1e6, 1), "window_bits": c_.value,
                         "passes": -(-(n * w_.value) // ((1 << 32) - 1))}
        key.close(); lib.free(d)
        lib.check(lib.c.mira_trim(0, None))
        ex["msm_reference_largest"] = big
    except Exception as e:
        ex["msm_reference_largest"] = {"error": repr(e)}

    # ---- the boundary's real cost: commit(&self, v: &[C::Scalar]) receives HOST memory
    # (src/commitment.rs:78).  mira_msm cuts the scalars into point chunks whose PCIe copies run beside
    # the kernels of the previous chunk; beside it the same call with one up-front copy (chunking off)
    # and the HBM-resident commit of the headline.
    try:
        from mira_amd import _lib as L
        lib.check(lib.c.mira_msm_set_window_bits(16))
        n = 1 << 22
        key = cm.CommitmentKey.synthetic(cm.CURVE_BN256, n)
        d = cm.synth_scalars_device(cm.CURVE_BN256, n)
        sc = lib.download(d, (n, 4))                           # ordinary pageable host memory

        def med(fn, reps=7):
            fn()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter(); out = fn(); ts.append((time.perf_counter() - t0) * 1e3)
            return sorted(ts)[reps // 2], out
        t_dev, p_dev = med(lambda: key.commit_device(d, n))
        t_host, p_host = med(lambda: key.commit(sc))
        lib.tune(L.TUNE_HOST_CHUNK_MIN_N, 1 << 40)
        t_one, p_one = med(lambda: key.commit(sc))
        lib.tune(L.TUNE_HOST_CHUNK_MIN_N, -1)
        ex["msm_2p22_host_scalars"] = {"hbm_resident_ms": round(t_dev, 3), "host_scalars_ms": round(t_host, 3), "host_scalars_one_copy_ms": round(t_one, 3),
                                       "pcie_inclusive_over_resident": round(t_host / t_dev, 3), "M_pairs_per_s_pcie_inclusive": round(n / t_host / 1e3, 1),
                                       "same_point": bool((p_dev == p_host).all() and (p_dev == p_one).all()),
                                       "note": "128 MiB of pageable host scalars per call; chunks of 2^18, 2^19, 2^20, ... pairs"}
        key.close(); lib.free(d)
    except Exception as e:
        ex["msm_2p22_host_scalars"] = {"error": repr(e)}
    finally:
        lib.check(lib.c.mira_msm_set_window_bits(0))

    # ---- the same 2^22 MSM over fixed-base window tables (opt-in mode) ----
    try:
        n = 1 << 22
        key = cm.CommitmentKey.synthetic(cm.CURVE_BN256, n)
        d = cm.synth_scalars_device(cm.CURVE_BN256, n)
        ref = key.commit_device(d, n)
        t0 = time.perf_counter(); key.precompute(); pre_s = time.perf_counter() - t0
        key.commit_device(d, n)
        lib.check(lib.c.mira_set_timing(1))
        reps, acc = 5, {}
        t0 = time.perf_counter()
        for _ in range(reps):
            out = key.commit_device(d, n)
            for name, ms in lib.timings():
                acc[name] = acc.get(name, 0.0) + ms / reps
        dt = (time.perf_counter() - t0) / reps
        lib.check(lib.c.mira_set_timing(0))
        ex["msm_2p22_fixed_base"] = {"M_pairs_per_s": round(n / dt / 1e6, 1), "ms": round(dt * 1e3, 3), "stages_ms": {a: round(b, 4) for a, b in acc.items()},
                                     "same_point_as_per_window_path": bool((out == ref).all()),
                                     "precompute_s": round(pre_s, 3), "table_bytes": 13 * n * 64,
                                     "note": "mira_msm_precompute: 13 window tables 2^(20w) P_i in HBM, one set of 2^19 buckets, 13 additions "
                                             "per pair; the tables depend on the key only and are built once per key, outside this timing"}
        key.close(); lib.free(d)
    except Exception as e:
        ex["msm_2p22_fixed_base"] = {"error": repr(e)}

    # ---- witness folding W1 + r W2 (src/plonk/mod.rs:1099-1110): HBM-bound ----
    try:
        from mira_amd import fold as FD
        n = 14 << 17                                           # the primary witness vector of a k = 17 step
        # eight independent (W1, W2, out) sets = 1.4 GB, visited in turn: every call streams from and to
        # HBM (one 176 MB set alone would sit in the 256 MiB Infinity Cache)
        sets = 8
        d1s = [cm.synth_scalars_device(cm.CURVE_BN256, n, seed=0x57 + 2 * i, kind=1) for i in range(sets)]
        d2s = [cm.synth_scalars_device(cm.CURVE_BN256, n, seed=0x58 + 2 * i) for i in range(sets)]
        dos = [lib.alloc(n * 32) for _ in range(sets)]
        d1, d2, do = d1s[0], d2s[0], dos[0]
        r = lib.download(cm.synth_scalars_device(cm.CURVE_BN256, 1, seed=0x59), (1, 4))[0]
        for i in range(sets):
            FD.fold_witness_device(FD.FIELD_FR, dos[i], d1s[i], d2s[i], r, n)
        lib.check(lib.c.mira_set_timing(1))
        ms, ms_cached = 0.0, 0.0
        for rep in range(2):
            for i in range(sets):
                FD.fold_witness_device(FD.FIELD_FR, dos[i], d1s[i], d2s[i], r, n)
                ms += dict(lib.timings())["fold_witness"] / (2 * sets)
        for _ in range(5):                                     # the same set again and again: cache-resident
            FD.fold_witness_device(FD.FIELD_FR, do, d1, d2, r, n)
            ms_cached += dict(lib.timings())["fold_witness"] / 5
        lib.check(lib.c.mira_set_timing(0))
        gbs = 96 * n / (ms * 1e-3) / 1e9
        ex["fold_witness_14x2p17"] = {"ms": round(ms, 4), "G_elements_per_s": round(n / ms / 1e6, 2),
                                      "ms_same_buffers_infinity_cache_resident": round(ms_cached, 4),
                                      "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                   "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes": 96 * n,
                                                   "working_set_bytes": 96 * n * sets}}
        if with_cpu:
            from oracle import cref as C
            m = 1 << 20
            a, b2 = lib.download(d1, (m, 4)), lib.download(d2, (m, 4))
            t0 = time.perf_counter(); want = C.fold_witness(1, a, b2, r); dtc = time.perf_counter() - t0
            ex["fold_witness_14x2p17"]["cpu_baseline"] = {"value": round(m / dtc / 1e9, 4), "unit": "G elements/s", "cores": C.num_threads(), "kind": "port",
                                                          "sample": f"first 2^20 elements, {dtc * 1e3:.1f} ms"}
            ex["fold_witness_14x2p17"]["bit_exact_vs_oracle_2p20"] = bool((lib.download(do, (m, 4)) == want).all())
        for p in d1s + d2s + dos:
            lib.free(p)
    except Exception as e:
        ex["fold_witness_14x2p17"] = {"error": repr(e)}

    # ---- fold-step MSM schedule at k = 17 -----------------------------
    # per curve: one witness commit (witness-like scalars) + the cross-term commits (uniform), the
    # latter both one call at a time (as the reference issues them) and as one batched submission
    try:
        k = 17
        n = 1 << k
        from harness import main_gate as MG_sched
        plan = MG_sched.fold_step_msm_schedule(k)              # {curve: (witness length, cross terms)}: (14 << k, 6) and (7 << k, 5), derived from the reference's configure functions
        keys, wit, cross = {}, {}, {}
        for c, (nw, cnt) in plan.items():
            keys[c] = cm.CommitmentKey.synthetic(c, nw, seed=0x464F4C44 + c)
            wit[c] = cm.synth_scalars_device(c, nw, seed=0x1000 + c, kind=1)
            cross[c] = lib.alloc(cnt * n * 32)
            for i in range(cnt):
                lib.check(lib.c.mira_synth_scalars_device(c, n, 0, 0x2000 + 16 * c + i, 0, ctypes.c_void_p(cross[c] + i * n * 32)))

        def run(batched):
            pts = []
            for c, (nw, cnt) in plan.items():
                pts.append(keys[c].commit_device(wit[c], nw))
                if batched:
                    pts.extend(keys[c].commit_batch_device(cross[c], n, cnt))
                else:
                    pts.extend(keys[c].commit_device(cross[c] + i * n * 32, n) for i in range(cnt))
            return pts
        # The figures WITHOUT suffix are the per-window path on plain keys, as in rounds 1 - 3: the library's automatic endomorphism
        # copy (round 4: built at the first commit that takes the GLV split) is switched off for them, so that they stay comparable;
        # the *_glv figures below are what a caller gets since round 4 WITHOUT any opt-in.
        from mira_amd import _lib as L_
        lib.tune(L_.TUNE_GLV_AUTO_MAX_LOG, 0)
        SETTLE = 15           # commits of a shape before it is timed: the library checks the planner's width against its neighbours on the first ones (MIRA_TUNE_WIDTH_TRIALS)
        for _ in range(SETTLE):
            run(False); run(True)                               # warm-up
        def median_ms(batched, reps=5):
            ts, pts = [], None
            for _ in range(reps):
                t0 = time.perf_counter(); pts = run(batched); ts.append((time.perf_counter() - t0) * 1e3)
            return sorted(ts)[reps // 2], pts
        seq_ms, seq_pts = median_ms(False)
        bat_ms, bat_pts = median_ms(True)
        # the same schedule as a Rust caller issues it: every vector in HOST memory, through mira_msm /
        # mira_msm_batch (src/nifs/vanilla/mod.rs:124-127 commits one by one; src/plonk/mod.rs:680-688)
        h_wit = {c: lib.download(wit[c], (plan[c][0], 4)) for c in plan}
        h_cross = {c: [lib.download(cross[c] + i * n * 32, (n, 4)) for i in range(plan[c][1])] for c in plan}

        def run_host(batched):
            pts = []
            for c, (nw, cnt) in plan.items():
                pts.append(keys[c].commit(h_wit[c]))
                if batched:
                    pts.extend(keys[c].commit_batch(h_cross[c]))
                else:
                    pts.extend(keys[c].commit(v) for v in h_cross[c])
            return pts
        for _ in range(SETTLE):
            run_host(False); run_host(True)
        hs, hb = [], []
        for _ in range(5):
            t0 = time.perf_counter(); hseq_pts = run_host(False); hs.append((time.perf_counter() - t0) * 1e3)
            t0 = time.perf_counter(); hbat_pts = run_host(True); hb.append((time.perf_counter() - t0) * 1e3)
        single = []
        for _ in range(40):                                     # the clock a sustained load runs at (the legs in front leave the GPU waiting for PCIe)
            keys[cm.CURVE_BN256].commit_device(cross[cm.CURVE_BN256], n)
        for _ in range(9):
            t0 = time.perf_counter(); keys[cm.CURVE_BN256].commit_device(cross[cm.CURVE_BN256], n); single.append((time.perf_counter() - t0) * 1e3)
        # the library's default since round 4, 2 x the key's HBM: the endomorphism copy, built by the first commit that takes the
        # split -- every scalar as two 127-bit halves over half the windows wherever the planners estimate the split ahead; the same
        # 13 calls, one per commit, and the batched form
        lib.tune(L_.TUNE_GLV_AUTO_MAX_LOG, -1)                   # the default: both keys get their copies inside the first commits below
        for _ in range(SETTLE):
            run(False); run(True)
        tglv, single_glv = [], []
        for _ in range(5):
            t0 = time.perf_counter(); glv_pts = run(False); tglv.append((time.perf_counter() - t0) * 1e3)
        for _ in range(40):                                     # the clock a sustained load runs at (the legs in front leave the GPU waiting for PCIe)
            keys[cm.CURVE_BN256].commit_device(cross[cm.CURVE_BN256], n)
        for _ in range(9):
            t0 = time.perf_counter(); keys[cm.CURVE_BN256].commit_device(cross[cm.CURVE_BN256], n); single_glv.append((time.perf_counter() - t0) * 1e3)
        run(True)
        tglvb = []
        for _ in range(5):
            t0 = time.perf_counter(); glvb_pts = run(True); tglvb.append((time.perf_counter() - t0) * 1e3)
        # opt-in: shared-bucket fixed-base tables on both keys (mira_msm_precompute_ex(handle, 15) and (handle, 13):
        # 18 + 20 x the key's HBM): all windows share one bucket set, no Horner epilogue; every commit takes the set that
        # is fastest for its length and, from the second commit of a shape on, for the bit lengths of its scalars (the
        # witness commits: 13 bits, dense 2^17-pair commits and the batches: 15) -- the same 13 calls, one per commit
        for c in plan:
            for width in (15, 13):
                keys[c].precompute(width)
        for _ in range(6):                                       # (the library times every set of a key on the first commits of a shape, twice each)
            run(False); run(True)
        t16, single16 = [], []
        for _ in range(5):
            t0 = time.perf_counter(); t16_pts = run(False); t16.append((time.perf_counter() - t0) * 1e3)
        for _ in range(40):                                     # the clock a sustained load runs at (the legs in front leave the GPU waiting for PCIe)
            keys[cm.CURVE_BN256].commit_device(cross[cm.CURVE_BN256], n)
        for _ in range(9):
            t0 = time.perf_counter(); keys[cm.CURVE_BN256].commit_device(cross[cm.CURVE_BN256], n); single16.append((time.perf_counter() - t0) * 1e3)
        run(True)
        t16b = []
        for _ in range(5):
            t0 = time.perf_counter(); t16b_pts = run(True); t16b.append((time.perf_counter() - t0) * 1e3)
        ex["fold_step_k17"] = {"msm_calls": 13, "pairs": sum(nw + cnt * n for nw, cnt in plan.values()),
                               "gpu_ms": round(bat_ms, 3), "gpu_ms_one_call_per_commit": round(seq_ms, 3),
                               "gpu_ms_host_scalars_one_call_per_commit": round(sorted(hs)[2], 3), "gpu_ms_host_scalars_batched": round(sorted(hb)[2], 3),
                               "one_commit_131072_pairs_ms": round(sorted(single)[4], 3),
                               "gpu_ms_one_call_per_commit_glv": round(sorted(tglv)[2], 3), "one_commit_131072_pairs_ms_glv": round(sorted(single_glv)[4], 3),
                               "gpu_ms_glv": round(sorted(tglvb)[2], 3),
                               "glv_same_points": bool(all((a == b).all() for a, b in zip(seq_pts, glv_pts)) and all((a == b).all() for a, b in zip(seq_pts, glvb_pts))),
                               "gpu_ms_one_call_per_commit_tables": round(sorted(t16)[2], 3), "one_commit_131072_pairs_ms_tables": round(sorted(single16)[4], 3),
                               "gpu_ms_tables": round(sorted(t16b)[2], 3), "table_widths": [13, 15],
                               "tables_same_points": bool(all((a == b).all() for a, b in zip(seq_pts, t16_pts)) and all((a == b).all() for a, b in zip(seq_pts, t16b_pts))),
                               "host_scalar_bytes": 32 * sum(nw + cnt * n for nw, cnt in plan.values()),
                               "batched_equals_sequential": bool(all((a == b).all() for a, b in zip(seq_pts, bat_pts))
                                                                 and all((a == b).all() for a, b in zip(seq_pts, hseq_pts))
                                                                 and all((a == b).all() for a, b in zip(seq_pts, hbat_pts))),
                               "note": "MSM schedule of one IVC fold step: per curve 1 witness commit + 6/5 cross-term commits "
                                       "(gpu_ms: scalars in HBM, cross terms as one mira_msm_batch per curve; *_host_scalars_*: every vector "
                                       "handed over in host memory, PCIe included); the Rust driver (examples/groth16) cannot be built here"}
        if with_cpu:
            from oracle import cref as C
            cpu_in = []
            for c, (nw, cnt) in plan.items():
                bases = keys[c].download()
                cpu_in.append((c, bases, lib.download(wit[c], (nw, 4))))
                for i in range(cnt):
                    cpu_in.append((c, bases[:n], lib.download(cross[c] + i * n * 32, (n, 4))))
            t0 = time.perf_counter()
            cpu_pts = [C.msm_pippenger(c, sc, bs) for c, bs, sc in cpu_in]
            cpu_ms = (time.perf_counter() - t0) * 1e3
            ex["fold_step_k17"].update({"cpu_ms": round(cpu_ms, 1), "cpu_cores": C.num_threads(), "cpu_kind": "port",
                                        "bit_exact_all_13": bool(all((a == b).all() for a, b in zip(bat_pts, cpu_pts)))})
    except Exception as e:
        ex["fold_step_k17"] = {"error": repr(e)}

    # ---- cross-term evaluation at k = 17 --------------------------------
    # The reference's own graphs: S.custom_gates_lookup_compressed.grouped().iter_from_first()
    # (src/nifs/vanilla/mod.rs:100-104) for the circuits of an IVC step -- the MainGate<5> gate
    # (src/main_gate.rs:543-589), homogenised and grouped by the power of the folding variable
    # (mira_amd/main_gate.py, expression.py, grouped_poly.py; pinned by the reference's Display tests).
    # Primary circuit (BN256 scalars): two gates compressed with a challenge, degree 6, 6 cross terms over
    # 30 fixed + 2 x 14 advice columns.  Secondary (Grumpkin scalars): one gate, degree 5, 5 cross terms
    # over 15 fixed + 2 x 7 advice columns.  2^17 rows each; outputs stay in HBM for the commits.
    try:
        from harness import graph_evaluator as G
        from harness import main_gate as MG
        k = 17
        n = 1 << k
        res = {}
        for name, c, gates, field in (("primary_bn256", cm.CURVE_BN256, 2, G.FIELD_FR), ("secondary_grumpkin", cm.CURVE_GRUMPKIN, 1, G.FIELD_FQ)):
            cg, ctx = MG.compressed_circuit(5, gates)
            evs = [G.GraphEvaluator.new(t, field) for t in cg.grouped.iter_from_first()]
            d_fix = cm.synth_scalars_device(c, ctx.num_fixed * n, seed=0x3000 + c)
            d_w1 = cm.synth_scalars_device(c, ctx.num_advice * n, seed=0x3100 + c)
            d_w2 = cm.synth_scalars_device(c, ctx.num_advice * n, seed=0x3200 + c, kind=1)
            chal = [(0x1234567 + 977 * j) ** 7 % G.MODULUS[field] for j in range(2 * ctx.num_challenges)]
            dom = G.PlonkEvalDomain(ctx.num_advice, 0, chal, [], [d_fix + j * n * 32 for j in range(ctx.num_fixed)],
                                    [(d_w1, ctx.num_advice * n)], [(d_w2, ctx.num_advice * n)], n)
            cols = dom.columns()
            d_out = lib.alloc(len(evs) * n * 32)
            run = lambda: G.GraphEvaluator.evaluate_batch_device(evs, cols, chal, n, [d_out + i * n * 32 for i in range(len(evs))])     # one mira_graph_eval_batch
            run()
            walls = []
            for _ in range(7):                       # median: see the NTT leg
                t0 = time.perf_counter(); run(); walls.append((time.perf_counter() - t0) * 1e3)
            wall = sorted(walls)[3]
            lib.check(lib.c.mira_set_timing(1))
            kerns = []
            for ev in evs:
                ev.evaluate_device(cols, chal, n, d_out=d_out)
                ev.evaluate_device(cols, chal, n, d_out=d_out)
                kerns.append(round(dict(lib.timings())["graph_eval"], 4))
            lib.check(lib.c.mira_set_timing(0))
            run()
            ncalc = [ev.num_intermediates for ev in evs]
            # the same d vectors by evaluation + interpolation (CrossTermPlan: d + 1 evaluations of the gate polynomial f,
            # one linear combination per term) -- what the fold-step leg below uses
            plan = G.CrossTermPlan.from_compressed_gates(cg, ctx, field)
            d_plan = lib.alloc(len(evs) * n * 32)
            run_plan = lambda: plan.evaluate_device(cols, chal, n, d_plan)
            run_plan()
            walls_p = []
            for _ in range(7):
                t0 = time.perf_counter(); run_plan(); walls_p.append((time.perf_counter() - t0) * 1e3)
            same = bool((lib.download(d_plan, (len(evs), n, 4)) == lib.download(d_out, (len(evs), n, 4))).all())
            # once per circuit: a kernel of its own for every evaluation point, compiled at run time from the graph's
            # instruction stream (mira_graph_specialize, hiprtc) -- intermediates in registers, no decoding
            t0 = time.perf_counter(); spec_ok = plan.specialize(cols, len(chal)); spec_s = time.perf_counter() - t0
            walls_s, same_s = [], None
            if spec_ok:
                run_plan()
                for _ in range(7):
                    t0 = time.perf_counter(); run_plan(); walls_s.append((time.perf_counter() - t0) * 1e3)
                same_s = bool((lib.download(d_plan, (len(evs), n, 4)) == lib.download(d_out, (len(evs), n, 4))).all())
            lib.free(d_plan)
            res[name] = {"gates": gates, "degree": cg.degree, "graphs": len(evs), "rows": n, "fixed_columns": ctx.num_fixed, "advice_columns": ctx.num_advice,
                         "calculations_per_graph": ncalc, "ms_grouped_graphs": round(wall, 3), "kernel_ms_per_graph_alone": kerns,
                         "G_calculations_per_s": round(sum(ncalc) * n / wall / 1e6, 2),
                         "ms": round(sorted(walls_s)[3], 3) if spec_ok else round(sorted(walls_p)[3], 3), "ms_interpreted": round(sorted(walls_p)[3], 3),
                         "specialized": bool(spec_ok), "specialize_s": round(spec_s, 1), "specialized_equals_grouped": same_s,
                         "calculations_per_point": plan.num_calculations, "points": [str(x) for x in plan.points],
                         "interpolated_equals_grouped": same}
            if with_cpu:
                from oracle import cref as C
                host_cols = list(lib.download(d_fix, (ctx.num_fixed, n, 4))) + list(lib.download(d_w1, (ctx.num_advice, n, 4))) + list(lib.download(d_w2, (ctx.num_advice, n, 4)))
                chal_m = G.to_montgomery(chal, field)
                fo = C.FIELD_FR if field == G.FIELD_FR else C.FIELD_FQ
                t0 = time.perf_counter()
                want = []
                for ev in evs:
                    code, consts, rots = ev.flatten()
                    want.append(C.graph_eval(fo, code, ev.num_intermediates, consts, rots, host_cols, chal_m, n))
                dtc = (time.perf_counter() - t0) * 1e3
                got = lib.download(d_out, (len(evs), n, 4))
                res[name].update({"cpu_ms": round(dtc, 1), "cpu_cores": C.num_threads(), "cpu_kind": "port",
                                  "bit_exact_all_rows": bool(all((got[i] == want[i]).all() for i in range(len(evs))))})
            for p in (d_fix, d_w1, d_w2, d_out):
                lib.free(p)
        res["ms"] = round(res["primary_bn256"]["ms"] + res["secondary_grumpkin"]["ms"], 3)
        res["ms_interpreted"] = round(res["primary_bn256"]["ms_interpreted"] + res["secondary_grumpkin"]["ms_interpreted"], 3)
        res["ms_grouped_graphs"] = round(res["primary_bn256"]["ms_grouped_graphs"] + res["secondary_grumpkin"]["ms_grouped_graphs"], 3)
        res["note"] = ("the d cross terms of the MainGate<5> circuits over 2^17 rows.  ms_grouped_graphs: the reference's own graphs (grouped terms 1..d of the "
                       "homogenised, challenge-compressed gate), one batched submission per circuit.  ms: the same vectors, bit for bit, from d + 1 evaluations of "
                       "the gate polynomial at W1 + x W2 and one linear combination per term (CrossTermPlan), every evaluation point through a kernel of its own compiled at "
                       "run time from the graph (mira_graph_specialize, once per circuit: specialize_s); ms_interpreted: the same through the graph interpreter.  "
                       "Outputs stay in HBM for the batched commit")
        ex["cross_term_eval_k17"] = res
    except Exception as e:
        import traceback
        ex["cross_term_eval_k17"] = {"error": repr(e), "trace": traceback.format_exc()[-600:]}

    # ---- one NIFS fold step at k = 17, the whole device-resident chain (BASELINE configs[3]) -------
    # per curve, in the reference's order: generate_plonk_trace's witness commit
    # (src/plonk/mod.rs:680-688) -> commit_cross_terms = row-wise evaluation of the d cross-term
    # graphs + their commits (src/nifs/vanilla/mod.rs:100-127) -> RelaxedPlonkWitness::fold (W and E,
    # src/plonk/mod.rs:1097-1134) -> the commitment side of RelaxedPlonkInstance::fold (:986-999,
    # 1049-1053).  Primary: BN256, two MainGate<5> (14 advice, 30 fixed columns), 6 cross terms;
    # secondary: Grumpkin, one MainGate<5> (7 advice, 15 fixed), 5 cross terms -- the graphs of the leg above.
    # The Rust driver cannot be built here: witnesses are witness-like scalars (not a satisfying trace: the
    # arithmetic does not care), fixed columns uniform, the challenges synthetic.  CPU: the oracle's
    # restatements of the same calls on the effective host cores, same inputs, every output compared.
    try:
        from harness import graph_evaluator as G
        from harness import main_gate as MG
        from mira_amd import fold as FD
        k = 17
        n = 1 << k
        shape = {cm.CURVE_BN256: (2, G.FIELD_FR), cm.CURVE_GRUMPKIN: (1, G.FIELD_FQ)}
        st = {}
        for c, (gates, field) in shape.items():
            cg, ctx = MG.compressed_circuit(5, gates)
            ncol, cnt = ctx.num_advice, cg.degree
            nw = ncol * n
            d_w1 = cm.synth_scalars_device(c, nw, seed=0x5100 + c, kind=1)       # accumulator witness
            d_w2 = cm.synth_scalars_device(c, nw, seed=0x5200 + c, kind=1)       # the step's new witness
            d_fix = cm.synth_scalars_device(c, ctx.num_fixed * n, seed=0x5300 + c)
            d_e = cm.synth_scalars_device(c, n, seed=0x5400 + c)
            chal = [(0x1234567 + 977 * j) ** 7 % G.MODULUS[field] for j in range(2 * ctx.num_challenges)]   # [c1.., u1, c2.., u2]
            dom = G.PlonkEvalDomain(ncol, 0, chal, [], [d_fix + j * n * 32 for j in range(ctx.num_fixed)], [(d_w1, nw)], [(d_w2, nw)], n)
            st[c] = dict(key=cm.CommitmentKey.synthetic(c, nw, seed=0x5500 + c), d_w1=d_w1, d_w2=d_w2, d_fix=d_fix, d_e=d_e, cols=dom.columns(), chal=chal, nw=nw,
                         cnt=cnt, field=field, ctx=ctx, evs=[G.GraphEvaluator.new(t, field) for t in cg.grouped.iter_from_first()], d_terms=lib.alloc(cnt * n * 32),
                         plan=G.CrossTermPlan.from_compressed_gates(cg, ctx, field),
                         d_wout=lib.alloc(nw * 32), d_enew=lib.alloc(n * 32), r_int=(0x5EED0000 + 7919 * c) ** 5 % G.MODULUS[field],
                         acc_w=cm.CommitmentKey.default_value(), acc_e=cm.CommitmentKey.default_value())
            st[c]["r"] = G.to_montgomery([st[c]["r_int"]], field)[0]
            st[c]["term_ptrs"] = [st[c]["d_terms"] + i * n * 32 for i in range(cnt)]

        def fold_step():
            spans = {"witness_commit": 0.0, "evaluation": 0.0, "commit": 0.0, "fold": 0.0}
            outs = {}
            for c, s_ in st.items():
                t0 = time.perf_counter()
                w_commit = s_["key"].commit_device(s_["d_w2"], s_["nw"])
                t1 = time.perf_counter()
                s_["plan"].evaluate_device(s_["cols"], s_["chal"], n, s_["d_terms"])       # the d cross terms (checked below against the reference's grouped graphs)
                t2 = time.perf_counter()
                t_commits = s_["key"].commit_batch_device(s_["d_terms"], n, s_["cnt"])
                t3 = time.perf_counter()
                d_e_new = s_["d_enew"]                                              # W' = W1 + r W2 and E' = E + sum r^(k+1) T_k, one submission
                FD.fold_relaxed_witness_device(s_["field"], s_["d_wout"], s_["d_w1"], s_["d_w2"], s_["nw"], d_e_new, s_["d_e"], s_["term_ptrs"], s_["r"], n)
                # W1 + r W2 and E_commit + sum r^(k+1) T_k (src/plonk/mod.rs:986-999, 1049-1053), one parallel region on the host
                folded_ws, folded_e = FD.fold_instance_commitments(c, s_["acc_w"], w_commit, s_["r"], s_["acc_e"], t_commits)
                folded_w = folded_ws[0]
                t4 = time.perf_counter()
                spans["witness_commit"] += t1 - t0; spans["evaluation"] += t2 - t1; spans["commit"] += t3 - t2; spans["fold"] += t4 - t3
                outs[c] = dict(w_commit=w_commit, t_commits=t_commits, d_e_new=d_e_new, folded_w=folded_w, folded_e=folded_e)
            return spans, outs
        from mira_amd import _lib as L_
        lib.tune(L_.TUNE_GLV_AUTO_MAX_LOG, 0)                   # `ms`: plain keys as in rounds 1 - 3 (comparable); `ms_glv` below: the library's default since round 4
        for _ in range(15):                                     # the width trials of every shape of commit settle here (MIRA_TUNE_WIDTH_TRIALS)
            fold_step()
        walls_i = []
        for _ in range(5):
            t0 = time.perf_counter(); spans_i, last_i = fold_step(); walls_i.append(((time.perf_counter() - t0) * 1e3, spans_i))
        wall_i, spans_i = sorted(walls_i, key=lambda x: x[0])[2]
        # once per circuit: every evaluation point gets its own run-time compiled kernel (mira_graph_specialize)
        t0 = time.perf_counter()
        spec_ok = all(s_["plan"].specialize(s_["cols"], len(s_["chal"])) for s_ in st.values())
        spec_s = time.perf_counter() - t0
        fold_step()
        walls, last = [], None
        for _ in range(5):
            t0 = time.perf_counter(); spans, last = fold_step(); walls.append(((time.perf_counter() - t0) * 1e3, spans))
        wall, spans = sorted(walls, key=lambda x: x[0])[2]
        same_spec = all((last[c]["w_commit"] == last_i[c]["w_commit"]).all() and (last[c]["t_commits"] == last_i[c]["t_commits"]).all()
                        and (last[c]["folded_e"] == last_i[c]["folded_e"]).all() for c in st)
        ex["nifs_fold_step_k17"] = {"ms": round(wall, 3), "spans_ms": {a: round(b * 1e3, 3) for a, b in spans.items()},
                                    "ms_interpreted_graphs": round(wall_i, 3), "spans_ms_interpreted_graphs": {a: round(b * 1e3, 3) for a, b in spans_i.items()},
                                    "graphs_specialized": bool(spec_ok), "specialize_s": round(spec_s, 1), "specialized_same_points": bool(same_spec),
                                    "rows": n, "advice_columns": [st[c]["ctx"].num_advice for c in st], "fixed_columns": [st[c]["ctx"].num_fixed for c in st],
                                    "cross_terms": [st[c]["cnt"] for c in st], "calculations_per_graph": [[ev.num_intermediates for ev in st[c]["evs"]] for c in st],
                                    "calculations_per_point": [st[c]["plan"].num_calculations for c in st],
                                    "note": "both curves, device-resident vectors: witness commit, the d cross terms of the MainGate<5> circuits (d + 1 evaluations of the gate polynomial "
                                            "+ interpolation: the same vectors as the reference's grouped graphs, which the CPU leg evaluates; every evaluation point through its own run-time "
                                            "compiled kernel, built once per circuit in specialize_s -- ms_interpreted_graphs: through the graph interpreter, as in earlier rounds), "
                                            "batched cross-term commits, W / E folding and instance folding (mira_g1_fold_commitments, host threads); span names follow the reference's tracing spans"}
        # the library's default since round 4 (2 x the keys' HBM): the same chain with the endomorphism copies of both keys, built by the
        # first commits below (the GLV split)
        lib.tune(L_.TUNE_GLV_AUTO_MAX_LOG, -1)
        for _ in range(15):
            fold_step()
        walls_g = []
        for _ in range(5):
            t0 = time.perf_counter(); spans_g, last_g = fold_step(); walls_g.append(((time.perf_counter() - t0) * 1e3, spans_g))
        wall_g, spans_g = sorted(walls_g, key=lambda x: x[0])[2]
        same_g = all((last_g[c]["w_commit"] == last[c]["w_commit"]).all() and (last_g[c]["t_commits"] == last[c]["t_commits"]).all()
                     and (last_g[c]["folded_e"] == last[c]["folded_e"]).all() for c in st)
        ex["nifs_fold_step_k17"].update({"ms_glv": round(wall_g, 3), "spans_ms_glv": {a: round(b * 1e3, 3) for a, b in spans_g.items()}, "glv_same_points": bool(same_g)})
        # opt-in: the same chain over shared-bucket fixed-base tables (mira_msm_precompute_ex(handle, 15), (handle, 13)) --
        # a commitment key is fixed for the whole IVC run, its tables are built once
        for s_ in st.values():
            for width in (15, 13):
                s_["key"].precompute(width)
        for _ in range(6):                                       # the trials among the two sets settle (two commits per set and shape)
            fold_step()
        walls16 = []
        for _ in range(5):
            t0 = time.perf_counter(); spans16, last16 = fold_step(); walls16.append(((time.perf_counter() - t0) * 1e3, spans16))
        wall16, spans16 = sorted(walls16, key=lambda x: x[0])[2]
        same16 = all((last16[c]["w_commit"] == last[c]["w_commit"]).all() and (last16[c]["t_commits"] == last[c]["t_commits"]).all()
                     and (last16[c]["folded_e"] == last[c]["folded_e"]).all() for c in st)
        ex["nifs_fold_step_k17"].update({"ms_tables": round(wall16, 3), "spans_ms_tables": {a: round(b * 1e3, 3) for a, b in spans16.items()}, "table_widths": [13, 15],
                                         "tables_same_points": bool(same16)})
        if with_cpu:
            from oracle import cref as C
            t0 = time.perf_counter()
            ok = True
            for c, s_ in st.items():
                fo = C.FIELD_FR if s_["field"] == G.FIELD_FR else C.FIELD_FQ
                ncol, nfix = s_["ctx"].num_advice, s_["ctx"].num_fixed
                bases = s_["key"].download()
                w1, w2 = lib.download(s_["d_w1"], (s_["nw"], 4)), lib.download(s_["d_w2"], (s_["nw"], 4))
                host_cols = list(lib.download(s_["d_fix"], (nfix, n, 4))) + [w1[j * n:(j + 1) * n] for j in range(ncol)] + [w2[j * n:(j + 1) * n] for j in range(ncol)]
                chal_m = G.to_montgomery(s_["chal"], s_["field"])
                want_w = C.msm_pippenger(c, w2, bases)
                terms = []
                for ev in s_["evs"]:
                    code, consts, rots = ev.flatten()
                    terms.append(C.graph_eval(fo, code, ev.num_intermediates, consts, rots, host_cols, chal_m, n))
                want_t = [C.msm_pippenger(c, t, bases[:n]) for t in terms]
                want_wf = C.fold_witness(fo, w1, w2, s_["r"])
                want_e = C.fold_error(fo, lib.download(s_["d_e"], (n, 4)), terms, s_["r"])
                o = last[c]
                ok &= bool((o["w_commit"] == want_w).all() and all((o["t_commits"][i] == want_t[i]).all() for i in range(s_["cnt"])))
                ok &= bool((lib.download(s_["d_wout"], (s_["nw"], 4)) == want_wf).all() and (lib.download(o["d_e_new"], (n, 4)) == want_e).all())
            cpu_ms = (time.perf_counter() - t0) * 1e3
            ex["nifs_fold_step_k17"].update({"cpu_ms": round(cpu_ms, 1), "cpu_cores": C.num_threads(), "cpu_kind": "port (commits, evaluation and folds of the oracle, incl. downloads)",
                                             "bit_exact_commits_terms_folds": ok})
        for c, s_ in st.items():
            for p in (s_["d_w1"], s_["d_w2"], s_["d_fix"], s_["d_e"], s_["d_terms"], s_["d_wout"], s_["d_enew"]):
                lib.free(p)
            s_["key"].close()
    except Exception as e:
        import traceback
        ex["nifs_fold_step_k17"] = {"error": repr(e), "trace": traceback.format_exc()[-600:]}

    # ---- specialised cross-term kernels, cold and cached (mira_graph_set_cache_dir): both circuits of the fold step specialised in a
    # fresh process that compiles every kernel, then in a second fresh process that finds the code objects in the directory
    try:
        import subprocess
        import tempfile
        probe = os.path.join(ROOT, "tools", "jit_cache_probe.py")
        legs = {}
        with tempfile.TemporaryDirectory() as cache_dir:
            os.chmod(cache_dir, 0o700)
            for leg in ("cold", "cached"):
                r = subprocess.run([sys.executable, probe, cache_dir], capture_output=True, text=True, timeout=600)
                legs[leg] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else {"error": r.stderr[-400:]}
            legs["files"] = len(os.listdir(cache_dir))
        for leg in ("cold", "cached"):
            if "error" not in legs[leg]:
                legs[leg + "_specialize_s"] = round(sum(v["seconds"] for v in legs[leg].values()), 3)
        ex["specialize_cold_and_cached"] = legs
    except Exception as e:
        ex["specialize_cold_and_cached"] = {"error": repr(e)}

    # ---- commitment-key cache file -> HBM -------------------------------------------
    # load_or_setup_cache (src/commitment.rs:134-166) = load_from_file + is_on_curve over every point; the only timing the
    # reference holds for it: 3.47 s for 2^27 bn256 points (the `get_or_create_commitment_key` span of the log fixture in
    # .scripts/build_profiling.py:213).  mira_msm_register_bases_file reads the 8 GiB file in 64 MiB chunks by several
    # threads into pinned memory, the copy and the read of chunk i + 1 beside the conversion + curve check of chunk i.
    # The file is a valid key of 2^22 generated points written 32 times over (the loader cannot tell).
    if not skip_key_load:
        try:
            import shutil
            import tempfile
            k = 27
            tmpdir = tempfile.mkdtemp(prefix="mira_key_")
            try:
                while k > 22 and shutil.disk_usage(tmpdir).free < (64 << k) + (2 << 30):
                    k -= 1                                             # what fits the scratch disk, noted in the result
                n, nb = 1 << k, 1 << 22
                base = cm.CommitmentKey.synthetic(cm.CURVE_BN256, nb, seed=0x4B4559)
                block = base.download().tobytes()
                path = os.path.join(tmpdir, f"{k}.bin")
                t0 = time.perf_counter()
                with open(path, "wb") as f:
                    for _ in range(n // nb):
                        f.write(block)
                    f.flush(); os.fsync(f.fileno())
                t_write = time.perf_counter() - t0
                res = {"points": n, "file_bytes": n * 64, "file_write_s": round(t_write, 2), "threads": min(8, os.cpu_count() or 1),
                       "reference_literal_s": 3.47, "reference_literal_source": ".scripts/build_profiling.py:213 (2^27 bn256 points, load + is_on_curve, unknown hardware)"}

                def load(validate):
                    t0 = time.perf_counter()
                    key = cm.CommitmentKey.load_from_file(cm.CURVE_BN256, path, k, validate=validate)
                    dt = time.perf_counter() - t0
                    return key, dt
                try:                                                   # evict the file from the page cache: a cold read from the scratch disk
                    fd = os.open(path, os.O_RDONLY)
                    os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
                    os.close(fd)
                    key, dt = load(True)
                    res["cold_load_validate_s"] = round(dt, 3)
                    key.close()
                except Exception as e:
                    res["cold_load_validate_s"] = repr(e)
                key, dt_v = load(True)                                 # page-cache warm from here on
                sample = bool((key.download(n - 4096, 4096) == base.download(nb - 4096, 4096)).all())
                key.close()
                key, dt_n = load(False)
                key.close()
                res.update({"load_validate_s": round(dt_v, 3), "load_s": round(dt_n, 3), "GB_per_s_validate": round(n * 64 / dt_v / 1e9, 2),
                            "GB_per_s": round(n * 64 / dt_n / 1e9, 2), "tail_matches_source": sample,
                            "note": "page-cache-warm file -> pinned -> HBM, resident-layout conversion and (validate) the curve check of every point included"})
                base.close()
                ex[f"key_load_2p{k}"] = res
            finally:
                shutil.rmtree(tmpdir, ignore_errors=True)
        except Exception as e:
            import traceback
            ex["key_load_2p27"] = {"error": repr(e), "trace": traceback.format_exc()[-500:]}

    # ---- ProtoGalaxy's weighted tree reduction: compute_F's shape at k = 17
    # with 8 gates -- 2^20 gate evaluations folded for 32 challenges
    try:
        levels, points = 20, 32
        n = 1 << levels
        d = cm.synth_scalars_device(cm.CURVE_BN256, n, seed=0x4000)
        d_w = cm.synth_scalars_device(cm.CURVE_BN256, points * levels, seed=0x4001)
        w = lib.download(d_w, (points * levels, 4))
        lib.free(d_w)
        out = np.zeros((points, 4), dtype=np.uint64)
        call = lambda: lib.check(lib.c.mira_pow_tree_reduce_device(1, ctypes.c_void_p(d), n, 0, w.ctypes.data_as(ctypes.c_void_p), points,
                                                                   out.ctypes.data_as(ctypes.c_void_p)))
        call()
        walls = []
        for _ in range(7):
            t0 = time.perf_counter(); call(); walls.append((time.perf_counter() - t0) * 1e3)
        ex["protogalaxy_tree_2p20x32"] = {"ms": round(sorted(walls)[3], 3), "leaves": n, "challenges": points,
                                          "G_leaf_challenge_pairs_per_s": round(n * points / sorted(walls)[3] / 1e6, 2)}
        if with_cpu:
            from oracle import cref as C
            host = lib.download(d, (n, 4))
            t0 = time.perf_counter(); want = C.pow_tree(1, host, w.reshape(points, levels, 4)); dtc = (time.perf_counter() - t0) * 1e3
            ex["protogalaxy_tree_2p20x32"].update({"cpu_ms": round(dtc, 1), "cpu_cores": min(points, C.num_threads()), "cpu_kind": "port",
                                                   "bit_exact": bool((out == want).all())})
        lib.free(d)
    except Exception as e:
        ex["protogalaxy_tree_2p20x32"] = {"error": repr(e)}
    return ex


if __name__ == "__main__":
    main()
