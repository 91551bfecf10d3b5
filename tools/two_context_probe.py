"""Development probe: what would two library contexts (a stream + workspaces per curve) buy a fold step?

The commits of the two curves of a k = 17 fold step are independent chains, serialised today by the library's one context.  An upper
bound on what overlapping them can give, measured without touching the library: TWO PROCESSES, one per curve, each running its half
of the step's commits (one witness commit + one batched cross-term commit) in a loop from a common start time, against ONE process
running both halves in turn.  The GPU runs the queues of two processes concurrently, as it would two streams of one process.

usage: python tools/two_context_probe.py [seconds]        (the parent never touches the GPU; it starts the children)
"""
import json, os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = 17


def child(curves, seconds, t_start):
    sys.path.insert(0, ROOT)
    import ctypes
    from mira_amd import _lib, commitment as cm
    lib = _lib.load()
    n = 1 << K
    plan = {cm.CURVE_BN256: (14 << K, 6), cm.CURVE_GRUMPKIN: (7 << K, 5)}
    keys, wit, cross = {}, {}, {}
    for c in curves:
        nw, cnt = plan[c]
        keys[c] = cm.CommitmentKey.synthetic(c, nw, seed=0x464F4C44 + c)
        wit[c] = cm.synth_scalars_device(c, nw, seed=0x1000 + c, kind=1)
        cross[c] = lib.alloc(cnt * n * 32)
        for i in range(cnt):
            lib.check(lib.c.mira_synth_scalars_device(c, n, 0, 0x2000 + 16 * c + i, 0, ctypes.c_void_p(cross[c] + i * n * 32)))

    def half(c):
        nw, cnt = plan[c]
        keys[c].commit_device(wit[c], nw)
        keys[c].commit_batch_device(cross[c], n, cnt)

    for _ in range(30):                                       # planner trials settle, clock warm
        for c in curves:
            half(c)
    while time.time() < t_start:
        pass
    t0 = time.perf_counter()
    per = {c: 0.0 for c in curves}
    steps = 0
    while time.perf_counter() - t0 < seconds:                # a fixed window: both processes are under contention for all of it
        for c in curves:
            t1 = time.perf_counter(); half(c); per[c] += time.perf_counter() - t1
        steps += 1
    wall = time.perf_counter() - t0
    print(json.dumps({"curves": curves, "steps": steps, "ms_per_step": wall / steps * 1e3, "ms_per_half": {str(c): round(v / steps * 1e3, 3) for c, v in per.items()}}), flush=True)


def run_children(groups, reps, lead_s):
    t_start = time.time() + lead_s
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", ",".join(map(str, g)), str(reps), repr(t_start)],
                              stdout=subprocess.PIPE, text=True) for g in groups]
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=280)
        if p.returncode != 0:
            raise SystemExit(f"child failed rc={p.returncode}")
        outs.append(json.loads([l for l in out.splitlines() if l.startswith("{")][-1]))
    return outs


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child([int(x) for x in sys.argv[2].split(",")], float(sys.argv[3]), float(sys.argv[4]))
        sys.exit(0)
    reps = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5       # seconds of the timed window
    for round_ in range(2):
        serial = run_children([[0, 1]], reps, 30.0)[0]
        pair = run_children([[0], [1]], reps, 30.0)
        together = max(p["ms_per_step"] for p in pair)
        print(f"one process, both curves in turn: {serial['ms_per_step']:.3f} ms per step (halves {serial['ms_per_half']})")
        print(f"two processes, one curve each:    {together:.3f} ms per step = the slower half under contention (bn256 {pair[0]['ms_per_step']:.3f}, grumpkin {pair[1]['ms_per_step']:.3f})"
              f"  -> {serial['ms_per_step'] / together:.2f} x", flush=True)
