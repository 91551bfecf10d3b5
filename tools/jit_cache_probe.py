"""mira_graph_set_cache_dir on the fold step's real graphs: time of CrossTermPlan.specialize for both MainGate<5> circuits in
a process that compiles every kernel and in a second process that finds them in the directory.
usage: python tools/jit_cache_probe.py            (parent: runs itself twice as a child on a temporary directory)"""
import json, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(cache_dir):
    from mira_amd import _lib, commitment as cm
    from harness import graph_evaluator as G, main_gate as MG
    lib = _lib.load()
    n = 1 << 12
    G.GraphEvaluator.set_jit_cache_dir(cache_dir, lib=lib)
    out = {}
    for c, gates, field in ((cm.CURVE_BN256, 2, G.FIELD_FR), (cm.CURVE_GRUMPKIN, 1, G.FIELD_FQ)):
        cg, ctx = MG.compressed_circuit(5, gates)
        nw = ctx.num_advice * n
        d_w1 = cm.synth_scalars_device(c, nw, seed=1, kind=1); d_w2 = cm.synth_scalars_device(c, nw, seed=2, kind=1)
        d_fix = cm.synth_scalars_device(c, ctx.num_fixed * n, seed=3)
        chal = [(0x1234567 + 977 * j) ** 7 % G.MODULUS[field] for j in range(2 * ctx.num_challenges)]
        dom = G.PlonkEvalDomain(ctx.num_advice, 0, chal, [], [d_fix + j * n * 32 for j in range(ctx.num_fixed)], [(d_w1, nw)], [(d_w2, nw)], n)
        plan = G.CrossTermPlan.from_compressed_gates(cg, ctx, field)
        t0 = time.perf_counter(); ok = plan.specialize(dom.columns(), len(chal), lib=lib); dt = time.perf_counter() - t0
        out["gates_%d" % gates] = {"ok": bool(ok), "seconds": round(dt, 3), "compiled_from_disk": G.GraphEvaluator.jit_stats(lib=lib)}
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        with tempfile.TemporaryDirectory() as d:
            for leg in ("first process (compiles)", "second process (reads the directory)"):
                r = subprocess.run([sys.executable, os.path.abspath(__file__), d], capture_output=True, text=True)
                print(leg, r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-800:], flush=True)
            print("files:", len(os.listdir(d)), "bytes:", sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d)))
