"""Development probe: the cross terms of the two MainGate<5> circuits at k = 17 -- grouped graphs and CrossTermPlan -- wall times.
usage: [MIRA_PROBE_LIB=tools/_variants/x.so] python tools/cross_term_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib
if os.environ.get("MIRA_PROBE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIRA_PROBE_LIB"])
from mira_amd import commitment as cm
from harness import graph_evaluator as G, main_gate as MG
lib = _lib.load()
n = 1 << 17
out = []
for c, gates, field in ((0, 2, G.FIELD_FR), (1, 1, G.FIELD_FQ)):
    cg, ctx = MG.compressed_circuit(5, gates)
    evs = [G.GraphEvaluator.new(t, field) for t in cg.grouped.iter_from_first()]
    plan = G.CrossTermPlan.from_compressed_gates(cg, ctx, field)
    d_fix = cm.synth_scalars_device(c, ctx.num_fixed * n, seed=1); d_w1 = cm.synth_scalars_device(c, ctx.num_advice * n, seed=2); d_w2 = cm.synth_scalars_device(c, ctx.num_advice * n, seed=3, kind=1)
    chal = [(77 + j) ** 9 % G.MODULUS[field] for j in range(2 * ctx.num_challenges)]
    dom = G.PlonkEvalDomain(ctx.num_advice, 0, chal, [], [d_fix + j * n * 32 for j in range(ctx.num_fixed)], [(d_w1, ctx.num_advice * n)], [(d_w2, ctx.num_advice * n)], n)
    cols = dom.columns()
    d_a, d_b = lib.alloc(len(evs) * n * 32), lib.alloc(len(evs) * n * 32)
    def med(fn):
        fn(); fn(); ts = []
        for _ in range(9):
            t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
        return sorted(ts)[4]
    tg = med(lambda: G.GraphEvaluator.evaluate_batch_device(evs, cols, chal, n, [d_a + i * n * 32 for i in range(len(evs))]))
    tp = med(lambda: plan.evaluate_device(cols, chal, n, d_b))
    same = bool((lib.download(d_a, (len(evs), n, 4)) == lib.download(d_b, (len(evs), n, 4))).all())
    out.append(f"gates {gates}: grouped {tg:.3f} ms, plan {tp:.3f} ms, same {same}")
print(os.environ.get("MIRA_PROBE_LIB", "tree"), " | ".join(out), flush=True)
