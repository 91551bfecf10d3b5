"""Development probe: the evaluation points of the MainGate<5> cross terms at 2^17 rows, interpreted against specialised."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm
from harness import graph_evaluator as G, main_gate as MG
lib = _lib.load()
if len(sys.argv) > 1: lib.tune(_lib.TUNE_JIT_LOADS_AHEAD, int(sys.argv[1]))
n = 1 << 17
for name, c, gates, field in (("primary_bn256", cm.CURVE_BN256, 2, G.FIELD_FR), ("secondary_grumpkin", cm.CURVE_GRUMPKIN, 1, G.FIELD_FQ)):
    cg, ctx = MG.compressed_circuit(5, gates)
    plan = G.CrossTermPlan.from_compressed_gates(cg, ctx, field)
    d_fix = cm.synth_scalars_device(c, ctx.num_fixed * n, seed=0x3000 + c)
    d_w1 = cm.synth_scalars_device(c, ctx.num_advice * n, seed=0x3100 + c)
    d_w2 = cm.synth_scalars_device(c, ctx.num_advice * n, seed=0x3200 + c, kind=1)
    chal = [(0x1234567 + 977 * j) ** 7 % G.MODULUS[field] for j in range(2 * ctx.num_challenges)]
    dom = G.PlonkEvalDomain(ctx.num_advice, 0, chal, [], [d_fix + j * n * 32 for j in range(ctx.num_fixed)], [(d_w1, ctx.num_advice * n)], [(d_w2, ctx.num_advice * n)], n)
    cols = dom.columns()
    d_out = lib.alloc(cg.degree * n * 32)
    def med():
        plan.evaluate_device(cols, chal, n, d_out)
        ts = []
        for _ in range(9):
            t0 = time.perf_counter(); plan.evaluate_device(cols, chal, n, d_out); ts.append((time.perf_counter() - t0) * 1e3)
        lib.check(lib.c.mira_set_timing(1)); plan.evaluate_device(cols, chal, n, d_out); st = {a: round(b, 3) for a, b in lib.timings()}; lib.check(lib.c.mira_set_timing(0))
        d_p = lib.alloc(len(plan.evaluators) * n * 32)
        outs = [d_p + j * n * 32 for j in range(len(plan.evaluators))]
        G.GraphEvaluator.evaluate_batch_device(plan.evaluators, cols, chal, n, outs)
        lib.check(lib.c.mira_set_timing(1)); G.GraphEvaluator.evaluate_batch_device(plan.evaluators, cols, chal, n, outs); st.update({a: round(b, 3) for a, b in lib.timings()}); lib.check(lib.c.mira_set_timing(0))
        t0 = time.perf_counter()
        for _ in range(5): G.GraphEvaluator.evaluate_batch_device(plan.evaluators, cols, chal, n, outs)
        st["batch_wall"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
        lib.free(d_p)
        return sorted(ts)[4], st
    t_i, st_i = med()
    want = lib.download(d_out, (cg.degree, n, 4))
    t0 = time.perf_counter(); ok = plan.specialize(cols, len(chal)); t_c = time.perf_counter() - t0
    t_j, st_j = med()
    same = bool((lib.download(d_out, (cg.degree, n, 4)) == want).all())
    print(f"{name}: {len(plan.evaluators)} points {plan.num_calculations} interpreted {t_i:.3f} ms {st_i}  specialised {t_j:.3f} ms {st_j}  compile {t_c:.1f} s ok {ok} same {same}", flush=True)
    if not ok: print(lib.c.mira_last_error())
