"""GPU parity of the run-time specialised cross-term kernels (mira_graph_specialize: a compiled graph's instruction stream
written out as straight-line HIP and compiled for the device with hiprtc) against the interpreter (k_graph_eval) and the
oracle: random gate-like graphs with selectors, rotations and challenges, both fields; the evaluation points of the
MainGate<5> circuits at 2^17 rows; the fallbacks (other column kinds than the kernel was built for, a mixed batch)."""
import random

import numpy as np
import pytest

from graph_cases import MODS, gate_like_expression, oracle_columns, random_expression
from helpers import ints_to_mont
from mira_amd import commitment as cm
from harness import graph_evaluator as G
from harness import main_gate as MG
from oracle import cref as C
from test_gpu_graph import device_columns, synth_data

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("field,log_rows,seed", [(1, 12, 11), (0, 15, 12)])
def test_specialised_random_graphs(gpu_lib, field, log_rows, seed):
    mod, n = MODS[field], 1 << log_rows
    arrs = synth_data(field, n, 2, 3, 7, 3, seed)
    ptrs, cols = device_columns(gpu_lib, arrs)
    rng = random.Random(seed)
    chal = ints_to_mont(arrs["challenges"], mod)
    try:
        evs, wants = [], []
        for k, nterms in enumerate((1, 5, 24, 9)):
            e = gate_like_expression(rng, nterms, 7, 12, 3) if k < 3 else random_expression(rng, 6, 12, 3)
            ge = G.GraphEvaluator.new(e, field)
            code, consts, rots = ge.flatten()
            d = ge.evaluate_device(cols, arrs["challenges"], n, lib=gpu_lib)            # interpreted
            got = gpu_lib.download(d, (n, 4)); gpu_lib.free(d)
            want = C.graph_eval(field, code, ge.num_intermediates, consts, rots, oracle_columns(arrs), chal, n)
            assert (got == want).all()
            assert not ge.is_specialized(len(arrs["challenges"]), len(cols), lib=gpu_lib)
            evs.append(ge); wants.append(want)
        src = evs[1].jit_source(cols, len(arrs["challenges"]), lib=gpu_lib)
        assert "mira_jit_eval" in src and "jit_bool(c" in src
        assert G.GraphEvaluator.specialize(evs[:3], cols, len(arrs["challenges"]), lib=gpu_lib), gpu_lib.c.mira_last_error()
        assert all(ev.is_specialized(len(arrs["challenges"]), len(cols), lib=gpu_lib) for ev in evs[:3])
        assert not evs[3].is_specialized(len(arrs["challenges"]), len(cols), lib=gpu_lib)
        for ev, want in zip(evs, wants):                                                 # one by one
            d = ev.evaluate_device(cols, arrs["challenges"], n, lib=gpu_lib)
            assert (gpu_lib.download(d, (n, 4)) == want).all()
            gpu_lib.free(d)
        # a batch that mixes specialised and interpreted graphs
        d_all = gpu_lib.alloc(len(evs) * n * 32)
        G.GraphEvaluator.evaluate_batch_device(evs, cols, arrs["challenges"], n, [d_all + k * n * 32 for k in range(len(evs))], lib=gpu_lib)
        got_all = gpu_lib.download(d_all, (len(evs), n, 4)); gpu_lib.free(d_all)
        assert all((got_all[k] == wants[k]).all() for k in range(len(evs)))
        # other challenges, fewer rows: the kernel is specialised on the program and the column kinds only
        chal2 = [(c * 7 + 5) % mod for c in arrs["challenges"]]
        m = n // 2 + 3
        d = evs[2].evaluate_device(cols, chal2, m, lib=gpu_lib)
        code, consts, rots = evs[2].flatten()
        short = dict(arrs, selectors=[s[:m] for s in arrs["selectors"]], fixed=[f[:m] for f in arrs["fixed"]], advice=[a[:m] for a in arrs["advice"]])
        ptrs2, cols2 = device_columns(gpu_lib, short)
        gpu_lib.free(d)
        d = evs[2].evaluate_device(cols2, chal2, m, lib=gpu_lib)
        want2 = C.graph_eval(field, code, evs[2].num_intermediates, consts, rots, oracle_columns(short), ints_to_mont(chal2, mod), m)
        assert (gpu_lib.download(d, (m, 4)) == want2).all()
        gpu_lib.free(d)
        for p in ptrs2:
            gpu_lib.free(p)
    finally:
        for p in ptrs:
            gpu_lib.free(p)


def test_specialised_kernel_is_bypassed_for_other_column_kinds(gpu_lib):
    """A kernel is built for the column kinds it was shown; an evaluation that hands a field column where it was built for a
    selector (or the reverse) must go through the interpreter -- and still be right."""
    field, n, seed = 1, 1 << 10, 21
    mod = MODS[field]
    arrs = synth_data(field, n, 2, 3, 7, 3, seed)
    ptrs, cols = device_columns(gpu_lib, arrs)
    rng = random.Random(seed)
    try:
        ge = G.GraphEvaluator.new(gate_like_expression(rng, 6, 7, 12, 3), field)
        assert G.GraphEvaluator.specialize([ge], cols, 3, lib=gpu_lib)
        code, consts, rots = ge.flatten()
        want = C.graph_eval(field, code, ge.num_intermediates, consts, rots, oracle_columns(arrs), ints_to_mont(arrs["challenges"], mod), n)
        d = ge.evaluate_device(cols, arrs["challenges"], n, lib=gpu_lib)
        assert (gpu_lib.download(d, (n, 4)) == want).all()
        gpu_lib.free(d)
        # the two selectors as field columns holding 0 / 1 in the reference's form: same values, other kinds
        one = ints_to_mont([1], mod)[0]
        as_field = [np.where(s[:, None] != 0, one[None, :], np.zeros(4, dtype=np.uint64)[None, :]).astype(np.uint64) for s in arrs["selectors"]]
        extra = []
        cols_f = list(cols)
        for k, a in enumerate(as_field):
            p = gpu_lib.alloc(a.nbytes); gpu_lib.upload(p, np.ascontiguousarray(a)); extra.append(p)
            cols_f[k] = (p, G.COL_FIELD)
        d = ge.evaluate_device(cols_f, arrs["challenges"], n, lib=gpu_lib)
        assert (gpu_lib.download(d, (n, 4)) == want).all()
        gpu_lib.free(d)
        for p in extra:
            gpu_lib.free(p)
    finally:
        for p in ptrs:
            gpu_lib.free(p)


@pytest.mark.parametrize("gates,field,curve", [(2, G.FIELD_FR, cm.CURVE_BN256), (1, G.FIELD_FQ, cm.CURVE_GRUMPKIN)])
def test_specialised_main_gate_cross_terms(gpu_lib, gates, field, curve):
    """The d cross terms of the MainGate<5> circuits at 2^17 rows (CrossTermPlan: d + 1 evaluation points + interpolation):
    specialised = interpreted, vector for vector; and the interpreted ones are what tests/test_gpu_cross_terms.py pins to the oracle."""
    n = 1 << 17
    cg, ctx = MG.compressed_circuit(5, gates)
    plan = G.CrossTermPlan.from_compressed_gates(cg, ctx, field)
    d_fix = cm.synth_scalars_device(curve, ctx.num_fixed * n, seed=0x4000 + curve)
    d_w1 = cm.synth_scalars_device(curve, ctx.num_advice * n, seed=0x4100 + curve)
    d_w2 = cm.synth_scalars_device(curve, ctx.num_advice * n, seed=0x4200 + curve, kind=1)
    chal = [(0x7654321 + 991 * j) ** 5 % G.MODULUS[field] for j in range(2 * ctx.num_challenges)]
    dom = G.PlonkEvalDomain(ctx.num_advice, 0, chal, [], [d_fix + j * n * 32 for j in range(ctx.num_fixed)],
                            [(d_w1, ctx.num_advice * n)], [(d_w2, ctx.num_advice * n)], n)
    cols = dom.columns()
    d_a, d_b = gpu_lib.alloc(cg.degree * n * 32), gpu_lib.alloc(cg.degree * n * 32)
    try:
        plan.evaluate_device(cols, chal, n, d_a, lib=gpu_lib)
        assert plan.specialize(cols, len(chal), lib=gpu_lib), gpu_lib.c.mira_last_error()
        assert all(ev.is_specialized(len(chal), len(cols), lib=gpu_lib) for ev in plan.evaluators)
        plan.evaluate_device(cols, chal, n, d_b, lib=gpu_lib)
        assert (gpu_lib.download(d_a, (cg.degree, n, 4)) == gpu_lib.download(d_b, (cg.degree, n, 4))).all()
    finally:
        for p in (d_a, d_b, d_fix, d_w1, d_w2):
            gpu_lib.free(p)


def test_code_objects_on_disk(gpu_lib, tmp_path):
    """mira_graph_set_cache_dir: a second PROCESS specialising the same graphs loads the kernels the first one compiled
    (same values); a truncated file and a file of another build environment are misses, not wrong kernels."""
    import json
    import os
    import subprocess
    import sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jit_cache_child.py")

    def run():
        out = subprocess.run([sys.executable, child, str(tmp_path), "1", "31"], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])
    first = run()
    if not first["ok"]:
        pytest.skip("no run-time compiler on this machine")
    assert (first["compiled"], first["from_disk"]) == (2, 0) and all(first["specialised"])
    files = sorted(p for p in os.listdir(tmp_path) if p.startswith("mira_jit_") and p.endswith(".bin"))
    assert len(files) == 2 and not [p for p in os.listdir(tmp_path) if ".tmp" in p]
    head = open(os.path.join(tmp_path, files[0]), "rb").read(400)
    assert head[:8] == b"MIRAJIT2" and b"arch gfx950" in head and b" hip " in head and b" hiprtc " in head and b" headers " in head     # keyed by GPU architecture and ROCm version
    assert all(os.stat(os.path.join(tmp_path, f)).st_mode & 0o077 == 0 for f in files)                                                # written for the user alone
    second = run()
    assert (second["compiled"], second["from_disk"]) == (0, 2) and all(second["specialised"])
    assert second["digest"] == first["digest"]
    # a truncated file, and one whose environment key differs in one byte: both are compiled again and rewritten
    a, b = (os.path.join(tmp_path, f) for f in files)
    blob = open(a, "rb").read()
    open(a, "wb").write(blob[:len(blob) // 2])
    blob_b = bytearray(open(b, "rb").read())
    blob_b[20] ^= 1                                               # inside the environment key (offset 16 .. 16 + its length)
    open(b, "wb").write(bytes(blob_b))
    third = run()
    assert (third["compiled"], third["from_disk"]) == (2, 0) and third["digest"] == first["digest"]
    assert open(a, "rb").read() == blob
    fourth = run()
    assert (fourth["compiled"], fourth["from_disk"]) == (0, 2) and fourth["digest"] == first["digest"]
    # a damaged code object (one byte of the code itself): the hash in the header does not match -> compiled again;
    # a file others may write is not taken either
    blob_a = bytearray(blob)
    blob_a[-100] ^= 0x40
    open(a, "wb").write(bytes(blob_a))
    os.chmod(b, 0o666)
    fifth = run()
    assert (fifth["compiled"], fifth["from_disk"]) == (2, 0) and fifth["digest"] == first["digest"]


def test_library_stays_usable_while_the_compiler_runs(gpu_lib):
    """mira_graph_specialize releases the library's lock for the compilation: another thread's evaluations (of the very graph
    being specialised: interpreted until the kernel is in place) and commits go on meanwhile, and a handle freed during the
    compilation is simply skipped."""
    import threading
    import time
    field, n, seed = 1, 1 << 10, 41
    mod = MODS[field]
    arrs = synth_data(field, n, 2, 3, 7, 3, seed)
    ptrs, cols = device_columns(gpu_lib, arrs)
    rng = random.Random(seed)
    key = cm.CommitmentKey.synthetic(cm.CURVE_BN256, 1 << 12, lib=gpu_lib)
    d_s = cm.synth_scalars_device(cm.CURVE_BN256, 1 << 12, seed=5)
    try:
        ge = G.GraphEvaluator.new(gate_like_expression(rng, 20, 7, 12, 3), field)
        doomed = G.GraphEvaluator.new(gate_like_expression(rng, 4, 6, 12, 3), field)
        code, consts, rots = ge.flatten()
        want = C.graph_eval(field, code, ge.num_intermediates, consts, rots, oracle_columns(arrs), ints_to_mont(arrs["challenges"], mod), n)
        want_point = key.commit_device(d_s, 1 << 12)
        nchal = len(arrs["challenges"])
        doomed.compiled(nchal, len(cols), gpu_lib)
        ge.compiled(nchal, len(cols), gpu_lib)
        result = {}

        def compile_thread():
            t0 = time.perf_counter()
            result["ok"] = G.GraphEvaluator.specialize([ge, doomed], cols, nchal, lib=gpu_lib)
            result["seconds"] = time.perf_counter() - t0
        th = threading.Thread(target=compile_thread)
        th.start()
        time.sleep(0.3)                                            # the compiler is running by now (a kernel takes seconds)
        done_meanwhile = 0
        freed = False
        while th.is_alive():
            d = ge.evaluate_device(cols, arrs["challenges"], n, lib=gpu_lib)
            assert (gpu_lib.download(d, (n, 4)) == want).all()
            gpu_lib.free(d)
            assert (key.commit_device(d_s, 1 << 12) == want_point).all()
            if not freed:
                doomed.close()                                     # mira_graph_free(h_doomed) while its kernel is being compiled
                freed = True
            if th.is_alive():
                done_meanwhile += 1
        th.join()
        assert result["ok"], gpu_lib.c.mira_last_error()
        assert done_meanwhile >= 3, (done_meanwhile, result)       # each round is ~1 ms; the compilation takes seconds
        assert ge.is_specialized(nchal, len(cols), lib=gpu_lib)
        d = ge.evaluate_device(cols, arrs["challenges"], n, lib=gpu_lib)
        assert (gpu_lib.download(d, (n, 4)) == want).all()
        gpu_lib.free(d)
    finally:
        key.close(); gpu_lib.free(d_s)
        for p in ptrs:
            gpu_lib.free(p)
