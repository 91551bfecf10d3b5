"""Child process of tests/test_gpu_graph_jit.py::test_code_objects_on_disk: specialises two gate-like graphs with the code
object directory set, evaluates them and prints one JSON line: what mira_graph_jit_stats reports and a digest of the values."""
import hashlib
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    cache_dir, field, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    from graph_cases import gate_like_expression
    from mira_amd import _lib
    from harness import graph_evaluator as G
    from test_gpu_graph import device_columns, synth_data
    lib = _lib.load()
    n = 1 << 10
    arrs = synth_data(field, n, 2, 3, 7, 3, seed)
    ptrs, cols = device_columns(lib, arrs)
    rng = random.Random(seed)
    evs = [G.GraphEvaluator.new(gate_like_expression(rng, nterms, 6, 12, 3), field) for nterms in (3, 6)]
    G.GraphEvaluator.set_jit_cache_dir(cache_dir, lib=lib)
    ok = G.GraphEvaluator.specialize(evs, cols, len(arrs["challenges"]), lib=lib)
    compiled, from_disk = G.GraphEvaluator.jit_stats(lib=lib)
    digest = hashlib.sha256()
    for ev in evs:
        d = ev.evaluate_device(cols, arrs["challenges"], n, lib=lib)
        digest.update(lib.download(d, (n, 4)).tobytes())
        lib.free(d)
    specialised = [ev.is_specialized(len(arrs["challenges"]), len(cols), lib=lib) for ev in evs]
    print(json.dumps({"ok": bool(ok), "compiled": compiled, "from_disk": from_disk, "specialised": specialised, "digest": digest.hexdigest()}))


if __name__ == "__main__":
    main()
