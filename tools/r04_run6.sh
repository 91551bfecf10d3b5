set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_msm.py tests/test_gpu_fold_step.py::test_fold_step_k17_schedule tests/test_gpu_graph_jit.py -x -q -m gpu > gpurun_out/r04_j_tests.txt 2>&1 || { tail -30 gpurun_out/r04_j_tests.txt; exit 1; }
tail -1 gpurun_out/r04_j_tests.txt
python bench.py > gpurun_out/r04_j_bench_full.json 2> gpurun_out/r04_j_bench_full.err || { tail -20 gpurun_out/r04_j_bench_full.err; exit 1; }
python - <<'PY'
import json
o=json.load(open('gpurun_out/r04_j_bench_full.json'))
print(o['value'], o['ms_per_step'], o['stages_ms'], o['roofline']['frac'])
e=o['extras']
for k in ('fold_step_k17','nifs_fold_step_k17','msm_sweep_16bit_windows','msm_reference_largest','specialize_cold_and_cached','ntt_2p24','cross_term_eval_k17'):
    v=e.get(k)
    if isinstance(v,dict):
        v={a:b for a,b in v.items() if a not in ('note',) and not isinstance(b,(list,))}
    print(k, json.dumps(v)[:1500])
PY
