#!/bin/bash
# Development helper for one gpurun call: GPU parity tests, then the headline bench.
# usage: tools/gpu_check.sh <tag> [bench args]
tag=${1:-x}; shift
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$tag.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_$tag.txt
timeout -k 10 500 python bench.py --steps 5 --warmup 2 "$@" > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/bench_$tag.json"))
print("value", d["value"], d["unit"], "ms/step", d["ms_per_step"])
print("stages", d["stages_ms"])
print("roofline", d["roofline"]["achieved"], d["roofline"]["frac"])
for k,v in d.get("extras",{}).items(): print(k, {a:b for a,b in v.items() if a not in ("note","roofline","cpu_baseline")})
print("parity", d.get("parity"))
PY
