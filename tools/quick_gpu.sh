#!/bin/bash
# Quick GPU check used while tuning kernels: the parity tests of the fused path, then the bench line.
# usage (on the GPU box, from the repository root): bash tools/quick_gpu.sh <tag> [pytest -k expression]
tag="$1"; sel="${2:-}"
if [ -n "$sel" ]; then
  timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "$sel" > gpurun_out/q_${tag}_suite.log 2>&1
else
  timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/q_${tag}_suite.log 2>&1
fi
rc=$?
tail -3 gpurun_out/q_${tag}_suite.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/q_${tag}_bench.json 2> gpurun_out/q_${tag}_bench.err || exit $?
python - <<PY
import json
d = json.load(open("gpurun_out/q_${tag}_bench.json"))
print("MDE/s", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"])
PY
