#!/bin/bash
# On the GPU box, from the repository root: every measurement profiles/rNN/ holds, from one build of the library.
#   bash tools/profile_round.sh r02
# Writes gpurun_out/<round>/...; copy what is to be kept into profiles/<round>/ afterwards (tools/collect_profiles.py).
set -o pipefail
R="${1:-r02}"
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out/$R"
mkdir -p "$OUT"
cd "$ROOT"
run() { echo "== $*"; "$@"; }
# 1. the bench line (Motorcycle-shaped, whole-frame CPU baselines)
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > "$OUT/bench_line.json" 2> "$OUT/bench_line.err" || exit 1
# 2. kernel-trace summary of the same command (without the CPU baselines)
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o pair --output-format csv -- \
    python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/pair_bench_under_rocprof.json" 2> "$OUT/trace.err") || exit 1
# 3. counters (five separate passes)
timeout -k 10 600 bash tools/pmc_passes.sh "$R/pmc_pair" > "$OUT/pmc_pair.log" 2>&1 || exit 1
# 4. the other paths and configurations
timeout -k 10 200 python bench.py --dist-single --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_single_rank_rccl_overlap.json" 2> "$OUT/dist.err" || exit 1
timeout -k 10 300 python bench.py --path two-call --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/two_call_bench_line.json" 2> "$OUT/twocall.err" || exit 1
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace2" -o twocall --output-format csv -- \
    python3 "$ROOT/bench.py" --path two-call --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> "$OUT/trace2.err") || exit 1
timeout -k 10 300 python bench.py --workload 4k --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/config4_4k_bench_line.json" 2> "$OUT/4k.err" || exit 1
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace4k" -o k4 --output-format csv -- \
    python3 "$ROOT/bench.py" --workload 4k --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2> "$OUT/trace4k.err") || exit 1
for f in guided median gaussian bilateral; do
  timeout -k 10 300 python bench.py --workload 8k --filter $f --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/config5_${f}_8k_bench_line.json" 2> "$OUT/8k_$f.err" || exit 1
done
timeout -k 10 600 python tools/param_table.py > "$OUT/param_table.txt" 2> "$OUT/param_table.err" || exit 1
echo done
