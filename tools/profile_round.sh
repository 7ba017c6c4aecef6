#!/bin/bash
# On the GPU box, from the repository root: every measurement profiles/rNN/ holds, from one build of the library.
#   bash tools/profile_round.sh r03
# Writes gpurun_out/<round>/...; copy what is to be kept into profiles/<round>/ afterwards (tools/collect_profiles.py).
set -o pipefail
R="${1:-r03}"
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out/$R"
mkdir -p "$OUT"
cd "$ROOT"
trace() {  # name, bench args...: rocprofv3 --kernel-trace --stats of a bench command (program after --, no launcher)
  local name="$1"; shift
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/$name" -o "$name" --output-format csv -- \
      python3 "$ROOT/bench.py" --no-cpu-baseline --no-precise "$@" > "$OUT/$name.json" 2> "$OUT/$name.err") || return 1
  echo "trace $name done"
}
# 1. the bench line as the driver runs it (Motorcycle-shaped, whole-frame CPU baselines)
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > "$OUT/bench_line.json" 2> "$OUT/bench_line.err" || exit 1
echo "bench line done"
# 2. kernel-trace summary of the same command (without the CPU legs)
trace pair --steps 20 --warmup 5 || exit 1
# 3. counters of the dominant kernel (five separate passes)
timeout -k 10 600 bash tools/pmc_passes.sh "$R/pmc_pair" > "$OUT/pmc_pair.log" 2>&1 || exit 1
# 4. the N > 1 code path on a one-rank group, with its kernel trace (which stream ran when)
timeout -k 10 200 python bench.py --dist-single --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_single_rank_rccl_overlap.json" 2> "$OUT/dist.err" || exit 1
trace dist_single --dist-single --steps 6 --warmup 2 || exit 1
# 5. the two-call path: bench line, kernel trace, HBM bytes of its kernels
timeout -k 10 300 python bench.py --path two-call --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/two_call_bench_line.json" 2> "$OUT/twocall.err" || exit 1
trace two_call --path two-call --steps 5 --warmup 2 || exit 1
timeout -k 10 600 bash tools/pmc_passes.sh "$R/pmc_two_call" --path two-call > "$OUT/pmc_two_call.log" 2>&1 || exit 1
# 6. BASELINE config 4 (4K, D = 512) and config 5 (8K, census radius 8, each filter), with kernel traces
timeout -k 10 300 python bench.py --workload 4k --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/config4_4k_bench_line.json" 2> "$OUT/4k.err" || exit 1
trace config4_4k --workload 4k --steps 10 --warmup 3 || exit 1
for f in guided median gaussian bilateral; do
  timeout -k 10 300 python bench.py --workload 8k --filter $f --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/config5_${f}_8k_bench_line.json" 2> "$OUT/8k_$f.err" || exit 1
  trace config5_$f --workload 8k --filter $f --steps 3 --warmup 1 || exit 1
done
for f in median bilateral gaussian; do
  timeout -k 10 600 bash tools/pmc_passes.sh "$R/pmc_$f" --workload 8k --filter $f > "$OUT/pmc_$f.log" 2>&1 || exit 1
done
# 7. the precise variant, the per-layer operator loop, the parameter table, the shard floor
for e in 0.005 0.0001; do
  timeout -k 10 200 python bench.py --eps $e --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/eps_$e.json" 2> "$OUT/eps_$e.err" || exit 1
done
timeout -k 10 200 python tools/precompute_loop.py 270 > "$OUT/precompute_loop.txt" 2> "$OUT/precompute_loop.err" || exit 1
timeout -k 10 600 python tools/param_table.py > "$OUT/param_table.txt" 2> "$OUT/param_table.err" || exit 1
timeout -k 10 400 python tools/shard_compute.py > "$OUT/shard_compute.txt" 2> "$OUT/shard_compute.err" || echo "shard_compute failed"
timeout -k 10 400 python tools/shard_compute.py 4k > "$OUT/shard_compute_4k.txt" 2> "$OUT/shard_compute_4k.err" || echo "shard_compute 4k failed"
echo done
