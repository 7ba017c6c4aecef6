#!/bin/bash
# Counter passes over a short bench run (one rocprofv3 process per pass, as the pool requires).
# Usage on the GPU box, from the repo root:  tools/pmc_passes.sh <out-dir-under-gpurun_out> [bench args]
set -eo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out/$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name="$1"; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d "$OUT/$name" -o "$name" --output-format csv -- \
      python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-precise "${BENCH_ARGS[@]}" > "$OUT/$name.log" 2>&1
  echo "pass $name done"
}
BENCH_ARGS=("$@")
run sq1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_INST_CYCLES_VMEM_RD
run grbm GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
VERSION="$(cd "$ROOT" && python3 -c 'import sys; sys.path.insert(0, "stereo-matching_amd"); import _native; print(_native.lib().stereo_hip_version().decode())')"
python3 "$ROOT/tools/pmc_summary.py" --library-version "$VERSION" "$OUT/sq1" "$OUT/sq2" "$OUT/grbm" "$OUT/fetch" "$OUT/write" > "$OUT/pmc_per_launch.json"
echo "summary in $OUT/pmc_per_launch.json"
