#!/bin/bash
# On the GPU box: kernel-trace summary lines of the kernels named by $1 (a grep pattern) for every tools/ab/libstereo_ab_*.so,
# two interleaved rounds.
pat="$1"
ROOT="$PWD"
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
for so in "$ROOT"/tools/ab/libstereo_ab_*.so; do
  n=$(basename $so .so); n=${n#libstereo_ab_}
  rm -rf "$ROOT/gpurun_out/trab_$n"
  STEREO_HIP_LIB=$so timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/trab_$n" -o t --output-format csv -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1 || { echo "$n failed"; continue; }
  find "$ROOT/gpurun_out/trab_$n" -name "*kernel_stats.csv" | xargs grep -h "$pat" | sed "s/^/$n /" | cut -c1-200
done
done
