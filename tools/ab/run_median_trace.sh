#!/bin/bash
# On the GPU box: per-kernel times of the 8K uint8 median for every tools/ab/libstereo_ab_*.so (rocprofv3 kernel trace).
ROOT="$PWD"
for so in tools/ab/libstereo_ab_*.so; do
  n=$(basename $so .so); n=${n#libstereo_ab_}
  export STEREO_HIP_LIB=$ROOT/$so
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/mt_$n -o mt --output-format csv -- python3 $ROOT/bench.py --workload 8k --filter median --steps 3 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/mt_$n.json 2> $ROOT/gpurun_out/mt_$n.err) || { echo "$n failed"; exit 1; }
  echo "== $n"; f=$(find $ROOT/gpurun_out/mt_$n -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 "$f" | head -5
done
