#!/bin/bash
# On the GPU box: kernel time of every tools/ab/libstereo_ab_*.so, two interleaved rounds.
for round in 1 2; do
for so in tools/ab/libstereo_ab_*.so; do
  n=$(basename $so .so); n=${n#libstereo_ab_}
  STEREO_HIP_LIB=$PWD/$so timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err || { echo "$n failed"; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/ab_$n.json')); print('$n', d['roofline']['kernel_ms'])"
done
done
