#!/bin/bash
# On the GPU box: 8K kernel times of the median, bilateral and Gaussian filters for every tools/ab/libstereo_ab_*.so, two rounds.
for round in 1 2; do
for so in tools/ab/libstereo_ab_*.so; do
  n=$(basename $so .so); n=${n#libstereo_ab_}
  for f in median bilateral gaussian; do
    STEREO_HIP_LIB=$PWD/$so timeout -k 10 200 python bench.py --workload 8k --filter $f --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/abf_${n}_$f.json 2> gpurun_out/abf_${n}_$f.err || { echo "$n $f failed"; continue; }
    python -c "
import json; d=json.load(open('gpurun_out/abf_${n}_$f.json')); print('$n', '$f', d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done
done
