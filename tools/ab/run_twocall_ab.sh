#!/bin/bash
# On the GPU box: kernel times of the two-call path (encoder, search on cost codes) for the library and every
# tools/ab/libstereo_ab_*.so, two rounds.
for round in 1 2; do
for so in stereo-matching_amd/libstereo_hip.so tools/ab/libstereo_ab_*.so; do
  n=$(basename $so .so); n=${n#libstereo_ab_}
  STEREO_HIP_LIB=$PWD/$so timeout -k 10 200 python bench.py --path two-call --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/abt_${n}.json 2> gpurun_out/abt_${n}.err || { echo "$n failed"; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/abt_${n}.json'))['two_call']; print('$n', 'encode', d['encode_kernel_ms'], 'search', d['search_kernel_ms'])"
done
done
