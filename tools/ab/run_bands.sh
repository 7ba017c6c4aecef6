#!/bin/bash
# On the GPU box: step and kernel time of the plain one-GPU path, of the band-reporting kernel without a process
# group (--bands N) and of the N > 1 code path on a one-rank group (--dist-single --bands N).
p() { python -c "
import json,sys; d=json.load(open('$1')); print('$2', 'step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'])"; }
for r in 1 2; do
timeout -k 10 120 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-precise > gpurun_out/bands_plain.json 2>gpurun_out/bands.err || exit 1; p gpurun_out/bands_plain.json plain
for b in 2 4; do
  timeout -k 10 120 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --bands $b > gpurun_out/bands_$b.json 2>gpurun_out/bands.err || exit 1; p gpurun_out/bands_$b.json "bands=$b"
done
for b in 1 4; do
  timeout -k 10 120 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --dist-single --bands $b > gpurun_out/bands_d$b.json 2>gpurun_out/bands.err || exit 1; p gpurun_out/bands_d$b.json "dist-single bands=$b"
done
done
