#!/bin/bash
# On the GPU box: tools/param_table.py rows that match the given words, for the shipped library and every tools/ab/libstereo_ab_*.so.
echo "== shipped"; timeout -k 10 300 python tools/param_table.py "$@" || exit 1
for so in tools/ab/libstereo_ab_*.so; do
  [ -e "$so" ] || continue
  echo "== $so"; STEREO_HIP_LIB=$PWD/$so timeout -k 10 300 python tools/param_table.py "$@" || exit 1
done
