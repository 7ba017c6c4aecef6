#!/bin/bash
# tools/ab/build_variant.sh <tag> [-DFLAG=...]...: a copy of the library whose two-column kernel is compiled with
# the given macros (tools/ab/libstereo_ab_<tag>.so; the other objects are the ones build.sh left behind).
set -euo pipefail
tag="$1"; shift
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
PKG="$ROOT/stereo-matching_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" -I"$PKG/csrc" -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-memory-clause -mllvm -amdgpu-schedule-relaxed-occupancy=true "$@" \
  -c "$PKG/csrc/guided_pair.hip" -o "$ROOT/tools/ab/gp_$tag.o"
objs=()
for o in "$PKG"/csrc/obj/*.o; do [[ "$(basename "$o")" == guided_pair.o ]] || objs+=("$o"); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/ab/libstereo_ab_$tag.so" "${objs[@]}" "$ROOT/tools/ab/gp_$tag.o"
echo "built libstereo_ab_$tag.so"
