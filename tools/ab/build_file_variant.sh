#!/bin/bash
# tools/ab/build_file_variant.sh <tag> <source.hip> [flags]...: a copy of the library in which one source file is
# compiled with the given extra flags (tools/ab/libstereo_ab_<tag>.so; the other objects are the ones build.sh left).
set -euo pipefail
tag="$1"; src="$2"; shift 2
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
PKG="$ROOT/stereo-matching_amd"
base="$(basename "${src%.hip}")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" -I"$PKG/csrc" "$@" \
  -c "$PKG/csrc/$base.hip" -o "$ROOT/tools/ab/${base}_$tag.o"
objs=()
for o in "$PKG"/csrc/obj/*.o; do [[ "$(basename "$o")" == "$base.o" ]] || objs+=("$o"); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/ab/libstereo_ab_$tag.so" "${objs[@]}" "$ROOT/tools/ab/${base}_$tag.o"
rm -f "$ROOT/tools/ab/${base}_$tag.o"
echo "built libstereo_ab_$tag.so"
