#!/bin/bash
# Build libstereo_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
OUT="$HERE/libstereo_hip.so"
SRCS=("$HERE"/csrc/*.hip)
mkdir -p "$HERE/csrc/obj"
# Build id = hash of every source that goes into the library and of this script (the compiler flags); stereo_hip_version() reports it, so that a counter
# summary under profiles/ can be matched with the library it was taken on.
BUILD_ID="$(cat "$HERE"/csrc/*.hip "$HERE"/csrc/*.hpp "$HERE/../include/stereo_hip.h" "$HERE/build.sh" | sha256sum | cut -c1-12)"
if [[ "$(cat "$HERE/csrc/obj/build_id.txt" 2>/dev/null)" != "$BUILD_ID" ]]; then
  rm -f "$HERE/csrc/obj/api.o"
  echo "$BUILD_ID" > "$HERE/csrc/obj/build_id.txt"
fi
OBJS=()
pids=()
for src in "${SRCS[@]}"; do
  obj="$HERE/csrc/obj/$(basename "${src%.hip}").o"
  OBJS+=("$obj")
  stale=0
  for dep in "$src" "$HERE"/csrc/*.hpp "$HERE/../include/stereo_hip.h" "$HERE/build.sh"; do
    [[ "$dep" -nt "$obj" ]] && stale=1
  done
  if [[ ! -f "$obj" || $stale -eq 1 ]]; then
    # The fused kernels keep two independent columns per lane in scalar registers: packing them into
    # v_pk_* instructions costs register shuffles and issues no faster on gfx950.
    file_flags=""
    case "$(basename "$src")" in
      # (two-column kernel: it runs two wavefronts per SIMD whatever its register count, so the default scheduler's
      #  care for occupancy buys nothing; same-box runs of library copies: the strategy that clusters memory
      #  operations 0.9 % faster than the default, the one that favours instruction-level parallelism 0.5 %;
      #  relaxed occupancy another 0.55 %.  The post-RA scheduler stays as it is: without it + 5.5 %, bottom-up
      #  + 2.5 %, bidirectional + 6 %)
      guided_pair.hip|guided_pair_codes.hip) file_flags="-fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-memory-clause -mllvm -amdgpu-schedule-relaxed-occupancy=true" ;;
      guided_fast.hip) file_flags="-fno-slp-vectorize" ;;
      api.hip) file_flags="-DSTEREO_BUILD_ID=\"$BUILD_ID\"" ;;
    esac
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$HERE/../include" -I"$HERE/csrc" \
      -Wall -Wno-unused-function $file_flags ${EXTRA_HIPCC_FLAGS:-} -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${OBJS[@]}"
echo "built $OUT"
