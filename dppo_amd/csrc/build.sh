#!/bin/bash
# Build libdppo_hip.so for gfx950 in-tree (dppo_amd/lib/).  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT" "$HERE/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I$HERE/../../include"
pids=()
build() { # src extra-flags...
  local src="$1"; shift
  local obj="$HERE/obj/$(basename "${src%.hip}").o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ -n "$(find "$HERE" ../../include -maxdepth 1 -name '*.h' -newer "$obj" 2>/dev/null)" ]; then
    $HIPCC $COMMON "$@" -c "$src" -o "$obj" &
    pids+=($!)
  fi
}
cd "$HERE"
build "$HERE/gemm.hip"
build "$HERE/fused.hip"
build "$HERE/sampler.hip" -ffp-contract=off
build "$HERE/ppo.hip" -ffp-contract=off
build "$HERE/api.hip"
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libdppo_hip.so" "$HERE"/obj/gemm.o "$HERE"/obj/fused.o "$HERE"/obj/sampler.o "$HERE"/obj/ppo.o "$HERE"/obj/api.o
echo "built $OUT/libdppo_hip.so"
