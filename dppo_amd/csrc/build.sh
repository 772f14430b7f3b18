#!/bin/bash
# Build libdppo_hip.so for gfx950 in-tree (dppo_amd/lib/).  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
OBJ="$HERE/obj"
LIB="libdppo_hip.so"
EXTRA=""
if [ -n "${DPPO_STAMPS:-}" ]; then  # debug variant with in-kernel phase stamps (tools/fused_bench.py --stamps)
  OBJ="$HERE/obj_stamps"; LIB="libdppo_hip_stamps.so"; EXTRA="-DDPPO_STAMPS"
fi
if [ -n "${DPPO_VARIANT:-}" ]; then  # experiment build: DPPO_VARIANT=name DPPO_CXXFLAGS="-D..." -> lib/libdppo_hip_name.so
  OBJ="$HERE/obj_${DPPO_VARIANT}"; LIB="libdppo_hip_${DPPO_VARIANT}.so"; EXTRA="${DPPO_CXXFLAGS:-}"
fi
mkdir -p "$OUT" "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I$HERE/../../include $EXTRA"
pids=()
build() { # src extra-flags...
  local src="$1"; shift
  local obj="$OBJ/$(basename "${src%.hip}").o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ -n "$(find "$HERE" ../../include -maxdepth 1 -name '*.h' -newer "$obj" 2>/dev/null)" ]; then
    $HIPCC $COMMON "$@" -c "$src" -o "$obj" &
    pids+=($!)
  fi
}
cd "$HERE"
build "$HERE/gemm.hip"
build "$HERE/fused.hip"
build "$HERE/sampler.hip" -ffp-contract=off
build "$HERE/sampler_split.hip" -ffp-contract=off
build "$HERE/ppo.hip" -ffp-contract=off
build "$HERE/gaussian.hip" -ffp-contract=off
build "$HERE/gmm.hip" -ffp-contract=off
build "$HERE/unet.hip" -ffp-contract=off
build "$HERE/vision.hip"
build "$HERE/api.hip"
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/$LIB" "$OBJ"/gemm.o "$OBJ"/fused.o "$OBJ"/sampler.o "$OBJ"/sampler_split.o "$OBJ"/ppo.o "$OBJ"/gaussian.o "$OBJ"/gmm.o "$OBJ"/unet.o "$OBJ"/vision.o "$OBJ"/api.o
echo "built $OUT/$LIB"
