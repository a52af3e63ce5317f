#!/bin/bash
# Builds A/B variants of libmi3drt.so side by side: tools/build_variants.sh name1="-DFLAG=1 ..." name2="..."  ->  tools/ab_<name>.so
# (git-ignored; they travel to the GPU box with the snapshot; run them with tools/ab.py)
cd "$(dirname "$0")/../er3t_amd/csrc" || exit 1
FLAGS="-O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function"
pids=()
for spec in "$@"; do
  name="${spec%%=*}"; defs="${spec#*=}"
  ( hipcc --offload-arch=gfx950 $FLAGS $defs -shared mi3d_api.hip -o ../../tools/ab_$name.so 2>&1 | grep -v "hip-link" ; echo "built ab_$name.so [$defs]" ) &
  pids+=($!)
  if (( ${#pids[@]} >= 4 )); then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
done
wait
