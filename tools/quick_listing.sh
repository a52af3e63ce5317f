#!/bin/bash
# a listing of the headline kernel alone in ~10 s (instead of the library's 50): tools/quick_listing.sh [extra flags] -> /tmp/quick.s, block A printed
cd "$(dirname "$0")/../er3t_amd/csrc" || exit 1
printf '#include "mi3d_kernels.hip"\n#include "mi3d_kernel_lean.hip"\n' > /tmp/quick_tu.hip
cp /tmp/quick_tu.hip ./_quick_tu.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DMI3D_ONLY_HEADLINE -DMI3D_MARKS "$@" -S --cuda-device-only _quick_tu.hip -o /tmp/quick.s 2>&1 | grep -i "error"
rm -f ./_quick_tu.hip
cd ../.. && python tools/isa_cost.py /tmp/quick.s k_transport_leanILb0ELb0ELi0ELi0ELi256E | head -4
