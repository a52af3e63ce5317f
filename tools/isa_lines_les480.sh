#!/bin/bash
# profiles/<round>/isa_lines_les480.txt: static vector / scalar / LDS / memory instructions and issue cost of the headline kernel by SOURCE range
# (the blocks between the MI3D_MARK lines of mi3d_kernel_lean.hip, the inlined device functions of mi3d_device.h), through the line table.
cd "$(dirname "$0")/.." || exit 1
S=er3t_amd/csrc/mi3d_kernel_lean.hip; D=er3t_amd/csrc/mi3d_device.h
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -gline-tables-only -S --cuda-device-only er3t_amd/csrc/mi3d_api.hip -o /tmp/lines.s 2>/dev/null || exit 1
ln() { grep -n "$2" $1 | head -1 | cut -d: -f1; }
A=$(ln $S 'MI3D_MARK("A")'); SCH=$(ln $S 'MI3D_MARK("BSCHED")'); C=$(ln $S 'MI3D_MARK("C")'); B0=$(ln $S 'MI3D_MARK("B0")'); B2=$(ln $S 'MI3D_MARK("B2")')
B5=$(ln $S 'MI3D_MARK("B5")'); B6=$(ln $S 'MI3D_MARK("B6")'); B4=$(ln $S 'MI3D_MARK("B4")'); B7=$(ln $S 'MI3D_MARK("B7")'); E=$(ln $S 'MI3D_MARK("END")')
KE=$(ln $S '^k_entry('); P0=$(ln $D 'inline void philox4x32_10'); P1=$(ln $D 'phase functions (∫P'); F0=$(ln $D 'inline float phase_eval_analytic'); F1=$(ln $D 'Tabulated phase functions in the lean kernels')
H0=$(ln $D 'inline float phase_eval_hg'); H1=$(ln $D 'surface: Ross-Thick')
echo "# k_transport_lean<false,false,0,0,256> ($(git rev-parse --short HEAD)): every instruction attributed to the source line its .loc names (tools/isa_lines.py)"
python tools/isa_lines.py /tmp/lines.s k_transport_leanILb0ELb0ELi0ELi0ELi256E \
  "mi3d_kernel_lean.hip:$A-$((SCH-1))=A: the voxel step" "mi3d_kernel_lean.hip:$SCH-$((C-1))=the pass's schedule" "mi3d_kernel_lean.hip:$C-$((B0-1))=C: a collision the walk has found" \
  "mi3d_kernel_lean.hip:$B0-$((B2-1))=B0: uniform layers" "mi3d_kernel_lean.hip:$B2-$((B5-1))=B2: the rarer events" "mi3d_kernel_lean.hip:$B5-$((B6-1))=B5: finish" \
  "mi3d_kernel_lean.hip:$B6-$((B4-1))=B6: Philox of the rarer events" "mi3d_kernel_lean.hip:$B4-$((B7-1))=B4: next photon, tally window" "mi3d_kernel_lean.hip:$B7-$((E-1))=B7: walk set-up" \
  "mi3d_device.h:$P0-$((P1-1))=Philox block, u01 (mi3d_device.h: inlined into C and B6)" "mi3d_device.h:$F0-$((F1-1))=phase functions (mi3d_device.h: C, B2, B5)" \
  "mi3d_device.h:$H0-$((H1-1))=HG, rotate_dir (mi3d_device.h: C, B5)" "mi3d_kernel_lean.hip:1-$((A-1))=prologue, tally macro, window bookkeeping" "mi3d_kernel_lean.hip:$E-$((KE-1))=epilogue"

