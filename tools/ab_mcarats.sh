#!/bin/bash
# A/B of this build against a real MCARaTS install on the committed input sets (tests/golden/ab/, 32 x 32 x 20 voxels, three
# runs of 1e6 photons per case, fixed seeds).  NOT run anywhere in this repository's pipeline (MCARaTS is an un-vendored
# third-party Fortran program: docs/source/tutorial/install.rst:39-48 of the reference): it is the one step that turns the
# oracle's "parity unpinned" into a pin.  On a machine that has both:
#
#     export MCARATS_V010_EXE=/path/to/mcarats-0.10.4/src/mcarats      # as er3t expects it (er3t/common.py:10)
#     bash tools/ab_mcarats.sh [outdir]
#
# Both solvers take the reference's command line `<exe> <Nphoton> <solver> <inp> <out>` (er3t/rtm/mca/mca_run.py:101-115); the
# GPU side is `python -m er3t_amd.rtm.mca.mca_exe`.  tools/ab_compare.py then prints, per case and output variable, the domain
# means, their difference in standard errors (three runs each) and the per-pixel z-scores (mean, std, fraction beyond 2 and 3).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$ROOT/gpurun_out/ab_mcarats}
if [ -z "$MCARATS_V010_EXE" ] || [ ! -x "$MCARATS_V010_EXE" ]; then
  echo "MCARATS_V010_EXE is not set to an executable: nothing to compare against" >&2
  exit 2
fi
export PYTHONPATH=$ROOT
N=1000000
for case in c2_nadir c2_slant c3_flux c4_absorb c5_lsrt c6_sea c7_allsky; do
  src=$ROOT/tests/golden/ab/$case
  for side in ref gpu; do
    mkdir -p $OUT/$case/$side
    cp $src/*.bin $src/*.inp.txt $OUT/$case/$side/          # side-file paths in the namelists are relative to the job directory
  done
  for r in 0 1 2; do
    inp=r0$r.g000.inp.txt; out=r0$r.g000.out.bin
    (cd $OUT/$case/ref && "$MCARATS_V010_EXE" $N 0 $OUT/$case/ref/$inp $OUT/$case/ref/$out > mcarats_r$r.log 2>&1)
    (cd $OUT/$case/gpu && python -m er3t_amd.rtm.mca.mca_exe $N 0 $OUT/$case/gpu/$inp $OUT/$case/gpu/$out)
  done
  python $ROOT/tools/ab_compare.py $case $OUT/$case/ref $OUT/$case/gpu
done
