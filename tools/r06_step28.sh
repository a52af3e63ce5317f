#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/final; mkdir -p $O
timeout -k 10 300 python bench.py --workload les480_flux --photons 5e7 --steps 8 --no-cpu-baseline --no-secondary > $O/bench_les480_flux_n1.json.log 2>> $O/bench_err.log
MI3D_OVERLAP_SORT=0 MI3D_OVERLAP_PRE=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kt2_les480_flux -o k --output-format csv -- python3 tools/pmc_run.py 1e8 les480_flux > $O/kt2_les480_flux.log 2>&1
timeout -k 10 200 python tools/sched_diag.py les480_flux 5e7 > $O/sched_diag_les480_flux.log 2>&1
