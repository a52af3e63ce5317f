#!/bin/bash
# small flux jobs (er3t's pattern: 16 g x 3 runs of a few million photons): per-job time and where it goes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/s45; rm -rf $O; mkdir -p $O
for n in 2e6 6e6 2e7; do AB_WORKLOAD=les128_flux timeout -k 10 120 python tools/small_runs.py $n 24 2>&1 | tail -1; done | tee $O/small_runs.log
AB_WORKLOAD=les128_flux timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kt -o k --output-format csv -- python3 tools/small_runs.py 6e6 24 > $O/kt.log 2>&1
python3 - $O/kt <<'PY' | tee -a $O/small_runs.log
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/k_kernel_stats.csv', recursive=True)[0]
tot = 0
rows = list(csv.DictReader(open(f)))
for row in rows: tot += float(row['TotalDurationNs'])
print('kernel time in all %.1f ms over 27 jobs of 6e6 photons' % (tot/1e6))
for row in rows[:14]:
    print('%-62s calls %4s total %8.2f ms  avg %7.1f us' % (row['Name'][:62], row['Calls'], float(row['TotalDurationNs'])/1e6, float(row['AverageNs'])/1e3))
PY
