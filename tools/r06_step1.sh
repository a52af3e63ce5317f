#!/bin/bash
# round 6, first GPU call: the hand-over test of ADVICE r5, today's flux baselines, and the TCP / TA / TD counters of the flux loop at 2 / 3 / 4
# workgroups per CU that VERDICT r5 item 1(a) asks for
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/c1; rm -rf $O; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pre_pass_beside or record_sort_beside or flux_tally_routes or flux_parity_les" > $O/pytest_subset.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_subset.log
for w in les128_flux les480_flux; do
  timeout -k 10 300 python bench.py --workload $w --steps 8 --no-pmc --no-cpu-baseline > $O/bench_$w.log 2>&1; echo "bench $w rc $?"
  python - $O/bench_$w.log <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{'):
        d=json.loads(ln); print(d['config']['workload'][:40], '%.4g photons/s' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'], d['roofline']['per_photon'])
PY
done
export MI3D_OVERLAP_SORT=0 MI3D_OVERLAP_PRE=0
for w in les480_flux les128_flux; do
 for wg in 2 3 4; do
  for c in "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TA_BUSY_avr TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM"; do
    n=$(echo $c | tr " " "_" | cut -c1-30)
    MI3D_FLUX_GRID_WG=$wg timeout -k 10 120 rocprofv3 --pmc $c -d $O/pmc_${w}_wg$wg/$n -o p --output-format csv -- python3 tools/pmc_run.py 5e7 $w > $O/pmc_${w}_wg${wg}_$n.log 2>&1 || echo "pass $w $wg $n failed"
  done
 done
done
python3 - $O <<'PY' > $O/pmc_flux_loop_by_wg.txt
import csv, glob, sys, collections, os
for d in sorted(glob.glob(sys.argv[1] + '/pmc_*_wg*')):
    if not os.path.isdir(d): continue
    tot = collections.defaultdict(float)
    for f in sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True)):
        for row in csv.DictReader(open(f)):
            if 'k_transport_flux' not in row['Kernel_Name']: continue
            tot[row['Counter_Name']] += float(row['Counter_Value'])
    print('==', os.path.basename(d))
    for c, v in sorted(tot.items()): print('  %-40s %18.0f' % (c, v))
PY
cat $O/pmc_flux_loop_by_wg.txt | head -80
grep -h photons $O/pmc_*_SQ_WAVE*.log | head -12
