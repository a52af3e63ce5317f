#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/c50; rm -rf $O; mkdir -p $O
export MI3D_OVERLAP_SORT=0 MI3D_OVERLAP_PRE=0
for w in les128_flux les480_flux; do
 for wg in 3 5; do
  for c in "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TA_BUSY_avr TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM"; do
    n=$(echo $c | tr " " "_" | cut -c1-30)
    MI3D_FLUX_GRID_WG=$wg timeout -k 10 120 rocprofv3 --pmc $c -d $O/pmc_${w}_wg$wg/$n -o p --output-format csv -- python3 tools/pmc_run.py 5e7 $w > $O/pmc_${w}_wg${wg}_$n.log 2>&1 || echo "pass $w $wg $n failed"
  done
 done
done
python3 - $O <<'PY' > $O/pmc_flux_loop_by_wg_final.txt
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
    if tot.get('TCP_TCC_READ_REQ_sum'): print('  -> L2 read latency %.0f cycles, loop %.1f M cycles per TCP, VALU busy %.2f, waves parked %.2f' % (tot['TCP_TCC_READ_REQ_LATENCY_sum']/tot['TCP_TCC_READ_REQ_sum'], tot['TCP_GATE_EN1_sum']/256e6, 4*tot['SQ_ACTIVE_INST_VALU']/tot['SQ_BUSY_CYCLES']/45.6, tot['SQ_WAIT_ANY']/tot['SQ_WAVE_CYCLES']))
PY
grep "==\|->" $O/pmc_flux_loop_by_wg_final.txt
