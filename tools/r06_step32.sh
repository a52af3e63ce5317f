#!/bin/bash
# headline loop: what does the walk pay per gathered line?  (a) ablation with a second gather per step, (b) the texture path's counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/s32; rm -rf $O; mkdir -p $O
L=tools
echo "== les480 5e8: a second gathered line per voxel step" | tee -a $O/ab.log
timeout -k 10 400 python tools/ab.py 5e8 $L/ab_base.so $L/ab_gather2.so $L/ab_base.so $L/ab_gather2.so 2>&1 | tee -a $O/ab.log
for c in "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TA_BUSY_avr TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM" "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TA_TCP_STATE_READ_sum"; do
  n=$(echo $c | tr " " "_" | cut -c1-30)
  for v in base gather2; do
    MI3D_LIBRARY=$PWD/tools/ab_$v.so timeout -k 10 120 rocprofv3 --pmc $c -d $O/pmc_$v/$n -o p --output-format csv -- python3 tools/pmc_run.py 2e8 les480 > $O/pmc_${v}_$n.log 2>&1 || echo "pass $v $n failed"
  done
done
python3 - $O <<'PY' > $O/pmc_headline_texture_path.txt
import csv, glob, sys, collections, os
for d in sorted(glob.glob(sys.argv[1] + '/pmc_*')):
    if not os.path.isdir(d): continue
    tot = collections.defaultdict(float)
    for f in sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True)):
        for row in csv.DictReader(open(f)):
            if 'k_transport_lean' not in row['Kernel_Name']: continue
            tot[row['Counter_Name']] += float(row['Counter_Value'])
    print('==', os.path.basename(d))
    for c, v in sorted(tot.items()): print('  %-40s %18.0f' % (c, v))
PY
cat $O/pmc_headline_texture_path.txt
