cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/c61/pmc; rm -rf $O; mkdir -p $O
export MI3D_OVERLAP_SORT=0 MI3D_OVERLAP_PRE=0
for c in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVES" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  timeout -k 10 120 rocprofv3 --pmc $c -d $O/$n -o p --output-format csv -- python3 tools/pmc_run.py 5e7 les480_flux > $O/$n.log 2>&1 || echo "pass $n failed"
done
python3 - $O <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float)
for f in sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'k_transport' not in k: continue
        tot[row['Counter_Name']] += float(row['Counter_Value'])
for c, v in sorted(tot.items()): print('%-44s %18.0f' % (c, v))
PY
