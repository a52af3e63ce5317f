#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/s39; rm -rf $O; mkdir -p $O
export MI3D_OVERLAP_SORT=0 MI3D_OVERLAP_PRE=0
for v in base keys; do
 for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_HIT_sum"; do
  n=$(echo $c | tr " " "_" | cut -c1-30)
  MI3D_LIBRARY=$PWD/tools/ab_$v.so timeout -k 10 120 rocprofv3 --pmc $c -d $O/pmc_$v/$n -o p --output-format csv -- python3 tools/pmc_run.py 5e7 les480_flux > $O/pmc_${v}_$n.log 2>&1 || echo "pass $v $n failed"
 done
done
python3 - $O <<'PY' > $O/pmc_sort_kernels.txt
import csv, glob, sys, collections, os
for d in sorted(glob.glob(sys.argv[1] + '/pmc_*')):
    if not os.path.isdir(d): continue
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True)):
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name']
            k = 'scatter' if 'k_tl_scatter' in k else 'runs_count' if 'k_tl_runsILb0' in k or 'k_tl_runs<false' in k else 'runs_write' if 'k_tl_runs' in k else None
            if k: tot[k][row['Counter_Name']] += float(row['Counter_Value'])
    print('==', os.path.basename(d))
    for k in sorted(tot):
        print(' ', k, '  '.join('%s=%.4g' % (c, v) for c, v in sorted(tot[k].items())))
PY
cat $O/pmc_sort_kernels.txt
