# counters of the flux chain's kernels (photon loop, record sort, sums), separate --pmc passes:  bash tools/pmc_flux.sh <outdir> [workload] [photons]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/pmc_flux}; W=${2:-les128_flux}; N=${3:-1e8}
rm -rf $O && mkdir -p $O
for c in "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_ATOMIC_RETURN SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_ANY" "TCC_EA0_WRREQ_STALL_sum TCC_WRITEBACK_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_REQ_sum"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  timeout -k 10 150 rocprofv3 --pmc $c -d $O/$n -o p --output-format csv -- python3 tools/pmc_run.py $N $W > $O/$n.log 2>&1 || echo "pass $n failed"
done
python3 - $O <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); dur = collections.defaultdict(float)
for f in sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)):
    seen = set()
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'k_tl_' not in k and 'k_transport' not in k: continue
        k = k.split('(')[0].replace('void mi3d::', '').replace('mi3d::', '')
        tot[(k, row['Counter_Name'])] += float(row['Counter_Value'])
        if 'End_Timestamp' in row and (k, row['Dispatch_Id']) not in seen:
            seen.add((k, row['Dispatch_Id'])); dur[(k, f)] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6
for (k, c), v in sorted(tot.items()): print('%-36s %-40s %18.0f' % (k, c, v))
for (k, f), v in sorted(dur.items()): print('%-36s %8.2f ms  %s' % (k, v, f.split('/')[-3] if f.count('/') > 2 else f))
PY
