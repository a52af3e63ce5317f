# counters of the ray kernel on the nine-view configuration (separate --pmc passes):  bash tools/pmc_rays.sh <outdir> [workload] [photons]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/pmc_rays}; W=${2:-les480_mv9}; N=${3:-4e7}
rm -rf $O && mkdir -p $O
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $c -d $O/$n -o p --output-format csv -- python3 tools/pmc_run.py $N $W > $O/$n.log 2>&1 || echo "pass $n failed"
done
python3 - $O <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); dur = collections.defaultdict(float); nl = collections.defaultdict(int)
for f in sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'k_rays' not in k and 'k_transport_lean' not in k: continue
        k = k.split('(')[0].replace('void mi3d::', '')
        tot[(k, row['Counter_Name'])] += float(row['Counter_Value'])
for (k, c), v in sorted(tot.items()): print('%-36s %-40s %16.0f' % (k, c, v))
PY
