# VALU / SALU / wait counters of one launch of a workload (separate --pmc passes, nothing combined with tracing):
#   bash tools/pmc_quick.sh <out dir under gpurun_out> [photons] [workload]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; N=${2:-200000000}; W=${3:-les480}; mkdir -p $O
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $c -d $O/pmc/$n -o p --output-format csv -- python3 tools/pmc_run.py $N $W > $O/pmc_$n.log 2>&1 || echo "pass $n failed"
done
python3 tools/pmc_parse.py $O/pmc/* > $O/pmc_summary.txt
cat $O/pmc_summary.txt
