# instruction mix and wait counters of the transport kernel (separate --pmc passes, no tracing combined)
#   bash tools/pmc_mix.sh <outdir> [workload] [photons]      (environment: MI3D_KERNEL, MI3D_TILE_COLS select the build/order)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/pmc_mix}; W=${2:-les480}; N=${3:-1e8}
rm -rf $O && mkdir -p $O
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_ATOMIC_sum GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $c -d $O/$n -o p --output-format csv -- python3 tools/pmc_run.py $N $W > $O/$n.log 2>&1 || echo "pass $n failed"
done
python3 tools/pmc_parse.py $O/* > $O/summary.txt
cat $O/summary.txt
