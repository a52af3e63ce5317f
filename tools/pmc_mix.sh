# instruction mix of the transport kernel on the bench scene (separate --pmc passes, kernel trace not combined)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc2 && mkdir -p gpurun_out/pmc2
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $c -d gpurun_out/pmc2/$n -o p --output-format csv -- python3 tools/pmc_run.py 1e8 > gpurun_out/pmc2_$n.log 2>&1
done
python tools/pmc_parse.py gpurun_out/pmc2 > gpurun_out/pmc2_summary.txt
cat gpurun_out/pmc2_summary.txt
