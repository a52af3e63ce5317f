# round 2: issue-rate and gather-ceiling microbenchmarks, and the FETCH_SIZE calibration on 16-byte per-lane gathers
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_micro; mkdir -p $O
timeout -k 10 120 tools/microbench/valu_rates > $O/valu_rates.log 2>&1
echo valu done
timeout -k 10 200 tools/microbench/gather_ceiling > $O/gather_ceiling.log 2>&1
echo gather done
# calibration: one configuration per table size, known number of 16-B gathers; kernel-trace apart from the counters
for lg in 17 22 25; do
  timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE -d $O/cal_fetch_$lg -o p --output-format csv -- tools/microbench/gather_ceiling $lg 5 2000 > $O/cal_fetch_$lg.log 2>&1
  timeout -k 10 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $O/cal_tcc_$lg -o p --output-format csv -- tools/microbench/gather_ceiling $lg 5 2000 > $O/cal_tcc_$lg.log 2>&1
done
echo cal done
python3 tools/pmc_parse.py $O/cal_fetch_17 $O/cal_tcc_17 $O/cal_fetch_22 $O/cal_tcc_22 $O/cal_fetch_25 $O/cal_tcc_25 > $O/cal_summary.txt 2>&1 || true
cat $O/valu_rates.log; tail -8 $O/gather_ceiling.log; cat $O/cal_summary.txt
