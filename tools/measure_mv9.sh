set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O; rm -rf $O/pmc9
cp profiles/traffic.json $O/traffic.json
N9=40000000
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $c -d $O/pmc9/$n -o p --output-format csv -- python3 tools/pmc_run.py $N9 les480_mv9 > $O/pmc9_$n.log 2>&1 || echo "pass $n failed"
done
python3 tools/pmc_parse.py $O/pmc9/* > $O/pmc_summary_les480_mv9.txt
python3 tools/make_traffic.py $O/pmc9 les480_mv9 $N9 $O/traffic.json "round r04, $(date -u +%Y-%m-%dT%H:%MZ), compact images for the ray kernel"
cp $O/traffic.json profiles/traffic.json
timeout -k 10 400 python bench.py --workload les480_mv9 --photons 2e8 --steps 8 --no-cpu-baseline --no-pmc > $O/bench_les480_mv9_n1.json.log 2>/dev/null
tail -1 $O/bench_les480_mv9_n1.json.log | cut -c1-300
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt9 -o mv9 --output-format csv -- python3 tools/pmc_run.py 4e7 les480_mv9 > $O/kt9.log 2>&1 || true
head -4 $O/kt9/mv9_kernel_stats.csv | cut -c1-180
