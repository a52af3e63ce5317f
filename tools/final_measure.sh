set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01b
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r01b/kt -o bench --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r01b/bench_under_rocprof.log 2>&1
echo kt done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r01b/fetch -o p --output-format csv -- python3 bench.py --steps 4 --warmup 0 --no-cpu-baseline > gpurun_out/r01b/pmc_fetch.log 2>&1
echo fetch done
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r01b/write -o p --output-format csv -- python3 bench.py --steps 4 --warmup 0 --no-cpu-baseline > gpurun_out/r01b/pmc_write.log 2>&1
echo write done
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU -d gpurun_out/r01b/valu -o p --output-format csv -- python3 bench.py --steps 4 --warmup 0 --no-cpu-baseline > gpurun_out/r01b/pmc_valu.log 2>&1
echo valu done
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d gpurun_out/r01b/tcc -o p --output-format csv -- python3 bench.py --steps 4 --warmup 0 --no-cpu-baseline > gpurun_out/r01b/pmc_tcc.log 2>&1
echo tcc done
python tools/make_traffic.py gpurun_out/r01b/fetch gpurun_out/r01b/write les480 1e8 gpurun_out/r01b/traffic.json gpurun_out/r01b/valu
cp gpurun_out/r01b/traffic.json profiles/traffic.json
python tools/pmc_parse.py gpurun_out/r01b/valu gpurun_out/r01b/tcc > gpurun_out/r01b/pmc_summary.txt
timeout -k 10 400 python bench.py > gpurun_out/r01b/bench_les480_n1.json.log 2> gpurun_out/r01b/bench_err.log
tail -1 gpurun_out/r01b/bench_les480_n1.json.log
timeout -k 10 300 python tools/time_workloads.py > gpurun_out/r01b/workloads.log 2>&1
cat gpurun_out/r01b/workloads.log
