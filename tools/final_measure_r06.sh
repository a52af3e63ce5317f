# Round-6 evidence run on the GPU box: bash tools/final_measure_r06.sh [A|B|C]   (each part within one gpurun call)
PART=${1:-A}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/final; mkdir -p $O
if [[ $PART == *A* ]]; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o bench --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pmc --no-secondary > $O/bench_under_rocprof.log 2>&1
echo kt done
timeout -k 10 700 python bench.py > $O/bench_les480_driver_command.json.log 2> $O/bench_err.log
tail -c 400 $O/bench_les480_driver_command.json.log; echo
echo part A done
fi
if [[ $PART == *B* ]]; then
timeout -k 10 600 python tools/make_traffic_live.py $O/traffic.json "round r06" les480:5e8 les480_mv9:4e7 les128_flux:1e8 les480_flux:5e7 les128_mie:2e8 les128:2e8 > $O/traffic.log 2>&1
echo traffic done
for w in "les128_flux 1e8 8" "les480_flux 5e7 8" "les480_mv9 2e8 8" "les128_mie 1e9 5"; do
  set -- $w
  timeout -k 10 300 python bench.py --workload $1 --photons $2 --steps $3 --no-cpu-baseline --no-secondary > $O/bench_$1_n1.json.log 2>> $O/bench_err.log || true
done
echo part B done
fi
if [[ $PART == *C* ]]; then
for w in "les128_flux 1e8" "les480_flux 1e8" "les128_mie 2e8" "les480_mv9 1e8"; do
  set -- $w
  MI3D_OVERLAP_SORT=0 MI3D_OVERLAP_PRE=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kt_$1 -o k --output-format csv -- python3 tools/pmc_run.py $2 $1 > $O/kt_$1.log 2>&1 || true
done
timeout -k 10 200 python tools/sched_diag.py les480 5e7 > $O/sched_diag_les480.log 2>&1 || true
timeout -k 10 200 python tools/sched_diag.py les128_flux 5e7 > $O/sched_diag_les128_flux.log 2>&1 || true
timeout -k 10 200 python tools/sched_diag.py les480_flux 5e7 > $O/sched_diag_les480_flux.log 2>&1 || true
timeout -k 10 200 python bench.py --gpus 1 --scaling strong --photons 1.25e8 --steps 20 --warmup 2 --no-cpu-baseline --no-secondary --no-pmc > $O/bench_les480_strong_share_of_8.json.log 2>> $O/bench_err.log || true
{ timeout -k 10 300 python tools/time_dropin.py; } > $O/dropin_pipeline_config3.log 2>&1 || true
timeout -k 10 300 python tools/soak_flux.py 3e8 > $O/soak_flux.log 2>&1 || true
echo part C done
fi
