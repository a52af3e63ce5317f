# Round-end measurements on the GPU box: kernel trace of the bench command, PMC passes of one launch of the bench workload
# (separate --pmc passes, nothing combined with tracing), the bench lines, the other workloads.
#   bash tools/final_measure.sh <round tag, e.g. r03> [A|B]      (two parts, each within one gpurun call; default: both)
set -e
R=${1:-r04}
PART=${2:-AB}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$R; mkdir -p $O
if [[ $PART == *A* ]]; then
# 1. kernel trace + stats of the bench command (5 steps of 1e9 photons)
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/kt -o bench --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $O/bench_under_rocprof.log 2>&1
echo kt done
# 2. PMC passes of one full-size launch (5e8 photons: what one launch of a bench step of 1e9 is)
N=500000000
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE TCC_EA0_ATOMIC_sum"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $c -d $O/pmc/$n -o p --output-format csv -- python3 tools/pmc_run.py $N les480 > $O/pmc_$n.log 2>&1 || echo "pass $n failed"
done
python3 tools/pmc_parse.py $O/pmc/* > $O/pmc_summary_les480.txt
python3 tools/make_traffic.py $O/pmc les480 $N $O/traffic.json "round ${R}, $(date -u +%Y-%m-%dT%H:%MZ), tools/final_measure.sh"
echo pmc done
# the same for the nine-view workload (its own roofline object)
N9=40000000
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $c -d $O/pmc9/$n -o p --output-format csv -- python3 tools/pmc_run.py $N9 les480_mv9 > $O/pmc9_$n.log 2>&1 || echo "pass $n failed"
done
python3 tools/pmc_parse.py $O/pmc9/* > $O/pmc_summary_les480_mv9.txt
python3 tools/make_traffic.py $O/pmc9 les480_mv9 $N9 $O/traffic.json "round ${R}, $(date -u +%Y-%m-%dT%H:%MZ), tools/final_measure.sh"
# ... and for the flux workload (config 3's flux leg): what its tallies cost at the memory side
NF=100000000
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE TCC_EA0_ATOMIC_sum"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $c -d $O/pmcf/$n -o p --output-format csv -- python3 tools/pmc_run.py $NF les128_flux > $O/pmcf_$n.log 2>&1 || echo "pass $n failed"
done
python3 tools/pmc_parse.py $O/pmcf/* > $O/pmc_summary_les128_flux.txt
python3 tools/make_traffic.py $O/pmcf les128_flux $NF $O/traffic.json "round ${R}, $(date -u +%Y-%m-%dT%H:%MZ), tools/final_measure.sh"
cp $O/traffic.json profiles/traffic.json
# 3. the bench lines (with the traffic figures just measured)
timeout -k 10 600 python bench.py > $O/bench_les480_n1.json.log 2> $O/bench_err.log
tail -1 $O/bench_les480_n1.json.log
timeout -k 10 400 python bench.py --workload les480_mv9 --photons 2e8 --steps 8 --no-cpu-baseline --no-pmc > $O/bench_les480_mv9_n1.json.log 2>> $O/bench_err.log
tail -1 $O/bench_les480_mv9_n1.json.log
timeout -k 10 400 python bench.py --workload les128 --photons 1e9 --steps 5 --no-cpu-baseline --no-pmc > $O/bench_les128_n1.json.log 2>> $O/bench_err.log
timeout -k 10 400 python bench.py --workload les128_flux --photons 1e8 --steps 5 --no-cpu-baseline --no-pmc > $O/bench_les128_flux_n1.json.log 2>> $O/bench_err.log || true
timeout -k 10 400 python bench.py --workload les128_aer --photons 1e9 --steps 5 --no-cpu-baseline --no-pmc > $O/bench_les128_aer_n1.json.log 2>> $O/bench_err.log || true
timeout -k 10 300 python bench.py --workload les480_flux --photons 1e8 --steps 5 --no-cpu-baseline --no-pmc > $O/bench_les480_flux_n1.json.log 2>> $O/bench_err.log || true
timeout -k 10 300 python bench.py --workload les128_cam --photons 5e7 --steps 5 --no-cpu-baseline --no-pmc > $O/bench_les128_cam_n1.json.log 2>> $O/bench_err.log || true
echo part A done
fi
if [[ $PART == *B* ]]; then
# strong scaling rehearsal on one GPU: the per-GPU share of config 4 / 5 on eight GPUs (1.25e8 photons per step): what the
# fixed cost of a step (sort, launch tail, fold, zeroing) does to the rate
timeout -k 10 300 python bench.py --gpus 1 --scaling strong --photons 1.25e8 --steps 20 --warmup 2 --no-cpu-baseline > $O/bench_les480_strong_share_of_8.json.log 2>> $O/bench_err.log || true
timeout -k 10 300 python bench.py --gpus 1 --scaling strong --photons 1.25e8 --steps 10 --warmup 2 --workload les480_mv9 --no-cpu-baseline > $O/bench_les480_mv9_strong_share_of_8.json.log 2>> $O/bench_err.log || true
# K16: the HIP path against the deterministic plane-parallel answer, the table behind the assertions
rm -f $O/k16_gpu_matrix.log; K16_LOG=$O/k16_gpu_matrix.log timeout -k 10 600 python -m pytest tests/test_k16.py -m gpu -q > $O/k16_pytest.log 2>&1 || true
# 4. scheduler diagnostics and microbenchmarks whose logs are kept
timeout -k 10 200 python tools/sched_diag.py les480 5e7 > $O/sched_diag_les480.log 2>&1
timeout -k 10 200 python tools/sched_rays.py les480_mv9 1e7 > $O/sched_diag_les480_mv9.log 2>&1 || true
timeout -k 10 200 tools/microbench/atomic_rates > $O/atomic_rates.log 2>&1 || true
{ timeout -k 10 300 python tools/time_dropin.py; MI3D_FUSED_SLOTS=1 timeout -k 10 300 python tools/time_dropin.py; } > $O/dropin_pipeline_config3.log 2>&1 || true
timeout -k 10 300 python tools/time_dropin.py --grid 480 > $O/dropin_pipeline_config4.log 2>&1 || true
timeout -k 10 300 python tools/weight_roulette_sweep.py 4e7 > $O/weight_roulette_sweep_mv9.log 2>&1 || true
# the flux workload: kernel trace (photon loop, record sort, record sum), scheduler diagnostics, the routes side by side, LDS atomic rates
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/ktf -o flux --output-format csv -- python3 tools/pmc_run.py 1e8 les128_flux > $O/ktf.log 2>&1 || true
timeout -k 10 200 python tools/sched_diag.py les128_flux 2e7 > $O/sched_diag_les128_flux.log 2>&1 || true
{ echo "records (default)"; AB_WORKLOAD=les128_flux timeout -k 10 200 python tools/ab.py 1e8 er3t_amd/libmi3drt.so; echo "an atomic per crossing (MI3D_TALLY_LISTS=0)"; MI3D_TALLY_LISTS=0 AB_WORKLOAD=les128_flux timeout -k 10 200 python tools/ab.py 1e8 er3t_amd/libmi3drt.so; echo "general kernel (MI3D_KERNEL=generic)"; MI3D_KERNEL=generic AB_WORKLOAD=les128_flux timeout -k 10 200 python tools/ab.py 1e8 er3t_amd/libmi3drt.so; } > $O/flux_tally_routes.log 2>&1 || true
timeout -k 10 100 tools/microbench/lds_atomic_rates > $O/lds_atomic_rates.log 2>&1 || true
# kernel trace of the nine-view workload
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt9 -o mv9 --output-format csv -- python3 tools/pmc_run.py 4e7 les480_mv9 > $O/kt9.log 2>&1 || true
fi
echo all done
