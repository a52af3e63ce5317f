set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_s1; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
timeout -k 10 400 python tools/tile_sweep.py 1e8 les480 0 16 24 32 48 64 -1 > $O/tile_sweep_les480.log 2>&1
cat $O/tile_sweep_les480.log
for c in 0 -1; do
  MI3D_TILE_COLS=$c timeout -k 10 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $O/tcc_$c -o p --output-format csv -- python3 tools/pmc_run.py 1e8 > $O/tcc_$c.log 2>&1
done
python3 tools/pmc_parse.py $O/tcc_0 $O/tcc_-1 | tee $O/tcc_summary.txt
