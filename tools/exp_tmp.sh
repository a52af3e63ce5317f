cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 || exit 1
w=les480_mv9
rocprofv3 --kernel-trace --stats -d gpurun_out/kt_$w -o x --output-format csv -- python3 tools/pmc_run.py 4e7 $w > gpurun_out/kt_$w.log 2>&1
echo $w; cat gpurun_out/kt_$w.log | tail -1; grep -E "k_rays|k_transport_lean" gpurun_out/kt_$w/x_kernel_stats.csv | sed 's/"void mi3d::\([a-z_]*\)<\([^>]*\)>[^"]*"/\1<\2>/' | cut -d, -f1-6
python3 tools/pmc_run.py 4e7 $w
