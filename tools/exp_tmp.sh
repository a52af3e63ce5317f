timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipeline" 2>&1 | tail -8
export AB_WORKLOAD=les480_mv9
python tools/ab.py 1e8 tools/ab_pipe.so
MI3D_PIPELINE=0 python tools/ab.py 1e8 tools/ab_pipe.so
MI3D_PIPE_P=1 MI3D_PIPE_R=5 python tools/ab.py 1e8 tools/ab_pipe.so
MI3D_PIPE_P=2 MI3D_PIPE_R=3 python tools/ab.py 1e8 tools/ab_pipe.so
