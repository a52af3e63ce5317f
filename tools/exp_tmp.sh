for b in 27 28 29; do echo "batch 2^$b"; MI3D_BATCH_LOG2=$b python tools/tile_sweep.py 1.08e9 les480 -1; done
