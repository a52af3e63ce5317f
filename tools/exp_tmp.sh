export AB_WORKLOAD=les480_mv9
python tools/ab.py 2e7 tools/ab_lean2.so tools/ab_lean3.so tools/ab_lean2.so tools/ab_lean3.so
