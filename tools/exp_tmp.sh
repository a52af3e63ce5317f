export AB_WORKLOAD=les480_mv9
python tools/ab.py 4e7 tools/ab_base.so tools/ab_vmajor.so tools/ab_base.so tools/ab_vmajor.so
