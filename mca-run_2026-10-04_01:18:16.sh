#!/bin/bash

/usr/local/bin/python -m er3t_amd.rtm.mca.mca_exe 1000000 0 /root/repo/tests/golden/ab/c7_allsky/r02.g000.inp.txt /root/repo/tests/golden/ab/c7_allsky/r02.g000.out.bin
/usr/local/bin/python -m er3t_amd.rtm.mca.mca_exe 1000000 0 /root/repo/tests/golden/ab/c7_allsky/r01.g000.inp.txt /root/repo/tests/golden/ab/c7_allsky/r01.g000.out.bin
/usr/local/bin/python -m er3t_amd.rtm.mca.mca_exe 1000000 0 /root/repo/tests/golden/ab/c7_allsky/r00.g000.inp.txt /root/repo/tests/golden/ab/c7_allsky/r00.g000.out.bin
