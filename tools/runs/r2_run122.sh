#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r122; mkdir -p $O
echo "== default build" >> $O/ext.txt
timeout -k 10 300 python3 tools/probe_ext_repro.py 40 >> $O/ext.txt 2>&1
echo "== everything built with -fno-slp-vectorize" >> $O/ext.txt
ETD_LIB_PATH=$PWD/etude_amd/variants/libetude_noslp_all.so timeout -k 10 300 python3 tools/probe_ext_repro.py 40 >> $O/ext.txt 2>&1
grep -v amdgpu.ids $O/ext.txt | cut -c1-400
echo "== bench, default build" >> $O/bench.txt
timeout -k 10 500 python3 bench.py >> $O/bench.txt 2>&1
echo "== bench, -fno-slp-vectorize" >> $O/bench.txt
ETD_LIB_PATH=$PWD/etude_amd/variants/libetude_noslp_all.so timeout -k 10 500 python3 bench.py >> $O/bench.txt 2>&1
grep "^==\|\"metric\"" $O/bench.txt | cut -c1-600
