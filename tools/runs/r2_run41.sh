#!/bin/bash
# host-bound check of the decode step + row-finish switch at the configs[3] shapes
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r41; mkdir -p $O
python3 tools/probe_graph_launch.py 54 320 96 2>&1 | grep engines | tee $O/probe.txt
python3 tools/probe_graph_launch.py 54 16 96 2>&1 | grep engines | tee -a $O/probe.txt
ETD_NO_GRAPH=1 python3 tools/probe_graph_launch.py 54 320 96 2>&1 | grep engines | sed 's/^/eager: /' | tee -a $O/probe.txt
for rf in 0 1; do for c in 512 3500; do ETD_ROWFIN=$rf python3 tools/bench_decoder_streams.py $c 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('rowfin=$rf', d['workload'][:60], d['ms_per_step'], d['roofline']['frac'], d['kernel_ms_per_step'])" | tee -a $O/streams.txt; done; done
