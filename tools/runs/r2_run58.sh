#!/bin/bash
# pure concurrent stepping (tools/ab_tokens.py) twice: same digests?
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r58; mkdir -p $O
ab() { timeout -k 10 300 python3 tools/ab_tokens.py "$@" 2>&1 | grep "^rep"; }
cmpab() { ab "$@" > $O/a.txt; ab "$@" > $O/b.txt; echo "== $TAG: $(diff $O/a.txt $O/b.txt | grep -c '^<') of $(wc -l < $O/a.txt) digests differ" | tee -a $O/det.txt; }
TAG="E=4 S=54 ctx 320 steps 160" cmpab 54 320 160 4
TAG="E=1 S=54" cmpab 54 320 160 1
TAG="E=4 S=54 ETD_NO_GRAPH" ETD_NO_GRAPH=1 cmpab 54 320 160 4
TAG="E=4 S=54 ETD_NO_ATTN_DOWN" ETD_NO_ATTN_DOWN=1 cmpab 54 320 160 4
TAG="E=4 S=54 ETD_NO_FUSED_STEP" ETD_NO_FUSED_STEP=1 cmpab 54 320 160 4
TAG="E=4 S=54 ETD_NO_ATTN_DOWN + ETD_NO_FUSED_STEP" ETD_NO_ATTN_DOWN=1 ETD_NO_FUSED_STEP=1 cmpab 54 320 160 4
TAG="E=4 S=32 (one row tile)" cmpab 32 320 160 4
TAG="E=4 S=54 ctx 64" cmpab 54 64 160 4
