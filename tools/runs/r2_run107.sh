#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r107; mkdir -p $O
for k in none k_embed k_enc_layer k_proj256_kv6 k_proj256_qkv k_attn k_post_attn k_proj256_row k_attn_frag k_post_attn_ffn k_freq2time k_heads; do
echo "== extractor launches only: $k" >> $O/race.txt
PROBE_EXT_CAPI=1 ETD_EXT_ONLY=$k timeout -k 10 300 python3 tools/probe_race.py 12 600 extractor >> $O/race.txt 2>&1 || exit 1
done
grep "^==\|^(a\|^(x) agg\|Error\|error" $O/race.txt
