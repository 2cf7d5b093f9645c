#!/bin/bash
# which Extract-stage launcher, alone on a second stream, makes the crossed packed add fail in a tiny self-checking kernel (tools/ubench/pk_async_check.hip mode 21)
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r130; mkdir -p $O
for k in none k_embed k_enc_layer k_proj256_kv6 k_proj256_qkv k_proj256_row k_proj256_ln k_attn k_attn_frag k_ffn_fused k_heads k_freq2time; do
  echo "== aggressor: $k" >> $O/pk.txt
  PROBE_PK_ASYNC=21 ETD_EXT_ONLY=$k timeout -k 10 200 python3 tools/probe_pk.py 1.5 432 40 >> $O/pk.txt 2>&1
done
grep "^==\|^beside the aggressor:\|(x) agg" $O/pk.txt | cut -c1-230
