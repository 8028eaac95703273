#!/bin/bash
# round 6, GPU call F: lone-backbone sampling with the one-launch sequence attention at every size
O=gpurun_out/r06f
mkdir -p $O
for n in 128 256; do
  for r in 1024 0; do
    echo "FD_SEQ_ATTN_MIN_ROWS=$r" >> $O/sample.log
    FD_SEQ_ATTN_MIN_ROWS=$r timeout 300 python tools/sample_probe.py $n 1 2>/dev/null | tail -1 >> $O/sample.log
  done
done
cat $O/sample.log
