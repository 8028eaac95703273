#!/bin/bash
# round 6, GPU call E: lone-backbone sampling with the key-split one-launch IPA attention from N=128 up (A never in HBM) against the
# launch sequence, kernels per captured step
O=gpurun_out/r06e
mkdir -p $O
for n in 128 256; do
  for cfg in "384 4" "100 4" "100 2" "100 8"; do
    set -- $cfg
    echo "FD_IPA_FLASH_SPLIT_MIN_N=$1 FD_IPA_FLASH_SPLITS=$2" >> $O/sample.log
    FD_IPA_FLASH_SPLIT_MIN_N=$1 FD_IPA_FLASH_SPLITS=$2 timeout 300 python tools/sample_probe.py $n 1 2>/dev/null | tail -1 >> $O/sample.log
  done
  for hp in 2 4; do
    echo "FD_IPA_FLASH_SPLIT_MIN_N=100 FD_IPA_FLASH_SPLITS=4 FD_IPA_FLASH_HPB=$hp" >> $O/sample.log
    FD_IPA_FLASH_SPLIT_MIN_N=100 FD_IPA_FLASH_SPLITS=4 FD_IPA_FLASH_HPB=$hp timeout 300 python tools/sample_probe.py $n 1 2>/dev/null | tail -1 >> $O/sample.log
  done
done
cat $O/sample.log
