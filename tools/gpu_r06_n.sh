#!/bin/bash
# round 6, GPU call N: fd_group_dw per item shape, the three kernel forms
O=gpurun_out/r06n
mkdir -p $O
for v in lock nolock v1; do
  unset FD_GROUP_DW_V1 FD_GROUP_DW_LOCKSTEP
  if [ $v = v1 ]; then export FD_GROUP_DW_V1=1; fi
  if [ $v = nolock ]; then export FD_GROUP_DW_LOCKSTEP=0; fi
  echo "== $v" >> $O/items.txt
  timeout 300 python tools/bench_group_dw_items.py 2>/dev/null >> $O/items.txt
done
cat $O/items.txt
