#!/bin/bash
# round 6, GPU call K: where fd_group_dw's time goes (timing-only ablations through FD_GROUP_DW_DEBUG: 1 no flush, 2 no MFMAs, 4 no split / LDS writes)
O=gpurun_out/r06k
mkdir -p $O
for d in 0 1 2 4 3 6 7; do
  echo "FD_GROUP_DW_DEBUG=$d" >> $O/group_dw_ablate.txt
  FD_GROUP_DW_DEBUG=$d timeout 200 python tools/bench_group_dw.py 3840 0 2>/dev/null | head -1 >> $O/group_dw_ablate.txt
  FD_GROUP_DW_DEBUG=$d timeout 200 python tools/bench_group_dw.py 3840 256 2>/dev/null | head -1 >> $O/group_dw_ablate.txt
done
cat $O/group_dw_ablate.txt
