#!/bin/bash
# round 6, GPU call O: where a fd_group_dw launch's fixed cost sits (four-wave form, timing-only ablations, kernel-only timing)
O=gpurun_out/r06o
mkdir -p $O
export FD_GROUP_DW_V1=1
for d in 0 1 2 4 6 7; do
  echo "== FD_GROUP_DW_DEBUG=$d (1 no flush, 2 no MFMAs, 4 no split / LDS writes)" >> $O/ablate.txt
  FD_GROUP_DW_DEBUG=$d timeout 300 python tools/bench_group_dw_items.py 2>/dev/null | head -3 >> $O/ablate.txt
done
cat $O/ablate.txt
