#!/bin/bash
# round 6, GPU call B: the pre-split-weight GEMM tiles (12-14): parity on the GPU, then the node-level table
O=gpurun_out/r06b
mkdir -p $O
timeout 600 python -m pytest tests/test_gemm_w.py -m gpu -x -q > $O/test_gemm_w.log 2>&1; tail -3 $O/test_gemm_w.log
timeout 600 python tools/bench_node_gemm.py 3840 > $O/node_gemm.log 2>&1
cat $O/node_gemm.log
