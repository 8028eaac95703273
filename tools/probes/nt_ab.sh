for r in 0 1; do echo "nt $r"; FD_GEMM_W_NT=$r python tools/bench_node_gemm.py 3840 2>/dev/null | grep -E "N= 6816|N=  960|N= 2688|N= 1280|N=  320 K=  320"; done
