#!/bin/bash
# round 5, GPU calls H-J: cache policy of the fused edge kernels' training saves.  The shipped library (non-temporal saves) against a
# variant built with -DEM_PLAIN_SAVES (probe build; built here with
#   EDGE_VARIANTS_EXTRA="plain_saves:-DFD_PROBE_BUILD,-DEM_PLAIN_SAVES" python tools/probes/edge_variants.py --build
# and swapped in on the box), alternating: the kernels alone, the training step, N=512 B=8 sampling.  Results: profiles/r05_ab.txt.
# (The round's calls also compared the hint on y / dy_out and on the outputs z' / dz: no further change, not kept.)
export EDGE_VARIANTS_EXTRA="plain_saves:-DFD_PROBE_BUILD,-DEM_PLAIN_SAVES"
for cfg in "30 128" "12 200" "2 512"; do
  set -- $cfg
  timeout 300 python tools/probes/edge_variants.py --rows-b $1 --n $2 2>&1 | grep -v amdgpu.ids
done
cp se3_diffusion_amd/lib/libfd_hip.so /tmp/base.so
run() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"; }
for i in 1 2 3; do
  cp /tmp/base.so se3_diffusion_amd/lib/libfd_hip.so; run shipped
  cp tools/probes/libfd_ev_plain_saves.so se3_diffusion_amd/lib/libfd_hip.so; run plain_saves
done
cp /tmp/base.so se3_diffusion_amd/lib/libfd_hip.so
