# roofline.traffic of bench.py: HBM bytes per launch of every kernel of the training step from PMC, calibrated
# (tools/pmc_roofline.py).  Counters in their own runs with --kernel-trace only.   bash tools/pmc_roofline.sh [round tag]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r03}
O=gpurun_out/pmc_roofline
rm -rf $O; mkdir -p $O
[ -x tools/probes/hbm_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probes/hbm_calib.hip -o tools/probes/hbm_calib
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/cf -o p --output-format csv -- tools/probes/hbm_calib > $O/cf.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/cw -o p --output-format csv -- tools/probes/hbm_calib > $O/cw.log 2>&1
CMD="python bench.py --steps 3 --warmup 1 --no-sampling --no-cpu-baseline"
FD_BENCH_PROFILE=1 FD_GRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/sf -o p --output-format csv -- $CMD > $O/sf.log 2>&1
FD_BENCH_PROFILE=1 FD_GRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/sw -o p --output-format csv -- $CMD > $O/sw.log 2>&1
python tools/pmc_roofline.py $O/cf $O/cw $O/sf $O/sw gpurun_out/${TAG}_pmc_traffic.json > gpurun_out/${TAG}_pmc_traffic.txt 2>&1
find $O -name "*.csv" -size +1M -delete
cat gpurun_out/${TAG}_pmc_traffic.txt
