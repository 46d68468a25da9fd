set -x
mkdir -p gpurun_out/r03b
timeout 900 python -m pytest tests/test_gpu_comm_failures.py -x -q 2>&1 | tail -5
for wl in era5_1deg_djf30 era5_025deg_480 era5_025deg_2k; do
CTK_SEAMSTATS=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra --workload $wl 2>&1 | grep -A9 SEAMSTATS | head -12 > gpurun_out/r03b/seamstats_$wl.txt
cat gpurun_out/r03b/seamstats_$wl.txt
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03b/trace -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/gpurun_out/r03b/trace.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/timeline.py gpurun_out/r03b/trace -2 > gpurun_out/r03b/timeline_1deg.txt; cat gpurun_out/r03b/timeline_1deg.txt
