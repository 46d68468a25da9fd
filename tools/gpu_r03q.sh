ROOT=$PWD; mkdir -p gpurun_out/r03q
cd /tmp && export TMPDIR=/tmp
CTK_FORCE_DIST=1 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/r03q/tr -o r -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-parity-check > $ROOT/gpurun_out/r03q/trace.log 2>&1
cd $ROOT; python tools/timeline.py gpurun_out/r03q/tr -2 > gpurun_out/r03q/timeline_sharded.txt; cat gpurun_out/r03q/timeline_sharded.txt
rm -rf gpurun_out/r03q/tr
CTK_HOSTPROF=1 CTK_FORCE_DIST=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra --no-parity-check 2>&1 | grep HOSTPROF | tail -1
