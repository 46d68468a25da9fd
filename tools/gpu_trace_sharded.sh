# tools/gpu_trace_sharded.sh <tag>: kernel timeline of one pass of the time-shard path at one rank (CTK_FORCE_DIST=1) -> gpurun_out/<tag>/
TAG=${1:-shtrace}; ROOT=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $ROOT/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
CTK_FORCE_DIST=1 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/$TAG/tr -o r -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-parity-check > $ROOT/gpurun_out/$TAG/trace.log 2>&1
cd $ROOT; python tools/timeline.py gpurun_out/$TAG/tr -2 > gpurun_out/$TAG/timeline_sharded.txt; cat gpurun_out/$TAG/timeline_sharded.txt
rm -rf gpurun_out/$TAG/tr
