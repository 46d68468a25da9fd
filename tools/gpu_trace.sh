# tools/gpu_trace.sh <tag> [workload] [pass index]: kernel timeline of one bench pass (rocprofv3 --kernel-trace) -> gpurun_out/<tag>/
TAG=${1:-trace}; WL=${2:-era5_1deg_djf30}; IDX=${3:--2}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $ROOT/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/$TAG/trace_$WL -o r -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --workload $WL > $ROOT/gpurun_out/$TAG/trace_$WL.log 2>&1
cd $ROOT; python tools/timeline.py gpurun_out/$TAG/trace_$WL $IDX > gpurun_out/$TAG/timeline_$WL.txt; cat gpurun_out/$TAG/timeline_$WL.txt
rm -rf gpurun_out/$TAG/trace_$WL
