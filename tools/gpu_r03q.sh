timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_sharded_inprocess.py tests/test_gpu_comm_failures.py tests/test_gpu_bench_dist.py -x -q 2>&1 | tail -4
ROOT=$PWD; mkdir -p gpurun_out/r03q
cd /tmp && export TMPDIR=/tmp
CTK_FORCE_DIST=1 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/r03q/tr -o r -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-parity-check > $ROOT/gpurun_out/r03q/trace.log 2>&1
cd $ROOT; python tools/timeline.py gpurun_out/r03q/tr -2 > gpurun_out/r03q/timeline_sharded.txt; tail -32 gpurun_out/r03q/timeline_sharded.txt
rm -rf gpurun_out/r03q/tr
for i in 1 2 3; do CTK_FORCE_DIST=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['config']['collectives_per_step'], d['kernels_ms']['host_seam_driver'])"; done
