set -x
mkdir -p gpurun_out/r03a
timeout 1500 python -m pytest tests/test_gpu_comm_failures.py tests/test_gpu_bench_dist.py tests/test_gpu_sharded.py -x -q 2>&1 | tail -25 > gpurun_out/r03a/tests_comm.log
tail -5 gpurun_out/r03a/tests_comm.log
CTK_FORCE_DIST=1 timeout 300 python bench.py --steps 10 --warmup 3 > gpurun_out/r03a/bench_rccl_world1.json 2> gpurun_out/r03a/bench_rccl_world1.err; tail -c 600 gpurun_out/r03a/bench_rccl_world1.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03a/bench_1deg.json 2> gpurun_out/r03a/bench_1deg.err; head -c 400 gpurun_out/r03a/bench_1deg.json
