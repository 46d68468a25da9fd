timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_sharded_inprocess.py tests/test_gpu_comm_failures.py tests/test_gpu_bench_dist.py -x -q 2>&1 | tail -8
CTK_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03p_world1.json 2>gpurun_out/r03p_world1.err; python -c "
import json; d=json.load(open('gpurun_out/r03p_world1.json')); print(d['ms_per_step'], d.get('collectives_per_step'), d['config'].get('parity'))"
CTK_NO_SPEC_X4=1 CTK_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('nospec', d['ms_per_step'], d.get('collectives_per_step'))"
