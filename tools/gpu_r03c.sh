set -x
mkdir -p gpurun_out/r03c
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "not processes" 2>&1 | tail -15
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03c/bench_1deg.json 2> gpurun_out/r03c/bench_1deg.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03c/bench_1deg.json'))
print(d['ms_per_step'], d['config']['n_tracked'], d['roofline']['frac'], d['kernels_ms'], d['workload_stats'])
PY
tail -3 gpurun_out/r03c/bench_1deg.err
CTK_ASYNC=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('sync path', d['ms_per_step'])"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --workload era5_025deg_480 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('025-480', d['ms_per_step'], d['config']['n_tracked'], d['workload_stats'])"
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --workload era5_025deg_2k 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('025-2k', d['ms_per_step'], d['config']['n_tracked'], d['workload_stats'])"
