timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "not processes" 2>&1 | grep -E "passed|failed"
for wl in era5_025deg_480 era5_025deg_2k; do for e in "X=1" "X=1"; do
env $e python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extra --workload $wl 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$wl', 'ms', round(d['ms_per_step'],4), 'thr', round(d['kernels_ms']['k_threshold'],4), 'rel', round(d['kernels_ms']['k_relabel'],4))"
done; done
bash tools/gpu_trace.sh r03t era5_025deg_480 > /dev/null; sed -n 1,8p gpurun_out/r03t/timeline_era5_025deg_480.txt
bash tools/gpu_trace.sh r03t era5_025deg_2k > /dev/null; sed -n 1,10p gpurun_out/r03t/timeline_era5_025deg_2k.txt
