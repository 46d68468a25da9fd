timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "not processes" 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5
for e in "CTK_OVL_W=0" "CTK_OVL_W=5" "CTK_OVL_W=6" "CTK_OVL_W=0" "CTK_OVL_W=5" "CTK_OVL_W=6"; do
env $e python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extra --workload era5_1deg_djf30 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$e', 'ms', round(d['ms_per_step'],4), 'thr', round(d['kernels_ms']['k_threshold'],4), 'rel', round(d['kernels_ms']['k_relabel'],4), 'mid', round(d['ms_per_step']-d['kernels_ms']['k_threshold']-d['kernels_ms']['k_relabel'],4))"
done
bash tools/gpu_trace.sh r03y era5_1deg_djf30 > /dev/null; grep "overlap" gpurun_out/r03y/timeline_era5_1deg_djf30.txt
CTK_OVL_W=5 bash tools/gpu_trace.sh r03y5 era5_1deg_djf30 > /dev/null; grep "overlap" gpurun_out/r03y5/timeline_era5_1deg_djf30.txt
CTK_OVL_W=6 bash tools/gpu_trace.sh r03y6 era5_1deg_djf30 > /dev/null; grep "overlap" gpurun_out/r03y6/timeline_era5_1deg_djf30.txt
