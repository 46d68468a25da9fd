bash tools/gpu_trace.sh r03z era5_1deg_djf30 > /dev/null; sed -n 1,6p gpurun_out/r03z/timeline_era5_1deg_djf30.txt
bash tools/gpu_trace.sh r03z era5_025deg_480 > /dev/null; sed -n 1,6p gpurun_out/r03z/timeline_era5_025deg_480.txt
for i in 1 2 3; do python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('ms', round(d['ms_per_step'],4), 'thr', round(d['kernels_ms']['k_threshold'],4), 'rel', round(d['kernels_ms']['k_relabel'],4))"; done
