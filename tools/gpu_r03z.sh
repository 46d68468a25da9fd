for e in "CTK_OVL_THREADS=1024" "CTK_OVL_THREADS=512" "CTK_OVL_THREADS=513" "CTK_OVL_THREADS=1024" "CTK_OVL_THREADS=512" "CTK_OVL_THREADS=513"; do
env $e python bench.py --steps 30 --warmup 3 --no-cpu-baseline --workload era5_025deg_480 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$e', 'ms', round(d['ms_per_step'],4), 'thr', round(d['kernels_ms']['k_threshold'],4), 'rel', round(d['kernels_ms']['k_relabel'],4), 'ovl', round(d['kernels_ms']['k_overlap'],4), 'mid', round(d['ms_per_step']-d['kernels_ms']['k_threshold']-d['kernels_ms']['k_relabel'],4))"
done
