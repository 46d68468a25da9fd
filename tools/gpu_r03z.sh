for rep in 1 2 3 4; do for e in "CTK_THRESHOLD=4" "CTK_THRESHOLD=48"; do
env $e python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$e', 'ms', round(d['ms_per_step'],4), 'thr', round(d['kernels_ms']['k_threshold'],4), 'rel', round(d['kernels_ms']['k_relabel'],4))"
done; done
for rep in 1 2; do for e in "CTK_THRESHOLD=4" "CTK_THRESHOLD=48"; do
env $e python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extra --workload era5_025deg_480 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('480 $e', 'ms', round(d['ms_per_step'],4), 'thr', round(d['kernels_ms']['k_threshold'],4), 'rel', round(d['kernels_ms']['k_relabel'],4))"
done; done
