# A/B of the fused (CTK_ASYNC=1) and the synchronous (CTK_ASYNC=0) one-call path on the same box, alternating
WL=${1:-era5_1deg_djf30}
for rep in 1 2 3; do for m in 0 1; do
CTK_ASYNC=$m python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra --workload $WL 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('async=$m', round(d['ms_per_step'],4), d['config']['n_tracked'], 'thr', round(d['kernels_ms']['k_threshold'],4), 'rel', round(d['kernels_ms']['k_relabel'],4))"
done; done
