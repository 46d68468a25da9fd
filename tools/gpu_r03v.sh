for e in "CTK_OVL_W=0" "CTK_OVL_W=5" "CTK_OVL_W=6" "CTK_OVL_W=8" "CTK_OVL_W=48" "CTK_OVL_W=46" "CTK_OVL_W=0"; do
env $e python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extra --workload era5_1deg_djf30 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$e', 'ms', round(d['ms_per_step'],4), 'thr', round(d['kernels_ms']['k_threshold'],4), 'rel', round(d['kernels_ms']['k_relabel'],4), 'mid', round(d['ms_per_step']-d['kernels_ms']['k_threshold']-d['kernels_ms']['k_relabel'],4))"
done
