for wl in era5_1deg_djf30 era5_025deg_480; do for e in "CTK_THRESHOLD=4" "CTK_THRESHOLD=48" "CTK_THRESHOLD=42" "CTK_THRESHOLD=4" "CTK_THRESHOLD=48"; do
env $e python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --workload $wl 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$wl $e', 'ms', round(d['ms_per_step'],4), 'thr', round(d['kernels_ms']['k_threshold'],4), 'frac', round(d['roofline']['frac'],3))"
done; done
