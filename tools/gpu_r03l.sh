for wl in era5_1deg_djf30 era5_025deg_2k era5_025deg_10yr; do for e in "X=1" "CTK_RELABEL_PLAIN=1"; do
env $e python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --workload $wl 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$wl $e', 'ms', round(d['ms_per_step'],3), 'rel', round(d['kernels_ms']['k_relabel'],4), 'thr', round(d['kernels_ms']['k_threshold'],4))"
done; done
