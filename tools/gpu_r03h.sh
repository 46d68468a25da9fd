timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "not processes" 2>&1 | tail -3
for wl in era5_025deg_480 era5_025deg_1k era5_025deg_2k; do
for env in "X=1" "CTK_RELABEL_V4=1" "CTK_RELABEL_ROWS=4" "CTK_RELABEL_ROWS=8 CTK_RELABEL_V4=1"; do
env $env python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --workload $wl 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$wl', '$env', 'ms', round(d['ms_per_step'],3), 'rel', round(d['kernels_ms']['k_relabel'],4), 'thr', round(d['kernels_ms']['k_threshold'],4), 'kernel', d['workload_stats']['relabel_kernel'])"
done; done
