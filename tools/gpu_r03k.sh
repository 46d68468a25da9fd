timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "not processes" 2>&1 | grep "passed\|failed"
for wl in era5_1deg_djf30 era5_025deg_480 era5_025deg_1k era5_025deg_2k era5_025deg_10yr; do
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --workload $wl 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$wl', 'ms', round(d['ms_per_step'],3), 'rel', round(d['kernels_ms']['k_relabel'],4), 'frac', round(4*d['roofline']['algorithmic_bytes_per_launch']/4/d['kernels_ms']['k_relabel']/1e6/8000,3), 'thr', round(d['kernels_ms']['k_threshold'],4), 'kernel', d['workload_stats']['relabel_kernel'], 'fused', d['workload_stats']['fused_pass'], 'n', d['config']['n_tracked'])"
done
