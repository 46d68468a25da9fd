for rep in 1 2; do for r in 16 21 10 32 42; do
CTK_THR_ROWS=$r python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('rows $r', 'ms', round(d['ms_per_step'],4), 'thr', round(d['kernels_ms']['k_threshold'],4))"
done; done
for r in 16 11 21 32; do CTK_THR_ROWS=$r python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extra --workload era5_025deg_480 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('480 rows $r', 'ms', round(d['ms_per_step'],4), 'thr', round(d['kernels_ms']['k_threshold'],4))"; done
