for e in "CTK_RC_THREADS=1024" "CTK_RC_THREADS=512" "CTK_RC_THREADS=256" "CTK_RC_THREADS=1024" "CTK_RC_THREADS=512"; do
env $e python bench.py --steps 30 --warmup 3 --no-cpu-baseline --workload era5_025deg_480 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$e', 'ms', round(d['ms_per_step'],4), 'thr', round(d['kernels_ms']['k_threshold'],4), 'rel', round(d['kernels_ms']['k_relabel'],4), 'scan', round(d['kernels_ms']['k_scan'],4), 'mid', round(d['ms_per_step']-d['kernels_ms']['k_threshold']-d['kernels_ms']['k_relabel'],4))"
done
for e in "CTK_RC_THREADS=256" "CTK_RC_THREADS=128" "CTK_RC_THREADS=512" "CTK_RC_THREADS=256" "CTK_RC_THREADS=128"; do
env $e python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('1deg $e', 'ms', round(d['ms_per_step'],4), 'thr', round(d['kernels_ms']['k_threshold'],4), 'rel', round(d['kernels_ms']['k_relabel'],4), 'scan', round(d['kernels_ms']['k_scan'],4), 'mid', round(d['ms_per_step']-d['kernels_ms']['k_threshold']-d['kernels_ms']['k_relabel'],4))"
done
