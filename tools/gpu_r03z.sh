timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "not processes" 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -3
for rep in 1 2 3; do for e in "CTK_SYNC_STREAM=1" "X=1"; do
env $e python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$e', 'ms', round(d['ms_per_step'],4), 'thr', round(d['kernels_ms']['k_threshold'],4), 'rel', round(d['kernels_ms']['k_relabel'],4))"
done; done
for e in "CTK_SYNC_STREAM=1" "X=1"; do env $e python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extra --workload era5_1deg_90 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('90 $e', 'ms', round(d['ms_per_step'],4))"; done
