timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "not processes and not rare_paths" 2>&1 | tail -4
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('1deg', d['ms_per_step'], d['config']['n_tracked'], d['workload_stats']['fused_pass'])"
bash tools/gpu_trace.sh r03d era5_1deg_djf30 | grep -v "k_rs_pass"
bash tools/gpu_trace.sh r03d era5_025deg_2k | grep -v "k_rs_pass"
