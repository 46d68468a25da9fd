timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "not processes and not rare_paths" 2>&1 | tail -3
for wl in era5_1deg_djf30 era5_025deg_2k; do CTK_SD_DBG=10 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --workload $wl 2>&1 | grep "SDDBG\|ms_per_step" | tail -2 | cut -c1-300; done
