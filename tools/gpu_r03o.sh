for wl in era5_025deg_2k era5_025deg_480 era5_1deg_djf30; do
CTK_SD_DBG=10 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --workload $wl 2>&1 | grep SDDBG | tail -1
CTK_ASYNC=0 CTK_SEAMSTATS=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --workload $wl 2>&1 | grep SEAMSTATS | tail -1
done
