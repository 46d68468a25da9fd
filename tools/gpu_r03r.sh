CTK_HOSTPROF=1 CTK_FORCE_DIST=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra --no-parity-check 2>&1 | grep HOSTPROF | tail -2
