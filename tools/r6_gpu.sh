#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r6_parity.txt 2>&1
echo "parity rc $?" >> gpurun_out/r6_parity.txt
tail -3 gpurun_out/r6_parity.txt
timeout 1500 python tools/r6_ab.py --rounds 4 --steps 40 base > gpurun_out/r6_ab6.txt 2>&1
tail -8 gpurun_out/r6_ab6.txt
timeout 1500 python tools/r6_ab.py --workload cesm_le_40x30yr --rounds 1 --steps 6 base > gpurun_out/r6_ab6_cesm.txt 2>&1
tail -4 gpurun_out/r6_ab6_cesm.txt
timeout 1500 python tools/r6_ab.py --workload era5_025deg_480 --rounds 2 --steps 20 base > gpurun_out/r6_ab6_025.txt 2>&1
tail -5 gpurun_out/r6_ab6_025.txt
