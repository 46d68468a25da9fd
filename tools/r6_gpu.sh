#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_lifecycle.py -x -q -m gpu > gpurun_out/r6_parity.txt 2>&1
echo "parity rc $?" >> gpurun_out/r6_parity.txt
tail -4 gpurun_out/r6_parity.txt
timeout 1500 python tools/r6_ab.py --rounds 3 --steps 40 base CTK_EXTENT_BLK=0 > gpurun_out/r6_ab5.txt 2>&1
tail -10 gpurun_out/r6_ab5.txt
timeout 1500 python tools/r6_ab.py --workload cesm_le_40x30yr --rounds 1 --steps 6 base CTK_EXTENT_BLK=0 > gpurun_out/r6_ab5_cesm.txt 2>&1
tail -6 gpurun_out/r6_ab5_cesm.txt
timeout 900 python tools/r6_ab.py --workload era5_025deg_10yr --rounds 1 --steps 6 base CTK_EXTENT_BLK=1 > gpurun_out/r6_ab5_10yr.txt 2>&1
tail -6 gpurun_out/r6_ab5_10yr.txt
