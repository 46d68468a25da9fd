#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r6_parity.txt 2>&1
echo "parity rc $?" >> gpurun_out/r6_parity.txt
tail -3 gpurun_out/r6_parity.txt
timeout 1500 python tools/r6_ab.py --rounds 3 --steps 30 base CTK_RC_THREADS=128 CTK_RC_THREADS=512 > gpurun_out/r6_ab9.txt 2>&1
tail -14 gpurun_out/r6_ab9.txt | cut -c1-330
timeout 1500 python tools/r6_ab.py --workload cesm_le_40x30yr --rounds 1 --steps 6 base > gpurun_out/r6_ab9_cesm.txt 2>&1
tail -4 gpurun_out/r6_ab9_cesm.txt | cut -c1-330
