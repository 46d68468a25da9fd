#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r6_parity.txt 2>&1
echo "parity rc $?" >> gpurun_out/r6_parity.txt
tail -4 gpurun_out/r6_parity.txt
timeout 1500 python tools/r6_ab.py --rounds 3 --steps 40 base CTK_COMPACT_LAUNCH=1 CTK_COUNT_OLD=1 CTK_COMPACT_LAUNCH=1,CTK_COUNT_OLD=1 > gpurun_out/r6_ab4.txt 2>&1
echo "ab rc $?" >> gpurun_out/r6_ab4.txt
tail -20 gpurun_out/r6_ab4.txt
