#!/bin/bash
# round 6, first visit of the GPU box: barrier microbench, core parity, A/B of the new pass structure
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 120 tools/exp/gridbar2 ) > gpurun_out/r6_gridbar2.txt 2>&1
echo "gridbar rc $?" >> gpurun_out/r6_gridbar2.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r6_parity.txt 2>&1
echo "parity rc $?" >> gpurun_out/r6_parity.txt
tail -5 gpurun_out/r6_parity.txt
timeout 1500 python tools/r6_ab.py --rounds 2 --steps 40 base CTK_EARLY_ZERO=0 CTK_EARLY_ZERO=0,CTK_THR_RC=0 CTK_SPARSE_ALL16=0 CTK_ZERO_GRID=512 CTK_ZERO_GRID=2048 CTK_SPARSE_ITERS=8 > gpurun_out/r6_ab1.txt 2>&1
echo "ab rc $?" >> gpurun_out/r6_ab1.txt
cat gpurun_out/r6_gridbar2.txt
tail -30 gpurun_out/r6_ab1.txt
