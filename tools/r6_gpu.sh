#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python tools/r6_ab.py --workload cesm_le_40x30yr --rounds 1 --steps 5 base CTK_XCD_THR=0 CTK_XCD_THR=1 CTK_XCD_THR=16 CTK_THR_ROWS=8 CTK_THR_ROWS=32 CTK_THR_ROWS=64 > gpurun_out/r6_ab11_cesm.txt 2>&1
tail -16 gpurun_out/r6_ab11_cesm.txt | cut -c1-300
