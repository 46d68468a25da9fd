#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python tools/r6_ab.py --rounds 2 --steps 20 base CTK_SD_DBG=10 CTK_SD_DBG=1 CTK_SD_DBG=2 CTK_SD_DBG=3 > gpurun_out/r6_ab14.txt 2>&1
grep -v "^----\|^distinct" gpurun_out/r6_ab14.txt | sed "s/ms  n=.*'k_resolve': \([0-9.]*\).*/ms k_resolve \1/" | cut -c1-200
