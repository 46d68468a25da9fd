#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python tools/r6_ab.py --steps 30 --rounds 2 base CTK_RELABEL_ROWS=8 CTK_RELABEL_ROWS=11 CTK_RELABEL_ROWS=22 CTK_RELABEL_ROWS=32 CTK_RELABEL_LDS_KB=16 CTK_RELABEL_LDS_KB=26 > gpurun_out/ab_rows_1deg.txt 2>&1
grep -v "^    SDDBG" gpurun_out/ab_rows_1deg.txt | tail -10
