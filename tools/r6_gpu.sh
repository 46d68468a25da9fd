#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep "passed\|failed" | tail -2
timeout -k 5 900 python tools/r6_ab.py --steps 40 --rounds 3 base > gpurun_out/ab_final_1deg.txt 2>&1
grep -v "^    SDDBG" gpurun_out/ab_final_1deg.txt | grep "^base" | head -3 | sed "s/cs=.*k_threshold/ k_threshold/" | cut -c1-300
