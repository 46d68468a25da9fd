#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
CTK_L2D_SMALL1=1 timeout -k 5 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
CTK_L2D_SMALL1=1 timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep "passed\|failed" | tail -2
timeout -k 5 900 python tools/r6_ab.py --steps 30 --rounds 3 base CTK_L2D_SMALL1=1 > gpurun_out/ab_l2d_small.txt 2>&1
grep -v "^    SDDBG" gpurun_out/ab_l2d_small.txt | grep "^base\|^CTK" | head -6 | sed "s/cs=.*'k_label2d': \([0-9.]*\).*/ label2d \1/"
grep -v "^    SDDBG" gpurun_out/ab_l2d_small.txt | tail -4
