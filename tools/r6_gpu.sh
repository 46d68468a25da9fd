#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.txt 2>&1
grep -n "passed\|failed\|error" gpurun_out/gpu_tests.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
(
timeout 1500 python tools/fuzz.py 1000000 12000
timeout 600 python tools/fuzz.py 1020000 3000 edge
timeout 600 python tools/fuzz.py 1030000 1500 f64
timeout 600 python tools/fuzz.py 1040000 1500 thr
timeout 900 python tools/fuzz_sharded.py 1050000 3000
CTK_SH_FORCE_SPLIT=1 timeout 600 python tools/fuzz_sharded.py 1060000 800
timeout 600 python tools/fuzz_stream.py 1070000 800
timeout 600 python tools/fuzz_lifecycle.py 1080000 1500
) > gpurun_out/fuzz_final.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" gpurun_out/fuzz_final.txt | tail -10
