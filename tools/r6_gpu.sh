#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_lifecycle.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -3
timeout 300 python tools/phase_probe_life.py 480 721 1440 > gpurun_out/life_probe.txt 2>&1
tail -3 gpurun_out/life_probe.txt
rm -rf gpurun_out/prof_life
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_life -o life -- python tools/lifecycle_probe.py 480 721 1440 > gpurun_out/life_probe_rocprof.txt 2>&1
grep -v "^W2026\|^E2026" gpurun_out/life_probe_rocprof.txt | tail -3
f=$(find gpurun_out/prof_life -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r06_lifecycle_025deg_480_kernel_stats.csv
grep "k_life" gpurun_out/r06_lifecycle_025deg_480_kernel_stats.csv | sed 's/([^)]*)//' | cut -c1-120
rm -rf gpurun_out/prof_life
timeout 900 python tools/fuzz_lifecycle.py 950000 3000 > gpurun_out/fuzz_life.txt 2>&1
tail -2 gpurun_out/fuzz_life.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "life or class or column" 2>&1 | grep "passed\|failed"
