#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_life
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_life -o life -- python tools/lifecycle_probe.py 480 721 1440 > gpurun_out/life_probe_rocprof.txt 2>&1
grep -v "^W2026\|^E2026" gpurun_out/life_probe_rocprof.txt | tail -8
f=$(find gpurun_out/prof_life -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r06_lifecycle_025deg_480_kernel_stats.csv
grep "k_life" gpurun_out/r06_lifecycle_025deg_480_kernel_stats.csv | cut -c1-40,260-
rm -rf gpurun_out/prof_life
