#!/bin/bash
# the builder's scratch script of round 6 (gpurun -- 'bash tools/r6_gpu.sh'): the full GPU suite, smoke() and a driver-style bench line
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.txt 2>&1
grep -n "passed\|failed\|error" gpurun_out/gpu_tests.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/bench_final.json') if l.startswith('{')][-1])
print(d['ms_per_step'], d['roofline']['frac'], d['workload_stats'].get('mask_ratio_x1000'), d.get('secondary_025deg',{}).get('ms_per_step'))"
timeout 600 python tools/fuzz.py 1400000 3000 edge 2>&1 | tail -1
