#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/phase_probe_pb.py era5_1deg_djf30 > gpurun_out/pb_probe_1deg_cached.txt 2>&1
tail -14 gpurun_out/pb_probe_1deg_cached.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "parity or golden or resolve or filter" 2>&1 | tail -5
timeout 600 python tools/r6_ab.py --steps 40 --rounds 3 base > gpurun_out/ab_pb_cached.txt 2>&1
tail -8 gpurun_out/ab_pb_cached.txt
timeout 600 python tools/r6_ab.py --workload era5_025deg_480 --steps 30 --rounds 2 base > gpurun_out/ab_pb_cached_025.txt 2>&1
tail -5 gpurun_out/ab_pb_cached_025.txt
