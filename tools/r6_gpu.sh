#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r6_parity.txt 2>&1
echo "parity rc $?" >> gpurun_out/r6_parity.txt
tail -3 gpurun_out/r6_parity.txt
python tools/phase_probe.py 2707 181 360 2>&1 | tail -2
timeout 1500 python tools/r6_ab.py --rounds 3 --steps 30 base > gpurun_out/r6_ab8.txt 2>&1
tail -6 gpurun_out/r6_ab8.txt | cut -c1-330
timeout 1500 python tools/r6_ab.py --workload cesm_le_40x30yr --rounds 1 --steps 6 base > gpurun_out/r6_ab8_cesm.txt 2>&1
tail -4 gpurun_out/r6_ab8_cesm.txt | cut -c1-330
timeout 1500 python tools/r6_ab.py --workload era5_025deg_480 --rounds 1 --steps 20 base > gpurun_out/r6_ab8_025.txt 2>&1
tail -4 gpurun_out/r6_ab8_025.txt | cut -c1-330
