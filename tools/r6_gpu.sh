#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
CTK_RELABEL_PERSIST=8 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or bench or randomized" > gpurun_out/r6_parity.txt 2>&1
tail -2 gpurun_out/r6_parity.txt
timeout 1500 python tools/r6_ab.py --rounds 3 --steps 30 base CTK_RELABEL_PERSIST=8 CTK_RELABEL_PERSIST=16 > gpurun_out/r6_ab13.txt 2>&1
tail -5 gpurun_out/r6_ab13.txt | cut -c1-200; grep -o "^[A-Za-z_=0-9]* .*'k_relabel': [0-9.]*" gpurun_out/r6_ab13.txt | sed "s/ .*'k_relabel'/ k_relabel/" 
timeout 1500 python tools/r6_ab.py --workload cesm_le_40x30yr --rounds 1 --steps 5 base CTK_RELABEL_PERSIST=8 CTK_RELABEL_PERSIST=16 > gpurun_out/r6_ab13_cesm.txt 2>&1
grep -o "^[A-Za-z_=0-9]* .*'k_relabel': [0-9.]*" gpurun_out/r6_ab13_cesm.txt | sed "s/ .*'k_relabel'/ k_relabel/"
timeout 1500 python tools/r6_ab.py --workload era5_025deg_480 --rounds 2 --steps 20 base CTK_RELABEL_PERSIST=8 > gpurun_out/r6_ab13_025.txt 2>&1
grep -o "^[A-Za-z_=0-9]* .*'k_relabel': [0-9.]*" gpurun_out/r6_ab13_025.txt | sed "s/ .*'k_relabel'/ k_relabel/"
timeout 1500 python tools/r6_ab.py --workload era5_025deg_10yr --rounds 1 --steps 5 base CTK_RELABEL_PERSIST=8 > gpurun_out/r6_ab13_10yr.txt 2>&1
grep -o "^[A-Za-z_=0-9]* .*'k_relabel': [0-9.]*" gpurun_out/r6_ab13_10yr.txt | sed "s/ .*'k_relabel'/ k_relabel/"
