#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
for w in 0 6 7 8; do echo "CTK_LS_WPE=$w"; CTK_LS_WPE=$w timeout -k 5 300 python tools/lifecycle_probe.py 480 721 1440 2>&1 | grep "tracked\|agree"; CTK_LS_WPE=$w timeout -k 5 300 python tools/lifecycle_probe.py 2>&1 | grep "tracked"; done
