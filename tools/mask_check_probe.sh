#!/bin/bash
# round 5: the mask placement check (tuner on) on eight handles per process, both bench shapes; CTK_HOSTTRACE prints what it did
mkdir -p gpurun_out/mcheck
{
for rep in 1 2 3; do
  for shape in 2707,181,360 480,721,1440; do
    SHAPE=$shape CTK_HOSTTRACE=1 python tools/thr_modes_probe.py 2>&1 | grep -E "mask placement|tune="
  done
done
} > gpurun_out/mcheck/probe.txt 2>&1
cat gpurun_out/mcheck/probe.txt
