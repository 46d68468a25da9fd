#!/bin/bash
# round 5: k_threshold_v7 with and without its mask stores (CTK_THR_STORE=2), eight handles each, placement check off.  (The run kept in
# profiles/r05_thr_store_modes.txt also had non-temporal stores (1) and words staged in LDS + one coalesced copy per workgroup (3, 4): removed.)
mkdir -p gpurun_out/store
export CTK_MASK_TUNE=0
for rep in 1 2; do
for m in 0 2; do
  echo -n "store=$m " ; CTK_THR_STORE=$m python tools/thr_modes_probe.py
done
done > gpurun_out/store/probe.txt 2>&1
for m in 0 2; do
  echo -n "store=$m " ; CTK_THR_STORE=$m SHAPE=480,721,1440 python tools/thr_modes_probe.py
done >> gpurun_out/store/probe.txt 2>&1
cat gpurun_out/store/probe.txt
