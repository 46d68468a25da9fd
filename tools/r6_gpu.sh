#!/bin/bash
# round 6: long seeded fuzz of the final library (GPU vs oracle / port)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(
timeout 3000 python tools/fuzz.py 800000 30000 2>&1 | tail -1
timeout 1500 python tools/fuzz.py 840000 10000 edge 2>&1 | tail -1
timeout 1200 python tools/fuzz.py 860000 5000 f64 2>&1 | tail -1
timeout 1200 python tools/fuzz.py 870000 5000 thr 2>&1 | tail -1
timeout 2400 python tools/fuzz_sharded.py 880000 10000 2>&1 | tail -1
CTK_SH_FORCE_SPLIT=1 timeout 1200 python tools/fuzz_sharded.py 895000 2000 2>&1 | tail -1
timeout 1200 python tools/fuzz_stream.py 900000 3000 2>&1 | tail -1
timeout 1200 python tools/fuzz_lifecycle.py 910000 5000 2>&1 | tail -1
) > gpurun_out/r06_fuzz_long.txt 2>&1
cat gpurun_out/r06_fuzz_long.txt
