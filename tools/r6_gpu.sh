#!/bin/bash
# round 6: the new test + seeded fuzz of the final library (GPU vs oracle / port)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sixteen" > gpurun_out/r6_parity.txt 2>&1
tail -3 gpurun_out/r6_parity.txt
(
timeout 1200 python tools/fuzz.py 600000 6000 2>&1 | tail -1
timeout 600 python tools/fuzz.py 610000 3000 edge 2>&1 | tail -1
timeout 600 python tools/fuzz.py 620000 1500 f64 2>&1 | tail -1
timeout 600 python tools/fuzz.py 630000 1500 thr 2>&1 | tail -1
timeout 900 python tools/fuzz_sharded.py 640000 3000 2>&1 | tail -1
CTK_SH_FORCE_SPLIT=1 timeout 600 python tools/fuzz_sharded.py 650000 800 2>&1 | tail -1
timeout 600 python tools/fuzz_stream.py 660000 800 2>&1 | tail -1
timeout 600 python tools/fuzz_lifecycle.py 670000 1500 2>&1 | tail -1
) > gpurun_out/r06_fuzz.txt 2>&1
cat gpurun_out/r06_fuzz.txt
