#!/bin/bash
# round 6: the capture for profiles/ with the FINAL library (kernel stats + PMC of four workloads, bench lines, SQ counters, 8-rank
# shm strong-scaling line) and the seeded fuzz (GPU vs oracle / port)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu_profile_all.sh r06 > gpurun_out/r06_profile_all.log 2>&1
bash tools/pmc_sq.sh r06_sq_1deg era5_1deg_djf30 > gpurun_out/r06_sq_1deg.log 2>&1
cp gpurun_out/r06_sq_1deg/sq_summary.md gpurun_out/r06_sq_1deg.md
bash tools/pmc_sq.sh r06_sq_cesm cesm_le_40x30yr > gpurun_out/r06_sq_cesm.log 2>&1
cp gpurun_out/r06_sq_cesm/sq_summary.md gpurun_out/r06_sq_cesm_40x30yr.md
rm -rf gpurun_out/r06_sq_1deg gpurun_out/r06_sq_cesm
cd "$GRAFT_REPO_ROOT"
CTK_DIST_BACKEND=shm timeout 900 python bench.py --gpus 8 --scaling strong --workload era5_025deg_10yr --steps 5 --warmup 2 --no-parity-check --strong-steps 0 > gpurun_out/r06_bench_shm_8ranks_one_gpu_strong_025deg_10yr.json 2> gpurun_out/r06_shm8.err
(
timeout 1200 python tools/fuzz.py 700000 6000 2>&1 | tail -1
timeout 600 python tools/fuzz.py 710000 3000 edge 2>&1 | tail -1
timeout 600 python tools/fuzz.py 720000 1500 f64 2>&1 | tail -1
timeout 600 python tools/fuzz.py 730000 1500 thr 2>&1 | tail -1
timeout 900 python tools/fuzz_sharded.py 740000 3000 2>&1 | tail -1
CTK_SH_FORCE_SPLIT=1 timeout 600 python tools/fuzz_sharded.py 750000 800 2>&1 | tail -1
timeout 600 python tools/fuzz_stream.py 760000 800 2>&1 | tail -1
timeout 600 python tools/fuzz_lifecycle.py 770000 1500 2>&1 | tail -1
) > gpurun_out/r06_fuzz.txt 2>&1
cat gpurun_out/r06_fuzz.txt
