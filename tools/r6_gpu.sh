#!/bin/bash
# round 6: the capture for profiles/ (kernel stats + PMC of four workloads, bench lines, SQ counters, 8-rank shm strong-scaling line)
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
tail -3 gpurun_out/r06_shm8.err
ls -la gpurun_out | grep r06
