#!/bin/bash
# tools/profile.sh <tag> [workload]  -- rocprofv3 kernel stats + HBM traffic counters for the bench workload.
# Run on the GPU box (gpurun); writes under gpurun_out/<tag>/ ; copy the summaries you keep into profiles/.
set -u
TAG=${1:-prof}
WL=${2:-era5_1deg_djf30}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra --workload $WL"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- $CMD > $OUT/trace.log 2>&1
# the bench line of the TRACED process (its per-kernel times and the kernel stats above are the same launches: round-4 verdict weak #4)
grep -a '^{"metric"' $OUT/trace.log | tail -1 > $OUT/traced_bench.json
# PMC passes, one counter group per run (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o r -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o r -- $CMD > $OUT/pmc_write.log 2>&1
ls -R $OUT | head -30
