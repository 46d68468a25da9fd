#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err
tail -3 gpurun_out/r6_bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_bench_default.json').read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "roofline", {k: d["roofline"][k] for k in ("kernel","frac","traffic","avg_kernel_ms")}, d["roofline"]["traffic_source"][:160])
print("pmc_live", d.get("pmc_live"))
print("stats", {k: d["workload_stats"][k] for k in ("mask_allocations_tried","mask_ratio_x1000","mask_check_us","mask_spacer_mb")})
print("sec", d["secondary_025deg"]["ms_per_step"], d["secondary_025deg"]["workload_stats"]["mask_allocations_tried"], d["secondary_025deg"]["workload_stats"]["mask_ratio_x1000"], d["secondary_025deg"]["workload_stats"].get("mask_check_us"))
print("e2e", d["e2e"]["ms_per_call"], "conc", d["concurrent_members"]["ms_per_slab"], "cpu", d["cpu_baseline"]["value"])
PY
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/r6_gputests.txt 2>&1
tail -6 gpurun_out/r6_gputests.txt
