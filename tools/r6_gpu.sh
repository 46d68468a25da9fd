#!/bin/bash
# round 6: what the driver runs at the end of the round -- the gpu test suite, smoke(), the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/r6_gputests.txt 2>&1
tail -4 gpurun_out/r6_gputests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err
tail -3 gpurun_out/r6_bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_bench_default.json').read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], {k: d["roofline"][k] for k in ("kernel","frac","traffic","avg_kernel_ms")})
print("stats", {k: d["workload_stats"][k] for k in ("mask_allocations_tried","mask_ratio_x1000","mask_check_us","mask_spacer_mb","fused_pass")})
print("ceil", {k:v for k,v in d["stream_ceiling"].items() if k!="note"})
print("sec", d["secondary_025deg"]["ms_per_step"], "e2e", d["e2e"]["ms_per_call"], "conc", d["concurrent_members"]["ms_per_slab"], "cpu", d["cpu_baseline"]["value"], d["secondary_025deg"]["cpu_baseline"]["flags_equal_gpu"])
PY
