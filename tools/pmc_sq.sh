#!/bin/bash
# tools/pmc_sq.sh <tag> [workload]  -- SQ counters (instructions per wave, where the wave cycles go) for every kernel of the bench pass.
# Run on the GPU box (gpurun); writes gpurun_out/<tag>/sq_summary.md.  Two passes (8 SQ slots per pass).
set -u
TAG=${1:-sq}
WL=${2:-era5_1deg_djf30}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra --workload $WL"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $OUT/sq1 -o r -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/sq2 -o r -- $CMD > $OUT/sq2.log 2>&1
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("sq1", "sq2"):
    for path in glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM",
         "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_BUSY_CYCLES"]
with open(out + "/sq_summary.md", "w") as fh:
    fh.write("per launch (mean); instruction counters also per wave\n\n| kernel | " + " | ".join(n.replace("SQ_", "") for n in names) + " | VALU/wave | SALU/wave | LDS/wave |\n|" + "---|" * (len(names) + 4) + "\n")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0])) / max(1, len(kv[1].get("SQ_WAVE_CYCLES", [0])))):
        m = {n: (sum(v[n]) / len(v[n]) if v.get(n) else 0.0) for n in names}
        w = max(m["SQ_WAVES"], 1.0)
        fh.write("| %s | " % k + " | ".join("%.0f" % m[n] for n in names) + " | %.0f | %.0f | %.0f |\n" % (m["SQ_INSTS_VALU"] / w, m["SQ_INSTS_SALU"] / w, m["SQ_INSTS_LDS"] / w))
print(open(out + "/sq_summary.md").read())
PY
