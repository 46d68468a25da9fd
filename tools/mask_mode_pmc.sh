#!/bin/bash
# tools/mask_mode_pmc.sh [tag]: PMC groups of k_threshold_v7, slow handle vs fast handle (tools/mask_mode_pmc.py). GPU box only.
set -u
TAG=${1:-maskpmc}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CTK_MASK_TUNE=0 NP=6
CMD="python $ROOT/tools/mask_mode_pmc.py"
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -o r -- $CMD > $OUT/g$i.log 2>&1
  grep MODES $OUT/g$i.log
done <<'GROUPS'
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum
TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
TCC_HIT_sum TCC_MISS_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum
TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_SRC_FIFO_FULL_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_64B_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE
TCC_STREAMING_REQ_sum TCC_NC_REQ_sum TCC_UC_REQ_sum TCC_CC_REQ_sum
GROUPS
python3 - "$OUT" <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
NP = int(os.environ.get("NP", "6"))
rows = []
for d in sorted(glob.glob(out + "/g*/")):
    per = collections.defaultdict(list)                      # counter -> [(dispatch id, value)]
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "k_threshold_v7" in r["Kernel_Name"]:
                per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for name, v in per.items():
        v.sort()
        v = [x for _, x in v][-2 * NP:]
        s, f = v[:NP], v[NP:]
        rows.append((name, sum(s) / len(s), sum(f) / len(f)))
with open(out + "/summary.md", "w") as fh:
    fh.write("k_threshold_v7, mean of %d launches on the slow handle and on the fast handle of one process\n\n| counter | slow | fast | slow / fast |\n|---|---|---|---|\n" % NP)
    for name, s, f in rows:
        fh.write("| %s | %.0f | %.0f | %s |\n" % (name, s, f, ("%.3f" % (s / f)) if f else "-"))
print(open(out + "/summary.md").read())
PY
