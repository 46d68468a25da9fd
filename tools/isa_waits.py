"""Which kernels wait for everything right behind a load?  hipcc places `s_waitcnt vmcnt(0)` at the join of a branch around a load (and
in front of an instruction that overwrites a register a load is still filling); vmcnt counts in order, so such a wait stalls the wave
for EVERY load it has in flight -- a level of loads that was meant to travel together degenerates into a chain (profiles/NOTES.md,
round 4: k_overlap -5 us, k_relabel_v5 -4 %, k_life_strips).  Lists, per kernel, its global loads and how many of them are followed
by a full wait within three instructions.

    cd contrack_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off --save-temps=obj -c ctk_api.hip -o /tmp/x.o
    python tools/isa_waits.py contrack_amd/csrc/ctk_api-hip-amdgcn-amd-amdhsa-gfx950.s [min_waits]
"""
import re
import shutil
import subprocess
import sys

path = sys.argv[1]
min_waits = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lines = open(path).read().split("\n")
cur, res = None, {}
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):\s", l)
    if m:
        cur = m.group(1)
        res[cur] = [0, 0]
        continue
    if cur is None:
        continue
    if l.startswith(".Lfunc_end"):
        cur = None
        continue
    if "global_load" in l or "buffer_load" in l:
        res[cur][0] += 1
        for j in range(i + 1, min(i + 4, len(lines))):
            if "s_waitcnt vmcnt(0)" in lines[j]:
                res[cur][1] += 1
                break
            if "global_load" in lines[j]:
                break
names = list(res)
filt = shutil.which("c++filt")
dem = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.split("\n") if filt else names
print("loads  followed by vmcnt(0)  kernel")
for n, d in sorted(zip(names, dem), key=lambda p: -res[p[0]][1]):
    if res[n][1] >= min_waits:
        print("%5d  %5d  %s" % (res[n][0], res[n][1], d[:140]))
