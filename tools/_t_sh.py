import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import golden_util
from contrack_amd import _native
from shard_inproc import sharded_threads
trks=[_native.Tracker(0) for _ in range(6)]
bad=0
import os
for name in (os.environ.get("CASES","").split(",") if os.environ.get("CASES") else golden_util.case_names()):
    g=golden_util.load(name)
    T=g["anom"].shape[0]
    op=_native.CMP_OPS[g["gorl"]]
    for n in (1,2,3,5):
        if n>T: continue
        cuts=[int(round(T*k/n)) for k in range(n+1)]
        if len(set(cuts))!=n+1: continue
        try:
            f,nt,st=sharded_threads(trks[:n],g["anom"],g["thr"],op,g["wrow"],g["overlap"],g["persistence"],g["twosided"],cuts)
            ok=np.array_equal(f,g["flag"]) and nt==len(np.unique(g["flag"]))-1
        except Exception as e:
            ok=False; print("EXC",name,n,repr(e)[:200])
        if not ok: bad+=1
        print(name,n,"ok" if ok else "FAIL", [ (s["filter_rounds"],s["shared_seam_rows"],s["ambiguous_decisions"]) for s in st] if ok or 'st' in dir() else "")
print("bad",bad)
