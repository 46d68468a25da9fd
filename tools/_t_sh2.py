import sys, os, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import golden_util
from contrack_amd import _native
from shard_inproc import sharded_threads
trks=[_native.Tracker(0) for _ in range(6)]
name=os.environ.get("CASE","T2"); n=int(os.environ.get("N","2"))
g=golden_util.load(name)
T=g["anom"].shape[0]
op=_native.CMP_OPS[g["gorl"]]
cuts=[int(round(T*k/n)) for k in range(n+1)]
f,nt,st=sharded_threads(trks[:n],g["anom"],g["thr"],op,g["wrow"],g["overlap"],g["persistence"],g["twosided"],cuts)
w=g["flag"]
print("cuts",cuts,"nt",nt,"want",len(np.unique(w))-1)
for t in range(T):
    a,b=f[t],w[t]
    if not np.array_equal(a,b):
        print("t",t,"got ids",np.unique(a),"want ids",np.unique(b),"mask equal",np.array_equal(a>0,b>0), "ndiff",(a!=b).sum())
print(st)
