"""times the sections of the sharded driver at world_size 1 over nccl (python/torch overhead of the N > 1 path)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29519")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from contrack_amd import _native, synth, dist as cdist
comm = cdist.TorchComm(device=0)
T, ny, nx = 2707, 181, 360
a = synth.smooth_field(T, ny, nx, seed=0)
lat, _ = synth.grid(ny, nx)
w = np.array(111 * np.float32(1) * 111 * np.float32(1) * np.cos(lat * np.pi / 180)).astype(np.float32)
thr = np.full(T, np.float64(np.float32(160)))
trk = _native.Tracker(0)
d_in, d_out = trk.malloc(a.nbytes), trk.malloc(a.nbytes)
trk.h2d(d_in, a)
eng = cdist.HipShardEngine(trk, comm, d_in, T, ny, nx, thr, 0, w, d_out)
def timed():
    ts = [time.perf_counter()]
    eng.label2d(False); trk.sync(); ts.append(time.perf_counter())
    eng.overlap(); trk.sync(); ts.append(time.perf_counter())
    ptr, nbytes = trk.shard_tables_dev(); ts.append(time.perf_counter())
    mine = comm.device_bytes(ptr.value, nbytes)
    blobs, sizes = comm.allgather_device(mine); ts.append(time.perf_counter())
    ext, n = trk.shard_resolve_dev([b.data_ptr() for b in blobs], sizes, 0, 0, 0.5, True); trk.sync(); ts.append(time.perf_counter())
    tmin, tmax = eng._wrap_ext(ext, n)
    comm.allreduce_min_max(tmin, tmax); ts.append(time.perf_counter())
    eng.write(5); ts.append(time.perf_counter())
    return np.diff(ts) * 1e3
for _ in range(3): timed()
acc = np.mean([timed() for _ in range(10)], axis=0)
print("ms: label2d %.3f overlap %.3f tables_dev %.3f allgather %.3f resolve_dev %.3f allreduce %.3f write %.3f  total %.3f" % (*acc, acc.sum()))
dist.destroy_process_group()
