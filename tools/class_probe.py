"""What a user of the drop-in class pays per call (tests/minixr.py stands in for xarray): run_contrack with host arrays, run_lifecycle,
calc_anom followed by run_contrack on the resident slab; cProfile of one call each.  python tools/class_probe.py"""
import os, sys, time, logging
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import minixr
minixr.install_as_xarray()
from contrack_amd import synth
from contrack_amd.contrack import contrack
T, ny, nx = 2707, 181, 360
a = synth.smooth_field(T, ny, nx, seed=0)
lat, lon = synth.grid(ny, nx)
time_ax = (np.datetime64("2000-12-01") + np.arange(T)).astype("datetime64[ns]")
ds = minixr.make_dataset(a, lat, lon, time=time_ax)
ds["time"].attrs = {}
c = contrack(ds=ds)
c.set_up(time_name="time", longitude_name="longitude", latitude_name="latitude")
for i in range(4):
    t0 = time.perf_counter()
    c.run_contrack(variable='anom', threshold=160, gorl='>=', overlap=0.5, persistence=5)
    print("run_contrack (class, host arrays): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter()
df = c.run_lifecycle(flag='flag', variable='anom')
print("run_lifecycle (class): %.1f ms, %d rows" % ((time.perf_counter() - t0) * 1e3, len(df)))
t0 = time.perf_counter()
df = c.run_lifecycle(flag='flag', variable='anom')
print("run_lifecycle (class) again: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
ds["z"] = minixr.DataArray(a + 5000, ("time", "latitude", "longitude"), attrs={"units": "m", "long_name": "Z500"})
for i in range(2):
    t0 = time.perf_counter()
    c.calc_anom(variable="z", window=31, smooth=2, groupby="dayofyear")
    t1 = time.perf_counter()
    c.run_contrack(variable='anom', threshold=160, gorl='>=', overlap=0.5, persistence=5)
    print("calc_anom %.1f ms, then run_contrack on the resident slab %.1f ms" % ((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))
import cProfile, pstats
pr = cProfile.Profile()
logging.disable(logging.CRITICAL)
pr.enable()
c.run_contrack(variable='anom', threshold=160, gorl='>=', overlap=0.5, persistence=5)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
pr = cProfile.Profile()
pr.enable()
df = c.run_lifecycle(flag='flag', variable='anom')
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
