"""Build container only (needs /root/reference): oracle/lifecycle_port.py against the UNMODIFIED reference's run_lifecycle
(under tests/minixr.py) on the random label planes of tests/life_util.py.   python tools/fuzz_lifecycle_port_vs_reference.py <first> <count>"""
import logging, os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import life_util, minixr, refimport
from oracle import lifecycle_port
first, count = int(sys.argv[1]), int(sys.argv[2])
cls = refimport.load()
bad, n = [], 0
for i in range(first, first + count):
    flag, field, lat, lon, wrow, _ = life_util.random_life_case(i)
    T, ny, nx = flag.shape
    if T < 2 or ny < 2 or nx < 2:
        continue
    ds = minixr.make_dataset(field, lat, lon)
    ds["flag"] = minixr.DataArray(flag, ("time", "latitude", "longitude"))
    c = cls(); c.read_xarray(ds)
    time = (np.datetime64("2001-01-01T00", "h") + np.arange(T) * 6).astype("datetime64[ns]")
    logging.disable(logging.CRITICAL)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            c.set_up(time_name="time", longitude_name="longitude", latitude_name="latitude", force=True)
            ds["time"].data = time
            df = c.run_lifecycle(flag="flag", variable="anom")
    except Exception as e:                                       # noqa: BLE001
        bad.append((i, "REF EXC", str(e)[:60])); continue
    finally:
        logging.disable(logging.NOTSET)
    w = np.array(111 * c._dlat * 111 * c._dlon * np.cos(np.asarray(lat) * np.pi / 180)).astype(np.float32)
    want = [(int(a), str(b), int(c_), int(d), float(e), float(f)) for a, b, c_, d, e, f in zip(df.Flag, df.Date, df.Longitude, df.Latitude, df.Intensity, df.Size)]
    got = lifecycle_port.run_lifecycle(flag, field, lat, lon, w, life_util.dates_of(time))
    got = [(int(a), str(b), int(c_), int(d), float(e), float(f)) for a, b, c_, d, e, f in got]
    n += 1
    if got != want:
        bad.append((i, flag.shape, [(x, y) for x, y in zip(got, want) if x != y][:2]))
print("lifecycle port vs reference %d..%d: compared %d, mismatches %d %s" % (first, first + count - 1, n, len(bad), bad[:3]))
