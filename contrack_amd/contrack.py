"""Drop-in `contrack` class: the public interface of steidani/ConTrack's `contrack.contrack`
(contrack/contrack.py:49-949) with `run_contrack` executed by the MI355X HIP path.

Same constructor, `read`, `read_xarray`, `set_up`, `calc_clim`, `calc_anom`, `run_contrack` signatures, the
same `ValueError` / `IOError` texts, the same INFO log lines and the same `flag` variable (dims of the input
variable, int32 ids identical to the reference's, same attrs).  `run_contrack`, `run_lifecycle`, `calc_clim` / `calc_anom`
and the percentile threshold of the reference's README run on the GPU; `calc_anom` leaves its slab resident in HBM so that
the following `run_contrack` does not cross PCIe on the way in.

xarray is imported lazily: the class itself only needs the small part of the Dataset/DataArray API listed in
tests/minixr.py, so it also works on any duck-typed dataset.  `track_numpy` is the array-level entry.

Stated deviations from the reference:
  * `flag` is written back with the INVERSE of the (time, lat, lon) permutation; the reference applies the
    forward permutation twice (contrack.py:778), which is only correct when the permutation is its own
    inverse -- for every such input (including the tested (time, lat, lon) order) both agree.
  * the library computes int32 ids (more than 2^31-2 ids raise instead of wrapping); the class hands `flag` out in the
    reference's dtype -- int32, int64 from 2^31-2 elements on, as scipy.ndimage.label does (contrack.py:687, :751).  `track_numpy`
    and the C ABI stay int32.
  * the numpy array behind `ds['anom']` after `calc_anom` is read-only (its twin stays in HBM for `run_contrack`); assign a
    new array to the variable to change it.
  * `run_contrack(..., chunk_steps=n)` (extension): stream the variable through the GPU in slices of n time steps.
"""
import logging
import os

import numpy as np

from . import _native

logger = logging.getLogger(__name__)
logging.basicConfig(format='%(levelname)s: %(message)s', level=logging.INFO)

_TRACKERS = {}


def _tracker(device=None):
    dev = int(os.environ.get("CONTRACK_DEVICE", "0")) if device is None else int(device)
    if dev not in _TRACKERS:
        _TRACKERS[dev] = _native.Tracker(dev)
    return _TRACKERS[dev]


# ------------------------------------------------------------------------------------------------
# array-level API
# ------------------------------------------------------------------------------------------------
def row_weights(lat, dlat, dlon):
    """Row weights as contrack.py:703-704 evaluates them: cos(lat*pi/180) in the dtype of `lat`, scaled by
    111*dlat*111*dlon left to right, cast to float32 (pole rows come out slightly negative in float32)."""
    weight_lat = np.cos(np.asarray(lat) * np.pi / 180)
    return np.array((111 * dlat * 111 * dlon * weight_lat)).astype(np.float32).reshape(-1)


def prepare_thresholds(threshold, T, data_dtype):
    """Per-timestep float64 thresholds thr[t] such that (double)x <op> thr[t] is the compare the reference
    evaluates (contrack.py:650-671 under numpy's promotion rules): a Python number is cast to the array's
    dtype; a float64 numpy scalar / array promotes the compare to float64."""
    data_dtype = np.dtype(data_dtype)
    if isinstance(threshold, (bool, int, float)) and not isinstance(threshold, np.generic):
        if data_dtype.kind == "f":
            val = np.asarray(threshold, dtype=data_dtype).astype(np.float64)
        else:
            val = np.float64(threshold)
        return np.full(int(T), val, dtype=np.float64)
    arr = np.asarray(threshold)
    if arr.dtype.kind not in "fiub":
        raise TypeError("threshold must be numeric")
    if data_dtype == np.float32 and (arr.dtype == np.float32 or arr.dtype == np.float16 or
                                     (arr.dtype.kind in "iub" and arr.dtype.itemsize <= 2)):
        arr = arr.astype(np.float32)
    out = np.broadcast_to(arr.astype(np.float64).reshape(-1) if arr.ndim else arr.astype(np.float64), (int(T),))
    return np.ascontiguousarray(out, dtype=np.float64)


def track_numpy(anom, wrow, threshold, gorl, overlap, persistence, twosided=True, device=None):
    """run_contrack on a (time, lat, lon) numpy slab.  Returns (flag int32 (T,ny,nx), n_tracked).

    anom float32 (other dtypes are compared exactly in float64 on the device), wrow float32 (ny,) from
    `row_weights`, threshold scalar or per-timestep vector, gorl in {'>=','<=','>','<','ge','le','gt','lt'}."""
    if gorl not in _native.CMP_OPS:
        raise ValueError(_native.GORL_ERRMSG)
    anom = np.asarray(anom)
    if anom.ndim != 3:
        raise ValueError("anom must be (time, lat, lon)")
    thr = prepare_thresholds(threshold, anom.shape[0], anom.dtype)
    trk = _tracker(device)
    if anom.dtype != np.float32:
        if anom.dtype.kind not in "fiub":
            raise TypeError("anom must be a real numeric array")
        return trk.track(anom.astype(np.float64), thr, _native.CMP_OPS[gorl], wrow, overlap, persistence, twosided, f64=True)
    return trk.track(anom, thr, _native.CMP_OPS[gorl], wrow, overlap, persistence, twosided)


# ------------------------------------------------------------------------------------------------
# the class
# ------------------------------------------------------------------------------------------------

def lifecycle_columns(rows, lat, lon, dates, tracker=None):
    """ctk_life_row records -> the columns of the reference's frame (contrack.py:876-906), rows sorted by (Flag, Date):
    dict of arrays Flag, Date, Longitude, Latitude, Intensity, Size.

    The device sums are float64 but in another order than the reference's (np.sum pairwise, np.bincount sequential).  That shows
    only where a result sits on a rounding boundary: a centre of mass that is an integer up to rounding (contours one pixel wide
    or high: int() then gives the cell or its neighbour) or a value at the edge of two decimals.  With the Tracker that produced
    `rows` at hand those rows -- a few per cent in practice -- are re-evaluated ON THE DEVICE in the reference's own summation orders
    (ctk_lifecycle_exact)."""
    nx, ny = len(lon), len(lat)
    if (rows["shift"] == -2).any():
        raise ValueError("attempt to get argmax of an empty sequence")                                  # np.argmax(np.diff(.)), :883
    with np.errstate(divide="ignore", invalid="ignore"):
        intensity = rows["swv"] / rows["area"]                                                          # :876
        com_y, com_x = rows["swvy"] / rows["swv"], rows["swvx"] / rows["swv"]                           # ndimage.center_of_mass
    area = rows["area"].copy()
    if tracker is not None and len(rows):
        def near_int(v):
            return np.abs(v - np.rint(v)) <= 1e-9 * np.maximum(1.0, np.abs(v))

        def near_half(v, rel):                              # v * 100 close to k + 0.5: round(v, 2) could go either way
            s = np.abs(v) * 100.0
            return np.abs(s - np.floor(s) - 0.5) <= rel * np.maximum(1.0, s)
        # The area is a sum of POSITIVE float64 terms: the device's value (exact integers, rounded once) and numpy's pairwise sum
        # differ by less than 5e-15 of it (depth of the pairwise tree x 2^-53) -- 1e-11 is generous, while 1e-6 would flag every
        # contour beyond 5000 km^2 (a quarter of all rows).  Sums of field values can cancel: their window stays wide.
        with np.errstate(invalid="ignore"):
            fragile = near_int(com_y) | near_int(com_x) | near_half(intensity, 1e-6) | near_half(area, 1e-11) | ~np.isfinite(com_y) | ~np.isfinite(com_x)
        idx = np.nonzero(fragile)[0]
        if len(idx):
            ex = tracker.lifecycle_exact(idx)
            intensity, com_y, com_x = intensity.copy(), com_y.copy(), com_x.copy()
            with np.errstate(divide="ignore", invalid="ignore"):
                area[idx] = ex["area"]
                intensity[idx] = ex["swv"] / ex["area"]
                com_y[idx], com_x[idx] = ex["sy"] / ex["s"], ex["sx"] / ex["s"]
    if not (np.isfinite(com_y).all() and np.isfinite(com_x).all()):
        raise ValueError("cannot convert float NaN to integer")                                         # int(center_of_mass[..]), :886
    iy, ix = np.trunc(com_y).astype(np.int64), np.trunc(com_x).astype(np.int64)
    if ((iy < -ny) | (iy >= ny) | (ix < -nx) | (ix >= nx)).any():
        raise IndexError("centre of mass outside the grid")
    shift = np.where(rows["shift"] > 0, rows["shift"], 0)
    ix = np.where(ix < 0, ix + nx, ix)                                                                  # Python indexing of the rolled axis
    lon_of = np.asarray(lon)[(ix + shift) % nx]                                                         # np.roll(lon, -shift)[ix], :884-887
    lat_of = np.asarray(lat)[iy]
    date = np.asarray(dates, dtype=object)[rows["t"]] if len(rows) else np.empty(0, dtype=object)
    cols = dict(Flag=rows["label"].astype(np.int64), Date=date,
                Longitude=np.trunc(lon_of).astype(np.int64), Latitude=np.trunc(lat_of).astype(np.int64),           # int(), :886-895
                Intensity=np.round(intensity, 2), Size=np.round(area, 2))                               # round(np.float64, 2), :899-900
    # the library returns (label, t) order; the reference sorts by the date STRING, which differs when the time axis
    # is not increasing
    order = sorted(range(len(rows)), key=lambda i: (cols["Flag"][i], cols["Date"][i])) \
        if any(a > b for a, b in zip(dates, dates[1:])) else None
    return cols if order is None else {k: v[order] for k, v in cols.items()}


def lifecycle_frame(rows, lat, lon, dates, tracker=None):
    """the same as a list of (Flag, Date, Longitude, Latitude, Intensity, Size) tuples"""
    c = lifecycle_columns(rows, lat, lon, dates, tracker)
    return [(int(f), d, int(lo), int(la), float(it), float(sz)) for f, d, lo, la, it, sz in
            zip(c["Flag"], c["Date"], c["Longitude"], c["Latitude"], c["Intensity"], c["Size"])]


INT64_FLAG_FROM = 2 ** 31 - 2       # elements from which scipy.ndimage.label (and the reference's 'flag') switch to int64


def _xr():
    import xarray as xr
    return xr


def _fingerprint(a):
    """cheap identity of a host array (address, shape, dtype and a strided sample of its bytes): is the copy that calc_anom left
    on the GPU still the array the dataset holds?  A new array, another shape or an in-place edit of a sampled element miss.
    Edits the sample cannot see are excluded differently: calc_anom hands the host copy out READ-ONLY (an in-place edit raises
    instead of silently diverging from its twin in HBM; assign a new array to the variable to change it), and an array that has
    been made writeable again never counts as resident."""
    a = np.asarray(a)
    if a.flags.writeable:
        return None
    flat = a.reshape(-1) if a.flags.c_contiguous else None
    sample = b"" if flat is None or flat.size == 0 else np.ascontiguousarray(flat[::max(1, flat.size // 2048)][:2048]).tobytes()
    return (a.__array_interface__["data"][0], a.shape, str(a.dtype), hash(sample))


class contrack(object):
    """contrack class -- interface of steidani/ConTrack (contrack/contrack.py:49), HIP-accelerated run_contrack."""

    num_of_contrack = 0

    def __init__(self, filename="", ds=None, **kwargs):
        """contrack(filename) reads a netCDF file, contrack(ds=dataset) wraps a dataset, contrack() is empty
        (contrack.py:58-88)."""
        if not filename:
            self.ds = None if ds is None else ds
            return
        try:
            self.ds = None
            self.read(filename, **kwargs)
        except (OSError, IOError, RuntimeError):
            try:
                self.read(filename, **kwargs)
            except Exception:
                raise IOError("Unkown fileformat. Known formats are netcdf.")
        contrack.num_of_contrack += 1

    def __repr__(self):
        try:
            return "\
            Xarray dataset with {} time steps. \n\
            Available fields: {}".format(self.ntime, ", ".join(self.variables))
        except AttributeError:
            return "\
            Empty contrack container.\n\
            Hint: use read() to load data."

    def __str__(self):
        return 'Class {}: \n{}'.format(self.__class__.__name__, self.ds)

    def __len__(self):
        return len(self.ds)

    def __getattr__(self, attr):
        if attr in self.__dict__:
            return getattr(self, attr)
        if attr == "ds":
            raise AttributeError(attr)
        return getattr(self.ds, attr)

    def __getitem__(self, key):
        return self.ds[key]

    # ---- properties (contrack.py:119-161) ------------------------------------------------------
    def _dim_size(self, name):
        dims = self.ds.dims
        try:
            return dims[name]
        except (TypeError, KeyError, IndexError):          # newer xarray: dims of a Dataset may be a plain view
            return self.ds.sizes[name]

    @property
    def ntime(self):
        if len(self.ds.dims) != 3:
            logger.warning("\nBe careful with the dimensions, you want dims = 3 and shape:\n(latitude, longitude, time)")
        return self._dim_size(self._get_name_time())

    @property
    def variables(self):
        return list(self.ds.data_vars)

    @property
    def dimensions(self):
        return list(self.ds.dims)

    @property
    def grid(self):
        if len(self.ds.dims) != 3:
            logger.warning("\nBe careful with the dimensions, you want dims = 3 and shape:\n(latitude, longitude, time)")
            return None
        print("\
        latitude: {} \n\
        longitude: {}".format(self._dim_size(self._get_name_latitude()), self._dim_size(self._get_name_longitude())))

    @property
    def dataset(self):
        return self.ds

    # ---- read (contrack.py:166-199) ---------------------------------------------------------------
    def read(self, filename, **kwargs):
        if self.ds is None:
            self.ds = _xr().open_dataset(filename, **kwargs)
            logger.debug('read: {}'.format(self.__str__))
        else:
            raise ValueError('contrack() is already set!')

    def read_xarray(self, ds):
        if self.ds is None:
            try:
                xr = _xr()
                ok = isinstance(ds, xr.core.dataset.Dataset)
            except ImportError:                              # no xarray installed: accept a duck-typed dataset
                ok = hasattr(ds, "dims") and hasattr(ds, "data_vars")
            if not ok:
                raise ValueError('ds has to be a xarray data set!')
            self.ds = ds
            logger.debug('read_xarray: {}'.format(self.__str__))
        else:
            raise ValueError('contrack() is already set!')

    # ---- set up (contrack.py:204-380) ----------------------------------------------------------------
    def set_up(self, time_name=None, longitude_name=None, latitude_name=None, force=False, write=True):
        self._time_name = self._get_name_time() if time_name is None else time_name
        self._longitude_name = self._get_name_longitude() if longitude_name is None else longitude_name
        self._latitude_name = self._get_name_latitude() if latitude_name is None else latitude_name
        if (self._longitude_name and self._latitude_name) is not None:
            self._dlon = self._get_resolution(self._longitude_name, force=force)
            self._dlat = self._get_resolution(self._latitude_name, force=force)
        if self._time_name is not None:
            self._dtime = self._get_resolution(self._time_name, force=force)
        if write:
            self._log_dim_names()

    def _log_dim_names(self):
        logger.info("\n time: '{}'\n longitude: '{}'\n latitude: '{}'\n".format(
            self._time_name, self._longitude_name, self._latitude_name))

    def _units_of(self, dim):
        out = []
        da = self.ds[dim]
        for holder in (getattr(da, "attrs", None), getattr(da, "encoding", None)):
            if holder and 'units' in holder:
                out.append(holder['units'])
        return out

    def _get_name_time(self):
        for dim in self.ds.dims:
            if any('since' in u for u in self._units_of(dim)) or dim in ['time']:
                return dim
        for name in self.ds.variables:
            data = self.ds[name].data
            try:
                first = data[0]
            except IndexError:
                first = data
            if isinstance(first, np.datetime64):
                return name
        logger.warning("\n 'time' dimension (dtype='datetime64[ns]') not found.")
        return None

    def _get_name_longitude(self):
        for dim in self.ds.dims:
            attrs = getattr(self.ds[dim], "attrs", {})
            if attrs.get('units') in ['degree_east', 'degrees_east'] or dim in ['lon', 'longitude', 'x']:
                return dim
        logger.warning("\n 'longitude' dimension (unit='degrees_east') not found.")
        return None

    def _get_name_latitude(self):
        for dim in self.ds.dims:
            attrs = getattr(self.ds[dim], "attrs", {})
            if attrs.get('units') in ['degree_north', 'degrees_north'] or dim in ['lat', 'latitude', 'y']:
                return dim
        logger.warning("\n 'latitude' dimension (unit='degrees_north') not found.")
        return None

    def _get_resolution(self, dim, force=False):
        """grid spacing in degrees / time step in hours (contrack.py:327-380)"""
        if dim == self._time_name:
            try:
                stamps = np.asarray(self.ds[dim].to_index().values)
                # (numpy arithmetic: pandas >= 2 refuses TimedeltaIndex.astype('timedelta64[h]'))
                delta = np.unique((stamps[1:] - stamps[:-1]).astype('timedelta64[h]'))
            except AttributeError:
                attrs = getattr(self.ds[dim], "attrs", {})
                if 'units' in attrs and 'days' in attrs['units']:
                    var = self.ds[dim].data
                    delta = np.unique(var[1:] - var[:-1])
                else:
                    raise ValueError('Can not decode time with unit {}'.format(attrs['units']))
        else:
            data = self.ds[dim].data
            delta = abs(np.unique(data[1:] - data[:-1]))
        if len(delta) > 1:
            errmsg = 'No regular grid found for dimension {}.\n\
            Hint: use set_up(force=True).'.format(dim)
            if force and dim != self._time_name:
                logging.warning(errmsg)
                logging.warning(' '.join(['force=True: using mean of non-equidistant', 'grid {}'.format(delta)]))
                delta = round(delta.mean(), 2)
            elif dim == self._time_name:
                logging.warning(errmsg)
            else:
                raise ValueError(errmsg)
        elif delta[0] == 0:
            raise ValueError('Two equivalent values found for dimension {}.'.format(dim))
        elif delta[0] < 0:
            raise ValueError(' '.join(['{} not increasing. This should', 'not happen?!']).format(dim))
        return delta

    def _ensure_set_up(self):
        logger.info("Set up dimensions...")
        if hasattr(self, '_time_name'):
            self._log_dim_names()
        else:
            self.set_up()

    # ---- pre-processing glue (contrack.py:386-581); plain xarray, not accelerated ------------------------
    def calculate_gph_from_gp(self, gp_name='z', gp_unit='m**2 s**-2', gph_name='z_height'):
        g = 9.80665
        if self.ds[gp_name].attrs['units'] != gp_unit:
            raise ValueError('Geopotential unit should be {} not {}'.format(gp_unit, self.ds[gp_name].attrs['units']))
        self.ds[gph_name] = (self.ds.variables[gp_name].dims, self.ds.variables[gp_name].data / g,
                             {'units': 'm', 'long_name': 'Geopotential Height', 'standard_name': 'geopotential height',
                              'history': 'Calculated from {} with g={}'.format(gp_name, g)})
        logger.info('Calculating GPH from GP... DONE')

    def calc_mean(self, variable):
        if not variable:
            return self['z'].mean(dim="time")
        if variable not in self.variables:
            logger.warning("\n Variable '{}' not found. Select from {}.".format(variable, self.variables))
            return None
        return self[variable].mean(dim="time")

    # ---- climatology / anomaly (contrack.py:458-581) on the device: SURVEY.md section 8(f) row N2 ------------------------
    def _group_ids(self, groupby):
        """(ids per timestep in 0..G-1, the G group values in ascending order) for time.<groupby> (dayofyear, month, ...)"""
        t = self.ds[self._time_name]
        try:
            vals = np.asarray(getattr(t.dt, groupby))
        except (AttributeError, TypeError):
            import pandas as pd
            vals = np.asarray(getattr(pd.DatetimeIndex(np.asarray(t.data)), groupby))
        uniq, ids = np.unique(vals, return_inverse=True)
        return ids.astype(np.int32), uniq

    def _slab_tll(self, variable):
        da = self.ds[variable]
        dims = tuple(da.dims)
        sort = [dims.index(d) for d in (self._time_name, self._latitude_name, self._longitude_name)]
        return np.asarray(da.data).transpose(sort), dims, sort

    def _wrap(self, like, data, dims, coords=None, attrs=None):
        """a labelled array of the same class as `like` (xarray.DataArray, or whatever duck-typed dataset is wrapped)"""
        try:                                         # (the climatology keeps the variable's name, as xarray's groupby().mean() does)
            return type(like)(data, dims=dims, coords=coords, attrs=attrs or {}, name=getattr(like, "name", None))
        except TypeError:
            return type(like)(data, dims=dims, coords=coords, attrs=attrs or {})

    def calc_clim(self, variable, window=1, groupby='dayofyear'):
        """climatological mean per `groupby` value, smoothed with a centred running mean over `window` groups; NaNs of the
        running mean (both ends of the axis) are replaced by the mean of the last `window` groups, as the reference does"""
        slab, dims, sort = self._slab_tll(variable)
        ids, uniq = self._group_ids(groupby)
        if slab.dtype.kind != "f":
            slab = slab.astype(np.float64)
        _, clim = _tracker().anomalies(slab, ids, len(uniq), window=window, smooth=1, want_anom=False, want_clim=True)
        da = self.ds[variable]
        coords = {groupby: uniq}
        for name in (self._latitude_name, self._longitude_name):
            coords[name] = np.asarray(self.ds[name].data)
        return self._wrap(da, clim, (groupby, self._latitude_name, self._longitude_name), coords)

    def calc_anom(self, variable, window=1, smooth=1, groupby='dayofyear', clim=None):
        """adds the variable 'anom': departure of `variable` from its climatology (calc_clim, or the one given as `clim`:
        a labelled array over `groupby` on this grid, or the path of one), smoothed with a centred running mean over `smooth`
        timesteps.  The slab also stays resident on the GPU: a following run_contrack(variable='anom') starts from HBM."""
        self._ensure_set_up()
        slab, dims, sort = self._slab_tll(variable)
        if slab.dtype.kind != "f":
            slab = slab.astype(np.float64)
        ids, uniq = self._group_ids(groupby)
        clim_arr = None
        if clim is None:
            logger.info('Calculating climatological mean from {}...'.format(variable))
            clim_txt = 'from {} with running window time steps {}'.format(variable, window)
        else:
            logger.info('Reading climatological mean from {}...'.format(clim))
            clim_txt = clim
            clim_mean = _xr().open_dataarray(clim) if isinstance(clim, str) else clim
            if groupby not in clim_mean.dims:
                raise ValueError("the climatology must have the dimension {!r}".format(groupby))
            if hasattr(clim_mean, "reindex"):       # regrid to this grid (nearest neighbour), as the reference does
                clim_mean = clim_mean.reindex(**{self._latitude_name: self.ds[self._latitude_name],
                                                 self._longitude_name: self.ds[self._longitude_name]}, method='nearest')
            cd = tuple(clim_mean.dims)
            carr = np.asarray(clim_mean.data).transpose([cd.index(d) for d in (groupby, self._latitude_name, self._longitude_name)])
            cvals = np.asarray(clim_mean[groupby].data if hasattr(clim_mean[groupby], "data") else clim_mean[groupby])
            pos = {v: i for i, v in enumerate(cvals.tolist())}
            missing = [v for v in uniq.tolist() if v not in pos]
            if missing:
                raise ValueError("the climatology has no {} {} (present in {!r}); it covers {}..{}".format(
                    groupby, missing[:5] + (['...'] if len(missing) > 5 else []), variable, cvals.min(), cvals.max()))
            clim_arr = np.stack([carr[pos[v]] for v in uniq.tolist()])       # one plane per group value present in the data
        anom, _ = _tracker().anomalies(slab, ids, len(uniq), window=window, smooth=smooth, clim=clim_arr, keep_resident=True)
        da = self.ds[variable]
        attrs = {'units': da.attrs['units'], 'long_name': da.attrs['long_name'] + ' Anomaly',
                 'standard_name': da.attrs['long_name'] + ' anomaly',
                 'history': ' '.join(['Calculated from {} with input attributes:', 'smoothing time steps = {},',
                                      'climatology = {}.']).format(variable, smooth, clim_txt)}
        anom.flags.writeable = False                         # (its twin stays in HBM for run_contrack: see _fingerprint)
        out = anom.transpose(np.argsort(sort))
        self.ds['anom'] = (dims, out, attrs)
        # (fingerprint of the host twin, identity of the slab in HBM: another instance's calc_anom on the shared handle changes
        # the second and this instance's run_contrack goes back to its own host array)
        self._anom_resident = (_fingerprint(np.asarray(self.ds['anom'].data)), _tracker().resident_generation())
        logger.info('Calculating Anomaly... DONE')

    def _resident_for(self, variable, arr, shape, is_f64):
        """True if `variable` is this instance's anomaly and its twin is still the slab resident in HBM"""
        res = getattr(self, "_anom_resident", None)
        if variable != 'anom' or res is None or res[0] is None:
            return False
        trk = _tracker()
        return res[0] == _fingerprint(np.asarray(arr)) and res[1] == trk.resident_generation() and \
            trk.resident_anom() == (shape[0], shape[1], shape[2], bool(is_f64))

    def percentile_threshold(self, variable='anom', q=0.90, lat_bounds=(50, 80)):
        """the more objective threshold of the reference's README (README.rst:150-151):
        block[variable].sel(latitude=band).quantile([q], dim='time').mean() -- the mean over the latitude band of the
        per-grid-point q-quantile over time.  Evaluated on the GPU (exact order statistics, numpy's linear interpolation)."""
        self._ensure_set_up()
        slab, dims, sort = self._slab_tll(variable)
        lat = np.asarray(self.ds[self._latitude_name].data, dtype=np.float64)
        rows = np.nonzero((lat >= min(lat_bounds)) & (lat <= max(lat_bounds)))[0]
        if len(rows) == 0 or not np.array_equal(rows, np.arange(rows[0], rows[-1] + 1)):
            raise ValueError("latitude band {} selects no contiguous rows".format(lat_bounds))
        if slab.dtype.kind != "f":
            slab = slab.astype(np.float64)
        resident = self._resident_for(variable, self.ds['anom'].data if variable == 'anom' else None, slab.shape, slab.dtype != np.float32)
        return _tracker().percentile(None if resident else slab, int(rows[0]), int(rows[-1]) + 1, q)

    # ---- the hot path (contrack.py:583-796) -----------------------------------------------------------------
    def _dayofyear(self):
        t = self.ds[self._time_name]
        try:
            return np.asarray(t.dt.dayofyear)
        except AttributeError:
            v = np.asarray(t.data).astype('datetime64[D]')
            return (v - v.astype('datetime64[Y]')).astype(int) + 1

    def _thresholds_per_step(self, threshold, T, dtype):
        """scalar, or a 1-D DataArray over 'dayofyear' (contrack.py:648-661) -> per-timestep values"""
        if hasattr(threshold, "dims") and hasattr(threshold, "data"):
            values = np.asarray(threshold.data)
            if 'dayofyear' in getattr(threshold, "dims", ()):
                coord = None
                try:
                    coord = np.asarray(threshold['dayofyear'].data)
                except Exception:
                    pass
                doy = self._dayofyear()
                if coord is None:
                    coord = np.arange(1, len(values) + 1)
                pos = {int(d): i for i, d in enumerate(coord)}
                values = np.array([values[pos[int(d)]] for d in doy], dtype=values.dtype)
            return prepare_thresholds(values, T, dtype)
        return prepare_thresholds(threshold, T, dtype)

    def run_contrack(self, variable, threshold, gorl, overlap, persistence, twosided=True, chunk_steps=None):
        """Spatial and temporal tracking of closed contours; adds the integer variable 'flag' to the dataset.

        variable: name of the input field; threshold: number or 1-D DataArray over 'dayofyear'; gorl: one of
        [>, >=, <, <=, ge, le, gt, lt]; overlap: fraction [0-1] of area overlap between consecutive steps;
        persistence: minimum life time in time steps; twosided: forward+backward overlap test (True) or forward
        only.
        chunk_steps (extension, not in the reference): stream the variable through the GPU in slices of that many time steps
        (0: about 256 MB each) instead of materialising it on the host and in HBM -- for a lazily loaded netCDF variable
        (xr.open_dataset) each slice is read from the file when its turn comes (`isel(time=slice)`), and the device holds four
        slices and the bit mask instead of twice the slab.  Same result."""
        logger.info("\nRun ConTrack \n########### \n    threshold:    {} {} \n    overlap:      {} \n"
                    "    persistence:  {} time steps".format(gorl, threshold, overlap, persistence))
        self._ensure_set_up()
        logger.info("Find individual contours...")
        if gorl not in _native.CMP_OPS:
            raise ValueError(_native.GORL_ERRMSG)
        da = self.ds[variable]
        dims = tuple(da.dims)
        sort = [dims.index(d) for d in (self._time_name, self._latitude_name, self._longitude_name)]
        lat = self.ds[self._latitude_name].data
        wrow = row_weights(lat, self._dlat, self._dlon)
        trk = _tracker()
        # threshold, 2-D labelling, overlap filter, 3-D tracking and persistence are ONE call into the library (the reference logs
        # them as it goes through them, contrack.py:646-772)
        logger.info("Apply overlap...")
        logger.info("Apply persistence...")
        if chunk_steps is not None:
            # the variable is read slice by slice and passes through chunk-sized device buffers
            flag, n_tracked = self._run_streaming(trk, da, dims, sort, threshold, gorl, wrow, overlap, persistence, twosided, int(chunk_steps))
            slab = None
        else:
            slab = np.asarray(da.data).transpose(sort)
            thr = self._thresholds_per_step(threshold, slab.shape[0], slab.dtype)
            if self._resident_for(variable, da.data, slab.shape, slab.dtype != np.float32):
                # calc_anom left this very slab in HBM: no host-to-device copy
                flag, n_tracked = trk.track_resident(thr, _native.CMP_OPS[gorl], wrow, overlap, persistence, twosided)
            elif slab.dtype == np.float32:
                flag, n_tracked = trk.track(np.ascontiguousarray(slab), thr, _native.CMP_OPS[gorl], wrow, overlap, persistence, twosided)
            else:
                flag, n_tracked = trk.track(np.ascontiguousarray(slab, dtype=np.float64), thr, _native.CMP_OPS[gorl], wrow, overlap,
                                            persistence, twosided, f64=True)
        if slab is not None and slab.nbytes > (4 << 30):
            trk.release_io()               # a big one-off slab: do not keep 2 x its size allocated on the GPU
        logger.info("Create new variable 'flag'...")
        # scipy.ndimage.label returns int32 labels, int64 from 2^31 - 2 elements on (scipy/_measurements.py:190-199), and with it the
        # reference's 'flag' (contrack.py:687, :751).  The library writes int32 ids (more than 2^31 - 2 ids are an error, never a
        # wrap-around); slabs of that size get the reference's dtype here.
        if flag.size >= INT64_FLAG_FROM:
            flag = flag.astype(np.int64)
        inverse = np.argsort(sort)
        attrs = {'units': 'flag', 'long_name': 'contrack flag', 'standard_name': 'contrack flag',
                 'history': ' '.join(['Calculated from {} with input attributes:', 'threshold = {} {},', 'overlap fraction = {},',
                                      'persistence time steps = {}.', 'twosided = {}']).format(
                     variable, gorl, threshold, overlap, persistence, twosided),
                 'reference': 'https://github.com/steidani/ConTrack'}
        self.ds['flag'] = (dims, flag.transpose(inverse), attrs)
        st = _tracker().stats()
        if st.get("off_fused_path_reason", 0):
            # not an error: the result is the same, the call was slower (exact areas / host resolver)
            why = {1: "the co-occurrence table had to be regrown (host resolver)", 2: "a removal cascade longer than 240 steps (host resolver)",
                   3: "table regrowth and a long removal cascade (host resolver)",
                   4: "{} overlap decisions on rounding boundaries were re-evaluated with numpy-order sums".format(st.get("exact_fixups", 0))}
            logger.info("run_contrack left the fused device path: " + why.get(st["off_fused_path_reason"], "host resolver"))
        logger.info("Running contrack... DONE\n{} contours tracked".format(n_tracked))

    def _run_streaming(self, trk, da, dims, sort, threshold, gorl, wrow, overlap, persistence, twosided, chunk_steps):
        """run_contrack with the variable read slice by slice"""
        shape = tuple(da.shape[i] for i in sort)                                  # (time, lat, lon)
        dtype = np.dtype(np.float32) if np.dtype(da.dtype) == np.float32 else np.dtype(np.float64)
        thr = self._thresholds_per_step(threshold, shape[0], dtype)
        tname = self._time_name

        def reader(t0, nt, out):
            part = da.isel(**{tname: slice(t0, t0 + nt)}) if hasattr(da, "isel") else None
            arr = np.asarray(part.data if part is not None else np.asarray(da.data).take(range(t0, t0 + nt), axis=dims.index(tname)))
            out[...] = arr.transpose(sort)
        return trk.track_stream(reader, thr, _native.CMP_OPS[gorl], wrow, overlap, persistence, twosided, shape=shape, dtype=dtype,
                                chunk_steps=chunk_steps)

    # ---- life cycle (contrack.py:798-906), consumer of `flag` ----------------------------
    def _time_labels(self):
        """'%Y%m%d_%H' per timestep (contrack.py:862)"""
        t = self.ds[self._time_name]
        try:
            return [str(v) for v in np.asarray(t.dt.strftime('%Y%m%d_%H').values)]
        except (AttributeError, TypeError):
            vals = np.asarray(t.data)
            if vals.dtype.kind == "M":
                import pandas as pd
                return [pd.Timestamp(v).strftime('%Y%m%d_%H') for v in vals]
            return [str(v) for v in vals]

    def run_lifecycle(self, flag, variable):
        """Intensity, size and centre of mass of every flagged contour at every time step.

        flag: name of the flag variable (output of run_contrack); variable: field used for intensity and centre of
        mass.  Returns a pandas DataFrame ['Flag', 'Date', 'Longitude', 'Latitude', 'Intensity', 'Size'] sorted by
        (Flag, Date) -- one row per (time step, flag id).

        The per-(time step, id) sums run on the GPU (ctk_lifecycle_*, include/contrack_hip.h); the divisions,
        int() truncations, coordinate look-ups and rounding of contrack.py:876-901 are done here on the few
        resulting rows."""
        import pandas as pd
        logger.info("\nRun Lifecycle \n########### \n    flag:    {}\n    variable:    {}".format(flag, variable))
        self._ensure_set_up()
        names = (self._time_name, self._latitude_name, self._longitude_name)

        def slab(name):
            da = self.ds[name]
            return np.asarray(da.data).transpose([tuple(da.dims).index(d) for d in names])

        flags, field = slab(flag), slab(variable)
        if flags.dtype.kind not in "iub":
            raise ValueError("flag variable {!r} is not an integer field".format(flag))
        if flags.size and (flags.dtype.itemsize > 4 or flags.dtype == np.uint32) and \
                (flags.max() > np.iinfo(np.int32).max or flags.min() < np.iinfo(np.int32).min):      # (nothing to check for int32 and narrower)
            raise ValueError("flag ids beyond int32")
        if field.dtype != np.float64:
            field = field.astype(np.float32, copy=False)
        lat = np.asarray(self.ds[self._latitude_name].data)
        lon = np.asarray(self.ds[self._longitude_name].data)
        wrow = row_weights(lat, self._dlat, self._dlon)                                                 # contrack.py:847-848
        trk = _tracker()
        resident = self._resident_for(variable, self.ds['anom'].data if variable == 'anom' else None, field.shape, field.dtype == np.float64)
        # (the anomaly slab calc_anom left in HBM: only the flags cross PCIe)
        rows = trk.lifecycle(flags, None, wrow, resident_f64=field.dtype == np.float64) if resident else trk.lifecycle(flags, field, wrow)
        return pd.DataFrame(lifecycle_columns(rows, lat, lon, self._time_labels(), _tracker()),
                            columns=['Flag', 'Date', 'Longitude', 'Latitude', 'Intensity', 'Size'])

    # ---- utility (contrack.py:912-949) ---------------------------------------------------------------------------
    def greatcircle_dist(self, lon1, lat1, lon2, lat2):
        """great-circle distance in km between (lon1, lat1) and (lon2, lat2) given in degrees"""
        a1, a2 = np.deg2rad(lat1), np.deg2rad(lat2)
        c = np.sin(a1) * np.sin(a2) + np.cos(a1) * np.cos(a2) * np.cos(np.deg2rad(lon1 - lon2))
        return 6371. * np.arccos(min(1., max(-1., c)))
