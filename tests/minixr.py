"""Tiny labelled-array stand-in used ONLY by the tests and the golden-vector generator.

The real `xarray` package is not installed in the build container nor on the GPU box, and
cannot be installed (no network).  The `contrack_amd.contrack` class is duck-typed against the
small part of the xarray API that the hot path touches (Dataset.__getitem__/__setitem__, .dims,
.data_vars, DataArray.dims/.data/.attrs); this module provides just that surface so that

  * tests/golden/make_golden.py can import the *unmodified* reference module from
    /root/reference (which does `import xarray as xr` at import time) and run it on numpy data,
  * tests/test_class_api.py can exercise our drop-in class without xarray.

It is test infrastructure: nothing in `contrack_amd/` imports it.
"""
import sys
import types
import numpy as np


class DataArray:
    def __init__(self, data, dims=None, coords=None, attrs=None, name=None):
        self.data = np.asarray(data)
        self.dims = tuple(dims) if dims is not None else tuple("dim_%d" % i for i in range(self.data.ndim))
        self.coords = dict(coords or {})
        self.attrs = dict(attrs or {})
        self.encoding = {}
        self.name = name

    # numpy protocol -------------------------------------------------------
    def __array__(self, dtype=None, copy=None):
        return self.data if dtype is None else self.data.astype(dtype)

    @property
    def values(self):
        return self.data

    @property
    def shape(self):
        return self.data.shape

    @property
    def dtype(self):
        return self.data.dtype

    def __len__(self):
        return len(self.data)

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.coords[key]
        return DataArray(self.data[key], self.dims[-self.data[key].ndim:] if self.data[key].ndim else ())

    def _cmp(self, other, op):
        other = other.data if isinstance(other, DataArray) else other
        return DataArray(op(self.data, other), self.dims, self.coords)

    def __ge__(self, o):
        return self._cmp(o, np.greater_equal)

    def __le__(self, o):
        return self._cmp(o, np.less_equal)

    def __gt__(self, o):
        return self._cmp(o, np.greater)

    def __lt__(self, o):
        return self._cmp(o, np.less)

    def to_index(self):
        if self.data.dtype.kind != "M":
            raise AttributeError("to_index")          # like dates xarray cannot decode
        return _Index(np.asarray(self.data))

    def transpose(self, *dims):
        if len(dims) == 1 and not isinstance(dims[0], str):
            dims = tuple(dims[0])
        order = [self.dims.index(d) for d in dims]
        return DataArray(self.data.transpose(order), dims, self.coords, self.attrs)

    def mean(self, dim=None):
        ax = self.dims.index(dim)
        return DataArray(self.data.mean(axis=ax), tuple(d for d in self.dims if d != dim))

    # the part of the API the reference's run_lifecycle touches (contrack.py:862-895) ----------------------
    def isel(self, **indexers):
        data, dims, coords = self.data, list(self.dims), dict(self.coords)
        for name, i in indexers.items():
            ax = dims.index(name)
            if isinstance(i, slice):                      # keeps the dimension
                data = data[(slice(None),) * ax + (i,)]
                if name in coords:
                    coords[name] = DataArray(np.asarray(coords[name].data)[i], (name,))
                continue
            data = np.take(data, i, axis=ax)
            dims.pop(ax)
            if name in coords:
                coords[name] = DataArray(np.take(coords[name].data, i, axis=0), ())
        return DataArray(data, dims, coords, self.attrs)

    def roll(self, roll_coords=False, **shifts):
        data, coords = self.data, dict(self.coords)
        for name, k in shifts.items():
            data = np.roll(data, k, axis=self.dims.index(name))
            if roll_coords and name in coords:
                coords[name] = DataArray(np.roll(coords[name].data, k), (name,), attrs=coords[name].attrs)
        return DataArray(data, self.dims, coords, self.attrs)

    # DataArray.groupby('time.dayofyear') <op> threshold-per-day-of-year: the reference's DataArray-threshold branch of
    # run_contrack (contrack.py:648-661) and nothing more
    def groupby(self, spec):
        name, _, what = spec.partition(".")
        if what != "dayofyear" or name not in self.dims:
            raise NotImplementedError("minixr groupby: only '<time dimension>.dayofyear' (%r)" % (spec,))
        return _GroupByDayOfYear(self, name)

    def reset_coords(self, names=None, drop=False):
        if not drop:
            raise NotImplementedError("minixr reset_coords: drop=True only")
        names = [names] if isinstance(names, str) else list(names or [])
        return DataArray(self.data, self.dims, {k: v for k, v in self.coords.items() if k not in names}, self.attrs, self.name)

    @property
    def dt(self):
        if self.data.dtype.kind != "M":
            raise TypeError("'.dt' accessor only available for DataArray with datetime64 timedelta64 dtype")
        return _DatetimeAccessor(self)


class _Index:
    """what DataArray.to_index() returns here: `.values`, slicing, and a difference that is a plain numpy timedelta64 array --
    the unmodified reference takes `(idx[1:] - idx[:-1]).astype('timedelta64[h]')` (contrack.py:335-339), which pandas >= 2
    refuses on its own TimedeltaIndex while numpy (and the pandas the reference was written against) accepts it"""

    def __init__(self, values):
        self.values = np.asarray(values)

    def __len__(self):
        return len(self.values)

    def __getitem__(self, key):
        v = self.values[key]
        return _Index(v) if isinstance(key, slice) else v

    def __sub__(self, other):
        return self.values - (other.values if isinstance(other, _Index) else other)

    def __iter__(self):
        return iter(self.values)


class _GroupByDayOfYear:
    """what `da.groupby('time.dayofyear') >= thr` evaluates in xarray: every time step compared with the threshold of ITS day of
    year (binary op between a groupby object and a DataArray indexed by the group label); the result carries a 'dayofyear'
    coordinate along time, which the reference drops again (reset_coords, contrack.py:661)"""

    def __init__(self, da, time_name):
        self._da, self._time = da, time_name
        tc = da.coords.get(time_name)
        if tc is None or np.asarray(tc.data).dtype.kind != "M":
            raise TypeError("groupby('%s.dayofyear') needs a datetime64 coordinate" % time_name)
        import pandas as pd
        self._doy = np.asarray(pd.DatetimeIndex(np.asarray(tc.data)).dayofyear)

    def _cmp(self, thr, op):
        if not isinstance(thr, DataArray) or thr.dims != ("dayofyear",):
            raise TypeError("the threshold must be a DataArray over 'dayofyear'")
        labels = np.asarray(thr.coords["dayofyear"].data if isinstance(thr.coords["dayofyear"], DataArray) else thr.coords["dayofyear"])
        pos = {int(v): i for i, v in enumerate(labels.tolist())}
        try:
            idx = np.array([pos[int(d)] for d in self._doy])
        except KeyError as e:
            raise KeyError("dayofyear %s is not in the threshold" % e)
        ax = self._da.dims.index(self._time)
        shape = [1] * self._da.data.ndim
        shape[ax] = -1
        per_step = np.asarray(thr.data)[idx].reshape(shape)              # (numpy promotes float32 data vs float64 thresholds to float64)
        coords = dict(self._da.coords)
        coords["dayofyear"] = DataArray(self._doy, (self._time,))
        return DataArray(op(self._da.data, per_step), self._da.dims, coords)

    def __ge__(self, o):
        return self._cmp(o, np.greater_equal)

    def __le__(self, o):
        return self._cmp(o, np.less_equal)

    def __gt__(self, o):
        return self._cmp(o, np.greater)

    def __lt__(self, o):
        return self._cmp(o, np.less)


class _DatetimeAccessor:
    def __init__(self, da):
        self._da = da

    def strftime(self, fmt):
        import pandas as pd
        flat = [pd.Timestamp(v).strftime(fmt) for v in np.asarray(self._da.data).reshape(-1)]
        return DataArray(np.array(flat, dtype=object).reshape(self._da.data.shape), self._da.dims)

    @property
    def dayofyear(self):
        import pandas as pd
        return DataArray(pd.DatetimeIndex(np.asarray(self._da.data).reshape(-1)).dayofyear.values, self._da.dims)


class Variable(DataArray):
    def __init__(self, dims, data, attrs=None):
        super().__init__(data, dims, attrs=attrs)


class _Dims(dict):
    """xarray's Dataset.dims: iterates names, maps name -> length."""


class Dataset:
    def __init__(self, data_vars=None, coords=None, attrs=None):
        self._coords = {}
        self._vars = {}
        self.attrs = dict(attrs or {})
        for k, v in (coords or {}).items():
            self._coords[k] = self._as_da(v, default_dims=(k,))
        for k, v in (data_vars or {}).items():
            self[k] = v

    @staticmethod
    def _as_da(v, default_dims=None):
        if isinstance(v, DataArray):
            return v
        if isinstance(v, tuple):
            dims, data = v[0], v[1]
            attrs = v[2] if len(v) > 2 else None
            if isinstance(dims, str):
                dims = (dims,)
            return DataArray(data, dims, attrs=attrs)
        return DataArray(v, default_dims)

    @property
    def dims(self):
        d = _Dims()
        for da in self._vars.values():
            for n, s in zip(da.dims, da.data.shape):
                d[n] = s
        for k, da in self._coords.items():
            d.setdefault(k, da.data.shape[0])
        # xarray sorts dims alphabetically for a Dataset opened from netCDF
        return _Dims(sorted(d.items()))

    @property
    def data_vars(self):
        return dict(self._vars)

    @property
    def variables(self):
        out = dict(self._coords)
        out.update(self._vars)
        return out

    @property
    def coords(self):
        return dict(self._coords)

    def __len__(self):
        return len(self._vars)

    def __contains__(self, k):
        return k in self._vars or k in self._coords

    def __getitem__(self, key):
        if key in self._vars:
            da = self._vars[key]
            da.coords = {d: self._coords[d] for d in da.dims if d in self._coords}
            return da
        return self._coords[key]

    def __setitem__(self, key, value):
        da = self._as_da(value)
        da.name = key
        self._vars[key] = da

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)


def where(cond, x, y):
    if isinstance(cond, DataArray):
        return DataArray(np.where(cond.data, x, y), cond.dims, cond.coords)
    return np.where(cond, x, y)


def open_dataset(filename, **kwargs):
    raise OSError("minixr cannot read files: %r" % (filename,))


def install_as_xarray():
    """Register this module as `xarray` if the real package is absent. Returns True if installed."""
    try:
        import xarray  # noqa: F401
        return False
    except ImportError:
        pass
    me = sys.modules[__name__]
    mod = types.ModuleType("xarray")
    for n in ("DataArray", "Dataset", "Variable", "where", "open_dataset"):
        setattr(mod, n, getattr(me, n))
    core = types.ModuleType("xarray.core")
    dsm = types.ModuleType("xarray.core.dataset")
    dsm.Dataset = Dataset
    core.dataset = dsm
    mod.core = core
    sys.modules["xarray"] = mod
    sys.modules["xarray.core"] = core
    sys.modules["xarray.core.dataset"] = dsm
    return True


def make_dataset(anom, lat, lon, time=None, var="anom", dims=("time", "latitude", "longitude"),
                 time_units="days since 2000-01-01"):
    """Build a Dataset shaped like the reference's test file (time, latitude, longitude)."""
    anom = np.asarray(anom)
    names = {"time": dims[0], "latitude": dims[1], "longitude": dims[2]}
    if time is None:
        time = np.arange(anom.shape[0], dtype=np.int64)
    coords = {
        names["time"]: DataArray(np.asarray(time), (names["time"],), attrs={"units": time_units}),
        names["latitude"]: DataArray(np.asarray(lat), (names["latitude"],), attrs={"units": "degrees_north"}),
        names["longitude"]: DataArray(np.asarray(lon), (names["longitude"],), attrs={"units": "degrees_east"}),
    }
    ds = Dataset(coords=coords)
    ds[var] = DataArray(anom, dims, attrs={"units": "m", "long_name": "Z500 anomaly"})
    return ds
