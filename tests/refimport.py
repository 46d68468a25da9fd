"""Import the UNMODIFIED reference module from /root/reference (build container only).

Used by tests/golden/make_golden.py and by the optional live cross-checks in
tests/test_oracle_vs_reference.py (skipped when /root/reference is absent, e.g. on the GPU box).
"""
import logging
import os
import sys
import warnings

REF_ROOT = "/root/reference"


def available():
    return os.path.exists(os.path.join(REF_ROOT, "contrack", "contrack.py"))


def load():
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import minixr
    minixr.install_as_xarray()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lvl = logging.root.manager.disable
        logging.disable(logging.CRITICAL)
        try:
            import contrack as refpkg
        finally:
            logging.disable(lvl)
    return refpkg.contrack


def run_reference(anom, lat, lon, threshold, gorl, overlap, persistence, twosided=True, force=False, time=None):
    """Run the reference's run_contrack on a (T,ny,nx) numpy slab; returns the int flag array."""
    import numpy as np
    import minixr
    cls = load()
    ds = minixr.make_dataset(anom, lat, lon) if time is None else minixr.make_dataset(anom, lat, lon, time=time)
    c = cls()
    c.read_xarray(ds)
    logging.disable(logging.CRITICAL)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if force:
                c.set_up(force=True)
            c.run_contrack(variable="anom", threshold=threshold, gorl=gorl, overlap=overlap,
                           persistence=persistence, twosided=twosided)
    finally:
        logging.disable(logging.NOTSET)
    return np.asarray(c.ds["flag"].data), c
