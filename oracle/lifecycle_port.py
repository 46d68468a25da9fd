"""TEST INFRASTRUCTURE -- scipy / numpy port of contrack.run_lifecycle (contrack/contrack.py:798-906), the checker for
the ctk_lifecycle_* reductions.  Same call sequence as the reference on plain numpy arrays (np.unique, masked np.sum,
np.roll, ndimage.center_of_mass); pinned against the reference itself by tests/golden/life/*.npz
(tests/golden/make_life_golden.py runs the unmodified reference class).  Only tests/ may import this module.
"""
import numpy as np


def run_lifecycle(flags, field, lat, lon, wrow, dates):
    """flags (T, ny, nx) int, field (T, ny, nx) float, wrow float32 row weights (contrack.py:847-848), dates: one
    label per time step.  Returns the reference's rows sorted by (Flag, Date)."""
    from scipy import ndimage
    wgrid = np.ones((len(lat), len(lon))) * np.asarray(wrow, dtype=np.float32)[:, None]                 # :848
    rows = []
    for i in range(flags.shape[0]):                                                                     # :860
        plane, values = flags[i], field[i]
        ids = np.unique(plane)                                                                          # :865
        for ident in ids[ids != 0]:
            member = plane == ident
            area = np.sum(wgrid[member])                                                                # :874
            intensity = np.sum(wgrid[member] * values[member]) / area                                   # :875-876
            lon_axis = lon
            if ident in plane[:, 0] and ident in plane[:, -1]:                                          # :880
                cols = np.unique(np.nonzero(member)[1])
                shift = cols[np.argmax(np.diff(cols)) + 1]                                              # :883
                plane_r, values_r = np.roll(plane, -shift, axis=1), np.roll(values, -shift, axis=1)     # :884-885
                lon_axis = np.roll(lon, -shift)
                com = ndimage.center_of_mass(values_r * wgrid, plane_r, [ident])                        # :886
            else:
                com = ndimage.center_of_mass(values * wgrid, plane, [ident])                            # :892
            rows.append((int(ident), dates[i], int(lon_axis[int(com[0][1])]), int(lat[int(com[0][0])]),
                         round(intensity, 2), round(area, 2)))
    return sorted(rows, key=lambda r: (r[0], r[1]))                                                     # :906
