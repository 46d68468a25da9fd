"""Deterministic synthetic Z500-anomaly-like slabs.

White noise -> Gaussian filter sigma = (2 steps, 6 deg, 6 deg) with longitude wrap -> std 100 -> +offset.
Used by the measurement scripts and the tests; not on the hot path.
"""
import numpy as np


def grid(ny, nx):
    """ERA5-style regular grid: lat 90..-90 (float32), lon 0..360-dlon (float32)."""
    lat = np.linspace(90.0, -90.0, ny, dtype=np.float32)
    lon = (np.arange(nx, dtype=np.float32) * np.float32(360.0 / nx)).astype(np.float32)
    return lat, lon


def smooth_field(T, ny, nx, seed=0, offset=35.0, sigma_t=2.0, sigma_deg=6.0, chunk=None):
    """float32 (T,ny,nx) slab; ~10 % of pixels exceed 160 with the default offset."""
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    sig_y = sigma_deg * (ny - 1) / 180.0
    sig_x = sigma_deg * nx / 360.0
    out = np.empty((T, ny, nx), dtype=np.float32)
    # filter in overlapping time chunks so that huge slabs do not need a second full-size temporary
    pad = int(4 * sigma_t + 1)
    if chunk is None:
        chunk = max(16, min(T, (1 << 26) // max(1, ny * nx)))
    noise_prev = None
    t = 0
    # generate noise plane-by-plane deterministically: one stream, consumed in order
    noise = rng.standard_normal((T + 2 * pad, ny, nx), dtype=np.float32) if T * ny * nx <= (1 << 28) else None
    if noise is not None:
        f = ndimage.gaussian_filter(noise, sigma=(sigma_t, sig_y, sig_x), mode=("nearest", "nearest", "wrap"))
        out[:] = f[pad:pad + T]
    else:
        # streaming variant: chunked along time with `pad` planes of halo on both sides
        buf = rng.standard_normal((min(T, chunk) + 2 * pad, ny, nx), dtype=np.float32)
        while t < T:
            n = min(chunk, T - t)
            if buf.shape[0] != n + 2 * pad:
                buf = buf[:n + 2 * pad]
            f = ndimage.gaussian_filter(buf, sigma=(sigma_t, sig_y, sig_x), mode=("nearest", "nearest", "wrap"))
            out[t:t + n] = f[pad:pad + n]
            t += n
            if t < T:
                n2 = min(chunk, T - t)
                nb = np.empty((n2 + 2 * pad, ny, nx), dtype=np.float32)
                nb[:2 * pad] = buf[n:n + 2 * pad]
                nb[2 * pad:] = rng.standard_normal((n2, ny, nx), dtype=np.float32)
                buf = nb
        del noise_prev
    s = out.std(dtype=np.float64)
    out *= np.float32(100.0 / s)
    out += np.float32(offset)
    return out
