"""Oracle: tile extraction, normalisation, CDF matching, median high-pass.

Test infrastructure only.  Follows the reference:
  * `helpers.normalize`        helpers.py:324-344
  * `helpers.compute_cdf`      helpers.py:433-464
  * `helpers.match_cdf`        helpers.py:467-493
  * `Tracker.extract_tile`     tracker.py:494-534
  * `Image.read` cached crop   image.py:180-186, 211-213
  * `Grid.snap_box/snap_xy/inbounds_xy`  raster.py:390-421, 343-388, 313-337
  * `Tracker.initialize_template`        tracker.py:536-561
  * search box                           tracker.py:580-603
"""
import numpy as np
import scipy.ndimage

from . import camera


def normalize(a):
    """helpers.py:344."""
    return (a - a.mean()) * (1 / a.std())


def compute_cdf(a, return_inverse=False):
    """helpers.py:458-464."""
    results = np.unique(a, return_inverse=return_inverse, return_counts=True)
    quantiles = np.cumsum(results[-1]) / a.size
    if return_inverse:
        return results[0], quantiles, results[1]
    return results[0], quantiles


def match_cdf(a, cdf):
    """helpers.py:489-493."""
    _, quantiles, inverse = compute_cdf(a, return_inverse=True)
    values = np.interp(quantiles, cdf[1], cdf[0])
    return values[inverse].reshape(a.shape)


def read_box(frame, box):
    """image.py:211-213: array[box[1]:box[3], box[0]:box[2]]."""
    return frame[box[1] : box[3], box[0] : box[2]]


def extract_tile(frame, box, histogram=None, return_histogram=False, highpass_size=(5, 5), highpass_mode="reflect"):
    """tracker.py:522-534 (`highpass_size` / `highpass_mode`: the entries of Tracker.highpass that :530 hands to
    scipy.ndimage.median_filter)."""
    tile = read_box(frame, box)
    if tile.ndim > 2:
        tile = tile.mean(axis=2)
    tile = normalize(tile)
    if histogram is not None:
        tile = match_cdf(tile, histogram)
    if return_histogram:
        returned_histogram = compute_cdf(tile, return_inverse=False)
    tile_low = scipy.ndimage.median_filter(tile, size=highpass_size, mode=highpass_mode)
    tile -= tile_low
    if return_histogram:
        return tile, returned_histogram
    return tile


def snap_box(uv, size, imgsz):
    """raster.py:414-421 + :372-388 with centers=False, edges=True on an image grid.

    Image grid: xlim = (0, nx), ylim = (0, ny), d = (1, 1) (observer.py:129).
    Raises IndexError like the reference when the box leaves the image.
    """
    halfsize = np.multiply(size, 0.5)
    xy_box = np.vstack((uv - halfsize, uv + halfsize))
    lo = np.zeros(2)
    hi = np.asarray(imgsz, dtype=float)
    inb = np.all((xy_box >= lo) & (xy_box <= hi), axis=1)
    if any(~inb):
        raise IndexError("Box extends beyond grid bounds")
    nxy = np.floor((xy_box - 0.0) / 1.0 + 0.5)
    return (nxy * 1.0 + 0.0).flatten().astype(int)


def initialize_template(frame, cam, mean_xyz, tile_size):
    """tracker.py:536-561.  Returns dict(box, duv, tile, histogram)."""
    uv = camera.xyz_to_uv(cam, np.asarray(mean_xyz, dtype=float)[None, 0:3]).ravel()
    box = snap_box(uv, tile_size, cam[6:8])
    template = {"box": box, "duv": uv - box.reshape(2, -1).mean(axis=0), "uv": uv}
    template["tile"], template["histogram"] = extract_tile(
        frame, box, return_histogram=True
    )
    return template


def search_box(uv, size, kx=3, ky=3):
    """tracker.py:580-595.  Returns the (2, 2) int box [[l, t], [r, b]] (pre-bounds-check).

    NaN in `uv` propagates through min/max; the int cast of NaN is then
    platform-defined garbage that fails the bounds test (tracker.py:597).
    """
    size = np.asarray(size)
    halfsize = size * 0.5
    box = np.vstack((uv.min(axis=0) - halfsize, uv.max(axis=0) + halfsize))
    ncols = ky - (np.diff(box[:, 0]) - size[0])
    if np.all(ncols > 0):
        box[:, 0] += np.hstack((-ncols, ncols)) * 0.5
    nrows = kx - (np.diff(box[:, 1]) - size[1])
    if np.all(nrows > 0):
        box[:, 1] += np.hstack((-nrows, nrows)) * 0.5
    with np.errstate(invalid="ignore"):
        return np.vstack((np.floor(box[0, :]), np.ceil(box[1, :]))).astype(int)


# ---- The restatement the HIP kernels implement (validated against the above) ----


def gray_key(frame_tile):
    """Integer sort key of a uint8 tile: the pixel (gray) or the channel sum (RGB).

    `tile.mean(axis=2)` (tracker.py:524) is sum/3 in float64, strictly monotone in
    the integer channel sum, so a histogram over the integer key has the same
    bins, in the same order, as `np.unique` over the float values.
    """
    if frame_tile.ndim > 2:
        return frame_tile.astype(np.int32).sum(axis=2)
    return frame_tile.astype(np.int32)


def median5x5_int(key):
    """Rank-12-of-25 selection with edge-repeating ('reflect') padding on integers."""
    return scipy.ndimage.median_filter(key, size=(5, 5), mode="reflect")


def template_from_key(key, channels):
    """Template tile + CDF via the integer-key formulation.

    normalize is affine increasing, so median(normalize(x)) == normalize(median(x)).
    """
    x = key.astype(float) / 3 if channels == 3 else key.astype(float)
    mean = x.mean()
    inv_std = 1 / x.std()
    nbins = 766 if channels == 3 else 256
    counts = np.bincount(key.ravel(), minlength=nbins)
    present = np.nonzero(counts)[0]
    xv = present.astype(float) / 3 if channels == 3 else present.astype(float)
    values = (xv - mean) * inv_std
    quantiles = np.cumsum(counts[present]) / key.size
    med = median5x5_int(key)
    xm = med.astype(float) / 3 if channels == 3 else med.astype(float)
    tile = (x - mean) * inv_std - (xm - mean) * inv_std
    return tile, (values, quantiles), mean, inv_std


def search_from_key(key, channels, histogram):
    """Search tile via 256/766-bin LUT + integer median.

    `normalize` before `match_cdf` is a no-op (it is monotone, and match_cdf only
    uses ranks), so the LUT maps raw integer key -> matched template value.
    """
    nbins = 766 if channels == 3 else 256
    counts = np.bincount(key.ravel(), minlength=nbins)
    present = np.nonzero(counts)[0]
    q = np.cumsum(counts[present]) / key.size
    lut = np.full(nbins, np.nan)
    lut[present] = np.interp(q, histogram[1], histogram[0])
    med = median5x5_int(key)
    return lut[key] - lut[med]
