"""CPU oracle for the pre-sweep image preparation that is scikit-image in the reference.

TEST INFRASTRUCTURE ONLY.  Nothing under ``helicon_amd/`` may import this module; it is imported by ``tests/`` only,
as the checker of ``helicon_amd.transform_image`` / ``rescale`` / ``down_scale`` /
``estimate_helix_rotation_center_diameter`` / ``auto_horizontalize`` and of ``helicon_amd.solver``'s non-cosine scores.

PARITY PINNED BY DERIVATION.  scikit-image is a dependency of the reference (pyproject.toml:17, unpinned) that is NOT
installed in the build container, so the reference's own functions for this row cannot be run here and no reference
output exists to pin against.  What this module restates, and from where:

* ``transform_image``      reference: src/helicon/lib/transforms.py:238-312 — composes ``AffineTransform`` objects and
  calls ``skimage.transform.warp(image, xform.inverse, mode, order)``.  scikit-image 0.25: ``AffineTransform(scale,
  rotation, translation).params`` = [[sx cos r, -sy sin r, tx], [sx sin r, sy cos r, ty], [0, 0, 1]]; ``a + b`` =
  ``b.params @ a.params``; ``warp`` of a 2-D image with a homography and order in 0..3 goes through the Cython loop
  ``_warp_fast`` (transform/_warps_cy.pyx) instantiated for the image's floating type with the matrix cast to that
  type: output pixel (r, c) samples the input at H (c, r, 1); order 1 = ``bilinear_interpolation`` (interpolation.pxd:
  corners at floor and ceil, out-of-image corners = cval, horizontal mixes in the image type, vertical mix in
  float64); then ``_clip_warp_output``.
* ``rescale`` / ``resize``  scikit-image 0.25 transform/_warps.py: ``output_shape = max(round(scale * shape), 1)``;
  ``factors = shape / output_shape``; with ``anti_aliasing`` ``scipy.ndimage.gaussian_filter(image, max(0, (factors -
  1) / 2), mode=ndi_mode)``; ``scipy.ndimage.zoom(filtered, 1 / factors, order, mode=ndi_mode, grid_mode=True)`` with
  ndi_mode = "mirror" for the default ``mode="reflect"``; clip to the input's range.  Here the two SciPy calls are MADE
  (SciPy 1.15.3 is installed), not restated.
* ``down_scale``            reference: src/helicon/lib/filters.py:375-412 (+ ``pad_to_size`` lib/transforms.py:441-479)
* binning                   reference: src/helicon/webApps/denovo3D/app.py:1911-1922
* ``estimate_helix_rotation_center_diameter``  reference: src/helicon/lib/analysis.py:645-728;
  ``skimage.morphology.closing(mask, mode="ignore")`` = grey dilation then grey erosion with the 3 x 3 cross, the
  outside of the image ignored (dilation: padded with the minimum, erosion: with the maximum), made here with
  ``scipy.ndimage.binary_dilation(border_value=0)`` / ``binary_erosion(border_value=1)``.
* ``rotate_shift_image``    reference: src/helicon/lib/transforms.py:315-369 (SciPy only; fixture G8 / G15 come from the
  reference itself)
* ``auto_horizontalize``    reference: src/helicon/webApps/denovo3D/utils.py:383-424
* ``set_to_periodic_range`` reference: src/helicon/lib/angular.py:84-108
* ``ssim_score`` / ``ms_ssim_score`` / ``mutual_information_score``  reference: src/helicon/lib/analysis.py:487-613 (the
  non-cosine scores of ``lsq_reconstruct``, solver_linear_regression.py:484-524): ``skimage.metrics.structural_similarity``
  (metrics/_structural_similarity.py, scikit-image 0.25, defaults) restated with ``scipy.ndimage.uniform_filter`` as
  scikit-image calls it, ``skimage.metrics.normalized_mutual_information`` with ``np.histogramdd`` and
  ``scipy.stats.entropy`` as scikit-image calls them
"""
from __future__ import annotations

import math

import numpy as np
from scipy import ndimage as ndi


# ---------------------------------------------------------------------------------------------------------------- warp
def _affine_params(scale=(1.0, 1.0), rotation=0.0, translation=(0.0, 0.0)) -> np.ndarray:
    sx, sy = scale
    tx, ty = translation
    return np.array([[sx * math.cos(rotation), -sy * math.sin(rotation), tx],
                     [sx * math.sin(rotation), sy * math.cos(rotation), ty],
                     [0.0, 0.0, 1.0]])


def transform_image_matrix(shape, scale=1.0, rotation=0.0, rotation_center=None, pre_translation=(0.0, 0.0),
                           post_translation=(0.0, 0.0)) -> np.ndarray:
    """The forward 3 x 3 matrix of transforms.py:290-307 ((x, y, 1) columns, as scikit-image keeps it)."""
    if rotation_center is None:
        rotation_center = np.array((shape[0], shape[1])) / 2.0
    elif not isinstance(rotation_center, np.ndarray):
        rotation_center = np.array(rotation_center)
    if isinstance(scale, (int, float)):
        scale = np.array((scale, scale))
    pre = _affine_params(translation=tuple(pre_translation[::-1]))
    to_c = _affine_params(translation=tuple(-rotation_center[::-1]))
    from_c = _affine_params(translation=tuple(rotation_center[::-1]))
    post = _affine_params(translation=tuple(post_translation[::-1]))
    centre = _affine_params(scale=tuple(scale[::-1]), rotation=np.deg2rad(rotation))
    # a + b applies a first: (b.params @ a.params)
    m = pre
    for nxt in (to_c, centre, from_c, post):
        m = nxt @ m
    return m


def warp_affine(image: np.ndarray, inverse_matrix: np.ndarray, order: int = 1, cval: float = 0.0, clip: bool = True) -> np.ndarray:
    """``skimage.transform.warp`` of a 2-D float image through ``_warp_fast`` (mode="constant")."""
    img = np.asarray(image)
    if img.dtype not in (np.float32, np.float64):
        img = img.astype(np.float64)
    T = img.dtype.type
    H = np.asarray(inverse_matrix, dtype=np.float64).astype(img.dtype)
    rows, cols = img.shape
    tfr, tfc = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    x, y = tfc.astype(img.dtype), tfr.astype(img.dtype)
    with np.errstate(all="ignore"):
        c = H[0, 0] * x + H[0, 1] * y + H[0, 2]
        r = H[1, 0] * x + H[1, 1] * y + H[1, 2]
        if not (H[2, 0] == 0 and H[2, 1] == 0 and H[2, 2] == 1):
            z = H[2, 0] * x + H[2, 1] * y + H[2, 2]
            c, r = c / z, r / z

    def pixel(rr, cc):
        inside = (rr >= 0) & (rr < rows) & (cc >= 0) & (cc < cols)
        out = np.full(rr.shape, T(cval), dtype=img.dtype)
        out[inside] = img[rr[inside], cc[inside]]
        return out

    if order == 0:
        rnd = lambda v: np.where(v >= 0, np.floor(v.astype(np.float64) + 0.5), np.ceil(v.astype(np.float64) - 0.5)).astype(np.int64)  # C round()
        out = pixel(rnd(r), rnd(c))
    elif order == 1:
        minr, minc = np.floor(r).astype(np.int64), np.floor(c).astype(np.int64)
        maxr, maxc = np.ceil(r).astype(np.int64), np.ceil(c).astype(np.int64)
        dr, dc = r - minr.astype(img.dtype), c - minc.astype(img.dtype)
        one = T(1)
        top = ((one - dc) * pixel(minr, minc) + dc * pixel(minr, maxc)).astype(np.float64)
        bottom = ((one - dc) * pixel(maxr, minc) + dc * pixel(maxr, maxc)).astype(np.float64)
        out = ((one - dr).astype(np.float64) * top + dr.astype(np.float64) * bottom).astype(img.dtype)
    else:
        raise NotImplementedError("orders 0 and 1")
    if clip:
        lo, hi = img.min(), img.max()
        if not (np.isnan(lo) or np.isnan(hi)):
            preserve = not (lo <= cval <= hi)
            keep = out == T(cval) if preserve else None
            out = np.clip(out, lo, hi)
            if preserve:
                out[keep] = T(cval)
    return out


def transform_image(image, scale=1.0, rotation=0.0, rotation_center=None, pre_translation=(0.0, 0.0),
                    post_translation=(0.0, 0.0), mode="constant", order=1) -> np.ndarray:
    if mode != "constant":
        raise NotImplementedError("mode='constant' (the reference's default and only use)")
    m = transform_image_matrix(image.shape, scale, rotation, rotation_center, pre_translation, post_translation)
    return warp_affine(image, np.linalg.inv(m), order=order)


# ------------------------------------------------------------------------------------------------------------- rescale
def resize(image: np.ndarray, output_shape, order: int = 3, anti_aliasing: bool = True, clip: bool = True) -> np.ndarray:
    img = np.asarray(image)
    if img.dtype not in (np.float32, np.float64):
        img = img.astype(np.float64)
    output_shape = tuple(int(v) for v in output_shape)
    factors = np.divide(img.shape, output_shape)
    lo, hi = img.min(), img.max()
    if anti_aliasing:
        sigma = np.maximum(0, (factors - 1) / 2)
        filtered = ndi.gaussian_filter(img, sigma, cval=0, mode="mirror")
    else:
        filtered = img
    out = ndi.zoom(filtered, [1 / f for f in factors], order=order, mode="mirror", cval=0, grid_mode=True)
    assert out.shape == output_shape, (out.shape, output_shape)
    if clip and not (np.isnan(lo) or np.isnan(hi)):
        out = np.clip(out, lo, hi)
    return out


def rescale(image: np.ndarray, scale: float, order: int = 3, anti_aliasing: bool = True, clip: bool = True) -> np.ndarray:
    shape = np.asarray(np.asarray(image).shape)
    output_shape = np.maximum(np.round(np.atleast_1d(scale) * shape), 1)
    return resize(image, output_shape, order=order, anti_aliasing=anti_aliasing, clip=clip)


def pad_to_size(data: np.ndarray, shape) -> np.ndarray:
    if data.shape == tuple(shape):
        return data
    ny, nx = data.shape
    my, mx = shape
    yb, xb = max(0, (my - ny) // 2), max(0, (mx - nx) // 2)
    return np.pad(data, ((yb, max(0, my - yb - ny)), (xb, max(0, mx - xb - nx))), mode="constant")


def down_scale(data: np.ndarray, target_apix: float, apix_orig: float) -> np.ndarray:
    if target_apix == apix_orig or target_apix < apix_orig:
        return data
    out = rescale(data, apix_orig / target_apix, order=3, anti_aliasing=True)
    ny, nx = out.shape
    return pad_to_size(out, (ny + ny % 2, nx + nx % 2))


def bin_image(image: np.ndarray, binning: int) -> np.ndarray:
    """app.py:1911-1922 (``preserve_range=True`` changes nothing for a floating image)."""
    return rescale(image, 1.0 / binning, order=3, anti_aliasing=True) if binning > 1 else image


# ----------------------------------------------------------------------------------------------------- helix estimates
_CROSS = ndi.generate_binary_structure(2, 1)


def closing_cross(mask: np.ndarray) -> np.ndarray:
    return ndi.binary_erosion(ndi.binary_dilation(mask, _CROSS, border_value=0), _CROSS, border_value=1)


def set_to_periodic_range(v, min=-180, max=180):
    if min <= v <= max:
        return v
    tmp = math.fmod(v - min, max - min)
    return tmp + min if tmp >= 0 else tmp + max


def estimate_helix_rotation_center_diameter(data, estimate_rotation=True, estimate_center=True, threshold=0):
    ny, nx = data.shape

    def weighted(mask, intensity):
        ys, xs = np.where(mask)
        if len(ys) < 2:
            return 0.0, 0.0, ny
        w = intensity[ys, xs].astype(np.float64)
        w = w - w.min() + 1e-8
        cw = w.sum()
        cy, cx = (ys * w).sum() / cw, (xs * w).sum() / cw
        uy, ux = ys - cy, xs - cx
        i_yy, i_xx, i_xy = (uy * uy * w).sum() / cw, (ux * ux * w).sum() / cw, (uy * ux * w).sum() / cw
        angle = np.rad2deg(0.5 * np.arctan2(2.0 * i_xy, i_yy - i_xx)) + 90.0
        if abs(angle) > 90.0:
            angle -= 180.0
        return angle, (ny // 2 - cy) if estimate_center else 0.0, int(ys.max() - ys.min() + 1)

    mask = closing_cross(data > threshold)
    if not mask.any():
        return 0.0, 0.0, ny
    if estimate_rotation:
        rotation, _, _ = weighted(mask, data)
        rotation = set_to_periodic_range(rotation, min=-180, max=180)
        rotated = transform_image(data, rotation=rotation)
    else:
        rotation, rotated = 0.0, data
    mask_rot = closing_cross(rotated > threshold)
    if not mask_rot.any():
        return rotation, 0.0, ny
    _, shift_y, diameter = weighted(mask_rot, rotated)
    return rotation, shift_y, diameter


def rotate_shift_image(data, angle=0, pre_shift=(0, 0), post_shift=(0, 0), rotation_center=None, order=1):
    if angle == 0 and pre_shift == [0, 0] and post_shift == [0, 0]:
        return data * 1.0
    ny, nx = data.shape
    if rotation_center is None:
        rotation_center = np.array((ny // 2, nx // 2), dtype=np.float32)
    ang = np.deg2rad(angle)
    m = np.array([[np.cos(ang), np.sin(ang)], [-np.sin(ang), np.cos(ang)]], dtype=np.float32)
    offset = -np.dot(m, np.array(post_shift, dtype=np.float32).T)
    offset += np.array(rotation_center, dtype=np.float32).T - np.dot(m, np.array(rotation_center, dtype=np.float32).T)
    offset += -np.array(pre_shift, dtype=np.float32).T
    return ndi.affine_transform(data, matrix=m, offset=offset, order=order, mode="constant")


def auto_horizontalize(data, refine=False):
    data_work = np.clip(data, 0, None)
    theta, shift_y, _ = estimate_helix_rotation_center_diameter(data)
    if refine:
        from scipy.optimize import fmin

        def score(x):
            tmp = rotate_shift_image(data_work, angle=x[0], post_shift=(x[1], 0))
            y = np.sum(tmp, axis=1)[1:]
            y += y[::-1]
            return -np.std(y)

        theta, shift_y = fmin(score, x0=(theta, shift_y), xtol=1e-2, disp=0)
    return rotate_shift_image(data, angle=theta, post_shift=(shift_y, 0), order=3), theta, shift_y


# ------------------------------------------------------------------------------------------- non-cosine scores (skimage.metrics)
def structural_similarity(im1, im2, data_range):
    """skimage.metrics.structural_similarity(im1, im2, data_range=...) with its defaults (metrics/_structural_similarity.py,
    scikit-image 0.25: 7 x 7 uniform window, sample covariance, K1 0.01, K2 0.03), in numpy's own float32 arithmetic, the
    uniform filters MADE with scipy.ndimage.uniform_filter as scikit-image makes them."""
    if min(im1.shape) < 7:
        raise ValueError("win_size exceeds image extent")
    float_type = np.float32 if im1.dtype in (np.float16, np.float32) else np.float64
    im1 = im1.astype(float_type, copy=False)
    im2 = im2.astype(float_type, copy=False)
    NP = 49
    cov_norm = NP / (NP - 1)
    ux, uy = ndi.uniform_filter(im1, size=7), ndi.uniform_filter(im2, size=7)
    uxx, uyy, uxy = ndi.uniform_filter(im1 * im1, size=7), ndi.uniform_filter(im2 * im2, size=7), ndi.uniform_filter(im1 * im2, size=7)
    vx = cov_norm * (uxx - ux * ux)
    vy = cov_norm * (uyy - uy * uy)
    vxy = cov_norm * (uxy - ux * uy)
    R = data_range
    C1, C2 = (0.01 * R) ** 2, (0.03 * R) ** 2
    A1, A2, B1, B2 = 2 * ux * uy + C1, 2 * vxy + C2, ux ** 2 + uy ** 2 + C1, vx + vy + C2
    S = (A1 * A2) / (B1 * B2)
    return S[3:-3, 3:-3].mean(dtype=np.float64)


def ssim_score(img1, img2):
    """helicon.ssim_score (lib/analysis.py:487-513)."""
    if img1.shape != img2.shape:
        raise ValueError(f"Image shapes must match: {img1.shape} vs {img2.shape}")
    try:
        data_range = max(img1.max() - img1.min(), img2.max() - img2.min())
        if data_range == 0:
            return 0.0
        return float(structural_similarity(img1, img2, data_range=data_range))
    except Exception:
        return 0.0


def ms_ssim_score(img1, img2):
    """helicon.ms_ssim_score (lib/analysis.py:516-582): SSIM at up to five scales, each half the last (``rescale(img, 0.5,
    anti_aliasing=True)``: linear interpolation for a floating image), combined as a weighted geometric mean."""
    if img1.shape != img2.shape:
        raise ValueError(f"Image shapes must match: {img1.shape} vs {img2.shape}")
    try:
        data_range = max(img1.max() - img1.min(), img2.max() - img2.min())
        if data_range == 0:
            return 0.0
        all_weights = np.array([0.0448, 0.2856, 0.3001, 0.2363, 0.1333])
        vals = []
        for i in range(len(all_weights)):
            h, w = img1.shape
            if h < 8 or w < 8:
                break
            vals.append(max(structural_similarity(img1, img2, data_range=data_range), 0.0))
            if i < len(all_weights) - 1:
                img1, img2 = rescale(img1, 0.5, order=1), rescale(img2, 0.5, order=1)
                data_range = max(img1.max() - img1.min(), img2.max() - img2.min())
                if data_range == 0:
                    break
        if not vals:
            return 0.0
        wts = all_weights[: len(vals)]
        wts = wts / wts.sum()
        result = 1.0
        for s_, w_ in zip(vals, wts):
            result *= s_ ** w_
        return float(result)
    except Exception:
        return 0.0


def mutual_information_score(img1, img2):
    """helicon.mutual_information_score (lib/analysis.py:585-613): skimage.metrics.normalized_mutual_information(bins=64) - 1;
    scikit-image forms it from np.histogramdd and scipy.stats.entropy."""
    from scipy.stats import entropy

    if img1.shape != img2.shape:
        raise ValueError(f"Image shapes must match: {img1.shape} vs {img2.shape}")
    try:
        hist, _ = np.histogramdd([np.reshape(img1, -1), np.reshape(img2, -1)], bins=64, density=True)
        H0, H1, H01 = entropy(np.sum(hist, axis=0)), entropy(np.sum(hist, axis=1)), entropy(np.reshape(hist, -1))
        return float((H0 + H1) / H01 - 1.0)
    except Exception:
        return 0.0
