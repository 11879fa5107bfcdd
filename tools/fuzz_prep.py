#!/usr/bin/env python3
"""Randomised agreement campaign of the device image preparation (csrc/image_prep.inc) with oracle/prep.py: random shapes,
scales, rotations, shifts and element types for transform_image, rescale / down_scale, rotate_shift_image(order 1 and 3) and the
helix estimates.  argv = [cases, seed].  Prints the worst differences; exits non-zero when a tolerance is exceeded."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from helicon_amd import denovo3D as D  # noqa: E402
from oracle import prep as P  # noqa: E402  (the checker)

if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
    worst = {"transform_image": 0.0, "transform_image_mismatch_share": 0.0, "rescale": 0.0, "rotate_shift_1": 0.0, "rotate_shift_3": 0.0,
             "estimate_rotation": 0.0, "estimate_shift": 0.0}
    bad = []
    for k in range(cases):
        ny, nx = int(rng.integers(9, 300)), int(rng.integers(9, 300))
        dtype = np.float32 if rng.random() < 0.7 else np.float64
        img = (rng.normal(size=(ny, nx)) * rng.uniform(0.1, 50) + rng.uniform(-5, 5)).astype(dtype)
        tol = 5e-6 * float(np.abs(img).max()) if dtype == np.float32 else 1e-10 * float(np.abs(img).max())
        # transform_image
        kw = dict(scale=float(rng.uniform(0.6, 1.6)) if rng.random() < 0.5 else (float(rng.uniform(0.7, 1.4)), float(rng.uniform(0.7, 1.4))),
                  rotation=float(rng.uniform(-180, 180)), pre_translation=(float(rng.uniform(-4, 4)), float(rng.uniform(-4, 4))),
                  post_translation=(float(rng.uniform(-6, 6)), float(rng.uniform(-6, 6))), order=int(rng.integers(0, 2)))
        if rng.random() < 0.3:
            kw["rotation_center"] = (float(rng.uniform(0, ny)), float(rng.uniform(0, nx)))
        got, want = D.transform_image(img, **kw), P.transform_image(img, **kw)
        diff = np.abs(got.astype(np.float64) - want.astype(np.float64))
        share = float((diff > tol).mean())      # (float32: a sample within rounding of a cell boundary may take the neighbouring cell)
        worst["transform_image_mismatch_share"] = max(worst["transform_image_mismatch_share"], share)
        worst["transform_image"] = max(worst["transform_image"], float(np.median(diff) / max(tol, 1e-300)))
        if share > (1e-3 if dtype == np.float32 else 0.0):
            bad.append(("transform_image", (ny, nx), dtype.__name__, kw, share))
        # rescale
        scale = float(rng.uniform(0.15, 1.3))
        okw = dict(order=3 if rng.random() < 0.8 else 1, anti_aliasing=bool(rng.random() < 0.85))
        got, want = D.rescale(img, scale, **okw), P.rescale(img, scale, **okw)
        d = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max()) if got.shape == want.shape else float("inf")
        worst["rescale"] = max(worst["rescale"], d / tol)
        if d > 4 * tol:
            bad.append(("rescale", (ny, nx), dtype.__name__, scale, okw, d, tol))
        # rotate_shift_image
        for order in (1, 3):
            a, ps = float(rng.uniform(-90, 90)), (float(rng.uniform(-5, 5)), float(rng.uniform(-5, 5)))
            f32 = img.astype(np.float32)
            got, want = D.rotate_shift_image(f32, a, (0, 0), ps, order=order), P.rotate_shift_image(f32, a, (0, 0), ps, order=order)
            d = float(np.abs(got - want).max())
            worst[f"rotate_shift_{order}"] = max(worst[f"rotate_shift_{order}"], d / (5e-6 * float(np.abs(f32).max())))
            if d > 2e-5 * float(np.abs(f32).max()):
                bad.append((f"rotate_shift_{order}", (ny, nx), a, ps, d))
        # helix estimates on a bar
        yy, xx = np.mgrid[0:ny, 0:nx].astype(np.float64)
        ang, sh = float(rng.uniform(-30, 30)), float(rng.uniform(-0.1, 0.1)) * ny
        t = np.deg2rad(ang)
        dist = -(xx - nx / 2) * np.sin(t) + (yy - ny / 2 - sh) * np.cos(t)
        bar = (np.exp(-0.5 * (dist / max(2.0, ny / 14)) ** 2) * (np.abs(dist) < ny / 6) * (1 + 0.1 * rng.random((ny, nx)))).astype(dtype)
        g, w = D.estimate_helix_rotation_center_diameter(bar), P.estimate_helix_rotation_center_diameter(bar)
        worst["estimate_rotation"] = max(worst["estimate_rotation"], abs(g[0] - w[0]))
        worst["estimate_shift"] = max(worst["estimate_shift"], abs(g[1] - w[1]))
        if abs(g[0] - w[0]) > 1e-4 or abs(g[1] - w[1]) > 1e-3 or abs(g[2] - w[2]) > 1:
            bad.append(("estimate", (ny, nx), dtype.__name__, g, w))
    print(f"{cases} random cases; worst (in units of the tolerance where one applies): {worst}")
    for b in bad[:10]:
        print("EXCEEDED", b)
    sys.exit(1 if bad else 0)
