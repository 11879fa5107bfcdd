"""The image-preparation oracle (oracle/prep.py) on the CPU: against the reference's own outputs where the reference
function runs here (fixture G15: ``rotate_shift_image(order=3)``, ``pad_to_size``, ``set_to_periodic_range`` — SciPy /
NumPy only) and, for the scikit-image calls it restates (scikit-image is not installed here: pinned by derivation),
against independent statements of the same operation."""
import numpy as np
import pytest
from scipy import ndimage as ndi

from oracle import prep as P


def _helix_image(ny=96, nx=160, angle=12.0, shift=5.0, seed=0):
    """A noisy bar through the box: `angle` degrees off horizontal, `shift` pixels off the middle row."""
    yy, xx = np.mgrid[0:ny, 0:nx].astype(np.float64)
    a = np.deg2rad(angle)
    d = -(xx - nx / 2) * np.sin(a) + (yy - ny / 2 - shift) * np.cos(a)
    img = np.exp(-0.5 * (d / 6.0) ** 2) * (1.0 + 0.3 * np.cos(0.4 * ((xx - nx / 2) * np.cos(a) + (yy - ny / 2) * np.sin(a))))
    img = img * (np.abs(d) < 14)
    return (img + 0.02 * np.random.default_rng(seed).random(img.shape) * (img > 0)).astype(np.float32)


def test_g15_reference_outputs(golden_dir):
    g = np.load(golden_dir / "g15_rotate_shift_cubic.npz")
    for k in range(5):
        a = g[f"case{k}_args"]
        got = P.rotate_shift_image(g[f"case{k}_image"], a[0], (a[1], a[2]), (a[3], a[4]), order=3)
        want = g[f"case{k}_out"]
        assert got.dtype == want.dtype
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)
    for k, target in enumerate([(8, 8), (6, 6), (8, 10), (7, 5)]):
        np.testing.assert_array_equal(P.pad_to_size(g[f"pad{k}_image"], target), g[f"pad{k}_out"])
    got = [P.set_to_periodic_range(float(v), min=-180, max=180) for v in g["periodic_in"]]
    np.testing.assert_array_equal(got, g["periodic_out"])


def test_warp_is_the_bilinear_resampling_scipy_computes_away_from_the_border():
    """skimage's fast warp and scipy's affine_transform(order=1) are the same interpolation in the interior (they differ
    in how a sample within one pixel of the border mixes with the constant): transform_image against affine_transform
    with the same matrix on (row, col) coordinates, float64."""
    rng = np.random.default_rng(1)
    img = rng.normal(size=(50, 70))
    for rot, post in [(10.0, (0.0, 0.0)), (-37.5, (2.5, -1.25)), (90.0, (0.0, 3.0))]:
        m = P.transform_image_matrix(img.shape, rotation=rot, post_translation=post)
        inv = np.linalg.inv(m)
        # (x, y) -> (row, col) ordering for scipy
        mat = np.array([[inv[1, 1], inv[1, 0]], [inv[0, 1], inv[0, 0]]])
        off = np.array([inv[1, 2], inv[0, 2]])
        want = ndi.affine_transform(img, mat, off, order=1, mode="constant")
        got = P.warp_affine(img, inv, order=1, clip=False)
        inside = ndi.binary_erosion(want != 0, iterations=2)
        np.testing.assert_allclose(got[inside], want[inside], rtol=0, atol=1e-12)
    same = P.transform_image(img.astype(np.float32))
    np.testing.assert_array_equal(same, img.astype(np.float32))    # identity
    clipped = P.transform_image(img, rotation=20.0)
    assert clipped.min() >= img.min() and clipped.max() <= img.max()


def test_warp_clip_keeps_a_cval_outside_the_range():
    img = np.full((8, 8), 5.0, dtype=np.float32)
    img[2, 3] = 7.0
    out = P.transform_image(img, post_translation=(3.0, 0.0))
    assert (out[:3] == 0).all() and out[3:].min() >= 5.0     # rows shifted in from outside keep cval = 0 < min


def test_rescale_shapes_identity_and_mean():
    rng = np.random.default_rng(2)
    img = rng.random((60, 90)).astype(np.float32)
    assert P.rescale(img, 0.5).shape == (30, 45)
    assert P.rescale(img, 1 / 3).shape == (20, 30)
    assert P.rescale(img, 0.37).shape == (22, 33)
    assert P.rescale(np.zeros((5, 5), np.float32), 0.5).shape == (2, 2)     # round half to even: 2.5 -> 2
    np.testing.assert_allclose(P.rescale(img, 1.0), img, rtol=0, atol=1e-6)
    smooth = ndi.gaussian_filter(rng.random((64, 64)), 4.0).astype(np.float32)
    assert abs(P.rescale(smooth, 0.5).mean() - smooth.mean()) < 2e-3
    d = P.down_scale(img, 3.0, 1.0)
    assert d.shape == (20, 30)
    d = P.down_scale(rng.random((50, 70)).astype(np.float32), 2.7, 1.0)    # 18.5 -> 18 (even), 25.9 -> 26
    assert d.shape[0] % 2 == 0 and d.shape[1] % 2 == 0
    assert P.down_scale(img, 1.0, 1.0) is img and P.down_scale(img, 0.5, 1.0) is img


def test_closing_and_helix_estimates():
    m = np.zeros((7, 11), bool)
    m[2:5, 2:9] = True
    m[3, 5] = False                    # a one-pixel hole closes, nothing else changes
    c = P.closing_cross(m)
    assert c[3, 5] and c.sum() == m.sum() + 1
    edge = np.zeros((5, 5), bool)
    edge[0, 0] = True                  # the border is ignored, not eroded away
    assert P.closing_cross(edge)[0, 0]
    for angle, shift in [(12.0, 5.0), (-25.0, -3.0), (0.0, 0.0)]:
        img = _helix_image(angle=angle, shift=shift)
        rot, dy, diameter = P.estimate_helix_rotation_center_diameter(img)
        assert abs(rot + angle) < 1.0, (rot, angle)     # the rotation that UNDOES the tilt
        assert 24 <= diameter <= 34
        # transform_image with the estimate lays the helix flat (what the estimate itself and app.py:2070 do with it) ...
        flat = P.transform_image(img, rotation=rot, post_translation=(dy, 0))
        rot2, dy2, _ = P.estimate_helix_rotation_center_diameter(flat)
        assert abs(rot2) < 1.0 and abs(dy2) < 1.0
        # ... while rotate_shift_image turns the OTHER way for the same angle (scipy's matrix is the output -> input map,
        # transform_image hands warp the inverse of its matrix), so auto_horizontalize(refine=False) — which passes the
        # estimate to rotate_shift_image, utils.py:401, 420 — doubles the tilt instead of removing it.  The oracle (and
        # the product) mirror that; the pipeline only ever calls it with refine=True.
        out, theta, sy = P.auto_horizontalize(img)
        assert theta == rot
        rot3, _, _ = P.estimate_helix_rotation_center_diameter(out)
        assert angle == 0.0 or abs(rot3) > 1.5 * abs(rot)      # (the box clips a bar at 50 degrees)
    assert P.estimate_helix_rotation_center_diameter(np.zeros((16, 16), np.float32)) == (0.0, 0.0, 16)


def test_non_cosine_scores_of_the_oracle():
    """The restated scikit-image metrics (pinned by derivation): identities and orderings they must satisfy."""
    rng = np.random.default_rng(5)
    a = ndi.gaussian_filter(rng.random((64, 128)), 2.0).astype(np.float32)
    b = (a + 0.02 * rng.standard_normal(a.shape)).astype(np.float32)
    c = (a + 0.2 * rng.standard_normal(a.shape)).astype(np.float32)
    assert P.ssim_score(a, a) == pytest.approx(1.0, abs=1e-6) and P.ms_ssim_score(a, a) == pytest.approx(1.0, abs=1e-6)
    assert 1.0 > P.ssim_score(a, b) > P.ssim_score(a, c) > 0
    assert 1.0 > P.ms_ssim_score(a, b) > P.ms_ssim_score(a, c) > 0
    assert P.mutual_information_score(a, a) == pytest.approx(1.0, abs=1e-9)          # H(a) + H(a) over H(a, a) = 2
    assert P.mutual_information_score(a, b) > P.mutual_information_score(a, c) > 0
    assert P.mutual_information_score(a, b) == pytest.approx(P.mutual_information_score(b, a), abs=1e-12)
    assert P.ssim_score(np.ones((16, 16), np.float32), np.ones((16, 16), np.float32)) == 0.0      # no range
    assert P.ssim_score(a[:5], b[:5]) == 0.0                                                     # the window does not fit: swallowed
    with pytest.raises(ValueError):
        P.ssim_score(a, b[:, :10])
    # the SSIM of the interior does not depend on how the filter treats the rim
    want = P.structural_similarity(a, b, data_range=float(max(np.ptp(a), np.ptp(b))))
    ux = ndi.uniform_filter(a.astype(np.float64), 7, mode="constant")
    assert np.isfinite(want) and ux.shape == a.shape
