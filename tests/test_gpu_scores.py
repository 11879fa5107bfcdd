"""The non-cosine scores of lsq_reconstruct (solver_linear_regression.py:484-524: "ssim", "ms_ssim", "mutual_information",
"composite"; helicon.ssim_score / ms_ssim_score / mutual_information_score, lib/analysis.py:487-613 — scikit-image in the
reference) on the device against oracle/prep.py's restatement (pinned by derivation: scikit-image is not installed beside
the reference), and their wiring into lsq_reconstruct against the same composition through the oracle."""
import numpy as np
import pytest
from scipy import ndimage as ndi

from helicon_amd import solver as S
from oracle import path_a as A
from oracle import prep as P

pytestmark = pytest.mark.gpu


def test_scores_against_the_oracle():
    rng = np.random.default_rng(6)
    for shape, noise in [((64, 128), 0.02), ((128, 64), 0.2), ((33, 47), 0.05), ((8, 9), 0.1), ((200, 300), 0.5)]:
        a = ndi.gaussian_filter(rng.random(shape), 1.5).astype(np.float32)
        b = (a + noise * rng.standard_normal(shape) * a.std()).astype(np.float32)
        assert S.ssim_score(a, b) == pytest.approx(P.ssim_score(a, b), abs=2e-6), shape
        assert S.ms_ssim_score(a, b) == pytest.approx(P.ms_ssim_score(a, b), abs=5e-6), shape
        assert S.mutual_information_score(a, b) == pytest.approx(P.mutual_information_score(a, b), abs=1e-12), shape
    a = rng.random((32, 32)).astype(np.float32)
    assert S.ssim_score(a, a) == pytest.approx(1.0, abs=1e-6) and S.mutual_information_score(a, a) == pytest.approx(1.0, abs=1e-9)
    assert S.ssim_score(np.ones((16, 16)), np.ones((16, 16))) == 0.0 and S.ssim_score(a[:5], a[:5]) == 0.0
    assert S.ms_ssim_score(a[:6], a[:6]) == 0.0
    with pytest.raises(ValueError):
        S.ssim_score(a, a[:, :10])
    # values on a bin edge and a constant image take np.histogramdd's conventions
    e = np.linspace(0, 1, 65).astype(np.float32).reshape(5, 13)
    f = np.full((5, 13), 0.25, np.float32)
    assert S.mutual_information_score(e, e[::-1].copy()) == pytest.approx(P.mutual_information_score(e, e[::-1].copy()), abs=1e-12)
    got, want = S.mutual_information_score(e, f), P.mutual_information_score(e, f)
    assert (np.isnan(got) and np.isnan(want)) or got == pytest.approx(want, abs=1e-12)


@pytest.mark.parametrize("interp", ["nn", "linear"])
def test_lsq_reconstruct_with_the_non_cosine_scores(golden_dir, interp):
    """solver:484-524 through helicon_amd.lsq_reconstruct: the prediction of the float32 map scattered to its pixels and
    scored against the transposed input region — against the same composition from the oracle's matrices and solution."""
    g = np.load(golden_dir / "g5_lsq.npz")
    img = g["helix_image"]
    kw = dict(reconstruct_diameter_2d_pixel=20, reconstruct_diameter_3d_pixel=20, reconstruct_length_2d_pixel=32, reconstruct_length_3d_pixel=6,
              sym_oversample=1, interpolation=interp)
    (rec_c, _, _), cos = S.lsq_reconstruct(img, 1.0, 29.0, 2.0, 1, **kw)
    _, _, parts = A.lsq_reconstruct(img, 1.0, 29.0, 2.0, 1, return_parts=True, **kw)
    x = rec_c[parts["mask"]].astype(np.float32)
    pred = np.asarray(parts["A_data"].astype(np.float64) @ x.astype(np.float64)).ravel().astype(np.float32)
    pred_2d = np.zeros((32, 20), np.float32)
    pred_2d.ravel()[parts["b_pid"]] = pred
    ref_2d = np.ascontiguousarray(img[16 - 10: 16 + 10, 16 - 16: 16 + 16].T)
    want = dict(ssim=P.ssim_score(pred_2d, ref_2d), ms_ssim=P.ms_ssim_score(pred_2d, ref_2d), mutual_information=P.mutual_information_score(pred_2d, ref_2d))
    want["composite"] = float(np.mean([A.cosine_similarity(pred, parts["b_data"]), want["ssim"], want["ms_ssim"], want["mutual_information"]]))
    for metric, w in want.items():
        (rec, _, _), got = S.lsq_reconstruct(img, 1.0, 29.0, 2.0, 1, score_metric=metric, **kw)
        np.testing.assert_array_equal(rec, rec_c)                    # the same solve, another score
        assert got == pytest.approx(w, abs=2e-3 if metric == "mutual_information" else 2e-5), (metric, got, w)
    assert S.lsq_reconstruct(img, 1.0, 29.0, 2.0, 1, score_metric="frc", **kw)[1] == cos        # falls through to cosine (solver:523)
    with pytest.raises(ValueError, match="shape mismatch"):
        S.lsq_reconstruct(img, 1.0, 29.0, 2.0, 1, score_metric="ssim", fsc_test=2, **kw)
