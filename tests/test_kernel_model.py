"""The kernel design (tests/kernel_model.py) against np.fft and the oracle — CPU only."""
import numpy as np
import pytest

from oracle import path_b as O
from tests import kernel_model as KM


@pytest.mark.parametrize("N", [32, 64, 128, 256, 512, 1024])
def test_lane_fft_matches_numpy(N):
    rng = np.random.default_rng(N)
    x = rng.normal(size=N) + 1j * rng.normal(size=N)
    T = N // 8
    V = KM.fft_lanes(x.reshape(8, T).T)
    X = np.fft.fft(x)
    np.testing.assert_allclose(V.T.reshape(-1), X, atol=1e-10)


@pytest.mark.parametrize("N", [32, 64, 128])
def test_two_pass_half_plane_matches_fft2(N):
    rng = np.random.default_rng(7)
    img = rng.normal(size=(N, N))
    F = KM.second_pass(KM.first_pass(img))
    ref = np.fft.fft2(img)
    np.testing.assert_allclose(F, ref[: N // 2 + 1, :], atol=1e-9)


@pytest.mark.parametrize("N", [32, 64])
@pytest.mark.parametrize("kind", ["band", "layer", "random"])
def test_half_plane_moments_equal_full_plane_masked_cc(N, kind):
    rng = np.random.default_rng(3)
    img_e = rng.normal(size=(N, N))
    img_s = O.simulate_helical_projection(1, 29.0, 10.0, 1, 0.4 * N * 2.0, 4.0, 0, 0, N, N, 2.0)
    if kind == "band":
        mask = O.radial_band_mask(N, N)
    elif kind == "layer":
        mask = O.layer_line_mask(N, N, axial_bins=[3, 6, 9], half_width=1)
    else:  # asymmetric mask that touches the self-conjugate rows/columns (index 0 = -Nyquist)
        mask = rng.random((N, N)) < 0.3
    pe = O.compute_power_spectra(img_e, 2.0)[0]
    ps = O.compute_power_spectra(img_s, 2.0)[0]
    ref = O.cross_correlation_coefficient(pe[mask], ps[mask])
    W = KM.half_plane_weights(mask)
    h = slice(0, N // 2 + 1)
    qe = np.log1p(np.abs(np.fft.fft2(img_e)))[h]
    qs = np.log1p(np.abs(KM.second_pass(KM.first_pass(img_s))))
    assert KM.pearson_from_moments(qs, qe, W) == pytest.approx(ref, abs=1e-12)
    assert W.sum() == mask.sum()


@pytest.mark.parametrize("csym,rot", [(1, 0.0), (3, 25.0)])
def test_run_table_identity_gives_the_oracles_spectrum(csym, rot):
    """The shared-twist pipelines' identity H[ky][x] = sum_c ex_c(x) G_c[ky]: one table
    for a twist serves every rise; the half-plane spectrum it leads to is the oracle's, up to the Gaussian
    tails below 2^-24 that the kernels drop."""
    n, apix, twist = 64, 2.0, 29.0
    d, br = 0.4 * n * apix, 2 * apix
    rises = (9.0, 10.0, 11.5)
    imax_t = int(np.ceil(n * apix / min(rises)))
    tab, rpx, sigma2 = KM.run_table(n, apix, twist, csym, rot, d, br, imax_t)
    for rise in rises:
        H = KM.first_pass_from_table(tab, rpx, sigma2, n, apix, rise, imax_t)
        F = KM.second_pass(H)
        img = O.simulate_helical_projection(1, twist, rise, csym, d, br, 0, 0, n, n, apix, rot=rot)
        ref = np.fft.fft2(img)[: n // 2 + 1]
        assert np.abs(F - ref).max() < 2e-5 * np.abs(ref).max()
