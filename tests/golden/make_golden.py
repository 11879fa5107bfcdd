#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by importing the REFERENCE itself.

Run in the build container only (the reference never travels to the GPU box):

    HELION_CACHE_DIR=/tmp/helicon_golden_cache PYTHONDONTWRITEBYTECODE=1 \
        PYTHONPATH=/root/reference/src python3 tests/golden/make_golden.py

Every array written is an INPUT or an OUTPUT of a reference function; no reference source
text is stored.  Stage 2 (``compute_power_spectra``) cannot run here (finufft is not
installed), so the composed fixtures G3 use the reference's own post-processing functions
(``normalize_percentile``, ``cross_correlation_coefficient``) around ``np.fft.fft2``, which is
what ``fft_rescale`` evaluates for default arguments (reference transforms.py:696-711).
"""
import json
import os
import sys
from pathlib import Path

import numpy as np

import helicon  # the reference
from helicon.lib import analysis, filters, angular, transforms
from helicon.webApps.denovo3D import utils

OUT = Path(__file__).resolve().parent


def versions():
    import scipy

    return dict(helicon=helicon.__version__, numpy=np.__version__, scipy=scipy.__version__,
                python=sys.version.split()[0])


def g1_simulate():
    """B1 outputs, deterministic branch (n=1, polymer=0) + seeded n>1 branch."""
    cases = [
        # (n, twist, rise, csym, diameter, ball_radius, ny, nx, apix, tilt, rot, psi, dy)
        (1, 29.0, 10.0, 1, 25.6, 4.0, 32, 32, 2.0, 0, 0, 0, 0),
        (1, 30.0, 5.0, 1, 40.0, 3.0, 32, 32, 2.0, 0, 0, 0, 0),
        (1, 30.0, 5.0, 1, 40.0, 3.0, 32, 32, 2.0, 5, 0, 10, 2),
        (1, -27.5, 7.25, 3, 60.0, 5.0, 64, 64, 2.0, 0, 30.0, 0, 0),
        (1, 1.2, 4.75, 1, 25.6, 2.0, 64, 64, 1.0, 0, 0, 0, 0),
        (1, 65.3, 23.1, 2, 100.0, 10.0, 64, 64, 5.0, 3.0, 15.0, -4.0, -6.0),
        (1, 12.0, 8.0, 1, 60.0, 6.0, 48, 96, 2.0, 0, 0, 0, 0),
        (1, -12.0, 8.0, 4, 60.0, 6.0, 48, 96, 2.0, 0, 45.0, 0, 3.5),
        (1, 179.1, 2.4, 1, 30.0, 2.5, 32, 32, 2.0, 0, 0, 0, 0),
        (1, 29.0, 10.0, 6, 50.0, 4.0, 64, 64, 2.0, 0, 0, 0, 0),
    ]
    arrs = {}
    for k, c in enumerate(cases):
        n, tw, rs, cs, d, br, ny, nx, apix, tilt, rot, psi, dy = c
        out = utils.simulate_helical_projection(n, tw, rs, cs, d, br, 0, 0, ny, nx, apix,
                                                tilt=tilt, rot=rot, psi=psi, dy=dy)
        arrs[f"case{k}_args"] = np.asarray(c, dtype=np.float64)
        arrs[f"case{k}_out"] = out
    # seeded multi-unit branch (reference tests/test_denovo3D_utils.py:107-143 inputs)
    for k, kw in enumerate([dict(), dict(tilt=5, psi=10, dy=2)]):
        np.random.seed(1234 + k)
        out = utils.simulate_helical_projection(10, 30, 5, 1, 40, 3, 0, 0, 32, 32, 2.0, **kw)
        arrs[f"multi{k}_seed"] = np.asarray([1234 + k])
        arrs[f"multi{k}_kw"] = np.asarray([kw.get("tilt", 0), kw.get("psi", 0), kw.get("dy", 0)], dtype=np.float64)
        arrs[f"multi{k}_out"] = out
    arrs["n_cases"] = np.asarray([len(cases)])
    np.savez_compressed(OUT / "g1_simulate.npz", **arrs)


def g2_scores():
    """B3 / A7 on seeded vectors, incl. zero-variance -> 0."""
    arrs = {}
    for n in (3, 1000, 65536):
        rng = np.random.default_rng(n)
        a = rng.normal(size=n)
        b = 0.3 * a + rng.normal(size=n)
        a32, b32 = a.astype(np.float32).astype(np.float64), b.astype(np.float32).astype(np.float64)
        if n <= 1000:  # larger inputs are regenerated from the seed (= n) by the test
            arrs[f"n{n}_a"] = a.astype(np.float32)
            arrs[f"n{n}_b"] = b.astype(np.float32)
        arrs[f"n{n}_cc"] = np.asarray([analysis.cross_correlation_coefficient(a32, b32)])
        arrs[f"n{n}_cos"] = np.asarray([analysis.cosine_similarity(a32, b32)])
    const = np.full(17, 2.5)
    ramp = np.arange(17, dtype=np.float64)
    arrs["const_cc"] = np.asarray([float(analysis.cross_correlation_coefficient(ramp, const))])
    arrs["zero_cos"] = np.asarray([float(analysis.cosine_similarity(ramp, const * 0))])
    np.savez_compressed(OUT / "g2_scores.npz", **arrs)


def _pwr(img, log):
    fft = np.fft.fftshift(np.fft.fft2(img.astype(np.complex128)))
    pwr = np.log1p(np.abs(fft)) if log else np.abs(fft)
    return filters.normalize_percentile(pwr, percentile=(0, 100))


def _band(n, r_lo, r_hi):
    k = np.arange(n) - n // 2
    r2 = k[:, None].astype(np.float64) ** 2 + k[None, :].astype(np.float64) ** 2
    return (r2 > r_lo**2) & (r2 < r_hi**2)


def g3_composed():
    """Composed Path-B scores on a noisy synthetic helix: 9 x 5 grid, log in {True, False}."""
    arrs = {}
    for n, apix, truth, twists, rises in (
        (64, 2.0, (29.0, 10.0, 1), np.arange(25.0, 33.0 + 0.5, 1.0), np.arange(8.0, 12.0 + 0.5, 1.0)),
        (128, 2.0, (29.0, 10.0, 1), np.arange(25.0, 33.0 + 0.5, 1.0), np.arange(8.0, 12.0 + 0.5, 1.0)),
        (64, 2.0, (-40.0, 7.0, 3), np.arange(-44.0, -36.0 + 0.5, 1.0), np.arange(5.0, 9.0 + 0.5, 1.0)),
    ):
        tw0, rs0, cs0 = truth
        d = 0.4 * n * apix
        br = 2 * apix
        clean = utils.simulate_helical_projection(1, tw0, rs0, cs0, d, br, 0, 0, n, n, apix)
        noise = np.random.default_rng(0).normal(0, 0.5 * clean.std(), clean.shape)
        img = (clean + noise).astype(np.float32)
        mask = _band(n, 2, n // 2 - 1)
        tag = f"n{n}_c{cs0}"
        arrs[f"{tag}_image"] = img
        arrs[f"{tag}_meta"] = np.asarray([n, apix, tw0, rs0, cs0, d, br], dtype=np.float64)
        arrs[f"{tag}_twists"] = twists
        arrs[f"{tag}_rises"] = rises
        for log in (True, False):
            pe = _pwr(img.astype(np.float64), log)
            sc = np.zeros((len(twists), len(rises)))
            for i, tw in enumerate(twists):
                for j, rs in enumerate(rises):
                    sim = utils.simulate_helical_projection(1, float(tw), float(rs), cs0, d, br, 0, 0, n, n, apix)
                    ps = _pwr(sim, log)
                    sc[i, j] = analysis.cross_correlation_coefficient(pe[mask], ps[mask])
            arrs[f"{tag}_scores_log{int(log)}"] = sc
            arrs[f"{tag}_argmax_log{int(log)}"] = np.asarray(np.unravel_index(np.argmax(sc), sc.shape))
    np.savez_compressed(OUT / "g3_composed.npz", **arrs)


def g3b_general_sizes():
    """The same composition on image sizes that are neither square nor powers of two (utils.py:31-47 and
    transforms.py:687-704 take any (ny, nx)): scores of a small twist-major grid, both log settings, and one
    simulated projection per size.  Sizes: 96 x 96 (2^5 3), 80 x 120 (ny != nx), 50 x 70 (factors 5 and 7),
    45 x 63 (odd sides)."""
    arrs = {}
    tags = []
    for ny, nx, apix, truth, twists, rises in (
        (96, 96, 2.0, (29.0, 10.0, 1), np.arange(26.0, 32.0 + 0.5, 1.0), np.arange(8.0, 12.0 + 0.25, 0.5)),
        (80, 120, 2.0, (-40.0, 7.0, 2), np.arange(-43.0, -37.0 + 0.5, 1.0), np.arange(5.0, 9.0 + 0.25, 0.5)),
        (50, 70, 3.0, (29.0, 12.0, 1), np.arange(27.0, 31.0 + 0.5, 1.0), np.arange(9.0, 15.0 + 0.25, 0.75)),
        (45, 63, 3.0, (55.0, 9.0, 3), np.arange(52.0, 58.0 + 0.5, 1.0), np.arange(7.0, 11.0 + 0.25, 0.5)),
    ):
        tw0, rs0, cs0 = truth
        d = 0.4 * ny * apix
        br = 2 * apix
        clean = utils.simulate_helical_projection(1, tw0, rs0, cs0, d, br, 0, 0, ny, nx, apix)
        noise = np.random.default_rng(1).normal(0, 0.5 * clean.std(), clean.shape)
        img = (clean + noise).astype(np.float32)
        ky = np.arange(ny) - ny // 2
        kx = np.arange(nx) - nx // 2
        r2 = ky[:, None].astype(np.float64) ** 2 + kx[None, :].astype(np.float64) ** 2
        mask = (r2 > 4.0) & (r2 < (min(ny, nx) // 2 - 1) ** 2)
        tag = f"s{ny}x{nx}"
        tags.append(tag)
        arrs[f"{tag}_clean"] = clean.astype(np.float32)  # reference output, stored in float32 (tolerance of its test: 5e-6)
        arrs[f"{tag}_image"] = img
        arrs[f"{tag}_meta"] = np.asarray([ny, nx, apix, tw0, rs0, cs0, d, br], dtype=np.float64)
        arrs[f"{tag}_twists"] = twists
        arrs[f"{tag}_rises"] = rises
        for log in (True, False):
            pe = _pwr(img.astype(np.float64), log)
            sc = np.zeros((len(twists), len(rises)))
            for i, tw in enumerate(twists):
                for j, rs in enumerate(rises):
                    sim = utils.simulate_helical_projection(1, float(tw), float(rs), cs0, d, br, 0, 0, ny, nx, apix)
                    sc[i, j] = analysis.cross_correlation_coefficient(pe[mask], _pwr(sim, log)[mask])
            arrs[f"{tag}_scores_log{int(log)}"] = sc
            arrs[f"{tag}_argmax_log{int(log)}"] = np.asarray(np.unravel_index(np.argmax(sc), sc.shape))
        arrs[f"{tag}_pwr_log1"] = _pwr(img.astype(np.float64), True).astype(np.float32)
    arrs["tags"] = np.asarray(tags)
    np.savez_compressed(OUT / "g3b_general_sizes.npz", **arrs)


def g3c_general_sizes_tilted():
    """G3b's composition with an out-of-plane tilt, an in-plane rotation and a shift (utils.py:31-47, 166-170 take any
    (ny, nx) with any tilt / psi / dy): scores of a small twist-major grid on 80 x 120 and on 48 x 74 (a row length with
    the prime factor 37), log spectrum, and one simulated projection per size."""
    arrs = {}
    tags = []
    for ny, nx, apix, truth, twists, rises, tilt, psi, dy in (
        (80, 120, 2.0, (-40.0, 7.0, 2), np.arange(-42.0, -38.0 + 0.5, 1.0), np.arange(6.0, 8.0 + 0.25, 0.5), 5.0, 10.0, 2.0),
        (48, 74, 2.5, (29.0, 10.0, 1), np.arange(27.0, 31.0 + 0.5, 1.0), np.arange(9.0, 11.0 + 0.25, 0.5), -4.0, 3.0, -1.5),
    ):
        tw0, rs0, cs0 = truth
        d = 0.4 * ny * apix
        br = 2 * apix
        kw = dict(tilt=tilt, psi=psi, dy=dy)
        clean = utils.simulate_helical_projection(1, tw0, rs0, cs0, d, br, 0, 0, ny, nx, apix, **kw)
        noise = np.random.default_rng(2).normal(0, 0.5 * clean.std(), clean.shape)
        img = (clean + noise).astype(np.float32)
        ky = np.arange(ny) - ny // 2
        kx = np.arange(nx) - nx // 2
        r2 = ky[:, None].astype(np.float64) ** 2 + kx[None, :].astype(np.float64) ** 2
        mask = (r2 > 4.0) & (r2 < (min(ny, nx) // 2 - 1) ** 2)
        tag = f"s{ny}x{nx}"
        tags.append(tag)
        arrs[f"{tag}_clean"] = clean.astype(np.float32)
        arrs[f"{tag}_image"] = img
        arrs[f"{tag}_meta"] = np.asarray([ny, nx, apix, tw0, rs0, cs0, d, br, tilt, psi, dy], dtype=np.float64)
        arrs[f"{tag}_twists"] = twists
        arrs[f"{tag}_rises"] = rises
        pe = _pwr(img.astype(np.float64), True)
        sc = np.zeros((len(twists), len(rises)))
        for i, tw in enumerate(twists):
            for j, rs in enumerate(rises):
                sim = utils.simulate_helical_projection(1, float(tw), float(rs), cs0, d, br, 0, 0, ny, nx, apix, **kw)
                sc[i, j] = analysis.cross_correlation_coefficient(pe[mask], _pwr(sim, True)[mask])
        arrs[f"{tag}_scores_log1"] = sc
        arrs[f"{tag}_argmax_log1"] = np.asarray(np.unravel_index(np.argmax(sc), sc.shape))
    arrs["tags"] = np.asarray(tags)
    np.savez_compressed(OUT / "g3c_general_sizes_tilted.npz", **arrs)
    print("g3c_general_sizes_tilted", tags)


def g4_path_a():
    """Path A building blocks: mask counts, symmetry-pair lists, Halton index lists,
    back-projected coordinates, and the CSR triplets of the NN data matrix and of the NN symmetry matrix on the inputs of
    the reference's own structural tests (tests/test_denovo3D_solver.py:8-175)."""
    from scipy.stats import qmc

    from helicon.webApps.denovo3D import solver_linear_regression as S

    arrs = {}
    for k, (nz, ny, nx, rmin, rmax) in enumerate([(4, 36, 36, 0, 17), (4, 64, 64, 0, 17), (8, 8, 8, 0, 3), (6, 20, 20, 3, 9)]):
        m = analysis.get_cylindrical_mask(nz, ny, nx, rmin=rmin, rmax=rmax)
        arrs[f"mask{k}_args"] = np.asarray([nz, ny, nx, rmin, rmax])
        arrs[f"mask{k}_count"] = np.asarray([np.count_nonzero(m)])
        arrs[f"mask{k}_first_nonzero"] = np.argwhere(m)[:5]
    for n in (7, 9, 16, 73):
        arrs[f"halton_{n}"] = qmc.Halton(d=1, scramble=False).integers(l_bounds=0, u_bounds=n, n=n)[:, 0]
    for k, (tw, rs, cs, nz) in enumerate([(30, 5, 1, 20), (30, 5, 2, 20), (-41.5, 3.7, 3, 12)]):
        pairs = S.sorted_hsym_csym_pairs(tw, rs, cs, nz)
        arrs[f"pairs{k}_args"] = np.asarray([tw, rs, cs, nz], dtype=np.float64)
        arrs[f"pairs{k}"] = np.asarray([[p[0], p[1], p[2], p[3], p[4], *p[5][0], *p[5][1]] for p in pairs], dtype=np.float64)
    img4 = np.arange(16, dtype=np.float32).reshape(4, 4)
    (X, Y, Z), vals = S.back_project_2d_coords_to_3d_coords(img4, 1.0, 4, 4)
    arrs.update(bp_image=img4, bp_X=X, bp_Y=Y, bp_Z=Z, bp_vals=vals)
    (X, Y, Z), vals = S.back_project_2d_coords_to_3d_coords(np.arange(48, dtype=np.float32).reshape(6, 8), 1.5, 4, 6)
    arrs.update(bp2_X=X, bp2_Y=Y, bp2_Z=Z, bp2_vals=vals)
    cases = [
        # image, scale, twist, rise, csym, tilt, psi, dy, D2d, L2d, D3d, D3d_inner, L3d, min_lines
        (np.eye(8, dtype=np.float32), 1.0, 30.0, 2.0, 1, 0.0, 0.0, 0.0, 8, 8, 8, 0, 8, 64),
        (np.random.default_rng(4).random((12, 16)).astype(np.float32), 1.0, -41.5, 3.7, 2, 3.0, -2.0, 0.5, 10, 14, 10, 2, 6, 300),
    ]
    for k, c in enumerate(cases):
        A, b, pid = S.build_A_data_matrix.func(*c, "nn", 0, 1) if hasattr(S.build_A_data_matrix, "func") else \
            S.build_A_data_matrix(*c, "nn", verbose=0, cpu=1)
        A = A.tocsr()
        A.sum_duplicates()
        A.sort_indices()
        arrs[f"adata{k}_image"] = c[0]
        arrs[f"adata{k}_args"] = np.asarray(c[1:], dtype=np.float64)
        arrs[f"adata{k}_indptr"], arrs[f"adata{k}_indices"], arrs[f"adata{k}_data"] = A.indptr, A.indices, A.data
        arrs[f"adata{k}_shape"] = np.asarray(A.shape)
        arrs[f"adata{k}_b"], arrs[f"adata{k}_pid"] = b, pid
    for k, c in enumerate([(8, 8, 8, 30.0, 2.0, 1, 0.0, 3.0, 50), (6, 12, 12, -41.5, 3.7, 2, 1.0, 5.0, 400)]):
        fn = getattr(S.build_A_helical_sym_matrix, "func", S.build_A_helical_sym_matrix)
        A, b = fn(*c, "nn", 0)
        A = A.tocsr()
        A.sort_indices()
        arrs[f"ahsym{k}_args"] = np.asarray(c, dtype=np.float64)
        arrs[f"ahsym{k}_indptr"], arrs[f"ahsym{k}_indices"], arrs[f"ahsym{k}_data"] = A.indptr, A.indices, A.data
        arrs[f"ahsym{k}_shape"] = np.asarray(A.shape)
    np.savez_compressed(OUT / "g4_path_a.npz", **arrs)


def g5_lsq():
    """lsq_reconstruct (model lsq, interpolation nn, cpu 1: the deterministic configuration) on the reference test's
    seed-42 rand(12, 12) (tests/test_denovo3D_solver.py:179-199) and on a 32 x 32 synthetic helix at three twists."""
    from helicon.webApps.denovo3D import solver_linear_regression as S

    arrs = {}
    np.random.seed(42)
    img = np.random.rand(12, 12).astype(np.float32)
    kw = dict(scale2d_to_3d=1.0, twist_degree=30.0, rise_pixel=2.0, csym=1, reconstruct_diameter_2d_pixel=8,
              reconstruct_diameter_3d_pixel=8, reconstruct_length_2d_pixel=8, reconstruct_length_3d_pixel=8,
              sym_oversample=1, interpolation="nn", algorithm=dict(model="lsq"), cpu=1)
    (rec, _, _), score = S.lsq_reconstruct(projection_image=img, **kw)
    arrs.update(seed42_image=img, seed42_rec3d=rec, seed42_score=np.asarray([score]),
                seed42_args=np.asarray([1.0, 30.0, 2.0, 1, 8, 8, 8, 8, 1], dtype=np.float64))
    n, apix = 32, 5.0
    d, br = 0.4 * n * apix, 2 * apix
    clean = utils.simulate_helical_projection(1, 29.0, 10.0, 1, d, br, 0, 0, n, n, apix)
    himg = (clean / clean.max()).astype(np.float32)
    arrs["helix_image"] = himg
    tws = np.asarray([25.0, 29.0, 33.0])
    scores = []
    for tw in tws:
        (rec, _, _), sc = S.lsq_reconstruct(projection_image=himg, scale2d_to_3d=1.0, twist_degree=float(tw), rise_pixel=2.0,
                                            csym=1, reconstruct_diameter_2d_pixel=20, reconstruct_diameter_3d_pixel=20,
                                            reconstruct_length_2d_pixel=32, reconstruct_length_3d_pixel=6, sym_oversample=1,
                                            interpolation="nn", algorithm=dict(model="lsq"), cpu=1)
        scores.append(sc)
        if tw == 29.0:
            arrs["helix_rec3d_29"] = rec
    arrs["helix_twists"] = tws
    arrs["helix_scores"] = np.asarray(scores)
    arrs["helix_args"] = np.asarray([1.0, 2.0, 1, 20, 20, 32, 6, 1], dtype=np.float64)
    np.savez_compressed(OUT / "g5_lsq.npz", **arrs)


def g4b_path_a_linear():
    """The trilinear branches of build_A_data_matrix (solver:1414-1503) and build_A_helical_sym_matrix (:910-1140), and
    lsq_reconstruct(interpolation="linear", model lsq) on the inputs of G4 / G5."""
    from helicon.webApps.denovo3D import solver_linear_regression as S

    arrs = {}
    cases = [
        (np.eye(8, dtype=np.float32), 1.0, 30.0, 2.0, 1, 0.0, 0.0, 0.0, 8, 8, 8, 0, 8, 64),
        (np.random.default_rng(4).random((12, 16)).astype(np.float32), 1.0, -41.5, 3.7, 2, 3.0, -2.0, 0.5, 10, 14, 10, 2, 6, 300),
        (np.random.default_rng(5).random((20, 24)).astype(np.float32), 0.8, 27.0, 2.3, 1, 0.0, 0.0, -0.25, 16, 20, 14, 0, 6, 1200),
    ]
    fn = getattr(S.build_A_data_matrix, "func", S.build_A_data_matrix)
    for k, c in enumerate(cases):
        A, b, pid = fn(*c, "linear", 0, 1)
        A = A.tocsr()
        A.sum_duplicates()
        A.sort_indices()
        arrs[f"adata{k}_image"] = c[0]
        arrs[f"adata{k}_args"] = np.asarray(c[1:], dtype=np.float64)
        arrs[f"adata{k}_indptr"], arrs[f"adata{k}_indices"], arrs[f"adata{k}_data"] = A.indptr, A.indices, A.data
        arrs[f"adata{k}_shape"] = np.asarray(A.shape)
        arrs[f"adata{k}_b"], arrs[f"adata{k}_pid"] = b, pid
    fn = getattr(S.build_A_helical_sym_matrix, "func", S.build_A_helical_sym_matrix)
    for k, c in enumerate([(8, 16, 16, 30.0, 2.0, 1, 0.0, 7.0, 200), (6, 20, 20, -41.5, 3.7, 2, 1.0, 9.0, 600)]):
        A, b = fn(*c, "linear", 0)
        A = A.tocsr()
        A.sort_indices()
        arrs[f"ahsym{k}_args"] = np.asarray(c, dtype=np.float64)
        arrs[f"ahsym{k}_indptr"], arrs[f"ahsym{k}_indices"], arrs[f"ahsym{k}_data"] = A.indptr, A.indices, A.data
        arrs[f"ahsym{k}_shape"] = np.asarray(A.shape)
    np.random.seed(42)
    img = np.random.rand(12, 12).astype(np.float32)
    (rec, _, _), score = S.lsq_reconstruct(projection_image=img, scale2d_to_3d=1.0, twist_degree=30.0, rise_pixel=2.0, csym=1,
                                           reconstruct_diameter_2d_pixel=8, reconstruct_diameter_3d_pixel=8,
                                           reconstruct_length_2d_pixel=8, reconstruct_length_3d_pixel=8, sym_oversample=1,
                                           interpolation="linear", algorithm=dict(model="lsq"), cpu=1)
    arrs.update(seed42_image=img, seed42_rec3d=rec, seed42_score=np.asarray([score]))
    n, apix = 32, 5.0
    d, br = 0.4 * n * apix, 2 * apix
    clean = utils.simulate_helical_projection(1, 29.0, 10.0, 1, d, br, 0, 0, n, n, apix)
    himg = (clean / clean.max()).astype(np.float32)
    arrs["helix_image"] = himg
    tws = np.asarray([25.0, 29.0, 33.0])
    scores = []
    for tw in tws:
        (rec, _, _), sc = S.lsq_reconstruct(projection_image=himg, scale2d_to_3d=1.0, twist_degree=float(tw), rise_pixel=2.0,
                                            csym=1, reconstruct_diameter_2d_pixel=20, reconstruct_diameter_3d_pixel=20,
                                            reconstruct_length_2d_pixel=32, reconstruct_length_3d_pixel=6, sym_oversample=1,
                                            interpolation="linear", algorithm=dict(model="lsq"), cpu=1)
        scores.append(sc)
        if tw == 29.0:
            arrs["helix_rec3d_29"] = rec
    arrs["helix_twists"] = tws
    arrs["helix_scores"] = np.asarray(scores)
    np.savez_compressed(OUT / "g4b_path_a_linear.npz", **arrs)


def g6_filters():
    rng = np.random.default_rng(6)
    x = rng.normal(size=(32, 32))
    arrs = dict(x=x)
    arrs["lp"] = filters.low_high_pass_filter(x, low_pass_fraction=0.3)
    arrs["hp"] = filters.low_high_pass_filter(x, high_pass_fraction=0.1)
    arrs["lphp"] = filters.low_high_pass_filter(x, low_pass_fraction=0.5, high_pass_fraction=0.05)
    # a second, larger image (the device path serves power-of-two sides 32..1024) with the pipeline's own
    # fractions (pipeline.py:183-188: low_pass 20 A at 2 A/pixel, high pass 2 / side)
    x64 = np.random.default_rng(66).normal(size=(64, 64)).astype(np.float32)
    arrs["x64"] = x64
    arrs["x64_lphp"] = filters.low_high_pass_filter(x64, low_pass_fraction=0.2, high_pass_fraction=2.0 / 64)
    arrs["thr_frac_0.2"] = filters.threshold_data(x, thresh_fraction=0.2)
    arrs["thr_frac_0"] = filters.threshold_data(x, thresh_fraction=0.0)
    arrs["thr_value_0.5"] = filters.threshold_data(x, thresh_value=0.5)
    arrs["thr_neg_in"] = -np.abs(x) - 0.25
    arrs["thr_neg_frac_0.5"] = filters.threshold_data(arrs["thr_neg_in"], thresh_fraction=0.5)
    arrs["norm_0_100"] = filters.normalize_percentile(x, (0, 100))
    arrs["norm_10_90"] = filters.normalize_percentile(x, (10, 90))
    arrs["periodic_in"] = np.asarray([-540.0, -181.0, -180.0, -0.5, 0.0, 179.99, 180.0, 180.5, 359.0, 725.0])
    arrs["periodic_out"] = np.asarray([angular.set_to_periodic_range(float(v), -180, 180) for v in arrs["periodic_in"]])
    np.savez_compressed(OUT / "g6_filters.npz", **arrs)


def _blob_volume(shape, seed):
    rng = np.random.default_rng(seed)
    nz, ny, nx = shape
    Z, Y, X = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    vol = np.zeros(shape)
    for _ in range(12):
        cz, cy, cx = rng.uniform(0.2, 0.8, 3) * np.asarray(shape)
        vol += rng.uniform(0.5, 1.5) * np.exp(-((Z - cz) ** 2 + (Y - cy) ** 2 + (X - cx) ** 2) / rng.uniform(3, 9))
    return vol.astype(np.float32)


def g7_helical_sym():
    """apply_helical_symmetry (transforms.py:58-165; pure-Python fallback of the numba kernel)."""
    cases = [
        # (shape, apix, twist, rise, csym, fraction, new_size, new_apix)
        ((20, 16, 16), 2.0, 30.0, 8.0, 1, 1.0, (20, 16, 16), None),
        ((18, 16, 14), 2.0, -41.5, 6.5, 3, 0.5, (26, 18, 18), 2.5),
        ((24, 20, 20), 1.5, 12.0, 4.75, 2, 1.0, (16, 14, 14), 1.5),
    ]
    arrs = {"n_cases": np.asarray([len(cases)])}
    for k, (shape, apix, tw, rs, cs, fr, ns, na) in enumerate(cases):
        vol = _blob_volume(shape, 70 + k)
        out = transforms.apply_helical_symmetry(vol, apix, tw, rs, csym=cs, fraction=fr, new_size=ns, new_apix=na, cpu=1)
        arrs[f"case{k}_in"] = vol
        arrs[f"case{k}_args"] = np.asarray([apix, tw, rs, cs, fr, *ns, -1.0 if na is None else na], dtype=np.float64)
        arrs[f"case{k}_out"] = np.asarray(out)
    np.savez_compressed(OUT / "g7_helical_sym.npz", **arrs)


def g8_rotate_shift():
    """helicon.rotate_shift_image (lib/transforms.py:315-369: scipy affine_transform, order 1, constant) and the app's
    is_vertical (webApps/denovo3D/utils.py:429-447)."""
    rng = np.random.default_rng(8)
    out = {}
    cases = [
        # (ny, nx, angle, pre_shift, post_shift, rotation_center)
        (32, 48, 7.5, (0, 0), (0, 0), None),
        (48, 32, -33.0, (1.5, -2.0), (0.25, 3.0), None),
        (40, 40, 90.0, (0, 0), (2, -1), None),
        (31, 45, 12.25, (0.5, 0.5), (0, 0), (10.0, 20.5)),
        (64, 64, 0.0, (3, 0), (0, -4), None),
        (25, 25, 180.0, (0, 0), (0, 0), None),
    ]
    for k, (ny, nx, ang, pre, post, rc) in enumerate(cases):
        img = rng.normal(size=(ny, nx)).astype(np.float32)
        got = transforms.rotate_shift_image(img, angle=ang, pre_shift=pre, post_shift=post,
                                            rotation_center=None if rc is None else np.array(rc))
        out[f"case{k}_image"] = img
        out[f"case{k}_args"] = np.array([ang, *pre, *post, *(rc if rc is not None else (np.nan, np.nan))], dtype=np.float64)
        out[f"case{k}_out"] = np.asarray(got)
    # is_vertical: a horizontal and a vertical bar, and noise images
    for k in range(6):
        img = rng.normal(size=(24, 36)).astype(np.float32)
        if k == 0:
            img[10:14, :] += 5
        if k == 1:
            img[:, 16:20] += 5
        out[f"vert{k}_image"] = img
        out[f"vert{k}"] = np.array([bool(utils.is_vertical(img))])
    np.savez_compressed(OUT / "g8_rotate_shift.npz", **out)
    print("g8_rotate_shift", len(cases))


def g16_refine_tilt_psi_dy():
    """refine_tilt_psi_dy (solver_linear_regression.py:550-841) on a 32 x 32 helix simulated with a tilt, an in-plane
    rotation and a shift: the refined parameters, the map and the score, for both projectors and both branches of its
    solver (positive: lsq_linear; unbounded: lsqr); and lsq_reconstruct with refine_tilt_psi_dy_range (solver:384-439) end to
    end.  The box (D2d = D3d = 20) keeps the number of rays the same under the perturbations — the function's own b is the
    first geometry's, so it raises as soon as a perturbed geometry gains or loses a ray."""
    from helicon.webApps.denovo3D import solver_linear_regression as S

    arrs = {}
    n, apix = 32, 5.0
    d, br = 0.4 * n * apix, 2 * apix
    kw = dict(scale2d_to_3d=1.0, twist_degree=29.0, rise_pixel=2.0, csym=1, reconstruct_diameter_2d_pixel=20,
              reconstruct_length_2d_pixel=32, reconstruct_diameter_3d_pixel=20, reconstruct_diameter_3d_inner_pixel=0,
              reconstruct_length_3d_pixel=6, sym_oversample=1)
    cases = [("nn", 1, (2.0, 1.5, 1.0)), ("nn", 0, (2.0, 1.5, 1.0)), ("linear", 1, (2.0, 1.5, 1.0)), ("nn", 1, (0.0, 6.0, -2.0))]
    for k, (interp, pos, (tilt, psi, dy)) in enumerate(cases):
        clean = utils.simulate_helical_projection(1, 29.0, 10.0, 1, d, br, 0, 0, n, n, apix, tilt=tilt, psi=psi, dy=dy * apix)
        himg = (clean / clean.max()).astype(np.float32)
        arrs[f"case{k}_image"] = himg
        arrs[f"case{k}_args"] = np.asarray([{"nn": 0, "linear": 1}[interp], pos, tilt, psi, dy], dtype=np.float64)
        try:
            t0, t1, t2, x, score = S.refine_tilt_psi_dy(projection_image=himg, interpolation=interp, x_init=None, positive_constraint=pos,
                                                        bounds_tilt=(-5.0, 5.0), bounds_psi=(-8.0, 8.0), bounds_dy=(-3.0, 3.0), verbose=0, cpu=1, **kw)
        except ValueError as e:   # (its b is the first geometry's: a geometry with another number of rays fails SciPy's shape check)
            arrs[f"case{k}_raised"] = np.asarray([1])
            print("g16 case", k, interp, pos, "raised", str(e)[:80], flush=True)
            continue
        arrs[f"case{k}_t"] = np.asarray([t0, t1, t2], dtype=np.float64)
        arrs[f"case{k}_x"] = np.asarray(x, dtype=np.float64)
        arrs[f"case{k}_score"] = np.asarray([score], dtype=np.float64)
        print("g16 case", k, interp, pos, [round(float(v), 5) for v in (t0, t1, t2)], round(float(score), 6), flush=True)
    # end to end: lsq_reconstruct with the refinement switched on (fsc_test 0: the score is the refinement's); unbounded, the
    # branch that went through above
    himg = arrs["case1_image"]
    if hasattr(S.lsq_reconstruct, "_refined_params"):
        S.lsq_reconstruct._refined_params = {}
    (rec, _, _), score = S.lsq_reconstruct(projection_image=himg, positive_constraint=0, interpolation="nn", algorithm=dict(model="lsq"), cpu=1,
                                           refine_tilt_psi_dy_range=dict(tilt=5.0, psi=8.0, dy=3.0), **{k_: v for k_, v in kw.items() if k_ != "reconstruct_diameter_3d_inner_pixel"})
    rp = getattr(S.lsq_reconstruct, "_refined_params", {})
    arrs["e2e_rec3d"] = rec
    arrs["e2e_score"] = np.asarray([score], dtype=np.float64)
    arrs["e2e_refined"] = np.asarray([rp.get("tilt", np.nan), rp.get("psi", np.nan), rp.get("dy", np.nan)], dtype=np.float64)
    arrs["kw"] = np.asarray([1.0, 29.0, 2.0, 1, 20, 32, 20, 0, 6, 1], dtype=np.float64)
    print("g16 e2e", round(float(score), 6), arrs["e2e_refined"], flush=True)
    np.savez_compressed(OUT / "g16_refine_tilt_psi_dy.npz", **arrs)


def g15_rotate_shift_cubic():
    """helicon.rotate_shift_image(order=3) (lib/transforms.py:315-369: scipy affine_transform, cubic spline, constant) — what
    auto_horizontalize's last step calls (webApps/denovo3D/utils.py:420-423) —, helicon.pad_to_size (lib/transforms.py:441-479)
    and helicon.set_to_periodic_range (lib/angular.py:84-108)."""
    rng = np.random.default_rng(15)
    out = {}
    cases = [
        (32, 48, 7.5, (0, 0), (0, 0)),
        (48, 32, -33.0, (1.5, -2.0), (0.25, 3.0)),
        (40, 40, 90.0, (0, 0), (2, -1)),
        (64, 64, 0.0, (0, 0), (-4.3, 0)),
        (25, 31, 1.75, (0, 0), (2.125, 0)),
    ]
    for k, (ny, nx, ang, pre, post) in enumerate(cases):
        img = rng.normal(size=(ny, nx)).astype(np.float32)
        got = transforms.rotate_shift_image(img, angle=ang, pre_shift=pre, post_shift=post, order=3)
        out[f"case{k}_image"] = img
        out[f"case{k}_args"] = np.array([ang, *pre, *post], dtype=np.float64)
        out[f"case{k}_out"] = np.asarray(got)
    for k, (shape, target) in enumerate([((5, 7), (8, 8)), ((6, 6), (6, 6)), ((7, 9), (8, 10)), ((4, 4), (7, 5))]):
        img = rng.normal(size=shape).astype(np.float32)
        out[f"pad{k}_image"] = img
        out[f"pad{k}_out"] = np.asarray(transforms.pad_to_size(img, target))
    vals = np.array([-540.0, -181.0, -180.0, -90.5, 0.0, 179.9, 180.0, 180.1, 271.0, 725.0])
    out["periodic_in"] = vals
    out["periodic_out"] = np.array([angular.set_to_periodic_range(float(v), min=-180, max=180) for v in vals])
    np.savez_compressed(OUT / "g15_rotate_shift_cubic.npz", **out)
    print("g15_rotate_shift_cubic", len(cases))


def g9_process_one_task():
    """The reference's own task function (webApps/denovo3D/pipeline.py:84-496) on a small helix, in the configuration the
    GPU build reproduces end to end: no rescale (target_apix2d = apix2d_orig), no tilt / psi / dy, model "lsq"."""
    from helicon.webApps.denovo3D import pipeline

    ny, nx, apix = 32, 48, 5.0
    img = utils.simulate_helical_projection(n=1, twist=29.0, rise=10.0, csym=1, helical_diameter=60.0, ball_radius=10.0,
                                            polymer=0, planarity=0, ny=ny, nx=nx, apix=apix)
    img = np.asarray(img, dtype=np.float32)
    img = img + np.random.default_rng(9).normal(0, 0.05 * img.std(), img.shape).astype(np.float32)
    out = {"image": img}
    cases = [
        # (twist, rise, csym, interpolation, thresh_fraction, target_apix3d, tube_diameter, low_pass)
        (29.0, 10.0, 1, "nn", -1, 5.0, 100.0, 0),
        (31.0, 10.0, 1, "nn", -1, 5.0, 100.0, 0),
        (29.0, 10.0, 1, "linear", -1, 0, 100.0, 0),
        (29.0, 10.0, 2, "nn", 0.05, 5.0, 120.0, 25.0),
    ]
    for k, (tw, rs, cs, interp, thr, a3, td, lp) in enumerate(cases):
        res = pipeline.process_one_task(0, 1, img.copy(), "mem", 1, tw, rs, (rs, rs), cs, 0.0, (0, 0), 0.0, 0, 0.0, 0,
                                        apix, "", lp, 0, 0, a3, apix, thr, -1, -1, td, 0, -1, 1, interp, 0, 1, "cosine",
                                        {"model": "lsq"}, 0, 1)
        score, ret, meta = res
        out[f"case{k}_args"] = np.array([tw, rs, cs, {"nn": 0, "linear": 1}[interp], thr, a3, td, lp], dtype=np.float64)
        out[f"case{k}_score"] = np.array([score], dtype=np.float64)
        out[f"case{k}_x_proj"] = np.asarray(ret[0])
        out[f"case{k}_y_proj"] = np.asarray(ret[1])
        out[f"case{k}_z_sections"] = np.asarray(ret[2])
        out[f"case{k}_rec3d"] = np.asarray(ret[3][0])
        out[f"case{k}_dims"] = np.array(ret[4:8], dtype=np.int64)
        out[f"case{k}_data_orig"] = np.asarray(meta[0])
        out[f"case{k}_meta"] = np.array([meta[3], meta[4], meta[5], meta[6], meta[7], meta[8], meta[9], meta[10]], dtype=np.float64)
        print("g9 case", k, score, ret[4:8], np.asarray(ret[0]).shape, np.asarray(ret[3][0]).shape)
    np.savez_compressed(OUT / "g9_process_one_task.npz", **out)


def g10_transform_map():
    """helicon.transform_map (lib/transforms.py:168-235: Rotation.from_euler("ZYZ") + scipy map_coordinates, cubic) on
    small volumes, and the reference's task function with tilt / psi / dy (the only place it resamples the map)."""
    from helicon.webApps.denovo3D import pipeline

    rng = np.random.default_rng(10)
    out = {}
    cases = [
        # (shape, scale, rot, tilt, psi, dx, dy, dz)
        ((12, 16, 20), 1.0, 0.0, 5.0, -7.0, 0.0, 1.5, 0.0),
        ((10, 14, 14), 1.1, 30.0, 10.0, 20.0, 1.0, -2.0, 0.5),
        ((9, 11, 13), 0.9, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0),
        ((16, 8, 8), 1.0, 0.0, 0.0, 12.0, 0.0, 0.0, 0.0),
    ]
    for k, (shape, sc, rot, tilt, psi, dx, dy, dz) in enumerate(cases):
        vol = rng.normal(size=shape).astype(np.float32)
        got = transforms.transform_map(vol, scale=sc, rot=rot, tilt=tilt, psi=psi, dx=dx, dy=dy, dz=dz)
        out[f"case{k}_vol"] = vol
        out[f"case{k}_args"] = np.array([sc, rot, tilt, psi, dx, dy, dz], dtype=np.float64)
        out[f"case{k}_out"] = np.asarray(got)
        print("g10 case", k, np.asarray(got).dtype, np.asarray(got).shape)
    ny, nx, apix = 32, 48, 5.0
    img = utils.simulate_helical_projection(n=1, twist=29.0, rise=10.0, csym=1, helical_diameter=60.0, ball_radius=10.0,
                                            polymer=0, planarity=0, ny=ny, nx=nx, apix=apix, tilt=3.0, psi=2.0, dy=5.0)
    img = np.asarray(img, dtype=np.float32)
    out["task_image"] = img
    score, ret, meta = pipeline.process_one_task(0, 1, img.copy(), "mem", 1, 29.0, 10.0, (10.0, 10.0), 1, 3.0, (0, 0), 2.0, 0, 5.0, 0,
                                                 apix, "", 0, 0, 0, 5.0, apix, -1, -1, -1, 100.0, 0, -1, 1, "nn", 0, 1, "cosine",
                                                 {"model": "lsq"}, 0, 1)
    out["task_score"] = np.array([score])
    out["task_x_proj"], out["task_y_proj"], out["task_z_sections"] = (np.asarray(v) for v in ret[:3])
    out["task_rec3d"] = np.asarray(ret[3][0])
    out["task_dims"] = np.array(ret[4:8], dtype=np.int64)
    print("g10 task", score, ret[4:8])
    np.savez_compressed(OUT / "g10_transform_map.npz", **out)


def g11_fsc_halves():
    """lsq_reconstruct with fsc_test 2, 3, 4 (solver:448-482, 526-547): the maps of the two pixel halves and the
    combined score s0 / 2 + (s1 + s2) / 4, on the helix of fixture G9."""
    from helicon.webApps.denovo3D.solver_linear_regression import lsq_reconstruct

    g9 = np.load(OUT / "g9_process_one_task.npz")
    img = g9["image"]
    out = {"image": img}
    for mode in (2, 3, 4):
        (rec, r1, r2), score = lsq_reconstruct(img, 1.0, 29.0, 2.0, 1, reconstruct_diameter_2d_pixel=20, reconstruct_diameter_3d_pixel=20,
                                               reconstruct_length_2d_pixel=48, reconstruct_length_3d_pixel=6, sym_oversample=1,
                                               interpolation="nn", fsc_test=mode, algorithm={"model": "lsq"})
        out[f"mode{mode}_score"] = np.array([score])
        out[f"mode{mode}_rec"], out[f"mode{mode}_rec1"], out[f"mode{mode}_rec2"] = rec, r1, r2
        print("g11 mode", mode, score)
    np.savez_compressed(OUT / "g11_fsc_halves.npz", **out)


def g12_polymer():
    """simulate_helical_projection(polymer=1) (utils.py:125-136 over random_polymer, :192-333), replayed from
    np.random.seed: the random-walk asymmetric unit itself (as the lattice code receives it) and the projection."""
    out = {}
    cases = [  # (seed, n, twist, rise, csym, diameter, ball_radius, planarity, ny, nx, apix, tilt, psi, dy)
        (7, 10, 30.0, 5.0, 1, 40.0, 3.0, 0.9, 32, 32, 2.0, 0, 0, 0),          # the reference's own test (test_denovo3D_utils.py:145)
        (8, 12, -20.0, 6.0, 2, 60.0, 3.0, 0.5, 48, 64, 2.0, 0, 0, 0),
        (9, 8, 41.0, 9.0, 3, 50.0, 2.5, 0.0, 64, 64, 2.0, 4.0, -3.0, 1.5),
    ]
    for k, c in enumerate(cases):
        seed, n, tw, rs, cs, d, br, pl, ny, nx, apix, tilt, psi, dy = c
        np.random.seed(seed)
        centers = utils.random_polymer(n_atoms=n, rmin=0, rmax=d / 2, csym=cs, planarity=pl)
        np.random.seed(seed)
        img = utils.simulate_helical_projection(n, tw, rs, cs, d, br, 1, pl, ny, nx, apix, tilt=tilt, psi=psi, dy=dy)
        out[f"case{k}_args"] = np.asarray(c, dtype=np.float64)
        out[f"case{k}_polymer"] = centers
        out[f"case{k}_out"] = img
        print("g12", k, centers.shape, float(img.max()))
    out["n_cases"] = np.asarray([len(cases)])
    np.savez_compressed(OUT / "g12_polymer.npz", **out)


def g13_fsc_random():
    """lsq_reconstruct with fsc_test = 1 (solver:186-189: the pixel ids as list(set(.)), np.random.shuffle, first half),
    replayed from np.random.seed, on the helix of fixture G9."""
    from helicon.webApps.denovo3D.solver_linear_regression import lsq_reconstruct

    g9 = np.load(OUT / "g9_process_one_task.npz")
    img = g9["image"]
    out = {"image": img}
    for k, seed in enumerate((3, 11)):
        np.random.seed(seed)
        (rec, r1, r2), score = lsq_reconstruct(img, 1.0, 29.0, 2.0, 1, reconstruct_diameter_2d_pixel=20, reconstruct_diameter_3d_pixel=20,
                                               reconstruct_length_2d_pixel=48, reconstruct_length_3d_pixel=6, sym_oversample=1,
                                               interpolation="nn", fsc_test=1, algorithm={"model": "lsq"})
        out[f"seed{k}"] = np.array([seed])
        out[f"seed{k}_score"] = np.array([score])
        out[f"seed{k}_rec"], out[f"seed{k}_rec1"], out[f"seed{k}_rec2"] = rec, r1, r2
        print("g13 seed", seed, score)
    np.savez_compressed(OUT / "g13_fsc_random.npz", **out)


def g14_sklearn_models():
    """lsq_reconstruct with the scikit-learn models of solve_equations (solver:270-342): elasticnet (the app's default,
    app.py:555-558), lasso, ridge, lreg — on the helix of fixture G5 at three twists, both projectors, positivity rule on.
    ElasticNet / Lasso use selection="random" with the global NumPy RNG: five seeds each, all scores stored (their spread is
    the reference's own run-to-run band) and the seed-0 map (the solution, for its objective value).  lreg (dense NNLS, 200 s
    per call without numba) only at the true twist."""
    from helicon.webApps.denovo3D.solver_linear_regression import lsq_reconstruct
    import sklearn

    g5 = np.load(OUT / "g5_lsq.npz")
    img = g5["helix_image"]
    kw = dict(reconstruct_diameter_2d_pixel=20, reconstruct_diameter_3d_pixel=20, reconstruct_length_2d_pixel=32,
              reconstruct_length_3d_pixel=6, sym_oversample=1)
    out = {"image": img, "twists": np.asarray([25.0, 29.0, 33.0]), "seeds": np.arange(5), "sklearn_version": np.asarray(sklearn.__version__)}
    for model in ("elasticnet", "lasso", "ridge", "lreg"):
        for interp in ("nn", "linear"):
            twists = (29.0,) if model == "lreg" else (25.0, 29.0, 33.0)
            seeds = range(5) if model in ("elasticnet", "lasso") else range(1)
            scores = np.zeros((len(twists), len(list(seeds))))
            for ti, tw in enumerate(twists):
                for si, seed in enumerate(seeds):
                    np.random.seed(seed)
                    (rec, _, _), score = lsq_reconstruct(img.copy(), 1.0, tw, 2.0, 1, interpolation=interp,
                                                         algorithm=dict(model=model, l1_ratio=0.5), **kw)
                    scores[ti, si] = score
                    if si == 0:
                        out[f"{model}_{interp}_rec_{int(tw)}"] = rec
            out[f"{model}_{interp}_scores"] = scores
            print("g14", model, interp, scores.round(6).tolist())
    np.savez_compressed(OUT / "g14_sklearn_models.npz", **out)


if __name__ == "__main__":
    assert "reference" in os.path.abspath(helicon.__file__), helicon.__file__
    makers = [g1_simulate, g2_scores, g3_composed, g3b_general_sizes, g3c_general_sizes_tilted, g4_path_a, g4b_path_a_linear, g5_lsq, g6_filters,
              g7_helical_sym, g8_rotate_shift, g9_process_one_task, g10_transform_map, g11_fsc_halves, g12_polymer, g13_fsc_random, g15_rotate_shift_cubic, g16_refine_tilt_psi_dy,
              g14_sklearn_models]
    only = set(sys.argv[1:])   # e.g. "g8_rotate_shift": regenerate just that fixture
    for make in makers:
        if not only or make.__name__ in only:
            make()
    (OUT / "VERSIONS.json").write_text(json.dumps(versions(), indent=1) + "\n")
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size)
