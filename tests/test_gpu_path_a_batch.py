"""Path A for many candidates at once (hh_pab_*, helicon_amd.lsq_reconstruct_batch) against the reference's own
lsq_reconstruct outputs (fixture G5), the NumPy / SciPy oracle, and itself: a candidate's result must not depend on
which other candidates share its batch, nor on the run.  solver_linear_regression.py:31-547; app.py:2473-2476."""
import numpy as np
import pytest

from helicon_amd.solver import PathABatch, PathAProblem, hh_pa_params, lsq_reconstruct, lsq_reconstruct_batch
from oracle import path_a as A

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["sliced", "general"])
def product_path(request, monkeypatch):
    """Both product implementations: slice-major with LDS staging (tilt = psi = 0) and the general gather path
    (HH_PAB_GENERAL forces it where the sliced one would apply)."""
    if request.param == "general":
        monkeypatch.setenv("HH_PAB_GENERAL", "1")
    else:
        monkeypatch.delenv("HH_PAB_GENERAL", raising=False)
    return request.param


def _helix_kw(g):
    s2, rs, cs, d2, d3, l2, l3, ov = g["helix_args"]
    return float(s2), float(rs), int(cs), dict(reconstruct_diameter_2d_pixel=int(d2), reconstruct_diameter_3d_pixel=int(d3),
                                               reconstruct_length_2d_pixel=int(l2), reconstruct_length_3d_pixel=int(l3),
                                               sym_oversample=ov)


def test_batch_reproduces_the_reference_scores(golden_dir, product_path):
    """G5's three twists in ONE batch: the reference's scores to 1e-4, the 29-degree volume to the solver's tolerance."""
    g = np.load(golden_dir / "g5_lsq.npz")
    s2, rs, cs, kw = _helix_kw(g)
    cands = [(float(tw), rs, cs) for tw in g["helix_twists"]]
    stats = {}
    res = lsq_reconstruct_batch(g["helix_image"], s2, cands, stats=stats, **kw)
    for (maps, score), want, tw in zip(res, g["helix_scores"], g["helix_twists"]):
        assert score == pytest.approx(float(want), abs=1e-4), tw
        if tw == 29.0:
            assert np.abs(maps[0] - g["helix_rec3d_29"]).max() < 5e-3 * np.abs(g["helix_rec3d_29"]).max()
    assert stats["launches"] > 0 and stats["host_syncs"] < stats["launches"] / 8    # the host only polls
    assert all(st in (1, 2, 3, -1, 0) for st, *_ in stats["info"])


def test_batch_composition_and_runs_do_not_change_a_candidate(golden_dir):
    """The same candidate alone, among 2 others, among 11 others, twice: bit-identical scores and maps (no quantity of
    a candidate depends on another's; every sum has one order)."""
    g = np.load(golden_dir / "g5_lsq.npz")
    s2, rs, cs, kw = _helix_kw(g)
    twists = [25.0 + 0.75 * k for k in range(12)]
    big = lsq_reconstruct_batch(g["helix_image"], s2, [(t, rs, cs) for t in twists], **kw)
    again = lsq_reconstruct_batch(g["helix_image"], s2, [(t, rs, cs) for t in twists], **kw)
    small = lsq_reconstruct_batch(g["helix_image"], s2, [(t, rs, cs) for t in twists[4:7]], **kw)
    split = lsq_reconstruct_batch(g["helix_image"], s2, [(t, rs, cs) for t in twists], batch=5, **kw)
    for k in range(12):
        assert big[k][1] == again[k][1] == split[k][1]
        np.testing.assert_array_equal(big[k][0][0], again[k][0][0])
        np.testing.assert_array_equal(big[k][0][0], split[k][0][0])
    for k in range(3):
        assert small[k][1] == big[4 + k][1]
        np.testing.assert_array_equal(small[k][0][0], big[4 + k][0][0])
    one = lsq_reconstruct(g["helix_image"], s2, twists[5], rs, cs, **kw)
    assert one[1] == big[5][1]
    np.testing.assert_array_equal(one[0][0], big[5][0][0])


def test_batch_against_the_oracle_with_mixed_candidates(golden_dir, product_path):
    """Candidates that differ in twist, rise AND csym, bounded and unbounded, clipped prediction: each equals the oracle's
    lsq_reconstruct (scores 1e-4; the unbounded ones 1e-5, they are a single LSMR solve)."""
    g = np.load(golden_dir / "g5_lsq.npz")
    img = g["helix_image"]
    kw = dict(reconstruct_diameter_2d_pixel=20, reconstruct_diameter_3d_pixel=20, reconstruct_length_2d_pixel=32,
              reconstruct_length_3d_pixel=6)
    cands = [(29.0, 2.0, 1), (31.0, 2.5, 1), (58.0, 4.0, 2), (-29.0, 2.0, 1), (27.5, 1.7, 1)]
    for pc, tol in ((-1, 1e-4), (0, 1e-5), (1, 1e-4)):
        res = lsq_reconstruct_batch(img, 1.0, cands, positive_constraint=pc, thresh_fraction=0.0, **kw)
        for (maps, score), (tw, rs, cs) in zip(res, cands):
            (rec_o, _, _), s_o = A.lsq_reconstruct(img, 1.0, tw, rs, cs, positive_constraint=pc, thresh_fraction=0.0, **kw)
            assert score == pytest.approx(s_o, abs=tol), (pc, tw, rs, cs)
            assert np.abs(maps[0] - rec_o).max() < 1e-2 * max(1e-6, np.abs(rec_o).max()), (pc, tw, rs, cs)


def test_batch_with_tilt_psi_dy_against_the_oracle(golden_dir):
    """Out-of-plane tilt, in-plane rotation and a shift: rays cross slices, so this is the general product path."""
    g = np.load(golden_dir / "g5_lsq.npz")
    img = g["helix_image"]
    kw = dict(reconstruct_diameter_2d_pixel=20, reconstruct_diameter_3d_pixel=20, reconstruct_length_2d_pixel=32,
              reconstruct_length_3d_pixel=6, tilt_degree=3.0, psi_degree=-2.0, dy_pixel=0.75)
    cands = [(29.0, 2.0, 1), (30.0, 2.2, 1)]
    res = lsq_reconstruct_batch(img, 1.0, cands, **kw)
    for (maps, score), (tw, rs, cs) in zip(res, cands):
        (rec_o, _, _), s_o = A.lsq_reconstruct(img, 1.0, tw, rs, cs, **kw)
        assert score == pytest.approx(s_o, abs=1e-4), (tw, rs)
        assert np.abs(maps[0] - rec_o).max() < 1e-2 * np.abs(rec_o).max()


def test_batch_rows_equal_the_single_candidate_problem(golden_dir):
    """The batch's set-up (rays that exist, b, pixel ids, symmetry pairs) is the single-candidate hh_pa's, candidate by
    candidate, including the half sets."""
    g = np.load(golden_dir / "g5_lsq.npz")
    img = np.ascontiguousarray(g["helix_image"], dtype=np.float32)
    base = dict(scale2d_to_3d=1.0, csym=1, tilt_degree=2.0, psi_degree=1.0, dy_pixel=0.5, reconstruct_diameter_2d_pixel=20,
                reconstruct_length_2d_pixel=32, reconstruct_diameter_3d_pixel=20, reconstruct_diameter_3d_inner_pixel=4,
                reconstruct_length_3d_pixel=6, min_projection_lines=700, min_sym_pairs=700)
    specs = [(29.0, 2.0, 0, 0), (31.0, 2.5, 3, 1), (31.0, 2.5, 3, 2), (27.0, 1.5, 2, 1)]
    params = [hh_pa_params(1.0, tw, rs, 1, 2.0, 1.0, 0.5, 20, 32, 20, 4, 6, 700, 700, 0, mode, half) for tw, rs, mode, half in specs]
    with PathABatch(img, params) as B:
        for c, (tw, rs, mode, half) in enumerate(specs):
            with PathAProblem(img, twist_degree=tw, rise_pixel=rs, fsc_mode=mode, fsc_half=half, **base) as P:
                assert (B.n, int(B.m_data[c]), int(B.m_sym[c]), int(B.n_ops[c])) == (P.n, P.m_data, P.m_sym, P.n_ops)
                b, pid = B.rhs(c)
                np.testing.assert_array_equal(b, P.b_data)
                np.testing.assert_array_equal(pid, P.b_pid)


def test_batch_rejects_what_it_does_not_do():
    img = np.ones((16, 16), dtype=np.float32)
    ok = hh_pa_params(1.0, 29.0, 2.0, 1, 0.0, 0.0, 0.0, 12, 16, 12, 0, 4, 100, 100, 0, 0, 0)
    lin = hh_pa_params(1.0, 29.0, 2.0, 1, 0.0, 0.0, 0.0, 12, 16, 12, 0, 4, 100, 100, 1, 0, 0)
    other_box = hh_pa_params(1.0, 29.0, 2.0, 1, 0.0, 0.0, 0.0, 12, 16, 10, 0, 4, 100, 100, 0, 0, 0)
    with pytest.raises(ValueError):
        PathABatch(img, [ok, lin])
    with pytest.raises(ValueError):
        PathABatch(img, [ok, other_box])


def test_device_symmetry_pairs_equal_the_sequential_rule(golden_dir, monkeypatch):
    """build_A_helical_sym_matrix's order-dependent de-duplication (solver:1142-1298), built on the device as "smallest
    walk index per unordered pair" (hash table + atomicMin), against the host's sequential restatement that the
    single-candidate hh_pa uses (itself bit-exact against the reference's matrix, fixture G4): same pairs, same order —
    several pairs of operations, csym 2 and 3, negative twist, half-integer rise, an inner radius, early stop."""
    g = np.load(golden_dir / "g5_lsq.npz")
    img = np.ascontiguousarray(g["helix_image"], dtype=np.float32)
    specs = [(29.0, 2.0, 1, 700), (31.0, 2.5, 1, 2000), (58.0, 4.0, 2, 1500), (-29.0, 2.0, 1, 50), (40.0, 1.0, 3, 5000),
             (27.5, 0.7, 1, 100000)]
    params = [hh_pa_params(1.0, tw, rs, cs, 0.0, 0.0, 0.0, 20, 32, 20, 4, 6, 700, want, 0, 0, 0) for tw, rs, cs, want in specs]
    monkeypatch.delenv("HH_PAB_HOST_SYM", raising=False)
    with PathABatch(img, params) as B:
        dev = [B.sym_pairs(c) for c in range(len(specs))]
    monkeypatch.setenv("HH_PAB_HOST_SYM", "1")
    with PathABatch(img, params) as B:
        host = [B.sym_pairs(c) for c in range(len(specs))]
    for d, h, sp in zip(dev, host, specs):
        assert len(h) > 0
        np.testing.assert_array_equal(d, h, err_msg=str(sp))


def test_full_load_is_reproducible():
    """128 candidates at the reference app's working size (64 x 128 image, 48k unknowns, ~100k rows each) fill the
    device; two solves of the same object must agree bit for bit and the solver's self-check must stay at zero.  (Round 3
    found the per-candidate state read stale across launches from a compute unit's own cache — one candidate-solve in a
    thousand took one LSMR iteration more; the state now crosses launches through agent-scope loads and stores.)"""
    import helicon_amd as H

    ny, nx, l3, target = 64, 128, 16, 47952
    eng = H.SweepEngine((ny, nx))
    eng.set_geometry(apix=5.0, helical_diameter=0.5 * ny * 5.0, ball_radius=10.0)
    img = eng.simulate(29.0, 20.0, 1).astype(np.float32)
    k = 128
    params = [hh_pa_params(1.0, float(t), 4.0, 1, 0.0, 0.0, 0.0, ny, nx, ny, 0, l3, target, target, 0, 0, 0)
              for t in np.linspace(27.0, 31.0, k)]
    with PathABatch(img, params) as B:
        a = B.solve(np.ones(k, dtype=np.int32), 0)
        ca = B.counters()
        b = B.solve(np.ones(k, dtype=np.int32), 0)
        cb = B.counters()
    assert ca["self_check_failures"] == 0 and cb["self_check_failures"] == 0
    np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    assert int(np.argmax(a[1])) in range(k // 2 - 4, k // 2 + 4)       # the truth (29 degrees) is in the middle of the list


def test_device_row_enumeration_equals_the_host_one(golden_dir, monkeypatch):
    """Which rays exist, their order, b and the pixel ids: enumerated on the device from per-slice counts (no per-ray data
    over PCIe) against the host loop that the half-set batches still use — same tables, and bit-identical solves (the
    device order of the rows is the same stable slice-major sort), on the sliced path and, with tilt / psi, the general one."""
    g = np.load(golden_dir / "g5_lsq.npz")
    img = np.ascontiguousarray(g["helix_image"], dtype=np.float32)
    for tilt, psi, dy in ((0.0, 0.0, 0.0), (0.0, 0.0, 0.75), (3.0, -2.0, 0.5)):
        specs = [(29.0, 2.0, 1, 700), (31.0, 2.5, 1, 2000), (58.0, 4.0, 2, 1500), (-29.0, 2.0, 1, 300), (27.5, 0.7, 3, 4000)]
        params = [hh_pa_params(1.0, tw, rs, cs, tilt, psi, dy, 20, 32, 20, 0, 6, want, want, 0, 0, 0) for tw, rs, cs, want in specs]
        out = {}
        for mode in ("device", "host"):
            if mode == "host":
                monkeypatch.setenv("HH_PAB_HOST_ROWS", "1")
            else:
                monkeypatch.delenv("HH_PAB_HOST_ROWS", raising=False)
            with PathABatch(img, params) as B:
                rhs = [B.rhs(c) for c in range(len(specs))]
                x, scores, info = B.solve(np.ones(len(specs), dtype=np.int32), 0)
                out[mode] = (B.m_data.copy(), B.n_ops.copy(), rhs, x, scores, info)
        d, h = out["device"], out["host"]
        np.testing.assert_array_equal(d[0], h[0])
        np.testing.assert_array_equal(d[1], h[1])
        for (bd, pd), (bh, ph) in zip(d[2], h[2]):
            np.testing.assert_array_equal(bd, bh)
            np.testing.assert_array_equal(pd, ph)
        np.testing.assert_array_equal(d[3], h[3])
        np.testing.assert_array_equal(d[4], h[4])
        np.testing.assert_array_equal(d[5], h[5])


def test_trilinear_batch_against_the_single_candidate_projector_and_the_oracle(golden_dir):
    """interpolation="linear" (the app's default, app.py:577-585) through the group solver (round 4: slice-major 2 x 2 x 2
    products, rays recomputed in float64 — csrc/path_a_linear.inc) against (a) the single-candidate hh_pa projector, whose
    matrices equal the reference's entry for entry (fixture G4b, tests/test_gpu_path_a.py): the products themselves to 1e-12
    (hh_pab_matvec / hh_pab_rmatvec: the solve's own kernels), UNBOUNDED solves — one LSMR solve each at lsq_linear's loose
    tolerance, where a norm's last bit can cost or save an iteration: 1e-4 — and (b) the float64 oracle and the reference's own scores (G4b) on the bounded
    ones at the tolerance that solve allows (2e-3, see test_lsq_reconstruct_trilinear)."""
    g = np.load(golden_dir / "g4b_path_a_linear.npz")
    kw = dict(reconstruct_diameter_2d_pixel=20, reconstruct_diameter_3d_pixel=20, reconstruct_length_2d_pixel=32,
              reconstruct_length_3d_pixel=6, sym_oversample=1)
    img = g["helix_image"]
    cands = [(float(tw), 2.0, 1) for tw in g["helix_twists"]] + [(58.0, 4.0, 2), (-29.0, 2.0, 1), (27.5, 1.7, 1), (31.0, 2.5, 1)]
    stats = {}
    rng = np.random.default_rng(5)
    params = [hh_pa_params(1.0, tw, rs, cs, 0.0, 0.0, 0.0, 20, 32, 20, 0, 6, 960, 960, 1, 0, 0) for tw, rs, cs in cands]
    with PathABatch(img, params) as B:
        for c, (tw, rs, cs) in enumerate(cands):
            with PathAProblem(img, scale2d_to_3d=1.0, twist_degree=tw, rise_pixel=rs, csym=cs, tilt_degree=0, psi_degree=0, dy_pixel=0,
                              reconstruct_diameter_2d_pixel=20, reconstruct_length_2d_pixel=32, reconstruct_diameter_3d_pixel=20,
                              reconstruct_diameter_3d_inner_pixel=0, reconstruct_length_3d_pixel=6, min_projection_lines=960,
                              min_sym_pairs=960, interpolation="linear") as P:
                assert (B.n, int(B.m_data[c]), int(B.m_sym[c]), int(B.n_ops[c])) == (P.n, P.m_data, P.m_sym, P.n_ops)
                b, pid = B.rhs(c)
                np.testing.assert_array_equal(b, P.b_data)
                np.testing.assert_array_equal(pid, P.b_pid)
                x, y = rng.normal(size=P.n), rng.normal(size=P.m)
                np.testing.assert_allclose(B.matvec(c, x), P.matvec(x), rtol=0, atol=1e-12)
                np.testing.assert_allclose(B.rmatvec(c, y), P.rmatvec(y), rtol=0, atol=1e-12)
    free = lsq_reconstruct_batch(img, 1.0, cands, positive_constraint=0, interpolation="linear", stats=stats, **kw)
    assert stats.get("path") != "hh_pa" and stats["self_check_failures"] == 0 and stats["launches"] > 0
    for (maps, score), (tw, rs, cs) in zip(free, cands):
        (rec_1, _, _), s_1 = lsq_reconstruct(img, 1.0, tw, rs, cs, positive_constraint=0, interpolation="linear", _single=True, **kw)
        assert score == pytest.approx(s_1, abs=1e-4), (tw, rs, cs)
        assert np.abs(maps[0] - rec_1).max() < 1e-2 * np.abs(rec_1).max(), (tw, rs, cs)
    (rec_o, _, _), s_o = A.lsq_reconstruct(img, 1.0, 29.0, 2.0, 1, positive_constraint=0, interpolation="linear", **kw)
    assert free[1][1] == pytest.approx(s_o, abs=1e-4)
    bounded = lsq_reconstruct_batch(img, 1.0, cands[:3], interpolation="linear", **kw)
    for (maps, score), want, (tw, rs, cs) in zip(bounded, g["helix_scores"], cands):
        (rec_o, _, _), s_o = A.lsq_reconstruct(img, 1.0, tw, rs, cs, interpolation="linear", **kw)
        assert score == pytest.approx(s_o, abs=2e-3) and score == pytest.approx(float(want), abs=2e-3), tw
        assert A.cosine_similarity(maps[0].ravel(), rec_o.ravel()) > 0.995
    assert int(np.argmax([s for _, s in bounded])) == 1


def test_trilinear_batch_composition_runs_and_half_sets(golden_dir):
    """The trilinear group solver: a candidate's result does not depend on its batch or on the run (bit for bit), the
    single call is a batch of one, return_3d=False gives the same scores, fsc_test = 2 returns three maps and the
    combined score of its three solves."""
    g = np.load(golden_dir / "g5_lsq.npz")
    s2, rs, cs, kw = _helix_kw(g)
    twists = [25.0 + 0.75 * k for k in range(12)]
    cands = [(t, rs, cs) for t in twists]
    big = lsq_reconstruct_batch(g["helix_image"], s2, cands, interpolation="linear", **kw)
    again = lsq_reconstruct_batch(g["helix_image"], s2, cands, interpolation="linear", **kw)
    split = lsq_reconstruct_batch(g["helix_image"], s2, cands, interpolation="linear", batch=5, streams=3, **kw)
    none = lsq_reconstruct_batch(g["helix_image"], s2, cands, interpolation="linear", return_3d=False, **kw)
    for k in range(12):
        assert big[k][1] == again[k][1] == split[k][1] == none[k][1]
        np.testing.assert_array_equal(big[k][0][0], again[k][0][0])
        np.testing.assert_array_equal(big[k][0][0], split[k][0][0])
        assert none[k][0] == (None, None, None)
    one = lsq_reconstruct(g["helix_image"], s2, twists[5], rs, cs, interpolation="linear", **kw)
    assert one[1] == big[5][1]
    np.testing.assert_array_equal(one[0][0], big[5][0][0])
    (full, h1, h2), score = lsq_reconstruct(g["helix_image"], s2, 29.0, rs, cs, interpolation="linear", fsc_test=2, **kw)
    assert h1 is not None and h2 is not None and h1.shape == full.shape
    s_full = lsq_reconstruct(g["helix_image"], s2, 29.0, rs, cs, interpolation="linear", **kw)[1]
    assert abs(score - s_full) < 0.1 and not np.array_equal(h1, h2)


def test_trilinear_with_tilt_takes_the_single_candidate_path(golden_dir):
    """Out-of-plane tilt: a ray's samples cross cell layers, the group solver has no such products — the candidates go
    through hh_pa from a thread pool, same results in list order."""
    g = np.load(golden_dir / "g5_lsq.npz")
    s2, rs, cs, kw = _helix_kw(g)
    cands = [(27.0, rs, cs), (29.0, rs, cs)]
    stats = {}
    res = lsq_reconstruct_batch(g["helix_image"], s2, cands, tilt_degree=3.0, interpolation="linear", streams=2, stats=stats, **kw)
    assert stats["path"] == "hh_pa"
    for (maps, score), (tw, r, c) in zip(res, cands):
        (want_map, _, _), want = lsq_reconstruct(g["helix_image"], s2, tw, r, c, tilt_degree=3.0, interpolation="linear", **kw)
        assert score == want
        np.testing.assert_array_equal(maps[0], want_map)


def test_full_load_trilinear_and_concurrent_groups_are_reproducible():
    """ADVICE (round 3): the load at which stale per-candidate state had shown (K = 256) and the shape that ships (several
    groups on several streams) had no reproducibility test.  256 nearest-neighbour candidates in one group, twice; 512
    candidates as 4 groups on 4 streams against one group on one stream (nn), and 128 trilinear candidates two ways: the
    scores must agree bit for bit and the solver's self-check must stay at zero (it is an error now: hh_pab_solve returns
    HH_ERR_STATE)."""
    import helicon_amd as H

    ny, nx, l3 = 64, 128, 16
    eng = H.SweepEngine((ny, nx))
    eng.set_geometry(apix=5.0, helical_diameter=0.5 * ny * 5.0, ball_radius=10.0)
    img = eng.simulate(29.0, 20.0, 1).astype(np.float32)
    kw = dict(reconstruct_diameter_2d_pixel=ny, reconstruct_diameter_3d_pixel=ny, reconstruct_length_2d_pixel=nx,
              reconstruct_length_3d_pixel=l3, return_3d=False)

    def scores(n, interp, batch, streams):
        st = {}
        res = lsq_reconstruct_batch(img, 1.0, [(float(t), 4.0, 1) for t in np.linspace(27.0, 31.0, n)], interpolation=interp,
                                    batch=batch, streams=streams, stats=st, **kw)
        assert st["self_check_failures"] == 0
        return np.array([s for _, s in res]), np.array(st["info"])

    a, ia = scores(256, "nn", 256, 1)
    b, ib = scores(256, "nn", 256, 1)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(ia, ib)
    c, ic = scores(512, "nn", 128, 4)
    d, id_ = scores(512, "nn", 512, 1)
    np.testing.assert_array_equal(c, d)
    np.testing.assert_array_equal(ic, id_)
    e, _ = scores(128, "linear", 32, 4)
    f, _ = scores(128, "linear", 128, 1)
    np.testing.assert_array_equal(e, f)
    assert abs(int(np.argmax(a)) - 128) <= 8 and abs(int(np.argmax(f)) - 64) <= 6       # the truth (29 degrees) is mid-list
