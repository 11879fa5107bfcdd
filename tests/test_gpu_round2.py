"""Round-2 GPU parity: the BASELINE configurations at their own sizes (C4: 1024^2 fused pipeline against
the oracle; C5: 64 segments x 512^2 through the fused pass + the MFMA contraction), the drop-in task
function under a thread pool, several contexts in one process, the device arg-max and the filtered
spectrum.  Everything goes through the C ABI (ctypes); tolerances as in test_gpu_parity.py.
"""
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import helicon_amd as H
from helicon_amd.grid import build_grid
from oracle import path_b as O

pytestmark = pytest.mark.gpu

SCORE_TOL = 2e-4


def _clean_and_noisy(eng, truth, seeds, sigma=0.5):
    clean = eng.simulate(*truth)
    imgs = [(clean + np.random.default_rng(s).normal(0, sigma * clean.std(), clean.shape)).astype(np.float32)
            for s in seeds]
    return clean, np.stack(imgs)


# ---------------------------------------------------------------------------- C5 at its own size
def test_c5_64_segments_at_512_fused_and_contraction_against_oracle():
    """BASELINE config 5 at size: 64 noisy segments x 512^2 against one twist-major sub-grid (3 twists x 64
    rises) through the fused pass with q stores + k_segment_corr (K = 131,584 bins per candidate).  Every
    segment's row equals that segment's single-segment sweep, 10 (segment, candidate) pairs equal the oracle,
    and every segment's arg-max is the truth (analysis.py:777-799 via alignment.py:144-147)."""
    n, apix = 512, 1.0
    d, br = 0.4 * n * apix, 2 * apix
    twists = np.array([1.19, 1.20, 1.21])
    rises = 4.75 + 0.005 * np.arange(-32, 32)            # 64 per run
    grid = build_grid(twists, rises, (1,), tube_length=n * apix)
    truth = 1 * 64 + 32
    assert tuple(grid.params[truth, :2]) == (1.20, 4.75)
    with H.SweepEngine(n) as eng:
        eng.set_geometry(apix=apix, helical_diameter=d, ball_radius=br)
        _, imgs = _clean_and_noisy(eng, (1.20, 4.75, 1), range(64))
        eng.set_reference(imgs)
        multi = eng.sweep(grid.params)
        assert eng.last_first_pass == "fused" and multi.shape == (64, 192)
        assert np.array_equal(multi, eng.sweep(grid.params))          # bit-reproducible
        eng.set_table_path(1)                                           # second pass with q stores + contraction
        multi_two = eng.sweep(grid.params)
        assert eng.last_first_pass == "run_tables"
        eng.set_table_path(2)
        np.testing.assert_allclose(multi, multi_two, rtol=0, atol=2e-5)
        for s in range(64):
            eng.set_reference(imgs[s])
            single = eng.sweep(grid.params)[0]
            np.testing.assert_allclose(multi[s], single, rtol=0, atol=2e-6, err_msg=f"segment {s}")
    assert (np.argmax(multi, axis=1) == truth).all()
    mask = O.radial_band_mask(n, n)
    pairs = [(0, truth), (0, 0), (7, 191), (13, 64), (31, 95), (32, 100), (40, 17), (63, truth), (63, 128), (50, 63)]
    for s, g in pairs:
        ref = O.sweep_cpu(imgs[s], grid.params[[g], :3], mask, apix=apix, helical_diameter=d, ball_radius=br)[0]
        assert abs(multi[s, g] - ref) < SCORE_TOL, (s, g, multi[s, g], ref)


# ---------------------------------------------------------------------------- C4 at its own size
def test_c4_1024_every_pipeline_against_oracle():
    """BASELINE config 4's image size through all three pipelines against the oracle directly: 14 candidates
    of a twist-major sub-grid (2 twists x 7 rises... padded to runs of 8 so the shared-twist planner takes it).
    N = 1024 is the only instantiation whose transforms span two wavefronts (workgroup barriers in the
    exchanges)."""
    n, apix = 1024, 1.0
    d, br = 0.4 * n * apix, 2 * apix
    twists = np.array([2.38, 2.40])
    rises = 9.5 + 0.01 * np.arange(-4, 4)                 # 8 per run
    grid = build_grid(twists, rises, (2,), tube_length=n * apix)
    truth = 1 * 8 + 4
    with H.SweepEngine(n) as eng:
        eng.set_geometry(apix=apix, helical_diameter=d, ball_radius=br)
        _, imgs = _clean_and_noisy(eng, (2.40, 9.5, 2), [3])
        eng.set_reference(imgs[0])
        fused = eng.sweep(grid.params)[0]
        assert eng.last_first_pass == "fused"
        eng.set_table_path(1)
        tables = eng.sweep(grid.params)[0]
        assert eng.last_first_pass == "run_tables"
        eng.set_table_path(0)
        transform = eng.sweep(grid.params)[0]
        assert eng.last_first_pass == "transform"
    pick = np.array([0, 1, 2, 3, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15])
    ref = O.sweep_cpu(imgs[0], grid.params[pick, :3], O.radial_band_mask(n, n), apix=apix, helical_diameter=d,
                      ball_radius=br)
    for name, got in (("fused", fused), ("run_tables", tables), ("transform", transform)):
        np.testing.assert_allclose(got[pick], ref, rtol=0, atol=SCORE_TOL, err_msg=name)
        assert int(np.argmax(got)) == truth, name


# ---------------------------------------------------------------------------- contexts and devices
def test_two_contexts_on_one_device_keep_their_own_kernel_attributes():
    """The dynamic-LDS limit of the big-LDS kernels is tracked per context (it is a per-device attribute): a
    second context created after the first has already launched must launch the fused pass (> 64 KB of LDS)
    and get identical scores; destroying one context leaves the other usable."""
    n, apix = 512, 1.0
    d, br = 0.4 * n * apix, 2 * apix
    grid = build_grid(np.array([1.19, 1.20]), 4.75 + 0.005 * np.arange(-8, 8), (1,), tube_length=n * apix)
    a = H.SweepEngine(n)
    a.set_geometry(apix=apix, helical_diameter=d, ball_radius=br)
    _, imgs = _clean_and_noisy(a, (1.20, 4.75, 1), [0])
    a.set_reference(imgs[0])
    sa = a.sweep(grid.params)
    assert a.last_first_pass == "fused"
    b = H.SweepEngine(n)
    b.set_geometry(apix=apix, helical_diameter=d, ball_radius=br)
    b.set_reference(imgs[0])
    sb = b.sweep(grid.params)
    assert b.last_first_pass == "fused"
    np.testing.assert_array_equal(sa, sb)
    a.close()
    np.testing.assert_array_equal(sb, b.sweep(grid.params))
    b.close()


def test_failed_set_reference_leaves_no_reference():
    """A set_reference that fails (empty mask) must not leave the old sizes behind: the next sweep reports the
    missing reference instead of launching with stale tables."""
    n, apix = 64, 2.0
    img = np.random.default_rng(0).normal(size=(n, n)).astype(np.float32)
    with H.SweepEngine(n) as eng:
        eng.set_geometry(apix=apix, helical_diameter=0.4 * n * apix, ball_radius=2 * apix)
        eng.set_reference(img)
        ok = eng.sweep(np.array([[29.0, 10.0, 1, 0.0]]))
        with pytest.raises(ValueError):
            eng.set_reference(img, mask=np.zeros((n, n), bool))
        # the empty mask is rejected before the old tables are touched: the old reference still answers
        np.testing.assert_array_equal(ok, eng.sweep(np.array([[29.0, 10.0, 1, 0.0]])))
        eng.set_reference(np.stack([img, img[::-1].copy()]))          # S = 2 after S = 1: buffers regrow
        two = eng.sweep(np.array([[29.0, 10.0, 1, 0.0]]))
        assert two.shape == (2, 1) and two[0, 0] == pytest.approx(ok[0, 0], abs=2e-6)


def test_device_argmax_matches_numpy_rule():
    import torch

    rng = np.random.default_rng(5)
    sc = rng.normal(size=(5, 100_003)).astype(np.float32)
    sc[0, 17] = sc[0, 90_000] = 9.0          # tie: lowest index wins
    sc[1, :] = np.nan                         # all NaN -> 0
    sc[2, 5] = np.nan
    sc[2, 6] = 8.0
    sc[3, 0] = np.inf
    sc[4, -1] = 7.5
    t = torch.from_numpy(sc).cuda()
    with H.SweepEngine(64) as eng:
        got = eng.argmax_device(t.data_ptr(), 5, sc.shape[1])
    assert got.tolist() == [17, 0, 6, 0, sc.shape[1] - 1]


# ---------------------------------------------------------------------------- the task function under a pool
def _task_args(data, twist, rise, *, n=None, image_file=None, image_index=0, algorithm=None, target_apix2d=5.0,
               low_pass=0, thresh_fraction=-1, transpose=0):
    """The 36-tuple of app.py:2407-2446 (apix 5 A/pixel, the app's default target_apix2d = target_apix3d = 5)."""
    n = data.shape[0] if n is None else n
    apix = 5.0
    return (0, 1, data, image_file, image_index, twist, rise, (rise, rise), 1, 0.0, (0, 0), 0.0, (0, 0), 0.0, (0, 0),
            apix, "", low_pass, transpose, 0, 5.0, target_apix2d, thresh_fraction, -1, n * apix, 0.4 * n * apix, 0,
            -1, -1, "linear", 0, 0, "cosine", algorithm or {}, 0, 1)


def test_process_one_task_from_a_thread_pool_like_the_app():
    """app.py:2473-2476 submits one task per (twist, rise) pair to a ThreadPoolExecutor, all with the same image and
    the app's default target_apix2d = 5 (here = the image's own pixel size, pipeline.py:268-272: no rescale).
    200 tasks return the sweep's scores, in any completion order, at > 2,000 calls/s after the first call."""
    n, apix = 256, 5.0
    d, br = 0.4 * n * apix, 2 * apix
    clean = O.simulate_helical_projection(1, 29.0, 25.0, 1, d, br, 0, 0, n, n, apix)
    img = (clean + np.random.default_rng(0).normal(0, 0.5 * clean.std(), clean.shape)).astype(np.float32)
    twists = 29.0 + 0.2 * np.arange(-10, 10)
    rises = 25.0 + 0.25 * np.arange(-5, 5)
    grid = build_grid(twists, rises, (1,), tube_length=n * apix)
    algo = dict(helical_diameter=d, ball_radius=br)
    ref = H.sweep(img, twists, rises, (1,), apix=apix, helical_diameter=d, ball_radius=br).scores.reshape(-1)
    H.process_one_task(*_task_args(img, 29.0, 25.0, algorithm=algo))           # first call prepares the reference
    tasks = [_task_args(img, float(tw), float(rs), algorithm=algo) for tw, rs, _, _ in grid.params]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=8) as pool:
        futs = [pool.submit(H.process_one_task, *t) for t in tasks]
        res = [f.result() for f in futs]
    dt = time.perf_counter() - t0
    got = np.array([r[0] for r in res])
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)
    for r, (tw, rs, _, _) in zip(res, grid.params):
        assert r[2][5] == tw and r[2][6] == rs and r[2][4] == apix and r[1][:4] == (None,) * 4
    rate = len(tasks) / dt
    assert rate > 2000, f"{rate:.0f} calls/s"
    # two threads with different images must not see each other's reference (engine session lock)
    other = img[::-1].copy()
    ref_o = H.sweep(other, twists[:4], rises[:4], (1,), apix=apix, helical_diameter=d, ball_radius=br).scores.reshape(-1)
    mixed = []
    for k, (tw, rs) in enumerate([(t, r) for t in twists[:4] for r in rises[:4]]):
        mixed.append((_task_args(img, float(tw), float(rs), algorithm=algo), ref[(k // 4) * len(rises) + k % 4]))
        mixed.append((_task_args(other, float(tw), float(rs), algorithm=algo), ref_o[k]))
    with ThreadPoolExecutor(max_workers=8) as pool:
        outs = list(pool.map(lambda a: H.process_one_task(*a[0])[0], mixed))
    np.testing.assert_allclose(outs, [m[1] for m in mixed], rtol=0, atol=2e-6)


def test_process_one_task_reads_one_based_image_index(tmp_path):
    """pipeline.py:212 reads slice imageIndex - 1 of the stack."""
    from helicon_amd.mrc import write_mrc

    n, apix = 64, 5.0
    rng = np.random.default_rng(1)
    stack = rng.normal(size=(3, n, n)).astype(np.float32)
    path = tmp_path / "stack.mrcs"
    write_mrc(path, stack, apix)
    algo = dict(helical_diameter=0.4 * n * apix, ball_radius=2 * apix)
    for k in (1, 2, 3):
        out = H.process_one_task(*_task_args(None, 29.0, 25.0, n=n, image_file=str(path), image_index=k, algorithm=algo))
        np.testing.assert_array_equal(out[2][0], stack[k - 1])
        assert out[2][2] == k
    with pytest.raises((OSError, IndexError, ValueError)):
        H.process_one_task(*_task_args(None, 29.0, 25.0, n=n, image_file=str(path), image_index=4, algorithm=algo))


# ---------------------------------------------------------------------------- filtered spectrum
@pytest.mark.parametrize("n", [64, 256])
@pytest.mark.parametrize("lp,hp", [(0.3, 0.0), (0.0, 0.05), (0.4, 0.02)])
def test_compute_power_spectra_with_low_and_high_pass(n, lp, hp):
    """transforms.py:811-817: low_high_pass_filter on log1p|F|, then min-max normalisation."""
    img = np.random.default_rng(n).normal(size=(n, n)).astype(np.float32)
    got, _ = H.compute_power_spectra(img, 2.0, low_pass_fraction=lp, high_pass_fraction=hp)
    ref, _ = O.compute_power_spectra(img.astype(np.float64), 2.0, low_pass_fraction=lp, high_pass_fraction=hp)
    np.testing.assert_allclose(got, ref, rtol=0, atol=5e-5)


# ---------------------------------------------------------------------------- the collective through the C ABI
def test_c_abi_allgather_single_rank_rccl():
    """hh_comm_unique_id / hh_comm_init / hh_allgather / hh_argmax_device: the sweep -> all-gather -> arg-max step
    of a multi-GPU run with no host framework in the data path (world size 1 here; torch only allocates)."""
    import torch

    n, apix = 64, 2.0
    d, br = 0.4 * n * apix, 2 * apix
    clean = O.simulate_helical_projection(1, 29.0, 10.0, 1, d, br, 0, 0, n, n, apix)
    imgs = np.stack([(clean + np.random.default_rng(s).normal(0, 0.5 * clean.std(), clean.shape)).astype(np.float32)
                     for s in range(2)])
    grid = build_grid(np.arange(25.0, 33.5, 1.0), np.arange(8.0, 12.5, 0.5), (1,), tube_length=n * apix)
    g, per = len(grid), len(grid) + 7                                   # a padded send buffer, like a short shard's
    with H.SweepEngine(n) as eng:
        eng.set_geometry(apix=apix, helical_diameter=d, ball_radius=br)
        eng.set_reference(imgs)
        ref = eng.sweep(grid.params)
        eng.comm_init(0, 1, H.SweepEngine.comm_unique_id())
        dp = torch.from_numpy(grid.params).cuda()
        send = torch.full((2, per), float("nan"), dtype=torch.float32, device="cuda")
        recv = torch.zeros((1, 2, per), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        eng.sweep_device(dp.data_ptr(), g, send.data_ptr(), host_params=grid.params, ld_scores=per)
        eng.allgather(send.data_ptr(), 2 * per, recv.data_ptr())
        best = eng.argmax_device(recv.data_ptr(), 2, per, per)
        eng.synchronize()
        got = recv.cpu().numpy()[0]
        eng.comm_destroy()
    np.testing.assert_array_equal(got[:, :g], ref)
    assert np.isnan(got[:, g:]).all()
    assert best.tolist() == np.argmax(ref, axis=1).tolist()


# ---------------------------------------------------------------------------- image preparation: rotate / shift, transpose
def test_rotate_shift_image_reproduces_the_reference(golden_dir):
    """helicon.rotate_shift_image (lib/transforms.py:315-369; scipy affine_transform order 1, constant) on the device
    against the reference's outputs (fixture G8): the same pixels fall outside the input, values within 1e-6."""
    g = np.load(golden_dir / "g8_rotate_shift.npz")
    for k in range(6):
        a = g[f"case{k}_args"]
        rc = None if np.isnan(a[5]) else np.array((a[5], a[6]))
        got = H.rotate_shift_image(g[f"case{k}_image"], a[0], (a[1], a[2]), (a[3], a[4]), rc)
        want = g[f"case{k}_out"]
        assert got.shape == want.shape and got.dtype == want.dtype
        np.testing.assert_array_equal(got == 0, want == 0)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)
    big = np.random.default_rng(3).normal(size=(700, 900)).astype(np.float32)
    np.testing.assert_allclose(H.rotate_shift_image(big, 17.0, (2.5, -1.0), (0.0, 4.0)),
                               O.rotate_shift_image(big, 17.0, (2.5, -1.0), (0.0, 4.0)), rtol=0, atol=2e-6)
    assert H.rotate_shift_image(big, 0, [0, 0], [0, 0]) is not big      # the reference's early return: a copy (data * 1.0)
    with pytest.raises(NotImplementedError):
        H.rotate_shift_image(big, 5.0, order=2)
    for k in range(6):
        assert H.is_vertical(g[f"vert{k}_image"]) == bool(g[f"vert{k}"][0])


def test_process_one_task_transposes_a_vertical_image():
    """pipeline.py:202-203: transpose < 0 transposes when utils.is_vertical says so — a vertical helix scores like its
    transposed (horizontal) self, and a horizontal one is left alone."""
    n, apix = 128, 5.0
    d, br = 0.4 * n * apix, 2 * apix
    img = O.simulate_helical_projection(1, 29.0, 25.0, 1, d, br, 0, 0, n, n, apix).astype(np.float32)
    algo = dict(helical_diameter=d, ball_radius=br)
    assert not H.is_vertical(img) and H.is_vertical(img.T)
    s_h = H.process_one_task(*_task_args(img, 29.0, 25.0, algorithm=algo))[0]
    s_auto = H.process_one_task(*_task_args(np.ascontiguousarray(img.T), 29.0, 25.0, algorithm=algo, transpose=-1))[0]
    s_same = H.process_one_task(*_task_args(img, 29.0, 25.0, algorithm=algo, transpose=-1))[0]
    s_forced = H.process_one_task(*_task_args(np.ascontiguousarray(img.T), 29.0, 25.0, algorithm=algo, transpose=1))[0]
    assert s_auto == pytest.approx(s_h, abs=1e-6) and s_same == pytest.approx(s_h, abs=1e-6) and s_forced == pytest.approx(s_h, abs=1e-6)
    assert s_h > 0.9


# ---------------------------------------------------------------------------- launch schedule of the fused pass
@pytest.mark.parametrize("n,n_twists,n_rises", [(128, 37, 93), (64, 300, 40), (256, 9, 701), (512, 50, 250), (128, 1, 3000)])
def test_fused_launch_schedules_agree_with_the_transform_pipeline(n, n_twists, n_rises):
    """The fused pass cuts a launch's runs into workgroups by the device's resident-workgroup count (long workgroups for
    whole rounds, finer pieces for the last round; one rank's share of a strong-scaling run is the (512, 50, 250) shape):
    whatever the cut, every candidate is scored once and like the general pipeline scores it."""
    apix = 2.0
    eng = H.SweepEngine(n)
    eng.set_geometry(apix=apix, helical_diameter=0.45 * n * apix, ball_radius=2.5 * apix)
    rng = np.random.default_rng(n + n_twists)
    twists = np.round(np.sort(rng.uniform(-60, 60, n_twists)), 3)
    rises = 6.0 * apix * (1.0 + 2e-4 * np.arange(n_rises))
    img = eng.simulate(float(twists[n_twists // 2]), float(rises[n_rises // 2]), 1)
    eng.set_reference((img + rng.normal(0, 0.3 * img.std(), img.shape)).astype(np.float32)[None], H.radial_band_mask(n, n))
    params = np.array([[tw, rs, 1, 0.0] for tw in twists for rs in rises])
    eng.set_table_path(2)
    got = eng.sweep(params)
    assert eng.last_first_pass == "fused"
    eng.set_table_path(0)
    step = max(1, len(params) // 4000)          # the general pipeline on a strided sample (and on the grid's two ends)
    pick = np.unique(np.r_[0:min(300, len(params)), np.arange(0, len(params), step), len(params) - min(300, len(params)):len(params)])
    want = eng.sweep(params[pick])
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got[:, pick], want, rtol=0, atol=5e-5)


def test_memory_report_and_segment_buffers_follow_the_sweep():
    """hh_memory_bytes: a context reports what it holds; the several-segment buffers are sized by the sweeps that ran, not
    by the largest batch any sweep could use (round 2: 12.9 GB of masked spectra for a 2-segment, 1-candidate call)."""
    eng = H.SweepEngine(512)
    eng.set_geometry(apix=1.0, helical_diameter=0.4 * 512, ball_radius=2.0)
    clean = eng.simulate(1.2, 4.75, 1)
    m0 = eng.memory_bytes()
    assert m0["total"] == sum(v for k, v in m0.items() if k != "total") and m0["segment_buffers"] == 0
    eng.set_reference(np.stack([clean, clean[::-1].copy()]))
    one = eng.sweep(np.array([[1.2, 4.75, 1, 0.0]]))
    assert one.shape == (2, 1) and one[0, 0] > 0.99
    m1 = eng.memory_bytes()
    assert 0 < m1["segment_buffers"] < 2**30, m1
    g = H.build_grid(H.sweep_axis(1.0, 1.4, 0.02), H.sweep_axis(4.5, 5.0, 0.005), (1,), tube_length=512.0)   # 21 x 101 runs
    sc = eng.sweep(g.params)
    m2 = eng.memory_bytes()
    assert eng.last_first_pass == "fused" and m2["segment_buffers"] > m1["segment_buffers"] and m2["run_tables"] > 0
    assert int(np.argmax(sc[0])) == int(np.argmin(np.abs(g.params[:, 0] - 1.2) + np.abs(g.params[:, 1] - 4.75)))
    np.testing.assert_allclose(eng.sweep(np.array([[1.2, 4.75, 1, 0.0]])), one, rtol=0, atol=1e-6)   # the grown buffers serve small calls
    eng.close()


def test_power_spectrum_with_fourier_zoom_against_the_oracle():
    """compute_power_spectra(cutoff_res=, output_size=) (transforms.py:771-820 over fft_rescale :663-713): the device's
    direct non-uniform transform against the oracle's (the sum finufft approximates to 1e-6) — zoom in, zoom out,
    rectangular and odd shapes, both log settings, with the Gaussian filter on top; and the default arguments still take
    the sweep's own transform and agree with the zoom kernels evaluated at the same frequencies."""
    rng = np.random.default_rng(11)
    eng = H.SweepEngine(64)
    eng.set_geometry(apix=2.0, helical_diameter=50.0, ball_radius=4.0)
    helix = eng.simulate(29.0, 10.0, 1)
    eng.close()
    cases = [
        (helix, 2.0, (8.0, 8.0), (64, 64), True),       # zoom in by 2, same size
        (helix, 2.0, (6.0, 10.0), (96, 48), True),      # anisotropic, other size
        (helix, 2.0, None, (128, 128), False),          # finer sampling only
        (rng.normal(size=(45, 63)), 1.5, (5.0, 7.0), (50, 33), True),    # odd, rectangular in and out
        (rng.normal(size=(40, 72)), 1.0, (2.0, 2.0), (40, 72), True),    # the default sampling through the zoom kernels
    ]
    for img, apix, cut, osz, log in cases:
        if cut == (2.0 * apix, 2.0 * apix) and tuple(osz) == img.shape:   # the wrapper routes these to the sweep's transform
            pw, ph = _zoom_direct(img, apix, cut, osz, log)
            pw_o, ph_o = _oracle_direct(img, apix, cut, osz, log)
        else:
            pw, ph = H.compute_power_spectra(img, apix, cutoff_res=cut, output_size=osz, log=log)
            pw_o, ph_o = O.compute_power_spectra(img.astype(np.float32), apix, cutoff_res=cut, output_size=osz, log=log)
        assert pw.shape == tuple(osz)
        np.testing.assert_allclose(pw, pw_o, rtol=0, atol=2e-5)
        strong = pw_o > 0.2                                  # the phase of a near-zero coefficient is noise
        d = np.angle(np.exp(1j * (ph - ph_o)))
        assert np.abs(d[strong]).max() < 1e-3
    pw, _ = H.compute_power_spectra(helix, 2.0, cutoff_res=(8.0, 8.0), output_size=(64, 64), low_pass_fraction=0.5, high_pass_fraction=0.05)
    pw_o, _ = O.compute_power_spectra(helix.astype(np.float32), 2.0, cutoff_res=(8.0, 8.0), output_size=(64, 64), low_pass_fraction=0.5,
                                      high_pass_fraction=0.05)
    np.testing.assert_allclose(pw, pw_o, rtol=0, atol=5e-5)


def _zoom_direct(img, apix, cut, osz, log):
    """The zoom entry point at the DEFAULT frequencies (the Python wrapper would route those to the sweep's transform)."""
    import ctypes as C

    from helicon_amd import _lib

    a = np.ascontiguousarray(img, dtype=np.float32)
    pw = np.empty(osz, dtype=np.float32)
    ph = np.empty(osz, dtype=np.float32)
    f32 = C.POINTER(C.c_float)
    _lib.check(_lib.lib().hh_power_spectrum_zoom(0, a.ctypes.data_as(f32), a.shape[0], a.shape[1], osz[0], osz[1], apix, cut[0], cut[1],
                                                 1 if log else 0, pw.ctypes.data_as(f32), ph.ctypes.data_as(f32)), None)
    return pw.astype(np.float64), ph.astype(np.float64)


def _oracle_direct(img, apix, cut, osz, log):
    fft = np.fft.fftshift(O.fft_rescale(img.astype(np.float32), apix=apix, cutoff_res=cut, output_size=osz))
    pwr = O.normalize_percentile(np.log1p(np.abs(fft)) if log else np.abs(fft), (0, 100))
    return pwr, np.angle(fft)
