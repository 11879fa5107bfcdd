"""Host-side logic (no GPU): grid construction, masks, sharding, and the library's exports."""
import re
import sys
from pathlib import Path

import numpy as np
import pytest

import helicon_amd as H
from helicon_amd import _lib
from oracle import path_b as O

ROOT = Path(__file__).resolve().parent.parent


def test_grid_matches_oracle_driver():
    tw = O.sweep_axis(0.01, 4.00, 0.01)
    rs = O.sweep_axis(4.000, 5.245, 0.005)
    np.testing.assert_array_equal(H.sweep_axis(0.01, 4.00, 0.01), tw)
    np.testing.assert_array_equal(H.sweep_axis(4.000, 5.245, 0.005), rs)
    g = H.build_grid(tw, rs, (1, 3), tube_length=512.0)
    p, valid = O.build_candidates(tw, rs, (1, 3), tube_length=512.0)
    assert g.shape == (2, 400, 250) and len(g) == 200000
    np.testing.assert_array_equal(g.params[:, :3], p)
    np.testing.assert_array_equal(g.valid, valid)
    assert g.unravel(250 * 400 + 251) == (1, 1, 1)


def test_grid_filters_and_wrap():
    g = H.build_grid([0.001, 1.0, 190.0], [0.001, 4.75, 40.0], (1, 2), tube_length=64.0)
    p, valid = O.build_candidates([0.001, 1.0, 190.0], [0.001, 4.75, 40.0], (1, 2), tube_length=64.0)
    np.testing.assert_array_equal(g.params[:, :3], p)
    np.testing.assert_array_equal(g.valid, valid)
    assert g.params[6, 0] == -170.0
    for v in (-540.0, -181.0, -180.0, 0.0, 180.0, 180.5, 725.0):
        assert H.set_to_periodic_range(v) == O.set_to_periodic_range(v)
    with pytest.raises(ValueError):
        H.build_grid([1.0], [1.0], (0,), tube_length=10)


def test_masks_match_oracle():
    for n in (32, 64, 512):
        np.testing.assert_array_equal(H.radial_band_mask(n, n), O.radial_band_mask(n, n))
    np.testing.assert_array_equal(H.layer_line_mask(64, 64, axial_bins=[3, 9], half_width=1),
                                  O.layer_line_mask(64, 64, axial_bins=[3, 9], half_width=1))
    m = H.radial_band_mask(64, 64)
    assert not m[32, 32] and not m[32, 34] and m[32, 35] and not m[0].any() and not m[:, 0].any()


def test_shard_bounds_cover_exactly_once():
    for n, w in ((100000, 8), (600000, 8), (7, 8), (0, 4), (13, 2), (1000, 1)):
        seen = np.zeros(n, dtype=int)
        per = None
        for r in range(w):
            lo, hi, per = H.shard_bounds(n, r, w)
            assert 0 <= lo <= hi <= n and hi - lo <= per
            seen[lo:hi] += 1
        assert (seen == 1).all()
        assert per * w >= n
    # run-aligned shards (what sweep_distributed uses): every shard starts on a twist, still an exact cover
    for n, w, align in ((100000, 8, 250), (100000, 3, 250), (600000, 7, 250), (4000, 8, 100), (90, 4, 30)):
        seen = np.zeros(n, dtype=int)
        for r in range(w):
            lo, hi, per = H.shard_bounds(n, r, w, align)
            assert lo % align == 0 and per % align == 0 and 0 <= lo <= hi <= n
            seen[lo:hi] += 1
        assert (seen == 1).all()


def test_header_symbols_all_exported_and_bound():
    hdr = (ROOT / "include" / "helicon_hip.h").read_text()
    declared = set(re.findall(r"^\s*(?:int64_t|int|void|const char\*)\s+(hh_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 20
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    L = _lib.lib()  # raises if the .so is missing or lacks a symbol; no GPU call is made
    for name in declared:
        assert hasattr(L, name)
    assert L.hh_abi_version() == 1
    assert L.hh_algorithmic_bytes(512) == 3153920
    assert L.hh_algorithmic_bytes(256) == 790528
    assert L.hh_algorithmic_bytes(1024) == 12599296


def test_no_cxx_exception_crosses_the_c_abi():
    """SURVEY 8(b) "never exit()": the host code under the entry points uses std::vector / std::thread, and round 3 saw a
    Python process aborted inside hh_pab_create (std::terminate).  Every entry point is now a function-try-block; the
    self-test raises the exceptions that code can meet INSIDE a guarded entry point and must come back as a status and a
    message — an absurd element count, an allocation no machine can serve, a thread that cannot start, a foreign type."""
    L = _lib.lib()
    want = {0: (-4, b"length_error"), 1: (-4, b"bad_alloc"), 2: (-5, b"Resource temporarily unavailable"), 3: (-5, b"unknown C++ exception")}
    for kind, (code, text) in want.items():
        assert L.hh_selftest_exception(kind) == code
        msg = L.hh_last_error(None)
        assert msg.startswith(b"hh_selftest_exception: ") and text in msg, msg
    assert L.hh_selftest_exception(99) == 0
    # and the source agrees: no entry point with a body of its own is left outside a try block
    import re
    csrc = ROOT / "helicon_amd" / "csrc"
    for f in ("helicon_hip.hip", "fourier_zoom.inc", "path_a_batch.inc", "path_a_host.inc", "image_prep.inc"):
        text_ = (csrc / f).read_text()
        for m in re.finditer(r'^(?:extern "C" )?(?:int|int64_t|void) (hh_\w+)\([^;{]*\)\s*(try)?\s*\{(.*)$', text_, re.M):
            one_liner = m.group(3).rstrip().endswith("}")
            assert m.group(2) == "try" or one_liner, (f, m.group(1))


def test_argmax_rule_is_lowest_index_and_ignores_nan():
    import ctypes as C
    L = _lib.lib()
    s = np.array([np.nan, 0.5, 0.7, 0.7, -np.inf, np.nan], dtype=np.float32)
    idx = C.c_int64(-1)
    assert L.hh_argmax(s.ctypes.data_as(C.POINTER(C.c_float)), s.size, C.byref(idx)) == 0
    assert idx.value == 2


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(H.HeliconHipError):
        H.SweepEngine(64)
    with pytest.raises(H.HeliconHipError):
        H.SweepEngine((48, 96))            # a general size: valid, but there is still no CPU path
    with pytest.raises(ValueError):
        H.SweepEngine(4)


def test_image_preparation_entry_points_check_their_arguments_and_have_no_cpu_path():
    """The four entry points of csrc/image_prep.inc: bad arguments are refused before anything touches a device (HH_ERR_ARG,
    a message), and without a GPU a valid call is a loud error — there is no CPU fallback; the Python mirrors refuse what they
    do not provide."""
    import ctypes as C
    import torch
    from helicon_amd import denovo3D as D

    L = _lib.lib()
    img = np.ones((8, 8), dtype=np.float32)
    out = np.empty_like(img)
    eye = (C.c_double * 9)(1, 0, 0, 0, 1, 0, 0, 0, 1)
    m8 = (C.c_double * 8)()
    assert L.hh_warp_affine_2d(0, img.ctypes.data, 0, 8, 8, eye, 3, 0.0, 1, out.ctypes.data) == -1          # order 3: not provided
    assert b"hh_warp_affine_2d" in L.hh_last_error(None)
    assert L.hh_warp_affine_2d(0, None, 0, 8, 8, eye, 1, 0.0, 1, out.ctypes.data) == -1
    assert L.hh_rescale_2d(0, img.ctypes.data, 0, 8, 8, 0, 4, 3, 1, 1, out.ctypes.data) == -1              # empty output
    assert L.hh_rescale_2d(0, img.ctypes.data, 0, 8, 8, 4, 4, 2, 1, 1, out.ctypes.data) == -1              # order 2
    assert L.hh_helix_moments(0, img.ctypes.data, 0, 0, 8, 0.0, m8) == -1
    assert L.hh_affine_transform_2d_cubic(0, None, 8, 8, (C.c_double * 4)(1, 0, 0, 1), (C.c_double * 2)(), None) == -1
    with pytest.raises(NotImplementedError):
        D.transform_image(img, order=3)
    with pytest.raises(NotImplementedError):
        D.rescale(img, 0.5, order=2)
    with pytest.raises(TypeError):
        D.down_scale(np.ones((8, 8), dtype=np.int16), 2.0, 1.0)
    assert D.down_scale(img, 1.0, 1.0) is img                                                               # nothing to do: no device needed
    np.testing.assert_array_equal(D.pad_to_size(img, (10, 9)).shape, (10, 9))
    if not torch.cuda.is_available():
        for call in (lambda: D.transform_image(img, rotation=5.0), lambda: D.rescale(img, 0.5), lambda: D.estimate_helix_rotation_center_diameter(img),
                     lambda: D.rotate_shift_image(img, 5.0, order=3)):
            with pytest.raises(H.HeliconHipError, match="no such HIP device"):
                call()


def test_product_never_imports_oracle():
    for f in (ROOT / "helicon_amd").rglob("*.py"):
        src = f.read_text()
        assert "import oracle" not in src and "from oracle" not in src, f


def test_batch_driver_arguments():
    import argparse

    from helicon_amd import denovo3DBatch as B

    a = B.add_args(argparse.ArgumentParser()).parse_args(
        ["x.npy", "--apix", "2", "--twist", "25", "33", "0.2", "--rise", "8", "13", "0.2", "--csym", "1", "3"])
    assert a.csym == [1, 3] and a.twist == [25.0, 33.0, 0.2] and not a.no_log and a.top == 10
    assert len(H.sweep_axis(*a.twist)) == 41 and len(H.sweep_axis(*a.rise)) == 26


def test_mrc_round_trip_and_reader_errors(tmp_path):
    from helicon_amd import mrc

    rng = np.random.default_rng(0)
    stack = rng.normal(size=(3, 8, 12)).astype(np.float32)
    mrc.write_mrc(tmp_path / "s.mrcs", stack, apix=1.25)
    data, apix = mrc.read_mrc(tmp_path / "s.mrcs")
    assert data.shape == (3, 8, 12) and apix == pytest.approx(1.25)
    np.testing.assert_array_equal(np.asarray(data), stack)
    assert mrc.image_shape(tmp_path / "s.mrcs") == (12, 8, 3)
    np.testing.assert_array_equal(mrc.read_image_2d(tmp_path / "s.mrcs", 2), stack[2])
    with pytest.raises(OSError):
        mrc.read_image_2d(tmp_path / "s.mrcs", 3)
    with pytest.raises(OSError):
        mrc.read_image_2d(tmp_path / "missing.mrc", 0)
    # big-endian int16 file with an extended header
    import struct
    hdr = bytearray(1024)
    struct.pack_into(">4i", hdr, 0, 4, 2, 1, 1)
    struct.pack_into(">3i", hdr, 28, 4, 2, 1)
    struct.pack_into(">3f", hdr, 40, 8.0, 4.0, 2.0)
    struct.pack_into(">i", hdr, 92, 16)
    hdr[212:216] = b"\x11\x11\x00\x00"
    vals = np.arange(8, dtype=">i2")
    (tmp_path / "b.mrc").write_bytes(bytes(hdr) + b"\0" * 16 + vals.tobytes())
    data, apix = mrc.read_mrc(tmp_path / "b.mrc")
    assert apix == pytest.approx(2.0)
    np.testing.assert_array_equal(np.asarray(data), np.arange(8).reshape(1, 2, 4))


def _fracs(obj, path=""):
    """Every value stored under a key named `frac` (or ending in `_frac`) anywhere in a JSON-like object."""
    if isinstance(obj, dict):
        for k, v in obj.items():
            if (k == "frac" or k.endswith("_frac")) and isinstance(v, (int, float)):
                yield path + "/" + k, v
            else:
                yield from _fracs(v, path + "/" + k)


def _bench_module():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_bench_roofline_object_has_the_contract_keys_and_no_fraction_above_one():
    """bench.py's roofline() on canned profile numbers (no GPU): the keys the driver and the judge read, and no
    `frac` that is not a fraction of a physical peak (the round-1 line said "hbm", frac 1.6 for the fused pass)."""
    bench = _bench_module()
    b_alg = 4 * 512 * 512 + 16 * 512 * 257
    prof = dict(ms_first_pass=12.0, n_first_pass=250, ms_second_pass=14.5, n_second_pass=250, ms_finalize=0.01,
                n_finalize=1, ms_centres=0.2, n_centres=10, candidates=62500, candidates_total=1000000)
    for pipeline in ("transform", "run_tables"):
        r = bench.roofline(prof, 512, b_alg, pipeline)
        assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernels"} <= set(r)
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0 < r["frac"] < 1
        assert r["moved_bytes_per_candidate"] == 16 * 512 * 257           # no phantom 4 N^2
        assert r["hbm_model"]["alg_bytes_per_candidate"] == b_alg
        assert len(r["kernels"]) == 2 and all("avg_us" in k for k in r["kernels"].values())
        assert all(0 <= v <= 1 for _, v in _fracs(r))
    # the round-1 measurement: 100k candidates in 24 ms of k_fused_pass
    fused = dict(prof, ms_first_pass=1.3, n_first_pass=10, ms_second_pass=240.0, n_second_pass=40, candidates=1000000)
    r = bench.roofline(fused, 512, b_alg, "fused")
    assert r["pipeline"] == "fused" and "k_fused_pass" in r["kernels"]
    assert r["bound"] == "fp32_vector" and r["unit"] == "TFLOP/s" and r["peak"] == pytest.approx(157.3)
    assert 0.25 < r["frac"] < 0.4 and r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    assert r["hbm_model"]["ratio_to_hbm_peak"] > 1.0 and "not a bound" in r["hbm_model"]["note"]
    fr = dict(_fracs(r))
    assert fr and all(0 <= v <= 1 for v in fr.values()), fr
    if r.get("hbm_measured"):
        assert r["hbm_measured"]["hbm_measured_frac"] < 0.2 and r["traffic_source"]
    assert r["traffic"] is None or r["traffic"] < 0.1 * b_alg * r["candidates_per_launch"]


class _StubEngine:
    """Stands in for SweepEngine in host-logic tests: records what the task function asks of the device."""

    def __init__(self):
        import contextlib
        import threading

        self.lock = threading.RLock()
        self.ref_calls, self.refs, self.sweeps = 0, [], 0
        self._ref_key = None
        self._ctx = contextlib.nullcontext

    def session(self):
        return self.lock

    def set_geometry(self, **kw):
        self.geom = kw

    def set_reference(self, images, mask=None, log=True, key=None):
        if key is not None and key == self._ref_key:
            return
        self.ref_calls += 1
        self.refs.append(np.array(images, copy=True))
        self._ref_key = key

    def sweep(self, params):
        self.sweeps += 1
        return np.full((1, len(params)), 0.25, dtype=np.float32)


def _task(data, twist, rise, n=16, image_file=None, image_index=0, target_apix2d=5.0):
    apix = 5.0
    return (0, 1, data, image_file, image_index, twist, rise, (rise, rise), 1, 0.0, (0, 0), 0.0, (0, 0), 0.0, (0, 0),
            apix, "", 0, 0, 0, 5.0, target_apix2d, -1, -1, n * apix, 0.4 * n * apix, 0, -1, -1, "linear", 0, 0,
            "cosine", {}, 0, 1)


def test_process_one_task_host_logic_with_a_stub_engine(tmp_path, monkeypatch):
    """pipeline.py:211-218, 268-272 without a GPU: 1-based imageIndex, blank image -> None, the app's default
    target_apix2d (5) accepted when it does not exceed the image's pixel size, one reference upload for many calls
    on the same image, and loud refusal of the options that need scikit-image."""
    from helicon_amd import denovo3D as D
    from helicon_amd.mrc import write_mrc

    stub = _StubEngine()
    monkeypatch.setattr(D, "_engine", lambda side, device=0: stub)
    monkeypatch.setattr(D, "_SUPPORTED_N", (16, 32))
    rng = np.random.default_rng(2)
    stack = rng.normal(size=(3, 16, 16)).astype(np.float32)
    write_mrc(tmp_path / "s.mrcs", stack, 5.0)
    for k in (1, 2, 3):
        out = D.process_one_task(*_task(None, 29.0, 25.0, image_file=str(tmp_path / "s.mrcs"), image_index=k))
        np.testing.assert_array_equal(out[2][0], stack[k - 1])
        np.testing.assert_array_equal(stub.refs[-1][-16:], stack[k - 1])
        assert out[0] == 0.25 and out[2][2] == k and out[2][4] == 5.0
    with pytest.raises(OSError):
        D.process_one_task(*_task(None, 29.0, 25.0, image_file=str(tmp_path / "s.mrcs"), image_index=4))
    assert D.process_one_task(*_task(np.ones((16, 16)), 29.0, 25.0)) is None       # blank
    calls = stub.ref_calls
    img = stack[0].copy()
    for tw in (28.0, 29.0, 30.0):
        D.process_one_task(*_task(img, tw, 25.0))
    assert stub.ref_calls == calls + 1          # stack[0] was last seen two references ago: one new upload, then cached
    D.process_one_task(*_task(img, 29.0, 25.0, target_apix2d=2.0))                  # finer than the image: no rescale
    assert stub.ref_calls == calls + 1
    # coarser than the image: down_scale runs on the device (hh_rescale_2d) — without one that is a loud error, not a fallback
    with pytest.raises(H.HeliconHipError, match="no such HIP device"):
        D.process_one_task(*_task(img, 29.0, 25.0, target_apix2d=10.0))
    with monkeypatch.context() as mp:
        mp.setattr(D, "down_scale", lambda d, target, orig, device=0: np.asarray(d)[::2, ::2] * 1.0)
        out = D.process_one_task(*_task(img, 29.0, 25.0, target_apix2d=10.0))
        assert out[2][0].shape == (8, 8) and out[2][4] == 10.0                      # the prepared image and ITS pixel size
    t = list(_task(img, 29.0, 25.0))
    t[16] = "tv"
    with pytest.raises(NotImplementedError):
        D.process_one_task(*t)
    before = img.copy()
    t = list(_task(img, 29.0, 25.0))
    t[22] = 0.1                                                                     # thresh_fraction
    monkeypatch.setattr(D, "threshold_data", lambda d, thresh_fraction=None, device=0: np.clip(d, d.max() * thresh_fraction, None) - d.max() * thresh_fraction)
    D.process_one_task(*t)
    np.testing.assert_array_equal(img, before)   # the caller's array is not modified


def test_no_process_wide_kernel_attribute_flags():
    """hipFuncAttributeMaxDynamicSharedMemorySize is per device: the library tracks it per context."""
    src = (ROOT / "helicon_amd" / "csrc" / "helicon_hip.hip").read_text()
    assert "static bool attr_done" not in src and "ensure_lds_attr" in src


def test_fused_launch_schedule_covers_every_candidate_once():
    """hh_fused_schedule (pure host arithmetic, loads without a GPU): whatever the shape, the plan's layers cover each run's
    candidates exactly once, workgroups hold at least 16 candidates unless the run is shorter, and the cases DESIGN.md
    quotes come out as quoted (C2 in one launch: whole runs, 25 rounds; an eighth of it: 48 whole runs + 2 runs in 8)."""
    import ctypes as C

    from helicon_amd import _lib

    L = _lib.lib()

    def plan(runs, run_len, n_kb, slots):
        out = (C.c_int32 * 6)()
        assert L.hh_fused_schedule(runs, run_len, n_kb, slots, out) == 0
        return tuple(out)

    assert plan(400, 250, 32, 512) == (400, 1, 250, 1, 250, 400)
    ra, ga, ca, gb, cb, layers = plan(50, 250, 32, 512)
    assert (ra, ga, ca) == (48, 1, 250) and gb == 8 and cb == 32 and layers == 48 + 2 * 8
    rng = np.random.default_rng(0)
    for _ in range(300):
        runs, run_len = int(rng.integers(1, 3000)), int(rng.integers(1, 5000))
        n_kb, slots = int(rng.integers(1, 65)), int(rng.choice([256, 512, 1024, 2048]))
        ra, ga, ca, gb, cb, layers = plan(runs, run_len, n_kb, slots)
        assert 0 <= ra <= runs and ga >= 1 and gb >= 1 and ca >= 1 and cb >= 1
        assert layers == ra * ga + (runs - ra) * gb
        for g, c in ((ga, ca), (gb, cb)):                       # g layers of c candidates cover a run, none is empty
            assert g * c >= run_len and (g - 1) * c < run_len
            assert c >= min(16, run_len) or g == 1
        # the last region's workgroups are never longer than the first region's
        assert cb <= ca
    out = (C.c_int32 * 6)()
    assert L.hh_fused_schedule(0, 10, 1, 512, out) != 0 and L.hh_fused_schedule(5, 10, 1, 0, out) != 0


def test_general_size_plan_picks_factor_pairs_that_fit():
    """hh_general_plan (pure host arithmetic): every row length that is a product of two 7-smooth numbers <= 32 gets a
    pair r1 >= r2 with r1 r2 = nx, whole rows on a wavefront, an LDS size inside the CU's 160 KB — two buffers for the
    column factors only when that costs no resident workgroup; other lengths fall back to the Stockham kernel."""
    import ctypes as C

    from helicon_amd import _lib

    L = _lib.lib()

    def plan(nx, rows_lds=101, kg=7):
        out = (C.c_int64 * 6)()
        assert L.hh_general_plan(nx, rows_lds, kg, out) == 0
        return tuple(out)

    def smooth(r):
        for p in (2, 3, 5, 7):
            while r % p == 0:
                r //= p
        return r == 1

    assert plan(400, 201)[:5] == (20, 20, 12, 256, 1)          # 76 KB with one factor buffer: two workgroups per CU
    assert plan(200)[:5] == (20, 10, 12, 256, 2)               # short rows: three workgroups either way, double-buffered
    assert plan(154)[:2] == (0, 0) and plan(2 * 37)[:2] == (0, 0)
    covered = 0
    for nx in range(32, 1025):
        r1, r2, rpb, threads, halves, lds = plan(nx)
        has_pair = any(nx % b == 0 and b <= nx // b <= 32 and smooth(b) and smooth(nx // b) for b in range(4, 33))
        assert (r1 > 0) == has_pair, nx
        if not r1:
            continue
        covered += 1
        assert r1 * r2 == nx and 4 <= r2 <= r1 <= 32 and smooth(r1) and smooth(r2)
        assert rpb == 4 * (64 // r1) and threads == 256 and halves in (1, 2) and 0 < lds <= 160 * 1024
        if halves == 2:                                        # the second buffer did not cost a resident workgroup
            one = lds - (kg_bytes := 7 * ((nx + 3) // 4 * 4) * 4) - ((((nx + 3) // 4) + 4 + 3) // 4 * 4) * 4
            assert (160 * 1024) // lds >= min((160 * 1024) // one, 2 if nx > 256 else 3), (nx, lds, one, kg_bytes)
    assert covered >= 100
    assert plan(400, 4001)[:2] == (0, 0)                       # a table slice that cannot fit LDS: not this kernel
    assert plan(1024, 101, 32)[:2] == (0, 0)                   # more factor rows than the register prefetch holds
    out = (C.c_int64 * 6)()
    assert L.hh_general_plan(0, 1, 1, out) != 0 and L.hh_general_plan(64, 1, 33, out) != 0



_ORDER_PROBE = r"""
import sys
sys.path.insert(0, {root!r})
order = {order!r}
def load_lib():
    from helicon_amd import _lib
    _lib.lib()
def load_torch():
    import torch  # noqa: F401
for step in order:
    load_lib() if step == "lib" else load_torch()
from helicon_amd import _lib
hip = _lib.hip_runtime_paths()
hsa = _lib._mapped("libhsa-runtime64")
print("HIP", len(hip), "HSA", len(hsa), hip, hsa)
"""


@pytest.mark.parametrize("order", [("lib", "torch"), ("torch", "lib")])
def test_one_hip_runtime_whatever_the_import_order(order):
    """Round 2's "No HIP GPUs are available": libhelicon_hip.so loaded before torch used to leave TWO HIP (and HSA)
    runtimes in the process — torch's wheel carries its own copy under another file name — and the second one to
    initialise cannot open the device.  _lib._bind_hip_runtime() makes both orders end with one."""
    import subprocess
    out = subprocess.run([sys.executable, "-c", _ORDER_PROBE.format(root=str(ROOT), order=order)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("HIP")][-1].split()
    assert line[1] == "1" and line[3] == "1", out.stdout


@pytest.mark.gpu
def test_torch_works_after_the_library_has_used_the_device():
    """The failing order of round 2, in a fresh process: create and destroy contexts through the C ABI FIRST, then import
    torch and make its first allocation.  (tests/conftest.py used to hide this by initialising torch first.)"""
    import subprocess
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "import helicon_amd as H\n"
        "for n in (64, 128, (48, 96)):\n"
        "    e = H.SweepEngine(n); e.set_geometry(apix=2.0, helical_diameter=40.0, ball_radius=4.0)\n"
        "    img = e.simulate(29.0, 10.0, 1); e.close()\n"
        "import torch\n"
        "t = torch.arange(8, device='cuda').float().sum().item()\n"
        "from helicon_amd import _lib\n"
        "print('RUNTIMES', len(_lib.hip_runtime_paths()), 'SUM', t)\n" % str(ROOT))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "RUNTIMES 1 SUM 28.0" in out.stdout, out.stdout + out.stderr
