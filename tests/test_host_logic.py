"""Host-side logic (no GPU): grid construction, masks, sharding, and the library's exports."""
import re
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
    with pytest.raises(ValueError):
        H.SweepEngine(48)


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


def test_bench_roofline_object_has_the_contract_keys_for_every_pipeline():
    """bench.py's roofline() on canned profile numbers (no GPU): the keys the driver and the judge read."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    b_alg = 4 * 512 * 512 + 16 * 512 * 257
    prof = dict(ms_first_pass=12.0, n_first_pass=250, ms_second_pass=14.5, n_second_pass=250, ms_finalize=0.01,
                n_finalize=1, ms_centres=0.2, n_centres=10, candidates=62500, candidates_total=1000000)
    for pipeline in ("transform", "run_tables"):
        r = bench.roofline(prof, 512, b_alg, pipeline)
        assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernels"} <= set(r)
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0 < r["frac"] < 1
        assert len(r["kernels"]) == 2 and all("avg_us" in k for k in r["kernels"].values())
    fused = dict(prof, ms_first_pass=1.3, n_first_pass=10, ms_second_pass=250.0, n_second_pass=40, candidates=1000000)
    r = bench.roofline(fused, 512, b_alg, "fused")
    assert r["pipeline"] == "fused" and "k_fused_pass" in r["kernels"] and "valu" in r and "note" in r
    assert r["frac"] > 1.0 > r["valu"]["frac"] > 0          # B_alg is not moved; the vector figure is the bound
    assert r["traffic"] is None or r["traffic"] < 0.1 * b_alg * r["candidates_per_launch"]
