#!/usr/bin/env python3
"""Path A on the GPU at the size the reference app works at after its binning to target_apix2d (a 64 x 128 pixel
projection, 64-voxel cylinder): the batched device-resident solver (helicon_amd.lsq_reconstruct_batch, nearest
neighbour) over K candidates, set-up and solve timed apart, with its launch / synchronisation counters and the bytes
its products move; then the single-candidate calls (nn through a batch of one, trilinear through hh_pa), and with
`--oracle` the CPU oracle's time for one call (seconds).  `--json` prints one JSON object (bench.py's path_a leg)."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import helicon_amd as H  # noqa: E402
from helicon_amd.solver import PathABatch, hh_pa_params, lsq_reconstruct, lsq_reconstruct_batch  # noqa: E402

NY, NX, L3 = 64, 128, 16
KW = dict(reconstruct_diameter_2d_pixel=NY, reconstruct_diameter_3d_pixel=NY, reconstruct_length_2d_pixel=NX,
          reconstruct_length_3d_pixel=L3)
# the reference's row target min(2**26, max(D2d * L2d, unknowns) * sym_oversample) (solver:148-150, 168-170) for this box
# with sym_oversample = 1: 47,952 voxels in the cylinder (round 2's tool, and this one until the profile of this round,
# passed D2d * L2d = 8,192 here: a six times smaller data block than lsq_reconstruct builds)
TARGET = 47952


def test_image():
    eng = H.SweepEngine((NY, NX))
    eng.set_geometry(apix=5.0, helical_diameter=0.5 * NY * 5.0, ball_radius=10.0)
    return eng.simulate(29.0, 20.0, 1).astype(np.float32)


def batch_run(image, k, repeat=2):
    """K candidates around the truth (twist 27 .. 31 degrees, rise 4 px): returns timings and counters of the best run."""
    twists = np.linspace(27.0, 31.0, k)
    cands = [(float(t), 4.0, 1) for t in twists]
    best = None
    for _ in range(repeat):
        params = [hh_pa_params(1.0, t, r, c, 0.0, 0.0, 0.0, NY, NX, NY, 0, L3, TARGET, TARGET, 0, 0, 0) for t, r, c in cands]
        t0 = time.perf_counter()
        B = PathABatch(image, params)
        t1 = time.perf_counter()
        _, scores, info = B.solve(np.ones(k, dtype=np.int32), 0, want_x=False)
        t2 = time.perf_counter()
        cnt = B.counters()
        run = dict(k=k, setup_s=t1 - t0, solve_s=t2 - t1, total_s=t2 - t0, candidates_per_s=k / (t2 - t0),
                   solve_candidates_per_s=k / (t2 - t1), unknowns=B.n, data_rows=int(B.m_data.mean()), sym_rows=int(B.m_sym.mean()),
                   device_bytes=B.device_bytes, lsmr_iterations=int(info[:, 3].sum()), lsmr_solves=int(info[:, 2].sum()),
                   outer_iterations=int(info[:, 1].sum()), best_twist=float(twists[int(np.argmax(scores))]),
                   best_score=float(scores.max()), **cnt)
        # Algorithmic bytes of ONE LSMR iteration of one candidate (Path A): every array an iteration must touch,
        # once per kernel that needs it — the uint16 map (padded rows) in both products, the symmetry pairs (A x) and their
        # transposed lists (A^T y), u read + written (A x) and read (A^T y), v read twice and written once, h / hbar / x read
        # and written; with the trust-region step's operator [A diag(d); diag(root)] also d (both products), root (both)
        # and the n extra entries of u.
        n, md, ms = float(B.n), float(np.mean(B.m_data)), float(np.mean(B.m_sym))
        mp = 2 * md * ((NY + 63) // 64 * 64) * 2
        plain = mp + 16 * ms + 4 * n + 8 * (3 * (md + ms) + 9 * n)
        aug = mp + 16 * ms + 4 * n + 8 * (3 * (md + ms) + 16 * n)
        first = int(info[:, 4].sum())
        run["bytes_per_lsmr_iteration_plain"] = plain
        run["bytes_per_lsmr_iteration_augmented"] = aug
        run["first_solve_iterations"] = first
        run["bytes_per_lsmr_iteration"] = (plain * first + aug * (run["lsmr_iterations"] - first)) / max(1, run["lsmr_iterations"])
        B.close()
        if best is None or run["total_s"] < best["total_s"]:
            best = run
    return best


def main():
    image = test_image()
    lsq_reconstruct(image, 1.0, 29.0, 4.0, 1, interpolation="nn", **KW)   # warm (module load, first allocations)
    runs = [batch_run(image, k) for k in ((256,) if "--json" in sys.argv else (1, 16, 64, 256, 512))]
    if "--json" in sys.argv:
        print(json.dumps(runs[-1]))
        return
    for r in runs:
        per_it = r["solve_s"] / max(1, r["lsmr_iterations"])
        print(f"batch of {r['k']:4d}: set-up {r['setup_s'] * 1e3:7.1f} ms, solve {r['solve_s'] * 1e3:7.1f} ms -> "
              f"{r['candidates_per_s']:7.1f} candidates/s ({r['solve_candidates_per_s']:.1f} solve only); unknowns {r['unknowns']}, "
              f"rows {r['data_rows']} + {r['sym_rows']}; LSMR iterations {r['lsmr_iterations']} in {r['lsmr_solves']} solves, "
              f"{r['outer_iterations']} trust-region iterations; {r['launches']} launches, {r['host_syncs']} host syncs, "
              f"{r['lsmr_iterations_queued']} iterations queued; {per_it * 1e6:.2f} us per candidate-iteration = "
              f"{r['bytes_per_lsmr_iteration'] / per_it / 1e12:.2f} TB/s of algorithmic traffic; device memory "
              f"{r['device_bytes'] / 2**30:.2f} GiB; best twist {r['best_twist']:.3f} ({r['best_score']:.4f})", flush=True)
    for interp in ("nn", "linear"):
        t0 = time.perf_counter()
        scores = [lsq_reconstruct(image, 1.0, tw, 4.0, 1, interpolation=interp, **KW)[1] for tw in (27.0, 29.0, 31.0)]
        t_call = (time.perf_counter() - t0) / 3
        print(f"{interp}: one lsq_reconstruct call {t_call * 1e3:.1f} ms; scores at 27/29/31 deg {np.round(scores, 4)}", flush=True)
    t0 = time.perf_counter()
    res = lsq_reconstruct_batch(image, 1.0, [(t, 4.0, 1) for t in np.linspace(27, 31, 100)], return_3d=True, **KW)
    print(f"lsq_reconstruct_batch, 100 candidates with their maps: {time.perf_counter() - t0:.3f} s; best "
          f"{np.linspace(27, 31, 100)[int(np.argmax([s for _, s in res]))]:.3f}", flush=True)
    # the whole thing as a user calls it: groups of `batch` candidates on `streams` concurrent HIP streams
    for total, batch, streams in ((1024, 256, 1), (1024, 128, 4), (1024, 64, 8), (1024, 64, 16), (1024, 32, 16), (2048, 128, 8)):
        tw = np.linspace(27.0, 31.0, total)
        st = {}
        t0 = time.perf_counter()
        res = lsq_reconstruct_batch(image, 1.0, [(t, 4.0, 1) for t in tw], return_3d=False, batch=batch, streams=streams, stats=st, **KW)
        dt = time.perf_counter() - t0
        print(f"lsq_reconstruct_batch: {total} candidates, groups of {batch} on {streams} stream(s): {dt:.3f} s = {total / dt:.1f} "
              f"candidates/s (set-up included); {st['launches']} launches, {st['host_syncs']} host syncs; best "
              f"{tw[int(np.argmax([s for _, s in res]))]:.3f}", flush=True)
    if "--oracle" in sys.argv:
        from oracle import path_a as A
        t0 = time.perf_counter()
        s = A.lsq_reconstruct(image, 1.0, 29.0, 4.0, 1, interpolation="nn", **KW)[1]
        print(f"CPU oracle (NumPy / SciPy restatement), nn: {time.perf_counter() - t0:.1f} s, score {s:.4f}")


if __name__ == "__main__":
    main()
