#!/usr/bin/env python3
"""Path A on the GPU (helicon_amd.lsq_reconstruct) at the size the reference app works at after its binning to
target_apix2d (a 64 x 128 pixel projection, 64-voxel cylinder): set-up and solve times, nearest neighbour and trilinear,
with the CPU oracle's time for the same call beside them (`--oracle`; minutes)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import helicon_amd as H  # noqa: E402
from helicon_amd.solver import PathAProblem, lsq_reconstruct  # noqa: E402


def main():
    ny, nx, l3 = 64, 128, 16
    eng = H.SweepEngine((ny, nx))
    eng.set_geometry(apix=5.0, helical_diameter=0.5 * ny * 5.0, ball_radius=10.0)
    image = eng.simulate(29.0, 20.0, 1).astype(np.float32)
    kw = dict(reconstruct_diameter_2d_pixel=ny, reconstruct_diameter_3d_pixel=ny, reconstruct_length_2d_pixel=nx,
              reconstruct_length_3d_pixel=l3)
    for interp in ("nn", "linear"):
        lsq_reconstruct(image, 1.0, 29.0, 4.0, 1, interpolation=interp, **kw)   # warm
        t0 = time.perf_counter()
        P = PathAProblem(image, scale2d_to_3d=1.0, twist_degree=29.0, rise_pixel=4.0, csym=1, tilt_degree=0, psi_degree=0,
                         dy_pixel=0, reconstruct_diameter_3d_inner_pixel=0, min_projection_lines=ny * nx,
                         min_sym_pairs=ny * nx, interpolation=interp, **kw)
        t_setup = time.perf_counter() - t0
        x = np.random.default_rng(0).normal(size=P.n)
        t0 = time.perf_counter()
        for _ in range(20):
            y = P.matvec(x)
            P.rmatvec(y)
        t_pair = (time.perf_counter() - t0) / 20
        dims = (P.n, P.m_data, P.m_sym, P.n_ops)
        P.close()
        t0 = time.perf_counter()
        scores = [lsq_reconstruct(image, 1.0, tw, 4.0, 1, interpolation=interp, **kw)[1] for tw in (27.0, 29.0, 31.0)]
        t_call = (time.perf_counter() - t0) / 3
        print(f"{interp}: unknowns {dims[0]}, data rows {dims[1]}, symmetry rows {dims[2]}, operations {dims[3]}; set-up "
              f"{t_setup * 1e3:.1f} ms, A x + A^T y (host vectors) {t_pair * 1e3:.2f} ms, lsq_reconstruct {t_call * 1e3:.1f} ms; "
              f"scores at 27/29/31 deg {np.round(scores, 4)}", flush=True)
    # the reference drives its scorer from a thread pool (app.py:2473-2476): calls on different candidates overlap on the
    # device (one hh_pa and one stream each; ctypes releases the GIL)
    from concurrent.futures import ThreadPoolExecutor
    twists = [27.0 + 0.25 * k for k in range(32)]
    for threads in (1, 4, 8, 16):
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=threads) as pool:
            scores = list(pool.map(lambda tw: lsq_reconstruct(image, 1.0, tw, 4.0, 1, interpolation="nn", **kw)[1], twists))
        dt = time.perf_counter() - t0
        print(f"nn, {len(twists)} candidates from {threads} thread(s): {len(twists) / dt:.1f} candidates/s; best twist "
              f"{twists[int(np.argmax(scores))]}", flush=True)
    if "--oracle" in sys.argv:
        from oracle import path_a as A
        t0 = time.perf_counter()
        s = A.lsq_reconstruct(image, 1.0, 29.0, 4.0, 1, interpolation="nn", **kw)[1]
        print(f"CPU oracle (NumPy / SciPy restatement), nn: {time.perf_counter() - t0:.1f} s, score {s:.4f}")


if __name__ == "__main__":
    main()
