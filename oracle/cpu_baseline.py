"""CPU baseline leg of bench.py: the NumPy oracle (a port of the reference's Path-B composition,
kind = "port") timed on the host cores over a bounded sample of the C2 workload.

TEST/BENCH INFRASTRUCTURE ONLY — the thing measured here is the reported CPU baseline, never
the product path.  Worker processes are spawned (not forked) so this is safe to call from a
process that will later initialise the GPU.
"""
import multiprocessing as mp
import os
import time

import numpy as np


def _init():
    # one NumPy thread per worker: the reference pins OMP_NUM_THREADS=1 on import as well
    # (src/helicon/lib/transforms.py:8-14) and parallelises over candidates with a thread pool
    os.environ["OMP_NUM_THREADS"] = "1"


def _score_chunk(job):
    from oracle import path_b as O

    pwr_exp, mask, params, kw = job
    return [O.score_candidate(pwr_exp, mask, tw, rs, cs, **kw) for tw, rs, cs in params]


def run(*, n, apix, helical_diameter, ball_radius, truth, twists, rises, cores, n_candidates, seed=0):
    from oracle import path_b as O

    tw0, rs0, cs0 = truth
    clean = O.simulate_helical_projection(1, tw0, rs0, cs0, helical_diameter, ball_radius, 0, 0, n, n, apix)
    img = (clean + np.random.default_rng(seed).normal(0, 0.5 * clean.std(), clean.shape)).astype(np.float32)
    mask = O.radial_band_mask(n, n)
    pwr_exp = O.reference_spectrum(img, apix, log=True)
    params, valid = O.build_candidates(twists, rises, [cs0], tube_length=n * apix)
    params = params[valid]
    pick = np.linspace(0, len(params) - 1, n_candidates).astype(int)  # strided over the whole grid
    sample = params[pick]
    kw = dict(apix=apix, helical_diameter=helical_diameter, ball_radius=ball_radius, log=True)
    chunks = [(pwr_exp, mask, c, kw) for c in np.array_split(sample, cores * 4) if len(c)]
    ctx = mp.get_context("spawn")
    with ctx.Pool(cores, initializer=_init) as pool:
        pool.map(_score_chunk, chunks[:cores])  # warm the workers (imports, page-in) untimed
        t0 = time.perf_counter()
        scores = pool.map(_score_chunk, chunks, chunksize=1)
        dt = time.perf_counter() - t0
    scores = np.concatenate([np.asarray(s) for s in scores])
    return {
        "value": len(sample) / dt,
        "unit": "candidates/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{len(sample)} of the {len(params)} C2 candidates (strided), {n}x{n}, "
                  f"NumPy oracle, {cores} worker processes x 1 thread, {dt:.1f} s",
        "best_in_sample": [float(x) for x in sample[int(np.argmax(scores))]],
    }
