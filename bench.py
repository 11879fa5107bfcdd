#!/usr/bin/env python3
"""Headline benchmark: helical-parameter candidates/s on a 512x512 image over a 100k-point
(twist, rise) grid (BASELINE.json configs[1] = SURVEY.md section 8d "C2"), one MI355X per rank.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the sweep over this rank's 100k-candidate shard with the parameters and the
experimental spectrum already resident in HBM, followed (N > 1) by the RCCL all-gather of the
scores.  Weak scaling: every rank owns one full 400 x 250 grid at Csym = 1 (rank r sweeps it at
azimuthal phase rot = 7.5 r degrees, so all ranks do identical work): 100k x N candidates per step.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md "HBM3E peak BW")


def c2_workload(n=512):
    from helicon_amd.grid import build_grid, sweep_axis

    apix = 1.0
    twists = sweep_axis(0.01, 4.00, 0.01)
    rises = sweep_axis(4.000, 5.245, 0.005)
    return dict(n=n, apix=apix, truth=(1.20, 4.75, 1), helical_diameter=0.4 * n * apix, ball_radius=2 * apix,
                twists=twists, rises=rises, build_grid=build_grid)


def baseline_metric():
    """The metric string of BASELINE.json (the file travels with the repo); a literal fallback otherwise."""
    try:
        return json.loads((ROOT / "BASELINE.json").read_text())["metric"]
    except Exception:
        return "helical-param candidates/sec (512\u00b2 image, 100k-pt grid) + HBM roofline %"


def usable_cores():
    """Host cores this job may really use: the affinity mask, clipped by the cgroup CPU quota and
    by the 16-core share a one-GPU box grants (override with HELICON_CPU_CORES)."""
    env = os.environ.get("HELICON_CPU_CORES")
    if env:
        return max(1, int(env))
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 16)


def cpu_baseline_leg(w, sample_per_core=48):
    """The CPU oracle (NumPy port of the reference path) on this host's cores, bounded sample."""
    from oracle import cpu_baseline

    cores = usable_cores()
    return cpu_baseline.run(n=w["n"], apix=w["apix"], helical_diameter=w["helical_diameter"],
                            ball_radius=w["ball_radius"], truth=w["truth"], twists=w["twists"], rises=w["rises"],
                            cores=cores, n_candidates=sample_per_core * cores)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=512, help="image side (default: the C2 workload)")
    ap.add_argument("--max-batch", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-launch HIP events")
    ap.add_argument("--profile-period", type=int, default=16,
                    help="HIP events around the launches of every k-th batch of the timed sweeps (1 = all)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo + --same-device rehearses the N > 1 path on a one-GPU box")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--first-pass", default="auto", choices=["auto", "tables", "transform"],
                    help="auto: the library's choice (fused pass on twist-major grids); tables: run tables + "
                         "second pass through the HBM intermediate; transform: raster + two transforms per candidate")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    w = c2_workload(args.n)

    # CPU baseline first, before this process touches the GPU (it uses worker processes)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_leg(w)

    import torch
    import torch.distributed as dist

    import helicon_amd as H
    from helicon_amd.distributed import gather_scores

    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    n = w["n"]
    eng = H.SweepEngine(n, device=local_rank, max_batch=args.max_batch)
    eng.set_geometry(apix=w["apix"], helical_diameter=w["helical_diameter"], ball_radius=w["ball_radius"])
    tw0, rs0, cs0 = w["truth"]
    clean = eng.simulate(tw0, rs0, cs0)
    noise = np.random.default_rng(0).normal(0, 0.5 * clean.std(), clean.shape)
    image = (clean + noise).astype(np.float32)
    eng.set_reference(image, H.radial_band_mask(n, n), log=True)

    grid = w["build_grid"](w["twists"], w["rises"], (1,), tube_length=n * w["apix"], rot=7.5 * rank)
    assert grid.valid.all()
    g_local = len(grid)
    stream = torch.cuda.current_stream(dev)
    eng.set_stream(stream.cuda_stream)
    d_params = torch.from_numpy(grid.params).to(dev)
    d_scores = torch.empty((1, g_local), dtype=torch.float32, device=dev)
    h_params = grid.params if args.first_pass != "transform" else None  # the host mirror lets the library see the runs
    eng.set_table_path({"auto": 2, "tables": 1, "transform": 0}[args.first_pass])

    def step():
        eng.sweep_device(d_params.data_ptr(), g_local, d_scores.data_ptr(), host_params=h_params)
        if world > 1:
            local = d_scores if args.backend == "nccl" else d_scores.cpu()
            return gather_scores(local, g_local * world, g_local)
        return d_scores

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # set-up: the library sizes its device buffers (run tables, column factors, moments) on the first sweep of
    # a given list, like an allocation; do that before the W warm-up steps so that W = 0 still times steady state
    step()
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    if not args.no_profile:
        # the fused pipeline runs a sweep in a handful of long launches: time all of them (events around a few
        # launches cost nothing), so the averages are over the same launches rocprofv3 sees
        eng.profile(1 if eng.last_first_pass == "fused" else args.profile_period)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full = step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_get() if not args.no_profile else None
    if prof is not None:
        prof["candidates_total"] = g_local * args.steps
    eng.profile(0)

    # the host-pointer API (params H2D + scores D2H inside the call), reported beside `value`
    host_api = None
    if world == 1:
        eng.sweep(grid.params[:1024])
        th = time.perf_counter()
        eng.sweep(grid.params)
        host_api = g_local / (time.perf_counter() - th)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness of what was timed: rank 0's grid (rot = 0) must have its arg-max at the truth
    scores = full.cpu().numpy().reshape(world, g_local)
    best = int(np.argmax(scores[0]))
    _, bt, br_ = np.unravel_index(best, (1, len(w["twists"]), len(w["rises"])))
    best_pair = (round(float(grid.params[best, 0]), 6), round(float(grid.params[best, 1]), 6))

    if rank == 0:
        total = g_local * world * args.steps
        value = total / elapsed
        b_alg = eng.algorithmic_bytes()
        out = {
            "metric": baseline_metric(),
            "value": value,
            "unit": "candidates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"C2: {n}x{n} synthetic helix (twist 1.20, rise 4.75, csym 1, noise 0.5 std), "
                            f"400x250 (twist, rise) grid per GPU, Csym = 1, rot = 7.5 deg x rank, radial-band mask, log1p|F|",
                "image": n, "grid_per_gpu": g_local, "candidates_per_step": g_local * world,
                "batch": eng.max_batch, "first_pass": eng.last_first_pass, "parallelism": f"grid-shard x{world} + all-gather(scores)",
            },
            "argmax": {"twist": best_pair[0], "rise": best_pair[1], "is_truth": best_pair == (tw0, rs0)},
            "hbm_roofline_frac_wall": value / world * b_alg / HBM_PEAK,
        }
        if host_api is not None:
            out["host_api_value"] = host_api  # hh_sweep with host buffers, PCIe-inclusive
        if prof is not None and prof["n_second_pass"] > 0:
            out["roofline"] = roofline(prof, n, b_alg, eng.last_first_pass)
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["gpu_over_cpu"] = value / cpu["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


F32_VECTOR_PEAK = 157.3e12  # FLOP/s, MI355X fp32 vector (= fp32 MFMA) peak, MI355X_MICROARCH.md

PIPELINES = {
    # pipeline -> ((JSON name, traffic.json key, what it moves per candidate), ...) for profile slots 0 and 1
    "transform": (("k_first_pass", "first_pass", "writes the half spectrum"),
                  ("k_second_pass", "second_pass", "reads the half spectrum")),
    "run_tables": (("k_first_pass_table", "first_pass_table", "writes the half spectrum"),
                   ("k_second_pass", "second_pass", "reads the half spectrum")),
    "fused": (("k_column_factors", "column_factors", "first batch of a sweep only; later batches ride in k_fused_pass"),
              ("k_fused_pass", "fused_pass", "no intermediate: table slice + column factors from L2, moments out")),
}


def roofline(prof, n, b_alg, pipeline):
    """achieved = B_alg(N) bytes per candidate x candidates / device time of the
    launches that process them, all measured with HIP events on the sweep's stream over sampled
    batches of the timed region, against the 8 TB/s HBM peak.  B_alg = 4 N^2 + 16 N (N/2+1) prices a
    pipeline that writes the column-transformed half spectrum to HBM and reads it back.  The
    `transform` and `run_tables` pipelines do exactly that (`kernels` gives each kernel's own half
    and its measured traffic).  The `fused` pipeline keeps the intermediate in LDS, so it moves
    almost none of B_alg (`traffic` << `achieved` x time) and the same formula can exceed 1.0: the
    section-8d roofline does not bound it.  The kernel is compute-side limited (vector and LDS pipes each
    about 60 % busy, waves waiting on LDS round trips at 4 waves per SIMD: DESIGN.md section 4); `valu`
    prices the section-8d FLOP count (5 N^2 log2 N per candidate) against the fp32 vector peak.
    `traffic` = measured bytes per launch (bytes per candidate from the calibrated FETCH_SIZE /
    WRITE_SIZE passes in profiles/traffic.json x the candidates of an average sampled launch)."""
    half = 8 * n * (n // 2 + 1)
    slots = (("ms_first_pass", "n_first_pass"), ("ms_second_pass", "n_second_pass"))
    cand = prof["candidates"]
    tfile = ROOT / "profiles" / "traffic.json"  # written from the rocprofv3 --pmc passes
    measured = {}
    if tfile.exists():
        try:
            measured = json.loads(tfile.read_text()).get(f"n{n}", {})
        except Exception:
            measured = {}
    kernels = {}
    main_launches = max(prof["n_first_pass"], prof["n_second_pass"])
    for (name, tkey, what), (ms_key, n_key) in zip(PIPELINES[pipeline], slots):
        launches = prof[n_key]
        if launches == 0:
            continue
        avg_us = 1e3 * prof[ms_key] / launches
        per_launch = cand / launches
        per_cand = measured.get(tkey)  # measured bytes per candidate (profiles/traffic.json)
        entry = {"launches": launches, "avg_us": avg_us, "role": what,
                 "traffic": per_cand * per_launch if per_cand is not None else None}
        if pipeline != "fused":
            gbps = half * per_launch / (avg_us * 1e-6) / 1e9
            entry.update({"candidates_per_launch": per_launch, "alg_bytes_per_candidate": half, "GBps": gbps,
                          "frac": gbps / (HBM_PEAK / 1e9)})
        elif name == "k_fused_pass":
            entry["candidates_per_launch"] = cand / launches
        kernels[name] = entry
    # run-table builds (one launch per sweep) are timed on every sweep but serve all of its batches:
    # scale them to the sampled share of the candidates
    share = cand / max(1, prof.get("candidates_total", cand))
    device_ms = prof["ms_first_pass"] + prof["ms_second_pass"] + prof["ms_finalize"] + prof["ms_centres"] * share
    achieved = b_alg * cand / (device_ms * 1e-3) / 1e9
    traffic = None
    dominant = [v for k, v in kernels.items() if k != "k_column_factors"]
    if dominant and all(v["traffic"] is not None for v in dominant):
        traffic = sum(v["traffic"] for v in dominant)
    out = {
        "bound": "hbm", "pipeline": pipeline,
        "kernel": " + ".join(k for k in kernels if k != "k_column_factors") + " (per batch)",
        "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / (HBM_PEAK / 1e9),
        "traffic": traffic, "alg_bytes_per_candidate": b_alg,
        "candidates_per_launch": cand / main_launches, "device_ms_sampled": device_ms,
        "run_table_ms_per_sweep": (prof["ms_centres"] / prof["n_centres"]) if prof["n_centres"] else None,
        "kernels": kernels,
    }
    if pipeline == "fused":
        flops = 5.0 * n * n * np.log2(n)  # r2c 2-D FFT, 11.8 MFLOP at 512
        tf = flops * cand / (device_ms * 1e-3) / 1e12
        out["note"] = ("fused pass: the half spectrum never goes to HBM, so B_alg is not moved and frac can exceed 1; "
                       "the kernel is limited on the compute side (vector + LDS pipes, see valu and DESIGN.md section 4)")
        out["valu"] = {"bound": "fp32 vector", "alg_flop_per_candidate": flops, "achieved": tf,
                       "peak": F32_VECTOR_PEAK / 1e12, "unit": "TFLOP/s", "frac": tf / (F32_VECTOR_PEAK / 1e12)}
    return out


if __name__ == "__main__":
    main()
