#!/usr/bin/env python3
"""Headline benchmark: helical-parameter candidates/s on a 512x512 image over a 100k-point
(twist, rise) grid (BASELINE.json configs[1] = SURVEY.md section 8d "C2").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C3|C4|C5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--config` picks the BASELINE configuration whose grid is sharded over the ranks: C2 (default; configs[1]), C3 (C2's grid
x Csym 1..6, configs[2]), C4 (1024^2, 500 x 500 grid, configs[3]), C5 (64 segments x 512^2 against one 200 x 100 grid,
per-segment arg-max, configs[4]).  At N = 1 with the default config the line also carries one timed leg per other
configuration, a general (non-power-of-two) size and the batched Path-A scorer under "pipelines", each with its own
roofline object.

A step = one pass of the sweep over the C2 grid with the candidate list and the experimental spectrum
already resident in HBM: every rank sweeps its contiguous shard of the ONE 400 x 250 grid (whole twists per
rank, helicon_amd.distributed.ShardedSweep), the scores are all-gathered (RCCL; N > 1 only) and every rank
takes the arg-max on the device.  That is the north_star's experiment, so N > 1 is STRONG scaling by default;
`--scaling weak` gives every rank its own full grid instead.  `python bench.py --gpus N` without a launcher
starts its own N rank processes (children, before this process touches the GPU) and relays rank 0's line.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK = 8.0e12            # B/s, MI355X spec (MI355X_MICROARCH.md "HBM3E peak BW")
F32_VECTOR_PEAK = 157.3e12   # FLOP/s, MI355X fp32 vector (= fp32 MFMA) peak, MI355X_MICROARCH.md


def c2_workload(n=512):
    from helicon_amd.grid import build_grid, sweep_axis

    apix = 1.0
    twists = sweep_axis(0.01, 4.00, 0.01)
    rises = sweep_axis(4.000, 5.245, 0.005)
    return dict(n=n, apix=apix, truth=(1.20, 4.75, 1), helical_diameter=0.4 * n * apix, ball_radius=2 * apix,
                twists=twists, rises=rises, build_grid=build_grid, csyms=[1], segments=1, name="C2")


def config_workload(name, n_override=None):
    """SURVEY.md section 8d's synthetic inputs for the BASELINE configurations (all: truth (1.20, 4.75, 1), apix 1,
    diameter 0.4 N, ball radius 2, noise 0.5 std from default_rng(segment))."""
    from helicon_amd.grid import sweep_axis

    w = c2_workload(n_override or (1024 if name == "C4" else 512))
    w["name"] = name
    if name == "C3":
        w["csyms"] = [1, 2, 3, 4, 5, 6]
    elif name == "C4":
        w["twists"], w["rises"] = sweep_axis(0.01, 5.00, 0.01), sweep_axis(4.000, 6.495, 0.005)     # 500 x 500
    elif name == "C5":
        w["twists"], w["rises"] = sweep_axis(0.02, 4.00, 0.02), sweep_axis(4.25, 5.24, 0.01)        # 200 x 100
        w["segments"] = 64
    return w


def baseline_metric():
    """The metric string of BASELINE.json (the file travels with the repo); a literal fallback otherwise."""
    try:
        return json.loads((ROOT / "BASELINE.json").read_text())["metric"]
    except Exception:
        return "helical-param candidates/sec (512² image, 100k-pt grid) + HBM roofline %"


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


def self_launch(n_ranks):
    """`python bench.py --gpus N` without a launcher: run the N ranks as children of this process (which has
    not imported torch, let alone touched the GPU) and exit with their status.  Never an exec."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def load_factory(spec):
    """`module:callable` -> the engine factory (default: helicon_amd.SweepEngine).  Tests rehearse the launch and
    collective plumbing on CPU with a stand-in engine; the product path is the default."""
    import importlib

    mod, _, name = spec.partition(":")
    return getattr(importlib.import_module(mod), name)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--config", default="C2", choices=["C2", "C3", "C4", "C5"],
                    help="BASELINE configuration whose grid is sharded over the ranks (default C2, the metric's own)")
    ap.add_argument("--segments", type=int, default=0, help="override the number of segments (tests: C5 at reduced size)")
    ap.add_argument("--grid-stride", type=int, default=1, help="keep every k-th twist of the configuration's grid (tests)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--side", dest="n", type=int, default=0, help="image side (default: the configuration's)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the one C2 grid sharded over the ranks (default, the north_star's experiment); "
                         "weak = a full grid per rank (rot = 7.5 deg x rank)")
    ap.add_argument("--csyms", type=int, nargs="+", default=None, help="Csym values of the grid (default: the configuration's)")
    ap.add_argument("--max-batch", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the run-tables / transform / C3 legs at N = 1")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-launch HIP events")
    ap.add_argument("--profile-period", type=int, default=16,
                    help="HIP events around the launches of every k-th batch of the timed sweeps (1 = all)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo + --same-device rehearses the N > 1 path on a one-GPU box")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--first-pass", default="auto", choices=["auto", "tables", "transform"],
                    help="auto: the library's choice (fused pass on twist-major grids); tables: run tables + "
                         "second pass through the HBM intermediate; transform: raster + two transforms per candidate")
    ap.add_argument("--engine", default="helicon_amd:SweepEngine", help=argparse.SUPPRESS)
    ap.add_argument("--dump-scores", default=None, help="rank 0 writes the gathered scores [S, G] of the last step (.npy)")
    return ap.parse_args(argv)


class Timed:
    """W warm-up steps, then exactly K steps bracketed by barrier + device synchronise; max over ranks."""

    def __init__(self, torch, dist, dev, world, on_gpu, backend):
        self.torch, self.dist, self.dev, self.world, self.on_gpu, self.backend = torch, dist, dev, world, on_gpu, backend

    def fence(self):
        if self.on_gpu:
            self.torch.cuda.synchronize(self.dev)
        if self.world > 1:
            self.dist.barrier()
            if self.on_gpu:
                self.torch.cuda.synchronize(self.dev)

    def run(self, step, warmup, steps, before_timed=None):
        self.fence()
        for _ in range(warmup):
            step()
        self.fence()
        if before_timed:
            before_timed()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.fence()
        elapsed = time.perf_counter() - t0
        if self.world > 1:
            t = self.torch.tensor([elapsed], dtype=self.torch.float64,
                                  device=self.dev if (self.backend == "nccl" and self.on_gpu) else "cpu")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    w = config_workload(args.config, args.n or None)
    args.n = w["n"]
    if args.csyms is None:
        args.csyms = list(w["csyms"])
    if args.segments:
        w["segments"] = args.segments
    if args.grid_stride > 1:
        w["twists"] = w["twists"][:: args.grid_stride]

    # CPU baseline first, before this process touches the GPU (it uses worker processes)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == "C2":
        cpu = cpu_baseline_leg(w)

    import torch
    import torch.distributed as dist

    import helicon_amd as H
    from helicon_amd.distributed import ShardedSweep

    Engine = load_factory(args.engine)
    on_gpu = args.engine == "helicon_amd:SweepEngine"
    if args.same_device:
        local_rank = 0
    dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    if on_gpu:
        torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl" and on_gpu:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    timed = Timed(torch, dist, dev, world, on_gpu, args.backend)

    n = w["n"]
    eng = Engine(n, device=local_rank, max_batch=args.max_batch)
    eng.set_geometry(apix=w["apix"], helical_diameter=w["helical_diameter"], ball_radius=w["ball_radius"])
    tw0, rs0, cs0 = w["truth"]
    clean = eng.simulate(tw0, rs0, cs0)
    image = np.stack([(clean + np.random.default_rng(seg).normal(0, 0.5 * clean.std(), clean.shape)).astype(np.float32)
                      for seg in range(w["segments"])])
    eng.set_reference(image if w["segments"] > 1 else image[0], H.radial_band_mask(n, n), log=True)

    n_rises = len(w["rises"])
    strong = args.scaling == "strong" or world == 1

    def make_sweep(csyms, first_pass="auto"):
        """(ShardedSweep, the whole candidate list): the list sharded over the ranks (strong) or a full grid per
        rank (weak; the list is then the concatenation of the ranks' grids)."""
        def grid_of(r):
            g = w["build_grid"](w["twists"], w["rises"], tuple(csyms), tube_length=n * w["apix"], rot=7.5 * r)
            assert g.valid.all()
            return g.params

        eng.set_table_path({"auto": 2, "tables": 1, "transform": 0}[first_pass])
        if strong:
            full = grid_of(0)
            sh = ShardedSweep(eng, full, align=n_rises, device=dev)
        else:
            full = np.concatenate([grid_of(r) for r in range(world)])
            sh = ShardedSweep(eng, full, align=len(full) // world, device=dev)
        if first_pass == "transform":
            sh.h_params = None  # no host mirror: the library cannot see the runs
        return sh, full

    sh, params_all = make_sweep(args.csyms, args.first_pass)
    g_total = sh.n_total

    # set-up: the library sizes its device buffers (run tables, column factors, moments) on the first sweep of
    # a given list, like an allocation; do that before the W warm-up steps so that W = 0 still times steady state
    sh.step()

    def start_profile():
        if not args.no_profile and on_gpu:
            # the fused pipeline runs a sweep in a handful of long launches: time all of them (events around a few
            # launches cost nothing), so the averages are over the same launches rocprofv3 sees
            eng.profile(1 if eng.last_first_pass == "fused" else args.profile_period)

    # the timed step: sweep kernels + all-gather + device arg-max + the results' copy to (pinned) host memory — SURVEY.md
    # section 8d's metric counts the result D2H; inputs (candidate list, reference spectrum) are resident in HBM
    def timed_step():
        sh.step(results_to_host=True)

    elapsed = timed.run(timed_step, args.warmup, args.steps, before_timed=start_profile)
    prof = eng.profile_get() if (not args.no_profile and on_gpu) else None
    if prof is not None:
        prof["candidates_total"] = sh.n_local * args.steps
        eng.profile(0)
    pipeline = eng.last_first_pass

    # correctness of what was timed: the arg-max of the gathered scores must be the synthetic truth
    best_all = sh.best_index()
    best = int(best_all[0])
    scores = sh.scores()
    for seg in range(scores.shape[0]):
        assert int(best_all[seg]) == int(np.argmax(np.where(np.isnan(scores[seg]), -np.inf, scores[seg]))), "device arg-max != host arg-max"
    if sh.host_scores is not None and world == 1:   # what the timed step left in host memory is the same thing
        assert np.array_equal(np.asarray(sh.host_scores).reshape(scores.shape[0], -1)[:, : scores.shape[1]], scores, equal_nan=True)
    best_pair = (round(float(params_all[best, 0]), 6), round(float(params_all[best, 1]), 6), int(params_all[best, 2]))
    if rank == 0 and args.dump_scores:
        np.save(args.dump_scores, scores)

    # where a step's time goes on this rank (untimed extra steps, device events between the phases)
    phases = phase_times(torch, sh, dev, on_gpu) if on_gpu else None

    # the host-pointer API (params H2D + scores D2H inside the call), reported beside `value`
    host_api = None
    extra = {}
    if world == 1 and on_gpu:
        eng.set_stream(None)
        eng.sweep(params_all)   # (untimed: the host-side buffers of this entry point; a whole sweep, so that every
        th = time.perf_counter()  #  k_fused_pass launch of this process is one — the profiler's average means something)
        eng.sweep(params_all)
        host_api = g_total / (time.perf_counter() - th)
        if not args.no_extra_legs and args.first_pass == "auto" and args.config == "C2" and args.csyms == [1]:
            extra = extra_legs(args, eng, timed, make_sweep, n)
            extra.update(config_legs(args, torch, dev))

    if rank == 0:
        total = g_total * args.steps
        value = total / elapsed
        b_alg = eng.algorithmic_bytes() if on_gpu else 0
        out = {
            "metric": baseline_metric(),
            "value": value,
            "unit": "candidates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"{w['name']}: {w['segments']} x {n}x{n} synthetic helix (twist 1.20, rise 4.75, csym 1, noise 0.5 std), ONE "
                             f"{len(w['twists'])}x{len(w['rises'])} (twist, rise) grid x Csym {args.csyms} "
                             + ("sharded over the ranks in whole twists" if strong else "per GPU (rot = 7.5 deg x rank)")
                             + ", radial-band mask, log1p|F|"),
                "image": n, "candidates_per_step": g_total, "candidates_per_rank": sh.per,
                "batch": getattr(eng, "max_batch", 0), "first_pass": pipeline, "segments": w["segments"],
                "timed_region": "sweep kernels + all-gather + device arg-max + scores/indices D2H into pinned host memory; "
                                "candidate list and reference spectrum resident in HBM",
                "parallelism": f"grid-shard x{world} + all-gather(scores) + device arg-max",
            },
            "argmax": {"index": best, "twist_rise_csym": best_pair,
                       "is_truth": best_pair == (tw0, rs0, cs0),
                       "segments_at_truth": int(sum(int(b) == best for b in best_all)), "segments": int(len(best_all))},
        }
        if phases is not None:
            out["step_phases_rank0"] = phases
        if host_api is not None:
            out["host_api_value"] = host_api  # hh_sweep with host buffers, PCIe-inclusive
        if prof is not None and prof["n_second_pass"] > 0:
            out["roofline"] = roofline(prof, n, b_alg, pipeline)
        if extra:
            out["pipelines"] = extra
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["gpu_over_cpu"] = value / cpu["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def phase_times(torch, sh, dev, on_gpu, reps=5):
    """ms per step of the sweep kernels, the all-gather and the arg-max on this rank (device events on the
    stream all three are queued on), and the host time to queue one step (`ms_fixed_host`, which the GPU hides
    as long as it is shorter than the device time)."""
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps)]
    torch.cuda.synchronize(dev)
    host = 0.0
    for e in ev:
        t0 = time.perf_counter()
        e[0].record()
        sh.sweep()
        e[1].record()
        sh.gather()
        e[2].record()
        sh.argmax()
        e[3].record()
        host += time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    ms = np.array([[a.elapsed_time(b) for a, b in zip(e[:-1], e[1:])] for e in ev]).mean(axis=0)
    return {"ms_sweep": float(ms[0]), "ms_allgather": float(ms[1]), "ms_argmax": float(ms[2]),
            "ms_fixed_host": 1e3 * host / reps}


def extra_legs(args, eng, timed, make_sweep, n):
    """N = 1 only: the two pipelines that DO move the intermediate through HBM, and C3 (the 600k list), each
    timed like the headline (fewer steps), so the driver's own run carries their numbers."""
    out = {}
    b_alg = eng.algorithmic_bytes()
    for name, fp, csyms, steps in (("run_tables", "tables", [1], 3), ("transform", "transform", [1], 3),
                                   ("c3_csym_1_to_6", "auto", [1, 2, 3, 4, 5, 6], 2)):
        sh, _ = make_sweep(csyms, fp)
        sh.step()

        def start():
            eng.profile(1 if eng.last_first_pass == "fused" else args.profile_period)

        el = timed.run(sh.step, 1, steps, before_timed=start)
        prof = eng.profile_get()
        prof["candidates_total"] = sh.n_local * steps
        eng.profile(0)
        leg = {"value": sh.n_total * steps / el, "unit": "candidates/s", "steps": steps, "candidates_per_step": sh.n_total,
               "first_pass": eng.last_first_pass, "argmax_index": int(sh.best_index()[0])}
        if prof["n_second_pass"] > 0:
            leg["roofline"] = roofline(prof, n, b_alg, eng.last_first_pass)
        out[name] = leg
        del sh
    eng.set_table_path(2)
    return out


def _time_device(torch, dev, fn, reps):
    """Mean device milliseconds of fn() (events on torch's current stream, which the engine is bound to)."""
    fn()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / reps


def config_legs(args, torch, dev):
    """N = 1: the other BASELINE configurations, a general image size and the batched Path-A scorer, each swept a few
    times on this GPU with its own roofline object — so that the driver's run, not only the builder's, times them."""
    import helicon_amd as H
    from helicon_amd.distributed import ShardedSweep
    from helicon_amd.grid import build_grid, sweep_axis

    out = {}

    def engine_for(shape, segments, apix, truth, noise):
        eng = H.SweepEngine(shape, device=dev.index or 0)
        geom = dict(apix=apix, helical_diameter=0.4 * eng.ny * apix, ball_radius=2 * apix)
        eng.set_geometry(**geom)
        clean = eng.simulate(truth[0], truth[1], 1)
        imgs = np.stack([(clean + np.random.default_rng(sg).normal(0, noise * clean.std(), clean.shape)).astype(np.float32)
                         for sg in range(segments)])
        eng.set_reference(imgs if segments > 1 else imgs[0], None, log=True)
        return eng, imgs, geom

    def sweep_leg(name, shape, segments, twists, rises, reps, apix=1.0, truth=(1.20, 4.75), noise=0.5):
        """One BASELINE configuration or general image size: timed steps, then (untimed) the checks of what was timed — the
        device arg-max of every segment against a host arg-max of the same scores, the synthetic truth, and 8 sampled
        scores (the best candidate among them) against the CPU oracle."""
        from oracle import path_b as O

        eng, imgs, geom = engine_for(shape, segments, apix, truth, noise)
        ny, nx = eng.ny, eng.nx
        grid = build_grid(twists, rises, (1,), tube_length=float(nx) * apix)
        sh = ShardedSweep(eng, grid.params, align=len(rises), device=dev)
        sh.step()
        ms = _time_device(torch, dev, lambda: sh.step(results_to_host=True), reps)
        best = sh.best_index()
        scores = np.asarray(sh.scores()).reshape(segments, -1)
        host_best = [int(np.nanargmax(scores[sg])) for sg in range(segments)]
        truth_i = int(np.argmin(np.abs(grid.params[:, 0] - truth[0]) + np.abs(grid.params[:, 1] - truth[1])))
        sample = sorted(set([int(best[0]), truth_i] + [int(v) for v in np.linspace(0, len(grid) - 1, 6)]))
        ref = O.sweep_cpu(imgs[0], grid.params[sample, :3], O.radial_band_mask(ny, nx), **geom)
        oracle_err = float(np.abs(scores[0, sample] - ref).max())
        cps = len(grid) / (ms * 1e-3)
        leg = {"value": cps, "unit": "candidates/s", "ms_per_step": ms, "steps": reps, "candidates_per_step": len(grid),
               "segments": segments, "image": [ny, nx], "apix": apix, "first_pass": eng.last_first_pass,
               "noise_std": noise, "segments_at_truth": int(sum(int(b) == truth_i for b in best)),
               "truth_rank_segment0": int((scores[0] > scores[0, truth_i]).sum()),   # candidates scoring above the truth (0: it leads)
               "best_twist_rise_segment0": [float(grid.params[int(best[0]), 0]), float(grid.params[int(best[0]), 1])],
               "argmax_equals_host_argmax": bool(all(int(b) == h for b, h in zip(best, host_best))),
               "oracle_max_abs_err": oracle_err, "oracle_samples": len(sample)}
        if not leg["argmax_equals_host_argmax"] or oracle_err > 2e-4:
            leg["check_failed"] = True
        flops = 2.5 * ny * nx * np.log2(ny * nx)        # SURVEY.md section 8d's 5 N^2 log2 N for an ny x nx image
        if segments == 1:
            tf = flops * cps / 1e12
            leg["roofline"] = {"bound": "fp32_vector", "achieved": tf, "peak": F32_VECTOR_PEAK / 1e12, "unit": "TFLOP/s",
                               "frac": tf / (F32_VECTOR_PEAK / 1e12), "alg_flop_per_candidate": flops, "traffic": None,
                               "kernel": "k_fused_pass" if shape == ny == nx and (ny & (ny - 1)) == 0 else "k_gen_rows (nx = R1 R2; gen_rows.hip)",
                               "note": "device time of the whole step (sweep + arg-max + D2H), events on the sweep's stream"}
        else:
            # several segments: the masked spectrum q of every candidate is stored once and read once (K bins x 4 B each way);
            # tuned sizes up to 512 keep only the bins with weight (round 4), in slices of 512
            k_bins = (ny // 2 + 1) * nx
            if shape == ny == nx and (ny & (ny - 1)) == 0 and ny <= 512:
                m = H.radial_band_mask(ny, nx)
                k_bins = -(-int(m[ny // 2:, :].sum() + m[0, :].sum()) // 512) * 512
            moved = 2.0 * 4 * k_bins
            gbps = moved * cps / 1e9
            leg["scores_per_s"] = cps * segments
            leg["roofline"] = {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": gbps / (HBM_PEAK / 1e9),
                               "moved_bytes_per_candidate": moved, "traffic": None,
                               "kernel": ("k_fused_pass<EPI_QSTORE>" if shape == ny == nx and (ny & (ny - 1)) == 0 else "k_gen_rows (q store)") + " + k_segment_corr",
                               "note": "q = masked log-spectrum of a candidate, written by the fused pass and read by the "
                                       "segment contraction; device time of the whole step"}
        out[name] = leg
        sh = None
        eng.close()

    tw, rs = sweep_axis(0.01, 4.00, 0.01), sweep_axis(4.000, 5.245, 0.005)
    # General image sizes at the pixel size the reference app bins such boxes to (2 - 5 A, app.py:1911-1922): a 200-pixel box
    # at 1 A holds 42 rises of a 1.2-degree twist and cannot tell the truth from its neighbours (round 3's leg sat at 0 of 1).
    tw5, rs5 = sweep_axis(0.05, 20.0, 0.05), sweep_axis(20.0, 26.225, 0.025)     # 400 x 250, truth (6.0, 23.75) a grid point
    legs = (
        ("C4_1024", 1024, 1, sweep_axis(0.01, 5.00, 0.01), sweep_axis(4.000, 6.495, 0.005), 2, {}),
        ("C5_64_segments", 512, 64, sweep_axis(0.02, 4.00, 0.02), sweep_axis(4.25, 5.24, 0.01), 5, {}),
        ("general_400", (400, 400), 1, tw5[70:170], rs5, 3, dict(apix=5.0, truth=(6.0, 23.75))),
        # (a 200-pixel box holds 42 subunits and the rise grid's step is a thousandth of the rise: at noise 0.5 and 0.25 std the
        # next rise leads by 1e-5 in the CPU oracle too; 0.1 std separates them by 3e-4)
        ("general_200", (200, 200), 1, tw5[20:220], rs5, 3, dict(apix=5.0, truth=(6.0, 23.75), noise=0.1)),
        ("general_400_64_segments", (400, 400), 64, tw5[70:170], rs5, 3, dict(apix=5.0, truth=(6.0, 23.75))),
    )
    for name, shape, segments, tws, rss, reps, kw in legs:
        try:
            sweep_leg(name, shape, segments, tws, rss, reps, **kw)
        except Exception as ex:   # a leg must not take the headline down with it
            out[name] = {"error": f"{type(ex).__name__}: {ex}"}
    for name, interp in (("path_a", "nn"), ("path_a_linear", "linear")):
        try:
            out[name] = path_a_leg(dev.index or 0, interpolation=interp)
        except Exception as ex:
            out[name] = {"error": f"{type(ex).__name__}: {ex}"}
    return out


def path_a_leg(device, total=1024, interpolation="nn"):
    """The reference's shipped scorer (sparse least squares + cosine, solver_linear_regression.py:31-547) batched on the
    device: `total` (twist, rise) candidates of a 64 x 128 image (the size the reference app works at after binning to
    target_apix2d), set-up included; "nn" = nearest-neighbour projector, "linear" = trilinear (the app's default,
    app.py:577-585).  Untimed: the best candidate's score against the CPU oracle's lsq_reconstruct, the arg-max against the
    synthetic truth.  Roofline: HBM; bytes = LSMR iterations x the arrays an iteration must touch (Path A)."""
    import helicon_amd as H
    from helicon_amd.solver import lsq_reconstruct_batch
    from oracle import path_a as A

    ny, nx, l3 = 64, 128, 16
    eng = H.SweepEngine((ny, nx), device=device)
    eng.set_geometry(apix=5.0, helical_diameter=0.5 * ny * 5.0, ball_radius=10.0)
    image = eng.simulate(29.0, 20.0, 1).astype(np.float32)
    eng.close()
    box = dict(reconstruct_diameter_2d_pixel=ny, reconstruct_diameter_3d_pixel=ny, reconstruct_length_2d_pixel=nx,
               reconstruct_length_3d_pixel=l3)
    kw = dict(box, return_3d=False, device=device, interpolation=interpolation)
    if interpolation == "linear":
        total = min(total, 512)
    tw = np.linspace(27.0, 31.0, total)
    lsq_reconstruct_batch(image, 1.0, [(t, 4.0, 1) for t in tw[:: total // 16]], **kw)   # warm
    st = {}
    t0 = time.perf_counter()
    res = lsq_reconstruct_batch(image, 1.0, [(float(t), 4.0, 1) for t in tw], batch=512, streams=2, stats=st, **kw)
    dt = time.perf_counter() - t0
    info = np.asarray(st["info"])
    iters, first = int(info[:, 3].sum()), int(info[:, 4].sum())
    scores = np.array([sc for _, sc in res])
    best_i = int(np.argmax(scores))
    # the best candidate through the CPU oracle (6 s for "nn", 30 s for "linear": one candidate)
    t1 = time.perf_counter()
    want = A.lsq_reconstruct(image, 1.0, float(tw[best_i]), 4.0, 1, interpolation=interpolation, **box)[1]
    oracle_s = time.perf_counter() - t1
    err = abs(float(scores[best_i]) - float(want))
    n, md, ms = 47952.0, 33291.0, 59424.0    # unknowns, data rows, symmetry rows of this box (tools/path_a_bench.py prints them)
    if interpolation == "nn":
        mp = 2 * md * 64 * 2                 # the uint16 ray map, both products
        sym = 16 * ms + 4 * n                # pairs (A x) and their transposed lists (A^T y)
        kernel = "k_pabs_matvec<1> + k_pabs_rmatvec<1> (one LSMR iteration of every candidate that is not done)"
    else:
        mp = 2 * 4 * 2.7e6                   # footprint lists (2.7 MB per candidate), read by each of the 4 groups in both products
        sym = 128 * ms + 8 * 16 * ms + 4 * n   # 16-entry rows (A x) and their transposed (row, weight) lists (A^T y)
        kernel = "k_pabf_matvec<1> + k_pabf_tail<1> + k_pabf_scatter<1> + k_pabl_finish<1>"
    plain = mp + sym + 8 * (3 * (md + ms) + 9 * n)
    aug = mp + sym + 8 * (3 * (md + ms) + 16 * n)
    moved = plain * first + aug * (iters - first)
    gbps = moved / dt / 1e9
    leg = {"value": total / dt, "unit": "candidates/s", "candidates": total, "seconds": dt, "set_up_included": True,
           "interpolation": interpolation, "lsmr_iterations": iters, "self_check_failures": int(st.get("self_check_failures", 0)),
           "best_twist": float(tw[best_i]), "truth_twist": 29.0, "best_is_truth": bool(abs(float(tw[best_i]) - 29.0) <= 0.15),
           "oracle_score_of_best": float(want), "oracle_abs_err": err, "oracle_seconds_for_one_candidate": oracle_s,
           "groups": st.get("groups"), "round2_value": 45.2 if interpolation == "nn" else 11.3,
           "roofline": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": gbps / (HBM_PEAK / 1e9),
                        "alg_bytes_per_lsmr_iteration": {"plain": plain, "augmented": aug}, "traffic": None, "kernel": kernel,
                        "note": "wall time of the whole call (set-up, all trust-region steps, host polling) against the bytes of "
                                "its LSMR iterations only; DESIGN.md (Path A) says what bounds the kernels"}}
    # the bounded trust-region solve is loosely converged (lsq_linear tol 1e-2): 2e-3 is what it reproduces (tests/test_gpu_path_a.py)
    if err > (1e-4 if interpolation == "nn" else 5e-3) or not leg["best_is_truth"] or leg["self_check_failures"]:
        leg["check_failed"] = True
    return leg


PIPELINES = {
    # pipeline -> ((JSON name, traffic.json key, what it moves per candidate), ...) for profile slots 0 and 1
    "transform": (("k_first_pass", "first_pass", "writes the half spectrum"),
                  ("k_second_pass", "second_pass", "reads the half spectrum")),
    "run_tables": (("k_first_pass_table", "first_pass_table", "writes the half spectrum"),
                   ("k_second_pass", "second_pass", "reads the half spectrum")),
    "fused": (("k_column_factors", "column_factors", "first batch of a sweep only; later batches ride in k_fused_pass"),
              ("k_fused_pass", "fused_pass", "no intermediate: table slice + column factors from L2, moments out")),
}


def traffic_table(n):
    """Measured bytes per candidate per kernel (profiles/traffic.json, written from separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes by tools/traffic_parse.py) and a label saying where they come from."""
    tfile = ROOT / "profiles" / "traffic.json"
    try:
        doc = json.loads(tfile.read_text())
        return doc.get(f"n{n}", {}), f"profiles/traffic.json ({doc.get('source', 'rocprofv3 --pmc passes, not this run')})"
    except Exception:
        return {}, None


def roofline(prof, n, b_alg, pipeline):
    """One object per timed pipeline; every `frac` in it is a fraction of a physical peak (<= 1).

    * two-pass pipelines (`transform`, `run_tables`): bound = HBM.  The kernels write the column-transformed half
      spectrum (8 N (N/2+1) bytes per candidate) and read it back; the simulated image is never stored, so the bytes
      moved are 16 N (N/2+1), not SURVEY section 8d's B_alg (which adds 4 N^2 for "reading the image").  `achieved` =
      moved bytes x candidates / device time of the sampled launches (HIP events on the sweep's stream).
    * `fused`: the half spectrum stays in LDS, HBM sees ~5 % of those bytes, so HBM does not bound the kernel; the
      bound that applies is the fp32 vector pipe.  `achieved` = section 8d's FLOP count (5 N^2 log2 N per candidate)
      x candidates / device time against the 157.3 TFLOP/s fp32 vector peak.  The section-8d HBM model is kept as
      `hbm_model` (a two-pass-equivalent rate, explicitly not a bound) and the measured traffic as `hbm_measured`.
    `traffic` = measured HBM-side bytes per launch of the dominant kernel(s), from `traffic_source` (a static file of
    earlier counter passes — counters cannot be collected inside a timed run), or null."""
    half = 8 * n * (n // 2 + 1)
    slots = (("ms_first_pass", "n_first_pass"), ("ms_second_pass", "n_second_pass"))
    cand = prof["candidates"]
    measured, source = traffic_table(n)
    kernels = {}
    main_launches = max(prof["n_first_pass"], prof["n_second_pass"])
    for (name, tkey, what), (ms_key, n_key) in zip(PIPELINES[pipeline], slots):
        launches = prof[n_key]
        if launches == 0:
            continue
        avg_us = 1e3 * prof[ms_key] / launches
        per_launch = cand / launches
        per_cand = measured.get(tkey)  # measured bytes per candidate
        entry = {"launches": launches, "avg_us": avg_us, "role": what,
                 "traffic": per_cand * per_launch if per_cand is not None else None}
        if pipeline != "fused":
            gbps = half * per_launch / (avg_us * 1e-6) / 1e9
            entry.update({"candidates_per_launch": per_launch, "moved_bytes_per_candidate": half, "GBps": gbps,
                          "frac": gbps / (HBM_PEAK / 1e9)})
        elif name == "k_fused_pass":
            entry["candidates_per_launch"] = per_launch
        kernels[name] = entry
    # run-table builds (one launch per sweep) are timed on every sweep but serve all of its batches:
    # scale them to the sampled share of the candidates
    share = cand / max(1, prof.get("candidates_total", cand))
    device_ms = prof["ms_first_pass"] + prof["ms_second_pass"] + prof["ms_finalize"] + prof["ms_centres"] * share
    dominant = [v for k, v in kernels.items() if k != "k_column_factors"]
    traffic = None
    if dominant and all(v["traffic"] is not None for v in dominant):
        traffic = sum(v["traffic"] for v in dominant)
    cand_per_s = cand / (device_ms * 1e-3)
    out = {
        "pipeline": pipeline,
        "kernel": " + ".join(k for k in kernels if k != "k_column_factors") + " (per batch)",
        "candidates_per_launch": cand / main_launches, "device_ms_sampled": device_ms,
        "run_table_ms_per_sweep": (prof["ms_centres"] / prof["n_centres"]) if prof["n_centres"] else None,
        "traffic": traffic, "traffic_source": source,
        "kernels": kernels,
    }
    model_gbps = b_alg * cand_per_s / 1e9
    if pipeline == "fused":
        flops = 5.0 * n * n * np.log2(n)  # r2c 2-D FFT, 11.8 MFLOP at 512
        tf = flops * cand_per_s / 1e12
        out.update({"bound": "fp32_vector", "achieved": tf, "peak": F32_VECTOR_PEAK / 1e12, "unit": "TFLOP/s",
                    "frac": tf / (F32_VECTOR_PEAK / 1e12), "alg_flop_per_candidate": flops})
        out["hbm_model"] = {"alg_bytes_per_candidate": b_alg, "two_pass_equivalent_GBps": model_gbps,
                            "ratio_to_hbm_peak": model_gbps / (HBM_PEAK / 1e9),
                            "note": "not a bound: section 8d's B_alg prices a half spectrum written to and read from HBM; "
                                    "the fused pass keeps it in LDS, so these bytes are not moved"}
        per_cand = measured.get("fused_pass")
        if per_cand is not None:
            gb = per_cand * cand_per_s / 1e9
            out["hbm_measured"] = {"bytes_per_candidate": per_cand, "GBps": gb, "hbm_measured_frac": gb / (HBM_PEAK / 1e9),
                                   "traffic_source": source}
    else:
        moved = 2 * half
        gbps = moved * cand_per_s / 1e9
        out.update({"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": gbps / (HBM_PEAK / 1e9), "moved_bytes_per_candidate": moved,
                    "hbm_model": {"alg_bytes_per_candidate": b_alg, "GBps": model_gbps,
                                  "ratio_to_hbm_peak": model_gbps / (HBM_PEAK / 1e9),
                                  "note": "section 8d's B_alg counts 4 N^2 for reading an image that no pipeline stores"}})
    return out


if __name__ == "__main__":
    main()
