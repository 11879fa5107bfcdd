#!/usr/bin/env python3
"""Operator-level comparison of the batched Path-A products (hh_pab_matvec / hh_pab_rmatvec) with the single-candidate
projector (hh_pa), candidate by candidate: dims, right-hand sides, A x, A^T y.  argv: nn | linear"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from helicon_amd.solver import PathABatch, PathAProblem, hh_pa_params  # noqa: E402

interp = sys.argv[1] if len(sys.argv) > 1 else "linear"
g = np.load(Path(__file__).resolve().parent.parent / "tests" / "golden" / "g5_lsq.npz")
img = np.ascontiguousarray(g["helix_image"], dtype=np.float32)
code = 1 if interp == "linear" else 0
specs = [(25.0, 2.0, 1), (29.0, 2.0, 1), (58.0, 4.0, 2), (-29.0, 2.0, 1), (27.5, 1.7, 1), (31.0, 2.5, 1)]
want = 960
params = [hh_pa_params(1.0, tw, rs, cs, 0.0, 0.0, 0.0, 20, 32, 20, 0, 6, want, want, code, 0, 0) for tw, rs, cs in specs]
rng = np.random.default_rng(0)
with PathABatch(img, params) as B:
    for c, (tw, rs, cs) in enumerate(specs):
        with PathAProblem(img, scale2d_to_3d=1.0, twist_degree=tw, rise_pixel=rs, csym=cs, tilt_degree=0, psi_degree=0, dy_pixel=0,
                          reconstruct_diameter_2d_pixel=20, reconstruct_length_2d_pixel=32, reconstruct_diameter_3d_pixel=20,
                          reconstruct_diameter_3d_inner_pixel=0, reconstruct_length_3d_pixel=6, min_projection_lines=want,
                          min_sym_pairs=want, interpolation=interp) as P:
            dims = (B.n, int(B.m_data[c]), int(B.m_sym[c]), int(B.n_ops[c])), (P.n, P.m_data, P.m_sym, P.n_ops)
            b, pid = B.rhs(c)
            same_rhs = dims[0][1] == dims[1][1] and np.array_equal(b, P.b_data) and np.array_equal(pid, P.b_pid)
            line = f"{interp} cand {c} {specs[c]}: dims batch {dims[0]} single {dims[1]} rhs equal {same_rhs}"
            if dims[0] == dims[1]:
                x = rng.normal(size=P.n)
                y = rng.normal(size=P.m)
                ya, yb = B.matvec(c, x), P.matvec(x)
                ga, gb = B.rmatvec(c, y), P.rmatvec(y)
                bad = np.argsort(-np.abs(ya - yb))[:3]
                line += (f"; A x max diff {np.abs(ya - yb).max():.3e} (data {np.abs(ya - yb)[:P.m_data].max():.3e}, scale {np.abs(yb).max():.2f}) "
                         f"worst rows {bad.tolist()}; A^T y max diff {np.abs(ga - gb).max():.3e} (scale {np.abs(gb).max():.2f})")
            print(line, flush=True)

if "--dense" in sys.argv:
    tw, rs, cs = specs[1]
    with PathABatch(img, params[1:2]) as B, PathAProblem(img, scale2d_to_3d=1.0, twist_degree=tw, rise_pixel=rs, csym=cs, tilt_degree=0,
                                                         psi_degree=0, dy_pixel=0, reconstruct_diameter_2d_pixel=20,
                                                         reconstruct_length_2d_pixel=32, reconstruct_diameter_3d_pixel=20,
                                                         reconstruct_diameter_3d_inner_pixel=0, reconstruct_length_3d_pixel=6,
                                                         min_projection_lines=want, min_sym_pairs=want, interpolation=interp) as P:
        n = P.n
        Da = np.stack([B.matvec(0, np.eye(n)[v]) for v in range(n)], axis=1)
        Db = np.stack([P.matvec(np.eye(n)[v]) for v in range(n)], axis=1)
        nsl = n // 6
        print("dense", Da.shape, "nnz batch", int((Da != 0).sum()), "single", int((Db != 0).sum()), "row sums batch", Da[:P.m_data].sum(1)[:6], "single", Db[:P.m_data].sum(1)[:6])
        for r in (0, 1, 100, 500, 1000):
            ia, ib = np.nonzero(Da[r])[0], np.nonzero(Db[r])[0]
            print("row", r, "batch planes", sorted(set((ia // nsl).tolist())), "single planes", sorted(set((ib // nsl).tolist())),
                  "in-plane idx batch", (ia % nsl)[:8].tolist(), "single", (ib % nsl)[:8].tolist(),
                  "w batch", np.round(Da[r, ia][:6], 4).tolist(), "single", np.round(Db[r, ib][:6], 4).tolist())
        bad = np.nonzero(np.abs(Da - Db).max(axis=1) > 1e-9)[0]
        print("rows that differ:", len(bad), "of", Da.shape[0], "first", bad[:40].tolist())
        pid = P.b_pid
        for r in bad[:12]:
            if r >= P.m_data:
                print("sym row", r - P.m_data, "batch", Da[r][np.nonzero(Da[r])[0]][:8], "single", Db[r][np.nonzero(Db[r])[0]][:8]); continue
            ib = np.nonzero(Db[r])[0]
            ia = np.nonzero(Da[r])[0]
            print("row", r, "pid (k, j)", divmod(int(pid[r]), 20), "single planes", sorted(set((ib // nsl).tolist())), "sum", Db[r].sum(), "batch planes",
                  sorted(set((ia // nsl).tolist())), "sum", Da[r].sum(), "max diff", np.abs(Da[r] - Db[r]).max())
        ks = sorted(set(int(pid[r]) // 20 for r in bad if r < P.m_data))
        print("columns k of the differing data rows:", ks)
        # and the transposed product, column by column of A^T = row by row of A
        m = P.m
        Ta = np.stack([B.rmatvec(0, np.eye(m)[r]) for r in range(0, m, 7)], axis=0)
        Tb = np.stack([P.rmatvec(np.eye(m)[r]) for r in range(0, m, 7)], axis=0)
        print("A^T rows sampled:", Ta.shape[0], "max diff vs single", np.abs(Ta - Tb).max(), "max diff vs own forward", np.abs(Ta - Da[::7]).max())
        for r in list(range(0, 14)) + [1030, 1036]:
            ia, ib = np.nonzero(Da[r])[0], np.nonzero(Db[r])[0]
            print("row", r, divmod(int(pid[r]), 20), "batch idx", ia[:10].tolist(), np.round(Da[r, ia][:10], 3).tolist(), "| single idx", ib[:10].tolist(), np.round(Db[r, ib][:10], 3).tolist())
        # is the batch a row permutation of the single?
        used = set()
        hits = 0
        for r in range(P.m_data):
            cand = np.nonzero(np.abs(Db[:P.m_data] - Da[r]).max(axis=1) < 1e-9)[0]
            if len(cand):
                hits += 1
                if r < 30 or r > P.m_data - 5: print("batch row", r, "== single row(s)", cand[:4].tolist())
        print("batch rows equal to some single row:", hits, "of", P.m_data)
