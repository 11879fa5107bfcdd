"""CPU oracle for Path A (the reference's shipped denovo3D scorer): sparse least-squares reconstruction of the
helical volume from one projection + cosine score, nearest-neighbour interpolation, ``model="lsq"``.

TEST INFRASTRUCTURE ONLY — a restatement of the reference's algorithm used to check the HIP path; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.  Pinned by tests/golden/g4_path_a.npz and
g5_lsq.npz (outputs of the reference itself, tests/golden/make_golden.py).

Restated functions (paths under /root/reference/src/helicon/):
  get_cylindrical_mask                  lib/analysis.py:731-774
  back_project_2d_coords_to_3d_coords   webApps/denovo3D/solver_linear_regression.py:1657-1746
  halton_order                          scipy.stats.qmc.Halton(d=1, scramble=False).integers(0, n, n) as used at
                                        solver_linear_regression.py:1566-1571, 1785-1790 (van der Corput base 2)
  sorted_hsym_csym_pairs                solver_linear_regression.py:1749-1791
  build_A_data_matrix (nn)              solver_linear_regression.py:1304-1654
  build_A_helical_sym_matrix (nn)       solver_linear_regression.py:847-1298
  lsmr                                  scipy.sparse.linalg.lsmr (Fong & Saunders 2011) as called by
                                        scipy.optimize.lsq_linear for an unbounded sparse problem
                                        (solver_linear_regression.py:258-269: tol=1e-2 -> atol = btol = 1e-4, maxiter 1000)
  lsq_reconstruct (lsq, nn)             solver_linear_regression.py:31-547
  cosine_similarity                     lib/analysis.py:802-821
"""
from __future__ import annotations

import itertools

import numpy as np
from scipy.sparse import csr_matrix, vstack


def get_cylindrical_mask(nz, ny, nx, rmin=0, rmax=-1, return_xyz=False):
    k = np.arange(0, nz, dtype=np.int32) - nz // 2
    j = np.arange(0, ny, dtype=np.int32) - ny // 2
    i = np.arange(0, nx, dtype=np.int32) - nx // 2
    Z, Y, X = np.meshgrid(k, j, i, indexing="ij")
    if rmax < 0:
        rmax = ny // 2 - 1
    mask = X * X + Y * Y < rmax * rmax
    if 0 < rmin < rmax:
        mask &= X * X + Y * Y >= rmin * rmin
    return (mask, (Z, Y, X)) if return_xyz else mask


def _quat_axis(axis: str, deg: float) -> np.ndarray:
    a = np.deg2rad(deg)
    q = np.zeros(4)
    q["xyz".index(axis)] = np.sin(a / 2)
    q[3] = np.cos(a / 2)
    return q


def _quat_mul(p, q):
    out = np.empty(4)
    out[:3] = p[3] * q[:3] + q[3] * p[:3] + np.cross(p[:3], q[:3])
    out[3] = p[3] * q[3] - np.dot(p[:3], q[:3])
    return out


def _quat_matrix(q) -> np.ndarray:
    """Rotation matrix of a (not re-normalised) quaternion (x, y, z, w), entry by entry as
    scipy.spatial.transform.Rotation.as_matrix computes it (checked bit for bit against scipy 1.15.3)."""
    x, y, z, w = q
    x2, y2, z2, w2 = x * x, y * y, z * z, w * w
    xy, zw, xz, yw, yz, xw = x * y, z * w, x * z, y * w, y * z, x * w
    return np.array([[x2 - y2 - z2 + w2, 2 * (xy - zw), 2 * (xz + yw)],
                     [2 * (xy + zw), -x2 + y2 - z2 + w2, 2 * (yz - xw)],
                     [2 * (xz - yw), 2 * (yz + xw), -x2 - y2 + z2 + w2]])


def euler_matrix(seq: str, angles) -> np.ndarray:
    """Rotation.from_euler(seq, angles, degrees=True).as_matrix() for lower-case (extrinsic) sequences."""
    angles = np.atleast_1d(angles)
    q = _quat_axis(seq[0], angles[0])
    for ax, a in zip(seq[1:], angles[1:]):
        q = _quat_mul(_quat_axis(ax, a), q)
    return _quat_matrix(q)


def rot_apply(m, coords, inverse=False):
    """Rotation.apply: out_j = v_0 m[j,0] + v_1 m[j,1] + v_2 m[j,2] (m[., j] for the inverse), summed left to right."""
    return np.einsum("kj,ik->ij" if inverse else "jk,ik->ij", m, coords)


def back_project_2d_coords_to_3d_coords(image, scale2d_to_3d, reconstruct_diameter_2d_pixel=-1,
                                        reconstruct_length_2d_pixel=-1):
    ny, nx = image.shape
    d2 = ny if reconstruct_diameter_2d_pixel <= 0 else reconstruct_diameter_2d_pixel
    l2 = nx if reconstruct_length_2d_pixel <= 0 else reconstruct_length_2d_pixel
    d2, l2 = int(np.rint(d2)), int(np.rint(l2))
    k = np.arange(0, d2, dtype=np.int32) - d2 // 2
    j = np.arange(0, d2, dtype=np.int32) - d2 // 2
    i = np.arange(0, l2, dtype=np.int32) - l2 // 2
    region = image[np.ix_(j + ny // 2, i + nx // 2)]
    Z, Y, X = np.meshgrid(k.astype(np.float32), j.astype(np.float32), i.astype(np.float32), indexing="ij")
    coords = np.vstack((X.ravel(), Y.ravel(), Z.ravel())).transpose().astype(np.float64)
    coords = rot_apply(euler_matrix("y", 90), coords, inverse=True)
    if scale2d_to_3d != 1.0:
        coords *= scale2d_to_3d
    shp = (d2, d2, l2)
    X2 = np.swapaxes(coords[:, 0].reshape(shp), 0, 2)
    Y2 = np.swapaxes(coords[:, 1].reshape(shp), 0, 2)
    Z2 = np.swapaxes(coords[:, 2].reshape(shp), 0, 2)
    assert X2[:, :, 0].shape[::-1] == region.shape
    return (X2, Y2, Z2), region


def halton_order(n: int) -> np.ndarray:
    """floor(vdC_2(i) * n), i = 0 .. n-1 (vdC_2(0) = 0): the index list qmc.Halton(d=1, scramble=False).integers
    gives.  For n not a power of two some indices repeat and some never occur (the reference keeps that)."""
    out = np.empty(n, dtype=np.int64)
    for i in range(n):
        f, r, k = 0.5, 0.0, i
        while k:
            if k & 1:
                r += f
            k >>= 1
            f *= 0.5
        out[i] = int(np.floor(r * n))
    return out


def sorted_hsym_csym_pairs(twist, rise, csym, nz):
    hsym_max = max(1, int(np.ceil(nz / (2 * rise))))
    hcsyms = itertools.product(range(-hsym_max, hsym_max + 1), range(csym))
    rows = []
    for p in itertools.combinations(hcsyms, r=2):
        (h1, c1), (h2, c2) = p
        a1 = twist * h1 + c1 * 360 / csym
        a2 = twist * h2 + c2 * 360 / csym
        angle = round(abs((a2 - a1 + 180) % 360 - 180), 2)
        rows.append((angle, abs(h1 + h2), abs(h1 - h2), abs(h1), abs(h2), p))
    rows.sort(key=lambda x: x[:-1])
    return [rows[int(i)] for i in halton_order(len(rows))]


def hcsym_order(twist_degree, rise_pixel, csym, reconstruct_length_3d_pixel, nz):
    """Symmetry operations (h, c) in the order build_A_data_matrix visits them (solver:1556-1571)."""
    hsym_max = max(1, int(np.ceil(reconstruct_length_3d_pixel + nz) / 2 / rise_pixel))
    hcsyms = list(itertools.product(range(-hsym_max, hsym_max + 1), range(csym)))
    hcsyms.sort(key=lambda x: (abs(x[0]), x[1]))
    return [hcsyms[int(i)] for i in halton_order(len(hcsyms))]


def build_A_data_matrix(image, scale2d_to_3d, twist_degree, rise_pixel, csym, tilt_degree, psi_degree, dy_pixel,
                        reconstruct_diameter_2d_pixel, reconstruct_length_2d_pixel, reconstruct_diameter_3d_pixel,
                        reconstruct_diameter_3d_inner_pixel, reconstruct_length_3d_pixel, min_projection_lines,
                        interpolation="nn"):
    assert interpolation in ("nn", "linear"), "the oracle restates the nearest-neighbour and the trilinear branch"
    (X0, Y0, Z0), pixel_vals = back_project_2d_coords_to_3d_coords(image, scale2d_to_3d, reconstruct_diameter_2d_pixel,
                                                                   reconstruct_length_2d_pixel)
    rmin = reconstruct_diameter_3d_inner_pixel / 2
    rmax = reconstruct_diameter_3d_pixel // 2 - 1
    nz, ny, nx = X0.shape
    if reconstruct_length_3d_pixel <= 0:
        reconstruct_length_3d_pixel = nz
    mask = get_cylindrical_mask(reconstruct_length_3d_pixel, ny, nx, rmin=rmin, rmax=rmax)
    n_x = int(np.count_nonzero(mask))
    rank = np.zeros(mask.shape, dtype=np.int64) - 1
    rank[np.nonzero(mask)] = np.arange(n_x)
    coords0 = np.vstack((X0.ravel(), Y0.ravel(), Z0.ravel())).transpose().copy()
    coords0[:, 1] -= dy_pixel
    coords0 = rot_apply(euler_matrix("yx", (tilt_degree, psi_degree)), coords0, inverse=True)
    mz, my, mx = mask.shape
    blocks, bs, pids = [], [], []
    n_b = 0
    for hi, ci in hcsym_order(twist_degree, rise_pixel, csym, reconstruct_length_3d_pixel, nz):
        coords = rot_apply(euler_matrix("z", twist_degree * hi + 360 * ci / csym), coords0, inverse=True)
        coords[:, 2] -= hi * rise_pixel
        X = coords[:, 0].reshape((nz, ny, nx)) + nx // 2
        Y = coords[:, 1].reshape((nz, ny, nx)) + ny // 2
        Z = coords[:, 2].reshape((nz, ny, nx)) + reconstruct_length_3d_pixel // 2
        if interpolation == "linear":
            # solver:1414-1503: int() truncates towards zero; a sample counts when its whole 2x2x2 cell lies in the
            # cylinder; weights in float64, summed per (row, voxel), stored as float32
            zi, yi, xi = np.trunc(Z).astype(np.int64), np.trunc(Y).astype(np.int64), np.trunc(X).astype(np.int64)
            ok = (zi >= 0) & (zi + 1 <= mz - 1) & (yi >= 0) & (yi + 1 <= my - 1) & (xi >= 0) & (xi + 1 <= mx - 1)
            corner = np.full((8,) + zi.shape, -1, dtype=np.int64)
            for c, (dz, dy_, dx) in enumerate(itertools.product((0, 1), (0, 1), (0, 1))):   # 000, 001, 010, ... (z, y, x)
                corner[c][ok] = rank[zi[ok] + dz, yi[ok] + dy_, xi[ok] + dx]
            hit = ok & (corner >= 0).all(axis=0)
            zf, yf, xf = Z - zi, Y - yi, X - xi
            wts = [(1 - zf) * (1 - yf) * (1 - xf), (1 - zf) * (1 - yf) * xf, (1 - zf) * yf * (1 - xf), (1 - zf) * yf * xf,
                   zf * (1 - yf) * (1 - xf), zf * (1 - yf) * xf, zf * yf * (1 - xf), zf * yf * xf]
            has = hit.any(axis=2)
            row_of = np.cumsum(has.ravel()).reshape(has.shape) - 1
            kk, jj, _ = np.nonzero(hit)
            rows = np.tile(row_of[kk, jj], 8)
            cols = np.concatenate([corner[c][hit] for c in range(8)])
            vals = np.concatenate([wts[c][hit] for c in range(8)])
            n_rows = int(has.sum())
            if n_rows:
                m64 = csr_matrix((vals, (rows, cols)), shape=(n_rows, n_x), dtype=np.float64)
                m64.sum_duplicates()
                blocks.append(m64.astype(np.float32))
        else:
            zi, yi, xi = np.rint(Z).astype(np.int64), np.rint(Y).astype(np.int64), np.rint(X).astype(np.int64)
            ok = (zi >= 0) & (zi <= mz - 1) & (yi >= 0) & (yi <= my - 1) & (xi >= 0) & (xi <= mx - 1)
            idx = np.full(zi.shape, -1, dtype=np.int64)
            idx[ok] = rank[zi[ok], yi[ok], xi[ok]]   # -1 outside the cylinder
            hit = idx >= 0                             # [k, j, i]
            has = hit.any(axis=2)                      # rows exist for rays with at least one sample in the mask
            row_of = np.cumsum(has.ravel()).reshape(has.shape) - 1
            kk, jj, _ = np.nonzero(hit)
            rows = row_of[kk, jj]
            cols = idx[hit]
            n_rows = int(has.sum())
            if n_rows:
                blocks.append(csr_matrix((np.ones(len(cols), dtype=np.float32), (rows, cols)), shape=(n_rows, n_x),
                                         dtype=np.float32))   # duplicates add up: an entry is a hit count
        if n_rows:
            k_idx, j_idx = np.nonzero(has)
            bs.append(pixel_vals[j_idx, k_idx].astype(np.float32))
            pids.append((k_idx * ny + j_idx).astype(np.int32))
        n_b += n_rows
        if min_projection_lines > 0 and n_b > min_projection_lines:
            break
    return vstack(blocks).tocsr(), np.concatenate(bs).astype(np.float32), np.concatenate(pids)


def _build_A_helical_sym_matrix_linear(nz, ny, nx, twist_degree, rise_pixel, csym, rmin, rmax, min_sym_pairs):
    """The trilinear branch (solver:910-1140), quirks included: a pair counts only when |dz|, |dy| and |dx| of the two
    truncated positions are all >= 3; de-duplication by the ROUNDED positions' ranks; the weights of corners 110 and 111
    are xf*yf*(1-xf) and xf*yf*zf."""
    pairs = sorted_hsym_csym_pairs(twist_degree, rise_pixel, csym, nz)
    mask, (Z, Y, X) = get_cylindrical_mask(nz, ny, nx, rmin=rmin, rmax=rmax, return_xyz=True)
    n_x = int(np.count_nonzero(mask))
    mz_i, my_i, mx_i = np.nonzero(mask)
    rank = np.zeros(mask.shape, dtype=np.int64) - 1
    rank[(mz_i, my_i, mx_i)] = np.arange(n_x)
    xyz = np.vstack((X.ravel(), Y.ravel(), Z.ravel())).transpose().astype(np.float64)
    seen = {-1}
    r_idx, c_idx, vals = [], [], []
    row_count = 0
    corners = list(itertools.product((0, 1), (0, 1), (0, 1)))  # (dz, dy, dx): 000, 001, 010, 011, 100, 101, 110, 111

    def weights(zf, yf, xf):
        return [(1 - zf) * (1 - yf) * (1 - xf), (1 - zf) * (1 - yf) * xf, (1 - zf) * yf * (1 - xf), (1 - zf) * yf * xf,
                zf * (1 - yf) * (1 - xf), zf * (1 - yf) * xf, xf * yf * (1 - xf), xf * yf * zf]

    for p in pairs:
        (h_i, c_i), (h_j, c_j) = p[-1]

        def image_of(h, c):
            t = rot_apply(euler_matrix("z", twist_degree * h + c * 360 / csym), xyz, inverse=False)
            return (t[:, 0].reshape(mask.shape) + nx // 2, t[:, 1].reshape(mask.shape) + ny // 2,
                    t[:, 2].reshape(mask.shape) + nz // 2 + rise_pixel * h)

        Xi, Yi, Zi = image_of(h_i, c_i)
        Xj, Yj, Zj = image_of(h_j, c_j)
        added = 0
        for m in range(n_x):
            k, j, i = mz_i[m], my_i[m], mx_i[m]
            a = (int(Zi[k, j, i]), int(Yi[k, j, i]), int(Xi[k, j, i]))
            b = (int(Zj[k, j, i]), int(Yj[k, j, i]), int(Xj[k, j, i]))
            lim = (nz, ny, nx)
            if any(not (0 <= a[q] and a[q] + 1 <= lim[q] - 1 and 0 <= b[q] and b[q] + 1 <= lim[q] - 1) for q in range(3)):
                continue
            ia = [rank[a[0] + dz, a[1] + dy_, a[2] + dx] for dz, dy_, dx in corners]
            ib = [rank[b[0] + dz, b[1] + dy_, b[2] + dx] for dz, dy_, dx in corners]
            if min(ia) < 0 or min(ib) < 0:
                continue
            if abs(a[0] - b[0]) < 3 or abs(a[1] - b[1]) < 3 or abs(a[2] - b[2]) < 3:
                continue
            ir = rank[round(Zi[k, j, i]), round(Yi[k, j, i]), round(Xi[k, j, i])]
            jr = rank[round(Zj[k, j, i]), round(Yj[k, j, i]), round(Xj[k, j, i])]
            pid = int(ir) * n_x + int(jr)
            if pid in seen:
                continue
            seen.add(pid)
            seen.add(int(jr) * n_x + int(ir))
            wa = weights(Zi[k, j, i] - a[0], Yi[k, j, i] - a[1], Xi[k, j, i] - a[2])
            wb = weights(Zj[k, j, i] - b[0], Yj[k, j, i] - b[1], Xj[k, j, i] - b[2])
            for q in range(8):
                r_idx.append(row_count + added)
                c_idx.append(int(ia[q]))
                vals.append(wa[q])
            for q in range(8):
                r_idx.append(row_count + added)
                c_idx.append(int(ib[q]))
                vals.append(-wb[q])
            added += 1
        row_count += added
        if row_count >= min_sym_pairs:
            break
    if not row_count:
        return None, None
    A = csr_matrix((np.asarray(vals, dtype=np.float32), (r_idx, c_idx)), shape=(row_count, n_x), dtype=np.float32)
    return A, np.zeros(row_count, dtype=np.float32)


def build_A_helical_sym_matrix(nz, ny, nx, twist_degree, rise_pixel, csym, rmin, rmax, min_sym_pairs,
                               interpolation="nn"):
    if interpolation == "linear":
        return _build_A_helical_sym_matrix_linear(nz, ny, nx, twist_degree, rise_pixel, csym, rmin, rmax, min_sym_pairs)
    assert interpolation == "nn"
    pairs = sorted_hsym_csym_pairs(twist_degree, rise_pixel, csym, nz)
    mask, (Z, Y, X) = get_cylindrical_mask(nz, ny, nx, rmin=rmin, rmax=rmax, return_xyz=True)
    n_x = int(np.count_nonzero(mask))
    mz_i, my_i, mx_i = np.nonzero(mask)
    rank = np.zeros(mask.shape, dtype=np.int64) - 1
    rank[(mz_i, my_i, mx_i)] = np.arange(n_x)
    xyz = np.vstack((X.ravel(), Y.ravel(), Z.ravel())).transpose().astype(np.float64)
    seen = {-1}
    rows_i, rows_j = [], []
    row_count = 0
    for p in pairs:
        (h_i, c_i), (h_j, c_j) = p[-1]

        def image_of(h, c):
            t = rot_apply(euler_matrix("z", twist_degree * h + c * 360 / csym), xyz, inverse=False)
            return (np.rint(t[:, 0].reshape(mask.shape) + nx // 2).astype(np.int64),
                    np.rint(t[:, 1].reshape(mask.shape) + ny // 2).astype(np.int64),
                    np.rint(t[:, 2].reshape(mask.shape) + nz // 2 + rise_pixel * h).astype(np.int64))

        xi, yi, zi = image_of(h_i, c_i)
        xj, yj, zj = image_of(h_j, c_j)
        added = 0
        for m in range(n_x):
            k, j, i = mz_i[m], my_i[m], mx_i[m]
            a = (zi[k, j, i], yi[k, j, i], xi[k, j, i])
            b = (zj[k, j, i], yj[k, j, i], xj[k, j, i])
            if not (0 <= a[0] < nz and 0 <= b[0] < nz and 0 <= a[1] < ny and 0 <= b[1] < ny and 0 <= a[2] < nx and 0 <= b[2] < nx):
                continue
            ia, ib = rank[a], rank[b]
            if ia < 0 or ib < 0:
                continue
            pid = int(ia) * n_x + int(ib)
            if pid in seen:
                continue
            seen.add(pid)
            seen.add(int(ib) * n_x + int(ia))
            rows_i.append(int(ia))
            rows_j.append(int(ib))
            added += 1
        row_count += added
        if row_count >= min_sym_pairs:
            break
    if not row_count:
        return None, None
    r = np.arange(row_count)
    A = csr_matrix((np.r_[np.ones(row_count, np.float32), -np.ones(row_count, np.float32)],
                    (np.r_[r, r], np.r_[rows_i, rows_j])), shape=(row_count, n_x), dtype=np.float32)
    return A, np.zeros(row_count, dtype=np.float32)


def lsmr(A, b, atol=1e-4, btol=1e-4, conlim=1e8, maxiter=1000):
    """LSMR (Fong & Saunders, SIAM J. Sci. Comput. 33, 2011) for min ||A x - b||, no damping, x0 = 0; the stopping
    rules of scipy.sparse.linalg.lsmr.  Returns (x, istop, itn, normr, normar)."""
    A = A.tocsr() if hasattr(A, "tocsr") else A
    At = A.T.tocsr() if hasattr(A, "tocsr") else A.T
    b = np.asarray(b, dtype=np.float64)
    m, n = A.shape
    u = b.copy()
    normb = np.linalg.norm(b)
    beta = normb
    x = np.zeros(n)
    if beta > 0:
        u /= beta
        v = At @ u
        alpha = np.linalg.norm(v)
    else:
        v = np.zeros(n)
        alpha = 0.0
    if alpha > 0:
        v /= alpha
    zetabar, alphabar, rho, rhobar, cbar, sbar = alpha * beta, alpha, 1.0, 1.0, 1.0, 0.0
    h, hbar = v.copy(), np.zeros(n)
    betadd, betad, rhodold, tautildeold, thetatilde, zeta, d = beta, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0
    normA2, maxrbar, minrbar = alpha * alpha, 0.0, 1e100
    normA, condA, normx = np.sqrt(normA2), 1.0, 0.0
    normr, normar = beta, alpha * beta
    ctol = 1.0 / conlim if conlim > 0 else 0.0
    itn, istop = 0, 0
    if normar == 0:
        return x, istop, itn, normr, normar
    if normb == 0:
        return x, istop, itn, normr, normar

    def sym_ortho(a, b_):
        if b_ == 0:
            return np.sign(a), 0.0, abs(a)
        if a == 0:
            return 0.0, np.sign(b_), abs(b_)
        if abs(b_) > abs(a):
            tau = a / b_
            s = np.sign(b_) / np.sqrt(1 + tau * tau)
            return s * tau, s, b_ / s
        tau = b_ / a
        c = np.sign(a) / np.sqrt(1 + tau * tau)
        return c, c * tau, a / c

    while itn < maxiter:
        itn += 1
        u *= -alpha
        u += A @ v
        beta = np.linalg.norm(u)
        if beta > 0:
            u *= 1 / beta
            v *= -beta
            v += At @ u
            alpha = np.linalg.norm(v)
            if alpha > 0:
                v *= 1 / alpha
        chat, shat, alphahat = sym_ortho(alphabar, 0.0)
        rhoold = rho
        c, s, rho = sym_ortho(alphahat, beta)
        thetanew = s * alpha
        alphabar = c * alpha
        rhobarold, zetaold = rhobar, zeta
        thetabar = sbar * rho
        rhotemp = cbar * rho
        cbar, sbar, rhobar = sym_ortho(cbar * rho, thetanew)
        zeta = cbar * zetabar
        zetabar = -sbar * zetabar
        hbar *= -(thetabar * rho / (rhoold * rhobarold))
        hbar += h
        x += (zeta / (rho * rhobar)) * hbar
        h *= -(thetanew / rho)
        h += v
        betaacute = chat * betadd
        betacheck = -shat * betadd
        betahat = c * betaacute
        betadd = -s * betaacute
        thetatildeold = thetatilde
        ctildeold, stildeold, rhotildeold = sym_ortho(rhodold, thetabar)
        thetatilde = stildeold * rhobar
        rhodold = ctildeold * rhobar
        betad = -stildeold * betad + ctildeold * betahat
        tautildeold = (zetaold - thetatildeold * tautildeold) / rhotildeold
        taud = (zeta - thetatilde * tautildeold) / rhodold
        d = d + betacheck * betacheck
        normr = np.sqrt(d + (betad - taud) ** 2 + betadd * betadd)
        normA2 = normA2 + beta * beta
        normA = np.sqrt(normA2)
        normA2 = normA2 + alpha * alpha
        maxrbar = max(maxrbar, rhobarold)
        if itn > 1:
            minrbar = min(minrbar, rhobarold)
        condA = max(maxrbar, rhotemp) / min(minrbar, rhotemp)
        normar = abs(zetabar)
        normx = np.linalg.norm(x)
        test1 = normr / normb
        test2 = normar / (normA * normr) if (normA * normr) != 0 else np.inf
        test3 = 1 / condA
        t1 = test1 / (1 + normA * normx / normb)
        rtol = btol + atol * normA * normx / normb
        if itn >= maxiter:
            istop = 7
        if 1 + test3 <= 1:
            istop = 6
        if 1 + test2 <= 1:
            istop = 5
        if 1 + t1 <= 1:
            istop = 4
        if test3 <= ctol:
            istop = 3
        if test2 <= atol:
            istop = 2
        if test1 <= rtol:
            istop = 1
        if istop > 0:
            break
    return x, istop, itn, normr, normar


# ---- scipy.optimize.lsq_linear(method="trf", lsq_solver="lsmr") for a sparse A with finite or infinite bounds --------
# (scipy/optimize/_lsq/lsq_linear.py, trf_linear.py, common.py of scipy 1.15; the reference calls it at
# solver_linear_regression.py:258-269 with tol=1e-2, max_iter=200, lsmr_maxiter=1000, lsmr_tol="auto")
EPS = np.finfo(float).eps


def _in_bounds(x, lb, ub):
    return np.all((x >= lb) & (x <= ub))


def _reflective_transformation(y, lb, ub):
    if _in_bounds(y, lb, ub):
        return y, np.ones_like(y)
    lbf, ubf = np.isfinite(lb), np.isfinite(ub)
    x = y.copy()
    g_neg = np.zeros_like(y, dtype=bool)
    m = lbf & ~ubf
    x[m] = np.maximum(y[m], 2 * lb[m] - y[m])
    g_neg[m] = y[m] < lb[m]
    m = ~lbf & ubf
    x[m] = np.minimum(y[m], 2 * ub[m] - y[m])
    g_neg[m] = y[m] > ub[m]
    m = lbf & ubf
    d = ub - lb
    t = np.remainder(y[m] - lb[m], 2 * d[m])
    x[m] = lb[m] + np.minimum(t, 2 * d[m] - t)
    g_neg[m] = t > d[m]
    g = np.ones_like(y)
    g[g_neg] = -1
    return x, g


def _find_active(x, lb, ub, rtol=1e-10):
    active = np.zeros_like(x, dtype=int)
    if rtol == 0:
        active[x <= lb] = -1
        active[x >= ub] = 1
        return active
    lower, upper = x - lb, ub - x
    lt, ut = rtol * np.maximum(1, np.abs(lb)), rtol * np.maximum(1, np.abs(ub))
    active[np.isfinite(lb) & (lower <= np.minimum(upper, lt))] = -1
    active[np.isfinite(ub) & (upper <= np.minimum(lower, ut))] = 1
    return active


def _make_strictly_feasible(x, lb, ub, rstep=1e-10):
    xn = x.copy()
    active = _find_active(x, lb, ub, rstep)
    lo, up = active == -1, active == 1
    if rstep == 0:
        xn[lo] = np.nextafter(lb[lo], ub[lo])
        xn[up] = np.nextafter(ub[up], lb[up])
    else:
        xn[lo] = lb[lo] + rstep * np.maximum(1, np.abs(lb[lo]))
        xn[up] = ub[up] - rstep * np.maximum(1, np.abs(ub[up]))
    tight = (xn < lb) | (xn > ub)
    xn[tight] = 0.5 * (lb[tight] + ub[tight])
    return xn


def _cl_scaling(x, g, lb, ub):
    v, dv = np.ones_like(x), np.zeros_like(x)
    m = (g < 0) & np.isfinite(ub)
    v[m] = ub[m] - x[m]
    dv[m] = -1
    m = (g > 0) & np.isfinite(lb)
    v[m] = x[m] - lb[m]
    dv[m] = 1
    return v, dv


def _step_to_bound(x, s, lb, ub):
    nz = np.nonzero(s)
    steps = np.full_like(x, np.inf)
    with np.errstate(over="ignore"):
        steps[nz] = np.maximum((lb - x)[nz] / s[nz], (ub - x)[nz] / s[nz])
    mn = np.min(steps)
    return mn, np.equal(steps, mn) * np.sign(s).astype(int)


def _quad_1d(Jdot, g, s, diag=None, s0=None):
    v = Jdot(s)
    a = np.dot(v, v)
    if diag is not None:
        a += np.dot(s * diag, s)
    a *= 0.5
    b = np.dot(g, s)
    if s0 is None:
        return a, b
    u = Jdot(s0)
    b += np.dot(u, v)
    c = 0.5 * np.dot(u, u) + np.dot(g, s0)
    if diag is not None:
        b += np.dot(s0 * diag, s)
        c += 0.5 * np.dot(s0 * diag, s0)
    return a, b, c


def _min_quad_1d(a, b, lb, ub, c=0):
    t = [lb, ub]
    if a != 0:
        ext = -0.5 * b / a
        if lb < ext < ub:
            t.append(ext)
    t = np.asarray(t)
    y = t * (a * t + b) + c
    k = np.argmin(y)
    return t[k], y[k]


def _eval_quad(Jdot, g, s, diag=None):
    Js = Jdot(s)
    q = np.dot(Js, Js)
    if diag is not None:
        q += np.dot(s * diag, s)
    return 0.5 * q + np.dot(s, g)


class _Aug:
    """[A diag(d); diag(root)] as the operator lsmr needs (regularized_lsq_operator of right_multiplied_operator)."""

    def __init__(self, A, At, d, root):
        self.A, self.At, self.d, self.root = A, At, d, root
        self.shape = (A.shape[0] + A.shape[1], A.shape[1])
        self.T = _AugT(self)

    def __matmul__(self, x):
        return np.hstack((self.A @ (x * self.d), self.root * x))


class _AugT:
    def __init__(self, p):
        self.p = p

    def __matmul__(self, y):
        m = self.p.A.shape[0]
        return self.p.d * (self.p.At @ y[:m]) + self.p.root * y[m:]


def lsq_linear_trf(A, b, lb, ub, tol=1e-2, max_iter=200, lsmr_maxiter=1000):
    """lsq_linear(A, b, bounds=(lb, ub), tol=tol, max_iter=max_iter, lsmr_maxiter=lsmr_maxiter, lsmr_tol="auto")
    for sparse A: unconstrained LSMR first; if it violates the bounds, the trust-region-reflective iteration."""
    A = A.tocsr()
    At = A.T.tocsr()
    m, n = A.shape
    b = np.asarray(b, dtype=np.float64)
    lb = np.full(n, lb, dtype=np.float64) if np.ndim(lb) == 0 else np.asarray(lb, dtype=np.float64)
    ub = np.full(n, ub, dtype=np.float64) if np.ndim(ub) == 0 else np.asarray(ub, dtype=np.float64)
    x_lsq = lsmr(A, b, atol=1e-2 * tol, btol=1e-2 * tol, maxiter=lsmr_maxiter)[0]
    if _in_bounds(x_lsq, lb, ub):
        return x_lsq, 3, 0
    x, _ = _reflective_transformation(x_lsq, lb, ub)
    x = _make_strictly_feasible(x, lb, ub, rstep=0.1)
    r = A @ x - b
    g = At @ r
    cost = 0.5 * np.dot(r, r)
    status = None
    it = -1
    Adot = lambda s_: A @ s_  # noqa: E731
    for it in range(max_iter):
        v, dv = _cl_scaling(x, g, lb, ub)
        g_norm = np.linalg.norm(g * v, ord=np.inf)
        if g_norm < tol:
            status = 1
        if status is not None:
            break
        diag_h = g * dv
        root = diag_h ** 0.5
        d = v ** 0.5
        g_h = d * g
        Ahdot = lambda s_, d=d: A @ (s_ * d)  # noqa: E731
        aug = _Aug(A, At, d, root)
        r_aug = np.concatenate((r, np.zeros(n)))
        eta = 1e-2 * min(0.5, g_norm)
        ltol = max(EPS, min(0.1, eta * g_norm))
        p_h = -_lsmr_op(aug, r_aug, ltol, lsmr_maxiter)
        p = d * p_h
        p_dot_g = np.dot(p, g)
        if p_dot_g > 0:
            status = -1
        theta = 1 - min(0.005, g_norm)
        # ---- select_step
        if _in_bounds(x + p, lb, ub):
            step = p
        else:
            p_stride, hits = _step_to_bound(x, p, lb, ub)
            r_h = np.copy(p_h)
            r_h[hits.astype(bool)] *= -1
            rr = d * r_h
            p = p * p_stride
            p_h = p_h * p_stride
            x_on = x + p
            r_su, _ = _step_to_bound(x_on, rr, lb, ub)
            r_sl = (1 - theta) * r_su
            r_su *= theta
            if r_su > 0:
                a_, b_, c_ = _quad_1d(Ahdot, g_h, r_h, s0=p_h, diag=diag_h)
                r_stride, r_value = _min_quad_1d(a_, b_, r_sl, r_su, c=c_)
                r_h = p_h + r_h * r_stride
                rr = d * r_h
            else:
                r_value = np.inf
            p_h = p_h * theta
            p = p * theta
            p_value = _eval_quad(Ahdot, g_h, p_h, diag=diag_h)
            ag_h = -g_h
            ag = d * ag_h
            ag_su, _ = _step_to_bound(x, ag, lb, ub)
            ag_su *= theta
            a_, b_ = _quad_1d(Ahdot, g_h, ag_h, diag=diag_h)
            ag_stride, ag_value = _min_quad_1d(a_, b_, 0, ag_su)
            ag = ag * ag_stride
            if p_value < r_value and p_value < ag_value:
                step = p
            elif r_value < p_value and r_value < ag_value:
                step = rr
            else:
                step = ag
        cost_change = -_eval_quad(Adot, g, step)
        if cost_change < 0:  # backtracking
            alpha = 1.0
            while True:
                x_new, _ = _reflective_transformation(x + alpha * p, lb, ub)
                step = x_new - x
                cost_change = -_eval_quad(Adot, g, step)
                if cost_change > -0.1 * alpha * p_dot_g:
                    break
                alpha *= 0.5
            if np.any(_find_active(x_new, lb, ub) != 0):
                x_new, _ = _reflective_transformation(x + theta * alpha * p, lb, ub)
                x_new = _make_strictly_feasible(x_new, lb, ub, rstep=0)
                step = x_new - x
                cost_change = -_eval_quad(Adot, g, step)
            # (scipy returns the OLD x from backtracking(): `return x, step, cost_change`)
        else:
            x = _make_strictly_feasible(x + step, lb, ub, rstep=0)
        r = A @ x - b
        g = At @ r
        if cost_change < tol * cost:
            status = 2
        cost = 0.5 * np.dot(r, r)
    if status is None:
        status = 0
    return x, status, it + 1


def _lsmr_op(op, b, tol, maxiter):
    """lsmr for an operator object with `@` and `.T @` (the augmented system of the trust-region step)."""

    class _W:
        shape = op.shape

        def tocsr(self):
            return self

        def __matmul__(self, x):
            return op @ x

        @property
        def T(self):
            return _WT()

    class _WT:
        def tocsr(self):
            return self

        def __matmul__(self, y):
            return op.T @ y

    return lsmr(_W(), b, atol=tol, btol=tol, maxiter=maxiter)[0]


def cosine_similarity(a, b):
    norm = np.linalg.norm(a) * np.linalg.norm(b)
    return 0 if norm == 0 else np.sum(a * b) / norm


def lsq_reconstruct(projection_image, scale2d_to_3d, twist_degree, rise_pixel, csym=1, tilt_degree=0, psi_degree=0,
                    dy_pixel=0, thresh_fraction=-1, positive_constraint=-1, reconstruct_diameter_3d_inner_pixel=0,
                    reconstruct_diameter_2d_pixel=-1, reconstruct_diameter_3d_pixel=-1, reconstruct_length_2d_pixel=-1,
                    reconstruct_length_3d_pixel=-1, sym_oversample=1, interpolation="nn", return_parts=False, fsc_test=0):
    """``algorithm=dict(model="lsq")``, ``fsc_test=0``, ``score_metric="cosine"``: ((rec3d, None, None), score)."""
    rmin = reconstruct_diameter_3d_inner_pixel / 2
    rmax = reconstruct_diameter_3d_pixel // 2 - 1
    mask = get_cylindrical_mask(reconstruct_length_3d_pixel, reconstruct_diameter_3d_pixel, reconstruct_diameter_3d_pixel,
                                rmin=rmin, rmax=rmax)
    mz, my, mx = mask.shape
    n3 = int(np.count_nonzero(mask))
    n2 = reconstruct_diameter_2d_pixel * reconstruct_length_2d_pixel
    target = min(2**26, int(max(n2, n3) * sym_oversample))
    A_data, b_data, b_pid = build_A_data_matrix(projection_image, scale2d_to_3d, twist_degree, rise_pixel, csym,
                                                tilt_degree, psi_degree, dy_pixel, reconstruct_diameter_2d_pixel,
                                                reconstruct_length_2d_pixel, reconstruct_diameter_3d_pixel,
                                                reconstruct_diameter_3d_inner_pixel, reconstruct_length_3d_pixel,
                                                target, interpolation)
    A_hsym, b_hsym = build_A_helical_sym_matrix(mz, my, mx, twist_degree, rise_pixel, csym, rmin, rmax, target,
                                                interpolation)
    if A_hsym is not None:
        A = vstack((A_data, A_hsym)).tocsr()
        b = np.concatenate((b_data, b_hsym))
    else:
        A, b = A_data, b_data
    pitch_pixel = round(rise_pixel * 360 / abs(twist_degree))
    positive = positive_constraint > 0 or (positive_constraint < 0 and pitch_pixel > round(reconstruct_length_3d_pixel * 2))
    lb, ub = (0.0, float(np.max(b_data))) if positive else (-np.inf, np.inf)   # solver:245-256
    x = lsq_linear_trf(A, b, lb, ub, tol=1e-2, max_iter=200, lsmr_maxiter=1000)[0].astype(np.float32)
    pred = A_data.dot(x)
    if thresh_fraction >= 0:
        pred = np.clip(pred, 0, None)
    score = cosine_similarity(pred, b_data)
    rec3d = np.zeros(mask.shape, dtype=np.float32)
    rec3d[mask] = x
    if fsc_test and fsc_test >= 1:
        # split_A_b (solver:175-203), deterministic modes: the rows of every second pixel id (2), of the lower half of the
        # ids (3), of the outer thirds (4+) against the rest; each half is solved with the same symmetry block and its
        # own upper bound; score = s0 / 2 + (s1 + s2) / 4 (solver:526-529)
        ids = sorted(set(b_pid.tolist()))
        n = len(ids)
        if fsc_test == 1:   # solver:186-189: the ids in the set's own iteration order, shuffled by the global RNG
            ids = list(set(b_pid))
            np.random.shuffle(ids)
            set1 = ids[: n // 2]
        else:
            set1 = ids[::2] if fsc_test == 2 else ids[: n // 3] + ids[n * 2 // 3:] if fsc_test >= 4 else ids[: n // 2]
        is1 = np.isin(b_pid, set1)
        halves, scores = [], [score]
        for sel in (is1, ~is1):
            Ah, bh = A_data[sel], b_data[sel]
            Af = vstack((Ah, A_hsym)).tocsr() if A_hsym is not None else Ah
            bf = np.concatenate((bh, b_hsym)) if A_hsym is not None else bh
            lbh, ubh = (0.0, float(np.max(bh))) if positive else (-np.inf, np.inf)
            xh = lsq_linear_trf(Af, bf, lbh, ubh, tol=1e-2, max_iter=200, lsmr_maxiter=1000)[0].astype(np.float32)
            ph = Ah.dot(xh)
            if thresh_fraction >= 0:
                ph = np.clip(ph, 0, None)
            scores.append(cosine_similarity(ph, bh))
            rh = np.zeros(mask.shape, dtype=np.float32)
            rh[mask] = xh
            halves.append(rh)
        return (rec3d, halves[0], halves[1]), scores[0] / 2 + (scores[1] + scores[2]) / 4
    if return_parts:
        return (rec3d, None, None), score, dict(A_data=A_data, b_data=b_data, b_pid=b_pid, A_hsym=A_hsym, x=x, mask=mask)
    return (rec3d, None, None), score
