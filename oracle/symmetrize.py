"""CPU oracle for ``apply_helical_symmetry`` (reference: src/helicon/lib/transforms.py:58-165).

TEST INFRASTRUCTURE ONLY (see oracle/path_b.py).  The reference function is a numba kernel (pure
Python loops without numba); this restatement keeps its arithmetic — float64 coordinates and
weights, float32 accumulation in the reference's (hi, k, ci) order, the ``(i - nx / 2)`` float vs
``(j - ny // 2)`` integer centre quirk (transforms.py:117-121), the z-range taken from the 1 %
profile threshold (transforms.py:92-99) — and vectorises only the two innermost (j, i) loops.
Pinned by tests/golden/g7_helical_sym.npz, generated from the reference itself.
"""
import numpy as np


def apply_helical_symmetry(data, apix, twist_degree, rise_angstrom, csym=1, fraction=1.0, new_size=None,
                           new_apix=None, cpu=1):
    if new_apix is None:
        new_apix = apix
    nz0, ny0, nx0 = data.shape
    if new_size is None:  # the reference cannot unpack None (transforms.py:78-79); treat it as "same size"
        new_size = data.shape
    new_size = tuple(int(v) for v in new_size)
    if new_size != data.shape:
        nz1, ny1, nx1 = new_size
        nz2, ny2, nx2 = max(nz0, nz1), max(ny0, ny1), max(nx0, nx1)
        data_work = np.zeros((nz2, ny2, nx2), dtype=np.float32)
    else:
        data_work = np.zeros((nz0, ny0, nx0), dtype=np.float32)
    nz, ny, nx = data_work.shape
    w = np.zeros((nz, ny, nx), dtype=np.float32)

    hsym_max = max(1, int(nz * new_apix / rise_angstrom))
    profile_z = np.sum(np.sum(data, axis=-1), axis=-1)
    threshold = 0.01 * np.max(profile_z)
    non_zero_indices = np.where(profile_z > threshold)[0]
    z0 = non_zero_indices[0]
    z1 = non_zero_indices[-1]
    zmid = (z0 + z1) // 2 + (z0 + z1) % 2
    z0 = max(z0, zmid - int(nz0 * fraction + 0.5) // 2)
    z1 = min(z1, zmid + int(nz0 * fraction + 0.5) // 2)

    jj = (np.arange(ny) - ny // 2).astype(np.float64)[:, None]
    ii = (np.arange(nx) - nx / 2)[None, :]
    for hi in range(-hsym_max, hsym_max + 1):
        for k in range(nz):
            k2 = ((k - nz // 2) * new_apix + hi * rise_angstrom) / apix + nz0 // 2
            if k2 < z0 or k2 >= z1:
                continue
            kf, kc = int(np.floor(k2)), int(np.ceil(k2))
            wk = k2 - kf
            for ci in range(csym):
                rot = np.deg2rad(twist_degree * hi + 360 * ci / csym)
                c, s = np.cos(rot), np.sin(rot)
                j2 = (c * jj + s * ii) * new_apix / apix + ny0 // 2
                i2 = (-s * jj + c * ii) * new_apix / apix + nx0 // 2
                jf, jc = np.floor(j2).astype(np.int64), np.ceil(j2).astype(np.int64)
                jf_, ic_ = jf, None
                i_f, i_c = np.floor(i2).astype(np.int64), np.ceil(i2).astype(np.int64)
                ok = (jf >= 0) & (jf < ny0 - 1) & (i_f >= 0) & (i_f < nx0 - 1)
                if not ok.any():
                    continue
                jf, jc, i_f, i_c = jf[ok], jc[ok], i_f[ok], i_c[ok]
                wj, wi = j2[ok] - jf, i2[ok] - i_f
                d0, d1 = data[kf], data[kc]
                val = ((1 - wk) * (1 - wj) * (1 - wi) * d0[jf, i_f]
                       + (1 - wk) * (1 - wj) * wi * d0[jf, i_c]
                       + (1 - wk) * wj * (1 - wi) * d0[jc, i_f]
                       + (1 - wk) * wj * wi * d0[jc, i_c]
                       + wk * (1 - wj) * (1 - wi) * d1[jf, i_f]
                       + wk * (1 - wj) * wi * d1[jf, i_c]
                       + wk * wj * (1 - wi) * d1[jc, i_f]
                       + wk * wj * wi * d1[jc, i_c])
                plane = data_work[k]
                plane[ok] = (plane[ok].astype(np.float64) + val).astype(np.float32)
                w[k][ok] += np.float32(1.0)
    mask = w > 0
    data_work = np.where(mask, data_work / np.where(mask, w, 1), data_work)
    if data_work.shape != new_size:
        nz1, ny1, nx1 = new_size
        data_work = data_work[nz // 2 - nz1 // 2: nz // 2 + nz1 // 2,
                              ny // 2 - ny1 // 2: ny // 2 + ny1 // 2,
                              nx // 2 - nx1 // 2: nx // 2 + nx1 // 2]
    return data_work


# ---- helicon.transform_map (lib/transforms.py:168-235) ------------------------------------------------------------------
def _spline_prefilter_axis(c, axis):
    """scipy.ndimage.spline_filter1d(order=3, mode="constant"): the cubic B-spline's one pole sqrt(3) - 2, causal and
    anti-causal recursions with MIRROR initialisation (ni_splines.c treats "constant" like "mirror")."""
    z = np.sqrt(3.0) - 2.0
    c = np.moveaxis(np.array(c, dtype=np.float64), axis, 0)
    n = c.shape[0]
    if n < 2:
        return np.moveaxis(c, 0, axis)
    c *= (1.0 - z) * (1.0 - 1.0 / z)
    zn1 = z ** (n - 1)
    c0 = c[0] + zn1 * c[n - 1]
    zi = z
    for i in range(1, n - 1):
        c0 = c0 + zi * (c[i] + zn1 * c[n - 1 - i])
        zi *= z
    c[0] = c0 / (1.0 - zn1 * zn1)
    for i in range(1, n):
        c[i] += z * c[i - 1]
    c[n - 1] = (z * c[n - 2] + c[n - 1]) * z / (z * z - 1.0)
    for i in range(n - 2, -1, -1):
        c[i] = z * (c[i + 1] - c[i])
    return np.moveaxis(c, 0, axis)


def _mirror(idx, n):
    if n <= 1:
        return np.zeros_like(idx)
    s2 = 2 * n - 2
    idx = np.abs(idx) % s2
    return np.where(idx >= n, s2 - idx, idx)


def map_coordinates_cubic(data, coords):
    """scipy.ndimage.map_coordinates(data, coords, order=3) with its defaults (mode="constant", cval=0, prefilter):
    a point outside [0, n - 1] on any axis gives 0; inside, 4 x 4 x 4 cubic B-spline taps with mirrored indices."""
    c = np.asarray(data, dtype=np.float64)
    for ax in range(c.ndim):
        c = _spline_prefilter_axis(c, ax)
    coords = np.asarray(coords, dtype=np.float64)
    shape = c.shape
    inside = np.ones(coords.shape[1], dtype=bool)
    for d in range(3):
        inside &= (coords[d] >= 0) & (coords[d] <= shape[d] - 1)
    out = np.zeros(coords.shape[1], dtype=np.float64)
    cc = coords[:, inside]
    idx, w = [], []
    for d in range(3):
        f = np.floor(cc[d])
        y = cc[d] - f
        zc = 1.0 - y
        w1 = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0
        w2 = (zc * zc * (zc - 2.0) * 3.0 + 4.0) / 6.0
        w0 = zc * zc * zc / 6.0
        w3 = 1.0 - w0 - w1 - w2
        w.append((w0, w1, w2, w3))
        start = f.astype(np.int64) - 1
        idx.append([_mirror(start + t, shape[d]) for t in range(4)])
    acc = np.zeros(cc.shape[1], dtype=np.float64)
    for a in range(4):
        for b in range(4):
            for e in range(4):
                acc += c[idx[0][a], idx[1][b], idx[2][e]] * (w[0][a] * w[1][b] * w[2][e])
    out[inside] = acc
    return out.astype(np.asarray(data).dtype if np.asarray(data).dtype.kind == "f" else np.float64)


def transform_map(data, scale=1.0, rot=0, tilt=0, psi=0, dx=0, dy=0, dz=0):
    """lib/transforms.py:168-235: rotate (intrinsic ZYZ Euler angles: Rz(rot) Ry(tilt) Rz(psi)), scale and shift the
    sampling grid about the volume's centre voxel, resample with the cubic spline above."""
    if scale == 1 and rot == 0 and tilt == 0 and psi == 0 and dx == 0 and dy == 0 and dz == 0:
        return data
    nz, ny, nx = data.shape
    Z, Y, X = np.meshgrid(np.arange(nz, dtype=np.int32) - nz // 2, np.arange(ny, dtype=np.int32) - ny // 2,
                          np.arange(nx, dtype=np.int32) - nx // 2, indexing="ij")
    if scale != 1.0:
        Z, Y, X = Z * scale, Y * scale, X * scale
    xyz = np.vstack((X.ravel(), Y.ravel(), Z.ravel())).astype(np.float64)

    def rz(a):
        c, s = np.cos(np.deg2rad(a)), np.sin(np.deg2rad(a))
        return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])

    def ry(a):
        c, s = np.cos(np.deg2rad(a)), np.sin(np.deg2rad(a))
        return np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])

    m = rz(rot) @ ry(tilt) @ rz(psi)
    p = m @ xyz
    p[0] += nx // 2 - dx
    p[1] += ny // 2 - dy
    p[2] += nz // 2 - dz
    return map_coordinates_cubic(data, p[[2, 1, 0]]).reshape((nz, ny, nx))
