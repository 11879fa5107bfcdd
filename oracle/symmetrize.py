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
