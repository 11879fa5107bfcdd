"""Path A on the GPU: ``lsq_reconstruct`` with the reference's signature
(src/helicon/webApps/denovo3D/solver_linear_regression.py:31-56) for ``interpolation="nn"`` / ``"linear"`` and
``algorithm["model"]`` in lsq, elasticnet (the reference app's default), lasso, ridge, lreg.

What runs where:

* libhelicon_hip.so, ``hh_pab_*`` (csrc/path_a_batch.inc, path_a_linear.inc, path_a_factored.inc): K candidates of one image
  set up and solved together on the device — the implicit system (the reference's ``build_A_data_matrix`` /
  ``build_A_helical_sym_matrix``, never materialised as matrices), LSMR and the whole trust-region-reflective loop of
  ``scipy.optimize.lsq_linear`` (model "lsq"), or the accelerated proximal-gradient minimiser of the scikit-learn models'
  common objective (``hh_pab_solve_prox``), the float32 map, the prediction and the cosine score.  ``lsq_reconstruct`` is
  a batch of one (three with half sets);
* ``hh_pa_*`` (csrc/path_a.inc): the single-candidate projector, kept for trilinear candidates with tilt / psi (rays that
  cross cell layers) — products on the device, the trust-region glue of this module on the host — and as the parity
  reference of the group solver's products;
* ``refine_tilt_psi_dy`` (:550-841, switched on by ``refine_tilt_psi_dy_range``, :384-439): the Gauss-Newton loop of the
  reference with every matrix an ``hh_pa`` operator on the device and SciPy's own ``lsq_linear`` / ``lsqr`` driving them;
* this module: argument handling, the positivity rule (:352-355), grouping, half sets, the volume assembly (:532-547).

* the non-cosine scores of :484-524 ("ssim", "ms_ssim", "mutual_information", "composite"; ``helicon.ssim_score`` etc.,
  lib/analysis.py:487-613 — scikit-image in the reference, pinned by derivation like the image preparation): ``ssim_score``,
  ``ms_ssim_score``, ``mutual_information_score`` on the device (``hh_ssim_2d``, ``hh_joint_histogram``).

Not provided (``NotImplementedError``): model "ard".  There is no CPU fallback: without the library or a GPU the call raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

__all__ = ["lsq_reconstruct", "lsq_reconstruct_batch", "PathAProblem", "PathABatch", "get_cylindrical_mask", "cosine_similarity"]

EPS = np.finfo(float).eps


hh_pa_params = _lib.hh_pa_params
_f64p = C.POINTER(C.c_double)


def _p(a):
    return None if a is None else a.ctypes.data_as(_f64p)


def get_cylindrical_mask(nz, ny, nx, rmin=0, rmax=-1, return_xyz=False):
    """The reference's cylinder (lib/analysis.py:731-774) — voxels whose centred (y, x) offset lies strictly inside radius
    ``rmax`` (default ``ny // 2 - 1``) and, when ``0 < rmin < rmax``, on or outside ``rmin`` — as one 2-D disk repeated
    along z.  The C-order rank of this mask is the order of the unknowns (the device ranks the same disk in
    ``pa_mask_rank``)."""
    if rmax < 0:
        rmax = ny // 2 - 1
    oy, ox = np.ogrid[-(ny // 2): ny - ny // 2, -(nx // 2): nx - nx // 2]
    radius2 = oy.astype(np.int64) ** 2 + ox.astype(np.int64) ** 2
    disk = radius2 < rmax * rmax
    if 0 < rmin < rmax:
        disk &= radius2 >= rmin * rmin
    mask = np.repeat(disk[None, :, :], nz, axis=0)
    if not return_xyz:
        return mask
    offsets = [np.arange(-(n // 2), n - n // 2, dtype=np.int32) for n in (nz, ny, nx)]
    return mask, tuple(np.meshgrid(*offsets, indexing="ij"))


def cosine_similarity(a, b):
    """lib/analysis.py:802-821 on host vectors (the prediction A_data x comes back from the device)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    norm = np.sqrt(_dot(a, a)) * np.sqrt(_dot(b, b))
    return 0 if norm == 0 else _dot(a, b) / norm


def _f32_pair(img1, img2):
    a, b = np.asarray(img1), np.asarray(img2)
    if a.shape != b.shape:
        raise ValueError(f"Image shapes must match: {a.shape} vs {b.shape}")   # lib/analysis.py:504-505
    return np.ascontiguousarray(a, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)


def _data_range(a, b):
    return max(a.max() - a.min(), b.max() - b.min())


def _ssim(a, b, data_range, device):
    out = C.c_double()
    _lib.check(_lib.lib().hh_ssim_2d(int(device), a.ctypes.data_as(C.POINTER(C.c_float)), b.ctypes.data_as(C.POINTER(C.c_float)),
                                     a.shape[0], a.shape[1], float(data_range), C.byref(out)), None)
    return float(out.value)


def ssim_score(img1, img2, *, device=0):
    """``helicon.ssim_score`` (lib/analysis.py:487-513): scikit-image's structural similarity with its defaults (7 x 7 uniform
    window) at ``data_range`` = the larger of the two images' ranges, on the device (``hh_ssim_2d``); 0 for constant images
    and for images the window does not fit (the reference swallows scikit-image's error and returns 0)."""
    a, b = _f32_pair(img1, img2)
    r = _data_range(a, b)
    if r == 0 or a.ndim != 2 or min(a.shape) < 7:
        return 0.0
    return _ssim(a, b, r, device)


def ms_ssim_score(img1, img2, *, device=0):
    """``helicon.ms_ssim_score`` (lib/analysis.py:516-582): SSIM at up to five scales, each half the last (scikit-image's
    ``rescale(img, 0.5, anti_aliasing=True)``: ``helicon_amd.rescale`` with linear interpolation), combined as a weighted
    geometric mean."""
    from .denovo3D import rescale

    a, b = _f32_pair(img1, img2)
    r = _data_range(a, b)
    if r == 0 or a.ndim != 2:
        return 0.0
    weights = np.array([0.0448, 0.2856, 0.3001, 0.2363, 0.1333])
    vals = []
    for i in range(len(weights)):
        if a.shape[0] < 8 or a.shape[1] < 8:
            break
        vals.append(max(_ssim(a, b, r, device), 0.0))
        if i < len(weights) - 1:
            a, b = rescale(a, 0.5, order=1, device=device), rescale(b, 0.5, order=1, device=device)
            r = _data_range(a, b)
            if r == 0:
                break
    if not vals:
        return 0.0
    wts = weights[: len(vals)] / weights[: len(vals)].sum()
    result = 1.0
    for s_, w_ in zip(vals, wts):
        result *= s_ ** w_
    return float(result)


def mutual_information_score(img1, img2, *, device=0):
    """``helicon.mutual_information_score`` (lib/analysis.py:585-613): scikit-image's normalised mutual information with 64
    bins, minus 1 — (H(a) + H(b)) / H(a, b) - 1 from the joint histogram, counted on the device (``hh_joint_histogram``; the
    edges are ``np.histogramdd``'s: ``np.linspace(min, max, 65)`` per image)."""
    a, b = _f32_pair(img1, img2)
    bins = 64

    def edges(v):
        # np.histogramdd's own edges: np.linspace of the image's float32 extremes — float32 edges under NumPy 2's promotion
        # rules — handed to the device as the float64 values they are
        lo, hi = v.min(), v.max()
        if lo == hi:   # np.histogramdd widens an empty range by one half either side
            lo, hi = lo - 0.5, hi + 0.5
        return np.linspace(lo, hi, bins + 1)

    ea32, eb32 = edges(a), edges(b)
    ea, eb = np.ascontiguousarray(ea32, dtype=np.float64), np.ascontiguousarray(eb32, dtype=np.float64)
    counts = np.zeros((bins, bins), dtype=np.int64)
    _lib.check(_lib.lib().hh_joint_histogram(int(device), a.ctypes.data_as(C.POINTER(C.c_float)), b.ctypes.data_as(C.POINTER(C.c_float)),
                                             a.size, _p(ea), _p(eb), bins, counts.ctypes.data_as(C.POINTER(C.c_int64))), None)
    # density=True, as scikit-image asks for it: counts over the bins' own widths (np.diff of those edges: not all equal in their
    # last float32 digits) and over the total — the widths do not cancel exactly in the entropies
    hist = counts.astype(np.float64)
    total = hist.sum()
    hist = hist / np.diff(ea32).reshape(bins, 1)
    hist = hist / np.diff(eb32).reshape(1, bins)
    hist = hist / total

    def entropy(counts):   # scipy.stats.entropy: normalise, -sum p log p over p > 0
        pk = counts.astype(np.float64).ravel()
        pk = pk / pk.sum()
        pk = pk[pk > 0]
        return float(-(pk * np.log(pk)).sum())

    h01 = entropy(hist)
    if h01 == 0:   # (scikit-image divides by zero here and the reference returns what comes out: nan - 1)
        return float("nan")
    return (entropy(hist.sum(axis=0)) + entropy(hist.sum(axis=1))) / h01 - 1.0


_SCORES_2D = ("ssim", "ms_ssim", "mutual_information", "composite")


def _score_2d(metric, pred, b_data, pid, img, d2, l2, device):
    """The 2-D scores of solver:484-524: the prediction scattered to its pixels of the (L2d, D2d) region against the
    transposed input region."""
    pred_2d = np.zeros((l2, d2), dtype=np.float32)
    pred_2d.ravel()[pid] = pred
    ny, nx = img.shape
    ref_2d = np.ascontiguousarray(img[ny // 2 - d2 // 2: ny // 2 + d2 // 2, nx // 2 - l2 // 2: nx // 2 + l2 // 2].T, dtype=np.float32)
    if metric == "ssim":
        return ssim_score(pred_2d, ref_2d, device=device)
    if metric == "ms_ssim":
        return ms_ssim_score(pred_2d, ref_2d, device=device)
    if metric == "mutual_information":
        return mutual_information_score(pred_2d, ref_2d, device=device)
    parts = [cosine_similarity(pred, b_data), ssim_score(pred_2d, ref_2d, device=device), ms_ssim_score(pred_2d, ref_2d, device=device),
             mutual_information_score(pred_2d, ref_2d, device=device)]
    return float(np.mean(parts))


class PathAProblem:
    """One candidate's implicit least-squares system on the device (``hh_pa``)."""

    def __init__(self, image, *, scale2d_to_3d, twist_degree, rise_pixel, csym, tilt_degree, psi_degree, dy_pixel,
                 reconstruct_diameter_2d_pixel, reconstruct_length_2d_pixel, reconstruct_diameter_3d_pixel,
                 reconstruct_diameter_3d_inner_pixel, reconstruct_length_3d_pixel, min_projection_lines, min_sym_pairs,
                 interpolation="nn", fsc_mode=0, fsc_half=0, device=0):
        self._L = _lib.lib()
        img = np.ascontiguousarray(image, dtype=np.float32)
        if img.ndim != 2:
            raise ValueError("projection_image must be 2-D")
        q = hh_pa_params(float(scale2d_to_3d), float(twist_degree), float(rise_pixel), int(csym), float(tilt_degree),
                         float(psi_degree), float(dy_pixel), int(reconstruct_diameter_2d_pixel),
                         int(reconstruct_length_2d_pixel), int(reconstruct_diameter_3d_pixel),
                         int(reconstruct_diameter_3d_inner_pixel), int(reconstruct_length_3d_pixel),
                         int(min_projection_lines), int(min_sym_pairs), {"nn": 0, "linear": 1}[interpolation],
                         int(fsc_mode), int(fsc_half))
        self._h = C.c_void_p()
        rc = self._L.hh_pa_create(C.byref(self._h), int(device), img.ctypes.data_as(C.POINTER(C.c_float)), img.shape[0],
                                  img.shape[1], C.byref(q))
        if rc:
            msg = self._L.hh_pa_last_error(None)
            kind = ValueError if rc == -1 else _lib.HeliconHipError
            raise kind(f"hh_pa_create error {rc}: {msg.decode() if msg else '?'}")
        dims = (C.c_int64 * 4)()
        self._L.hh_pa_dims(self._h, dims)
        self.n, self.m_data, self.m_sym, self.n_ops = (int(v) for v in dims)
        self.m = self.m_data + self.m_sym
        self.b_data = np.empty(self.m_data, dtype=np.float32)
        self.b_pid = np.empty(self.m_data, dtype=np.int32)
        self._L.hh_pa_get_rhs(self._h, self.b_data.ctypes.data_as(C.POINTER(C.c_float)),
                              self.b_pid.ctypes.data_as(C.POINTER(C.c_int32)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.hh_pa_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            msg = self._L.hh_pa_last_error(self._h)
            raise _lib.HeliconHipError(f"libhelicon_hip (Path A) error {rc}: {msg.decode() if msg else '?'}")

    def matvec(self, x, d=None, root=None):
        """[A diag(d); diag(root)] x — A = [A_data; A_hsym]."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        d = None if d is None else np.ascontiguousarray(d, dtype=np.float64)
        root = None if root is None else np.ascontiguousarray(root, dtype=np.float64)
        y = np.empty(self.m + (self.n if root is not None else 0), dtype=np.float64)
        self._check(self._L.hh_pa_matvec(self._h, _p(x), _p(d), _p(root), _p(y)))
        return y

    def rmatvec(self, y, d=None, root=None):
        y = np.ascontiguousarray(y, dtype=np.float64)
        d = None if d is None else np.ascontiguousarray(d, dtype=np.float64)
        root = None if root is None else np.ascontiguousarray(root, dtype=np.float64)
        g = np.empty(self.n, dtype=np.float64)
        self._check(self._L.hh_pa_rmatvec(self._h, _p(y), _p(d), _p(root), _p(g)))
        return g

    def lsmr(self, rhs, d=None, root=None, atol=1e-6, btol=1e-6, conlim=1e8, maxiter=1000):
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        d = None if d is None else np.ascontiguousarray(d, dtype=np.float64)
        root = None if root is None else np.ascontiguousarray(root, dtype=np.float64)
        x = np.empty(self.n, dtype=np.float64)
        info = (C.c_int * 2)()
        norms = (C.c_double * 2)()
        self._check(self._L.hh_pa_lsmr(self._h, _p(rhs), _p(d), _p(root), float(atol), float(btol), float(conlim),
                                       int(maxiter), _p(x), info, norms))
        return x, int(info[0]), int(info[1]), float(norms[0]), float(norms[1])


class PathABatch:
    """K candidates of one image and one reconstruction box, set up and solved together on the device (``hh_pab``):
    what the reference's thread pool does one ``process_one_task`` at a time (app.py:2473-2476)."""

    def __init__(self, image, params, device=0):
        self._L = _lib.lib()
        img = np.ascontiguousarray(image, dtype=np.float32)
        if img.ndim != 2:
            raise ValueError("projection_image must be 2-D")
        self.count = len(params)
        arr = (hh_pa_params * self.count)(*params)
        self._h = C.c_void_p()
        rc = self._L.hh_pab_create(C.byref(self._h), int(device), img.ctypes.data_as(C.POINTER(C.c_float)), img.shape[0],
                                   img.shape[1], arr, self.count)
        if rc:
            msg = self._L.hh_pab_last_error(None)
            kind = ValueError if rc == -1 else _lib.HeliconHipError
            raise kind(f"hh_pab_create error {rc}: {msg.decode() if msg else '?'}")
        dims = (C.c_int64 * 5)()
        rows = np.empty((self.count, 3), dtype=np.int64)
        self._L.hh_pab_dims(self._h, dims, rows.ctypes.data_as(C.POINTER(C.c_int64)))
        self.n = int(dims[1])
        self.device_bytes = int(dims[4])
        self.m_data, self.m_sym, self.n_ops = rows[:, 0].copy(), rows[:, 1].copy(), rows[:, 2].copy()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.hh_pab_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def rhs(self, c):
        b = np.empty(int(self.m_data[c]), dtype=np.float32)
        pid = np.empty(int(self.m_data[c]), dtype=np.int32)
        self._L.hh_pab_get_rhs(self._h, int(c), b.ctypes.data_as(C.POINTER(C.c_float)), pid.ctypes.data_as(C.POINTER(C.c_int32)))
        return b, pid

    def sym_pairs(self, c):
        """Symmetry rows of candidate c as voxel-rank pairs, in the reference's row order."""
        out = np.empty((int(self.m_sym[c]), 2), dtype=np.int32)
        rc = self._L.hh_pab_get_pairs(self._h, int(c), out.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc:
            raise _lib.HeliconHipError(f"hh_pab_get_pairs error {rc}")
        return out

    def solve(self, positive, clip, tol=1e-2, max_iter=200, lsmr_maxiter=1000, want_x=True):
        """``lsq_linear`` + cosine score for every candidate: (x float32 [K, n] or None, scores [K], info [K, 5])."""
        pos = np.ascontiguousarray(np.broadcast_to(np.asarray(positive, dtype=np.int32), (self.count,)))
        clp = np.ascontiguousarray(np.broadcast_to(np.asarray(clip, dtype=np.int32), (self.count,)))
        x = np.empty((self.count, self.n), dtype=np.float32) if want_x else None
        scores = np.empty(self.count, dtype=np.float64)
        info = np.empty((self.count, 5), dtype=np.int32)
        i32 = C.POINTER(C.c_int32)
        rc = self._L.hh_pab_solve(self._h, pos.ctypes.data_as(i32), clp.ctypes.data_as(i32), float(tol), int(max_iter),
                                  int(lsmr_maxiter), None if x is None else x.ctypes.data_as(C.POINTER(C.c_float)),
                                  scores.ctypes.data_as(_f64p), info.ctypes.data_as(i32))
        if rc:
            msg = self._L.hh_pab_last_error(self._h)
            raise _lib.HeliconHipError(f"libhelicon_hip (Path A batch) error {rc}: {msg.decode() if msg else '?'}")
        return x, scores, info

    def solve_prox(self, positive, clip, alpha, l1_ratio, ridge_form=False, tol=1e-7, max_iter=5000, want_x=True):
        """The scikit-learn models of solve_equations (solver:270-342) as the minimiser of their common objective
        (``hh_pab_solve_prox``): (x float32 [K, n] or None, scores [K], info [K, 3] = iterations / converged / non-zeros,
        objective [K])."""
        pos = np.ascontiguousarray(np.broadcast_to(np.asarray(positive, dtype=np.int32), (self.count,)))
        clp = np.ascontiguousarray(np.broadcast_to(np.asarray(clip, dtype=np.int32), (self.count,)))
        al = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha, dtype=np.float64), (self.count,)))
        x = np.empty((self.count, self.n), dtype=np.float32) if want_x else None
        scores = np.empty(self.count, dtype=np.float64)
        info = np.empty((self.count, 3), dtype=np.int32)
        obj = np.empty(self.count, dtype=np.float64)
        i32 = C.POINTER(C.c_int32)
        self._check(self._L.hh_pab_solve_prox(self._h, pos.ctypes.data_as(i32), clp.ctypes.data_as(i32), _p(al), float(l1_ratio),
                                              1 if ridge_form else 0, float(tol), int(max_iter),
                                              None if x is None else x.ctypes.data_as(C.POINTER(C.c_float)), _p(scores),
                                              info.ctypes.data_as(i32), _p(obj)))
        return x, scores, info, obj

    def matvec(self, c, x):
        """A_c x through the solve's own product kernels (rows in the reference's order)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(int(self.m_data[c] + self.m_sym[c]), dtype=np.float64)
        self._check(self._L.hh_pab_matvec(self._h, int(c), _p(x), _p(y)))
        return y

    def rmatvec(self, c, y):
        y = np.ascontiguousarray(y, dtype=np.float64)
        g = np.empty(self.n, dtype=np.float64)
        self._check(self._L.hh_pab_rmatvec(self._h, int(c), _p(y), _p(g)))
        return g

    def _check(self, rc):
        if rc:
            msg = self._L.hh_pab_last_error(self._h)
            raise _lib.HeliconHipError(f"libhelicon_hip (Path A batch) error {rc}: {msg.decode() if msg else '?'}")

    def counters(self):
        out = (C.c_int64 * 4)()
        self._L.hh_pab_counters(self._h, out)
        return {"launches": int(out[0]), "host_syncs": int(out[1]), "lsmr_iterations_queued": int(out[2]),
                "self_check_failures": int(out[3])}


# ---- scipy.optimize.lsq_linear(method="trf", lsq_solver="lsmr", lsmr_tol="auto") around the device operator ----------
def _dot(a, b):
    """Inner product as an index-ordered pairwise sum (``np.add.reduce``), not BLAS: one thread, one summation order.
    ``np.dot`` hands these 50k-element vectors to OpenBLAS, whose worker threads (one per host core: 256 on the GPU
    box, under a 16-core quota) made a 0.1 s solve take 0.4 - 0.7 s, and whose blocking makes the rounding depend on
    the thread count — which the loosely converged trust-region iteration amplifies to 1e-3 in the score."""
    return float(np.add.reduce(a * b))


def _in_bounds(x, lb, ub):
    return np.all((x >= lb) & (x <= ub))


def _reflect(y, lb, ub):  # common.py: reflective_transformation (finite bounds on both sides, or none)
    if _in_bounds(y, lb, ub):
        return y
    d = ub - lb
    t = np.remainder(y - lb, 2 * d)
    return lb + np.minimum(t, 2 * d - t)


def _active(x, lb, ub, rtol=1e-10):
    act = np.zeros_like(x, dtype=int)
    if rtol == 0:
        act[x <= lb] = -1
        act[x >= ub] = 1
        return act
    lower, upper = x - lb, ub - x
    lt, ut = rtol * np.maximum(1, np.abs(lb)), rtol * np.maximum(1, np.abs(ub))
    act[lower <= np.minimum(upper, lt)] = -1
    act[upper <= np.minimum(lower, ut)] = 1
    return act


def _strictly_feasible(x, lb, ub, rstep=1e-10):
    xn = x.copy()
    act = _active(x, lb, ub, rstep)
    lo, up = act == -1, act == 1
    if rstep == 0:
        xn[lo] = np.nextafter(lb[lo], ub[lo])
        xn[up] = np.nextafter(ub[up], lb[up])
    else:
        xn[lo] = lb[lo] + rstep * np.maximum(1, np.abs(lb[lo]))
        xn[up] = ub[up] - rstep * np.maximum(1, np.abs(ub[up]))
    tight = (xn < lb) | (xn > ub)
    xn[tight] = 0.5 * (lb[tight] + ub[tight])
    return xn


def _step_to_bound(x, s, lb, ub):
    nz = np.nonzero(s)
    steps = np.full_like(x, np.inf)
    with np.errstate(over="ignore"):
        steps[nz] = np.maximum((lb - x)[nz] / s[nz], (ub - x)[nz] / s[nz])
    mn = np.min(steps)
    return mn, np.equal(steps, mn) * np.sign(s).astype(int)


def _quad_1d(jdot, g, s, diag=None, s0=None):
    v = jdot(s)
    a = _dot(v, v)
    if diag is not None:
        a += _dot(s * diag, s)
    a *= 0.5
    b = _dot(g, s)
    if s0 is None:
        return a, b
    u = jdot(s0)
    b += _dot(u, v)
    c = 0.5 * _dot(u, u) + _dot(g, s0)
    if diag is not None:
        b += _dot(s0 * diag, s)
        c += 0.5 * _dot(s0 * diag, s0)
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


def _eval_quad(jdot, g, s, diag=None):
    js = jdot(s)
    q = _dot(js, js)
    if diag is not None:
        q += _dot(s * diag, s)
    return 0.5 * q + _dot(s, g)


def _solve_bounded(P: PathAProblem, b, lb, ub, tol=1e-2, max_iter=200, lsmr_maxiter=1000):
    """lsq_linear(A, b, bounds=(lb, ub), tol, max_iter, lsmr_maxiter, lsmr_tol="auto"): unconstrained LSMR first; if it
    leaves the box, the trust-region-reflective iteration (every LSMR solve and every product with A on the device)."""
    n = P.n
    x_lsq = P.lsmr(b, atol=1e-2 * tol, btol=1e-2 * tol, maxiter=lsmr_maxiter)[0]
    if not np.isfinite(lb) and not np.isfinite(ub):
        return x_lsq, 3, 0
    lb = np.full(n, lb, dtype=np.float64)
    ub = np.full(n, ub, dtype=np.float64)
    if _in_bounds(x_lsq, lb, ub):
        return x_lsq, 3, 0
    x = _strictly_feasible(_reflect(x_lsq, lb, ub), lb, ub, rstep=0.1)
    r = P.matvec(x) - b
    g = P.rmatvec(r)
    cost = 0.5 * _dot(r, r)
    status, it = None, -1
    adot = P.matvec
    for it in range(max_iter):
        v, dv = np.ones(n), np.zeros(n)          # CL_scaling_vector
        mk = g < 0
        v[mk] = ub[mk] - x[mk]
        dv[mk] = -1
        mk = g > 0
        v[mk] = x[mk] - lb[mk]
        dv[mk] = 1
        g_norm = np.linalg.norm(g * v, ord=np.inf)
        if g_norm < tol:
            status = 1
            break
        diag_h = g * dv
        root = diag_h ** 0.5
        d = v ** 0.5
        g_h = d * g
        ahdot = lambda s_, d=d: P.matvec(s_, d=d)  # noqa: E731
        eta = 1e-2 * min(0.5, g_norm)
        ltol = max(EPS, min(0.1, eta * g_norm))
        p_h = -P.lsmr(np.concatenate((r, np.zeros(n))), d=d, root=root, atol=ltol, btol=ltol, maxiter=lsmr_maxiter)[0]
        p = d * p_h
        p_dot_g = _dot(p, g)
        if p_dot_g > 0:
            status = -1
        theta = 1 - min(0.005, g_norm)
        if _in_bounds(x + p, lb, ub):            # select_step
            step = p
        else:
            p_stride, hits = _step_to_bound(x, p, lb, ub)
            r_h = np.copy(p_h)
            r_h[hits.astype(bool)] *= -1
            rr = d * r_h
            p = p * p_stride
            p_h = p_h * p_stride
            r_su, _ = _step_to_bound(x + p, rr, lb, ub)
            r_sl = (1 - theta) * r_su
            r_su *= theta
            if r_su > 0:
                a_, b_, c_ = _quad_1d(ahdot, g_h, r_h, s0=p_h, diag=diag_h)
                r_stride, r_value = _min_quad_1d(a_, b_, r_sl, r_su, c=c_)
                r_h = p_h + r_h * r_stride
                rr = d * r_h
            else:
                r_value = np.inf
            p_h = p_h * theta
            p = p * theta
            p_value = _eval_quad(ahdot, g_h, p_h, diag=diag_h)
            ag_h = -g_h
            ag = d * ag_h
            ag_su, _ = _step_to_bound(x, ag, lb, ub)
            ag_su *= theta
            a_, b_ = _quad_1d(ahdot, g_h, ag_h, diag=diag_h)
            ag_stride, ag_value = _min_quad_1d(a_, b_, 0, ag_su)
            ag = ag * ag_stride
            if p_value < r_value and p_value < ag_value:
                step = p
            elif r_value < p_value and r_value < ag_value:
                step = rr
            else:
                step = ag
        cost_change = -_eval_quad(adot, g, step)
        if cost_change < 0:                      # backtracking (scipy keeps the old x here)
            alpha = 1.0
            while True:
                x_new = _reflect(x + alpha * p, lb, ub)
                step = x_new - x
                cost_change = -_eval_quad(adot, g, step)
                if cost_change > -0.1 * alpha * p_dot_g:
                    break
                alpha *= 0.5
            if np.any(_active(x_new, lb, ub) != 0):
                x_new = _strictly_feasible(_reflect(x + theta * alpha * p, lb, ub), lb, ub, rstep=0)
                step = x_new - x
                cost_change = -_eval_quad(adot, g, step)
        else:
            x = _strictly_feasible(x + step, lb, ub, rstep=0)
        r = P.matvec(x) - b
        g = P.rmatvec(r)
        if cost_change < tol * cost:
            status = 2
        cost = 0.5 * _dot(r, r)
        if status is not None:
            break
    return x, (0 if status is None else status), it + 1


def _box(img, reconstruct_diameter_2d_pixel, reconstruct_length_2d_pixel, reconstruct_diameter_3d_pixel,
         reconstruct_length_3d_pixel, reconstruct_diameter_3d_inner_pixel, sym_oversample):
    """The integer geometry of lsq_reconstruct's first lines (solver:116-173): sizes, the cylinder and the row target."""
    d3, l3 = int(reconstruct_diameter_3d_pixel), int(reconstruct_length_3d_pixel)
    d2 = int(reconstruct_diameter_2d_pixel) if reconstruct_diameter_2d_pixel > 0 else img.shape[0]
    l2 = int(reconstruct_length_2d_pixel) if reconstruct_length_2d_pixel > 0 else img.shape[1]
    mask = get_cylindrical_mask(l3, d3, d3, rmin=reconstruct_diameter_3d_inner_pixel / 2, rmax=d3 // 2 - 1)
    n3 = int(np.count_nonzero(mask))
    target = min(2**26, int(max(d2 * l2, n3) * sym_oversample))                   # solver:148-150, 168-170
    return d2, l2, d3, l3, mask, n3, target


def _model_of(algorithm):
    """algorithm dict of lsq_reconstruct -> None for "lsq", else (alpha, l1_ratio, ridge_form) of the common objective
    (solver:270-342: the defaults of the reference's constructors)."""
    algorithm = algorithm or {}
    model = algorithm.get("model", "lsq")
    if model == "lsq":
        return None
    if model == "elasticnet":
        return float(algorithm.get("alpha", 1e-4)), float(algorithm.get("l1_ratio", 0.5)), False
    if model == "lasso":
        return float(algorithm.get("alpha", 1e-4)), 1.0, False
    if model == "ridge":
        return float(algorithm.get("alpha", 1)), 0.0, True
    if model == "lreg":
        return 0.0, 0.0, False
    raise NotImplementedError(f"algorithm model {model!r}: the GPU path has lsq, elasticnet, lasso, ridge and lreg (ard is a dense "
                              "Bayesian fit in the reference)")


def lsq_reconstruct_batch(projection_image, scale2d_to_3d, candidates, tilt_degree=0, psi_degree=0, dy_pixel=0,
                          thresh_fraction=-1, positive_constraint=-1, reconstruct_diameter_3d_inner_pixel=0,
                          reconstruct_diameter_2d_pixel=-1, reconstruct_diameter_3d_pixel=-1, reconstruct_length_2d_pixel=-1,
                          reconstruct_length_3d_pixel=-1, sym_oversample=1, fsc_test=0, *, interpolation="nn", return_3d=True,
                          device=0, batch=128, streams=8, stats=None, algorithm=None, score_metric="cosine"):
    """``lsq_reconstruct`` (solver_linear_regression.py:31-547; nearest-neighbour projector, model "lsq", cosine score)
    for MANY (twist_degree, rise_pixel, csym) candidates of one image: the loop the reference's driver runs as a thread
    pool over ``process_one_task`` (app.py:2473-2476).  ``candidates`` is a sequence of ``(twist, rise, csym)``; the
    other arguments have the meaning and defaults of ``lsq_reconstruct``.

    Up to ``batch`` candidates form one device-resident group (``hh_pab_*``): every LSMR iteration and every
    trust-region step is one launch for the whole group, whose candidates wait for each other at the trust-region
    steps.  Up to ``streams`` groups run at once, each from its own host thread on its own HIP stream, so one group's
    stragglers overlap with the other groups' full launches.  A candidate's result does not depend on how the list is
    cut.  Returns ``[((rec3d, half1, half2), score), ...]`` in the order of ``candidates`` (maps are ``None`` with
    ``return_3d=False``).  ``stats``, if a dict, receives launch / synchronisation counters.

    ``interpolation="linear"`` (the app's default, app.py:577-585) has no group solver yet: the candidates then go one
    by one through ``lsq_reconstruct`` (``hh_pa``: products on the device, trust-region glue on the host) from
    ``streams`` threads — the same results in the same order, at that path's rate."""
    img = np.asarray(projection_image)
    cands = [(float(t), float(r), int(c)) for t, r, c in candidates]
    if interpolation not in ("nn", "linear"):
        raise ValueError("interpolation must be 'nn' or 'linear'")
    model = _model_of(algorithm)
    if model is not None and (tilt_degree != 0 or psi_degree != 0):
        raise NotImplementedError("the scikit-learn models run on the slice-major products: tilt = psi = 0")

    def one_by_one():
        """The single-candidate path (hh_pa) from a thread pool: trilinear candidates the group solver cannot slice."""
        from concurrent.futures import ThreadPoolExecutor

        def one(c):
            maps, score = lsq_reconstruct(img, scale2d_to_3d, c[0], c[1], c[2], tilt_degree, psi_degree, dy_pixel, thresh_fraction,
                                          positive_constraint, reconstruct_diameter_3d_inner_pixel, reconstruct_diameter_2d_pixel,
                                          reconstruct_diameter_3d_pixel, reconstruct_length_2d_pixel, reconstruct_length_3d_pixel,
                                          sym_oversample, interpolation, fsc_test, score_metric=score_metric, device=device, _single=True)
            return (maps if return_3d else (None, None, None)), score

        with ThreadPoolExecutor(max_workers=max(1, min(int(streams), len(cands)))) as pool:
            res = list(pool.map(one, cands))
        if stats is not None:
            stats.update(groups=len(cands), launches=0, host_syncs=0, lsmr_iterations_queued=0, self_check_failures=0, info=[], path="hh_pa")
        return res

    if interpolation == "linear" and (tilt_degree != 0 or psi_degree != 0) and model is None:
        if fsc_test == 1:
            raise NotImplementedError("fsc_test=1 with trilinear interpolation needs tilt = psi = 0 (the group solver)")
        return one_by_one()
    d2, l2, d3, l3, mask, n3, target = _box(img, reconstruct_diameter_2d_pixel, reconstruct_length_2d_pixel,
                                            reconstruct_diameter_3d_pixel, reconstruct_length_3d_pixel,
                                            reconstruct_diameter_3d_inner_pixel, sym_oversample)
    random_split = fsc_test == 1     # solver:186-189: halves from np.random.shuffle of the pixel ids (global RNG)
    halves = (0, 1, 2) if fsc_test and fsc_test > 1 else (0,)
    per = 3 if fsc_test and fsc_test >= 1 else 1
    step = max(1, int(batch) // per)
    if random_split:
        streams = 1                  # the draws follow the order of the list
    # groups of equal size rather than full groups and a remainder.  A short list is NOT spread thin over the streams: the
    # device runs the groups' launches mostly one after the other whatever the stream count (Path A), so 100
    # candidates take 0.25 s as two groups of 50 and 0.33 s as eight groups of 12
    n_groups = max(1, -(-len(cands) // step), min(int(streams), len(cands) // 48))
    bounds = [round(k * len(cands) / n_groups) for k in range(n_groups + 1)]

    def params_of(tw, rs, cs, mode, half, ids=None):
        q = hh_pa_params(float(scale2d_to_3d), tw, rs, cs, float(tilt_degree), float(psi_degree), float(dy_pixel), d2, l2, d3,
                         int(reconstruct_diameter_3d_inner_pixel), l3, int(target), int(target), 1 if interpolation == "linear" else 0,
                         int(mode) if half else 0, half)
        if ids is not None:
            q.n_fsc_ids = len(ids)
            q.fsc_ids = ids.ctypes.data_as(C.POINTER(C.c_int32))
        return q

    def solve_batch(params, positive):
        with PathABatch(img, params, device=device) as B:
            if B.n != n3:
                raise ValueError("the cylinder does not fit the 2-D region's box (reconstruct_diameter_2d_pixel must hold "
                                 "the 3-D diameter): the reference's two masks would rank the voxels differently")
            clip = 1 if thresh_fraction >= 0 else 0
            two_d = score_metric in _SCORES_2D
            want_x = return_3d or two_d
            if model is None:
                x, scores, info = B.solve(positive, clip, want_x=want_x)
            else:
                alpha, rho, ridge_form = model
                al = np.full(len(params), alpha, dtype=np.float64)
                x, scores, info3, _ = B.solve_prox(positive, clip, al, rho, ridge_form, want_x=want_x)
                # solver:331-338: an all-zero solution is refitted with alpha / 10 (elasticnet, lasso, ridge) until it is not
                for _ in range(12):
                    zero = info3[:, 2] == 0
                    if not zero.any() or alpha == 0:
                        break
                    al = np.where(zero, al * 0.1, al)
                    x2, s2, i2, _ = B.solve_prox(positive, clip, al, rho, ridge_form, want_x=want_x)
                    scores = np.where(zero, s2, scores)
                    info3 = np.where(zero[:, None], i2, info3)
                    if want_x:
                        x = np.where(zero[:, None], x2, x)
                info = np.concatenate([np.where(info3[:, 1:2] == 1, 1, 0), info3[:, :1], np.zeros_like(info3[:, :1]), info3[:, :1],
                                       np.zeros_like(info3[:, :1])], axis=1)   # status, iterations, -, iterations, -
            if two_d:
                # solver:484-524: the prediction A_data x (x as the float32 map), clipped with thresh_fraction >= 0, scattered to
                # its pixels and scored against the input region
                scores = np.array(scores, dtype=np.float64)
                for c in range(len(params)):
                    b_c, pid_c = B.rhs(c)
                    pred = B.matvec(c, x[c].astype(np.float64))[: len(b_c)].astype(np.float32)
                    if clip:
                        pred = np.clip(pred, 0, None)
                    scores[c] = _score_2d(score_metric, pred, b_c, pid_c, img, d2, l2, device)
            pids = [B.rhs(c)[1] for c in range(len(params))] if random_split else None
            counters = dict(B.counters(), device_bytes=B.device_bytes, info=info.tolist())
        return x, scores, counters, pids

    def run_group(g):
        chunk = cands[bounds[g]: bounds[g + 1]]
        params, positive = [], []
        for tw, rs, cs in chunk:
            pitch_pixel = round(rs * 360 / abs(tw))
            pos = positive_constraint > 0 or (positive_constraint < 0 and pitch_pixel > round(l3 * 2))    # solver:352-355
            for half in halves:
                params.append(params_of(tw, rs, cs, fsc_test, half))
                positive.append(1 if pos else 0)
        x, scores, counters, pids = solve_batch(params, positive)
        if random_split:
            # the whole-image solves above, then the two halves of every candidate's pixels: list(set(ids)) in the set's own
            # order, shuffled by the global NumPy RNG, first half of the shuffled ids against the rest
            keep, params2, positive2 = [], [], []
            for k, (tw, rs, cs) in enumerate(chunk):
                ids = list(set(pids[k]))
                np.random.shuffle(ids)
                first = np.ascontiguousarray(ids[: len(ids) // 2], dtype=np.int32)
                keep.append(first)
                for half in (1, 2):
                    params2.append(params_of(tw, rs, cs, 1, half, first))
                    positive2.append(positive[k])
            x2, scores2, counters2, _ = solve_batch(params2, positive2)
            for key in ("launches", "host_syncs", "lsmr_iterations_queued", "self_check_failures"):
                counters[key] += counters2[key]
            counters["info"] += counters2["info"]
            # interleave: (full, half 1, half 2) per candidate
            scores = np.stack([scores, scores2[0::2], scores2[1::2]], axis=1).reshape(-1)
            if return_3d:
                x = np.stack([x, x2[0::2], x2[1::2]], axis=1).reshape(-1, x.shape[1])
        res = []
        for k in range(len(chunk)):
            sc = scores[k * per: (k + 1) * per]
            score = float(sc[0] / 2 + (sc[1] + sc[2]) / 4 if per == 3 else sc[0])                          # solver:526-529
            maps = [None, None, None]
            if return_3d:
                for h in range(per):
                    rec = np.zeros(mask.shape, dtype=np.float32)
                    rec[mask] = x[k * per + h]
                    maps[h] = rec
            res.append(((maps[0], maps[1], maps[2]), score))
        return res, counters

    try:
        if n_groups == 1 or streams <= 1:
            done = [run_group(g) for g in range(n_groups)]
        else:
            from concurrent.futures import ThreadPoolExecutor

            with ThreadPoolExecutor(max_workers=min(int(streams), n_groups)) as pool:
                done = list(pool.map(run_group, range(n_groups)))
    except ValueError as e:
        # a trilinear ray whose samples straddle cell layers (or a cylinder whose two planes do not fit the LDS): the
        # group solver has no general form of those products, the single-candidate path has
        if interpolation == "linear" and "not sliceable" in str(e) and fsc_test != 1:
            return one_by_one()
        raise
    out = []
    for res, counters in done:
        out.extend(res)
        if stats is not None:
            for k in ("launches", "host_syncs", "lsmr_iterations_queued", "self_check_failures"):
                stats[k] = stats.get(k, 0) + counters[k]
            stats["device_bytes"] = stats.get("device_bytes", 0) + counters["device_bytes"]
            stats.setdefault("info", []).extend(counters["info"])
    if stats is not None:
        stats["groups"] = n_groups
        stats["path"] = "hh_pab"
    return out


def refine_tilt_psi_dy(projection_image, scale2d_to_3d, twist_degree, rise_pixel, csym, reconstruct_diameter_2d_pixel,
                       reconstruct_length_2d_pixel, reconstruct_diameter_3d_pixel, reconstruct_diameter_3d_inner_pixel,
                       reconstruct_length_3d_pixel, sym_oversample, interpolation, x_init, tilt_0=0.0, psi_0=0.0, dy_0=0.0,
                       delta_tilt=0.5, delta_psi=1.0, delta_dy=0.2, max_iter=5, tol_tilt=0.05, tol_psi=0.1, tol_dy=0.05,
                       bounds_tilt=(-30.0, 30.0), bounds_psi=(-45.0, 45.0), bounds_dy=(-5.0, 5.0), positive_constraint=-1,
                       algorithm=None, verbose=0, cpu=1, *, device=0):
    """solver_linear_regression.py:550-841: Gauss-Newton refinement of (tilt, psi, dy) with a finite-difference Jacobian of
    the predicted projection.  Returns ``(tilt, psi, dy, x, cosine score)``.

    Every matrix of the reference is an implicit operator on the device here (``hh_pa``, one per geometry: the general
    projector with tilt and psi); what drives them is the reference's own driver — ``scipy.optimize.lsq_linear(bounds=(0,
    max b), max_iter=200)`` under the positivity rule, ``scipy.sparse.linalg.lsqr(atol=btol=1e-6)`` otherwise — called on a
    ``LinearOperator`` whose products run on the device, so tolerances and stopping rules are SciPy's, not a restatement.
    ``x_init`` is accepted and ignored, like the reference (it solves the base system afresh).  The reference pairs every new
    geometry's matrix with the FIRST geometry's right-hand side and so fails in SciPy's shape check as soon as a perturbed
    geometry gains or loses a ray; the same condition raises ``ValueError`` here."""
    from scipy.optimize import lsq_linear
    from scipy.sparse.linalg import LinearOperator, lsqr

    img = np.asarray(projection_image)
    d2, l2 = int(reconstruct_diameter_2d_pixel), int(reconstruct_length_2d_pixel)
    d3, l3 = int(reconstruct_diameter_3d_pixel), int(reconstruct_length_3d_pixel)
    t = np.array([tilt_0, psi_0, dy_0], dtype=np.float64)
    deltas = np.array([delta_tilt, delta_psi, delta_dy], dtype=np.float64)
    lo = np.array([bounds_tilt[0], bounds_psi[0], bounds_dy[0]], dtype=np.float64)
    hi = np.array([bounds_tilt[1], bounds_psi[1], bounds_dy[1]], dtype=np.float64)
    target = min(2**26, int(max(d2 * l2, d3 * d3 * l3) * sym_oversample))   # solver:636-643, 663-665: the BOX's voxels here
    pitch_pixel = round(rise_pixel * 360 / abs(twist_degree))
    positive = positive_constraint > 0 or (positive_constraint < 0 and pitch_pixel > round(l3 * 2))

    def problem(tt):
        return PathAProblem(img, scale2d_to_3d=scale2d_to_3d, twist_degree=twist_degree, rise_pixel=rise_pixel, csym=csym,
                            tilt_degree=float(tt[0]), psi_degree=float(tt[1]), dy_pixel=float(tt[2]),
                            reconstruct_diameter_2d_pixel=d2, reconstruct_length_2d_pixel=l2, reconstruct_diameter_3d_pixel=d3,
                            reconstruct_diameter_3d_inner_pixel=reconstruct_diameter_3d_inner_pixel,
                            reconstruct_length_3d_pixel=l3, min_projection_lines=target, min_sym_pairs=target,
                            interpolation=interpolation, device=device)

    def solve(P, b_data):   # _solve_system (solver:697-716)
        if P.m_data != len(b_data):
            raise ValueError("Inconsistent shapes between `A` and `b`: the perturbed geometry has "
                             f"{P.m_data} rays, the first one {len(b_data)} (the reference fails the same way)")
        op = LinearOperator((P.m, P.n), matvec=P.matvec, rmatvec=P.rmatvec, dtype=np.float64)
        b = np.concatenate((b_data.astype(np.float64), np.zeros(P.m_sym)))
        if positive:
            return lsq_linear(op, b, bounds=(0.0, float(np.max(b_data))), max_iter=200).x
        return lsqr(op, b, atol=1e-6, btol=1e-6)[0]

    with problem(t) as P0:
        b_data = P0.b_data.copy()
        x_cur = solve(P0, b_data)
        p_0 = P0.matvec(x_cur)[: P0.m_data]
    n_base = len(b_data)
    b64 = b_data.astype(np.float64)
    for _it in range(int(max_iter)):
        J = np.zeros((n_base, 3), dtype=np.float64)
        for i in range(3):
            t_pert = t.copy()
            t_pert[i] = np.clip(t_pert[i] + deltas[i], lo[i], hi[i])
            actual = t_pert[i] - t[i]
            if abs(actual) > 1e-12:
                with problem(t_pert) as Pp:
                    p_pert = Pp.matvec(x_cur)[: Pp.m_data]
                nc = min(n_base, len(p_pert))
                J[:nc, i] = (p_pert[:nc] - p_0[:nc]) / actual
        r_0 = p_0 - b64
        G = J.T @ J
        g = J.T @ r_0
        cond = np.linalg.cond(G) if np.linalg.det(G) != 0 else float("inf")
        if cond > 1e10:
            G = G + 1e-6 * np.diag(np.diag(G))
        try:
            delta_t = np.linalg.solve(G, -g)
        except np.linalg.LinAlgError:
            break
        t_new = np.clip(t + delta_t, lo, hi)
        step = t_new - t
        t = t_new
        if abs(step[0]) < tol_tilt and abs(step[1]) < tol_psi and abs(step[2]) < tol_dy:
            break
        with problem(t) as Pn:
            x_cur = solve(Pn, b_data)
            p_0 = Pn.matvec(x_cur)[: Pn.m_data]
    return float(t[0]), float(t[1]), float(t[2]), x_cur, cosine_similarity(p_0, b64)


def lsq_reconstruct(projection_image, scale2d_to_3d, twist_degree, rise_pixel, csym=1, tilt_degree=0, psi_degree=0,
                    dy_pixel=0, thresh_fraction=-1, positive_constraint=-1, reconstruct_diameter_3d_inner_pixel=0,
                    reconstruct_diameter_2d_pixel=-1, reconstruct_diameter_3d_pixel=-1, reconstruct_length_2d_pixel=-1,
                    reconstruct_length_3d_pixel=-1, sym_oversample=1, interpolation="nn", fsc_test=0,
                    score_metric="cosine", target_apix2d=5.0, verbose=0, algorithm=dict(model="lsq"),
                    refine_tilt_psi_dy_range=None, cpu=1, *, device=0, _single=False):
    """solver_linear_regression.py:31-547 for ``interpolation`` "nn" or "linear", ``algorithm["model"] == "lsq"``,
    cosine score: returns ``((rec3d float32 (L3d, D3d, D3d), None, None), score)``, or with ``fsc_test`` 2, 3 or 4 the
    maps of the two pixel halves as well and ``score = s_full / 2 + (s_half1 + s_half2) / 4``.

    The solve runs in float64 throughout.  With "linear" the reference's first LSMR call runs in float32 (its matrix is
    float32 and NumPy 2 keeps that type), so its score is reproducible to about 1e-3 only — the tolerance of the parity
    tests against its fixture; against the float64 oracle the agreement is that of "nn"."""
    if interpolation not in ("nn", "linear"):
        raise ValueError("interpolation must be 'nn' or 'linear'")
    model = _model_of(algorithm)
    if score_metric not in ("cosine", "frc") + _SCORES_2D:  # "frc" is documented but never dispatched: it falls through to cosine
        score_metric = "cosine"                              # (anything else falls through to cosine too, solver:523-524)
    if score_metric in _SCORES_2D and fsc_test and fsc_test >= 1:
        # solver:499-501 scatters a HALF set's prediction with the FULL set's pixel ids: NumPy refuses the assignment
        raise ValueError("shape mismatch: a half set's prediction cannot be scattered with the full set's pixel ids "
                         "(the reference fails the same way for 2-D scores with fsc_test >= 1)")
    refine = refine_tilt_psi_dy_range is not None and any(v > 0 for v in (refine_tilt_psi_dy_range.get(k, 0) for k in ("tilt", "psi", "dy")))
    img = np.asarray(projection_image)
    if _single and model is not None:
        raise NotImplementedError("the scikit-learn models run in the group solver")
    if refine:
        return _lsq_reconstruct_refined(img, scale2d_to_3d, twist_degree, rise_pixel, csym, tilt_degree, psi_degree, dy_pixel,
                                        thresh_fraction, positive_constraint, reconstruct_diameter_3d_inner_pixel,
                                        reconstruct_diameter_2d_pixel, reconstruct_diameter_3d_pixel, reconstruct_length_2d_pixel,
                                        reconstruct_length_3d_pixel, sym_oversample, interpolation, fsc_test, algorithm,
                                        refine_tilt_psi_dy_range, device)
    if interpolation == "nn" or not _single:   # the device-resident solver (a batch of one, or of three with half sets)
        return lsq_reconstruct_batch(img, scale2d_to_3d, [(twist_degree, rise_pixel, csym)], tilt_degree, psi_degree, dy_pixel,
                                     thresh_fraction, positive_constraint, reconstruct_diameter_3d_inner_pixel,
                                     reconstruct_diameter_2d_pixel, reconstruct_diameter_3d_pixel, reconstruct_length_2d_pixel,
                                     reconstruct_length_3d_pixel, sym_oversample, fsc_test, interpolation=interpolation, device=device,
                                     algorithm=algorithm, score_metric=score_metric)[0]
    d3, l3 = int(reconstruct_diameter_3d_pixel), int(reconstruct_length_3d_pixel)
    d2 = int(reconstruct_diameter_2d_pixel) if reconstruct_diameter_2d_pixel > 0 else img.shape[0]
    l2 = int(reconstruct_length_2d_pixel) if reconstruct_length_2d_pixel > 0 else img.shape[1]
    rmin = reconstruct_diameter_3d_inner_pixel / 2
    rmax = d3 // 2 - 1
    mask = get_cylindrical_mask(l3, d3, d3, rmin=rmin, rmax=rmax)
    n3 = int(np.count_nonzero(mask))
    target = min(2**26, int(max(d2 * l2, n3) * sym_oversample))                   # solver:148-150, 168-170
    pitch_pixel = round(rise_pixel * 360 / abs(twist_degree))
    positive = positive_constraint > 0 or (positive_constraint < 0 and pitch_pixel > round(l3 * 2))   # solver:352-355
    xs, scores = [], []
    # the whole image, then (fsc_test >= 2: solver:448-482) the two halves of its pixels, each with the same symmetry block
    for half in ((0, 1, 2) if fsc_test and fsc_test > 1 else (0,)):
        with PathAProblem(img, scale2d_to_3d=scale2d_to_3d, twist_degree=twist_degree, rise_pixel=rise_pixel, csym=csym,
                          tilt_degree=tilt_degree, psi_degree=psi_degree, dy_pixel=dy_pixel,
                          reconstruct_diameter_2d_pixel=d2, reconstruct_length_2d_pixel=l2, reconstruct_diameter_3d_pixel=d3,
                          reconstruct_diameter_3d_inner_pixel=reconstruct_diameter_3d_inner_pixel,
                          reconstruct_length_3d_pixel=l3, min_projection_lines=target, min_sym_pairs=target,
                          interpolation=interpolation, fsc_mode=int(fsc_test) if half else 0, fsc_half=half, device=device) as P:
            if P.n != n3:
                raise ValueError("the cylinder does not fit the 2-D region's box (reconstruct_diameter_2d_pixel must hold "
                                 "the 3-D diameter): the reference's two masks would rank the voxels differently")
            b = np.concatenate((P.b_data.astype(np.float64), np.zeros(P.m_sym)))
            lb, ub = (0.0, float(np.max(P.b_data))) if positive else (-np.inf, np.inf)                    # solver:245-256
            x, _, _ = _solve_bounded(P, b, lb, ub, tol=1e-2, max_iter=200, lsmr_maxiter=1000)
            x = x.astype(np.float32)
            pred = P.matvec(x.astype(np.float64))[: P.m_data]
            if thresh_fraction >= 0:
                pred = np.clip(pred, 0, None)
            xs.append(x)
            if score_metric in _SCORES_2D:
                scores.append(_score_2d(score_metric, pred.astype(np.float32), P.b_data, P.b_pid, img, d2, l2, device))
            else:
                scores.append(cosine_similarity(pred, P.b_data.astype(np.float64)))
    score = scores[0] / 2 + (scores[1] + scores[2]) / 4 if len(scores) == 3 else scores[0]                # solver:526-529
    maps = []
    for x in xs:
        rec = np.zeros(mask.shape, dtype=np.float32)
        rec[mask] = x
        maps.append(rec)
    if len(maps) == 3:
        return (maps[0], maps[1], maps[2]), score
    rec3d = maps[0]
    return (rec3d, None, None), score



def _lsq_reconstruct_refined(img, scale2d_to_3d, twist_degree, rise_pixel, csym, tilt_degree, psi_degree, dy_pixel, thresh_fraction,
                             positive_constraint, d3_inner, d2, d3, l2, l3, sym_oversample, interpolation, fsc_test, algorithm,
                             r_range, device):
    """lsq_reconstruct with ``refine_tilt_psi_dy_range`` (solver:384-439): the ordinary solve, then the refinement started
    from tilt = psi = dy = 0 (as the reference starts it, whatever the task's own angles), whose map and score replace the
    solve's whenever they exist (the solve's own score is ``None`` at that point for every model, solver:330-342 with
    ``train_fraction = 1``).  With half sets the three scores are recomputed from the ORIGINAL geometry's rows (solver:484-524):
    ``A_data x_refined`` for the full set.  The refined parameters are kept in ``lsq_reconstruct._refined_params`` for the
    caller's projections (pipeline.py:419-428 reads and clears them)."""
    d2 = int(d2) if d2 > 0 else img.shape[0]
    l2 = int(l2) if l2 > 0 else img.shape[1]
    (rec, h1, h2), _score = lsq_reconstruct_batch(img, scale2d_to_3d, [(twist_degree, rise_pixel, csym)], tilt_degree, psi_degree, dy_pixel,
                                                 thresh_fraction, positive_constraint, d3_inner, d2, d3, l2, l3, sym_oversample, fsc_test,
                                                 interpolation=interpolation, device=device, algorithm=algorithm)[0]
    tilt, psi, dy, x, score_refined = refine_tilt_psi_dy(
        img, scale2d_to_3d, twist_degree, rise_pixel, csym, d2, l2, d3, d3_inner, l3, sym_oversample, interpolation, None,
        delta_tilt=r_range.get("delta_tilt", 0.5), delta_psi=r_range.get("delta_psi", 1.0), delta_dy=r_range.get("delta_dy", 0.2),
        max_iter=r_range.get("max_iter", 5), bounds_tilt=(-r_range.get("tilt", 30.0), r_range.get("tilt", 30.0)),
        bounds_psi=(-r_range.get("psi", 45.0), r_range.get("psi", 45.0)), bounds_dy=(-r_range.get("dy", 5.0), r_range.get("dy", 5.0)),
        positive_constraint=positive_constraint, algorithm=algorithm, device=device)
    lsq_reconstruct._refined_params = {"tilt": tilt, "psi": psi, "dy": dy}
    mask = get_cylindrical_mask(int(l3), int(d3), int(d3), rmin=d3_inner / 2, rmax=int(d3) // 2 - 1)
    rec = np.zeros(mask.shape, dtype=np.float32)
    rec[mask] = x.astype(np.float32)
    if not (fsc_test and fsc_test >= 1):
        return (rec, None, None), score_refined
    # half sets: every score again from the rows of the task's own geometry
    scores = []
    target = min(2**26, int(max(d2 * l2, int(np.count_nonzero(mask))) * sym_oversample))
    for half, vol in ((0, rec), (1, h1), (2, h2)):
        with PathAProblem(img, scale2d_to_3d=scale2d_to_3d, twist_degree=twist_degree, rise_pixel=rise_pixel, csym=csym,
                          tilt_degree=tilt_degree, psi_degree=psi_degree, dy_pixel=dy_pixel, reconstruct_diameter_2d_pixel=d2,
                          reconstruct_length_2d_pixel=l2, reconstruct_diameter_3d_pixel=int(d3), reconstruct_diameter_3d_inner_pixel=d3_inner,
                          reconstruct_length_3d_pixel=int(l3), min_projection_lines=target, min_sym_pairs=target, interpolation=interpolation,
                          fsc_mode=int(fsc_test) if half else 0, fsc_half=half, device=device) as P:
            pred = P.matvec(vol[mask].astype(np.float64))[: P.m_data]
            if thresh_fraction >= 0:
                pred = np.clip(pred, 0, None)
            scores.append(cosine_similarity(pred, P.b_data.astype(np.float64)))
    return (rec, h1, h2), scores[0] / 2 + (scores[1] + scores[2]) / 4
