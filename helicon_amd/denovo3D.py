"""Host mirror of the reference interface for the denovo3D parameter sweep (Path B).

Same names, argument order and return shapes as the reference callables it stands in for:

* ``simulate_helical_projection``   src/helicon/webApps/denovo3D/utils.py:31-189
* ``compute_power_spectra``         src/helicon/lib/transforms.py:771-820
* ``cross_correlation_coefficient`` src/helicon/lib/analysis.py:777-799
* ``cosine_similarity``             src/helicon/lib/analysis.py:802-821
* ``process_one_task``              src/helicon/webApps/denovo3D/pipeline.py:85-497 (tuple layout)
* ``sweep`` / ``SweepEngine``       replace the thread pool of app.py:2455-2523

All arithmetic happens in libhelicon_hip.so (hand-written gfx950 kernels) through ctypes;
there is no NumPy/SciPy fallback — without the library or a GPU these functions raise.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import threading
from dataclasses import dataclass

import numpy as np

from . import _lib
from .grid import CandidateGrid, build_grid, radial_band_mask, set_to_periodic_range

__all__ = [
    "SweepEngine",
    "SweepResult",
    "sweep",
    "simulate_helical_projection",
    "compute_power_spectra",
    "cross_correlation_coefficient",
    "cosine_similarity",
    "process_one_task",
    "low_high_pass_filter",
    "threshold_data",
    "apply_helical_symmetry",
    "rotate_shift_image",
    "transform_image",
    "rescale",
    "down_scale",
    "pad_to_size",
    "estimate_helix_rotation_center_diameter",
    "auto_horizontalize",
    "transform_map",
    "is_vertical",
    "units_to_cylindrical",
]

_SUPPORTED_N = (32, 64, 128, 256, 512, 1024)   # square sides served by the tuned kernels
_MIN_SIDE, _MAX_SIDE = 8, 1024                    # every other (ny, nx) in this range: runtime-sized kernels


def _largest_prime_factor(n: int) -> int:
    p, m, best = 2, int(n), 1
    while p * p <= m:
        while m % p == 0:
            best, m = p, m // p
        p += 1
    return max(best, m) if m > 1 else best


def _image_shape(ny, nx) -> tuple[int, int]:
    """Validate an image shape for the gfx950 path (``hh_create2``): sides in [8, 1024]."""
    ny, nx = int(ny), int(nx)
    if not (_MIN_SIDE <= ny <= _MAX_SIDE and _MIN_SIDE <= nx <= _MAX_SIDE):
        raise ValueError(f"image sides must lie in [{_MIN_SIDE}, {_MAX_SIDE}]; got ({ny}, {nx})")
    return ny, nx   # (a row length with a prime factor above 31 is served by the direct path: slow, not refused)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def units_to_cylindrical(centers_0: np.ndarray) -> np.ndarray:
    """Cartesian asymmetric-unit positions (columns: projection axis, image-row axis, helical axis —
    the reference's ``centers_0``, utils.py:138-151) -> (radius A, azimuth rad, axial A) float64."""
    c = np.asarray(centers_0, dtype=np.float64).reshape(-1, 3)
    out = np.empty_like(c)
    out[:, 0] = np.hypot(c[:, 0], c[:, 1])
    out[:, 1] = np.arctan2(c[:, 1], c[:, 0])
    out[:, 2] = c[:, 2]
    return out


class SweepEngine:
    """One ``hh_ctx``: a device, an image side, device workspaces.  The context itself is not
    thread-safe; every call takes the engine's re-entrant lock, and a caller that configures and then
    scores (geometry, reference, sweep) holds it across the whole sequence with ``session()`` — the
    reference calls its task function from pool threads (app.py:2473-2476), and two threads with
    different images must not interleave between ``set_reference`` and ``sweep``."""

    def __init__(self, n, device: int = 0, max_batch: int = 0):
        """``n``: the side of a square image, or ``(ny, nx)``.  Square power-of-two sides 32...1024 run the tuned
        kernels; any other shape the runtime-sized ones (same results, lower throughput; ``self.general``)."""
        ny, nx = (n, n) if np.isscalar(n) else n
        self.ny, self.nx = _image_shape(ny, nx)
        self.n = self.ny if self.ny == self.nx else None
        self.general = not (self.ny == self.nx and self.ny in _SUPPORTED_N)
        self._L = _lib.lib()
        self._ctx = C.c_void_p()
        self.device = int(device)
        self._lock = threading.RLock()
        _lib.check(self._L.hh_create2(C.byref(self._ctx), self.device, self.ny, self.nx, int(max_batch)), None)
        self.max_batch = int(self._L.hh_max_batch(self._ctx))
        self.n_segments = 0
        self._geom_key = None
        self._ref_key = None

    @contextlib.contextmanager
    def session(self):
        """Hold the engine for a configure-and-score sequence (re-entrant)."""
        with self._lock:
            yield self

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._L.hh_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        _lib.check(rc, self._ctx)

    # -- configuration ----------------------------------------------------------------------
    def set_stream(self, hip_stream: int | None):
        """Run on a caller-owned hipStream_t given as an int (``torch.cuda.current_stream().cuda_stream``;
        0 is the device's null stream = torch's default stream); ``None`` returns to the engine's own stream."""
        with self._lock:
            if hip_stream is None:
                self._check(self._L.hh_use_own_stream(self._ctx))
            else:
                self._check(self._L.hh_set_stream(self._ctx, C.c_void_p(int(hip_stream))))

    def set_geometry(self, *, apix, helical_diameter, ball_radius, tilt=0.0, psi=0.0, dy=0.0,
                     units=None, tail_bits=0):
        key = (float(apix), float(helical_diameter), float(ball_radius), float(tilt), float(psi), float(dy),
               None if units is None else np.asarray(units, dtype=np.float64).tobytes(), int(tail_bits))
        # the reference asserts (utils.py:88); keep its exception type
        assert helical_diameter + ball_radius < self.ny * apix * 0.99
        g = _lib.hh_geom()
        g.apix, g.helical_diameter, g.ball_radius = float(apix), float(helical_diameter), float(ball_radius)
        g.tilt, g.psi, g.dy = float(tilt), float(psi), float(dy)
        g.tail_bits = int(tail_bits)
        u = None
        if units is not None:
            u = np.ascontiguousarray(np.asarray(units, dtype=np.float64).reshape(-1, 3))
            g.n_units = len(u)
            g.units = _ptr(u, C.c_double)
        else:
            g.n_units = 0
            g.units = None
        with self._lock:  # compare-and-set under the lock
            if key == self._geom_key:
                return
            self._geom_key = None
            self._check(self._L.hh_set_geometry(self._ctx, C.byref(g)))
            self._geom_key = key

    def set_reference(self, images, mask=None, log=True, key=None):
        """``key``: an optional hashable identity of (images, mask, log); when it equals the key of the
        reference the context already holds, the upload and the spectrum preparation are skipped."""
        if key is not None:
            with self._lock:
                if key == self._ref_key:
                    return
        imgs = _f32(images)
        if imgs.ndim == 2:
            imgs = imgs[None]
        if imgs.ndim != 3 or imgs.shape[1:] != (self.ny, self.nx):
            raise ValueError(f"images must be [S, {self.ny}, {self.nx}], got {imgs.shape}")
        if mask is None:
            mask = radial_band_mask(self.ny, self.nx)
        m = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
        if m.shape != (self.ny, self.nx):
            raise ValueError(f"mask must be [{self.ny}, {self.nx}] on the fftshifted plane")
        with self._lock:
            try:
                self._check(self._L.hh_set_reference(self._ctx, _ptr(imgs, C.c_float), imgs.shape[0],
                                                     _ptr(m, C.c_uint8), 1 if log else 0))
            except ValueError:  # HH_ERR_ARG: rejected before anything changed, the old reference stands
                raise
            except Exception:   # the library dropped the old reference (sweeps now report HH_ERR_STATE)
                self.n_segments = 0
                self._ref_key = None
                raise
            self.n_segments = imgs.shape[0]
            self._ref_key = key

    # -- the hot path -----------------------------------------------------------------------
    def sweep(self, params) -> np.ndarray:
        """params [G, 4] float64 (twist, rise, csym, rot) -> scores [S, G] float32 (host in, host out)."""
        p = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, 4)
        out = np.empty((max(self.n_segments, 1), len(p)), dtype=np.float32)
        with self._lock:
            self._check(self._L.hh_sweep(self._ctx, _ptr(p, C.c_double), len(p), _ptr(out, C.c_float)))
        return out

    def sweep_device(self, d_params: int, n_candidates: int, d_scores: int, host_params=None, ld_scores: int = 0):
        """Device pointers (ints): params [G, 4] float64, scores [S, G] float32; asynchronous on
        the engine's stream.  ``host_params`` = the same [G, 4] list on the host, if the caller has
        it: it lets the library take the shared-twist first pass on twist-major grids
        (``hh_sweep_device_mirrored``).  ``ld_scores`` > G: row stride of the score buffer
        (``hh_sweep_device_strided``), e.g. the padded send buffer of an all-gather."""
        with self._lock:
            hp = None
            if host_params is not None:
                hp = np.ascontiguousarray(host_params, dtype=np.float64)
                if hp.shape != (int(n_candidates), 4):
                    raise ValueError("host_params must be [n_candidates, 4]")
            if ld_scores:
                self._check(self._L.hh_sweep_device_strided(self._ctx, C.c_void_p(d_params),
                                                            _ptr(hp, C.c_double) if hp is not None else None,
                                                            int(n_candidates), C.c_void_p(d_scores), int(ld_scores)))
            elif hp is None:
                self._check(self._L.hh_sweep_device(self._ctx, C.c_void_p(d_params), int(n_candidates),
                                                    C.c_void_p(d_scores)))
            else:
                self._check(self._L.hh_sweep_device_mirrored(self._ctx, C.c_void_p(d_params), _ptr(hp, C.c_double),
                                                             int(n_candidates), C.c_void_p(d_scores)))

    def argmax_device(self, d_scores: int, n_rows: int, n: int, ld: int = 0, d_index: int | None = None):
        """Per-row arg-max of device-resident scores (row r at ``d_scores + r * ld``; lowest index on ties, NaN
        never wins).  With ``d_index`` (a device int64 buffer) the call is asynchronous and returns None;
        otherwise the indices come back as a NumPy array."""
        if d_index is not None:
            with self._lock:
                self._check(self._L.hh_argmax_device(self._ctx, C.c_void_p(d_scores), int(n_rows), int(n), int(ld),
                                                     C.c_void_p(d_index), None))
            return None
        out = np.zeros(int(n_rows), dtype=np.int64)
        with self._lock:
            self._check(self._L.hh_argmax_device(self._ctx, C.c_void_p(d_scores), int(n_rows), int(n), int(ld),
                                                 None, _ptr(out, C.c_int64)))
        return out

    # -- the collective of a multi-GPU sweep through the C ABI (RCCL; no torch.distributed needed) ------------
    @staticmethod
    def comm_unique_id() -> bytes:
        """128 bytes made by ONE rank and handed to all others (file, socket, MPI ...): ``hh_comm_unique_id``."""
        buf = C.create_string_buffer(128)
        _lib.check(_lib.lib().hh_comm_unique_id(buf), None)
        return bytes(buf.raw)

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        buf = C.create_string_buffer(unique_id, 128)
        with self._lock:
            self._check(self._L.hh_comm_init(self._ctx, int(rank), int(world), buf))

    def allgather(self, d_send: int, count: int, d_recv: int):
        """``count`` floats of every rank, rank-major, into ``d_recv`` (device pointers); queued on the engine's
        stream behind the sweep that wrote ``d_send``."""
        with self._lock:
            self._check(self._L.hh_allgather(self._ctx, C.c_void_p(d_send), int(count), C.c_void_p(d_recv)))

    def comm_destroy(self):
        with self._lock:
            self._check(self._L.hh_comm_destroy(self._ctx))

    def set_table_path(self, mode=2):
        """How runs of candidates that share (twist, csym, rot) are swept: 0 / False = like any other
        list (raster + two transforms per candidate), 1 = run tables + second pass, 2 / True = the
        fused pass where it fits (default).  Scores agree to float32 rounding."""
        mode = 2 if mode is True else int(mode)
        with self._lock:
            self._check(self._L.hh_set_table_path(self._ctx, mode))

    @property
    def last_first_pass(self) -> str:
        """Pipeline of the last sweep: "transform" (raster + column transform per candidate),
        "run_tables" (shared-twist tables + second pass) or "fused" (shared-twist, no intermediate)."""
        return {1: "run_tables", 2: "fused"}.get(self._L.hh_last_first_pass(self._ctx), "transform")

    def synchronize(self):
        with self._lock:
            self._check(self._L.hh_synchronize(self._ctx))

    # -- the primitives ---------------------------------------------------------------------
    def simulate(self, twist, rise, csym, rot=0.0) -> np.ndarray:
        p = np.array([twist, rise, csym, rot], dtype=np.float64)
        out = np.empty((self.ny, self.nx), dtype=np.float32)
        with self._lock:
            self._check(self._L.hh_simulate(self._ctx, _ptr(p, C.c_double), _ptr(out, C.c_float)))
        return out

    def power_spectrum(self, image, log=True, want_phase=True):
        img = _f32(image)
        if img.shape != (self.ny, self.nx):
            raise ValueError(f"image must be [{self.ny}, {self.nx}]")
        pwr = np.empty((self.ny, self.nx), dtype=np.float32)
        phase = np.empty((self.ny, self.nx), dtype=np.float32) if want_phase else None
        with self._lock:
            self._check(self._L.hh_power_spectrum(self._ctx, _ptr(img, C.c_float), 1 if log else 0,
                                                  _ptr(pwr, C.c_float),
                                                  _ptr(phase, C.c_float) if want_phase else None))
        return pwr, phase

    def low_high_pass_filter(self, image, low_pass_fraction=0.0, high_pass_fraction=0.0) -> np.ndarray:
        img = _f32(image)
        if img.shape != (self.ny, self.nx):
            raise ValueError(f"image must be [{self.ny}, {self.nx}]")
        out = np.empty_like(img)
        with self._lock:
            self._check(self._L.hh_low_high_pass_filter(self._ctx, _ptr(img, C.c_float), float(low_pass_fraction),
                                                        float(high_pass_fraction), _ptr(out, C.c_float)))
        return out

    def threshold_data(self, data, thresh_fraction=None, thresh_value=None) -> np.ndarray:
        x = _f32(data)
        if thresh_fraction is not None and thresh_fraction >= 0:  # filters.py:303-309
            use, thr = 1, float(thresh_fraction)
        elif thresh_value is not None:
            use, thr = 0, float(thresh_value)
        else:
            return x
        out = np.empty_like(x)
        with self._lock:
            self._check(self._L.hh_threshold_data(self._ctx, _ptr(x, C.c_float), x.size, use, thr, _ptr(out, C.c_float)))
        return out

    def _pair(self, a, b, f32name, f64name):
        a = np.asarray(a)
        b = np.asarray(b)
        if a.shape != b.shape:
            raise ValueError("operands must have the same shape")
        if a.size == 0:
            return 0
        out = C.c_double(0.0)
        if a.dtype == np.float32 and b.dtype == np.float32:
            x, y = np.ascontiguousarray(a).ravel(), np.ascontiguousarray(b).ravel()
            fn, ct = getattr(self._L, f32name), C.c_float
        else:
            x = np.ascontiguousarray(a, dtype=np.float64).ravel()
            y = np.ascontiguousarray(b, dtype=np.float64).ravel()
            fn, ct = getattr(self._L, f64name), C.c_double
        with self._lock:
            self._check(fn(self._ctx, _ptr(x, ct), _ptr(y, ct), x.size, C.byref(out)))
        return out.value

    def cross_correlation_coefficient(self, a, b):
        return self._pair(a, b, "hh_cross_correlation", "hh_cross_correlation_f64")

    def cosine_similarity(self, a, b):
        return self._pair(a, b, "hh_cosine_similarity", "hh_cosine_similarity_f64")

    # -- profiling --------------------------------------------------------------------------
    def profile(self, period: int):
        """0 = off; k >= 1 = record HIP events around the launches of every k-th batch."""
        with self._lock:
            self._check(self._L.hh_profile_enable(self._ctx, int(period)))
            self._check(self._L.hh_profile_reset(self._ctx))

    def profile_get(self) -> dict:
        p = _lib.hh_profile()
        with self._lock:
            self._check(self._L.hh_profile_get(self._ctx, C.byref(p)))
        return {name: getattr(p, name) for name, _ in p._fields_}

    def calibrate_traffic(self, mode: int, nbytes: int):
        """Profiling aid: one launch moving ``nbytes`` with the sweep's read (0) / write (1) shape."""
        with self._lock:
            self._check(self._L.hh_calibrate_traffic(self._ctx, int(mode), int(nbytes)))

    def memory_bytes(self) -> dict:
        """Device memory this engine holds right now (``hh_memory_bytes``): total and its parts."""
        parts = (C.c_int64 * 5)()
        total = self._L.hh_memory_bytes(self._ctx, parts)
        names = ("run_tables", "column_factors", "intermediate", "segment_buffers", "other")
        return dict(total=int(total), **{k: int(v) for k, v in zip(names, parts)})

    def algorithmic_bytes(self) -> int:
        return 4 * self.ny * self.nx + 16 * self.nx * (self.ny // 2 + 1)  # = hh_algorithmic_bytes(n) for a square


# ------------------------------------------------------------------------------------------
# module-level engines, one per (side, device)
# ------------------------------------------------------------------------------------------
_engines: dict = {}
_engines_lock = threading.Lock()


def _engine(shape, device: int = 0) -> SweepEngine:
    shape = (int(shape), int(shape)) if np.isscalar(shape) else (int(shape[0]), int(shape[1]))
    with _engines_lock:
        e = _engines.get((shape, device))
        if e is None:
            e = _engines[(shape, device)] = SweepEngine(shape, device)
        return e


def _square_side(ny, nx):
    """Shape check of the entry points that need the tuned kernels (the Fourier filter)."""
    if ny != nx or ny not in _SUPPORTED_N:
        raise ValueError(f"this function handles square images with side in {_SUPPORTED_N}; got ({ny}, {nx})")
    return int(ny)


# ------------------------------------------------------------------------------------------
# drop-in primitives
# ------------------------------------------------------------------------------------------
def simulate_helical_projection(n, twist, rise, csym, helical_diameter, ball_radius, polymer, planarity,
                                ny, nx, apix, tilt=0, rot=0, psi=0, dy=0, *, device=0):
    """utils.py:31-47.  ``n > 1`` draws the asymmetric unit from the global NumPy RNG exactly as the
    reference does (utils.py:139-144); ``polymer=1`` grows it as the reference's self-avoiding random walk
    (``helicon_amd.polymer``, utils.py:125-136, 192-333), also from the global RNG.  Either way ``np.random.seed``
    replays the reference's image."""
    assert helical_diameter + ball_radius < ny * apix * 0.99  # utils.py:88
    assert n >= 1
    if not rise > 0:
        raise ValueError("negative dimensions are not allowed")  # what np.zeros raises in the reference
    eng = _engine(_image_shape(ny, nx), device)
    units = None
    if polymer:
        from .polymer import polymer_units

        units = units_to_cylindrical(polymer_units(n, helical_diameter, int(csym), planarity))
        if len(units) > 64:   # HH_MAX_UNITS
            raise ValueError(f"the polymer has {len(units)} atoms (with their csym copies); the device lattice takes up to 64 per "
                             "asymmetric unit")
        rot = 0.0             # the reference's polymer branch never uses `rot` (utils.py:125-136)
    elif n > 1:  # the reference's three draws, in its order (utils.py:140-144); rot is added on the device
        r = np.sqrt(np.random.uniform(0, helical_diameter**2 / 4, n))
        angle = np.random.uniform(-np.pi, np.pi, n)
        z = np.random.uniform(-rise / 2, rise / 2, n)
        units = np.stack([r, angle, z], axis=1)
    with eng.session():
        eng.set_geometry(apix=apix, helical_diameter=helical_diameter, ball_radius=ball_radius,
                         tilt=tilt, psi=psi, dy=dy, units=units)
        return eng.simulate(twist, rise, int(csym), float(rot)).astype(np.float64)


def compute_power_spectra(data, apix, cutoff_res=None, output_size=None, log=True,
                          low_pass_fraction=0, high_pass_fraction=0, *, device=0):
    """transforms.py:771-820; returns ``(pwr, phase)``.  Default Fourier sampling: the sweep's own transform kernels.  With
    ``cutoff_res`` / ``output_size`` (the Fourier-space zoom of ``fft_rescale``, transforms.py:663-713): the direct
    non-uniform transform ``hh_power_spectrum_zoom`` — what the reference's finufft call approximates to 1e-6."""
    data = np.asarray(data)
    if data.ndim != 2:
        raise NotImplementedError("only 2D images are on the accelerated path")
    zoom = (cutoff_res is not None and tuple(cutoff_res) != (2 * apix, 2 * apix)) or \
           (output_size is not None and tuple(output_size) != tuple(data.shape))
    # an odd side: the reference's (-1)^(u + v) does not undo the shift of finufft's centred modes there, so its transform
    # is fft2 times a unit-modulus phase ramp (same amplitudes, other phases) — the direct sum reproduces exactly that
    if zoom or data.shape[0] % 2 or data.shape[1] % 2:
        cy, cx = (float(v) for v in cutoff_res) if cutoff_res else (2.0 * apix, 2.0 * apix)
        ony, onx = (int(v) for v in output_size) if output_size else data.shape
        img = np.ascontiguousarray(data, dtype=np.float32)
        pwr = np.empty((ony, onx), dtype=np.float32)
        phase = np.empty((ony, onx), dtype=np.float32)
        L = _lib.lib()
        _lib.check(L.hh_power_spectrum_zoom(int(device), _ptr(img, C.c_float), img.shape[0], img.shape[1], ony, onx, float(apix),
                                            cy, cx, 1 if log else 0, _ptr(pwr, C.c_float), _ptr(phase, C.c_float)), None)
        if 0 < low_pass_fraction < 1 or 0 < high_pass_fraction < 1:   # on the normalised spectrum: see below
            f = low_high_pass_filter(pwr, low_pass_fraction, high_pass_fraction, device=device)
            vmin, vmax = float(f.min()), float(f.max())
            pwr = (f - vmin) / (vmax - vmin) if vmax != vmin else f
        return pwr.astype(np.float64), phase.astype(np.float64)
    eng = _engine(_image_shape(*data.shape), device)
    with eng.session():
        pwr, phase = eng.power_spectrum(data, log=log, want_phase=True)
        if 0 < low_pass_fraction < 1 or 0 < high_pass_fraction < 1:
            # transforms.py:811-817 filters log1p|F| and then min-max normalises.  The device spectrum is already
            # normalised, x -> a x + b with a > 0; the filter maps the constant b to another constant and min-max
            # normalisation removes any positive affine map, so normalise(filter(a x + b)) = normalise(filter(x)).
            f = eng.low_high_pass_filter(pwr, low_pass_fraction, high_pass_fraction)
            vmin, vmax = float(f.min()), float(f.max())
            pwr = (f - vmin) / (vmax - vmin) if vmax != vmin else f  # filters.py:276-280
    return pwr.astype(np.float64), phase.astype(np.float64)


def cross_correlation_coefficient(a, b, *, device=0):
    """analysis.py:777-799."""
    return _engine(64, device).cross_correlation_coefficient(a, b)


def cosine_similarity(a, b, *, device=0):
    """analysis.py:802-821."""
    return _engine(64, device).cosine_similarity(a, b)


def low_high_pass_filter(data, low_pass_fraction=0, high_pass_fraction=0, *, device=0):
    """``helicon.low_high_pass_filter`` (lib/filters.py:314-372) for a 2-D image of any size: Gaussian low / high
    pass in Fourier space on the device (square power-of-two sides: the sweep's own float32 transforms; anything else:
    direct float64 transforms), float64 result like the reference's.  3-D input is outside the accelerated path."""
    d = np.asarray(data)
    if d.ndim == 3:
        raise NotImplementedError("3-D low_high_pass_filter is outside the accelerated path")
    if d.ndim != 2:
        raise ValueError("Input data must be a 2D or 3D array.")  # filters.py:336-337
    return _engine(_image_shape(*d.shape), device).low_high_pass_filter(d, low_pass_fraction, high_pass_fraction).astype(np.float64)


def threshold_data(data, thresh_fraction=None, thresh_value=None, *, device=0):
    """``helicon.threshold_data`` (lib/filters.py:283-311) on the device; returns the input unchanged
    when neither threshold applies, like the reference."""
    d = np.asarray(data)
    if not ((thresh_fraction is not None and thresh_fraction >= 0) or thresh_value is not None):
        return data
    eng = _engine(32, device)  # the kernel does not depend on the image side; any context of the device serves
    return eng.threshold_data(d, thresh_fraction, thresh_value).astype(d.dtype if d.dtype.kind == "f" else np.float64)


def rotate_shift_image(data, angle=0, pre_shift=(0, 0), post_shift=(0, 0), rotation_center=None, order=1, *, device=0):
    """``helicon.rotate_shift_image`` (lib/transforms.py:315-369): rotate by ``angle`` degrees about ``rotation_center``
    (default: the pixel (ny // 2, nx // 2)) with (y, x) shifts before and after, resampled as
    ``scipy.ndimage.affine_transform(order, mode="constant")`` does — on the device (``hh_affine_transform_2d`` for
    order 1, the reference's default; ``hh_affine_transform_2d_cubic`` for order 3, what ``auto_horizontalize`` asks for).
    The 2 x 2 matrix and the offset are float32 quantities in the reference; they are formed the same way here."""
    d = np.asarray(data)
    if d.ndim != 2:
        raise ValueError("data must be a 2D image")
    if order not in (1, 3):
        raise NotImplementedError("the device resampler provides order=1 (the reference's default) and order=3")
    if angle == 0 and pre_shift == [0, 0] and post_shift == [0, 0]:   # (the reference's test: lists only)
        return d * 1.0
    ny, nx = d.shape
    centre = np.array((ny // 2, nx // 2), dtype=np.float32) if rotation_center is None else np.array(rotation_center, dtype=np.float32)
    a = np.deg2rad(angle)
    m = np.array([[np.cos(a), np.sin(a)], [-np.sin(a), np.cos(a)]], dtype=np.float32)
    offset = -np.dot(m, np.array(post_shift, dtype=np.float32))   # the shift after the rotation
    offset += centre - np.dot(m, centre)                          # rotation about the centre
    offset += -np.array(pre_shift, dtype=np.float32)              # the shift before it
    img = np.ascontiguousarray(d, dtype=np.float32)
    out = np.empty_like(img)
    mat = (C.c_double * 4)(*[float(v) for v in m.ravel()])
    off = (C.c_double * 2)(*[float(v) for v in offset])
    fn = _lib.lib().hh_affine_transform_2d if order == 1 else _lib.lib().hh_affine_transform_2d_cubic
    _lib.check(fn(int(device), _ptr(img, C.c_float), ny, nx, mat, off, _ptr(out, C.c_float)), None)
    return out if d.dtype == np.float32 else out.astype(d.dtype if d.dtype.kind == "f" else np.float64)


def _float_image(image, integers_as_values=False):
    """A 2-D image in the floating type scikit-image would compute in (float32 stays, float16 -> float32, float64 stays).
    scikit-image rescales INTEGER images to [0, 1] (``img_as_float``) before it resamples them; that convention is not
    reproduced — integer images are refused unless the caller's result does not depend on a scale of the intensities."""
    d = np.asarray(image)
    if d.ndim != 2:
        raise ValueError("image must be 2D")
    if d.dtype.kind != "f" and not integers_as_values:
        raise TypeError("a floating-point image is expected (scikit-image would rescale an integer image to [0, 1] first: convert it)")
    return np.ascontiguousarray(d, dtype=np.float32 if d.dtype in (np.float32, np.float16) else np.float64)


def _affine_params(scale=(1.0, 1.0), rotation=0.0, translation=(0.0, 0.0)):
    # skimage.transform.AffineTransform(scale=(sx, sy), rotation, translation=(tx, ty)).params on (x, y, 1) columns
    sx, sy = scale
    return np.array([[sx * np.cos(rotation), -sy * np.sin(rotation), translation[0]],
                     [sx * np.sin(rotation), sy * np.cos(rotation), translation[1]],
                     [0.0, 0.0, 1.0]])


def transform_image(image, scale=1.0, rotation=0.0, rotation_center=None, pre_translation=(0.0, 0.0),
                    post_translation=(0.0, 0.0), mode="constant", order=1, *, device=0):
    """``helicon.transform_image`` (lib/transforms.py:238-312): translate, rotate / scale about ``rotation_center``
    (default (ny / 2, nx / 2)), translate — resampled as ``skimage.transform.warp(image, xform.inverse, mode, order)``
    does for a 2-D image (its fast path: the image's own floating type, corners outside the image = 0, result clipped
    to the input's range), on the device (``hh_warp_affine_2d``).  The transforms are composed on the host exactly as
    the reference composes its ``AffineTransform`` objects (pre_translation -> to centre -> rotation / scale -> back ->
    post_translation; (y, x) arguments reversed to scikit-image's (x, y)).  Pinned by derivation (scikit-image is not
    installed beside the reference here): ``oracle/prep.py``."""
    if mode != "constant":
        raise NotImplementedError("mode='constant' is provided (the reference's default, and its only use)")
    if order not in (0, 1):
        raise NotImplementedError("orders 0 and 1 are provided (1 is the reference's default, and its only use)")
    img = _float_image(image)
    centre = np.array(img.shape) / 2.0 if rotation_center is None else np.asarray(rotation_center, dtype=np.float64)
    sc = np.array((scale, scale), dtype=np.float64) if isinstance(scale, (int, float)) else np.asarray(scale, dtype=np.float64)
    m = _affine_params(translation=tuple(pre_translation[::-1]))
    for nxt in (_affine_params(translation=tuple(-centre[::-1])), _affine_params(scale=tuple(sc[::-1]), rotation=np.deg2rad(rotation)),
                _affine_params(translation=tuple(centre[::-1])), _affine_params(translation=tuple(post_translation[::-1]))):
        m = nxt @ m   # a + b applies a first
    inv = np.ascontiguousarray(np.linalg.inv(m), dtype=np.float64)
    out = np.empty_like(img)
    _lib.check(_lib.lib().hh_warp_affine_2d(int(device), img.ctypes.data, int(img.dtype == np.float64), img.shape[0], img.shape[1],
                                            _ptr(inv, C.c_double), int(order), 0.0, 1, out.ctypes.data), None)
    return out


def rescale(image, scale, order=3, anti_aliasing=True, clip=True, *, device=0):
    """``skimage.transform.rescale(image, scale, order, anti_aliasing)`` of a 2-D image with the defaults the reference
    leaves alone (mode "reflect", clip) — what the app's binning (app.py:1911-1922) and ``down_scale`` call — on the
    device (``hh_rescale_2d``): output shape ``max(round(scale * shape), 1)``, a Gaussian of sigma (1 / scale - 1) / 2
    against aliasing, cubic (or linear) spline zoom on pixel-area coordinates, clip to the input's range."""
    if order not in (1, 3):
        raise NotImplementedError("orders 1 and 3 are provided (3 is what the reference asks for)")
    img = _float_image(image)
    out_shape = np.maximum(np.round(np.atleast_1d(scale) * np.asarray(img.shape)), 1).astype(int)
    out = np.empty(tuple(out_shape), dtype=img.dtype)
    _lib.check(_lib.lib().hh_rescale_2d(int(device), img.ctypes.data, int(img.dtype == np.float64), img.shape[0], img.shape[1],
                                        int(out_shape[0]), int(out_shape[1]), int(order), int(bool(anti_aliasing)), int(bool(clip)),
                                        out.ctypes.data), None)
    return out


def pad_to_size(data, shape):
    """``helicon.pad_to_size`` (lib/transforms.py:441-479) for 2-D images: zero padding, the smaller half first."""
    d = np.asarray(data)
    if d.shape == tuple(shape):
        return d
    ny, nx = d.shape
    my, mx = shape
    yb, xb = max(0, (my - ny) // 2), max(0, (mx - nx) // 2)
    return np.pad(d, ((yb, max(0, my - yb - ny)), (xb, max(0, mx - xb - nx))), mode="constant")


def down_scale(data, target_apix, apix_orig, *, device=0):
    """``helicon.down_scale`` (lib/filters.py:375-412): resample to a LARGER pixel size (``rescale`` with anti-aliasing,
    cubic) and pad to even sides; a target at or below the image's own pixel size returns the image unchanged."""
    if target_apix == apix_orig or target_apix < apix_orig:
        return data
    out = rescale(data, apix_orig / target_apix, order=3, anti_aliasing=True, device=device)
    ny, nx = out.shape
    return pad_to_size(out, (ny + ny % 2, nx + nx % 2))


def estimate_helix_rotation_center_diameter(data, estimate_rotation=True, estimate_center=True, threshold=0, *, device=0):
    """``helicon.estimate_helix_rotation_center_diameter`` (lib/analysis.py:645-728): the rotation (degrees) that makes
    the helix horizontal, the vertical shift (pixels) that centres it and its diameter (pixels), from the intensity-
    weighted second moments of the closed mask ``data > threshold`` — mask, closing and moments on the device
    (``hh_helix_moments``), the rotation in between with ``transform_image``."""
    img = _float_image(data, integers_as_values=True)   # (rotation, centre and extent do not depend on the intensity scale)
    ny, nx = img.shape

    def weighted(im):
        m = (C.c_double * 8)()
        _lib.check(_lib.lib().hh_helix_moments(int(device), im.ctypes.data, int(im.dtype == np.float64), ny, nx, float(threshold), m), None)
        count, cy, _cx, i_yy, i_xx, i_xy, first, last = list(m)
        if count < 1:
            return None
        if count < 2:
            return 0.0, 0.0, ny
        angle = np.rad2deg(0.5 * np.arctan2(2.0 * i_xy, i_yy - i_xx)) + 90.0
        if abs(angle) > 90.0:
            angle -= 180.0
        return angle, (ny // 2 - cy) if estimate_center else 0.0, int(last - first + 1)

    first = weighted(img)
    if first is None:
        return 0.0, 0.0, ny
    if estimate_rotation:
        rotation = set_to_periodic_range(first[0], min=-180, max=180)
        rotated = transform_image(img, rotation=rotation, device=device)
    else:
        rotation, rotated = 0.0, img
    second = weighted(rotated)
    if second is None:
        return rotation, 0.0, ny
    return rotation, second[1], second[2]


def auto_horizontalize(data, refine=False, *, device=0):
    """``auto_horizontalize`` (webApps/denovo3D/utils.py:383-424): rotate and shift the image so that the helix lies
    along x through the middle row — the estimate above, optionally refined by a Nelder-Mead search (``scipy.optimize.
    fmin``, as in the reference) on minus the standard deviation of the mirrored row profile, every evaluation a device
    resampling; the result resampled with cubic splines."""
    d = np.asarray(data)
    work = np.clip(d, 0, None)
    theta, shift_y, _ = estimate_helix_rotation_center_diameter(d, device=device)
    if refine:
        from scipy.optimize import fmin

        def score(x):
            tmp = rotate_shift_image(work, angle=x[0], post_shift=(x[1], 0), device=device)
            y = np.sum(tmp, axis=1)[1:]
            y = y + y[::-1]
            return -np.std(y)

        theta, shift_y = fmin(score, x0=(theta, shift_y), xtol=1e-2, disp=0)
    return rotate_shift_image(d, angle=theta, post_shift=(shift_y, 0), order=3, device=device), theta, shift_y


def transform_map(data, scale=1.0, rot=0, tilt=0, psi=0, dx=0, dy=0, dz=0, *, device=0):
    """``helicon.transform_map`` (lib/transforms.py:168-235): scale, rotate (intrinsic ZYZ Euler angles, degrees) and
    shift the sampling grid of a (nz, ny, nx) volume about its centre voxel and resample it like
    ``scipy.ndimage.map_coordinates(order=3)`` — on the device (``hh_transform_map``).  With every argument at its
    default the input itself is returned, like the reference."""
    if scale == 1 and rot == 0 and tilt == 0 and psi == 0 and dx == 0 and dy == 0 and dz == 0:
        return data
    vol = np.ascontiguousarray(data, dtype=np.float32)
    if vol.ndim != 3:
        raise ValueError("data must be a 3D volume (nz, ny, nx)")
    out = np.empty_like(vol)
    shape = (C.c_int32 * 3)(*vol.shape)
    _lib.check(_lib.lib().hh_transform_map(int(device), _ptr(vol, C.c_float), shape, float(scale), float(rot), float(tilt),
                                           float(psi), float(dx), float(dy), float(dz), _ptr(out, C.c_float)), None)
    d = np.asarray(data)
    return out if d.dtype == np.float32 else out.astype(d.dtype if d.dtype.kind == "f" else np.float64)


def is_vertical(data):
    """webApps/denovo3D/utils.py:429-447: the strongest column sum exceeds the strongest row sum."""
    d = np.asarray(data)
    return bool(np.max(np.sum(d, axis=0)) > np.max(np.sum(d, axis=1)))


def apply_helical_symmetry(data, apix, twist_degree, rise_angstrom, csym=1, fraction=1.0, new_size=None,
                           new_apix=None, cpu=1, *, device=0, return_kernel_ms=False):
    """transforms.py:58-74 (same positional signature; ``cpu`` is accepted and ignored).  ``new_size=None``
    means "same size" (the reference cannot unpack ``None``, transforms.py:78-79)."""
    vol = np.ascontiguousarray(data, dtype=np.float32)
    if vol.ndim != 3:
        raise ValueError("data must be a 3D volume (nz, ny, nx)")
    if new_apix is None:
        new_apix = apix
    if new_size is None:
        new_size = vol.shape
    L = _lib.lib()
    in_shape = (C.c_int32 * 3)(*vol.shape)
    want = (C.c_int32 * 3)(*[int(v) for v in new_size])
    out_shape = (C.c_int32 * 3)()
    args = (int(device), _ptr(vol, C.c_float), in_shape, float(apix), float(twist_degree), float(rise_angstrom),
            int(csym), float(fraction), want, float(new_apix))
    _lib.check(L.hh_apply_helical_symmetry(*args, None, out_shape, None), None)
    out = np.empty(tuple(out_shape), dtype=np.float32)
    ms = C.c_double(0.0)
    _lib.check(L.hh_apply_helical_symmetry(*args, _ptr(out, C.c_float), out_shape, C.byref(ms)), None)
    return (out, ms.value) if return_kernel_ms else out


# ------------------------------------------------------------------------------------------
# the sweep
# ------------------------------------------------------------------------------------------
@dataclass
class SweepResult:
    scores: np.ndarray      # [S, C, T, R] float32; skipped candidates are -inf
    grid: CandidateGrid
    best_index: np.ndarray  # [S] flat candidate index of the arg-max (lowest index on ties)
    best: list              # [S] tuples (twist, rise, csym, score)


def _argmax(scores_1d: np.ndarray) -> int:
    L = _lib.lib()
    s = _f32(scores_1d)
    idx = C.c_int64(0)
    _lib.check(L.hh_argmax(_ptr(s, C.c_float), s.size, C.byref(idx)), None)
    return int(idx.value)


def finish_sweep(scores: np.ndarray, grid: CandidateGrid) -> SweepResult:
    """scores [S, G] -> SweepResult (mask skipped candidates, arg-max per segment)."""
    scores = np.array(scores, dtype=np.float32, copy=True)
    scores[:, ~grid.valid] = -np.inf
    best_index = np.array([_argmax(s) for s in scores], dtype=np.int64)
    best = []
    for s, g in enumerate(best_index):
        tw, rs, cs, _ = grid.params[g]
        best.append((float(tw), float(rs), int(cs), float(scores[s, g])))
    return SweepResult(scores.reshape((scores.shape[0],) + grid.shape), grid, best_index, best)


def sweep(images, twists, rises, csyms=(1,), *, apix, helical_diameter, ball_radius, mask=None, log=True,
          rot=0.0, tilt=0.0, psi=0.0, dy=0.0, device=0, engine: SweepEngine | None = None) -> SweepResult:
    """Score every (csym, twist, rise) candidate against the experimental image(s) on one GPU.
    For several GPUs see ``helicon_amd.distributed.sweep_distributed``."""
    imgs = np.asarray(images)
    ny, nx = _image_shape(*imgs.shape[-2:])
    eng = engine or _engine((ny, nx), device)
    grid = build_grid(twists, rises, csyms, tube_length=nx * apix, rot=rot)
    from .distributed import harmless_rise

    params = grid.params.copy()
    params[~grid.valid, 1] = harmless_rise(grid)  # skipped pairs still occupy a slot
    with eng.session():
        eng.set_geometry(apix=apix, helical_diameter=helical_diameter, ball_radius=ball_radius,
                         tilt=tilt, psi=psi, dy=dy)
        eng.set_reference(imgs, mask, log=log)
        scores = eng.sweep(params)
    return finish_sweep(scores, grid)


# ------------------------------------------------------------------------------------------
# pipeline.process_one_task tuple layout (pipeline.py:85-122, 469-497), Path-B scoring
# ------------------------------------------------------------------------------------------
def _digest(a: np.ndarray) -> tuple:
    """Content identity of an array (shape, dtype, 64-bit hash of the bytes)."""
    a = np.ascontiguousarray(a)
    buf = memoryview(a).cast("B")
    try:
        import xxhash

        h = xxhash.xxh3_64_intdigest(buf)
    except ImportError:  # pragma: no cover - xxhash ships with the image
        import hashlib

        h = hashlib.sha1(buf).digest()
    return (a.shape, a.dtype.str, h)


_prepared: dict = {}          # (image digest, preparation options) -> prepared float image
_blank: dict = {}             # image digest -> np.std(image) == 0
_prepared_lock = threading.Lock()
_PREPARED_MAX = 16


def _prepare_base(data, apix, low_pass, transpose, horizontalize, device):
    """``prepare_data`` of pipeline.py:180-208 on one image: Gaussian low / high pass (:183-188), transpose (:202-203),
    ``auto_horizontalize(refine=True)`` (:204-208).  (The denoisers of :189-201 are scikit-image restoration filters and
    are refused by the caller.)"""
    if low_pass is not None and low_pass > 2 * apix:
        data = low_high_pass_filter(data, low_pass_fraction=2 * apix / low_pass,
                                    high_pass_fraction=2.0 / np.max(data.shape), device=device)
    if transpose is not None and (transpose > 0 or (transpose < 0 and is_vertical(data))):   # pipeline.py:202-203
        data = data.T
    if horizontalize:
        data, _theta, _shift = auto_horizontalize(data, refine=True, device=device)
    return data


def _prepare_task_image(data, apix, low_pass, transpose, thresh_fraction, tube_diameter, device, horizontalize=0,
                        target_apix2d=None):
    """pipeline.py:180-286 on one image: ``prepare_data`` (above), ``down_scale`` to ``target_apix2d`` when that is
    larger than the image's pixel size (:268-275), background subtraction + threshold + /max (:277-284;
    ``tube_diameter`` already resolved by the caller).  Unlike the reference (pipeline.py:282 works in place when no
    rescale happened) the caller's array is never modified."""
    data = _prepare_base(data, apix, low_pass, transpose, horizontalize, device)
    ny0 = data.shape[0]
    a2 = apix if target_apix2d is None or target_apix2d < apix else float(target_apix2d)
    data = down_scale(data, a2, apix, device=device)
    ny, nx = data.shape
    if thresh_fraction is not None and thresh_fraction >= 0:
        # pipeline.py:253-255: reconstruct_diameter = tube_diameter if 0 < tube_diameter < ny*apix else ny*apix (BEFORE the rescale)
        rec_d = tube_diameter if 0 < tube_diameter < ny0 * apix else ny0 * apix
        nr = min(ny // 2 - 1, int(np.ceil(rec_d / 2 / a2) + 1))
        data = np.asarray(data, dtype=np.float64) - np.median(np.asarray(data)[(ny // 2 - nr, ny // 2 + nr), :])
        data = threshold_data(data, thresh_fraction=thresh_fraction, device=device)
        data = data / np.max(data)
    return data


def lsq_box(ny, nx, apix, rise, rise_range, tilt_range, target_apix3d, tube_length, tube_diameter, tube_diameter_inner,
            reconstruct_length, sym_oversample, return_3d, orig=None):
    """The reconstruction box of one task — pipeline.py:242-349, the reference's integer arithmetic:
    ``(apix3d, D2d, L2d, D3d, D3d_inner, L3d, sym_oversample)``.  (ny, nx, apix) describe the image handed to the solver;
    ``orig`` = (ny, nx, apix) of the image BEFORE ``down_scale`` when the task rescaled it (the tube's length and the
    reconstruction's diameter / length are worked out on that one, pipeline.py:242-266)."""
    a2 = apix
    ny0, nx0, a0 = orig if orig is not None else (ny, nx, apix)
    # tube length, reconstruction diameter / length (pipeline.py:242-266)
    if tube_length < 0:
        tube_length = int(nx0 * a0) if tube_diameter > ny0 * a0 / 2 else round(np.sqrt((nx0 * a0) ** 2 / 4 - tube_diameter ** 2 / 4) * 2)
    rec_d = tube_diameter if 0 < tube_diameter < ny0 * a0 else ny0 * a0
    rec_d_inner = tube_diameter_inner if 0 < tube_diameter_inner < rec_d else 0
    if reconstruct_length < rise:
        reconstruct_length = max(min(3 * np.max(rise_range), tube_length),
                                 round(np.tan(np.deg2rad(np.max(np.abs(tilt_range)))) * tube_diameter * 3))
    # voxel size (pipeline.py:288-300)
    if target_apix3d < 0:
        vol = reconstruct_length * (rec_d ** 2 - rec_d_inner ** 2) / 4 * np.pi
        a3 = max(a2, round(np.power(vol / (nx * ny), 1 / 3) + 0.5))
    elif target_apix3d == 0:
        a3 = a2
    else:
        a3 = target_apix3d
    # box sizes in pixels, all even (pipeline.py:305-331)
    even = lambda v: v + v % 2  # noqa: E731
    d3 = even(int(round(rec_d / a3)))
    d3_inner = int(round(tube_diameter_inner / a3))
    d2 = even(int(round(rec_d / a2)))
    len2 = tube_length if 0 < tube_length < nx * a2 else nx * a2
    l2 = even(int(len2 / a2))
    if reconstruct_length > 0:
        l3 = even(max(int(np.ceil(rise / a3)), int(np.ceil(reconstruct_length / a3))))
    else:
        l3 = even(int(l2 * a2 / a3 + 0.5))
    if sym_oversample <= 0:  # pipeline.py:333-345
        ratio = 2 ** 20 / (l3 * (d3 ** 2 - d3_inner ** 2))
        if ratio < 10:
            sym_oversample = max(1, int(round(ratio)))
        elif ratio < 100:
            sym_oversample = max(1, int(round(ratio / 10)) * 10)
        else:
            sym_oversample = max(1, int(round(ratio / 100)) * 100)
        if return_3d:
            sym_oversample *= 2
    return a3, d2, l2, d3, d3_inner, l3, sym_oversample


def _task_lsq(data, prepared, imageFile, imageIndex, twist, rise, rise_range, csym, tilt, tilt_range, psi, dy, apix,
              target_apix3d, thresh_fraction, positive_constraint, tube_length, tube_diameter, tube_diameter_inner,
              reconstruct_length, sym_oversample, interpolation, fsc_test, return_3d, score_metric, opts, device, transpose,
              low_pass, horizontalize=0, target_apix2d=None, orig_shape=None, psi_range=0, dy_range=0):
    """pipeline.py:242-496: the reconstruction box from the tube's dimensions, ``lsq_reconstruct`` on the (possibly
    down-scaled) image, helical symmetrisation back on the INPUT's grid, projections and z sections.  ``apix`` is the
    input's pixel size (apix2d_orig), ``target_apix2d`` the solver image's (>= apix), ``orig_shape`` the image's shape
    before ``down_scale``."""
    from .solver import lsq_reconstruct

    # the image the reference would have at pipeline.py:286 (``prepared`` already went through the low pass, the
    # transpose, the rescale and, with thresh_fraction >= 0, the background subtraction + threshold + / max), and its
    # ``data_orig``
    img = np.asarray(prepared)
    ny, nx = img.shape
    a2 = apix if target_apix2d is None or target_apix2d < apix else float(target_apix2d)
    ny0, nx0 = orig_shape if orig_shape is not None else (ny, nx)
    if thresh_fraction is not None and thresh_fraction >= 0:
        # data_orig is the image after the in-place median subtraction (pipeline.py:277-282: ``data_orig = data`` aliases it)
        base = _prepare_task_image(data, apix, low_pass, transpose, None, tube_diameter, device, horizontalize, a2)
        rec_d0 = tube_diameter if 0 < tube_diameter < ny0 * apix else ny0 * apix
        nr = min(ny // 2 - 1, int(np.ceil(rec_d0 / 2 / a2) + 1))
        data_orig = np.asarray(base) - np.median(np.asarray(base)[(ny // 2 - nr, ny // 2 + nr), :])
        data_orig = data_orig.astype(np.asarray(base).dtype, copy=False)
    else:
        data_orig = img
    a3, d2, l2, d3, d3_inner, l3, sym_oversample = lsq_box(ny, nx, a2, rise, rise_range, tilt_range, target_apix3d, tube_length,
                                                          tube_diameter, tube_diameter_inner, reconstruct_length, sym_oversample,
                                                          return_3d, orig=(ny0, nx0, apix))
    model = {k: v for k, v in opts.items() if k in ("model", "alpha", "l1_ratio")}   # app.py:2385-2387: model, l1_ratio (+ alpha)
    model.setdefault("model", "lsq")
    # the local refinement's ranges (pipeline.py:357-369): only for the models that have it, only where a range is open
    refine_range = None
    if model.get("model", "lsq") in ("lsq", "elasticnet", "lasso", "ridge"):
        r_dict = {}
        if tilt_range is not None and tilt_range[1] > tilt_range[0]:
            r_dict["tilt"] = max(abs(tilt_range[0]), abs(tilt_range[1]))
        for name, rng in (("psi", psi_range), ("dy", dy_range)):
            width = float(np.max(np.abs(rng))) if rng is not None else 0.0   # (the app passes scalars; a (lo, hi) pair is read as its extent)
            if width > 0:
                r_dict[name] = width
        refine_range = r_dict or None
    lsq_reconstruct._refined_params = {}
    (rec3d, set1, set2), score = lsq_reconstruct(
        img, a2 / a3, twist, rise / a3, csym, tilt, psi, dy / a2, thresh_fraction=thresh_fraction,
        positive_constraint=positive_constraint, reconstruct_diameter_3d_inner_pixel=d3_inner,
        reconstruct_diameter_2d_pixel=d2, reconstruct_diameter_3d_pixel=d3, reconstruct_length_2d_pixel=l2,
        reconstruct_length_3d_pixel=l3, sym_oversample=sym_oversample, interpolation=interpolation, fsc_test=fsc_test,
        score_metric=score_metric, target_apix2d=a2, algorithm=model, refine_tilt_psi_dy_range=refine_range, device=device)
    refined = getattr(lsq_reconstruct, "_refined_params", {}) or {}   # pipeline.py:419-428: refined angles for the projections, consumed once
    lsq_reconstruct._refined_params = {}
    tilt_viz, psi_viz, dy_viz = refined.get("tilt", tilt), refined.get("psi", psi), refined.get("dy", dy)
    # the map on the input's grid (its size and pixel size BEFORE any rescale), at least 1.2 pitches long (pipeline.py:398-417)
    tw_eff = twist if abs(twist) < 90 else 180 - abs(twist)
    pitch_pixel = int(360 / abs(tw_eff) * rise / apix + 0.5) if abs(tw_eff) > 1e-2 else int(np.ceil(2 * rise / apix))
    new_length = max(nx0, int(pitch_pixel * 1.2))
    sym = apply_helical_symmetry(rec3d, a3, twist, rise, csym, new_size=(new_length, ny0, ny0), new_apix=apix, device=device)
    tilted = transform_map(sym, scale=1.0, tilt=tilt_viz, psi=psi_viz, dy=dy_viz / apix, device=device)   # pipeline.py:430-432
    x_proj = np.sum(tilted, axis=2).T
    y_proj = np.sum(tilted, axis=1).T
    y_max = y_proj.max()
    if y_max > 0:
        y_proj = y_proj * (x_proj.max() / y_max)
    per_rise = max(1, int(np.ceil(rise / apix)))
    z0 = sym.shape[0] // 2 - per_rise // 2
    z_sections = np.sum(sym[z0:z0 + per_rise], axis=0)
    lo, hi = z_sections.min(), z_sections.max()
    if hi > lo:
        z_sections = (z_sections - lo) * (x_proj.max() - x_proj.min()) / (hi - lo) + x_proj.min()
    return (
        score,
        (x_proj, y_proj, z_sections, (rec3d, set1, set2) if return_3d else None, d2, d3, l2, l3),
        (data_orig, imageFile, imageIndex, a3, a2, twist, rise, csym, tilt, psi, dy),
    )


def process_one_task(ti, ntasks, data, imageFile, imageIndex, twist, rise, rise_range, csym, tilt, tilt_range,
                     psi, psi_range, dy, dy_range, apix2d_orig, denoise, low_pass, transpose, horizontalize,
                     target_apix3d, target_apix2d, thresh_fraction, positive_constraint, tube_length,
                     tube_diameter, tube_diameter_inner, reconstruct_length, sym_oversample, interpolation,
                     fsc_test, return_3d, score_metric, algorithm, verbose, n_cpu=1):
    """Same positional signature and return layout as the reference task function, scored by the
    Fourier layer-line correlation instead of the sparse least-squares reconstruction:
    ``None`` for a blank image (pipeline.py:214-218), else
    ``(score, (None, None, None, None, D2d, D3d, L2d, L3d), (data, imageFile, imageIndex, apix3d,
    apix2d, twist, rise, csym, tilt, psi, dy))``.  ``algorithm`` may carry ``helical_diameter``,
    ``ball_radius``, ``mask``, ``log`` and ``device``.  ``imageIndex`` is 1-based like the reference's
    (pipeline.py:212 reads ``imageIndex - 1``).  ``target_apix2d`` below the image's own pixel size means
    "no rescale" (pipeline.py:268-272, filters.py:393-409); a larger one needs scikit-image's ``rescale``
    and is rejected, like denoise / horizontalize / auto tube diameter.

    ``algorithm["scorer"] = "lsq"`` runs the reference's OWN scorer instead (pipeline.py:286-496): the sparse
    least-squares reconstruction on the device (``helicon_amd.lsq_reconstruct``, model "lsq", interpolation "nn" or
    "linear"), its cosine score, the helically symmetrised map (``apply_helical_symmetry``) and its x / y projections
    and central z sections — the reference's complete return tuple (with tilt / psi / dy the projections are taken
    from the map resampled by ``transform_map``, like the reference's).

    The reference's pool calls this once per (twist, rise) pair with the same image (app.py:2473-2476): the
    prepared image is cached by content, and the engine keeps the reference spectrum it was last given, so
    repeated calls cost one 1-candidate sweep each."""
    if data is None:  # pipeline.py:211-212
        from .mrc import read_image_2d

        data = read_image_2d(imageFile, imageIndex - 1)
    data = np.asarray(data)
    dkey = _digest(data)
    with _prepared_lock:
        blank = _blank.get(dkey)
    if blank is None:
        blank = bool(np.std(data) == 0)
        with _prepared_lock:
            if len(_blank) >= 4 * _PREPARED_MAX:
                _blank.clear()
            _blank[dkey] = blank
    if blank:  # pipeline.py:214-218
        return None
    if denoise:
        raise NotImplementedError("denoise (scikit-image's restoration filters, pipeline.py:189-201) is outside the accelerated path")
    apix = float(apix2d_orig)
    if target_apix2d is None or target_apix2d < apix:  # pipeline.py:268-269
        target_apix2d = apix
    target_apix2d = float(target_apix2d)
    opts = dict(algorithm or {})
    device = int(opts.get("device", 0))
    mask = opts.get("mask")
    log = bool(opts.get("log", True))
    bkey = (dkey, apix, None if low_pass is None else float(low_pass), None if transpose is None else int(np.sign(transpose)),
            bool(horizontalize), device)
    orig_shape = None
    if horizontalize or target_apix2d != apix or (tube_diameter is not None and tube_diameter < 0):
        # the image after prepare_data (pipeline.py:220-231), kept by content: its shape is the "original" one of
        # pipeline.py:230-231 and the automatic tube diameter is estimated on it (:233-240)
        with _prepared_lock:
            base = _prepared.get(("base", bkey))
        if base is None:
            base = _prepare_base(data, apix, low_pass, transpose, horizontalize, device)
            with _prepared_lock:
                if len(_prepared) >= _PREPARED_MAX:
                    _prepared.pop(next(iter(_prepared)))
                _prepared[("base", bkey)] = base
        orig_shape = tuple(base.shape)
        if tube_diameter is not None and tube_diameter < 0:
            _rot, _shift, diameter_px = estimate_helix_rotation_center_diameter(base, device=device)
            tube_diameter = int(min(orig_shape[0], diameter_px) * apix * 2.5)
    pkey = (bkey, target_apix2d, None if thresh_fraction is None else float(thresh_fraction), float(tube_diameter))
    with _prepared_lock:
        prepared = _prepared.get(pkey)
    if prepared is None:
        prepared = _prepare_task_image(data, apix, low_pass, transpose, thresh_fraction, tube_diameter, device, horizontalize,
                                       target_apix2d)
        with _prepared_lock:
            if len(_prepared) >= _PREPARED_MAX:
                _prepared.pop(next(iter(_prepared)))
            _prepared[pkey] = prepared
    if opts.get("scorer", "spectrum") == "lsq":
        return _task_lsq(data, prepared, imageFile, imageIndex, twist, rise, rise_range, csym, tilt, tilt_range, psi, dy, apix,
                         target_apix3d, thresh_fraction, positive_constraint, tube_length, tube_diameter, tube_diameter_inner,
                         reconstruct_length, sym_oversample, interpolation, fsc_test, return_3d, score_metric, opts, device,
                         transpose, low_pass, horizontalize, target_apix2d, orig_shape, psi_range, dy_range)
    apix = target_apix2d   # the spectrum scorer works on the prepared image's own grid
    ny, nx = _image_shape(*prepared.shape)
    diameter = float(opts.get("helical_diameter", 0.4 * (tube_diameter if tube_diameter > 0 else ny * apix)))
    ball_radius = float(opts.get("ball_radius", 2.0 * apix))
    rkey = (pkey, None if mask is None else _digest(np.asarray(mask) != 0), log)
    eng = _engine((ny, nx), device)
    with eng.session():
        eng.set_geometry(apix=apix, helical_diameter=diameter, ball_radius=ball_radius, tilt=tilt, psi=psi, dy=dy)
        eng.set_reference(prepared, mask, log=log, key=rkey)
        score = float(eng.sweep(np.array([[twist, rise, csym, 0.0]]))[0, 0])
    apix3d = target_apix3d if (target_apix3d is not None and target_apix3d > 0) else apix
    return (
        score,
        (None, None, None, None, nx, ny, nx, nx),
        (prepared, imageFile, imageIndex, apix3d, apix, twist, rise, csym, tilt, psi, dy),
    )
