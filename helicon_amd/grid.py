"""Host-side logic of the sweep driver: candidate grid, candidate filter, Fourier masks.

Mirrors the headless part of the reference's ``run_denovo3D_reconstruction``
(src/helicon/webApps/denovo3D/app.py:2286-2452): axes are ``np.arange(min, max + step/2, step)``
(app.py:2319-2334), candidates are ``itertools.product(twists, rises)`` (app.py:2336-2338) for one
csym at a time, twist is wrapped to [-180, 180] and rounded to 6 decimals (app.py:2360) and
candidates with a tiny twist / tiny rise / too large rise are skipped (app.py:2389-2403).
"""
from __future__ import annotations

import itertools
import math
from dataclasses import dataclass

import numpy as np

__all__ = [
    "set_to_periodic_range",
    "sweep_axis",
    "CandidateGrid",
    "build_grid",
    "radial_band_mask",
    "layer_line_mask",
    "shard_bounds",
]


def set_to_periodic_range(v: float, min: float = -180, max: float = 180) -> float:
    """src/helicon/lib/angular.py:84-110."""
    if min <= v <= max:
        return v
    tmp = math.fmod(v - min, max - min)
    if tmp >= 0:
        tmp += min
    else:
        tmp += max
    return tmp


def sweep_axis(vmin: float, vmax: float, step: float) -> np.ndarray:
    if vmin < vmax:
        return np.arange(vmin, vmax + step / 2, step)
    return np.array([vmin], dtype=np.float64)


@dataclass
class CandidateGrid:
    """Flat candidate list, csym-major then twist then rise: g = (c*T + t)*R + r."""

    twists: np.ndarray
    rises: np.ndarray
    csyms: np.ndarray
    params: np.ndarray  # [G, 4] float64: twist (wrapped, rounded), rise, csym, rot
    valid: np.ndarray   # [G] bool — False where the reference's driver skips the pair

    @property
    def shape(self):
        return (len(self.csyms), len(self.twists), len(self.rises))

    def __len__(self):
        return len(self.params)

    def unravel(self, g: int):
        c, t, r = np.unravel_index(int(g), self.shape)
        return int(c), int(t), int(r)


def build_grid(twists, rises, csyms=(1,), *, tube_length: float, rot: float = 0.0) -> CandidateGrid:
    twists = np.atleast_1d(np.asarray(twists, dtype=np.float64))
    rises = np.atleast_1d(np.asarray(rises, dtype=np.float64))
    csyms = np.atleast_1d(np.asarray(csyms, dtype=np.int64))
    if (csyms < 1).any():
        raise ValueError("csym must be >= 1")
    tw = np.array([np.round(set_to_periodic_range(float(t), min=-180, max=180), 6) for t in twists])
    pairs = np.array(list(itertools.product(tw, rises)), dtype=np.float64).reshape(-1, 2)
    ok = ~((np.abs(pairs[:, 0]) < 0.01) | (np.abs(pairs[:, 1]) < 0.01) | (np.abs(pairs[:, 1]) >= tube_length / 2))
    n = len(pairs)
    params = np.empty((len(csyms) * n, 4), dtype=np.float64)
    for k, c in enumerate(csyms):
        params[k * n:(k + 1) * n, 0:2] = pairs
        params[k * n:(k + 1) * n, 2] = float(c)
    params[:, 3] = rot
    return CandidateGrid(twists, rises, csyms, params, np.tile(ok, len(csyms)))


def radial_band_mask(ny: int, nx: int, r_lo: float = 2.0, r_hi: float | None = None) -> np.ndarray:
    """``r_lo < r < r_hi`` on the fftshifted plane, r in pixels from DC at [ny//2, nx//2];
    default ``r_hi = min(ny, nx)//2 - 1``."""
    if r_hi is None:
        r_hi = min(ny, nx) // 2 - 1
    ky = (np.arange(ny) - ny // 2).astype(np.float64)
    kx = (np.arange(nx) - nx // 2).astype(np.float64)
    r2 = ky[:, None] ** 2 + kx[None, :] ** 2
    return (r2 > r_lo * r_lo) & (r2 < r_hi * r_hi)


def layer_line_mask(ny, nx, r_lo=2.0, r_hi=None, axial_bins=None, half_width=1) -> np.ndarray:
    """Radial band intersected with layer lines.  The helical axis is the image column axis, so a
    layer line is a COLUMN of the fftshifted plane at axial frequency ``+-axial_bins[j]`` bins."""
    m = radial_band_mask(ny, nx, r_lo, r_hi)
    if axial_bins is None:
        return m
    kx = np.abs(np.arange(nx) - nx // 2)
    sel = np.zeros(nx, dtype=bool)
    for b in axial_bins:
        sel |= np.abs(kx - int(b)) <= half_width
    return m & sel[None, :]


def shard_bounds(n_items: int, rank: int, world: int, align: int = 1) -> tuple[int, int, int]:
    """Contiguous block partition of the flat candidate index:
    rank k owns [k*per, min(G, (k+1)*per)), per = ceil(G/W) rounded up to a multiple of ``align``
    (the number of rises: shards then start on a twist, which keeps every shard a list of whole
    shared-twist runs for the library's fused pass).  Returns (lo, hi, per_rank)."""
    per = -(-n_items // world)
    if align > 1:
        per = -(-per // align) * align
    lo = min(n_items, rank * per)
    hi = min(n_items, lo + per)
    return lo, hi, per
