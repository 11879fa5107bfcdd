"""The random-walk asymmetric unit of ``simulate_helical_projection(polymer=1)``
(src/helicon/webApps/denovo3D/utils.py:125-136 over ``random_polymer``, :192-333), on the host.

A chain of C-alpha-like atoms (3.8 A apart, never closer than 80 % of that to any other atom or symmetry copy) is grown
inside a cylinder shell by a self-avoiding walk that prefers to keep its direction, flattened towards the x-y plane by
``planarity``; every atom is stored with its ``csym`` images about z.  The walk draws from NumPy's GLOBAL random state,
like the reference, so ``np.random.seed`` replays it draw for draw: the order of the draws and the floating-point
expressions that decide acceptance are the reference's (fixture G12 holds its outputs); the code is arranged around a
small walker object instead of nested closures.  The lattice and the raster then run on the device with these atoms as
the asymmetric unit (``hh_geom.units``).
"""
from __future__ import annotations

import numpy as np

__all__ = ["random_polymer", "polymer_units"]

BOND = 3.8          # Angstrom between consecutive atoms
CLEARANCE = 0.8     # closest approach, in bonds
ATTEMPTS = 10       # restarts of the chain, tries per atom, tries for the first atom


def _images(point: np.ndarray, csym: int) -> np.ndarray:
    """The point and its images under the csym-fold axis along z, as rows."""
    if csym <= 1:
        return point[None, :]
    from scipy.spatial.transform import Rotation

    rows = [point]
    for k in range(1, csym):
        rows.append(Rotation.from_euler("z", k * 360 / csym, degrees=True).apply(point))
    return np.vstack(rows)


def _distances(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    delta = a[:, None, :] - b[None, :, :]
    return np.sqrt(np.sum(delta**2, axis=-1))


def _clear_of(new: np.ndarray, placed: np.ndarray, limit: float) -> bool:
    """True when the new atoms keep the clearance among themselves and to everything placed so far."""
    if len(new) > 1:
        d = _distances(new, new)
        d[np.diag_indices_from(d)] = 1e10
        if np.any(d < limit):
            return False
    d = _distances(new, placed)
    if new.shape == placed.shape and np.allclose(new, placed):   # the first atom is checked against itself
        d[np.diag_indices_from(d)] = 1e10
    return not np.any(d < limit)


class _Walker:
    def __init__(self, rmin, rmax, csym, planarity):
        self.rmin, self.rmax, self.csym, self.planarity = rmin, rmax, csym, planarity

    def first_atom(self):
        radius = np.sqrt(np.random.uniform(self.rmin**2, self.rmax**2))
        phi = np.random.uniform(-np.pi, np.pi)
        return np.array([radius * np.sin(phi), radius * np.cos(phi), 0.0])

    def step_from(self, chain: np.ndarray) -> np.ndarray:
        """One more atom (with its images) a bond away from the chain's end; after ten tries outside the shell the
        last try is taken as it is."""
        tries = 1
        while True:
            out_of_plane = 90 * (1 - self.planarity)
            spread_z = np.abs(np.random.normal(0, out_of_plane / 3))
            spread_xy = 180 / 3
            tip = chain[-1, :]
            if len(chain) < 2:
                keep = tip * 0
            else:
                keep = tip - chain[-2, :]
                keep /= np.linalg.norm(keep)
                keep /= tries
                keep *= (self.rmax - np.linalg.norm(tip)) / self.rmax
            kick = np.random.normal(0, (spread_xy, spread_xy, spread_z))
            kick /= np.linalg.norm(kick)
            heading = (keep + kick) / np.linalg.norm(keep + kick)
            atom = tip + BOND * heading
            radius = np.linalg.norm(atom)
            if self.rmin <= radius <= self.rmax or tries > ATTEMPTS:
                return _images(atom, self.csym)
            tries += 1


def random_polymer(n_atoms=100, rmin=0, rmax=100, csym=1, planarity=0.9) -> np.ndarray:
    """utils.py:192-333: ``[atoms * csym, 3]`` coordinates (fewer atoms when the walk gets stuck ten times in a row)."""
    walker = _Walker(rmin, rmax, csym, planarity)
    limit = BOND * CLEARANCE
    placed_atoms = 0
    restarts = 0
    coords = np.zeros([csym * n_atoms, 3], dtype=float)
    while restarts < ATTEMPTS:
        coords = np.zeros([csym * n_atoms, 3], dtype=float)
        started = False
        for _ in range(ATTEMPTS):
            coords[0, :] = walker.first_atom()
            coords[0:csym, :] = _images(coords[0, :], csym)
            if _clear_of(coords[0:csym, :], coords[0:csym, :], limit):
                started = True
                placed_atoms = 1
                break
        if not started:
            break            # (the reference gives up here and returns what an earlier restart placed)
        for i in range(1, n_atoms):
            grown = False
            for _ in range(ATTEMPTS):
                chain = coords[: i * csym, :]
                images = walker.step_from(chain)
                if _clear_of(images, chain, limit):
                    coords[i * csym: (i + 1) * csym, :] = images
                    grown = True
                    placed_atoms = i + 1
                    break
            if not grown:
                break
        if placed_atoms == n_atoms:
            break
        restarts += 1
    return coords[: placed_atoms * csym, :]


def polymer_units(n, helical_diameter, csym, planarity) -> np.ndarray:
    """The asymmetric unit ``centers_0`` the lattice code receives for ``polymer=1`` (utils.py:125-136): the walk, turned by
    ``Ry(90)`` and with its axes reordered x, y, z -> z, y, x; columns (projection axis, image-row axis, helical axis)."""
    from scipy.spatial.transform import Rotation

    atoms = random_polymer(n_atoms=n, rmin=0, rmax=helical_diameter / 2, csym=csym, planarity=planarity)
    atoms = Rotation.from_euler("y", 90, degrees=True).apply(atoms)
    return atoms[:, [2, 1, 0]]
