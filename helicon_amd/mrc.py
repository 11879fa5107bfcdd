"""Minimal MRC2014 reader/writer for the headless driver (no ``mrcfile`` dependency).

Stands in for the two I/O calls on the sweep's edge: ``read_image_2d`` (src/helicon/lib/io_mrc.py:
71-100, used by pipeline.py:211-212 when a task carries a file name instead of an array) and the map
download of the app (app.py:1279-1287).  Supports the header fields those need: nx, ny, nz, mode
(0 int8, 1 int16, 2 float32, 6 uint16, 12 float16), the extended-header size, the pixel size, and both
byte orders (machine stamp)."""
from __future__ import annotations

import struct
from pathlib import Path

import numpy as np

_MODES = {0: np.int8, 1: np.int16, 2: np.float32, 6: np.uint16, 12: np.float16}


def _header(path):
    with open(path, "rb") as f:
        raw = f.read(1024)
    if len(raw) < 1024:
        raise OSError(f"{path}: not an MRC file (short header)")
    stamp = raw[212:214]
    order = ">" if stamp[:1] == b"\x11" else "<"
    nx, ny, nz, mode = struct.unpack(order + "4i", raw[0:16])
    mx, my, mz = struct.unpack(order + "3i", raw[28:40])
    cella = struct.unpack(order + "3f", raw[40:52])
    nsymbt = struct.unpack(order + "i", raw[92:96])[0]
    if mode not in _MODES or min(nx, ny, nz) <= 0:
        raise OSError(f"{path}: unsupported MRC mode {mode} or bad dimensions {(nx, ny, nz)}")
    apix = cella[0] / mx if mx > 0 else 0.0
    return dict(nx=nx, ny=ny, nz=nz, mode=mode, nsymbt=max(nsymbt, 0), order=order, apix=float(apix))


def image_shape(path) -> tuple[int, int, int]:
    """(nx, ny, nz) like io_mrc.py:55-68."""
    h = _header(path)
    return h["nx"], h["ny"], h["nz"]


def read_mrc(path) -> tuple[np.ndarray, float]:
    """Whole file as a [nz, ny, nx] array (memory-mapped) and the pixel size in Angstrom."""
    h = _header(path)
    dt = np.dtype(_MODES[h["mode"]]).newbyteorder(h["order"])
    data = np.memmap(path, dtype=dt, mode="r", offset=1024 + h["nsymbt"], shape=(h["nz"], h["ny"], h["nx"]))
    return data, h["apix"]


def read_image_2d(imageFile, i: int) -> np.ndarray:
    """Slice ``i`` of an MRC stack (io_mrc.py:71-100); out-of-range indices raise like the reference."""
    if not Path(imageFile).exists():
        raise OSError(f"cannot find image file {imageFile}")
    data, _ = read_mrc(imageFile)
    i = int(i)
    if not 0 <= i < data.shape[0]:
        raise OSError(f"the requested image {i} is out of the valid range [0, {data.shape[0]}) for image file {imageFile}")
    return np.asarray(data[i])


def write_mrc(path, data, apix: float = 1.0) -> None:
    """float32 [ny, nx] image or [nz, ny, nx] volume/stack, MRC2014 little-endian, mode 2."""
    a = np.ascontiguousarray(data, dtype="<f4")
    if a.ndim == 2:
        a = a[None]
    nz, ny, nx = a.shape
    hdr = bytearray(1024)
    struct.pack_into("<4i", hdr, 0, nx, ny, nz, 2)
    struct.pack_into("<3i", hdr, 28, nx, ny, nz)
    struct.pack_into("<3f", hdr, 40, nx * apix, ny * apix, nz * apix)
    struct.pack_into("<3f", hdr, 52, 90.0, 90.0, 90.0)
    struct.pack_into("<3i", hdr, 64, 1, 2, 3)
    struct.pack_into("<3f", hdr, 76, float(a.min()), float(a.max()), float(a.mean()))
    hdr[208:212] = b"MAP "
    hdr[212:216] = b"\x44\x44\x00\x00"
    struct.pack_into("<f", hdr, 216, float(a.std()))
    with open(path, "wb") as f:
        f.write(bytes(hdr))
        f.write(a.tobytes())
