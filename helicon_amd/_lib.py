"""ctypes binding of libhelicon_hip.so (include/helicon_hip.h).

There is deliberately no CPU fallback: if the shared library is missing, or no gfx950 device
is visible, every entry point raises ``HeliconHipError``.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

__all__ = ["HeliconHipError", "lib", "lib_path", "hip_runtime_paths", "hh_geom", "hh_profile", "hh_pa_params", "check", "EXPORTS"]

_HERE = Path(__file__).resolve().parent


class HeliconHipError(RuntimeError):
    """A libhelicon_hip call failed (message from ``hh_last_error``)."""


class hh_geom(C.Structure):
    _fields_ = [
        ("apix", C.c_double),
        ("helical_diameter", C.c_double),
        ("ball_radius", C.c_double),
        ("tilt", C.c_double),
        ("psi", C.c_double),
        ("dy", C.c_double),
        ("n_units", C.c_int32),
        ("tail_bits", C.c_int32),
        ("units", C.POINTER(C.c_double)),
    ]


class hh_profile(C.Structure):
    _fields_ = [
        ("ms_first_pass", C.c_double),
        ("ms_second_pass", C.c_double),
        ("ms_finalize", C.c_double),
        ("ms_centres", C.c_double),
        ("n_first_pass", C.c_int64),
        ("n_second_pass", C.c_int64),
        ("n_finalize", C.c_int64),
        ("n_centres", C.c_int64),
        ("candidates", C.c_int64),
    ]


class hh_pa_params(C.Structure):
    _fields_ = [
        ("scale2d_to_3d", C.c_double), ("twist_degree", C.c_double), ("rise_pixel", C.c_double),
        ("csym", C.c_int32),
        ("tilt_degree", C.c_double), ("psi_degree", C.c_double), ("dy_pixel", C.c_double),
        ("reconstruct_diameter_2d_pixel", C.c_int32), ("reconstruct_length_2d_pixel", C.c_int32),
        ("reconstruct_diameter_3d_pixel", C.c_int32), ("reconstruct_diameter_3d_inner_pixel", C.c_int32),
        ("reconstruct_length_3d_pixel", C.c_int32),
        ("min_projection_lines", C.c_int64), ("min_sym_pairs", C.c_int64),
        ("interpolation", C.c_int32), ("fsc_mode", C.c_int32), ("fsc_half", C.c_int32),
        ("n_fsc_ids", C.c_int32), ("fsc_ids", C.POINTER(C.c_int32)),
    ]


_ctx = C.c_void_p
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)

# name -> (restype, argtypes); exactly the symbols include/helicon_hip.h declares
EXPORTS = {
    "hh_abi_version": (C.c_int, []),
    "hh_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "hh_selftest_exception": (C.c_int, [C.c_int]),
    "hh_create": (C.c_int, [C.POINTER(_ctx), C.c_int, C.c_int, C.c_int]),
    "hh_create2": (C.c_int, [C.POINTER(_ctx), C.c_int, C.c_int, C.c_int, C.c_int]),
    "hh_destroy": (None, [_ctx]),
    "hh_max_batch": (C.c_int, [_ctx]),
    "hh_memory_bytes": (C.c_int64, [_ctx, C.POINTER(C.c_int64)]),
    "hh_last_error": (C.c_char_p, [_ctx]),
    "hh_set_stream": (C.c_int, [_ctx, C.c_void_p]),
    "hh_use_own_stream": (C.c_int, [_ctx]),
    "hh_set_geometry": (C.c_int, [_ctx, C.POINTER(hh_geom)]),
    "hh_set_reference": (C.c_int, [_ctx, _f32p, C.c_int, C.POINTER(C.c_uint8), C.c_int]),
    "hh_sweep": (C.c_int, [_ctx, _f64p, C.c_int64, _f32p]),
    "hh_sweep_device": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_void_p]),
    "hh_sweep_device_mirrored": (C.c_int, [_ctx, C.c_void_p, _f64p, C.c_int64, C.c_void_p]),
    "hh_sweep_device_strided": (C.c_int, [_ctx, C.c_void_p, _f64p, C.c_int64, C.c_void_p, C.c_int64]),
    "hh_set_table_path": (C.c_int, [_ctx, C.c_int]),
    "hh_last_first_pass": (C.c_int, [_ctx]),
    "hh_low_high_pass_filter": (C.c_int, [_ctx, _f32p, C.c_double, C.c_double, _f32p]),
    "hh_threshold_data": (C.c_int, [_ctx, _f32p, C.c_int64, C.c_int, C.c_double, _f32p]),
    "hh_argmax": (C.c_int, [_f32p, C.c_int64, C.POINTER(C.c_int64)]),
    "hh_argmax_device": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.POINTER(C.c_int64)]),
    "hh_comm_unique_id": (C.c_int, [C.c_void_p]),
    "hh_comm_init": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_void_p]),
    "hh_allgather": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_void_p]),
    "hh_comm_destroy": (C.c_int, [_ctx]),
    "hh_simulate": (C.c_int, [_ctx, _f64p, _f32p]),
    "hh_power_spectrum": (C.c_int, [_ctx, _f32p, C.c_int, _f32p, _f32p]),
    "hh_power_spectrum_zoom": (C.c_int, [C.c_int, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                        C.c_int, _f32p, _f32p]),
    "hh_cross_correlation": (C.c_int, [_ctx, _f32p, _f32p, C.c_int64, _f64p]),
    "hh_cosine_similarity": (C.c_int, [_ctx, _f32p, _f32p, C.c_int64, _f64p]),
    "hh_cross_correlation_f64": (C.c_int, [_ctx, _f64p, _f64p, C.c_int64, _f64p]),
    "hh_cosine_similarity_f64": (C.c_int, [_ctx, _f64p, _f64p, C.c_int64, _f64p]),
    "hh_fused_schedule": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)]),
    "hh_general_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "hh_affine_transform_2d": (C.c_int, [C.c_int, _f32p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), _f32p]),
    "hh_affine_transform_2d_cubic": (C.c_int, [C.c_int, _f32p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), _f32p]),
    "hh_warp_affine_2d": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int, C.c_double, C.c_int,
                                    C.c_void_p]),
    "hh_rescale_2d": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hh_ssim_2d": (C.c_int, [C.c_int, _f32p, _f32p, C.c_int, C.c_int, C.c_double, _f64p]),
    "hh_joint_histogram": (C.c_int, [C.c_int, _f32p, _f32p, C.c_int64, _f64p, _f64p, C.c_int, C.POINTER(C.c_int64)]),
    "hh_helix_moments": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, _f64p]),
    "hh_transform_map": (C.c_int, [C.c_int, _f32p, C.POINTER(C.c_int32), C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                   C.c_double, C.c_double, _f32p]),
    "hh_apply_helical_symmetry": (C.c_int, [C.c_int, _f32p, C.POINTER(C.c_int32), C.c_double, C.c_double, C.c_double,
                                            C.c_int, C.c_double, C.POINTER(C.c_int32), C.c_double, _f32p,
                                            C.POINTER(C.c_int32), _f64p]),
    "hh_synchronize": (C.c_int, [_ctx]),
    "hh_profile_enable": (C.c_int, [_ctx, C.c_int]),
    "hh_profile_reset": (C.c_int, [_ctx]),
    "hh_profile_get": (C.c_int, [_ctx, C.POINTER(hh_profile)]),
    "hh_calibrate_traffic": (C.c_int, [_ctx, C.c_int, C.c_int64]),
    "hh_algorithmic_bytes": (C.c_int64, [C.c_int]),
    # Path A slice (helicon_amd/solver.py)
    "hh_pa_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, _f32p, C.c_int, C.c_int, C.POINTER(hh_pa_params)]),
    "hh_pa_destroy": (None, [C.c_void_p]),
    "hh_pa_last_error": (C.c_char_p, [C.c_void_p]),
    "hh_pa_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "hh_pa_get_rhs": (C.c_int, [C.c_void_p, _f32p, C.POINTER(C.c_int32)]),
    "hh_pa_matvec": (C.c_int, [C.c_void_p, _f64p, _f64p, _f64p, _f64p]),
    "hh_pa_rmatvec": (C.c_int, [C.c_void_p, _f64p, _f64p, _f64p, _f64p]),
    "hh_pa_lsmr": (C.c_int, [C.c_void_p, _f64p, _f64p, _f64p, C.c_double, C.c_double, C.c_double, C.c_int, _f64p,
                             C.POINTER(C.c_int), _f64p]),
    # Path A, many candidates at once (helicon_amd/solver.py: lsq_reconstruct_batch)
    "hh_pab_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, _f32p, C.c_int, C.c_int, C.POINTER(hh_pa_params), C.c_int]),
    "hh_pab_destroy": (None, [C.c_void_p]),
    "hh_pab_last_error": (C.c_char_p, [C.c_void_p]),
    "hh_pab_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "hh_pab_get_rhs": (C.c_int, [C.c_void_p, C.c_int, _f32p, C.POINTER(C.c_int32)]),
    "hh_pab_get_pairs": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int32)]),
    "hh_pab_solve": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_double, C.c_int, C.c_int, _f32p, _f64p,
                               C.POINTER(C.c_int32)]),
    "hh_pab_check_ray_arithmetic": (C.c_int, [C.POINTER(hh_pa_params), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "hh_pab_matvec": (C.c_int, [C.c_void_p, C.c_int, _f64p, _f64p]),
    "hh_pab_rmatvec": (C.c_int, [C.c_void_p, C.c_int, _f64p, _f64p]),
    "hh_pab_solve_prox": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _f64p, C.c_double, C.c_int, C.c_double, C.c_int,
                                    _f32p, _f64p, C.POINTER(C.c_int32), _f64p]),
    "hh_pab_counters": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
}

_lib = None
_runtime = None  # path of the HIP runtime this process is bound to (diagnostics: hip_runtime_path())


def lib_path() -> Path:
    env = os.environ.get("HELICON_HIP_LIB")
    return Path(env) if env else _HERE / "libhelicon_hip.so"


def _mapped(fragment: str):
    """Paths of the shared objects mapped into this process whose name contains `fragment`."""
    found = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(None, 1)[-1]
                if fragment in path and path not in found:
                    found.append(path)
    except OSError:
        pass
    return found


def _bind_hip_runtime() -> None:
    """One HIP runtime per process, whatever the import order.

    libhelicon_hip.so asks the loader for ``libamdhip64.so.7`` (the soname).  PyTorch's ROCm wheels ship their OWN copy of
    the runtime in ``torch/lib`` and ask for it as ``libamdhip64.so`` (and ``libhsa-runtime64.so``).  glibc matches a
    request against loaded objects by requested name or soname, so:

    * torch first, this library second: ``libamdhip64.so.7`` matches the soname of torch's copy — one runtime;
    * this library first, ``import torch`` later: ``libamdhip64.so`` matches neither the name nor the soname of the
      system copy already loaded, torch maps its own — TWO HIP runtimes and TWO HSA runtimes in the process.  The second
      one to initialise cannot open the device, and torch reports "No HIP GPUs are available" (round 2 saw exactly that
      in test sessions whose first torch.cuda use came after the library's; tests/conftest.py papered over it by
      initialising torch first).

    So before the library is loaded, and only if no HIP runtime is mapped yet, the runtime of an INSTALLED torch (found
    without importing it) is loaded by path: both this library (by soname) and a later ``import torch`` (same file: the
    loader recognises the inode) then bind to it.  ``HELICON_HIP_RUNTIME=system`` keeps the system runtime instead,
    ``HELICON_HIP_RUNTIME=/path/to/libamdhip64.so`` names one.
    """
    global _runtime
    mapped = _mapped("libamdhip64")
    if mapped:  # someone (torch, another extension) got there first: the soname match binds us to it
        _runtime = mapped[0]
        return
    choice = os.environ.get("HELICON_HIP_RUNTIME", "auto")
    if choice == "system":
        return
    path = None
    if choice not in ("auto", "torch"):
        path = Path(choice)
    else:
        import importlib.util

        try:
            spec = importlib.util.find_spec("torch")  # locates the package, does not import it
        except (ImportError, ValueError):
            spec = None
        if spec is not None and spec.origin:
            cand = Path(spec.origin).resolve().parent / "lib" / "libamdhip64.so"
            if cand.exists():
                path = cand
    if path is None:
        return
    # the preload only helps if the loader will match libhelicon_hip.so's request (DT_NEEDED libamdhip64.so.N) against
    # this file's SONAME; a wheel that bundles another ROCm major would leave the library to map the system runtime as
    # well — the two-runtime state this function exists to prevent — so such a candidate is skipped
    want, have = _needed_hip_soname(lib_path()), _soname(path)
    if choice in ("auto", "torch") and want and have and want != have:
        return
    try:
        C.CDLL(str(path), mode=C.RTLD_GLOBAL)
        _runtime = str(path)
    except OSError as e:
        raise HeliconHipError(f"cannot load the HIP runtime {path}: {e} (set HELICON_HIP_RUNTIME=system to use "
                              "the one libhelicon_hip.so was linked against)") from e


def _dynamic_strings(path, tag: int):
    """Strings of the ELF dynamic section entries with the given tag (1 = DT_NEEDED, 14 = DT_SONAME); [] if unreadable."""
    import struct

    try:
        data = Path(path).read_bytes()
        if data[:4] != b"\x7fELF" or data[4] != 2 or data[5] != 1:   # 64-bit little-endian only
            return []
        shoff, = struct.unpack_from("<Q", data, 0x28)
        shentsize, shnum = struct.unpack_from("<HH", data, 0x3A)
        sections = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
        out = []
        for sec in sections:
            if sec[1] != 6:   # SHT_DYNAMIC
                continue
            strtab = sections[sec[6]]
            for off in range(sec[4], sec[4] + sec[5], 16):
                t, v = struct.unpack_from("<qQ", data, off)
                if t == 0:
                    break
                if t == tag:
                    start = strtab[4] + v
                    out.append(data[start: data.index(b"\0", start)].decode())
        return out
    except (OSError, struct.error, ValueError, IndexError):
        return []


def _soname(path):
    names = _dynamic_strings(path, 14)
    return names[0] if names else None


def _needed_hip_soname(path):
    for name in _dynamic_strings(path, 1):
        if name.startswith("libamdhip64"):
            return name
    return None


def hip_runtime_paths():
    """Every libamdhip64 mapped into the process right now (more than one entry is the two-runtime condition)."""
    return _mapped("libamdhip64")


def lib():
    """Load (once) and return the shared library with prototypes attached."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not path.exists():
        raise HeliconHipError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); helicon_amd has no CPU fallback"
        )
    _bind_hip_runtime()
    try:
        handle = C.CDLL(str(path))
    except OSError as e:  # e.g. libamdhip64 missing
        raise HeliconHipError(f"cannot load {path}: {e}") from e
    for name, (res, args) in EXPORTS.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise HeliconHipError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if handle.hh_abi_version() != 1:
        raise HeliconHipError(f"{path}: ABI version {handle.hh_abi_version()} != 1")
    runtimes = {os.path.realpath(p) for p in hip_runtime_paths()}
    if len(runtimes) > 1:   # two HIP runtimes in one process: the second to initialise finds no device
        raise HeliconHipError(f"two HIP runtimes are mapped into this process ({sorted(runtimes)}): set HELICON_HIP_RUNTIME=system "
                              "(or to the runtime's path) so that libhelicon_hip.so and every other extension share one")
    _lib = handle
    return _lib


def check(rc: int, ctx=None) -> None:
    if rc == 0:
        return
    msg = lib().hh_last_error(ctx)
    kind = {-1: ValueError, -3: HeliconHipError}.get(rc, HeliconHipError)
    raise kind(f"libhelicon_hip error {rc}: {msg.decode() if msg else '?'}")
