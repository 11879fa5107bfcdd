#!/usr/bin/env python3
"""Evidence for DESIGN.md section 6 ("one HIP runtime per process"): with HELICON_HIP_RUNTIME=system the library binds
the system libamdhip64.so.7 and a LATER `import torch` maps torch's own copy beside it; the second runtime to
initialise cannot open the device.  Prints what each order maps and what torch then says.  Run once, in a fresh
process per case (this script starts them)."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CODE = r"""
import sys
sys.path.insert(0, %r)
import helicon_amd as H
from helicon_amd import _lib
e = H.SweepEngine(64); e.set_geometry(apix=2.0, helical_diameter=40.0, ball_radius=4.0); e.simulate(29.0, 10.0, 1); e.close()
print("after the library :", _lib.hip_runtime_paths(), flush=True)
import torch
print("after import torch:", _lib.hip_runtime_paths(), flush=True)
try:
    print("torch allocation  :", torch.arange(8, device="cuda").float().sum().item(), flush=True)
except Exception as ex:
    print("torch allocation  : FAILED:", type(ex).__name__, str(ex).splitlines()[0], flush=True)
""" % str(ROOT)

for mode in ("system", "auto"):
    env = dict(os.environ, HELICON_HIP_RUNTIME=mode)
    print(f"--- HELICON_HIP_RUNTIME={mode} (library first, torch second) ---", flush=True)
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=600)
    print(out.stdout, end="")
    if out.returncode:
        print("exit code", out.returncode, out.stderr[-800:])
