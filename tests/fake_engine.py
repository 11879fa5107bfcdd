"""A stand-in for helicon_amd.SweepEngine that needs no GPU: TEST INFRASTRUCTURE for the launch / shard /
all-gather / arg-max plumbing of bench.py and helicon_amd.distributed (world-size-2 gloo tests on CPU).
Its "scores" are a closed-form function of the candidate parameters, so any rank count must reproduce the
single-rank result bit for bit.  It reads and writes the raw pointers it is given, like the library does."""
import ctypes as C

import numpy as np


def fake_scores(params: np.ndarray, segment: int = 0) -> np.ndarray:
    tw, rs, cs, rot = params[:, 0], params[:, 1], params[:, 2], params[:, 3]
    s = np.exp(-((tw - 1.2) / 0.7) ** 2 - ((rs - 4.75) / 0.4) ** 2) / cs + 1e-3 * np.cos(3.0 * tw + rs + 0.1 * segment)
    return (s - 1e-4 * np.abs(rot)).astype(np.float32)


class FakeEngine:
    def __init__(self, n, device=0, max_batch=0):
        self.n, self.device, self.max_batch = n, device, max_batch or 256
        self.n_segments = 0
        self.last_first_pass = "fake"

    def set_geometry(self, **kw):
        pass

    def simulate(self, twist, rise, csym, rot=0.0):
        y, x = np.mgrid[: self.n, : self.n]
        return np.cos(0.1 * x + 0.01 * twist * y).astype(np.float32)

    def set_reference(self, images, mask=None, log=True, key=None):
        images = np.asarray(images)
        self.n_segments = 1 if images.ndim == 2 else images.shape[0]

    def set_table_path(self, mode=2):
        pass

    def set_stream(self, s):
        pass

    def sweep_device(self, d_params, n_candidates, d_scores, host_params=None, ld_scores=0):
        p = np.ctypeslib.as_array(C.cast(d_params, C.POINTER(C.c_double)), shape=(n_candidates, 4))
        ld = ld_scores or n_candidates
        out = np.ctypeslib.as_array(C.cast(d_scores, C.POINTER(C.c_float)), shape=(max(self.n_segments, 1), ld))
        for s in range(max(self.n_segments, 1)):
            out[s, :n_candidates] = fake_scores(p, s)

    def sweep(self, params):
        p = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, 4)
        return np.stack([fake_scores(p, s) for s in range(max(self.n_segments, 1))])
