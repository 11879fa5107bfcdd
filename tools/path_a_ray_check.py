import sys, ctypes as C
sys.path.insert(0, '/root/repo')
from helicon_amd import _lib
L = _lib.lib()
dev = int(sys.argv[1]) if len(sys.argv) > 1 else -1
for tw, rs, cs, scale, dy in ((29.0, 2.0, 1, 1.0, 0.0), (27.5, 1.7, 1, 1.0, 0.0), (58.0, 4.0, 2, 1.0, 0.5), (29.0, 2.0, 1, 0.8, 0.0)):
    q = _lib.hh_pa_params(scale, tw, rs, cs, 0.0, 0.0, dy, 20, 32, 20, 0, 6, 960, 960, 1, 0, 0)
    out = (C.c_int64 * 8)()
    rc = L.hh_pab_check_ray_arithmetic(C.byref(q), 20, 32, dev, out)
    print(tw, rs, cs, scale, dy, rc, list(out))
