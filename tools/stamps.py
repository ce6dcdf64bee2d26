#!/usr/bin/env python3
"""Developer diagnostic: per-phase cycle shares of the fused build kernel (stamped variant)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from structure_from_motion_amd import _capi, synthetic  # noqa: E402

p = synthetic.make_config(sys.argv[1] if len(sys.argv) > 1 else "venice1778")
h = _capi.Handle(_capi.Arrays.from_dict(p))
h.build(1e-4)
n = _capi.analyze(h.arr)["n_chunks"]
buf = np.zeros((n, 12), dtype=np.uint64)
lib = _capi.load()
lib.ba_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
_capi.check(lib.ba_debug_stamps(h.h, buf.ctypes.data, buf.size))
d = np.diff(buf[:, :11].astype(np.int64), axis=1)
names = ["zero+stage+barrier", "phase1 load+linearize", "seg-reduce V + barrier", "1b gather U", "2a cholesky",
         "block reductions", "2b T + barrier", "-", "3 schur + barrier", "4 flush"]
tot = d.sum(1)
print(f"chunks {n}; cycles per chunk (s_memtime ticks @100MHz? see guide): median total {np.median(tot):.0f}")
for i, nm in enumerate(names[:d.shape[1]]):
    print(f"  {nm:26s} median {np.median(d[:, i]):9.0f}  mean {d[:, i].mean():9.0f}  share {d[:, i].sum() / tot.sum() * 100:5.1f} %")
