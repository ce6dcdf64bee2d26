#!/usr/bin/env python3
"""Developer: LM steps on the Ladybug shape with the blocked dense solver forced (for rocprofv3)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from structure_from_motion_amd import _capi, synthetic
p = synthetic.make_config(sys.argv[1] if len(sys.argv) > 1 else "ladybug1723")
with _capi.Handle(_capi.Arrays.from_dict(p), loss="huber", f_scale=1.0, linear_solver="dense_cholesky") as h:
    s = h.run(1); h.synchronize()
    t0 = time.perf_counter(); s = h.run(3); h.synchronize(); dt = time.perf_counter() - t0
    print("steps", s["iterations"], "ms per LM step %.2f" % (dt * 1e3 / 3), "pcg", s["pcg_iterations"])
