#!/usr/bin/env python3
"""Developer tool: solve one synthetic config on the GPU with the per-iteration table."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from structure_from_motion_amd import _capi, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("config")
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--max-iter", type=int, default=100)
ap.add_argument("--pcg-tol", type=float, default=1e-8)
ap.add_argument("--pcg-max-iter", type=int, default=500)
ap.add_argument("--verbose", type=int, default=2)
ap.add_argument("--solver", default="auto")
ap.add_argument("--n-fixed", type=int, default=1)
ap.add_argument("--track", type=float, default=5.0)
a = ap.parse_args()
if a.config.startswith("street:"):
    nc, npt, no, seed = (int(v) for v in a.config.split(":")[1:])
    p = synthetic.make_street(nc, npt, no, seed=seed)
elif a.config.startswith("ring:"):
    nc, npt, seed = (int(v) for v in a.config.split(":")[1:])
    p = synthetic.make_ring(nc, npt, seed=seed, mean_track=a.track, outlier_frac=0.02 if a.track == 4.0 else 0.01, n_fixed=a.n_fixed)
else:
    p = synthetic.make_config(a.config, scale=a.scale)
arr = _capi.Arrays.from_dict(p)
print(p["name"], _capi.analyze(arr))
t0 = time.time()
s = _capi.solve(arr, loss="huber", f_scale=1.0, max_iter=a.max_iter, verbose=a.verbose, pcg_tol=a.pcg_tol,
                pcg_max_iter=a.pcg_max_iter, linear_solver=a.solver)
print(s, f"wall {time.time() - t0:.2f}s")
