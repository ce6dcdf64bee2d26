"""Developer: the LM paths of AUTO (blocked dense) and forced PCG on a small object-centric problem."""
import sys
import numpy as np
sys.path.insert(0, '/root/repo')
from structure_from_motion_amd import _capi, synthetic
p = synthetic.make_ring(300, 12000, seed=77, K=(1200.0, 1200.0, 960.0, 540.0), focal_jitter=0.05, noise=0.5, outlier_frac=0.01,
                        visibility="random", mean_track=7.0, max_track=20, n_fixed=2, radius=6.0)
for solver, kw in (("auto", {}), ("pcg", dict(pcg_tol=1e-10, pcg_max_iter=2000)), ("dense_cholesky", {})):
    arr = _capi.Arrays.from_dict(p)
    s = _capi.solve(arr, loss="huber", f_scale=1.0, max_iter=100, linear_solver=solver, **kw)
    print(solver, {k: s[k] for k in ("status", "iterations", "accepted", "pcg_iterations", "initial_cost", "final_cost", "t_solve_ms")})
