#!/usr/bin/env python3
"""Developer experiment: LM progress on the non-banded ring workload against the PCG tolerance."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from structure_from_motion_amd import _capi, synthetic
name = sys.argv[1] if len(sys.argv) > 1 else "ring1778_loop"
p = synthetic.make_config(name)
for tol, mx in ((1e-8, 500), (1e-8, 2000), (1e-2, 500), (1e-1, 500)):
    with _capi.Handle(_capi.Arrays.from_dict(p), loss="huber", f_scale=1.0, max_iter=300, pcg_tol=tol, pcg_max_iter=mx, linear_solver="pcg") as h:
        t0 = time.perf_counter(); costs = []
        for k in range(6):
            s = h.run(5); h.synchronize()
            costs.append("%.6e" % s["final_cost"])
            if s["status"] != 0: break
        dt = time.perf_counter() - t0
        print(f"tol {tol:g} max {mx}: steps {s['iterations']} accepted {s['accepted']} pcg {s['pcg_iterations']} status {s['status']} time {dt*1e3:.1f} ms costs {costs}")
