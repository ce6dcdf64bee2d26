import sys, time; sys.path.insert(0,'/root/repo')
from structure_from_motion_amd import _capi, synthetic
for name in ("ring1778_revisit", "ring1778_loop", "orbit1024"):
    p = synthetic.make_config(name)
    for tol, solver in ((1e-8, "auto"), (1e-6, "auto"), (5e-3, "auto"), (1e-8, "pcg"), (5e-3, "pcg")):
        if name == "ring1778_loop" and solver == "auto": continue
        arr = _capi.Arrays.from_dict(p)
        with _capi.Handle(arr, loss="huber", f_scale=1.0, max_iter=60, pcg_tol=tol, linear_solver=solver) as h:
            h.run(1); h.synchronize()
            t0 = time.perf_counter(); s = h.run(59); h.synchronize(); dt = time.perf_counter() - t0
        print("%-18s %-5s tol %.0e: status %d, %2d LM steps, %5d CG its, %.1f ms total, %.2f ms/step, cost %.8e" %
              (name, solver, tol, s["status"], s["iterations"], s["pcg_iterations"], dt*1e3, dt*1e3/max(s["iterations"]-1,1), s["final_cost"]), flush=True)
