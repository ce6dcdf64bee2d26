"""Developer: one workload for rocprofv3 (kernel trace or a PMC pass).
usage: prof_run.py <config> [det] [lm=<steps>] [solver=<name>] [builds=<n>] [schur_mode=auto|chunks|rows]"""
import sys
sys.path.insert(0, '/root/repo')
from structure_from_motion_amd import _capi, synthetic
name = sys.argv[1]
det = 1 if "det" in sys.argv[2:] else 0
kw = dict(a.split("=") for a in sys.argv[2:] if "=" in a)
p = synthetic.make_config(name)
opts = dict(loss="huber", f_scale=1.0, max_iter=300, deterministic=det)
if "solver" in kw:
    opts["linear_solver"] = kw["solver"]
if "pcg_tol" in kw:
    opts["pcg_tol"] = float(kw["pcg_tol"])
if "schur_mode" in kw:
    opts["schur_mode"] = kw["schur_mode"]
with _capi.Handle(_capi.Arrays.from_dict(p), **opts) as h:
    ms_pass, ms_k = h.build_bench(1e-4, warmup=2, reps=int(kw.get("builds", 10)))
    print("%s det=%d: build pass %.3f ms, chunk kernel %.3f ms" % (name, det, ms_pass, ms_k), flush=True)
    n = int(kw.get("lm", 0))
    if n:
        import time
        h.run(1); h.synchronize()
        t0 = time.perf_counter(); s = h.run(n); h.synchronize(); dt = time.perf_counter() - t0
        print("LM: %d steps, %d PCG iterations, %.3f ms per step, cost %.6e -> %.6e" %
              (s["iterations"], s["pcg_iterations"], dt * 1e3 / max(s["iterations"] - 1, 1), s["initial_cost"], s["final_cost"]), flush=True)
