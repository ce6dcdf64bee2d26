"""Build pass and LM step on a problem in its natural camera numbering and with the cameras numbered at random."""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from structure_from_motion_amd import _capi, synthetic

def renumber(p, seed):
    rng = np.random.default_rng(seed); nc = len(p["cam_pose"]); new_of_old = rng.permutation(nc); old_of_new = np.argsort(new_of_old)
    q = dict(p)
    for k in ("cam_pose", "cam_intr", "cam_fixed"): q[k] = np.ascontiguousarray(p[k][old_of_new])
    q["obs_cam"] = new_of_old[p["obs_cam"]].astype(np.int32)
    return q

name = sys.argv[1] if len(sys.argv) > 1 else "ladybug1723"
p = synthetic.make_config(name)
for tag, q in (("natural", p), ("shuffled", renumber(p, 3))):
    t0 = time.perf_counter()
    with _capi.Handle(_capi.Arrays.from_dict(q), loss="huber", f_scale=1.0, max_iter=300) as h:
        t_create = (time.perf_counter() - t0) * 1e3
        h.build_bench(1e-4, warmup=0, reps=3)
        ms_pass, ms_k = h.build_bench(1e-4, warmup=0, reps=10)
        t0 = time.perf_counter(); s = h.run(10); h.synchronize(); dt = time.perf_counter() - t0
    print("%-9s create %.1f ms, build pass %.3f ms (k_build %.3f), %d LM steps %.1f it/s, pcg %d, cost %.6e" %
          (tag, t_create, ms_pass, ms_k, s["iterations"], s["iterations"] / dt, s["pcg_iterations"], s["final_cost"]))
