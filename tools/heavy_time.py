"""Developer: what landmarks seen by (nearly) every camera cost a build pass: a street of n cameras plus k such points."""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from structure_from_motion_amd import _capi, synthetic
nc, k = int(sys.argv[1]) if len(sys.argv) > 1 else 1778, int(sys.argv[2]) if len(sys.argv) > 2 else 5
opts = {a.split("=")[0]: a.split("=")[1] for a in sys.argv[3:] if "=" in a}
p = synthetic.make_street(nc, 200000, 1000000, seed=3)
rng = np.random.default_rng(1)
npts = len(p["points"])
lm = np.stack([0.5 * nc + rng.uniform(-5, 5, k), rng.uniform(-3, 3, k), rng.uniform(8, 12, k) * nc], axis=1)   # far ahead of the street's middle
q = dict(p)
q["points"] = np.vstack([p["points"], lm])
oc = np.tile(np.arange(nc, dtype=np.int32), k); op = np.repeat(np.arange(npts, npts + k, dtype=np.int32), nc)
xy = synthetic.project_points(p["cam_pose"], p["cam_intr"], q["points"], oc, op) + rng.normal(0, 0.5, (len(oc), 2))
q["obs_cam"] = np.concatenate([p["obs_cam"], oc]); q["obs_pt"] = np.concatenate([p["obs_pt"], op]); q["obs_xy"] = np.vstack([p["obs_xy"], xy])
for name, prob in (("street", p), ("street + %d landmarks" % k, q)):
    arr = _capi.Arrays.from_dict(prob)
    a = _capi.analyze(arr)
    with _capi.Handle(arr, loss="huber", f_scale=1.0, **opts) as h:
        h.build_bench(1e-4, warmup=0, reps=2)
        ms, msk = h.build_bench(1e-4, warmup=0, reps=5)
    print("%-24s %d obs, heavy points %d: build pass %.3f ms (chunk kernel %.3f)" % (name, len(prob["obs_cam"]), a["n_heavy_points"], ms, msk), flush=True)
