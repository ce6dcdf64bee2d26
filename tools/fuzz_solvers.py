"""Random chains and rings (sizes, track lengths, gauge, camera numbering) through AUTO -- block cyclic reduction beyond
128 free cameras -- against the dense Cholesky factorisation forced on the same problem: same LM path, same minimum.
usage: python tools/fuzz_solvers.py [n_cases] [first_seed]"""
import sys
import numpy as np
sys.path.insert(0, '/root/repo')
from structure_from_motion_amd import _capi, synthetic


def renumber(p, rng):
    nc = len(p["cam_pose"]); new_of_old = rng.permutation(nc); old_of_new = np.argsort(new_of_old)
    q = dict(p)
    for k in ("cam_pose", "cam_intr", "cam_fixed"):
        q[k] = np.ascontiguousarray(p[k][old_of_new])
    q["obs_cam"] = new_of_old[p["obs_cam"]].astype(np.int32)
    return q


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for seed in range(seed0, seed0 + n_cases):
    rng = np.random.default_rng(seed)
    nc = int(rng.integers(135, int(sys.argv[3]) if len(sys.argv) > 3 else 700))
    mt = int(rng.integers(3, 21))
    ring = rng.random() < 0.35
    shuffle = rng.random() < 0.4
    n_fixed = int(rng.integers(1, 4))
    track = float(rng.uniform(2.5, min(mt, 8)))
    n_pts = int(nc * rng.uniform(15, 40))
    if ring:
        mt = min(mt, 10)   # (a shuffled ring folds: twice the bandwidth)
        p = synthetic.make_ring(nc, n_pts, seed=seed, mean_track=min(track, mt - 1), max_track=mt, n_fixed=n_fixed, outlier_frac=0.01,
                                radius=0.15 * nc, point_radius=(4.0, 20.0))
    else:
        p = synthetic.make_street(nc, n_pts, int(n_pts * track), seed=seed, outlier_frac=0.01, max_track=mt, n_fixed=n_fixed)
    if shuffle:
        p = renumber(p, rng)
    out = {}
    for solver in ("auto", "dense_cholesky"):
        arr = _capi.Arrays.from_dict(p)
        out[solver] = _capi.solve(arr, loss="huber", f_scale=1.0, max_iter=8, linear_solver=solver)
    a, d = out["auto"], out["dense_cholesky"]
    ok = (a["pcg_iterations"] == 0 and d["pcg_iterations"] == 0 and a["iterations"] == d["iterations"] and a["accepted"] == d["accepted"]
          and abs(a["final_cost"] - d["final_cost"]) <= 1e-8 * d["final_cost"])
    bad += not ok
    print("seed %3d: %s nc=%3d max_track=%2d fixed=%d %s  steps %d/%d acc %d/%d pcg %d/%d cost %.9e / %.9e  %s" %
          (seed, "ring  " if ring else "street", nc, mt, n_fixed, "shuffled" if shuffle else "natural ", a["iterations"], d["iterations"],
           a["accepted"], d["accepted"], a["pcg_iterations"], d["pcg_iterations"], a["final_cost"], d["final_cost"], "ok" if ok else "MISMATCH"),
          flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
