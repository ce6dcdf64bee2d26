"""Developer: tests/test_gpu_fuzz.py's differential test at larger, nastier sizes -- up to 300 cameras, tracks of up to 250
(wide chunks, cameras without a slot, many rounds of Schur groups, heavy points now and then) -- reduced system against the
oracle.  usage: python tools/fuzz_structure.py [n_cases] [first_seed]"""
import sys
import numpy as np
sys.path.insert(0, '/root/repo')
from oracle import ba_oracle as O
from structure_from_motion_amd import _capi, synthetic


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def problem(rng):
    nc = int(rng.integers(30, 301))
    n_pts = int(rng.integers(20, 400))
    base = synthetic.make_ring(nc, n_pts, seed=int(rng.integers(1 << 30)), mean_track=2.0, outlier_frac=0.0, radius=6.0)
    cam_pose, cam_intr, pts = base["cam_pose_gt"].copy(), base["cam_intr"].copy(), base["points_gt"].copy()
    oc, op = [], []
    style = int(rng.integers(0, 4))       # 0: short random tracks, 1: long random tracks, 2: windows, 3: mixture
    for q in range(n_pts):
        if rng.random() < 0.05:
            continue
        st = style if style < 3 else int(rng.integers(0, 3))
        if st == 0:
            cams = rng.choice(nc, size=int(rng.integers(1, 13)), replace=False)
        elif st == 1:
            hi = min(nc, 300 if rng.random() < 0.03 else 250)
            cams = rng.choice(nc, size=int(rng.integers(min(60, hi), hi + 1)), replace=False)
        else:
            L = int(rng.integers(2, min(nc, 70)))
            cams = (int(rng.integers(0, nc)) + np.arange(L)) % nc
        if rng.random() < 0.05:
            cams = np.concatenate([cams, cams[:1]])
        oc.extend(cams.tolist()); op.extend([q] * len(cams))
    oc, op = np.array(oc, np.int32), np.array(op, np.int32)
    perm = rng.permutation(len(oc))
    oc, op = oc[perm], op[perm]
    xy = synthetic.project_points(cam_pose, cam_intr, pts, oc, op) + rng.normal(0, 0.5, (len(oc), 2))
    bad = rng.random(len(oc)) < 0.03
    xy[bad] += rng.uniform(-40, 40, (int(bad.sum()), 2))
    fixed = np.zeros(nc, np.uint8)
    fixed[rng.choice(nc, size=int(rng.integers(1, max(2, nc // 5))), replace=False)] = 1
    cam_pose = cam_pose + rng.normal(0, 1e-3, cam_pose.shape) * (fixed[:, None] == 0)
    pts = pts + rng.normal(0, 1e-2, pts.shape)
    return dict(cam_pose=cam_pose, cam_intr=cam_intr, cam_fixed=fixed, points=pts, obs_cam=oc, obs_pt=op, obs_xy=xy), style


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
schur_mode = sys.argv[3] if len(sys.argv) > 3 else "auto"   # auto | chunks | rows
bad = 0
for seed in range(seed0, seed0 + n_cases):
    rng = np.random.default_rng(seed)
    p, style = problem(rng)
    loss = O.LOSSES[seed % len(O.LOSSES)]
    lam = float(rng.choice([1e-4, 1e-2, 1.0]))
    arr = _capi.Arrays.from_dict(p)
    a = _capi.analyze(arr)
    args = (p["cam_pose"], p["cam_intr"], p["points"], p["obs_cam"], p["obs_pt"])
    nb = O.normal_blocks(*args, p["obs_xy"], 1.0, loss, floor=1.0)
    S, gred, _, _, _ = O.reduced_system(nb, p["cam_fixed"], p["obs_cam"], p["obs_pt"], lam)
    with _capi.Handle(arr, loss=loss, f_scale=1.0, schur_mode=schur_mode) as h:
        h.build(lam)
        r = h.reduced_system()
        free = p["cam_fixed"] == 0
        e = dict(F=abs(h.cost() - nb["F"]) / max(nb["F"], 1e-300), V=rel(r["V"], nb["V"]), gp=rel(r["gp"], nb["gp"]),
                 U=rel(r["U"][free], nb["U"][free]), gc=rel(r["gc"][free], nb["gc"][free]), S=rel(r["S"], S), gred=rel(r["gred"], gred))
    ok = e["F"] < 1e-11 and e["V"] < 1e-11 and e["gp"] < 1e-10 and e["U"] < 1e-11 and e["gc"] < 1e-10 and e["S"] < 1e-7 and e["gred"] < 1e-7
    bad += not ok
    print("seed %3d style %d nc=%3d obs=%6d heavy=%d direct cam obs %6d direct pairs %7d max cams %3d  S %.1e gred %.1e U %.1e  %s" %
          (seed, style, len(p["cam_pose"]), len(p["obs_cam"]), a["n_heavy_points"], a["n_direct_cam_obs"], a["n_direct_pairs"], a["max_chunk_cams"],
           e["S"], e["gred"], e["U"], "ok" if ok else "MISMATCH " + str(e)), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
