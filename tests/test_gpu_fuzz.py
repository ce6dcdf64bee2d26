"""Randomised differential test: the HIP engine against the NumPy oracle on small problems with irregular structure
(random visibility, duplicate observations, random gauge, unobserved points, more cameras per chunk than LDS tiles,
every scipy loss), in both modes of the build pass (chunk kernel, rows mode).  Tolerances as in test_gpu_parity.py."""
import numpy as np
import pytest

from oracle import ba_oracle as O
from structure_from_motion_amd import _capi, synthetic

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def random_problem(rng):
    nc = int(rng.integers(2, 41))
    n_pts = int(rng.integers(1, 121))
    base = synthetic.make_ring(max(nc, 3), n_pts, seed=int(rng.integers(1 << 30)), mean_track=2.0, outlier_frac=0.0)
    cam_pose, cam_intr = base["cam_pose_gt"][:nc].copy(), base["cam_intr"][:nc].copy()
    pts = base["points_gt"].copy()
    oc, op = [], []
    wide = rng.random() < 0.3                                  # some problems: points seen by (nearly) every camera
    for q in range(n_pts):
        if rng.random() < 0.1:
            continue                                           # unobserved point
        k = int(rng.integers(1, (nc if wide else min(nc, 12)) + 1))
        cams = rng.choice(nc, size=k, replace=False)
        if rng.random() < 0.1:
            cams = np.concatenate([cams, cams[:1]])            # one camera sees the point twice
        oc.extend(cams.tolist()); op.extend([q] * len(cams))
    oc, op = np.array(oc, np.int32), np.array(op, np.int32)
    perm = rng.permutation(len(oc))
    oc, op = oc[perm], op[perm]
    xy = synthetic.project_points(cam_pose, cam_intr, pts, oc, op) + rng.normal(0, 0.5, (len(oc), 2))
    bad = rng.random(len(oc)) < 0.05
    xy[bad] += rng.uniform(-40, 40, (int(bad.sum()), 2))
    mode = rng.integers(0, 4)
    fixed = np.zeros(nc, np.uint8)
    if mode == 0:
        fixed[:] = 1                                           # points only
    elif mode == 1:
        fixed[rng.choice(nc, size=int(rng.integers(1, nc + 1)), replace=False)] = 1
    elif mode == 2:
        fixed[0] = 1
    cam_pose = cam_pose + rng.normal(0, 1e-3, cam_pose.shape) * (fixed[:, None] == 0)
    pts = pts + rng.normal(0, 1e-2, pts.shape)
    return dict(cam_pose=cam_pose, cam_intr=cam_intr, cam_fixed=fixed, points=pts, obs_cam=oc, obs_pt=op, obs_xy=xy)


@pytest.mark.parametrize("schur_mode", ["chunks", "rows"])
@pytest.mark.parametrize("seed", range(24))
def test_random_structure_against_oracle(gpu, seed, schur_mode):
    rng = np.random.default_rng(1000 + seed)
    p = random_problem(rng)
    loss = O.LOSSES[seed % len(O.LOSSES)]
    fs = float(rng.choice([0.7, 1.0, 1.5]))
    lam = float(rng.choice([1e-4, 1e-2, 1.0]))
    arr = _capi.Arrays.from_dict(p)
    args = (p["cam_pose"], p["cam_intr"], p["points"], p["obs_cam"], p["obs_pt"])
    f = O.residuals(*args, p["obs_xy"])
    np.testing.assert_allclose(_capi.eval_residuals(arr), f, rtol=1e-9, atol=1e-9)
    if len(f) == 0:
        return
    nb = O.normal_blocks(*args, p["obs_xy"], fs, loss, floor=1.0)
    S, gred, _, _, _ = O.reduced_system(nb, p["cam_fixed"], p["obs_cam"], p["obs_pt"], lam)
    with _capi.Handle(arr, loss=loss, f_scale=fs, schur_mode=schur_mode) as h:
        h.build(lam)
        r = h.reduced_system()
        free = p["cam_fixed"] == 0
        assert abs(h.cost() - nb["F"]) <= 1e-11 * max(nb["F"], 1e-300)
        assert rel(r["V"], nb["V"]) < 1e-11 and rel(r["gp"], nb["gp"]) < 1e-10
        if free.any():
            seen = np.zeros(len(free), bool); seen[p["obs_cam"]] = True
            assert rel(r["U"][free], nb["U"][free]) < 1e-11 and rel(r["gc"][free], nb["gc"][free]) < 1e-10
            # (1e-7: the oracle inverts V + lambda D, conditioned like 1 / lambda for rank-deficient V)
            assert rel(r["S"], S) < 1e-7 and rel(r["gred"], gred) < 1e-7


@pytest.mark.parametrize("seed", range(12))
def test_random_structure_solve_invariants(gpu, seed):
    """Properties that hold whatever the structure: the reported cost is the oracle's objective at the returned state,
    it never exceeds the initial cost, anchored cameras and unobserved points come back bit-identical."""
    rng = np.random.default_rng(5000 + seed)
    p = random_problem(rng)
    loss = O.LOSSES[seed % len(O.LOSSES)]
    solver = ["auto", "pcg", "dense_cholesky"][seed % 3]
    arr = _capi.Arrays.from_dict(p)
    s = _capi.solve(arr, loss=loss, f_scale=1.0, max_iter=40, linear_solver=solver, schur_mode=["auto", "rows"][(seed // 3) % 2])
    assert s["status"] >= 0
    assert s["final_cost"] <= s["initial_cost"] * (1 + 1e-12)
    f = O.residuals(arr.cam_pose, p["cam_intr"], arr.points, p["obs_cam"], p["obs_pt"], p["obs_xy"])
    c = O.cost(f, 1.0, loss) if len(f) else 0.0
    assert abs(c - s["final_cost"]) <= 1e-10 * max(c, 1e-300)
    fixed = p["cam_fixed"] == 1
    np.testing.assert_array_equal(arr.cam_pose[fixed], p["cam_pose"][fixed])
    unobs = np.setdiff1d(np.arange(len(p["points"])), p["obs_pt"])
    np.testing.assert_array_equal(arr.points[unobs], p["points"][unobs])
    assert np.all(np.isfinite(arr.cam_pose)) and np.all(np.isfinite(arr.points))
