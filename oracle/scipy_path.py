"""TEST INFRASTRUCTURE ONLY -- CPU port of the reference's scipy configuration.

Restates, from scratch, how ``/root/reference/src/BundleAdjustment.py`` drives scipy so that
the same path can be (a) checked against the golden vectors captured from the reference itself
and (b) timed on the GPU box's host cores as ``cpu_baseline`` (kind "port") -- the reference's
Python cannot travel to the GPU box.

  * residual closure: one Python loop iteration per observation, one single-point projection per
    iteration, dict lookups for camera and point parameters (BundleAdjustment.py:102-152).
    ``cv2.projectPoints`` is replaced by ``oracle.cv2_standin.projectPoints`` (OpenCV is absent).
  * Jacobian sparsity: 2 rows x (6 camera + 3 point) columns per observation (:154-198)
  * solver call: least_squares(fun, x0, jac_sparsity=S, loss, f_scale, max_nfev,
    ftol=xtol=gtol=1e-8) with scipy defaults TRF / LSMR / 2-point (:210-220)

It is interpreter-bound and uses ONE core.
"""
import time

import numpy as np
from scipy.optimize import least_squares
from scipy.optimize._numdiff import approx_derivative, group_columns
from scipy.sparse import coo_matrix

from . import cv2_standin as cv2


def make_closure(cam_pose, cam_intr, cam_fixed, points, obs_cam, obs_pt, obs_xy):
    """-> (fun, x0, sparsity) in the reference's parameter layout (BundleAdjustment.py:46-65)."""
    n_c, n_p, n_o = len(cam_pose), len(points), len(obs_cam)
    fixed = np.asarray(cam_fixed).astype(bool)
    Ks = [np.array([[k[0], 0, k[2]], [0, k[1], k[3]], [0, 0, 1.0]]) for k in cam_intr]
    observations = [{"camera_id": int(c), "point_idx": int(p), "pixel": np.array(xy, dtype=np.float64)}
                    for c, p, xy in zip(obs_cam, obs_pt, obs_xy)]
    x0 = np.concatenate([cam_pose[~fixed].reshape(-1), np.asarray(points, dtype=np.float64).reshape(-1)])
    n_free = int((~fixed).sum())
    # the reference holds R per camera and re-derives the ANCHORED cameras' rvec from it on every call (:110-114)
    R_fixed = {c: cv2.Rodrigues(np.asarray(cam_pose[c, :3], dtype=np.float64).reshape(3, 1))[0] for c in range(n_c) if fixed[c]}

    def fun(params):
        idx = 0
        cam_params = {}
        for c in range(n_c):
            if fixed[c]:
                cam_params[c] = (cv2.Rodrigues(R_fixed[c])[0].flatten(), cam_pose[c, 3:])
            else:
                cam_params[c] = (params[idx:idx + 3], params[idx + 3:idx + 6])
                idx += 6
        point_coords = {}
        for i in range(n_p):
            point_coords[i] = params[idx:idx + 3]
            idx += 3
        out = []
        for ob in observations:
            rvec, tvec = cam_params[ob["camera_id"]]
            proj, _ = cv2.projectPoints(point_coords[ob["point_idx"]].reshape(1, 3), rvec, tvec, Ks[ob["camera_id"]], None)
            out.extend(proj[0, 0] - ob["pixel"])
        return np.array(out, dtype=np.float64)

    cam_off = np.full(n_c, -1, dtype=np.int64)
    cam_off[~fixed] = 6 * np.arange(n_free)
    rows, cols = [], []
    for k in range(n_o):
        co = cam_off[obs_cam[k]]
        cs = ([co + j for j in range(6)] if co >= 0 else []) + [6 * n_free + 3 * int(obs_pt[k]) + j for j in range(3)]
        for r in (2 * k, 2 * k + 1):
            rows.extend([r] * len(cs))
            cols.extend(cs)
    sparsity = coo_matrix((np.ones(len(rows), dtype=bool), (rows, cols)), shape=(2 * n_o, len(x0))).tocsr()
    return fun, x0, sparsity


def solve(prob, f_scale=1.0, max_nfev=300, loss="huber"):
    fun, x0, sparsity = make_closure(prob["cam_pose"], prob["cam_intr"], prob["cam_fixed"], prob["points"],
                                     prob["obs_cam"], prob["obs_pt"], prob["obs_xy"])
    return least_squares(fun, x0, jac_sparsity=sparsity, loss=loss, f_scale=f_scale, max_nfev=max_nfev,
                         ftol=1e-8, xtol=1e-8, gtol=1e-8)


def time_linearisation(prob, repeats=1):
    """Wall-clock of what the reference spends per Jacobian: f(x) plus one residual sweep per
    colour group (scipy _numdiff.py:636-699).  Returns dict(obs_per_s, seconds, sweeps, n_obs)."""
    fun, x0, sparsity = make_closure(prob["cam_pose"], prob["cam_intr"], prob["cam_fixed"], prob["points"],
                                     prob["obs_cam"], prob["obs_pt"], prob["obs_xy"])
    groups = group_columns(sparsity)
    n_groups = int(groups.max()) + 1
    t0 = time.perf_counter()
    for _ in range(repeats):
        f0 = fun(x0)
        approx_derivative(fun, x0, method="2-point", f0=f0, sparsity=(sparsity, groups))
    dt = (time.perf_counter() - t0) / repeats
    n_obs = len(prob["obs_cam"])
    return dict(obs_per_s=n_obs / dt, seconds=dt, sweeps=1 + n_groups, n_obs=n_obs)
