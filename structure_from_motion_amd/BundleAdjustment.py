"""Drop-in replacement for the reference's ``BundleAdjustment`` module, backed by the MI355X engine.

Same seven public names, call signatures, return containers and error behaviour as
``/root/reference/src/BundleAdjustment.py`` so that ``Reconstruction.py``'s
``from BundleAdjustment import *`` (``Reconstruction.py:9``) and its calls
(``inter_ba`` ``:253-255``, ``compute_ba_reprojection_errors`` ``:263``, ``final_ba``
``:302-304``) run unchanged.  What differs is behind the boundary: instead of
``scipy.optimize.least_squares`` with a finite-difference Jacobian over a Python residual loop
(``BundleAdjustment.py:102-152, 210-220``) the solve is a Levenberg-Marquardt / Schur-complement
engine in hand-written HIP (``csrc/``), reached through the C ABI of ``include/ba_hip.h``.

There is no CPU fallback: without ``libba_hip.so`` and a GPU the solve raises inside the
``try`` below and -- exactly like any other failure in the reference (``:238-240``) -- is
reported as ``[BA] Bundle adjustment error: ...`` with ``(cameras, points_3d, False)``.
"""
import itertools
import operator

import numpy as np

from . import _capi
from .geometry import matrices_to_rvecs, matrix_to_rvec, rvec_to_matrix

try:   # the container walks of the marshalling in C (csrc/ba_pyhelp.c); without it the NumPy / itertools paths below
    from . import _ba_pyhelp as _help
except ImportError:   # pragma: no cover  (built by csrc/build.sh next to libba_hip.so)
    _help = None

__all__ = [
    "global_bundle_adjustment", "prepare_observations_from_image_data", "prepare_cameras_from_image_data",
    "update_image_data_from_ba_results", "inter_ba", "final_ba", "compute_ba_reprojection_errors",
]

# Last engine summary (status, iterations, costs, timings) -- extra, not part of the reference API.
last_summary = None

# The process's one-time GPU costs (runtime, code objects, staging buffer: some 50 ms) are paid at import, where the
# reference pays for `import cv2` and scipy, not inside the driver's first inter_ba call.  BA_NO_WARM_UP=1 skips it; without
# the library or a GPU it does nothing (the calls themselves report that, as the reference reports any failure).
import os as _os
if not _os.environ.get("BA_NO_WARM_UP"):
    _capi.warm_up()


def _anchor_ids(cam_ids, visited_ids):
    """Gauge rule of BundleAdjustment.py:34-38: the first visited camera is always fixed, the
    second one too while fewer than 15 images are registered."""
    first = visited_ids[0] if len(visited_ids) > 0 else cam_ids[0]
    second = visited_ids[1] if len(visited_ids) > 1 else cam_ids[1]
    return first, second, len(visited_ids) < 15


def _reference_success(summary):
    """The reference's success rule (BundleAdjustment.py:223-224): scipy's ``result.success`` (status > 0) OR the
    quirk ``result.cost < result.fun.mean() * len(result.fun) * 0.95`` -- the cost against 0.95 x the SIGNED sum of
    the final residuals (``BaSummary.residual_sum``).  Returns (success, converged)."""
    converged = summary["status"] > 0
    return converged or summary["final_cost"] < summary["residual_sum"] * 0.95, converged


def _to_arrays(cameras, points_3d, observations, cam_ids, fixed_ids):
    """dict / list containers -> structure of arrays (the layout of ``BaProblem``)."""
    n_c = len(cam_ids)
    index_of = {cid: i for i, cid in enumerate(cam_ids)}
    cam_pose = np.empty((n_c, 6))
    cam_intr = np.empty((n_c, 4))
    cam_fixed = np.zeros(n_c, dtype=np.uint8)
    for i, cid in enumerate(cam_ids):
        cam = cameras[cid]
        cam_pose[i, :3] = matrix_to_rvec(cam["R"])
        cam_pose[i, 3:] = np.asarray(cam["t"], dtype=np.float64).reshape(3)
        K = np.asarray(cam["K"], dtype=np.float64)
        cam_intr[i] = (K[0, 0], K[1, 1], K[0, 2], K[1, 2])
        cam_fixed[i] = cid in fixed_ids
    # (bulk conversions instead of one Python statement per number; the values and their order are the same)
    n_p = len(points_3d)
    try:
        points = np.asarray(points_3d, dtype=np.float64)
        if points.ndim != 2 or points.shape[1] < 3:
            raise ValueError
        points = np.ascontiguousarray(points[:, :3])
    except (ValueError, TypeError):   # ragged rows: element by element
        points = np.array([[p[0], p[1], p[2]] for p in points_3d], dtype=np.float64).reshape(-1, 3)
    n_o = len(observations)
    obs_cam = np.fromiter((index_of[ob["camera_id"]] for ob in observations), dtype=np.int32, count=n_o)
    obs_pt = np.fromiter((ob["point_idx"] for ob in observations), dtype=np.int64, count=n_o)
    if n_o and (obs_pt.min() < -2**31 or obs_pt.max() >= 2**31):
        raise KeyError(int(obs_pt.max()))
    obs_pt = obs_pt.astype(np.int32)
    try:
        obs_xy = (np.concatenate([ob["pixel"] for ob in observations]).astype(np.float64, copy=False).reshape(n_o, 2)
                  if n_o else np.empty((0, 2)))
    except ValueError:   # a pixel that is not a pair of numbers in an array: element by element
        obs_xy = np.empty((n_o, 2))
        for k, ob in enumerate(observations):
            obs_xy[k] = ob["pixel"]
    if n_o and (obs_pt.min() < 0 or obs_pt.max() >= n_p):
        raise KeyError(int(obs_pt.max()))  # the reference's point_coords[point_idx] lookup fails the same way (:137)
    return _capi.Arrays(cam_pose, cam_intr, cam_fixed, points, obs_cam, obs_pt, obs_xy)


def global_bundle_adjustment(cameras, points_3d, observations, visited_ids,
                             robust_loss='huber', loss_f_scale=1.0,
                             max_nfev=100, verbose=0):
    """Global bundle adjustment with gauge anchoring over camera (rvec, t) and 3D points.

    Args / returns as the reference (BundleAdjustment.py:7-21):
        cameras      {img_id: {'R': 3x3, 't': 3x1, 'K': 3x3}}
        points_3d    [[x, y, z, r, g, b], ...]
        observations [{'camera_id', 'point_idx', 'pixel'}, ...]
        visited_ids  camera ids in registration order (decides the anchors)
    Returns (optimized_cameras, optimized_points, success); on failure the ORIGINAL
    ``cameras`` / ``points_3d`` objects are returned with ``False``.
    """
    global last_summary
    if len(cameras) < 3 or len(points_3d) < 10:
        print(f"[BA] Insufficient data: {len(cameras)} cameras, {len(points_3d)} points")
        return cameras, points_3d, False

    print(f"[BA] Starting BA with {len(cameras)} cameras, {len(points_3d)} points, {len(observations)} observations")

    cam_ids = list(cameras.keys())
    first_id, second_id, anchor_second = _anchor_ids(cam_ids, visited_ids)
    fixed_ids = {first_id, second_id} if anchor_second else {first_id}
    if verbose:
        print(f"[BA] Anchoring cameras: {[first_id, second_id] if anchor_second else [first_id]}")

    try:
        n_free = sum(1 for c in cam_ids if c not in fixed_ids)
        if 6 * n_free + 3 * len(points_3d) == 0:
            if verbose:
                print("[BA] All cameras anchored, no optimization needed")
            return cameras, points_3d, True

        arr = _to_arrays(cameras, points_3d, observations, cam_ids, fixed_ids)
        summary = _capi.solve(arr, loss=robust_loss, f_scale=float(loss_f_scale), max_iter=int(max_nfev),
                              ftol=1e-8, xtol=1e-8, gtol=1e-8, verbose=2 if verbose else 0)
        last_summary = summary

        success, converged = _reference_success(summary)
        if success:
            new_cameras = {}
            R_new = rvec_to_matrix(arr.cam_pose[:, :3])
            for i, cid in enumerate(cam_ids):
                if cid in fixed_ids:
                    new_cameras[cid] = cameras[cid].copy()
                else:
                    new_cameras[cid] = {"R": R_new[i].copy(), "t": arr.cam_pose[i, 3:].reshape(3, 1).copy(),
                                        "K": cameras[cid]["K"].copy()}
            new_points = [[arr.points[i, 0], arr.points[i, 1], arr.points[i, 2], p[3], p[4], p[5]]
                          for i, p in enumerate(points_3d)]
            if verbose:
                status = "converged" if converged else "improved significantly"
                print(f"[BA] Bundle adjustment {status}! Cost: {summary['final_cost']:.6f}")
            return new_cameras, new_points, True
        if verbose:
            print(f"[BA] Bundle adjustment failed: status {summary['status']} after {summary['iterations']} steps")
        return cameras, points_3d, False

    except Exception as e:  # the reference maps every failure to (inputs, False) (:238-240)
        print(f"[BA] Bundle adjustment error: {e}")
        return cameras, points_3d, False


def _valid_track_entries(cam_data, n_points):
    """(keypoint index, point index) pairs of one image that the reference would use
    (BundleAdjustment.py:260-264): reference set, keypoint exists, a Python int, in range."""
    _R, _t, _K, ref_array, _desc, kp = cam_data[:6]
    n_kp = len(kp)
    for kp_idx, point_idx in enumerate(ref_array):
        if point_idx != -1 and kp_idx < n_kp and isinstance(point_idx, int) and 0 <= point_idx < n_points:
            yield kp_idx, point_idx


def _track_arrays(cam_data, n_points):
    """(keypoint indices, point indices) of one image as arrays, under the reference's rules (:260-264)."""
    ref_array, kp = cam_data[3], cam_data[5]
    if _help is not None:
        raw = _help.track_entries(ref_array, len(kp), n_points)
        if raw is not None:
            a = np.frombuffer(raw, dtype=np.int64).reshape(-1, 2)
            return a[:, 0], a[:, 1]
    if isinstance(ref_array, list) and set(map(type, ref_array)) <= {int}:
        ref = np.asarray(ref_array, dtype=np.int64)
        ok = (ref != -1) & (ref >= 0) & (ref < n_points)
        ok[len(kp):] = False
        kidx = np.flatnonzero(ok)
        return kidx, ref[kidx]
    pairs = list(_valid_track_entries(cam_data, n_points))        # anything unusual: the generic rules
    if not pairs:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    a = np.array(pairs, dtype=np.int64)
    return a[:, 0], a[:, 1]


_PT = operator.attrgetter("pt")


def _keypoint_xy(kp, kidx):
    """(len(kidx), 2) float64 pixel coordinates ``kp[k].pt`` (:269).  One flat iterator into ``np.fromiter``: less than
    half the time of ``np.array`` over a list of tuples, and the keypoints are most of what a ``final_ba`` call reads."""
    if _help is not None:
        out = np.empty((len(kidx), 2), dtype=np.float64)
        if _help.keypoints_xy(kp, np.ascontiguousarray(kidx, dtype=np.int64), out):
            return out
    flat = itertools.chain.from_iterable(map(_PT, map(kp.__getitem__, kidx.tolist())))
    return np.fromiter(flat, dtype=np.float64, count=2 * len(kidx)).reshape(-1, 2)


def _rows_to_array(rows):
    """``np.asarray(rows, float64)`` for the driver's list of equally long point rows ([x, y, z, r, g, b] lists,
    ``Reconstruction.py:41-47``), through a flat iterator (about half the time); anything else goes to ``np.asarray``."""
    if _help is not None and isinstance(rows, list) and rows and isinstance(rows[0], (list, tuple)) and len(rows[0]) > 0:
        out = np.empty((len(rows), len(rows[0])), dtype=np.float64)
        if _help.rows_to_array(rows, out) is not None:
            return out
    if isinstance(rows, list) and rows and isinstance(rows[0], (list, tuple)):
        w = len(rows[0])
        if w and set(map(len, rows)) == {w}:
            try:
                return np.fromiter(itertools.chain.from_iterable(rows), dtype=np.float64, count=w * len(rows)).reshape(-1, w)
            except (TypeError, ValueError):
                pass
    return np.asarray(rows, dtype=np.float64)


def _image_data_to_arrays(image_data, all_points_3D, visited_ids, keypoints=None):
    """The driver's containers -> (Arrays without gauge, cam_ids, colours (Np, 3) or None), with NumPy instead of one
    Python object per observation.  Same camera order, observation order and validity rules as
    ``prepare_cameras_from_image_data`` + ``prepare_observations_from_image_data`` + ``_to_arrays`` (the tests compare
    them array by array).  ``keypoints(img_id, kp) -> (k, 2) float64`` lets a caller cache the pixel coordinates."""
    pts6 = _rows_to_array(all_points_3D)
    if pts6.ndim != 2 or pts6.shape[1] < 3:
        pts6 = np.array([[p[0], p[1], p[2]] for p in all_points_3D], dtype=np.float64).reshape(-1, 3)
    points = np.ascontiguousarray(pts6[:, :3])
    colours = pts6[:, 3:6] if pts6.shape[1] >= 6 else None
    n_points = len(points)
    usable = [i for i in visited_ids if i in image_data and len(image_data[i]) >= 6]
    cam_ids = list(dict.fromkeys(usable))
    index_of = {cid: i for i, cid in enumerate(cam_ids)}
    n_c = len(cam_ids)
    cam_pose, cam_intr = np.empty((n_c, 6)), np.empty((n_c, 4))
    Rs = np.empty((n_c, 3, 3))
    for i, cid in enumerate(cam_ids):
        R, t, K = image_data[cid][0], image_data[cid][1], image_data[cid][2]
        Rs[i] = np.asarray(R, dtype=np.float64).reshape(3, 3)
        cam_pose[i, 3:] = np.asarray(t, dtype=np.float64).reshape(3)
        K = np.asarray(K, dtype=np.float64)
        cam_intr[i] = (K[0, 0], K[1, 1], K[0, 2], K[1, 2])
    cam_pose[:, :3] = matrices_to_rvecs(Rs)       # (one batched SVD; matrix_to_rvec per camera otherwise)
    oc, op, oxy = [], [], []
    for img_id in usable:                         # per occurrence, like the reference's loop (:253)
        cam_data = image_data[img_id]
        kidx, pidx = _track_arrays(cam_data, n_points)
        if len(kidx):
            kp = cam_data[5]
            xy = keypoints(img_id, kp) if keypoints is not None else None
            oc.append(np.full(len(kidx), index_of[img_id], dtype=np.int32))
            op.append(pidx.astype(np.int32))
            oxy.append(xy[kidx] if xy is not None else _keypoint_xy(kp, kidx))
    obs_cam = np.concatenate(oc) if oc else np.zeros(0, np.int32)
    obs_pt = np.concatenate(op) if op else np.zeros(0, np.int32)
    obs_xy = np.concatenate(oxy) if oxy else np.zeros((0, 2))
    arr = _capi.Arrays(cam_pose, cam_intr, np.zeros(n_c, np.uint8), points, obs_cam, obs_pt, obs_xy)
    return arr, cam_ids, colours


def _ba_from_image_data(image_data, all_points_3D, visited_ids, loss_f_scale, max_nfev, verbose, min_obs=None,
                        keypoints=None, run=None):
    """What ``prepare_cameras`` + ``prepare_observations`` + ``global_bundle_adjustment`` + ``update_image_data`` do for
    ``inter_ba`` / ``final_ba`` (BundleAdjustment.py:316-350, 361-385), on arrays: same guards, prints, gauge, success rule
    and in-place update of ``image_data``; no list of observation dicts in between (82 k dicts and arrays per call at the
    Temple size).  ``min_obs``: inter_ba's extra guard (:325).  ``run(arr, **opts) -> summary`` replaces the one-shot
    ``_capi.solve`` (BASession keeps its handle there).  Returns (points, success, cam_ids, fixed_ids, attempted) -- attempted = got past inter_ba's own guard."""
    global last_summary
    arr, cam_ids, colours = _image_data_to_arrays(image_data, all_points_3D, visited_ids, keypoints)
    n_c, n_p, n_o = len(cam_ids), len(arr.points), len(arr.obs_cam)
    if min_obs is not None and (n_c < 3 or n_p < 10 or n_o < min_obs):
        if verbose:
            print("[BA] Insufficient data for BA")
        return all_points_3D, False, cam_ids, set(), False
    if n_c < 3 or n_p < 10:
        print(f"[BA] Insufficient data: {n_c} cameras, {n_p} points")
        return all_points_3D, False, cam_ids, set(), True
    print(f"[BA] Starting BA with {n_c} cameras, {n_p} points, {n_o} observations")
    first_id, second_id, anchor_second = _anchor_ids(cam_ids, visited_ids)
    fixed_ids = {first_id, second_id} if anchor_second else {first_id}
    if verbose:
        print(f"[BA] Anchoring cameras: {[first_id, second_id] if anchor_second else [first_id]}")
    for i, cid in enumerate(cam_ids):
        arr.cam_fixed[i] = cid in fixed_ids
    try:
        if n_o and (arr.obs_pt.min() < 0 or arr.obs_pt.max() >= n_p):
            raise KeyError(int(arr.obs_pt.max()))
        opts = dict(loss="huber", f_scale=float(loss_f_scale), max_iter=int(max_nfev), ftol=1e-8, xtol=1e-8, gtol=1e-8,
                    verbose=2 if verbose else 0)
        summary = run(arr, **opts) if run is not None else _capi.solve(arr, **opts)
        last_summary = summary
        success, converged = _reference_success(summary)
        if not success:
            if verbose:
                print(f"[BA] Bundle adjustment failed: status {summary['status']} after {summary['iterations']} steps")
            return all_points_3D, False, cam_ids, fixed_ids, True
        if verbose:
            print(f"[BA] Bundle adjustment {'converged' if converged else 'improved significantly'}! Cost: {summary['final_cost']:.6f}")
        R_new = rvec_to_matrix(arr.cam_pose[:, :3])
        for i, cid in enumerate(cam_ids):         # in-place update of the tuples (:298-306); anchors keep their pose
            entry = list(image_data[cid])
            if cid in fixed_ids:
                entry[0] = np.asarray(entry[0]).astype(np.float64)
                entry[1] = np.asarray(entry[1]).astype(np.float64).reshape(3, 1)
            else:
                entry[0], entry[1] = R_new[i].copy(), arr.cam_pose[i, 3:].reshape(3, 1).copy()
            image_data[cid] = tuple(entry)
        new_points = None
        if _help is not None:
            # (x, y, z from the solve, the colour entries are the caller's own objects -- as the reference's `+ colours`, :96-97)
            new_points = _help.rows_with_xyz(all_points_3D, np.ascontiguousarray(arr.points))
        if new_points is not None:
            pass
        elif colours is not None:
            new_points = np.concatenate([arr.points, colours], axis=1).tolist()
        else:
            new_points = [[x, y, z] + list(p[3:6]) for (x, y, z), p in zip(arr.points.tolist(), all_points_3D)]
        return new_points, True, cam_ids, fixed_ids, True
    except Exception as e:                        # as the reference: any failure -> (inputs, False) (:238-240)
        print(f"[BA] Bundle adjustment error: {e}")
        return all_points_3D, False, cam_ids, fixed_ids, True


def prepare_observations_from_image_data(image_data, visited_ids, all_points_3D):
    """image_data tuples -> observation dicts, camera-major in ``visited_ids`` order, keypoint
    order inside a camera (BundleAdjustment.py:243-272).  This defines the residual order."""
    observations = []
    n_points = len(all_points_3D)
    for img_id in visited_ids:
        if img_id not in image_data:
            continue
        cam_data = image_data[img_id]
        if len(cam_data) < 6:
            continue
        kp = cam_data[5]
        entries = list(_valid_track_entries(cam_data, n_points))
        if not entries:
            continue
        # (one float64 array per image, its rows handed out as the observations' pixels: same values, dtype and shape as
        # the reference's np.array(kp.pt, dtype=np.float64) per observation, a tenth of the time)
        xy = np.array([kp[k].pt for k, _ in entries], dtype=np.float64).reshape(-1, 2)
        observations.extend({"camera_id": img_id, "point_idx": point_idx, "pixel": xy[i]}
                            for i, (_k, point_idx) in enumerate(entries))
    return observations


def prepare_cameras_from_image_data(image_data, visited_ids):
    """image_data tuples -> cameras dict in ``visited_ids`` order, float64, t as (3,1)
    (BundleAdjustment.py:275-295)."""
    cameras = {}
    for img_id in visited_ids:
        if img_id not in image_data:
            continue
        cam_data = image_data[img_id]
        if len(cam_data) < 6:
            continue
        R, t, K = cam_data[0], cam_data[1], cam_data[2]
        cameras[img_id] = {"R": R.astype(np.float64), "t": t.astype(np.float64).reshape(3, 1), "K": K.astype(np.float64)}
    return cameras


def update_image_data_from_ba_results(image_data, optimized_cameras):
    """Write optimised R, t back into the image_data tuples, in place (BundleAdjustment.py:298-306)."""
    for img_id, cam in optimized_cameras.items():
        if img_id in image_data:
            entry = list(image_data[img_id])
            entry[0] = cam["R"].astype(np.float64)
            entry[1] = cam["t"].astype(np.float64)
            image_data[img_id] = tuple(entry)
    return image_data


def inter_ba(image_data, all_points_3D, visited_ids,
             ba_interval=10, verbose=True):
    """Intermediate BA every ``ba_interval`` registered images: huber, f_scale 1.5, max_nfev 150
    (BundleAdjustment.py:309-350).  Returns (points, image_data, ran_ok)."""
    if len(visited_ids) % ba_interval != 0 or len(visited_ids) < 3:
        return all_points_3D, image_data, False

    if verbose:
        print(f"\n{'=' * 50}")
        print(f"RUNNING BUNDLE ADJUSTMENT - {len(visited_ids)} images processed")
        print(f"{'=' * 50}")

    # (the reference goes through prepare_cameras / prepare_observations / global_bundle_adjustment /
    # update_image_data_from_ba_results here; _ba_from_image_data does the same on arrays)
    optimized_points, success, cam_ids, _, attempted = _ba_from_image_data(image_data, all_points_3D, visited_ids, 1.5, 150,
                                                                            verbose, min_obs=20)
    if success:
        if verbose:
            print(f"[BA] Successfully optimized {len(cam_ids)} cameras and {len(optimized_points)} points")
        return optimized_points, image_data, True
    if verbose and attempted:
        print("[BA] Bundle adjustment failed")
    return all_points_3D, image_data, False


def final_ba(image_data, all_points_3D, visited_ids, verbose=True):
    """Final BA over everything: huber, f_scale 1.0, max_nfev 300 (BundleAdjustment.py:353-385).
    Returns (points, image_data, ok)."""
    if verbose:
        print(f"\n{'=' * 60}")
        print("RUNNING FINAL BUNDLE ADJUSTMENT")
        print(f"{'=' * 60}")

    optimized_points, success = _ba_from_image_data(image_data, all_points_3D, visited_ids, 1.0, 300, verbose)[:2]
    if success:
        if verbose:
            print("[BA] Final BA completed successfully!")
        return optimized_points, image_data, True
    if verbose:
        print("[BA] Final BA failed")
    return all_points_3D, image_data, False


def compute_ba_reprojection_errors(image_data, all_points_3D, visited_ids):
    """Mean L2 reprojection error per registered camera, ``{img_id: error}``
    (BundleAdjustment.py:388-423 with utils.ReprojectionError, utils.py:271-276: ``K @ (R @ X + t)`` with the full K and
    with R exactly as stored in ``image_data`` -- no conversion to a rotation vector).
    One HIP launch for all cameras instead of one NumPy evaluation per camera."""
    ids, rts, Ks = [], [], []
    obs_cam, obs_pt, obs_xy = [], [], []
    n_points = len(all_points_3D)
    for img_id in visited_ids:
        if img_id not in image_data:
            continue
        cam_data = image_data[img_id]
        if len(cam_data) < 6:
            continue
        kp = cam_data[5]
        entries = list(_valid_track_entries(cam_data, n_points))
        if not entries:
            continue
        ci = len(ids)
        ids.append(img_id)
        rts.append(np.concatenate([np.asarray(cam_data[0], dtype=np.float64).reshape(9),
                                   np.asarray(cam_data[1], dtype=np.float64).reshape(3)]))
        Ks.append(np.asarray(cam_data[2], dtype=np.float64).reshape(9))
        kidx = np.fromiter((e[0] for e in entries), dtype=np.int64, count=len(entries))
        obs_cam.append(np.full(len(entries), ci, dtype=np.int32))
        obs_pt.append(np.fromiter((e[1] for e in entries), dtype=np.int32, count=len(entries)))
        obs_xy.append(np.array([kp[k].pt for k in kidx], dtype=np.float64).reshape(-1, 2))
    if not ids:
        return {}
    points = np.array([[p[0], p[1], p[2]] for p in all_points_3D], dtype=np.float64).reshape(-1, 3)
    n_c = len(ids)
    arr = _capi.Arrays(np.zeros((n_c, 6)), np.zeros((n_c, 4)), np.zeros(n_c, np.uint8), points,
                       np.concatenate(obs_cam), np.concatenate(obs_pt), np.concatenate(obs_xy))
    err = _capi.eval_reprojection_errors_rt(arr, np.array(rts), np.array(Ks))
    return {img_id: float(err[i]) for i, img_id in enumerate(ids)}
