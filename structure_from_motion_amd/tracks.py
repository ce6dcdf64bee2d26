"""Opt-in track bookkeeping next to the bundle-adjustment path.

NOT reference behaviour -- nothing here runs unless the driver calls it.  The reference's driver records, for every
freshly triangulated point, only the NEW image's keypoint (``Reconstruction.py:244-248``: ``ref[match.trainIdx] =
p_idx``); the partner image's reference array ``Ref_Old`` is never written, and the 2D-3D matches that PnP registered
the image with are thrown away (``utils.Correspondence2D_3D``, ``utils.py:278-312``, returns coordinates, no indices).
So seed-pair points have two observations and every other point exactly one: two residuals for three
unknowns, and bundle adjustment can move such a point along its viewing ray for free.

These helpers record the observations the driver already has in its hands at those two places, so that tracks have
length >= 2 and the adjustment constrains the structure:

* :func:`record_partner_observations` -- at ``Reconstruction.py:244-248``: the partner image's keypoint of every
  triangulated match also points at the new 3D point.
* :func:`correspondence_point_indices` + :func:`record_pnp_observations` -- at ``Reconstruction.py:107-145``: the
  inlier 2D-3D matches of the PnP step become observations of the EXISTING points by the new image.

They only edit the ``ref`` lists inside ``image_data`` (element 3 of the tuples, ``-1`` = no point), which is exactly
what ``prepare_observations_from_image_data`` reads (``BundleAdjustment.py:257-270``); an entry that already references
a point is never overwritten, and nothing is recorded for indices out of range.  Post-BA culling by reprojection error
is :meth:`structure_from_motion_amd.session.BASession.cull_points`.
"""
import numpy as np


def _pairs(idx_a, idx_b):
    a = np.asarray(idx_a, dtype=np.int64).reshape(-1)
    b = np.asarray(idx_b, dtype=np.int64).reshape(-1)
    if len(a) != len(b):
        raise ValueError("index arrays differ in length")
    return a, b


def match_indices(matches):
    """(queryIdx, trainIdx) arrays of a list of ``cv2.DMatch``-like objects, or of an (n, 2) integer array."""
    if len(matches) and hasattr(matches[0], "queryIdx"):
        return (np.fromiter((m.queryIdx for m in matches), dtype=np.int64, count=len(matches)),
                np.fromiter((m.trainIdx for m in matches), dtype=np.int64, count=len(matches)))
    m = np.asarray(matches, dtype=np.int64).reshape(-1, 2)
    return m[:, 0].copy(), m[:, 1].copy()


def _record(ref, kp_idx, point_idx, n_points):
    n = 0
    for k, p in zip(kp_idx.tolist(), point_idx.tolist()):
        if 0 <= k < len(ref) and 0 <= p < n_points and ref[k] == -1:
            ref[k] = int(p)          # a Python int: the reference's validity rule is isinstance(idx, int) (:262)
            n += 1
    return n


def record_partner_observations(image_data, partner_id, partner_kp_idx, point_idx, n_points):
    """The partner image of a triangulation also observes the new points.

    ``partner_kp_idx[i]`` is the partner image's keypoint of the i-th triangulated match (``match.queryIdx`` of
    ``valid_matches`` at ``Reconstruction.py:244``), ``point_idx[i]`` the index the new point got in ``all_points_3D``
    (``p_idx`` at ``:245``).  Returns how many observations were recorded."""
    kp_idx, pts = _pairs(partner_kp_idx, point_idx)
    if partner_id not in image_data or len(image_data[partner_id]) < 6:
        return 0
    return _record(image_data[partner_id][3], kp_idx, pts, n_points)


def correspondence_point_indices(image_data):
    """Point index of every row of the descriptor / 3D-point database that ``utils.Correspondence2D_3D`` builds
    (``utils.py:285-301``: for each image in ``image_data`` order, its keypoints that reference a point, in keypoint
    order).  ``match.trainIdx`` of that function's matches indexes this array -- the index the reference drops."""
    out = []
    for _img_id, cam_data in image_data.items():
        ref = np.asarray(cam_data[3])
        valid = np.flatnonzero(ref != -1)
        if len(valid):
            out.append(ref[valid].astype(np.int64))
    return np.concatenate(out) if out else np.zeros(0, np.int64)


def record_pnp_observations(image_data, new_id, new_kp_idx, point_idx, n_points, inliers=None):
    """The new image observes the EXISTING points it was registered with.

    ``new_kp_idx[i]`` / ``point_idx[i]``: keypoint of the new image and point index of the i-th 2D-3D match
    (``match.queryIdx`` and ``correspondence_point_indices(image_data)[match.trainIdx]``); ``inliers``: optional indices
    or boolean mask of the matches PnP-RANSAC kept (``Reconstruction.py:107-130``).  Call it after the new image's tuple
    is in ``image_data`` (``:251``).  Returns how many observations were recorded."""
    kp_idx, pts = _pairs(new_kp_idx, point_idx)
    if inliers is not None:
        inl = np.asarray(inliers).reshape(-1)
        sel = np.flatnonzero(inl) if inl.dtype == bool else inl.astype(np.int64)
        kp_idx, pts = kp_idx[sel], pts[sel]
    if new_id not in image_data or len(image_data[new_id]) < 6:
        return 0
    return _record(image_data[new_id][3], kp_idx, pts, n_points)


def track_lengths(image_data, visited_ids, n_points):
    """Observations per point over the registered images (the validity rules of BundleAdjustment.py:260-264)."""
    cnt = np.zeros(n_points, dtype=np.int64)
    for img_id in visited_ids:
        if img_id not in image_data or len(image_data[img_id]) < 6:
            continue
        ref, kp = image_data[img_id][3], image_data[img_id][5]
        for k, p in enumerate(ref):
            if p != -1 and k < len(kp) and isinstance(p, int) and 0 <= p < n_points:
                cnt[p] += 1
    return cnt
