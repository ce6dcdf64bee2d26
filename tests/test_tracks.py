"""track bookkeeping next to the path -- recording the partner-image and PnP observations the
reference's driver drops (Reconstruction.py:244-251, utils.py:278-312)."""
import copy

import numpy as np
import pytest

from structure_from_motion_amd import BundleAdjustment as BA
from structure_from_motion_amd import synthetic, tracks


class _M:
    def __init__(self, q, t):
        self.queryIdx, self.trainIdx = q, t


def _driver_like(n_cams=9, n_points=500, seed=2):
    """A reconstruction as the reference's driver leaves it (H4): every point was triangulated between the LAST camera
    of its track (the 'new' image, which gets the reference) and the one before it (the 'partner', whose keypoint exists
    but references nothing); earlier cameras of the track hold keypoints too (what PnP would have matched).
    Returns image_data, points, ids, and per point the (camera, keypoint) list of its whole true track."""
    p = synthetic.make_ring(n_cams, n_points, seed=seed, mean_track=3.2, max_track=5, n_fixed=2, noise=0.3, sig_p=0.0, sig_t=0.0, sig_r=0.0)
    ids, image_data, pts = synthetic.to_image_data(p)
    rng = np.random.default_rng(seed)
    truth = [[] for _ in range(n_points)]
    for c in ids:
        for k, q in enumerate(image_data[c][3]):
            truth[q].append((c, k))
    for q, tr in enumerate(truth):
        tr.sort()
        for c, k in tr[:-1]:
            image_data[c][3][k] = -1          # only the newest image keeps the reference
    # the seed pair's points keep both (Reconstruction.py:41-47)
    for q, tr in enumerate(truth):
        if len(tr) >= 2 and tr[0][0] == ids[0] and tr[1][0] == ids[1] and len(tr) == 2:
            for c, k in tr:
                image_data[c][3][k] = q
    # depth errors: what a two-view triangulation with a short baseline leaves behind
    gt = np.array([x[:3] for x in pts])
    for q in range(n_points):
        d = rng.normal(0, 0.08)
        c = truth[q][-1][0]
        R, t = image_data[c][0], image_data[c][1].reshape(3)
        centre = -R.T @ t
        ray = gt[q] - centre
        pts[q][:3] = list(gt[q] + d * ray)
    return image_data, pts, ids, truth, gt


def test_bookkeeping_rules():
    image_data, pts, ids, truth, _ = _driver_like()
    n = len(pts)
    before = tracks.track_lengths(image_data, ids, n)
    assert before.max() <= 2 and (before == 1).sum() > 0.5 * n                      # the H4 shape
    img = copy.deepcopy(image_data)
    # partner observations through DMatch-like objects and through plain arrays
    q = next(i for i, tr in enumerate(truth) if len(tr) >= 3 and before[i] == 1)
    (c_new, k_new), (c_old, k_old) = truth[q][-1], truth[q][-2]
    qi, ti = tracks.match_indices([_M(k_old, k_new)])
    assert tracks.record_partner_observations(img, c_old, qi, [q], n) == 1
    assert img[c_old][3][k_old] == q and isinstance(img[c_old][3][k_old], int)
    assert tracks.record_partner_observations(img, c_old, qi, [q + 1], n) == 0        # never overwrites
    assert tracks.record_partner_observations(img, c_old, [10 ** 6], [q], n) == 0     # keypoint out of range
    assert tracks.record_partner_observations(img, c_old, [k_old], [n], n) == 0       # point out of range
    assert tracks.record_partner_observations(img, 12345, [0], [0], n) == 0           # unknown image
    assert tracks.match_indices(np.array([[3, 4], [5, 6]]))[1].tolist() == [4, 6]
    # the index table of utils.Correspondence2D_3D's database: image order, keypoint order, referenced keypoints only
    tab = tracks.correspondence_point_indices(image_data)
    want = [r for c in image_data for r in image_data[c][3] if r != -1]
    assert tab.tolist() == want and len(tab) == int(before.sum())
    # PnP observations with an inlier mask / inlier indices
    c0, k0 = truth[q][0]
    row = int(np.flatnonzero(tab == q)[0])
    assert tracks.record_pnp_observations(img, c0, [k0, k0], [tab[row], tab[row]], n, inliers=np.array([False, True])) == 1
    assert img[c0][3][k0] == q
    assert tracks.record_pnp_observations(img, c0, [k0], [q], n, inliers=[0]) == 0    # already recorded
    after = tracks.track_lengths(img, ids, n)
    assert after[q] == before[q] + 2
    # what was recorded is exactly what prepare_observations now emits in addition
    o0 = BA.prepare_observations_from_image_data(image_data, ids, pts)
    o1 = BA.prepare_observations_from_image_data(img, ids, pts)
    assert len(o1) == len(o0) + 2


@pytest.mark.gpu
def test_recorded_tracks_constrain_the_structure(gpu):
    """With the reference's bookkeeping (tracks of 1 and 2) BA cannot correct a point's depth; with the partner and PnP
    observations recorded every point has its whole track and the same BA call brings the points back."""
    image_data, pts, ids, truth, gt = _driver_like()
    n = len(pts)
    rec = copy.deepcopy(image_data)
    n_rec = 0
    for q, tr in enumerate(truth):
        if len(tr) < 2:
            continue
        (c_old, k_old) = tr[-2]
        n_rec += tracks.record_partner_observations(rec, c_old, [k_old], [q], n)
        for c, k in tr[:-2]:
            n_rec += tracks.record_pnp_observations(rec, c, [k], [q], n)
    assert n_rec > 0 and tracks.track_lengths(rec, ids, n).min() >= 2
    err0 = np.linalg.norm(np.array([x[:3] for x in pts]) - gt, axis=1)
    out = {}
    for name, d in (("reference", copy.deepcopy(image_data)), ("recorded", rec)):
        p2, d2, ok = BA.final_ba(d, pts, ids, verbose=False)
        assert ok is True
        out[name] = np.linalg.norm(np.array([x[:3] for x in p2]) - gt, axis=1)
    multi = np.array([len(tr) >= 2 for tr in truth])
    assert np.median(out["recorded"][multi]) < 0.2 * np.median(err0[multi])       # depth recovered
    assert np.median(out["reference"][multi]) > 0.5 * np.median(err0[multi])      # single observations cannot
