"""CPU: the C-ABI library loads and exports what include/ba_hip.h declares; host-side structure
analysis; the marshalling layer against the reference's own marshalling (golden)."""
import ctypes
import os
import re

import numpy as np
import pytest
from conftest import ROOT, load_golden, problem_of

from structure_from_motion_amd import BundleAdjustment as BA
from structure_from_motion_amd import _capi, geometry, synthetic


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ba_hip.h")).read()
    declared = set(re.findall(r"\b(ba_[a-z_]+)\s*\(", hdr))
    assert declared == set(_capi.EXPORTED)
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _capi.load().ba_version() >= 100


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_capi.BaProblem) == 4 + 4 + 8 + 7 * 8 + 8
    assert ctypes.sizeof(_capi.BaOptions) == 8 + 6 * 8 + 6 * 4 + 8
    assert ctypes.sizeof(_capi.BaSummary) == 16 + 7 * 8 + 16


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = synthetic.make_street(6, 40, 160, seed=1)
    arr = _capi.Arrays.from_dict(p)
    with pytest.raises(_capi.BaError):
        _capi.eval_residuals(arr)
    cams = {i: {"R": np.eye(3), "t": np.zeros((3, 1)), "K": np.eye(3)} for i in range(4)}
    pts = [[0.0, 0.0, 5.0, 0, 0, 0]] * 12
    obs = [{"camera_id": 0, "point_idx": 0, "pixel": np.zeros(2)}]
    out = BA.global_bundle_adjustment(cams, pts, obs, [0, 1, 2, 3])
    assert out[2] is False and out[0] is cams and out[1] is pts


@pytest.mark.parametrize("kind", ["street", "ring", "sfm", "long"])
def test_structure_analysis(kind):
    if kind == "street":
        p = synthetic.make_street(40, 3000, 13000, seed=3)
    elif kind == "ring":
        p = synthetic.make_ring(47, 2000, seed=4, mean_track=6.0)
    elif kind == "sfm":
        p = synthetic.make_ring(11, 1500, seed=5, visibility="sfm", n_fixed=2)
    else:
        p = synthetic.make_ring(60, 300, seed=6, mean_track=20.0)
    a = _capi.analyze(_capi.Arrays.from_dict(p))
    n_obs = len(p["obs_cam"])
    assert a["n_free_cams"] == int((p["cam_fixed"] == 0).sum())
    assert a["n_observed_points"] == len(np.unique(p["obs_pt"]))
    assert a["n_chunks"] >= int(np.ceil(n_obs / 256))
    assert a["max_track"] == np.bincount(p["obs_pt"]).max()
    # pair entries = sum over points of L_free (L_free + 1) / 2
    free = p["cam_fixed"][p["obs_cam"]] == 0
    lf = np.bincount(p["obs_pt"][free], minlength=len(p["points"]))
    assert a["n_pair_entries"] == int((lf * (lf + 1) // 2).sum())
    # upper pattern = distinct co-visible free camera pairs + every diagonal
    fidx = np.cumsum(p["cam_fixed"] == 0) - 1
    pairs = set((i, i) for i in range(a["n_free_cams"]))
    order = np.argsort(p["obs_pt"], kind="stable")
    oc, op = p["obs_cam"][order], p["obs_pt"][order]
    st = np.searchsorted(op, np.arange(len(p["points"]) + 1))
    for q in range(len(p["points"])):
        cs = sorted(fidx[c] for c in oc[st[q]:st[q + 1]] if p["cam_fixed"][c] == 0)
        for i in range(len(cs)):
            for j in range(i, len(cs)):
                pairs.add((cs[i], cs[j]))
    assert a["nnz_blocks_upper"] == len(pairs)
    assert a["nnz_blocks_full"] == 2 * len(pairs) - a["n_free_cams"]


def test_analyze_rejects_bad_indices_and_routes_long_tracks():
    p = synthetic.make_street(6, 40, 160, seed=1)
    bad = dict(p); bad["obs_pt"] = p["obs_pt"].copy(); bad["obs_pt"][3] = 40
    with pytest.raises(_capi.BaError):
        _capi.analyze(_capi.Arrays.from_dict(bad))
    # a track longer than a chunk (256 observations) is a "heavy point": no chunk, every camera pair in the pattern
    n = 300
    long = dict(cam_pose=np.zeros((n, 6)), cam_intr=np.ones((n, 4)), cam_fixed=np.zeros(n, np.uint8),
                points=np.zeros((2, 3)), obs_cam=np.arange(n), obs_pt=np.zeros(n, np.int32), obs_xy=np.zeros((n, 2)))
    long["cam_fixed"][7] = 1
    a = _capi.analyze(_capi.Arrays.from_dict(long))
    assert a["n_heavy_points"] == 1 and a["max_track"] == n and a["n_chunks"] == 0 and a["n_observed_points"] == 1
    assert a["nnz_blocks_upper"] == (n - 1) * n // 2
    # 256 is still a chunk; mixed problems keep both kinds
    p = synthetic.make_ring(270, 40, seed=2, mean_track=12.0, n_landmarks=2)
    a = _capi.analyze(_capi.Arrays.from_dict(p))
    assert a["n_heavy_points"] == 2 and a["n_observed_points"] == 40 and a["n_chunks"] >= 2
    p = synthetic.make_ring(256, 40, seed=2, mean_track=12.0, n_landmarks=2)
    a = _capi.analyze(_capi.Arrays.from_dict(p))
    assert a["n_heavy_points"] == 0 and a["max_track"] == 256


def test_rodrigues_round_trip_matches_oracle():
    from oracle import ba_oracle as O
    from oracle import cv2_standin
    rng = np.random.default_rng(0)
    for _ in range(50):
        r = rng.normal(size=3) * rng.choice([1e-9, 1e-3, 0.5, 2.5])
        R = geometry.rvec_to_matrix(r)
        np.testing.assert_allclose(R, cv2_standin.rodrigues_vec_to_mat(r), atol=1e-15)
        np.testing.assert_allclose(R, O.rodrigues(r), atol=1e-15)
        np.testing.assert_allclose(geometry.matrix_to_rvec(R), cv2_standin.rodrigues_mat_to_vec(R), atol=1e-15)
        if 1e-4 < np.linalg.norm(r) < 3.0:  # (OpenCV returns exactly 0 below sin(theta) = 1e-5)
            np.testing.assert_allclose(geometry.matrix_to_rvec(R), r, atol=1e-9)
    # theta = pi branch
    for ax in np.eye(3):
        R = geometry.rvec_to_matrix(np.pi * ax)
        np.testing.assert_allclose(np.abs(geometry.matrix_to_rvec(R)), np.pi * ax, atol=1e-6)


class _KP:
    def __init__(self, xy):
        self.pt = (float(xy[0]), float(xy[1]))


def _image_data_from_golden(g):
    ids = [int(i) for i in g["ids"]]
    image_data = {}
    for i, cid in enumerate(ids):
        fx, fy, cx, cy = g["cam_intr"][i]
        K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]], dtype=np.float32)
        kp = [_KP(xy) for xy in g[f"kp_{cid}"]]
        ref = [int(v) for v in g[f"ref_{cid}"]]
        image_data[cid] = (g["cam_R"][i].copy(), g["cam_t"][i].reshape(3, 1).copy(), K, ref,
                           np.zeros((len(kp), 128), np.float32), kp)
    pts = [[float(p[0]), float(p[1]), float(p[2]), 0.1, 0.2, 0.3] for p in g["points"]]
    return ids, image_data, pts


def test_marshalling_matches_reference_order_and_types():
    g = load_golden("marshalling")
    ids, image_data, pts = _image_data_from_golden(g)
    obs = BA.prepare_observations_from_image_data(image_data, ids, pts)
    assert [o["camera_id"] for o in obs] == [int(v) for v in g["obs_cam_id"]]
    assert [o["point_idx"] for o in obs] == [int(v) for v in g["obs_pt"]]
    np.testing.assert_array_equal(np.array([o["pixel"] for o in obs]), g["obs_xy"])
    assert all(o["pixel"].dtype == np.float64 for o in obs)
    cams = BA.prepare_cameras_from_image_data(image_data, ids)
    assert list(cams.keys()) == ids
    np.testing.assert_array_equal(np.array([cams[c]["K"] for c in ids]), g["cams_K"])
    assert cams[ids[0]]["t"].shape == tuple(g["cams_t_shape"]) == (3, 1)
    assert all(cams[c][k].dtype == np.float64 for c in ids for k in "RtK")
    # ragged inputs the reference skips: unknown id, short tuple, non-int / out-of-range references
    image_data[999] = (1, 2)
    ref = list(image_data[ids[0]][3]); ref[0] = np.int64(3); ref[1] = 10 ** 6
    image_data[ids[0]] = image_data[ids[0]][:3] + (ref,) + image_data[ids[0]][4:]
    obs2 = BA.prepare_observations_from_image_data(image_data, ids + [999, 1234], pts)
    assert len(obs2) <= len(obs)
    assert BA.prepare_observations_from_image_data({}, ids, pts) == []
    # update: in place, slots 0 and 1 only
    new = {ids[1]: {"R": np.eye(3), "t": np.ones((3, 1)), "K": None}, 4242: {"R": np.eye(3), "t": np.ones((3, 1))}}
    out = BA.update_image_data_from_ba_results(image_data, new)
    assert out is image_data and np.array_equal(image_data[ids[1]][0], np.eye(3)) and 4242 not in image_data


def test_session_marshalling_equals_the_dict_protocol():
    """the array-native path builds the same BaProblem arrays, in the same order, as
    prepare_cameras + prepare_observations + _to_arrays -- also on ragged input."""
    from structure_from_motion_amd.session import BASession
    g = load_golden("marshalling")
    ids, image_data, pts = _image_data_from_golden(g)
    sess = BASession()

    def both(image_data, visited, pts):
        cams = BA.prepare_cameras_from_image_data(image_data, visited)
        obs = BA.prepare_observations_from_image_data(image_data, visited, pts)
        cam_ids = list(cams.keys())
        want = BA._to_arrays(cams, pts, obs, cam_ids, set())
        got, got_ids, colours = sess.marshal(image_data, pts, visited)
        assert got_ids == cam_ids
        for k in ("cam_pose", "cam_intr", "points", "obs_cam", "obs_pt", "obs_xy"):
            np.testing.assert_array_equal(getattr(got, k), getattr(want, k), err_msg=k)
            assert getattr(got, k).dtype == getattr(want, k).dtype
        np.testing.assert_array_equal(colours, np.array(pts)[:, 3:6])
        return got

    a = both(image_data, ids, pts)
    assert len(a.obs_cam) == len(g["obs_pt"])
    both(image_data, ids[:4], pts)                                    # a prefix, as the incremental driver calls it
    both(image_data, ids, pts[:30])                                   # references beyond the point list are dropped
    # ragged: unknown id, short tuple, a numpy integer and an out-of-range reference, more refs than keypoints
    image_data[999] = (1, 2)
    ref = list(image_data[ids[0]][3]); ref[0] = np.int64(3); ref[1] = 10 ** 6
    image_data[ids[0]] = image_data[ids[0]][:3] + (ref,) + image_data[ids[0]][4:]
    e = image_data[ids[1]]
    image_data[ids[1]] = e[:3] + (list(e[3]) + [5, 6],) + e[4:]
    both(image_data, ids + [999, 1234], pts)
    assert sess._kp_xy[ids[2]][2].shape == (len(image_data[ids[2]][5]), 2)   # keypoints converted once, cached
    # schedule / guards as inter_ba (:314, :325)
    assert sess.inter_ba(image_data, pts, ids[:3], ba_interval=2, verbose=False) == (pts, image_data, False)
    assert sess.inter_ba({}, pts, ids[:4], ba_interval=2, verbose=False)[2] is False


def test_reference_success_rule():
    """a7 (BundleAdjustment.py:223-224): success = status > 0 OR cost < 0.95 x the signed residual sum."""
    rule = BA._reference_success
    assert rule(dict(status=2, final_cost=10.0, residual_sum=-5.0)) == (True, True)
    assert rule(dict(status=0, final_cost=10.0, residual_sum=10.6)) == (True, False)     # the quirk: 10 < 0.95 * 10.6
    assert rule(dict(status=0, final_cost=10.0, residual_sum=10.5)) == (False, False)    # 10 < 9.975 is false
    assert rule(dict(status=0, final_cost=10.0, residual_sum=-100.0)) == (False, False)  # signed: a negative sum never passes


def test_session_keypoint_cache_is_tied_to_the_list_object():
    from structure_from_motion_amd.session import BASession
    sess = BASession()
    kp = [_KP((1.0, 2.0)), _KP((3.0, 4.0))]
    a = sess._keypoints(7, kp)
    assert sess._keypoints(7, kp) is a                      # same list object: cached
    kp2 = [_KP((5.0, 6.0)), _KP((7.0, 8.0))]                # another list of the same length: never the cached one
    b = sess._keypoints(7, kp2)
    assert b is not a and b[0, 0] == 5.0
    kp2[0].pt = (9.0, 9.0)                                  # in-place edit of the first / last keypoint is noticed
    assert sess._keypoints(7, kp2)[0, 0] == 9.0
    sess.invalidate(7)
    assert 7 not in sess._kp_xy


def test_guards_and_schedule_without_gpu():
    cams = {i: {"R": np.eye(3), "t": np.zeros((3, 1)), "K": np.eye(3)} for i in range(2)}
    pts = [[0, 0, 1, 0, 0, 0]] * 20
    r = BA.global_bundle_adjustment(cams, pts, [], [0, 1])
    assert r[0] is cams and r[1] is pts and r[2] is False          # < 3 cameras
    cams3 = {i: cams[0] for i in range(3)}
    r = BA.global_bundle_adjustment(cams3, pts[:9], [], [0, 1, 2])
    assert r[2] is False                                             # < 10 points
    image_data = {}
    r = BA.inter_ba(image_data, pts, [0, 1, 2], ba_interval=2, verbose=False)
    assert r == (pts, image_data, False)                             # 3 % 2 != 0
    r = BA.inter_ba(image_data, pts, [0, 1], ba_interval=2, verbose=False)
    assert r[2] is False                                             # fewer than 3 images
    r = BA.inter_ba(image_data, pts, [0, 1, 2, 3], ba_interval=2, verbose=False)
    assert r[2] is False                                             # no cameras prepared -> insufficient
    assert BA.compute_ba_reprojection_errors({}, pts, [0, 1]) == {}
    import BundleAdjustment as shim                                   # the root import shim
    assert set(shim.__all__) == set(BA.__all__) and shim.final_ba is BA.final_ba


def test_synthetic_shapes():
    p = synthetic.make_config("ladybug1723", scale=0.01)
    assert len(p["cam_pose"]) == 1723 and len(p["obs_cam"]) == int(678718 * 0.01)
    assert np.all(np.diff(p["obs_cam"]) >= 0)          # camera-major observation order
    assert np.bincount(p["obs_pt"]).min() >= 2
    t = synthetic.make_config("temple", scale=0.05)
    assert len(t["cam_pose"]) == 47 and t["cam_fixed"].sum() == 1
    f = synthetic.make_config("fountain", scale=0.05)
    assert len(f["cam_pose"]) == 11 and f["cam_fixed"].sum() == 2


def test_analyze_camera_slots_of_wide_chunks():
    """Chunk-local camera slots (csrc/ba_symbolic.h): up to 28 cameras of a chunk get a full LDS tile, up to 112 a slot
    ("wide" chunks: unstructured visibility), only the cameras beyond that add to HBM observation by observation."""
    p = synthetic.make_street(60, 1500, 6000, seed=3, max_track=12)
    a = _capi.analyze(_capi.Arrays.from_dict(p))
    assert a["max_chunk_cams"] <= 28 and a["n_direct_cam_obs"] == 0
    p = synthetic.make_ring(90, 1500, seed=78, visibility="random", mean_track=7.0, max_track=20, n_fixed=2, radius=6.0)
    a = _capi.analyze(_capi.Arrays.from_dict(p))
    assert 28 < a["max_chunk_cams"] <= 112 and a["n_direct_cam_obs"] == 0
    free = p["cam_fixed"][p["obs_cam"]] == 0
    cnt = np.bincount(p["obs_pt"][free], minlength=len(p["points"]))        # (pair entries: free cameras only)
    assert a["n_pair_entries"] == int((cnt * (cnt + 1) // 2).sum())
    p = synthetic.make_ring(400, 40, seed=79, visibility="random", mean_track=150.0, max_track=190, n_fixed=0, radius=6.0)
    a = _capi.analyze(_capi.Arrays.from_dict(p))
    cnt = np.bincount(p["obs_pt"], minlength=len(p["points"]))
    assert a["max_chunk_cams"] == 112 and a["n_heavy_points"] == 0
    assert a["n_direct_cam_obs"] == int(np.maximum(cnt - 112, 0).sum())     # (one point per chunk: its cameras past the 112th)
    assert a["n_pair_entries"] == int((cnt * (cnt + 1) // 2).sum())


def test_analysis_says_when_auto_takes_the_rows_mode():
    """AUTO's rule for the Schur accumulation (include/ba_hip.h, BaOptions.schur_mode): chunk-local pair groups that are
    mostly padding -- fewer than three real entries per group of eight, on a problem with at least 200 k entries."""
    street = _capi.analyze(_capi.Arrays.from_dict(synthetic.make_street(60, 9000, 40000, seed=2)))
    assert street["auto_rows_mode"] == 0 and street["n_pair_entries"] > 3 * street["n_groups"] > 0
    rnd = synthetic.make_ring(300, 9000, seed=5, K=(1200.0, 1200.0, 960.0, 540.0), visibility="random", mean_track=8.0, max_track=12,
                              n_fixed=2, radius=6.0, noise=0.5)
    a = _capi.analyze(_capi.Arrays.from_dict(rnd))
    assert a["auto_rows_mode"] == 1 and 200000 <= a["n_pair_entries"] < 3 * a["n_groups"]
    small = synthetic.make_ring(40, 900, seed=12, visibility="random", mean_track=5.0, max_track=9, n_fixed=2, radius=6.0, noise=0.5)
    assert _capi.analyze(_capi.Arrays.from_dict(small))["auto_rows_mode"] == 0     # too small to matter
    with pytest.raises(TypeError):
        _capi.default_options(schur_rows=1)
    assert _capi.default_options(schur_mode="rows").schur_mode == 2
