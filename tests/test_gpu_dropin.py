"""GPU: the drop-in Python interface (same calls as the reference's Reconstruction.py makes)."""
import copy

import numpy as np
import pytest
from conftest import load_golden
from test_host_logic import _image_data_from_golden

from oracle import ba_oracle as O
from structure_from_motion_amd import BundleAdjustment as BA
from structure_from_motion_amd.geometry import matrix_to_rvec

pytestmark = pytest.mark.gpu


def _errors(image_data, pts, ids):
    out = {}
    P = np.array([p[:3] for p in pts], dtype=np.float64)
    for cid in ids:
        R, t, K, ref, _, kp = image_data[cid]
        idx = [(k, r) for k, r in enumerate(ref) if r != -1]
        X = P[[r for _, r in idx]]
        xy = np.array([kp[k].pt for k, _ in idx])
        pc = R @ X.T + t.reshape(3, 1)
        ph = K.astype(np.float64) @ pc
        out[cid] = float(np.mean(np.linalg.norm(xy - (ph[:2] / ph[2]).T, axis=1)))
    return out


def test_final_and_inter_ba_against_reference_results(gpu):
    g = load_golden("marshalling")
    ids, image_data, pts = _image_data_from_golden(g)
    assert bool(g["final_ok"]) and bool(g["inter_ok"])
    for fn, key in ((lambda d: BA.final_ba(d, pts, ids, verbose=False), "final"),
                    (lambda d: BA.inter_ba(d, pts, ids, ba_interval=2, verbose=True), "inter")):
        d = copy.deepcopy(image_data)
        out_pts, out_d, ok = fn(d)
        assert ok is True and out_d is d
        assert len(out_pts) == len(pts) and all(len(p) == 6 for p in out_pts)
        assert [p[3:] for p in out_pts] == [p[3:] for p in pts]            # colours preserved
        assert out_pts is not pts and pts[0][0] == float(g["points"][0, 0])  # inputs not mutated
        for i, cid in enumerate(ids):
            assert out_d[cid][0].shape == (3, 3) and out_d[cid][1].shape == (3, 1)
            assert out_d[cid][0].dtype == np.float64 and out_d[cid][1].dtype == np.float64
        # anchored cameras (first two visited, < 15 images) unchanged
        for cid in ids[:2]:
            np.testing.assert_array_equal(out_d[cid][0], image_data[cid][0])
        # per-camera mean reprojection error within 1e-3 px of the reference's converged result
        ref_d = copy.deepcopy(image_data)
        for i, cid in enumerate(ids):
            e = list(ref_d[cid]); e[0] = g[f"{key}_R"][i]; e[1] = g[f"{key}_t"][i].reshape(3, 1); ref_d[cid] = tuple(e)
        ref_pts = [[*g[f"{key}_pts"][i], 0.1, 0.2, 0.3] for i in range(len(pts))]
        e_ref = _errors(ref_d, ref_pts, ids)
        e_our = BA.compute_ba_reprojection_errors(out_d, out_pts, ids)
        e_chk = _errors(out_d, out_pts, ids)
        for cid in ids:
            assert abs(e_our[cid] - e_chk[cid]) < 1e-9          # HIP reprojection-error kernel vs NumPy formula
            assert abs(e_our[cid] - e_ref[cid]) < 1e-3
    r = BA.inter_ba(copy.deepcopy(image_data), pts, ids[:5], ba_interval=2, verbose=False)
    assert r[2] is False and bool(g["inter_skip_ok"]) is False


def test_prints_are_the_references(gpu, capsys):
    """The [BA] lines final_ba / inter_ba print, in order, are the reference's (captured from the reference on this very
    fixture; the solver's own iteration table and result line differ by construction and are skipped)."""
    g = load_golden("marshalling")
    ids, image_data, pts = _image_data_from_golden(g)

    def lines(fn):
        capsys.readouterr()
        fn()
        out = capsys.readouterr().out.splitlines()
        return [ln for ln in out if ln.startswith(("[BA]", "RUNNING", "=")) and not ln.startswith("[BA] Bundle adjustment converged")]

    got = lines(lambda: BA.final_ba(copy.deepcopy(image_data), pts, ids, verbose=True))
    assert got == ["=" * 60, "RUNNING FINAL BUNDLE ADJUSTMENT", "=" * 60,
                   "[BA] Starting BA with 6 cameras, 60 points, 180 observations", "[BA] Anchoring cameras: [7, 3]",
                   "[BA] Final BA completed successfully!"]
    got = lines(lambda: BA.inter_ba(copy.deepcopy(image_data), pts, ids, ba_interval=2, verbose=True))
    assert got == ["=" * 50, "RUNNING BUNDLE ADJUSTMENT - 6 images processed", "=" * 50,
                   "[BA] Starting BA with 6 cameras, 60 points, 180 observations", "[BA] Anchoring cameras: [7, 3]",
                   "[BA] Successfully optimized 6 cameras and 60 points"]
    got = lines(lambda: BA.final_ba(copy.deepcopy(image_data), pts[:5], ids, verbose=True))
    assert got == ["=" * 60, "RUNNING FINAL BUNDLE ADJUSTMENT", "=" * 60, "[BA] Insufficient data: 6 cameras, 5 points",
                   "[BA] Final BA failed"]
    got = lines(lambda: BA.inter_ba(copy.deepcopy(image_data), pts[:9], ids, ba_interval=2, verbose=True))
    assert got == ["=" * 50, "RUNNING BUNDLE ADJUSTMENT - 6 images processed", "=" * 50, "[BA] Insufficient data for BA"]
    capsys.readouterr()
    BA.final_ba(copy.deepcopy(image_data), pts, ids, verbose=True)
    out = capsys.readouterr().out
    assert "[BA] Bundle adjustment converged! Cost: 0.8592" in out          # the reference printed 0.859267
    # silent calls print only the unconditional line (:27)
    got = lines(lambda: BA.final_ba(copy.deepcopy(image_data), pts, ids, verbose=False))
    assert got == ["[BA] Starting BA with 6 cameras, 60 points, 180 observations"]


def test_global_ba_containers_and_gauge(gpu):
    g = load_golden("ring16_dense")
    nc = len(g["cam_pose"])
    ids = [int(v) for v in g["cam_ids"]]
    R = O.rodrigues(g["cam_pose"][:, :3])
    cams = {}
    for i, cid in enumerate(ids):
        fx, fy, cx, cy = g["cam_intr"][i]
        cams[cid] = {"R": R[i].copy(), "t": g["cam_pose"][i, 3:].reshape(3, 1).copy(),
                     "K": np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])}
    pts = [[*map(float, p), *map(float, c)] for p, c in zip(g["points"], g["colors"])]
    obs = [{"camera_id": ids[int(c)], "point_idx": int(p), "pixel": xy.copy()}
           for c, p, xy in zip(g["obs_cam"], g["obs_pt"], g["obs_xy"])]
    visited = [int(v) for v in g["visited_ids"]]
    new_c, new_p, ok = BA.global_bundle_adjustment(cams, pts, obs, visited, robust_loss="huber",
                                                   loss_f_scale=1.0, max_nfev=300, verbose=1)
    assert ok is True and list(new_c.keys()) == ids and nc >= 15
    assert new_c[visited[0]]["R"] is cams[visited[0]]["R"]          # anchored: shallow copy shares arrays
    assert new_c[visited[1]]["R"] is not cams[visited[1]]["R"]      # >= 15 images: second camera is free
    assert tuple(g["ref_out_t_shape"]) == new_c[ids[-1]]["t"].shape == (3, 1)
    assert len(new_p[0]) == int(g["ref_out_point_len"]) == 6
    # cost of the returned containers through the oracle objective: within 1e-6 of the reference's converged cost
    cp = np.array([np.concatenate([matrix_to_rvec(new_c[c]["R"]), new_c[c]["t"].reshape(3)]) for c in ids])
    P = np.array([p[:3] for p in new_p], dtype=np.float64)
    f = O.residuals(cp, g["cam_intr"], P, g["obs_cam"], g["obs_pt"], g["obs_xy"])
    c = O.cost(f, 1.0, "huber")
    assert int(g["ref_status"]) > 0
    assert c <= float(g["ref_cost"]) * (1 + 1e-6) and c <= float(g["tight_cost"]) * (1 + 1e-6)
    assert BA.last_summary["status"] > 0


def test_success_quirk_branch_is_reached(gpu, monkeypatch):
    """a7, the branch scipy alone never takes: status == 0 (max_nfev spent) but cost < 0.95 x the SIGNED residual sum
    (BundleAdjustment.py:223-224) -> the reference returns the new containers with True.  Every observed pixel is
    moved so that all residuals start near +4 px (Huber's linear zone: cost ~ sum(f - 0.5) < 0.95 sum f), and the
    damping starts high (lambda0 = 30: about 3 % of the Gauss-Newton step) so the single allowed step stays there."""
    from structure_from_motion_amd import _capi
    g = load_golden("ring16_dense")
    ids = [int(v) for v in g["cam_ids"]]
    R = O.rodrigues(g["cam_pose"][:, :3])
    cams = {cid: {"R": R[i].copy(), "t": g["cam_pose"][i, 3:].reshape(3, 1).copy(),
                  "K": np.array([[g["cam_intr"][i, 0], 0, g["cam_intr"][i, 2]], [0, g["cam_intr"][i, 1], g["cam_intr"][i, 3]], [0, 0, 1.0]])}
            for i, cid in enumerate(ids)}
    pts = [[*map(float, p), 0.5, 0.5, 0.5] for p in g["points"]]
    proj = O.residuals(g["cam_pose"], g["cam_intr"], g["points"], g["obs_cam"], g["obs_pt"], np.zeros_like(g["obs_xy"])).reshape(-1, 2)
    obs = [{"camera_id": ids[int(c)], "point_idx": int(p), "pixel": xy - 4.0}
           for c, p, xy in zip(g["obs_cam"], g["obs_pt"], proj)]
    visited = [int(v) for v in g["visited_ids"]]
    real_defaults = _capi.default_options
    monkeypatch.setattr(_capi, "default_options", lambda **kw: real_defaults(**{**kw, "lambda0": 30.0}))
    new_c, new_p, ok = BA.global_bundle_adjustment(cams, pts, obs, visited, robust_loss="huber", loss_f_scale=1.0,
                                                   max_nfev=1, verbose=1)
    s = BA.last_summary
    assert s["status"] == 0 and s["iterations"] == 1
    assert s["final_cost"] < 0.95 * s["residual_sum"]                     # the quirk's condition holds ...
    assert ok is True and new_c is not cams and new_p is not pts           # ... and is honoured
    assert BA._reference_success(s) == (True, False)
    # without the quirk's condition a status-0 result is refused and the ORIGINAL objects come back
    obs2 = [{"camera_id": o["camera_id"], "point_idx": o["point_idx"], "pixel": o["pixel"] + 8.0} for o in obs]   # all residuals near -4
    out = BA.global_bundle_adjustment(cams, pts, obs2, visited, robust_loss="huber", loss_f_scale=1.0, max_nfev=1, verbose=0)
    assert BA.last_summary["status"] == 0 and BA.last_summary["residual_sum"] < 0
    assert out[2] is False and out[0] is cams and out[1] is pts


def _reference_errors(image_data, pts, ids):
    """utils.ReprojectionError (utils.py:271-276) as compute_ba_reprojection_errors applies it (:388-423)."""
    return _errors(image_data, pts, ids)


def test_reprojection_errors_use_R_as_stored(gpu):
    """a13: the reference multiplies by the 3x3 R exactly as image_data holds it (utils.py:273).  A float32-derived R, a
    slightly non-orthonormal R (a PnP pose before any BA) and a rotation below 1e-5 rad (which cv2.Rodrigues would turn
    into the zero vector) must all give the reference formula's numbers -- no R -> rvec -> R round trip."""
    g = load_golden("marshalling")
    ids, image_data, pts = _image_data_from_golden(g)
    rng = np.random.default_rng(3)
    e = list(image_data[ids[0]]); e[0] = e[0].astype(np.float32).astype(np.float64); image_data[ids[0]] = tuple(e)
    e = list(image_data[ids[1]]); e[0] = e[0] + 1e-3 * rng.normal(size=(3, 3)); image_data[ids[1]] = tuple(e)
    e = list(image_data[ids[2]]); e[0] = O.rodrigues(np.array([4e-6, -3e-6, 2e-6])); e[1] = np.array([[0.01], [0.0], [4.0]])
    image_data[ids[2]] = tuple(e)
    e = list(image_data[ids[3]]); e[0] = e[0].astype(np.float32); e[1] = e[1].astype(np.float32); image_data[ids[3]] = tuple(e)  # float32 containers
    want = _reference_errors(image_data, pts, ids)
    got = BA.compute_ba_reprojection_errors(image_data, pts, ids)
    assert list(got.keys()) == ids
    for cid in ids:
        assert abs(got[cid] - want[cid]) <= 1e-9 * max(1.0, want[cid]), cid
    # the rvec round trip the reference does NOT do would be visibly off for the perturbed cameras
    from structure_from_motion_amd.geometry import rvec_to_matrix
    d2 = copy.deepcopy(image_data)
    for cid in ids[:3]:
        e = list(d2[cid]); e[0] = rvec_to_matrix(matrix_to_rvec(e[0])); d2[cid] = tuple(e)
    off = _reference_errors(d2, pts, ids)
    assert max(abs(off[c] - want[c]) for c in ids[:3]) > 1e-4
    # a camera without valid references is left out, like the reference's `continue` (:413-414)
    e = list(image_data[ids[4]]); e[3] = [-1] * len(e[3]); image_data[ids[4]] = tuple(e)
    got = BA.compute_ba_reprojection_errors(image_data, pts, ids)
    assert ids[4] not in got and len(got) == len(ids) - 1


def test_resident_reprojection_errors_after_ba(gpu):
    """f2: right after a BA the session answers compute_ba_reprojection_errors from the handle's resident points and
    observations (Reconstruction.py:263 calls it after every BA); same numbers as the reference formula on the returned
    containers, and as the C-ABI probe with the handle's own poses."""
    from structure_from_motion_amd.session import BASession
    g = load_golden("marshalling")
    ids, image_data, pts = _image_data_from_golden(g)
    e = list(image_data[ids[0]]); e[0] = e[0] + 2e-4 * np.random.default_rng(1).normal(size=(3, 3)); image_data[ids[0]] = tuple(e)  # anchored, not quite orthonormal
    sess = BASession()
    p2, d2, ok = sess.final_ba(image_data, pts, ids, verbose=False)
    assert ok is True
    got = sess.compute_ba_reprojection_errors(d2, p2, ids)
    assert sess.resident_error_calls == 1                                  # the resident path ran
    want = _reference_errors(d2, p2, ids)
    assert list(got.keys()) == ids
    for cid in ids:
        assert abs(got[cid] - want[cid]) <= 1e-9 * max(1.0, want[cid])
    stateless = BA.compute_ba_reprojection_errors(d2, p2, ids)
    assert all(abs(stateless[c] - got[c]) <= 1e-9 for c in ids)
    # handle-level probe with the resident poses and the engine's own intrinsics: the free cameras agree with the above
    # (the anchored camera's stored R is not orthonormal; the engine holds its nearest rotation)
    err = sess._handle.reprojection_errors()
    for i, cid in enumerate(ids[2:], start=2):
        assert abs(err[i] - want[cid]) <= 1e-6
    # anything that is not the returned state falls back to the stateless path: a grown point list, another image set
    p3 = list(p2) + [[0.0, 0.0, 5.0, 0, 0, 0]]
    sess.compute_ba_reprojection_errors(d2, p3, ids)
    sess.compute_ba_reprojection_errors(d2, p2, ids[:4])
    e = list(d2[ids[3]]); e[0] = e[0].copy(); d2[ids[3]] = tuple(e)        # a replaced R object
    sess.compute_ba_reprojection_errors(d2, p2, ids)
    assert sess.resident_error_calls == 1
    sess.close()


def test_session_path_equals_module_functions(gpu):
    """BASession.final_ba / inter_ba return what the module functions return."""
    from structure_from_motion_amd.session import BASession
    g = load_golden("marshalling")
    ids, image_data, pts = _image_data_from_golden(g)
    sess = BASession()
    for name, kw in (("final_ba", {}), ("inter_ba", {"ba_interval": 2})):
        d1, d2 = copy.deepcopy(image_data), copy.deepcopy(image_data)
        p1, o1, ok1 = getattr(BA, name)(d1, pts, ids, verbose=False, **kw)
        p2, o2, ok2 = getattr(sess, name)(d2, pts, ids, verbose=False, **kw)
        assert ok1 is True and ok2 is True and o2 is d2
        np.testing.assert_allclose(np.array(p2), np.array(p1), rtol=0, atol=1e-7)
        assert [p[3:] for p in p2] == [list(p[3:]) for p in pts]          # colours preserved
        for cid in ids:
            np.testing.assert_allclose(d2[cid][0], d1[cid][0], rtol=0, atol=1e-8)
            np.testing.assert_allclose(d2[cid][1], d1[cid][1], rtol=0, atol=1e-7)
            assert d2[cid][0].dtype == np.float64 and d2[cid][1].shape == (3, 1)
            assert d2[cid][3] is image_data[cid][3] or d2[cid][3] == image_data[cid][3]
    assert sess.last_summary["status"] > 0
    assert sess.handle_reuses == 1        # inter_ba on the structure final_ba analysed: ba_set_values, no new handle
    sess.close()


def test_incremental_driver_pattern(gpu):
    """The reference driver's loop (Reconstruction.py:253-255): inter_ba after every new image on a growing set of
    registered images, results fed back into the next call.  The session (array marshalling, cached keypoints, kept
    handle) and the module functions must walk the same states."""
    from structure_from_motion_amd.session import BASession
    g = load_golden("marshalling")
    ids, image_data, pts = _image_data_from_golden(g)
    d1, d2 = copy.deepcopy(image_data), copy.deepcopy(image_data)
    p1, p2 = pts, pts
    sess = BASession()
    ran = 0
    for n in range(3, len(ids) + 1):
        for rep in range(2):                 # the second call sees the same structure: handle re-used, values re-uploaded
            p1, d1, ok1 = BA.inter_ba(d1, p1, ids[:n], ba_interval=1, verbose=False)
            p2, d2, ok2 = sess.inter_ba(d2, p2, ids[:n], ba_interval=1, verbose=False)
            assert ok1 == ok2
            ran += bool(ok2)
            np.testing.assert_allclose(np.array(p2)[:, :3], np.array(p1)[:, :3], rtol=0, atol=1e-5)
            for cid in ids[:n]:
                np.testing.assert_allclose(d2[cid][0], d1[cid][0], rtol=0, atol=1e-6)
                np.testing.assert_allclose(d2[cid][1], d1[cid][1], rtol=0, atol=1e-5)
    assert ran >= 2 and sess.handle_reuses >= ran // 2
    sess.close()


def test_session_appends_on_a_growing_reconstruction(gpu):
    """f1: the reference driver's loop on a reconstruction that GROWS the way Reconstruction.py:244-251 grows it (each
    new image brings its own points).  Every inter_ba of the session after the first extends the resident structure
    (ba_append) instead of building a new handle, and walks the states of the module functions."""
    from structure_from_motion_amd.session import BASession
    from structure_from_motion_amd import synthetic
    from test_gpu_parity import _grow_h4
    stages = _grow_h4(4, seed=3)
    ids_all, image_full, pts_full = synthetic.to_image_data(stages[-1])
    d1, d2 = {}, {}
    p1 = p2 = None
    sess = BASession()
    for k, st in enumerate(stages):
        ids = ids_all[:len(st["cam_pose"])]
        for c in ids:
            if c not in d1:
                d1[c] = copy.deepcopy(image_full[c]); d2[c] = copy.deepcopy(image_full[c])
        n_p = len(st["points"])
        p1 = (p1 or []) + copy.deepcopy(pts_full[len(p1 or []):n_p])
        p2 = (p2 or []) + copy.deepcopy(pts_full[len(p2 or []):n_p])
        r1 = BA.inter_ba(d1, p1, ids, ba_interval=1, verbose=False)
        r2 = sess.inter_ba(d2, p2, ids, ba_interval=1, verbose=False)
        assert r1[2] == r2[2] is True
        p1, p2 = r1[0], r2[0]
        np.testing.assert_allclose(np.array(p2)[:, :3], np.array(p1)[:, :3], rtol=0, atol=1e-5)
        for c in ids:
            np.testing.assert_allclose(d2[c][0], d1[c][0], rtol=0, atol=1e-6)
            np.testing.assert_allclose(d2[c][1], d1[c][1], rtol=0, atol=1e-5)
    assert sess.handle_reuses == len(stages) - 1 and sess.handle_appends == len(stages) - 1
    sess.close()


def test_opt_in_point_culling(gpu):
    """SURVEY 8(f) f4 (opt-in, not reference behaviour): a grossly wrong observation un-references its point."""
    from structure_from_motion_amd.session import BASession
    g = load_golden("marshalling")
    ids, image_data, pts = _image_data_from_golden(g)
    sess = BASession()
    assert len(sess.cull_points(image_data, pts, ids, max_error_px=1e4)) == 0
    k = next(i for i, r in enumerate(image_data[ids[2]][3]) if r != -1)
    victim = image_data[ids[2]][3][k]
    image_data[ids[2]][5][k].pt = (image_data[ids[2]][5][k].pt[0] + 500.0, image_data[ids[2]][5][k].pt[1])
    sess = BASession()                                                    # (keypoints changed: fresh cache)
    before = sum(r != -1 for c in ids for r in image_data[c][3])
    bad = sess.cull_points(image_data, pts, ids, max_error_px=100.0)
    assert bad.tolist() == [victim]
    assert all(r != victim for c in ids for r in image_data[c][3])
    assert sum(r != -1 for c in ids for r in image_data[c][3]) < before
    pts2, d2, ok = sess.final_ba(image_data, pts, ids, verbose=False)
    assert ok is True and pts2[victim][:3] == pts[victim][:3]             # unobserved point passes through


def test_verbose_2_prints_scipys_table(gpu, capfd):
    """The reference runs least_squares(verbose=2) (BundleAdjustment.py:219): scipy's progress table and closing lines.  The
    engine prints the same lines -- checked against scipy's own formatting functions: header, one row for the start and one per
    accepted step (centred 15-character columns), the termination message of the status and the summary line."""
    import contextlib
    import io
    from scipy.optimize._lsq import common
    from scipy.optimize._lsq.least_squares import TERMINATION_MESSAGES
    from structure_from_motion_amd import _capi, synthetic
    p = synthetic.make_street(6, 40, 160, seed=2)
    arr = _capi.Arrays.from_dict(p)
    capfd.readouterr()
    s = _capi.solve(arr, loss="huber", f_scale=1.0, max_iter=30, verbose=2)
    lines = [ln for ln in capfd.readouterr().out.split("\n") if ln.strip()]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        common.print_header_nonlinear()
    assert lines[0] == buf.getvalue().rstrip("\n")
    rows = lines[1:-2]
    assert len(rows) == s["accepted"] + 1
    for k, row in enumerate(rows):
        assert len(row) == 90
        f = row.split()
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            if k == 0:
                common.print_iteration_nonlinear(int(f[0]), int(f[1]), float(f[2]), None, None, float(f[3]))
            else:
                common.print_iteration_nonlinear(int(f[0]), int(f[1]), float(f[2]), float(f[3]), float(f[4]), float(f[5]))
        assert row == buf.getvalue().rstrip("\n") and int(f[0]) == k
    assert abs(float(rows[0].split()[2]) - s["initial_cost"]) <= 1e-4 * s["initial_cost"]
    assert abs(float(rows[-1].split()[2]) - s["final_cost"]) <= 1e-4 * s["final_cost"]
    assert lines[-2] == TERMINATION_MESSAGES[s["status"]]
    assert lines[-1] == "Function evaluations {}, initial cost {:.4e}, final cost {:.4e}, first-order optimality {:.2e}.".format(
        s["iterations"] + 1, s["initial_cost"], s["final_cost"], s["grad_inf"])
