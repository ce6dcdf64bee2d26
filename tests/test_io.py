"""On-disk formats next to the path: the reference's ASCII PLY and BAL text problems."""
import numpy as np

from structure_from_motion_amd import io as sfm_io
from structure_from_motion_amd import synthetic
from structure_from_motion_amd.geometry import project_points, rvec_to_matrix

# what utils.Toply writes first (utils.py:337-348; the same ten lines head src/temple_new.ply)
PLY_HEADER = ["ply", "format ascii 1.0", "element vertex {}", "property float x", "property float y",
              "property float z", "property uchar red", "property uchar green", "property uchar blue", "end_header"]


def test_ply_writer_matches_reference_format(tmp_path):
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(7, 3))
    clrs = rng.uniform(-0.1, 1.1, size=(7, 3))          # out-of-range colours are clipped (utils.py:335)
    fn = tmp_path / "out.ply"
    sfm_io.write_ply(pts, clrs, str(fn))
    lines = fn.read_text().splitlines()
    assert lines[:10] == [h.format(7) for h in PLY_HEADER]
    c8 = np.clip(clrs * 255.0, 0, 255).astype(np.uint8)
    for k in range(7):                                   # '{} {} {} {} {} {}'.format(...) of the reference loop
        assert lines[10 + k] == "{} {} {} {} {} {}".format(pts[k, 0], pts[k, 1], pts[k, 2], c8[k, 0], c8[k, 1], c8[k, 2])
    p2, c2 = sfm_io.read_ply(str(fn))
    np.testing.assert_array_equal(p2, pts)               # repr round-trips float64 exactly
    np.testing.assert_array_equal(c2, c8 / 255.0)
    sfm_io.points_list_to_ply([[1.0, 2.0, 3.0, 0.5, 0.25, 1.0]], str(fn))
    assert fn.read_text().splitlines()[10] == "1.0 2.0 3.0 127 63 255"
    sfm_io.write_ply(np.zeros((0, 3)), np.zeros((0, 3)), str(fn))
    p0, _ = sfm_io.read_ply(str(fn))
    assert p0.shape == (0, 3)


def _bal_project(cam9, X):
    """BAL's own camera model, straight from its definition."""
    P = X @ rvec_to_matrix(cam9[:3]).T + cam9[3:6]
    p = -P[:, :2] / P[:, 2:3]
    r2 = (p * p).sum(1)
    return cam9[6] * (1 + cam9[7] * r2 + cam9[8] * r2 * r2)[:, None] * p


def test_bal_reader_maps_the_camera_model(tmp_path):
    rng = np.random.default_rng(3)
    nc, npt = 5, 40
    cams = np.concatenate([rng.normal(scale=0.2, size=(nc, 3)), rng.normal(scale=0.3, size=(nc, 3)) + [0, 0, -6.0],
                           rng.uniform(500, 900, size=(nc, 1)), rng.normal(scale=1e-2, size=(nc, 1)),
                           rng.normal(scale=1e-3, size=(nc, 1))], axis=1)
    X = rng.normal(size=(npt, 3))
    oc = np.repeat(np.arange(nc), npt)
    op = np.tile(np.arange(npt), nc)
    xy = np.concatenate([_bal_project(cams[c], X) for c in range(nc)])
    fn = tmp_path / "p.txt"
    with open(fn, "w") as f:
        f.write(f"{nc} {npt} {len(oc)}\n")
        for c, p, (x, y) in zip(oc, op, xy):
            f.write(f"{c} {p}     {float(x)!r} {float(y)!r}\n")
        for v in cams.reshape(-1).tolist() + X.reshape(-1).tolist():
            f.write(f"{v!r}\n")
    prob, ex = sfm_io.read_bal(str(fn))
    assert prob["cam_fixed"].tolist() == [1, 1, 0, 0, 0]            # gauge rule for < 15 cameras
    np.testing.assert_array_equal(ex["k1"], cams[:, 7])
    # the path's pinhole model on the converted problem reproduces the (undistorted) BAL observations
    uv = project_points(prob["cam_pose"], prob["cam_intr"], prob["points"], prob["obs_cam"], prob["obs_pt"])
    np.testing.assert_allclose(uv, prob["obs_xy"], rtol=0, atol=1e-9)
    # round trip through the writer
    fn2 = tmp_path / "q.txt"
    sfm_io.write_bal(str(fn2), prob, ex)
    prob2, ex2 = sfm_io.read_bal(str(fn2))
    for k in ("cam_pose", "points", "obs_xy"):
        np.testing.assert_allclose(prob2[k], prob[k], rtol=0, atol=1e-9)
    np.testing.assert_array_equal(prob2["obs_cam"], prob["obs_cam"])
    np.testing.assert_array_equal(prob2["obs_pt"], prob["obs_pt"])
    tok = fn2.read_text().split()
    np.testing.assert_allclose(np.array(tok[3:3 + 4 * len(oc)], dtype=float).reshape(-1, 4)[:, 2:], xy, rtol=0, atol=1e-8)


def test_bal_writer_accepts_synthetic_shapes(tmp_path):
    p = synthetic.make_street(6, 50, 180, seed=2, n_fixed=2)
    p["cam_intr"][:, 1] = p["cam_intr"][:, 0]
    shift = p["cam_intr"][:, 2:].copy()
    p["obs_xy"] = p["obs_xy"] - shift[p["obs_cam"]]                # principal point to the origin
    p["cam_intr"][:, 2:] = 0.0
    fn = tmp_path / "s.txt"
    sfm_io.write_bal(str(fn), p)
    q, _ = sfm_io.read_bal(str(fn))
    uv_p = project_points(p["cam_pose"], p["cam_intr"], p["points"], p["obs_cam"], p["obs_pt"])
    uv_q = project_points(q["cam_pose"], q["cam_intr"], q["points"], q["obs_cam"], q["obs_pt"])
    np.testing.assert_allclose(uv_q, uv_p, rtol=0, atol=1e-8)
    np.testing.assert_allclose(q["obs_xy"], p["obs_xy"], rtol=0, atol=1e-9)
