"""GPU: the "rows" mode of the build pass (``BaOptions.schur_mode = 2``, csrc/ba_rows.h).

The reference hands any sparsity to scipy (BundleAdjustment.py:154-198, 210-220).  The chunk kernel relies on structure
inside a chunk (neighbouring points seeing the same cameras); for co-visibility without it the engine accumulates the
Schur complement from global per-block entry lists instead: per-observation rows in HBM, one owner per block of S.  Checked
here: the reduced system against the oracle on every structure kind (same tolerances as the chunk mode's tests), the LM
minimum against the chunk mode's, AUTO's choice, and the handle operations that depend on the analysed structure."""
import numpy as np
import pytest

from oracle import ba_oracle as O
from structure_from_motion_amd import _capi, synthetic

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def oracle_args(p):
    return p["cam_pose"], p["cam_intr"], p["points"], p["obs_cam"], p["obs_pt"]


def _problems():
    yield "street", synthetic.make_street(12, 300, 1300, seed=8, outlier_frac=0.05)
    yield "random", synthetic.make_ring(90, 1500, seed=78, K=(1200.0, 1200.0, 960.0, 540.0), focal_jitter=0.05, noise=0.5,
                                        outlier_frac=0.02, visibility="random", mean_track=7.0, max_track=20, n_fixed=2, radius=6.0)
    # landmarks seen by every camera (heavy points: rows from k_lin_heavy, pairs in the same global lists), unobserved points,
    # two fixed cameras
    yield "heavy", synthetic.make_ring(270, 110, seed=3, mean_track=30.0, max_track=60, n_landmarks=3, outlier_frac=0.05,
                                       n_fixed=2, n_unobserved=2)
    yield "long_tracks", synthetic.make_ring(400, 40, seed=79, K=(1200.0, 1200.0, 960.0, 540.0), focal_jitter=0.05, noise=0.5,
                                             visibility="random", mean_track=150.0, max_track=190, n_fixed=2, radius=6.0)
    base = synthetic.make_street(6, 40, 160, seed=2)
    p = dict(base)                                   # a camera that sees a point twice: both orders land in its diagonal block
    p["obs_cam"] = np.concatenate([base["obs_cam"], base["obs_cam"][:7]])
    p["obs_pt"] = np.concatenate([base["obs_pt"], base["obs_pt"][:7]])
    p["obs_xy"] = np.concatenate([base["obs_xy"], base["obs_xy"][:7] + 0.3])
    yield "twice", p
    q = dict(synthetic.make_street(8, 60, 250, seed=5))
    q["cam_fixed"] = np.ones(len(q["cam_pose"]), np.uint8)
    q["cam_fixed"][3] = 0                            # one free camera: no off-diagonal block at all
    yield "one_free_camera", q


@pytest.mark.parametrize("name,p", list(_problems()), ids=[n for n, _ in _problems()])
@pytest.mark.parametrize("loss", ["huber", "cauchy"])
def test_rows_mode_reduced_system_against_the_oracle(gpu, name, p, loss):
    lam = 1e-3
    nb = O.normal_blocks(*oracle_args(p), p["obs_xy"], 1.0, loss, floor=1.0)
    S, gred, _, _, _ = O.reduced_system(nb, p["cam_fixed"], p["obs_cam"], p["obs_pt"], lam)
    free = p["cam_fixed"] == 0
    with _capi.Handle(_capi.Arrays.from_dict(p), loss=loss, f_scale=1.0, schur_mode="rows") as h:
        h.build(lam)
        r = h.reduced_system()
        assert rel(r["U"][free], nb["U"][free]) < 1e-12 and rel(r["gc"][free], nb["gc"][free]) < 1e-11
        assert rel(r["V"], nb["V"]) < 1e-12 and rel(r["gp"], nb["gp"]) < 1e-11
        assert rel(r["S"], S) < 1e-9 and rel(r["gred"], gred) < 1e-9
        assert abs(h.cost() - nb["F"]) <= 1e-12 * nb["F"]
        h.build(lam)                                  # every sum of the three kernels has a fixed order: the same bits again
        r2 = h.reduced_system()
        for k in ("S", "gred", "U", "gc", "V", "gp"):
            np.testing.assert_array_equal(r[k], r2[k])


@pytest.mark.parametrize("solver", ["auto", "pcg"])
def test_rows_mode_lm_minimum_equals_the_chunk_modes(gpu, solver):
    p = synthetic.make_ring(60, 2500, seed=31, K=(1200.0, 1200.0, 960.0, 540.0), focal_jitter=0.05, noise=0.5, outlier_frac=0.02,
                            visibility="random", mean_track=6.0, max_track=16, n_fixed=2, radius=6.0)
    res = {}
    for mode in ("chunks", "rows"):
        arr = _capi.Arrays.from_dict(p)
        s = _capi.solve(arr, loss="huber", f_scale=1.0, max_iter=300, linear_solver=solver, pcg_tol=1e-10, schur_mode=mode)
        assert s["status"] > 0
        res[mode] = (s, arr.cam_pose.copy(), arr.points.copy())
    a, b = res["chunks"], res["rows"]
    assert abs(a[0]["final_cost"] - b[0]["final_cost"]) <= 1e-8 * a[0]["final_cost"]
    np.testing.assert_allclose(a[1], b[1], rtol=1e-5, atol=1e-6)
    # ... and the oracle's own LM loop from the same start ends in the same minimum
    o = O.lm_solve(p["cam_pose"], p["cam_intr"], p["cam_fixed"], p["points"], p["obs_cam"], p["obs_pt"], p["obs_xy"], f_scale=1.0, kind="huber", max_iter=300)
    assert abs(o["cost"] - b[0]["final_cost"]) <= 1e-6 * o["cost"]


def test_auto_takes_the_rows_mode_without_chunk_structure(gpu):
    """Random co-visibility, enough of it to matter: AUTO analyses twice and takes the rows mode (same system as forced chunks);
    on a street it stays with the chunks."""
    p = synthetic.make_ring(300, 9000, seed=5, K=(1200.0, 1200.0, 960.0, 540.0), visibility="random", mean_track=8.0, max_track=12,
                            n_fixed=2, radius=6.0, noise=0.5)
    arr = _capi.Arrays.from_dict(p)
    a = _capi.analyze(arr)
    assert a["auto_rows_mode"] == 1 and a["n_pair_entries"] < 3 * a["n_groups"]
    assert _capi.analyze(_capi.Arrays.from_dict(synthetic.make_street(60, 9000, 40000, seed=2)))["auto_rows_mode"] == 0
    lam = 1e-4
    out = {}
    for mode in ("auto", "chunks", "rows"):
        with _capi.Handle(arr, loss="huber", f_scale=1.0, schur_mode=mode) as h:
            h.build(lam)
            out[mode] = (h.reduced_blocks(), None)
    kc, bc = out["chunks"][0][0], out["chunks"][0][2]
    for mode in ("auto", "rows"):
        k, b = out[mode][0][0], out[mode][0][2]
        np.testing.assert_array_equal(k, kc)
        assert rel(b, bc) < 1e-12


def test_rows_mode_handle_operations(gpu):
    """set_values keeps the structure and its mode; append on a rows handle rebuilds the structure inside the handle and ends
    where a new handle on the grown problem ends."""
    p = synthetic.make_ring(40, 900, seed=12, visibility="random", mean_track=5.0, max_track=9, n_fixed=2, radius=6.0, noise=0.5)
    arr = _capi.Arrays.from_dict(p)
    with _capi.Handle(arr, loss="huber", f_scale=1.0, max_iter=40, schur_mode="rows") as h:
        s1 = h.run()
        st1 = h.state()
        h.set_values(_capi.Arrays.from_dict(p))
        s2 = h.run()
        st2 = h.state()
        assert s1["iterations"] == s2["iterations"] and abs(s1["final_cost"] - s2["final_cost"]) <= 1e-12 * s1["final_cost"]
        np.testing.assert_allclose(st1[0], st2[0], rtol=1e-9, atol=1e-10)
        # grow: 200 new points seen by the old cameras
        rng = np.random.default_rng(3)
        g = dict(p)
        npts = len(p["points"])
        new_pts = p["points"][rng.integers(0, npts, 200)] + rng.normal(scale=0.05, size=(200, 3))
        g["points"] = np.vstack([p["points"], new_pts])
        oc, op, oxy = [], [], []
        for k in range(200):
            for c in rng.choice(len(p["cam_pose"]), 4, replace=False):
                oc.append(c); op.append(npts + k)
        oc, op = np.asarray(oc, np.int32), np.asarray(op, np.int32)
        garr0 = _capi.Arrays.from_dict({**g, "obs_cam": np.concatenate([p["obs_cam"], oc]), "obs_pt": np.concatenate([p["obs_pt"], op]),
                                        "obs_xy": np.concatenate([p["obs_xy"], np.zeros((len(oc), 2))])})
        proj = O.residuals(garr0.cam_pose, garr0.cam_intr, garr0.points, garr0.obs_cam, garr0.obs_pt, garr0.obs_xy).reshape(-1, 2)
        g["obs_cam"], g["obs_pt"] = garr0.obs_cam, garr0.obs_pt
        g["obs_xy"] = np.concatenate([p["obs_xy"], proj[len(p["obs_cam"]):] + rng.normal(scale=0.5, size=(len(oc), 2))])
        in_place = h.append(_capi.Arrays.from_dict(g))
        assert not in_place                              # (the entry lists are global: rebuilt)
        sa = h.run()
    sn = _capi.solve(_capi.Arrays.from_dict(g), loss="huber", f_scale=1.0, max_iter=40, schur_mode="rows")
    assert sa["iterations"] == sn["iterations"] and abs(sa["final_cost"] - sn["final_cost"]) <= 1e-10 * sn["final_cost"]


def test_rows_mode_full_size_orbit(gpu):
    """The object-centric shape of bench.py's other_configs (1 024 cameras around one object, 956 k observations, every point
    seen by a random set of cameras): the oracle's blocks and S v at full size in the mode AUTO takes there, and the time of a
    build pass in both modes."""
    p = synthetic.make_config("orbit1024")
    arr = _capi.Arrays.from_dict(p)
    nb = O.normal_blocks(*oracle_args(p), p["obs_xy"], 1.0, "huber", floor=1.0)
    lam = 1e-4
    free = p["cam_fixed"] == 0
    v = np.random.default_rng(1).normal(size=6 * int(free.sum()))
    Sv, gred = O.reduced_matvec(nb, p["cam_fixed"], p["obs_cam"], p["obs_pt"], lam, v)
    assert _capi.analyze(arr)["auto_rows_mode"] == 1
    with _capi.Handle(arr, loss="huber", f_scale=1.0) as h:
        h.build(lam)
        assert abs(h.cost() - nb["F"]) <= 1e-12 * nb["F"]
        r = h.reduced_system(want_points=True, want_dense=False)
        assert rel(r["U"][free], nb["U"][free]) < 1e-10 and rel(r["gc"][free], nb["gc"][free]) < 1e-10
        assert rel(r["V"], nb["V"]) < 1e-10 and rel(r["gp"], nb["gp"]) < 1e-10
        assert rel(r["gred"], gred) < 1e-8
        assert rel(h.schur_matvec(v), Sv) < 1e-8
        ms_rows, _ = h.build_bench(lam, warmup=2, reps=10)
    with _capi.Handle(arr, loss="huber", f_scale=1.0, schur_mode="chunks") as h:
        ms_chunks, _ = h.build_bench(lam, warmup=2, reps=10)
    print("orbit1024 build pass: rows %.3f ms (%.3g obs/s), chunks %.3f ms" % (ms_rows, len(p["obs_cam"]) / ms_rows * 1e3, ms_chunks))
    assert len(p["obs_cam"]) / ms_rows * 1e3 >= 1e9     # north_star's target on this shape too


@pytest.mark.parametrize("kind", ["random", "heavy"])
def test_deterministic_flag_with_the_rows_mode(gpu, kind):
    """BaOptions.deterministic = 1 together with schur_mode = rows: the rows kernels have no atomics, so the mode is taken as it
    is, heavy points included (k_lin_heavy sums in a fixed order).  Two LM runs on fresh handles, every cost and every camera
    bit for bit."""
    if kind == "random":
        p = synthetic.make_ring(60, 2500, seed=31, K=(1200.0, 1200.0, 960.0, 540.0), focal_jitter=0.05, noise=0.5, outlier_frac=0.02,
                                visibility="random", mean_track=6.0, max_track=16, n_fixed=2, radius=6.0)
    else:
        p = synthetic.make_ring(270, 110, seed=3, mean_track=30.0, max_track=60, n_landmarks=3, outlier_frac=0.05, n_fixed=2, n_unobserved=2)
    runs = []
    for _rep in range(2):
        with _capi.Handle(_capi.Arrays.from_dict(p), loss="huber", f_scale=1.0, max_iter=25, deterministic=1, schur_mode="rows",
                          linear_solver="pcg", pcg_tol=1e-10) as h:
            costs = []
            for _ in range(25):
                s = h.run(1)
                costs.append(s["final_cost"])
                if s["status"] != 0:
                    break
            cp, pts = h.state()
        runs.append((costs, cp, pts))
    assert len(runs[0][0]) >= 4 and runs[0][0] == runs[1][0]
    np.testing.assert_array_equal(runs[0][1], runs[1][1])
    np.testing.assert_array_equal(runs[0][2], runs[1][2])
    sa = _capi.solve(_capi.Arrays.from_dict(p), loss="huber", f_scale=1.0, max_iter=25, linear_solver="pcg", pcg_tol=1e-10, schur_mode="chunks")
    assert abs(sa["final_cost"] - runs[0][0][-1]) <= 1e-8 * sa["final_cost"]


def _landmarks_problem(n_cams, n_landmarks, seed=4):
    """A street, plus landmarks seen by EVERY camera (tracks of n_cams > 256 observations: heavy points), one of them twice by
    camera 5, two cameras fixed."""
    p = dict(synthetic.make_street(n_cams, n_cams * 6, n_cams * 30, seed=seed, outlier_frac=0.02))
    rng = np.random.default_rng(seed)
    n0 = len(p["points"])
    # (the street's cameras sit at (i, 0, 0) and look down +z: far ahead of the middle of the street is in front of all of them)
    lm = np.stack([0.5 * n_cams + rng.uniform(-5.0, 5.0, n_landmarks), rng.uniform(-3.0, 3.0, n_landmarks),
                   rng.uniform(8.0, 12.0, n_landmarks) * n_cams], axis=1)
    p["points"] = np.concatenate([p["points"], lm])
    cams = np.tile(np.arange(n_cams, dtype=np.int32), n_landmarks)
    pts = np.repeat(np.arange(n0, n0 + n_landmarks, dtype=np.int32), n_cams)
    cams = np.concatenate([cams, [5]]).astype(np.int32)
    pts = np.concatenate([pts, [n0]]).astype(np.int32)
    xy = synthetic.project_points(p["cam_pose"], p["cam_intr"], p["points"], cams, pts) + rng.normal(0.0, 0.5, (len(cams), 2))
    perm = rng.permutation(len(p["obs_cam"]) + len(cams))      # (the caller's order is arbitrary)
    p["obs_cam"] = np.concatenate([p["obs_cam"], cams])[perm]
    p["obs_pt"] = np.concatenate([p["obs_pt"], pts])[perm]
    p["obs_xy"] = np.concatenate([p["obs_xy"], xy])[perm]
    p["cam_fixed"] = np.zeros(n_cams, np.uint8)
    p["cam_fixed"][[0, 7]] = 1
    return p


@pytest.mark.parametrize("mode", ["chunks", "rows"])
@pytest.mark.parametrize("loss", ["huber", "cauchy"])
def test_heavy_points_through_the_entry_lists(gpu, mode, loss):
    """Landmarks seen by all 300 cameras next to ordinary street tracks.  In the chunk mode the chunks go through k_build and the
    landmarks' pairs are ADDED from the global lists (k_lin_heavy, k_rows_schur<ADD>); in the rows mode they sit in the
    lists with everybody else.  Same reduced system as the oracle's either way."""
    p = _landmarks_problem(300, 4)
    lam = 1e-3
    a = _capi.analyze(_capi.Arrays.from_dict(p))
    assert a["n_heavy_points"] == 4 and a["max_track"] == 301
    nb = O.normal_blocks(*oracle_args(p), p["obs_xy"], 1.0, loss, floor=1.0)
    S, gred, _, _, _ = O.reduced_system(nb, p["cam_fixed"], p["obs_cam"], p["obs_pt"], lam)
    free = p["cam_fixed"] == 0
    with _capi.Handle(_capi.Arrays.from_dict(p), loss=loss, f_scale=1.0, schur_mode=mode) as h:
        h.build(lam)
        r = h.reduced_system()
        assert rel(r["U"][free], nb["U"][free]) < 1e-12 and rel(r["gc"][free], nb["gc"][free]) < 1e-11
        assert rel(r["V"], nb["V"]) < 1e-11 and rel(r["gp"], nb["gp"]) < 1e-11
        assert rel(r["S"], S) < 1e-9 and rel(r["gred"], gred) < 1e-9
        assert abs(h.cost() - nb["F"]) <= 1e-12 * nb["F"]


def test_heavy_points_cost_what_their_pairs_cost(gpu):
    """Five landmarks seen by all 1 778 cameras of a street: 5 x 1.6 M pairs.  One workgroup per landmark with atomics took
    100 ms per build pass; through the entry lists the pass stays within a few milliseconds."""
    p = _landmarks_problem(1778, 5)
    arr = _capi.Arrays.from_dict(p)
    with _capi.Handle(arr, loss="huber", f_scale=1.0) as h:
        ms, _ = h.build_bench(1e-3, warmup=1, reps=5)
    print("street1778 + 5 landmarks: build pass %.3f ms" % ms)
    assert ms < 10.0


@pytest.mark.parametrize("mode", ["chunks", "rows"])
def test_heavy_points_seen_by_fixed_cameras_only(gpu, mode):
    """A landmark whose every observation belongs to a FIXED camera gives the reduced system nothing (no pair, no camera block):
    in the chunk mode there is then no entry table at all -- only the point's own block and its cost."""
    p = dict(synthetic.make_street(300, 1800, 9000, seed=6, outlier_frac=0.02))
    rng = np.random.default_rng(6)
    n0 = len(p["points"])
    p["points"] = np.concatenate([p["points"], [[150.0, 1.0, 3000.0]]])
    cams = np.arange(270, dtype=np.int32)                       # the landmark: cameras 0 .. 269, all of them fixed below
    pts = np.full(270, n0, np.int32)
    xy = synthetic.project_points(p["cam_pose"], p["cam_intr"], p["points"], cams, pts) + rng.normal(0.0, 0.5, (270, 2))
    p["obs_cam"] = np.concatenate([p["obs_cam"], cams]); p["obs_pt"] = np.concatenate([p["obs_pt"], pts]); p["obs_xy"] = np.concatenate([p["obs_xy"], xy])
    p["cam_fixed"] = np.ones(300, np.uint8)
    p["cam_fixed"][270:] = 0
    lam = 1e-3
    assert _capi.analyze(_capi.Arrays.from_dict(p))["n_heavy_points"] == 1
    nb = O.normal_blocks(*oracle_args(p), p["obs_xy"], 1.0, "huber", floor=1.0)
    S, gred, _, _, _ = O.reduced_system(nb, p["cam_fixed"], p["obs_cam"], p["obs_pt"], lam)
    free = p["cam_fixed"] == 0
    with _capi.Handle(_capi.Arrays.from_dict(p), loss="huber", f_scale=1.0, schur_mode=mode) as h:
        h.build(lam)
        r = h.reduced_system()
        assert rel(r["U"][free], nb["U"][free]) < 1e-12 and rel(r["gc"][free], nb["gc"][free]) < 1e-11
        assert rel(r["V"], nb["V"]) < 1e-11 and rel(r["gp"], nb["gp"]) < 1e-11
        assert rel(r["S"], S) < 1e-9 and rel(r["gred"], gred) < 1e-9
        assert abs(h.cost() - nb["F"]) <= 1e-12 * nb["F"]
        s = h.run(5)
        assert s["final_cost"] < s["initial_cost"]
