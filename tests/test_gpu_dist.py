"""GPU: the real engine, point-sharded over two processes that share the one GPU of the test box
(collectives through torch.distributed; the multi-GPU run uses the same code over RCCL)."""
import numpy as np
import pytest
import torch.multiprocessing as mp
from dist_workers import engine_nccl_single_rank, engine_sharded_append, engine_sharded_empty_rank, engine_sharded_solve
from test_dist_cpu import _free_port

from structure_from_motion_amd import _capi, synthetic

pytestmark = pytest.mark.gpu


def test_two_rank_engine_solve_matches_single_handle(gpu, tmp_path):
    mp.spawn(engine_sharded_solve, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    p = synthetic.make_ring(16, 600, seed=8, mean_track=4.0, outlier_frac=0.02)
    arr = _capi.Arrays.from_dict(p)
    s = _capi.solve(arr, loss="huber", f_scale=1.0, max_iter=80)
    assert s["status"] > 0 and int(r0["status"]) > 0 and int(r0["status"]) == int(r1["status"])
    np.testing.assert_array_equal(r0["cp"], r1["cp"])                 # replicated cameras stay bit-identical
    assert abs(float(r0["cost"]) - float(r1["cost"])) == 0.0          # every rank sees the global cost
    assert abs(float(r0["cost"]) - s["final_cost"]) <= 1e-6 * s["final_cost"]
    pts = np.empty_like(p["points"])
    pts[r0["pidx"]] = r0["pts"]
    pts[r1["pidx"]] = r1["pts"]
    assert np.allclose(r0["cp"], arr.cam_pose, atol=1e-5) and np.allclose(pts, arr.points, atol=1e-4)


def test_rccl_callbacks_blocking_and_stream_ordered(gpu, tmp_path):
    """backend nccl (= RCCL) with one rank: the collective callbacks on tensors aliasing the library's device buffers,
    blocking and stream-ordered, reproduce the single-handle solve."""
    mp.spawn(engine_nccl_single_rank, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "nccl1.npz")
    p = synthetic.make_ring(16, 600, seed=8, mean_track=4.0, outlier_frac=0.02)
    arr = _capi.Arrays.from_dict(p)
    s = _capi.solve(arr, loss="huber", f_scale=1.0, max_iter=80)
    for name in ("blocking", "ordered", "native"):   # native = the library's own RCCL binding (ba_rccl_init), no Python in the loop
        assert int(r[name + "_status"]) == s["status"]
        assert abs(float(r[name + "_cost"]) - s["final_cost"]) <= 1e-9 * s["final_cost"]
        assert np.allclose(r[name + "_cp"], arr.cam_pose, atol=1e-6) and np.allclose(r[name + "_pts"], arr.points, atol=1e-5)
    print("build pass with collectives: blocking %.3f ms, stream-ordered %.3f ms, library-side RCCL %.3f ms"
          % (float(r["blocking_ms"]), float(r["ordered_ms"]), float(r["native_ms"])))


def test_sharded_solves_use_the_cyclic_reduction_beyond_the_dense_size(gpu, tmp_path, monkeypatch):
    """203 cameras on a street: AUTO's linear solver is the block cyclic reduction, run by every rank on the all-reduced
    reduced camera system.  Two gloo ranks, and the library's own RCCL binding with one rank, against the single handle:
    no PCG iteration anywhere, same number of LM steps, same minimum."""
    monkeypatch.setenv("BA_TEST_PROBLEM", "street203")
    p = synthetic.make_street(203, 5000, 22000, seed=51, outlier_frac=0.01)
    arr = _capi.Arrays.from_dict(p)
    s = _capi.solve(arr, loss="huber", f_scale=1.0, max_iter=25)
    assert s["pcg_iterations"] == 0 and s["accepted"] >= 3
    mp.spawn(engine_sharded_solve, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert int(r0["pcg"]) == 0 and int(r1["pcg"]) == 0
    assert int(r0["iters"]) == int(r1["iters"]) == s["iterations"] and int(r0["status"]) == int(r1["status"]) == s["status"]
    np.testing.assert_array_equal(r0["cp"], r1["cp"])
    assert abs(float(r0["cost"]) - s["final_cost"]) <= 1e-8 * s["final_cost"]
    assert np.allclose(r0["cp"], arr.cam_pose, rtol=1e-6, atol=1e-6)
    mp.spawn(engine_nccl_single_rank, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "nccl1.npz")
    for name in ("blocking", "ordered", "native"):
        assert int(r[name + "_pcg"]) == 0 and int(r[name + "_iters"]) == s["iterations"] and int(r[name + "_status"]) == s["status"]
        assert abs(float(r[name + "_cost"]) - s["final_cost"]) <= 1e-8 * s["final_cost"]
        assert np.allclose(r[name + "_cp"], arr.cam_pose, rtol=1e-6, atol=1e-6)


def test_two_ranks_in_the_rows_mode(gpu, tmp_path, monkeypatch):
    """The rows mode of the build pass (csrc/ba_rows.h) on point shards: every rank stores the blocks its own points give --
    the ranks' patterns differ, so S is cleared first and the all-reduce adds the shards up -- against the single handle."""
    monkeypatch.setenv("BA_TEST_PROBLEM", "random60")
    monkeypatch.setenv("BA_TEST_SCHUR", "rows")
    p = synthetic.make_ring(60, 2500, seed=31, K=(1200.0, 1200.0, 960.0, 540.0), focal_jitter=0.05, noise=0.5, outlier_frac=0.02,
                            visibility="random", mean_track=6.0, max_track=16, n_fixed=2, radius=6.0)
    arr = _capi.Arrays.from_dict(p)
    s = _capi.solve(arr, loss="huber", f_scale=1.0, max_iter=300, schur_mode="rows")
    assert s["status"] > 0
    mp.spawn(engine_sharded_solve, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert int(r0["status"]) == int(r1["status"]) and int(r0["status"]) > 0
    np.testing.assert_array_equal(r0["cp"], r1["cp"])
    assert abs(float(r0["cost"]) - s["final_cost"]) <= 1e-7 * s["final_cost"]
    assert np.allclose(r0["cp"], arr.cam_pose, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("mode", ["chunks", "rows"])
def test_two_ranks_with_heavy_points(gpu, tmp_path, monkeypatch, mode):
    """Landmarks seen by all 270 cameras (heavy points: rows from k_lin_heavy, pairs through the global entry lists -- added behind
    the chunk kernel in the chunk mode) on two point shards: the rank that owns a landmark contributes its blocks, the all-reduce
    adds the shards up; against the single handle."""
    monkeypatch.setenv("BA_TEST_PROBLEM", "landmarks270")
    monkeypatch.setenv("BA_TEST_SCHUR", mode)
    p = synthetic.make_ring(270, 110, seed=3, mean_track=30.0, max_track=60, n_landmarks=3, outlier_frac=0.05, n_fixed=2, n_unobserved=2)
    arr = _capi.Arrays.from_dict(p)
    assert _capi.analyze(arr)["n_heavy_points"] == 3
    s = _capi.solve(arr, loss="huber", f_scale=1.0, max_iter=40, schur_mode=mode)
    mp.spawn(engine_sharded_solve, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert int(r0["status"]) == int(r1["status"]) == s["status"]
    np.testing.assert_array_equal(r0["cp"], r1["cp"])
    assert abs(float(r0["cost"]) - s["final_cost"]) <= 1e-7 * s["final_cost"]
    assert np.allclose(r0["cp"], arr.cam_pose, rtol=1e-5, atol=1e-5)


def test_rank_without_observations_stays_in_step(gpu, tmp_path):
    mp.spawn(engine_sharded_empty_rank, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "empty0.npz"), np.load(tmp_path / "empty1.npz")
    p = synthetic.make_street(6, 60, 260, seed=3, outlier_frac=0.02)
    p["cam_fixed"] = np.ones(6, np.uint8)
    arr = _capi.Arrays.from_dict(p)
    s = _capi.solve(arr, loss="huber", f_scale=1.0, max_iter=40)
    for r in (r0, r1):
        assert int(r["status"]) == s["status"] and int(r["iters"]) == s["iterations"]
        assert abs(float(r["cost"]) - s["final_cost"]) <= 1e-9 * s["final_cost"]
        assert abs(float(r["cost0"]) - s["initial_cost"]) <= 1e-12 * s["initial_cost"]
        assert abs(float(r["rsum"]) - s["residual_sum"]) <= 1e-7 * max(1.0, abs(s["residual_sum"]))
    assert float(r0["cost"]) == float(r1["cost"]) and float(r0["rsum"]) == float(r1["rsum"])


def test_two_ranks_with_ten_thousand_replicated_cameras(gpu, tmp_path, monkeypatch):
    """BASELINE.json config 5 through the sharded path: all 10 000 cameras replicated on two gloo ranks, 1/20 of the
    points and observations (1 M / 5 M) cut in two by dist.shard_points.  Three LM steps: every rank solves the all-reduced
    reduced camera system with the 10 000-camera cyclic reduction (no PCG iteration); same steps, same cost and the same
    cameras as the single handle."""
    monkeypatch.setenv("BA_TEST_PROBLEM", "city_20th")
    p = synthetic.make_street(10000, 1_000_000, 5_000_000, seed=1233 + 5)
    with _capi.Handle(_capi.Arrays.from_dict(p), loss="huber", f_scale=1.0, max_iter=3) as h:
        s = h.run()
        cp, _ = h.state()
    assert s["pcg_iterations"] == 0 and s["accepted"] >= 1
    mp.spawn(engine_sharded_solve, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert int(r0["pcg"]) == 0 and int(r1["pcg"]) == 0
    assert int(r0["iters"]) == int(r1["iters"]) == s["iterations"] and int(r0["status"]) == int(r1["status"]) == s["status"]
    np.testing.assert_array_equal(r0["cp"], r1["cp"])
    assert float(r0["cost"]) == float(r1["cost"])
    # (three LM steps on a 10 000-camera chain amplify the rounding of the two summation orders -- one segmented sum, or two
    # shards joined by the all-reduce: measured 1.8e-9 relative on the cost)
    assert abs(float(r0["cost"]) - s["final_cost"]) <= 1e-7 * s["final_cost"]
    assert np.allclose(r0["cp"], cp, rtol=1e-5, atol=1e-5)


def test_deterministic_mode_across_ranks(gpu, tmp_path, monkeypatch):
    """BaOptions.deterministic on a sharded solve (determinism): two runs of the two-rank solve give the
    same bits, and the two-rank cost equals the single handle's to 1e-13 (what differs is only where the partial sums of
    the two shards are joined: inside one segmented sum, or by the all-reduce)."""
    monkeypatch.setenv("BA_TEST_PROBLEM", "street203")
    monkeypatch.setenv("BA_TEST_DET", "1")
    p = synthetic.make_street(203, 5000, 22000, seed=51, outlier_frac=0.01)
    arr = _capi.Arrays.from_dict(p)
    s = _capi.solve(arr, loss="huber", f_scale=1.0, max_iter=25, deterministic=1)
    runs = []
    for rep in range(2):
        d = tmp_path / f"rep{rep}"
        d.mkdir()
        mp.spawn(engine_sharded_solve, args=(2, _free_port(), str(d)), nprocs=2, join=True)
        runs.append((np.load(d / "rank0.npz"), np.load(d / "rank1.npz")))
    (a0, a1), (b0, b1) = runs
    np.testing.assert_array_equal(a0["cp"], b0["cp"])
    np.testing.assert_array_equal(a0["pts"], b0["pts"])
    np.testing.assert_array_equal(a1["pts"], b1["pts"])
    assert float(a0["cost"]) == float(b0["cost"]) == float(a1["cost"])
    assert int(a0["iters"]) == s["iterations"] and int(a0["pcg"]) == 0
    assert abs(float(a0["cost"]) - s["final_cost"]) <= 1e-13 * s["final_cost"]


def growing_problem():
    """(first, full): a ring problem with 14 of its 16 cameras and the points they see, and the whole of it -- the last
    two cameras and the points only they see arrive later, at the END of the camera / point / observation arrays (what
    the reference's driver appends per registered image, Reconstruction.py:244-251)."""
    p = synthetic.make_ring(16, 900, seed=41, visibility="sfm", n_seed_points=120, n_fixed=2, noise=0.3)
    late = p["obs_cam"] >= 14
    late_pts = np.unique(p["obs_pt"][late])
    assert not np.isin(p["obs_pt"][~late], late_pts).any()          # the new images' points are their own
    old_pts = np.setdiff1d(np.arange(len(p["points"])), late_pts)
    order = np.concatenate([old_pts, late_pts])
    remap = np.empty(len(order), dtype=np.int64)
    remap[order] = np.arange(len(order))
    oo = np.concatenate([np.flatnonzero(~late), np.flatnonzero(late)])
    full = dict(cam_pose=p["cam_pose"], cam_intr=p["cam_intr"], cam_fixed=p["cam_fixed"], points=p["points"][order],
                obs_cam=p["obs_cam"][oo], obs_pt=remap[p["obs_pt"][oo]].astype(np.int32), obs_xy=p["obs_xy"][oo])
    n_o1, n_p1 = int((~late).sum()), len(old_pts)
    first = dict(cam_pose=full["cam_pose"][:14], cam_intr=full["cam_intr"][:14], cam_fixed=full["cam_fixed"][:14],
                 points=full["points"][:n_p1], obs_cam=full["obs_cam"][:n_o1], obs_pt=full["obs_pt"][:n_o1], obs_xy=full["obs_xy"][:n_o1])
    return first, full


def test_append_on_sharded_handles(gpu, tmp_path):
    """f1 on more than one rank: ba_append on every rank's handle + the pattern union again = a fresh single handle on
    the grown problem (same initial cost bit for bit apart from the summation order, same minimum)."""
    mp.spawn(engine_sharded_append, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "app0.npz"), np.load(tmp_path / "app1.npz")
    first, full = growing_problem()
    a1 = _capi.Arrays.from_dict(first)
    s1 = _capi.solve(a1, loss="huber", f_scale=1.5, max_iter=40)
    assert abs(float(r0["cost1"]) - s1["final_cost"]) <= 1e-8 * s1["final_cost"]
    # the grown problem from the state the sharded handles were given: old cameras / points as the caller holds them
    # (the reference's driver passes its current image_data; here: the initial values again), new ones initial
    a2 = _capi.Arrays.from_dict(full)
    s2 = _capi.solve(a2, loss="huber", f_scale=1.5, max_iter=40)
    for r in (r0, r1):
        assert abs(float(r["cost2_0"]) - s2["initial_cost"]) <= 1e-12 * s2["initial_cost"]
        assert int(r["status2"]) == s2["status"] and abs(float(r["cost2"]) - s2["final_cost"]) <= 1e-7 * s2["final_cost"]
    np.testing.assert_array_equal(r0["cp"], r1["cp"])
    assert np.allclose(r0["cp"], a2.cam_pose, atol=1e-5)
    pts = np.empty_like(full["points"])
    pts[r0["keep"]] = r0["pts"]
    pts[r1["keep"]] = r1["pts"]
    assert np.allclose(pts, a2.points, atol=1e-4)
