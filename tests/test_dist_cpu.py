"""CPU, two processes over gloo: the point-sharded algorithm (what dist.py + the collective hook do
on GPUs) reproduces the unsharded step -- sharding helper, pattern union, all-reduce of the partial
reduced systems, broadcast of the camera step, local back-substitution."""
import socket

import numpy as np
import torch.multiprocessing as mp
from dist_workers import oracle_sharded_step

from oracle import ba_oracle as O
from structure_from_motion_amd import synthetic
from structure_from_motion_amd.dist import shard_points


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_shard_points_partitions_everything():
    p = synthetic.make_street(9, 200, 900, seed=3)
    seen_obs, seen_pts = 0, []
    for r in range(3):
        sh, pidx = shard_points(p, r, 3)
        seen_obs += len(sh["obs_cam"])
        seen_pts.append(pidx)
        assert sh["obs_pt"].min() >= 0 and sh["obs_pt"].max() < len(sh["points"])
        np.testing.assert_array_equal(sh["points"], p["points"][pidx])
        assert len(sh["cam_pose"]) == 9                       # cameras replicated
    assert seen_obs == len(p["obs_cam"])
    np.testing.assert_array_equal(np.concatenate(seen_pts), np.arange(200))
    counts = [len(shard_points(p, r, 3)[0]["obs_cam"]) for r in range(3)]
    assert max(counts) - min(counts) < 0.1 * len(p["obs_cam"])  # balanced by observation count


def test_two_rank_sharded_step_equals_unsharded(tmp_path):
    mp.spawn(oracle_sharded_step, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    p = synthetic.make_street(7, 90, 400, seed=4, outlier_frac=0.03)
    nb = O.normal_blocks(p["cam_pose"], p["cam_intr"], p["points"], p["obs_cam"], p["obs_pt"], p["obs_xy"], 1.0, "huber")
    S, gred, _, _, _ = O.reduced_system(nb, p["cam_fixed"], p["obs_cam"], p["obs_pt"], 1e-3)
    dc, dp = O.lm_step(nb, p["cam_fixed"], p["obs_cam"], p["obs_pt"], 1e-3)
    for r in (r0, r1):
        assert np.allclose(r["S"], S, rtol=1e-11, atol=1e-9 * np.abs(S).max())
        assert np.allclose(r["g"], gred, rtol=1e-10, atol=1e-10 * np.abs(gred).max())
        assert np.allclose(r["dc"], dc, rtol=1e-7, atol=1e-12)
        assert np.allclose(r["dp"], dp[r["pidx"]], rtol=1e-6, atol=1e-10)
        assert abs(r["F"] - nb["F"]) <= 1e-12 * nb["F"]
    np.testing.assert_array_equal(r0["dc"], r1["dc"])            # bit-identical cameras on every rank
    np.testing.assert_array_equal(r0["uni"], r1["uni"])          # one shared block pattern
    assert len(np.concatenate([r0["pidx"], r1["pidx"]])) == 90
