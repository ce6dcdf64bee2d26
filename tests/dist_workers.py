"""Worker functions for the multi-process tests (spawned with torch.multiprocessing)."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def _engine_problem():
    """The problem of the engine workers: a 16-camera ring (dense solver) or, with BA_TEST_PROBLEM=street203, a street
    chain of 203 cameras -- beyond the one-workgroup dense solver, so AUTO takes the block cyclic reduction, which every
    rank then runs on the all-reduced system (what a multi-GPU run of the bench workload does)."""
    from structure_from_motion_amd import synthetic
    if os.environ.get("BA_TEST_PROBLEM") == "street203":
        return synthetic.make_street(203, 5000, 22000, seed=51, outlier_frac=0.01), 25
    if os.environ.get("BA_TEST_PROBLEM") == "city_20th":
        # BASELINE.json config 5 at 1/20 of its points and observations, ALL of its 10 000 cameras (replicated on every rank)
        return synthetic.make_street(10000, 1_000_000, 5_000_000, seed=1233 + 5), 3
    if os.environ.get("BA_TEST_PROBLEM") == "random60":
        # unstructured co-visibility (the rows mode of the build pass when BA_TEST_SCHUR=rows)
        return synthetic.make_ring(60, 2500, seed=31, K=(1200.0, 1200.0, 960.0, 540.0), focal_jitter=0.05, noise=0.5, outlier_frac=0.02,
                                   visibility="random", mean_track=6.0, max_track=16, n_fixed=2, radius=6.0), 300
    if os.environ.get("BA_TEST_PROBLEM") == "landmarks270":
        # landmarks seen by every camera (tracks no chunk holds: heavy points, their pairs through the global entry lists)
        return synthetic.make_ring(270, 110, seed=3, mean_track=30.0, max_track=60, n_landmarks=3, outlier_frac=0.05, n_fixed=2,
                                   n_unobserved=2), 40
    return synthetic.make_ring(16, 600, seed=8, mean_track=4.0, outlier_frac=0.02), 80


def oracle_sharded_step(rank, world, port, out_dir):
    """CPU: one damped Gauss-Newton step of a point-sharded problem with the oracle's math; the
    partial reduced systems are summed with a gloo all-reduce exactly as dist.py does on the GPU."""
    import torch
    dist = _init(rank, world, port)
    from oracle import ba_oracle as O
    from structure_from_motion_amd import synthetic
    from structure_from_motion_amd.dist import shard_points, union_pattern
    p = synthetic.make_street(7, 90, 400, seed=4, outlier_frac=0.03)
    sh, pidx = shard_points(p, rank, world)
    lam = 1e-3
    nb = O.normal_blocks(sh["cam_pose"], sh["cam_intr"], sh["points"], sh["obs_cam"], sh["obs_pt"], sh["obs_xy"], 1.0, "huber")
    S, gred, Vinv, _, _ = O.reduced_system(nb, sh["cam_fixed"], sh["obs_cam"], sh["obs_pt"], 0.0 if False else lam)
    # reduced_system adds U + lam*D(U) per rank; the engine damps AFTER the all-reduce, so rebuild that way:
    free = np.flatnonzero(sh["cam_fixed"] == 0)
    for k, c in enumerate(free):
        Dc = np.clip(np.diag(nb["U"][c]), O.D_MIN, O.D_MAX)
        S[6 * k:6 * k + 6, 6 * k:6 * k + 6] -= lam * np.diag(Dc)
    buf = torch.from_numpy(np.concatenate([S.reshape(-1), gred, nb["U"][free].reshape(-1), [nb["F"]]]))
    dist.all_reduce(buf)
    n = S.shape[0]
    S_all = buf[:n * n].numpy().reshape(n, n).copy()
    g_all = buf[n * n:n * n + n].numpy().copy()
    U_all = buf[n * n + n:-1].numpy().reshape(len(free), 6, 6)
    for k in range(len(free)):
        S_all[6 * k:6 * k + 6, 6 * k:6 * k + 6] += lam * np.diag(np.clip(np.diag(U_all[k]), O.D_MIN, O.D_MAX))
    dc_flat = np.linalg.solve(S_all, -g_all)
    t = torch.from_numpy(dc_flat.copy())
    dist.broadcast(t, src=0)
    dc = np.zeros((len(sh["cam_pose"]), 6)); dc[free] = t.numpy().reshape(-1, 6)
    rhs = nb["gp"].copy()
    np.add.at(rhs, sh["obs_pt"], np.einsum("nij,ni->nj", nb["W"], dc[sh["obs_cam"]]))
    dp = -np.einsum("nij,nj->ni", Vinv, rhs)
    # pattern union (co-visible free-camera pairs) through the same helper the GPU path uses
    fidx = np.cumsum(sh["cam_fixed"] == 0) - 1
    pairs = set()
    for q in range(len(sh["points"])):
        cs = sorted(fidx[c] for c in sh["obs_cam"][sh["obs_pt"] == q] if sh["cam_fixed"][c] == 0)
        pairs.update((a, b) for i, a in enumerate(cs) for b in cs[i:])
    local = np.array(sorted(pairs), dtype=np.int32).reshape(-1, 2)

    def gather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out
    uni = union_pattern(local, gather)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), S=S_all, g=g_all, dc=dc, dp=dp, pidx=pidx, F=float(buf[-1]), uni=uni)
    dist.destroy_process_group()


def engine_sharded_solve(rank, world, port, out_dir):
    """GPU (one device shared by both ranks): the real engine, point-sharded, collectives through
    torch.distributed (gloo moves the device buffers here; on a multi-GPU node the same code runs
    over RCCL)."""
    import torch
    dist = _init(rank, world, port)
    torch.cuda.set_device(0)
    from structure_from_motion_amd import synthetic
    from structure_from_motion_amd import dist as badist
    p, max_iter = _engine_problem()
    sh, pidx = badist.shard_points(p, rank, world)
    h = badist.create_sharded_handle(sh, rank, world, loss="huber", f_scale=1.0, max_iter=max_iter, device=0,
                                     verbose=int(os.environ.get("BA_TEST_VERBOSE", "0")),
                                     deterministic=int(os.environ.get("BA_TEST_DET", "0")),
                                     schur_mode=os.environ.get("BA_TEST_SCHUR", "auto"))
    s = h.run()
    cp, pts = h.state()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), cp=cp, pts=pts, pidx=pidx, cost=s["final_cost"], status=s["status"],
             iters=s["iterations"], pcg=s["pcg_iterations"])
    h.close()
    dist.destroy_process_group()


def engine_nccl_single_rank(rank, world, port, out_dir):
    """GPU: RCCL itself (backend nccl, a world of one rank) under the sharded code path with the collectives forced
    on -- the aliasing of the library's buffers by torch tensors and both orderings of the callback (blocking, and
    stream-ordered on the handle's stream) are exactly what an 8-GPU run uses; only the peers are missing."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from structure_from_motion_amd import synthetic
    from structure_from_motion_amd import dist as badist
    p, max_iter = _engine_problem()
    sh, pidx = badist.shard_points(p, 0, 1)
    out = {}
    for name, ordered, native in (("blocking", False, False), ("ordered", True, False), ("native", True, True)):
        h = badist.create_sharded_handle(sh, 0, 1, stream_ordered=ordered, force_collectives=True, native=native, loss="huber",
                                         f_scale=1.0, max_iter=max_iter, device=0)
        ms_pass, _ = h.build_bench(1e-4, warmup=2, reps=20)          # back-to-back passes, collectives in between
        s = h.run()
        cp, pts = h.state()
        out[name + "_cp"], out[name + "_pts"] = cp, pts
        out[name + "_cost"], out[name + "_status"], out[name + "_ms"] = s["final_cost"], s["status"], ms_pass
        out[name + "_pcg"], out[name + "_iters"] = s["pcg_iterations"], s["iterations"]
        h.close()
    np.savez(os.path.join(out_dir, "nccl1.npz"), **out)
    dist.destroy_process_group()


def engine_sharded_empty_rank(rank, world, port, out_dir):
    """GPU, two ranks on one device: every camera anchored and ALL observed points on rank 0 -- rank 1 owns unobserved
    points only and has nothing to optimise, but must stay in step with rank 0's collectives (it used to leave the LM
    loop on its own: deadlock), and both must report the WHOLE problem's cost and residual sum."""
    import torch
    dist = _init(rank, world, port)
    torch.cuda.set_device(0)
    from structure_from_motion_amd import synthetic
    from structure_from_motion_amd import dist as badist
    p = synthetic.make_street(6, 60, 260, seed=3, outlier_frac=0.02)
    fixed = np.ones(6, np.uint8)
    if rank == 0:
        sh = dict(p, cam_fixed=fixed)
    else:
        sh = dict(cam_pose=p["cam_pose"], cam_intr=p["cam_intr"], cam_fixed=fixed, points=np.ones((5, 3)),
                  obs_cam=np.zeros(0, np.int32), obs_pt=np.zeros(0, np.int32), obs_xy=np.zeros((0, 2)))
    h = badist.create_sharded_handle(sh, rank, world, loss="huber", f_scale=1.0, max_iter=40, device=0)
    s = h.run()
    np.savez(os.path.join(out_dir, f"empty{rank}.npz"), cost=s["final_cost"], rsum=s["residual_sum"], status=s["status"],
             iters=s["iterations"], cost0=s["initial_cost"])
    h.close()
    dist.destroy_process_group()


def engine_sharded_append(rank, world, port, out_dir):
    """GPU, two ranks on one device: the sharded handles solve a problem with its first cameras and points, then the
    problem GROWS (the remaining cameras, the points only they see, appended at the end of every rank's shard) --
    ba_append on each rank, the pattern union again, and the next solve."""
    import torch
    dist = _init(rank, world, port)
    torch.cuda.set_device(0)
    from structure_from_motion_amd import dist as badist
    from test_gpu_dist import growing_problem
    first, full = growing_problem()
    sh1, _ = badist.shard_points(first, rank, world)
    h = badist.create_sharded_handle(sh1, rank, world, loss="huber", f_scale=1.5, max_iter=40, device=0)
    s1 = h.run()
    # this rank's grown shard: its old points first (same order), then its share of the new points
    n1 = len(first["points"])
    new_pts = np.arange(n1, len(full["points"]))
    mine_new = new_pts[rank::world]
    old_idx = badist.shard_points(first, rank, world)[1]
    keep = np.concatenate([old_idx, mine_new])
    remap = -np.ones(len(full["points"]), dtype=np.int64)
    remap[keep] = np.arange(len(keep))
    m_old = np.isin(full["obs_pt"][:len(first["obs_cam"])], old_idx)
    sel_old = np.flatnonzero(m_old)
    sel_new = len(first["obs_cam"]) + np.flatnonzero(np.isin(full["obs_pt"][len(first["obs_cam"]):], mine_new))
    sel = np.concatenate([sel_old, sel_new])
    grown = dict(cam_pose=full["cam_pose"], cam_intr=full["cam_intr"], cam_fixed=full["cam_fixed"], points=full["points"][keep],
                 obs_cam=full["obs_cam"][sel], obs_pt=remap[full["obs_pt"][sel]].astype(np.int32), obs_xy=full["obs_xy"][sel])
    appended = badist.append_sharded(h, grown, world, loss="huber", f_scale=1.5, max_iter=40, device=0)
    s2 = h.run()
    cp, pts = h.state()
    np.savez(os.path.join(out_dir, f"app{rank}.npz"), cp=cp, pts=pts, keep=keep, cost1=s1["final_cost"], cost2=s2["final_cost"],
             status2=s2["status"], iters2=s2["iterations"], appended=int(appended), cost2_0=s2["initial_cost"])
    h.close()
    dist.destroy_process_group()
