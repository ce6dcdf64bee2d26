#!/usr/bin/env python3
"""Benchmark of the bundle-adjustment hot path on MI355X (contract: see the task brief).

A "step" is ONE residual + Jacobian + Schur-complement build pass (x, lambda) -> (cost, reduced
camera system, reduced gradient) over the whole synthetic problem, inputs already resident in HBM.

  python bench.py                         # 1 GPU, Venice-1778-shaped problem (BASELINE.json configs[3])
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

N > 1 (default --scaling strong, what BASELINE.json asks for): ONE problem, point-sharded over the N ranks
(cameras replicated); per step every rank builds its partial reduced camera system and the ranks all-reduce it over
RCCL (the library's own RCCL binding, no Python between kernels and collectives).  value = observations of the WHOLE
problem / max-over-ranks time.
  --workload venice1778   the Venice-shaped problem, cut N ways by dist.shard_points (same problem at every N)
  --workload city10k      10 000 cameras, 20 M points, 100 M observations in total, generated per shard (rank r makes
                          its 1/N of the points with its own seed; cameras from one seed)
--scaling weak: every rank owns a full-size shard of its own (N times the work).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
FP64_PEAK_TFLOPS = 78.6  # MI355X fp64 vector peak (same guide)


def algorithmic_bytes(n_obs, n_pts, n_cams):
    """one build pass, each array touched once = 40 B/obs + 96 B/point + 296 B/camera."""
    return 40.0 * n_obs + 96.0 * n_pts + 296.0 * n_cams


def cpu_baseline(prob, budget_obs):
    """The reference's scipy configuration (oracle/scipy_path.py, a from-scratch port) timed on this
    host: f(x) + one residual sweep per colour group = one finite-difference Jacobian, on a bounded
    sample (the first points of the workload and all their observations)."""
    from oracle import scipy_path
    cnt = np.bincount(prob["obs_pt"], minlength=len(prob["points"]))
    n_pts = int(np.searchsorted(np.cumsum(cnt), budget_obs)) + 1
    m = prob["obs_pt"] < n_pts
    sample = dict(cam_pose=prob["cam_pose"], cam_intr=prob["cam_intr"], cam_fixed=prob["cam_fixed"],
                  points=prob["points"][:n_pts], obs_cam=prob["obs_cam"][m], obs_pt=prob["obs_pt"][m],
                  obs_xy=prob["obs_xy"][m])
    r = scipy_path.time_linearisation(sample)
    # a fairer CPU bar: the oracle's vectorised NumPy restatement of the same pass -- residuals,
    # analytic Jacobian blocks, U / V / W / g accumulation, damped V inverse and one implicit Schur product
    from oracle import ba_oracle as O
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        nb = O.normal_blocks(sample["cam_pose"], sample["cam_intr"], sample["points"], sample["obs_cam"], sample["obs_pt"],
                             sample["obs_xy"], 1.0, "huber", floor=1.0)
        v = np.ones(6 * int((np.asarray(sample["cam_fixed"]) == 0).sum()))
        O.reduced_matvec(nb, sample["cam_fixed"], sample["obs_cam"], sample["obs_pt"], 1e-4, v)
        best = min(best, time.perf_counter() - t0)
    vec = {"value": r["n_obs"] / best, "unit": "observations/sec", "cores": 1, "kind": "port",
           "sample": "same sample; oracle/ba_oracle.py normal_blocks + one matrix-free Schur product (NumPy, analytic "
                     f"Jacobian), best of 3: {best:.2f} s"}
    return {"vectorised_numpy": vec, "value": r["obs_per_s"], "unit": "observations/sec", "cores": 1, "kind": "port",
            "sample": f"first {n_pts} points / {r['n_obs']} observations of the workload, residual closure + "
                      f"2-point sparse Jacobian = {r['sweeps']} residual sweeps in {r['seconds']:.1f} s "
                      f"(host has {os.cpu_count()} cores, the path is single-threaded Python)"}


def other_configs(_capi, synthetic):
    """BASELINE.json's other single-GPU configurations and the non-banded solver case, reported beside the headline
    line (not bench lines of their own)."""
    out = {}
    # Temple-ring-shaped final BA, three ways: (1) one ba_solve call from ready arrays; (2) final_ba of the drop-in
    # module from the reference's containers (image_data tuples, point lists) = what Reconstruction.py:302-304 pays,
    # marshalling included; (3) the same through BASession (array marshalling, cached keypoints)
    p = synthetic.make_config("temple")
    fx = os.path.join(ROOT, "tests", "golden", "temple_final.npz")
    temple_source = "synthetic.make_config('temple')"
    if os.path.exists(fx):
        # the fixture problem: the 20 467 vertices of the reference's own Temple cloud (src/temple_new.ply), 47 ring cameras,
        # at the reference's own x0 (tests/golden/make_golden.py --config-size)
        g = dict(np.load(fx, allow_pickle=False))
        free = g["cam_fixed"] == 0
        nf = int(free.sum())
        cam_pose = g["cam_pose"].copy()
        cam_pose[free] = g["x0"][:6 * nf].reshape(nf, 6)
        p = dict(cam_pose=cam_pose, cam_intr=g["cam_intr"], cam_fixed=g["cam_fixed"], points=g["x0"][6 * nf:].reshape(-1, 3).copy(),
                 obs_cam=g["obs_cam"], obs_pt=g["obs_pt"], obs_xy=g["obs_xy"])
        temple_source = "tests/golden/temple_final.npz (structure: the reference's temple_new.ply)"
    runs = []
    for _rep in range(3):          # (best of three, as for the host-API entries below; the first call of a process also pays
        arr = _capi.Arrays.from_dict(p)   # one-time costs: code objects, the page-locked staging buffer of the uploads)
        t0 = time.perf_counter()
        s = _capi.solve(arr, loss="huber", f_scale=1.0, max_iter=300)
        runs.append(((time.perf_counter() - t0) * 1e3, s))
    wall, s = min(runs, key=lambda r: r[0])
    out["temple_final_ba"] = {"cameras": len(p["cam_pose"]), "points": len(p["points"]), "observations": len(p["obs_cam"]),
                              "problem": temple_source,
                              "status": s["status"], "lm_steps": s["iterations"], "cost": [s["initial_cost"], s["final_cost"]],
                              "solve_ms": s["t_solve_ms"], "setup_ms": s["t_setup_ms"],
                              "wall_ms_from_arrays": wall, "wall_ms_from_arrays_first_call": runs[0][0]}
    try:
        import contextlib
        import io
        from structure_from_motion_amd import BundleAdjustment as BA
        from structure_from_motion_amd.session import BASession
        ids, image_data, pts = synthetic.to_image_data(p)
        import gc
        session = BASession()          # ONE session for all repetitions: its first call is the cold one, the rest are warm
        for name, fn in (("wall_ms_final_ba_dict_api", lambda d: BA.final_ba(d, pts, ids, verbose=False)),
                         ("wall_ms_final_ba_session", lambda d: session.final_ba(d, pts, ids, verbose=False))):
            times = []
            _pts = _d = None
            gc.collect()               # (the containers built above and the previous loop's results are the bench's own
                                       # garbage: a full collection over 100 k objects inside a timed call costs 60 ms)
            for _rep in range(4):      # (every call starts from the driver's containers again)
                d = dict(image_data)
                t0 = time.perf_counter()
                with contextlib.redirect_stdout(io.StringIO()):
                    _pts, _d, ok = fn(d)
                times.append((time.perf_counter() - t0) * 1e3)
            out["temple_final_ba"][name] = min(times[1:])                # warm: best of the later calls
            out["temple_final_ba"][name + "_first_call"] = times[0]      # cold: the first call
            out["temple_final_ba"][name.replace("wall_ms", "ok")] = bool(ok)
        out["temple_final_ba"]["session_handle_reuses"] = session.handle_reuses
        session.close()
    except Exception as e:  # never lose the headline line over a side measurement
        out["temple_final_ba"]["host_api_error"] = repr(e)
    # f1: the incremental driver's growth (Fountain-shaped, visibility as Reconstruction.py records it): the problem with
    # 9 of its 11 images resident, then the last two images and their points arrive -- ba_append against a new handle
    try:
        p = synthetic.make_config("fountain_sfm")
        m = p["obs_cam"] < 9
        npts = int(p["obs_pt"][m].max()) + 1
        first = dict(cam_pose=p["cam_pose"][:9], cam_intr=p["cam_intr"][:9], cam_fixed=p["cam_fixed"][:9], points=p["points"][:npts],
                     obs_cam=p["obs_cam"][m], obs_pt=p["obs_pt"][m], obs_xy=p["obs_xy"][m])
        assert np.all(p["obs_pt"][~m] >= npts)
        with _capi.Handle(_capi.Arrays.from_dict(first), loss="huber", f_scale=1.5, max_iter=150) as h:
            h.run()
            t0 = time.perf_counter()
            appended = h.append(_capi.Arrays.from_dict(p))
            t_app = (time.perf_counter() - t0) * 1e3
            s_app = h.run()
        t0 = time.perf_counter()
        with _capi.Handle(_capi.Arrays.from_dict(p), loss="huber", f_scale=1.5, max_iter=150) as h2:
            t_new = (time.perf_counter() - t0) * 1e3
            s_new = h2.run()
        out["fountain_incremental"] = {"observations": [int(m.sum()), len(m)], "appended_in_place": bool(appended),
                                       "append_ms": t_app, "new_handle_ms": t_new,
                                       "cost_after": [s_app["final_cost"], s_new["final_cost"]]}
    except Exception as e:
        out["fountain_incremental"] = {"error": repr(e)}
    # BaOptions.deterministic = 1 on the headline workload: the atomic-free build (owner-lane gathers, slab rows, fixed-order
    # segmented sum) and a whole LM step in that mode, beside the default (atomic) mode's numbers of the headline line
    try:
        pv = synthetic.make_config("venice1778")
        with _capi.Handle(_capi.Arrays.from_dict(pv), loss="huber", f_scale=1.0, max_iter=300, deterministic=1) as h:
            h.build_bench(1e-4, warmup=0, reps=3)
            ms_pass, ms_k = h.build_bench(1e-4, warmup=0, reps=20)
            h.run(1)
            h.synchronize()
            t0 = time.perf_counter()
            s = h.run(10)
            h.synchronize()
            dt = time.perf_counter() - t0
        steps = max(s["iterations"] - 1, 1)
        out["venice1778_deterministic"] = {"observations": len(pv["obs_cam"]), "build_ms": ms_pass, "k_build_det_ms": ms_k,
                                           "observations_per_sec": len(pv["obs_cam"]) / (ms_pass * 1e-3),
                                           "lm_iters_per_sec": steps / dt, "lm_pcg_iterations": s["pcg_iterations"]}
        del pv
    except Exception as e:
        out["venice1778_deterministic"] = {"error": repr(e)}
    p = synthetic.make_config("ladybug1723")
    with _capi.Handle(_capi.Arrays.from_dict(p), loss="huber", f_scale=1.0, max_iter=300) as h:
        h.build_bench(1e-4, warmup=0, reps=3)
        ms_pass, ms_k = h.build_bench(1e-4, warmup=0, reps=20)
        t0 = time.perf_counter()
        s = h.run(10)
        h.synchronize()
        dt = time.perf_counter() - t0
    out["ladybug1723"] = {"observations": len(p["obs_cam"]), "build_ms": ms_pass, "k_build_ms": ms_k,
                          "observations_per_sec": len(p["obs_cam"]) / (ms_pass * 1e-3),
                          "lm_iters_per_sec": max(s["iterations"], 1) / dt, "lm_pcg_iterations": s["pcg_iterations"]}
    # NOT banded: 1778 cameras on a closed ring (loop closure), long tracks.  Three ways: AUTO = block cyclic reduction
    # (direct, the ring of camera blocks halved per level); block-Jacobi PCG on the block-sparse S at the library's
    # default inner tolerance (relative residual 1e-8: an "exact" Newton step, capped at pcg_max_iter) and with an
    # inexact-Newton forcing term of 0.1, which loses nothing in LM progress on this problem
    p = synthetic.make_config("ring1778_loop")
    ring = {"cameras": len(p["cam_pose"]), "observations": len(p["obs_cam"]),
            "solver": "auto = block cyclic reduction (a closed ring is not banded); the pcg_* entries force block-Jacobi PCG"}
    for tag, kw in (("auto_cyclic_reduction", {}), ("pcg_tol_1e-8", dict(linear_solver="pcg", pcg_tol=1e-8)),
                    ("pcg_tol_0.1", dict(linear_solver="pcg", pcg_tol=0.1))):
        with _capi.Handle(_capi.Arrays.from_dict(p), loss="huber", f_scale=1.0, max_iter=300, **kw) as h:
            if "build_ms" not in ring:
                h.build_bench(1e-4, warmup=0, reps=3)
                ms_pass, ms_k = h.build_bench(1e-4, warmup=0, reps=10)
                ring["build_ms"] = ms_pass
                ring["observations_per_sec"] = len(p["obs_cam"]) / (ms_pass * 1e-3)
            t0 = time.perf_counter()
            s = h.run(20)
            h.synchronize()
            dt = time.perf_counter() - t0
        ring[tag] = {"lm_steps": s["iterations"], "lm_iters_per_sec": max(s["iterations"], 1) / dt,
                     "lm_pcg_iterations": s["pcg_iterations"],
                     "pcg_iterations_per_lm_step": s["pcg_iterations"] / max(s["iterations"], 1),
                     "cost": [s["initial_cost"], s["final_cost"]]}
    out["ring1778_loop"] = ring
    # The same ring with 1 % of its points seen a second time from far away (revisits): a ring PLUS a remainder of
    # far-apart camera pairs.  AUTO = CG on the whole S preconditioned by the cyclic reduction of its in-band part;
    # block-Jacobi PCG forced beside it
    p = synthetic.make_config("ring1778_revisit")
    rev = {"cameras": len(p["cam_pose"]), "observations": len(p["obs_cam"]),
           "solver": "auto = CG preconditioned by the cyclic reduction of the in-band part, default tolerance (adaptive forcing "
                     "term, BaOptions.pcg_tol = 0); auto_tol_1e-8: the same at a fixed 1e-8; pcg_tol_1e-8 forces block-Jacobi PCG"}
    for tag, kw in (("auto_bcr_preconditioned_cg", {}), ("auto_tol_1e-8", dict(pcg_tol=1e-8)), ("pcg_tol_1e-8", dict(linear_solver="pcg", pcg_tol=1e-8))):
        with _capi.Handle(_capi.Arrays.from_dict(p), loss="huber", f_scale=1.0, max_iter=300, **kw) as h:
            h.run(1)
            h.synchronize()
            t0 = time.perf_counter()
            s = h.run(20)
            h.synchronize()
            dt = time.perf_counter() - t0
        steps = max(s["iterations"] - 1, 1)
        rev[tag] = {"lm_steps": steps, "lm_iters_per_sec": steps / dt, "lm_pcg_iterations": s["pcg_iterations"],
                    "pcg_iterations_per_lm_step": s["pcg_iterations"] / max(s["iterations"], 1),
                    "cost": [s["initial_cost"], s["final_cost"]]}
    out["ring1778_revisit"] = rev
    # Object-centric capture: 1024 cameras around one object, every pair of cameras shares landmarks -- S is full, the
    # "genuine dense contraction" of BASELINE.json's north_star: AUTO = blocked dense Cholesky on the fp64 matrix cores
    # (ba_dense_big.h); block-Jacobi PCG forced beside it.  The flop rate charges the whole LM step minus the build pass
    # (solve + back-substitution + step tail) with the n^3 / 3 of the factorisation: a lower bound for the solver.
    p = synthetic.make_config("orbit1024")
    n_unk = 6 * int((np.asarray(p["cam_fixed"]) == 0).sum())
    orbit = {"cameras": len(p["cam_pose"]), "observations": len(p["obs_cam"]), "unknowns": n_unk,
             "solver": "auto = blocked dense Cholesky or block-Jacobi PCG at the default (adaptive) tolerance, whichever the handle times as "
                       "faster (full block pattern); dense_cholesky forces the blocked dense solver; pcg_tol_1e-8 forces block-Jacobi PCG at a fixed 1e-8"}
    for tag, kw in (("auto", {}), ("auto_dense_cholesky", dict(linear_solver="dense_cholesky")), ("pcg_tol_1e-8", dict(linear_solver="pcg", pcg_tol=1e-8))):
        with _capi.Handle(_capi.Arrays.from_dict(p), loss="huber", f_scale=1.0, max_iter=300, **kw) as h:
            if "build_ms" not in orbit:
                # (AUTO's build pass on this shape is the rows mode, csrc/ba_rows.h; the chunk kernel is timed beside it below)
                h.build_bench(1e-4, warmup=0, reps=3)
                orbit["build_ms"], orbit["build_kernels_ms"] = h.build_bench(1e-4, warmup=0, reps=10)
                orbit["build_obs_per_s"] = len(p["obs_cam"]) / orbit["build_ms"] * 1e3
                orbit["build_mode"] = "rows" if _capi.analyze(h.arr)["auto_rows_mode"] else "chunks"
            h.run(3 if tag == "auto" else 1)   # (AUTO times both solvers on its first steps, then keeps the faster one)
            h.synchronize()
            t0 = time.perf_counter()
            s = h.run(8)
            h.synchronize()
            dt = time.perf_counter() - t0
        steps = max(s["iterations"] - (3 if tag == "auto" else 1), 1)
        orbit[tag] = {"lm_steps": steps, "lm_iters_per_sec": steps / dt, "lm_pcg_iterations": s["pcg_iterations"],
                      "cost": [s["initial_cost"], s["final_cost"]]}
        if tag == "auto_dense_cholesky":
            solve_s = dt / steps - orbit["build_ms"] * 1e-3
            orbit[tag]["solve_ms_upper_bound"] = solve_s * 1e3
            orbit[tag]["fp64_mfma_tflops_lower_bound"] = n_unk ** 3 / 3.0 / solve_s / 1e12
            orbit[tag]["fp64_mfma_frac_of_78.6_lower_bound"] = n_unk ** 3 / 3.0 / solve_s / 78.6e12
    with _capi.Handle(_capi.Arrays.from_dict(p), loss="huber", f_scale=1.0, schur_mode="chunks") as h:
        h.build_bench(1e-4, warmup=0, reps=3)
        orbit["build_ms_chunk_kernel_forced"], _ = h.build_bench(1e-4, warmup=0, reps=10)
    out["orbit1024"] = orbit
    # Landmarks: a street of 1778 cameras (1 M observations) plus five points seen by EVERY camera -- tracks no chunk holds
    # ("heavy points": rows from k_lin_heavy, 7.9 M pairs through the global entry lists of csrc/ba_rows.h)
    nc, k = 1778, 5
    p = synthetic.make_street(nc, 200000, 1000000, seed=3)
    rng = np.random.default_rng(1)
    n0 = len(p["points"])
    lm = np.stack([0.5 * nc + rng.uniform(-5, 5, k), rng.uniform(-3, 3, k), rng.uniform(8, 12, k) * nc], axis=1)   # far ahead of the street
    q = dict(p)
    q["points"] = np.vstack([p["points"], lm])
    oc = np.tile(np.arange(nc, dtype=np.int32), k)
    op = np.repeat(np.arange(n0, n0 + k, dtype=np.int32), nc)
    xy = synthetic.project_points(p["cam_pose"], p["cam_intr"], q["points"], oc, op) + rng.normal(0, 0.5, (len(oc), 2))
    q["obs_cam"] = np.concatenate([p["obs_cam"], oc]); q["obs_pt"] = np.concatenate([p["obs_pt"], op]); q["obs_xy"] = np.vstack([p["obs_xy"], xy])
    lmk = {"cameras": nc, "landmarks": k, "landmark_pairs": k * nc * (nc + 1) // 2}
    for tag, prob in (("street", p), ("street_plus_landmarks", q)):
        with _capi.Handle(_capi.Arrays.from_dict(prob), loss="huber", f_scale=1.0) as h:
            h.build_bench(1e-4, warmup=0, reps=2)
            ms, _ = h.build_bench(1e-4, warmup=0, reps=5)
        lmk[tag] = {"observations": len(prob["obs_cam"]), "build_ms": ms}
    out["street1778_landmarks"] = lmk
    return out


def pmc_child(args):
    """What the PMC passes of live_traffic() profile: the workload, a handle, a few build passes.  No torch."""
    from structure_from_motion_amd import _capi, synthetic
    nc, n_pts, n_obs = synthetic.SHAPES[args.workload]
    n_pts, n_obs = max(int(n_pts * args.scale), 64), max(int(n_obs * args.scale), 256)
    idx = {"ladybug1723": 3, "venice1778": 4, "city10k": 5}[args.workload]
    prob = synthetic.make_street(nc, n_pts, n_obs, seed=1233 + idx)
    with _capi.Handle(_capi.Arrays.from_dict(prob), loss="huber", f_scale=1.0, max_iter=300) as h:
        h.build_bench(1e-4, warmup=1, reps=5)


def live_traffic(args):
    """HBM bytes per k_build launch, measured NOW: two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one
    pass) over a child process that runs this file with --pmc-child, counters read and corrected as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes (both in KiB per dispatch; gfx950 tallies the 128-byte requests of
    wide coalesced reads at 64 bytes: 2 x FETCH_SIZE, an upper bound for this kernel's narrow gathers; WRITE_SIZE is
    exact for float atomics).  Returns (bytes, detail) or (None, reason)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable,
                   os.path.abspath(__file__), "--pmc-child", "--workload", args.workload, "--scale", str(args.scale)]
            env = dict(os.environ, TMPDIR="/tmp")
            try:
                pr = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
                try:
                    pr.wait(timeout=90)
                except subprocess.TimeoutExpired:
                    os.killpg(pr.pid, signal.SIGKILL)      # the process group started here, nothing else
                    pr.wait()
                    return None, f"{counter} pass timed out"
            except OSError as e:
                return None, f"rocprofv3 did not start: {e}"
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if "k_build" in r["Kernel_Name"] and "heavy" not in r["Kernel_Name"] and r["Counter_Name"] == counter:
                            vals.append(float(r["Counter_Value"]))
            if not vals:
                return None, f"{counter} pass left no k_build rows (exit code {pr.returncode})"
            got[counter] = sum(vals) / len(vals) * 1024.0
    return 2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"], {
        "source": "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over this run's workload (separate passes)",
        "fetch_bytes_2x": 2.0 * got["FETCH_SIZE"], "write_bytes": got["WRITE_SIZE"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="venice1778", choices=["venice1778", "ladybug1723", "city10k"])
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = ONE problem cut N ways (default, BASELINE.json); weak = a full-size shard per rank")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink points/observations (debug only)")
    ap.add_argument("--lm-steps", type=int, default=10)
    ap.add_argument("--cpu-sample-obs", type=int, default=150000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="take roofline.traffic from profiles/traffic.json instead of measuring it")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)

    import torch
    import torch.distributed as dist
    from structure_from_motion_amd import _capi, synthetic
    from structure_from_motion_amd import dist as badist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    # BA_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share devices, the
    # collectives are staged through the host); never what the driver measures
    backend = os.environ.get("BA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    nc, n_pts, n_obs = synthetic.SHAPES[args.workload]
    n_pts, n_obs = max(int(n_pts * args.scale), 64), max(int(n_obs * args.scale), 256)
    idx = {"ladybug1723": 3, "venice1778": 4, "city10k": 5}[args.workload]
    strong = args.scaling == "strong"
    if not strong:
        # weak: rank r owns a full-size point shard of its own (own seed); cameras identical on every rank (cam_seed)
        prob = synthetic.make_street(nc, n_pts, n_obs, seed=1233 + idx + 1000 * rank, cam_seed=1233 + idx)
        total_obs, total_pts = n_obs * world, n_pts * world
        how = "every rank a full-size shard of its own"
    elif args.workload == "city10k":
        # ONE problem of n_pts points / n_obs observations in total, generated per shard (no rank ever
        # builds the whole 100 M-observation problem): rank r makes its 1/N of the points, cameras from one seed
        sp, so = n_pts // world, n_obs // world
        prob = synthetic.make_street(nc, sp, so, seed=1233 + idx + 1000 * rank, cam_seed=1233 + idx)
        total_obs, total_pts = so * world, sp * world
        how = "one problem generated per shard"
    else:
        # ONE problem, the same at every N (same seed on every rank), cut into N point shards balanced by observations
        full = synthetic.make_street(nc, n_pts, n_obs, seed=1233 + idx)
        prob = badist.shard_points(full, rank, world)[0] if world > 1 else full
        del full
        total_obs, total_pts = n_obs, n_pts
        how = "one problem cut by dist.shard_points"
    my_obs, my_pts = len(prob["obs_cam"]), len(prob["points"])
    arr = _capi.Arrays.from_dict(prob)
    lam = 1e-4
    opts = dict(loss="huber", f_scale=1.0, max_iter=300, device=local_rank)
    if world > 1:
        h = badist.create_sharded_handle(prob, rank, world, **opts)
    else:
        h = _capi.Handle(arr, **opts)

    def barrier():
        h.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    h.build_bench(lam, warmup=0, reps=max(args.warmup, 1))  # W untimed warm-up steps
    barrier()
    t0 = time.perf_counter()
    h._lib.ba_build_bench(h.h, lam, 0, args.steps, None, None)  # exactly K steps, no host syncs in between
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt * 1e3 / args.steps
    value = total_obs / (dt / args.steps)

    # dominant kernel: HIP events around its launch on the library's stream (per-rank shard)
    _, ms_kernel = h.build_bench(lam, warmup=0, reps=min(args.steps, 10))
    alg = algorithmic_bytes(my_obs, my_pts, nc)   # of this rank's launch
    achieved = alg / (ms_kernel * 1e-3) / 1e9
    traffic = None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf) and world == 1:   # (recorded for the whole problem on one GPU: says nothing about a shard's launch)
        traffic = json.load(open(tf)).get(f"{args.workload}@{args.scale:g}")
    roofline = {"bound": "hbm", "kernel": "k_build", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_detail": {"source": "profiles/traffic.json (recorded PMC passes)"} if traffic is not None else None,
                "algorithmic_bytes_per_launch": alg, "kernel_ms": ms_kernel}
    # the kernel sits near the ridge: the fp64 VALU ceiling next to the HBM one.  Algorithmic flops:
    # 430 per observation (projection, analytic blocks, U/V/W/g) + 108 + 108 (L + 1) per observation for the explicit
    # Schur complement at mean track length L + 50 per point (3x3 factor)
    track = my_obs / max(my_pts, 1)
    flops = my_obs * (430.0 + 108.0 + 108.0 * (track + 1.0)) + 50.0 * my_pts
    roofline["fp64_valu"] = {"achieved": flops / (ms_kernel * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": flops / (ms_kernel * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, "algorithmic_flops_per_launch": flops}

    # full LM iterations (build + PCG + back-substitution + trial cost + device-side accept logic)
    barrier()
    t0 = time.perf_counter()
    summ = h.run(args.lm_steps)
    barrier()
    lm_dt = time.perf_counter() - t0
    lm_rate = max(summ["iterations"], 1) / lm_dt

    out = {
        "metric": "observations/sec (residual+Jacobian+Schur build)", "value": value, "unit": "observations/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": args.scaling if world > 1 else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.workload} shape: {nc} cameras (replicated), {total_pts} points, {total_obs} observations in "
                               f"total over {world} GPU(s) ({how}); huber f_scale=1.0; lambda={lam:g}",
                   "scale": args.scale, "cameras": nc, "points_total": total_pts, "observations_total": total_obs,
                   "points_rank0": my_pts, "observations_rank0": my_obs,
                   "collectives": ("RCCL bound in libba_hip.so" if backend == "nccl" else backend) if world > 1 else None,
                   "parallelism": f"point-sharded x{world}" if world > 1 else "single GPU"},
        "lm_iters_per_sec": lm_rate, "lm_steps": summ["iterations"], "lm_pcg_iterations": summ["pcg_iterations"],
        "lm_cost": [summ["initial_cost"], summ["final_cost"]],
        "roofline": roofline,
    }
    h.close()
    if rank == 0 and world == 1 and not args.no_live_traffic:
        # (after the timed regions and with the handle released: the counter passes run in child processes)
        try:
            tb, detail = live_traffic(args)
        except Exception as e:  # never lose the headline line over a side measurement
            tb, detail = None, repr(e)
        if tb is not None:
            roofline["traffic"], roofline["traffic_detail"] = tb, detail
        elif roofline["traffic_detail"] is not None:
            roofline["traffic_detail"]["live_measurement_failed"] = detail
    if rank == 0 and world == 1 and not args.no_other_configs:
        out["other_configs"] = other_configs(_capi, synthetic)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(prob, args.cpu_sample_obs)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
