"""Seeded synthetic bundle-adjustment problems.

The reference ships no datasets (``/root/reference/.gitignore:1``), so every
benchmark / parity problem is synthetic: the *shape* (camera / point /
observation counts, track-length statistics, intrinsics) follows the configs of
BASELINE.json; the camera model is the reference's (rvec, t, fixed per-camera
fx, fy, cx, cy, no distortion -- ``BundleAdjustment.py:136-142``).

All generators return a dict of structure-of-arrays (the layout of the C-ABI
``BaProblem``): cam_pose (Nc,6), cam_intr (Nc,4), cam_fixed (Nc,) uint8,
points (Np,3), obs_cam (No,) int32, obs_pt (No,) int32, obs_xy (No,2).
Observation order is camera-major, as ``prepare_observations_from_image_data``
produces it (``BundleAdjustment.py:249-270``).
"""
import numpy as np

from .geometry import matrix_to_rvec, project_points, rvec_to_matrix

# (n_cams, n_points, n_obs) of the BASELINE.json configs
SHAPES = {
    "ladybug1723": (1723, 156502, 678718),
    "venice1778": (1778, 993923, 5001946),
    "city10k": (10000, 20000000, 100000000),
}


def _track_lengths(rng, n_points, n_obs, max_track, min_track=2):
    mean = n_obs / n_points
    L = min_track + rng.poisson(max(mean - min_track, 0.0), size=n_points)
    L = np.clip(L, min_track, max_track).astype(np.int64)
    diff = int(n_obs - L.sum())
    # hit n_obs exactly: nudge random points by +-1 inside [min_track, max_track]
    while diff != 0:
        step = 1 if diff > 0 else -1
        ok = np.flatnonzero((L + step >= min_track) & (L + step <= max_track))
        k = min(abs(diff), len(ok))
        if k == 0:
            break
        pick = rng.choice(ok, size=k, replace=False)
        L[pick] += step
        diff -= step * k
    return L


def _finish(rng, cam_pose_gt, cam_intr, points_gt, obs_cam, obs_pt, noise, outlier_frac,
            sig_r, sig_t, sig_p, cam_fixed, shuffle_in_cam=True):
    n_obs = len(obs_cam)
    xy = project_points(cam_pose_gt, cam_intr, points_gt, obs_cam, obs_pt)
    xy += rng.normal(0.0, noise, size=xy.shape)
    if outlier_frac > 0:
        bad = rng.random(n_obs) < outlier_frac
        nb = int(bad.sum())
        ang = rng.uniform(0, 2 * np.pi, nb)
        mag = rng.uniform(5.0, 50.0, nb)
        xy[bad] += np.stack([mag * np.cos(ang), mag * np.sin(ang)], axis=1)
    # camera-major observation order (then arbitrary "keypoint" order inside a camera)
    key = obs_cam.astype(np.int64) * (1 << 32) + (rng.permutation(n_obs) if shuffle_in_cam else np.arange(n_obs))
    order = np.argsort(key, kind="stable")
    obs_cam, obs_pt, xy = obs_cam[order], obs_pt[order], xy[order]
    cam_pose = cam_pose_gt.copy()
    free = ~cam_fixed.astype(bool)
    cam_pose[free, :3] += rng.normal(0.0, sig_r, size=(int(free.sum()), 3))
    cam_pose[free, 3:] += rng.normal(0.0, sig_t, size=(int(free.sum()), 3))
    points = points_gt + rng.normal(0.0, 1.0, size=points_gt.shape) * sig_p
    return dict(cam_pose=np.ascontiguousarray(cam_pose), cam_intr=np.ascontiguousarray(cam_intr),
                cam_fixed=cam_fixed.astype(np.uint8), points=np.ascontiguousarray(points),
                obs_cam=np.ascontiguousarray(obs_cam, dtype=np.int32),
                obs_pt=np.ascontiguousarray(obs_pt, dtype=np.int32),
                obs_xy=np.ascontiguousarray(xy),
                cam_pose_gt=cam_pose_gt, points_gt=points_gt)


def _revisits(rng, n_cams, obs_cam, obs_pt, first_cam, frac, length, ring):
    """A fraction of the points is seen a SECOND time by ``length`` consecutive cameras far away along the trajectory (a
    revisit / loop closure in the middle of a sequence, landmarks seen again from another street): the co-visibility
    pattern is then a chain (or ring) plus a remainder of far-apart camera pairs."""
    n_points = len(first_cam)
    pick = np.flatnonzero(rng.random(n_points) < frac)
    lo, hi = max(n_cams // 8, 2 * length + 44), max(n_cams // 2, 2 * length + 45)
    off = rng.integers(lo, hi, size=len(pick)) * rng.choice([-1, 1], size=len(pick))
    start = first_cam[pick] + off
    start = start % n_cams if ring else np.clip(start, 0, n_cams - length)
    extra_pt = np.repeat(pick, length)
    extra_cam = np.repeat(start, length) + np.tile(np.arange(length), len(pick))
    if ring:
        extra_cam %= n_cams
    key = np.unique(np.concatenate([obs_pt * n_cams + obs_cam, extra_pt * n_cams + extra_cam]))   # (no duplicate observation)
    return key % n_cams, key // n_cams


def make_street(n_cams, n_points, n_obs, seed=0, noise=0.5, outlier_frac=0.01, max_track=16,
                n_fixed=2, sig_r=1e-3, sig_t=1e-2, sig_p_rel=1e-2, point_offset=0, cam_seed=None,
                revisit_frac=0.0, revisit_len=3):
    """BAL-shaped problem: cameras on a street trajectory (baseline 1), points in a slab at
    depth 5..50 baselines, each observed by a contiguous window of cameras whose length follows
    2 + Poisson so that the mean track is n_obs / n_points.  ``cam_seed`` fixes the cameras
    independently of the points (used to generate point shards of one problem per rank).
    The first ``n_fixed`` cameras are gauge anchors; the default 2 removes the scale freedom as well
    (what the reference does below 15 images, ``BundleAdjustment.py:38``) so the problem has an
    isolated minimum."""
    rng = np.random.default_rng(seed)
    crng = np.random.default_rng(seed if cam_seed is None else cam_seed)
    max_track = min(max_track, n_cams)
    # cameras: centre (i, ~0, ~0), looking down +z with a small random rotation
    C = np.stack([np.arange(n_cams, dtype=np.float64),
                  crng.normal(0, 0.05, n_cams), crng.normal(0, 0.05, n_cams)], axis=1)
    rv = crng.normal(0.0, 0.02, size=(n_cams, 3))
    R = rvec_to_matrix(rv)
    t = -np.einsum("nij,nj->ni", R, C)
    cam_pose_gt = np.concatenate([rv, t], axis=1)
    f = crng.uniform(800.0, 1600.0, n_cams)
    cam_intr = np.stack([f, f, np.full(n_cams, 960.0), np.full(n_cams, 540.0)], axis=1)
    L = _track_lengths(rng, n_points, n_obs, max_track)
    start = (rng.random(n_points) * (n_cams - L + 1)).astype(np.int64)
    centre = start + 0.5 * (L - 1)
    depth = rng.uniform(5.0, 50.0, n_points)
    points_gt = np.stack([centre + rng.uniform(-1.0, 1.0, n_points),
                          rng.uniform(-3.0, 3.0, n_points), depth], axis=1)
    obs_pt = np.repeat(np.arange(n_points, dtype=np.int64), L)
    first = np.cumsum(L) - L
    within = np.arange(len(obs_pt), dtype=np.int64) - np.repeat(first, L)
    obs_cam = np.repeat(start, L) + within
    if revisit_frac > 0:
        obs_cam, obs_pt = _revisits(rng, n_cams, obs_cam, obs_pt, start, revisit_frac, revisit_len, ring=False)
    cam_fixed = np.zeros(n_cams, dtype=np.uint8)
    cam_fixed[:n_fixed] = 1
    prob = _finish(rng, cam_pose_gt, cam_intr, points_gt, obs_cam, obs_pt, noise, outlier_frac,
                   sig_r, sig_t, sig_p_rel * depth[:, None], cam_fixed)
    prob["name"] = f"street(nc={n_cams},np={n_points},no={len(obs_pt)},seed={seed})"
    return prob


def make_ring(n_cams, n_points, seed=0, K=(1520.4, 1525.9, 302.32, 246.87), focal_jitter=0.02,
              noise=0.35, outlier_frac=0.0, visibility="dense", mean_track=4.0, n_fixed=1,
              n_seed_points=None, radius=4.0, sig_r=1e-3, sig_t=1e-2, sig_p=1e-2, n_unobserved=0,
              max_track=24, n_landmarks=0, point_radius=None, revisit_frac=0.0, revisit_len=3):
    """Temple / Fountain-shaped problem: cameras on a ring (or arc) of ``radius`` looking at a
    unit blob of points.

    visibility='dense': each point is seen by a contiguous window of ring cameras (mean length
    ``mean_track``).  visibility='sfm': what the reference's driver actually produces
    (SURVEY.md H4): the first ``n_seed_points`` points have 2 observations (cameras 0 and 1),
    every other point exactly 1 (``Reconstruction.py:41-47, 244-251``).
    ``n_landmarks`` (dense only): the first points are seen by EVERY camera (tracks of ``n_cams``)."""
    rng = np.random.default_rng(seed)
    ang = np.linspace(0.0, 2 * np.pi, n_cams, endpoint=False) if n_cams > 12 else np.linspace(-0.9, 0.9, n_cams)
    C = np.stack([radius * np.sin(ang), rng.normal(0, 0.05, n_cams), -radius * np.cos(ang)], axis=1)
    cam_pose_gt = np.zeros((n_cams, 6))
    for i in range(n_cams):
        zax = -C[i] / np.linalg.norm(C[i])
        xax = np.cross([0.0, 1.0, 0.0], zax); xax /= np.linalg.norm(xax)
        yax = np.cross(zax, xax)
        Rm = np.stack([xax, yax, zax], axis=0)  # world -> camera
        cam_pose_gt[i, :3] = matrix_to_rvec(Rm)
        cam_pose_gt[i, 3:] = -Rm @ C[i]
    if point_radius is not None:
        # large rings: points in a band around the camera circle's inside, so that a camera sees the points near its
        # own azimuth from a sane distance (a unit blob in the middle of a 1778-camera ring would be seen by everybody)
        azp = rng.uniform(0.0, 2 * np.pi, n_points)
        rad = radius - rng.uniform(*point_radius, n_points)
        ring_points = np.stack([rad * np.sin(azp), rng.normal(0, 0.3, n_points), -rad * np.cos(azp)], axis=1)
    fj = 1.0 + rng.uniform(-focal_jitter, focal_jitter, n_cams)
    cam_intr = np.stack([K[0] * fj, K[1] * fj, np.full(n_cams, K[2]), np.full(n_cams, K[3])], axis=1)
    n_obs_pts = n_points - n_unobserved
    points_gt = rng.normal(0.0, 0.45, size=(n_points, 3)) if point_radius is None else ring_points
    if visibility == "dense":
        n_obs = int(round(mean_track * n_obs_pts))
        L = _track_lengths(rng, n_obs_pts, n_obs, max_track=min(n_cams, max_track))
        L[:n_landmarks] = n_cams
        # window centred on the camera nearest to the point's azimuth
        az = np.arctan2(points_gt[:n_obs_pts, 0], -points_gt[:n_obs_pts, 2])
        if n_cams > 64:   # (cameras are evenly spaced on the full circle: nearest by rounding, no Np x Nc table)
            near = np.rint((az % (2 * np.pi)) / (2 * np.pi) * n_cams).astype(np.int64) % n_cams
        else:
            near = np.argmin(np.abs(((az[:, None] - ang[None, :] + np.pi) % (2 * np.pi)) - np.pi), axis=1)
        start = near - L // 2
        obs_pt = np.repeat(np.arange(n_obs_pts, dtype=np.int64), L)
        first = np.cumsum(L) - L
        within = np.arange(len(obs_pt), dtype=np.int64) - np.repeat(first, L)
        if n_cams > 12:
            obs_cam = (np.repeat(start, L) + within) % n_cams
        else:
            start = np.clip(start, 0, n_cams - L)
            obs_cam = np.repeat(start, L) + within
        if revisit_frac > 0:
            obs_cam, obs_pt = _revisits(rng, n_cams, obs_cam, obs_pt, start % n_cams, revisit_frac, revisit_len, ring=n_cams > 12)
    elif visibility == "random":
        # object-centric capture: every point is seen by a random set of cameras from all around the ring (mean length
        # ``mean_track``) -- any two cameras share landmarks, the reduced camera system is (nearly) full
        n_obs = int(round(mean_track * n_obs_pts))
        L = _track_lengths(rng, n_obs_pts, n_obs, max_track=min(n_cams, max_track))
        obs_pt = np.repeat(np.arange(n_obs_pts, dtype=np.int64), L)
        obs_cam = rng.integers(0, n_cams, size=len(obs_pt))
        key = np.unique(obs_pt * n_cams + obs_cam)        # (a camera drawn twice for one point counts once)
        obs_pt, obs_cam = key // n_cams, key % n_cams
        cnt = np.bincount(obs_pt, minlength=n_obs_pts)
        lone = np.flatnonzero(cnt < 2)                      # keep every track at two observations or more
        if len(lone):
            first_cam = np.zeros(n_obs_pts, np.int64); first_cam[obs_pt[::-1]] = obs_cam[::-1]
            extra_cam = np.where(cnt[lone] == 0, 0, (first_cam[lone] + 1) % n_cams)
            obs_pt = np.concatenate([obs_pt, lone]); obs_cam = np.concatenate([obs_cam, extra_cam])
            zero = lone[cnt[lone] == 0]
            obs_pt = np.concatenate([obs_pt, zero]); obs_cam = np.concatenate([obs_cam, np.ones(len(zero), np.int64)])
    elif visibility == "sfm":
        ns = n_obs_pts // 8 if n_seed_points is None else n_seed_points
        pt_a = np.arange(ns, dtype=np.int64)
        rest = np.arange(ns, n_obs_pts, dtype=np.int64)
        cam_rest = 2 + (np.arange(len(rest)) * (n_cams - 2) // max(len(rest), 1))
        obs_pt = np.concatenate([pt_a, pt_a, rest])
        obs_cam = np.concatenate([np.zeros(ns, np.int64), np.ones(ns, np.int64), cam_rest.astype(np.int64)])
    else:
        raise ValueError(visibility)
    cam_fixed = np.zeros(n_cams, dtype=np.uint8)
    cam_fixed[:n_fixed] = 1
    prob = _finish(rng, cam_pose_gt, cam_intr, points_gt, obs_cam, obs_pt, noise, outlier_frac,
                   sig_r, sig_t, sig_p, cam_fixed)
    prob["name"] = f"ring(nc={n_cams},np={n_points},no={len(obs_pt)},vis={visibility},seed={seed})"
    return prob


def make_config(name, seed=None, scale=1.0):
    """The BASELINE.json configs as concrete problems.  ``scale`` < 1 shrinks points and
    observations proportionally (parity tests use reduced sizes; the bench uses 1.0)."""
    if name in ("fountain", "C1"):
        return make_ring(11, int(16908 * scale), seed=1235 if seed is None else seed,
                         K=(2759.48 * 1600 / 3072, 2764.16 * 1600 / 3072, 1520.69 * 1600 / 3072, 1006.81 * 1600 / 3072),
                         focal_jitter=0.0, noise=0.4, mean_track=3.5, n_fixed=2)
    if name == "fountain_sfm":
        # the visibility Reconstruction.py really records: the seed pair's points are seen twice
        # (cameras 0 and 1, both anchored below 15 images), every later point once, by the image that triangulated it
        return make_ring(11, int(16908 * scale), seed=1237 if seed is None else seed,
                         K=(2759.48 * 1600 / 3072, 2764.16 * 1600 / 3072, 1520.69 * 1600 / 3072, 1006.81 * 1600 / 3072),
                         focal_jitter=0.0, noise=0.4, visibility="sfm", n_seed_points=int(2500 * scale), n_fixed=2)
    if name == "temple_sfm":
        return make_ring(47, int(20467 * scale), seed=1238 if seed is None else seed, focal_jitter=0.02, noise=0.35,
                         visibility="sfm", n_seed_points=int(1500 * scale), n_fixed=1)
    if name in ("temple", "C2"):
        return make_ring(47, int(20467 * scale), seed=1236 if seed is None else seed, focal_jitter=0.02,
                         noise=0.35, mean_track=4.0, n_fixed=1)
    if name == "ring1778_loop":
        # NOT banded: the Ladybug-sized camera count on a closed ring (camera 1777 sees what camera 0 sees: loop closure)
        # with tracks of up to 40 cameras -- not a chain in any camera order; AUTO solves it by cyclic reduction on the ring
        # (bench.py reports LM iterations/s for that and for the block-Jacobi PCG forced on the same problem)
        return make_ring(1778, int(150000 * scale), seed=1239 if seed is None else seed, K=(1200.0, 1200.0, 960.0, 540.0),
                         focal_jitter=0.05, noise=0.5, outlier_frac=0.01, mean_track=8.0, max_track=40, n_fixed=2,
                         radius=283.0, point_radius=(4.0, 30.0), sig_t=2e-2, sig_p=5e-2)
    if name == "ring1778_revisit":
        # the closed ring above with 1 % of its points seen a second time from far away along the ring (revisits): a ring
        # PLUS a remainder of far-apart camera pairs -- no direct solver of the engine applies; AUTO = CG preconditioned by
        # the cyclic reduction of the in-band part
        return make_ring(1778, int(150000 * scale), seed=1239 if seed is None else seed, K=(1200.0, 1200.0, 960.0, 540.0),
                         focal_jitter=0.05, noise=0.5, outlier_frac=0.01, mean_track=8.0, max_track=40, n_fixed=2,
                         radius=283.0, point_radius=(4.0, 30.0), sig_t=2e-2, sig_p=5e-2, revisit_frac=0.01, revisit_len=3)
    if name == "orbit1024":
        # object-centric: 1024 cameras around one object, every point seen from all around -- S is full (6132 unknowns):
        # AUTO's blocked dense Cholesky (ba_dense_big.h), the "genuine dense contraction" of BASELINE.json's north_star
        return make_ring(1024, int(120000 * scale), seed=1240 if seed is None else seed, K=(1200.0, 1200.0, 960.0, 540.0),
                         focal_jitter=0.05, noise=0.5, outlier_frac=0.01, visibility="random", mean_track=8.0, max_track=24,
                         n_fixed=2, radius=6.0)
    if name in SHAPES:
        nc, np_, no = SHAPES[name]
        idx = {"ladybug1723": 3, "venice1778": 4, "city10k": 5}[name]
        return make_street(nc, max(int(np_ * scale), 16), max(int(no * scale), 32),
                           seed=1233 + idx if seed is None else seed)
    raise ValueError(f"unknown config {name!r}")


class KeyPoint:
    """Stand-in for ``cv2.KeyPoint``: the path only reads ``.pt`` (``BundleAdjustment.py:269``)."""
    __slots__ = ("pt",)

    def __init__(self, x, y):
        self.pt = (float(x), float(y))


def to_image_data(prob, ids=None):
    """A SoA problem as the containers the reference's driver holds (``Reconstruction.py:59-60``):
    ``image_data[id] = (R, t (3,1), K, ref list (-1 = none), descriptors, keypoints)`` and ``[[x, y, z, r, g, b], ...]``.
    Returns (visited_ids, image_data, all_points_3D)."""
    n_c = len(prob["cam_pose"])
    ids = list(range(n_c)) if ids is None else list(ids)
    R = rvec_to_matrix(prob["cam_pose"][:, :3])
    order = np.argsort(prob["obs_cam"], kind="stable")
    oc, op, oxy = prob["obs_cam"][order], prob["obs_pt"][order], prob["obs_xy"][order]
    start = np.searchsorted(oc, np.arange(n_c + 1))
    image_data = {}
    for i, cid in enumerate(ids):
        fx, fy, cx, cy = prob["cam_intr"][i]
        K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
        a, b = start[i], start[i + 1]
        kp = [KeyPoint(x, y) for x, y in oxy[a:b]]
        ref = [int(v) for v in op[a:b]]
        image_data[cid] = (R[i].copy(), prob["cam_pose"][i, 3:].reshape(3, 1).copy(), K, ref,
                           np.zeros((len(kp), 1), np.float32), kp)
    pts = [[float(x), float(y), float(z), 0.5, 0.5, 0.5] for x, y, z in prob["points"]]
    return ids, image_data, pts
