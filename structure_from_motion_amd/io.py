"""On-disk formats either side of the bundle-adjustment path.

* ASCII PLY as the reference writes it (``src/utils.py:334-353`` ``Toply``; header as in
  ``src/temple_new.ply:1-10``): ``x y z`` floats and ``red green blue`` uchar, colours given in 0..1.
* "Bundle Adjustment in the Large" (BAL) text problems, so that real Ladybug / Venice files can
  replace the synthetic shapes of ``synthetic.py`` when they are available.

BAL's camera model is ``p = -P / P.z`` with ``P = R X + t``, pixel ``= f (1 + k1 |p|^2 + k2 |p|^4) p``,
origin at the image centre.  The path's model is the reference's pinhole
(``u = fx x / z + cx``, no distortion, fixed intrinsics; ``BundleAdjustment.py:136-142``).  A BAL camera
maps onto it exactly for the pose part by a half turn about the optical axis, ``R' = D R``, ``t' = D t``,
``D = diag(-1, -1, 1)``, with ``fx = fy = f`` and ``cx = cy = 0``; because intrinsics are constants of the
path, radial distortion is removed from the OBSERVATIONS once, with the file's ``f, k1, k2``.
"""
import numpy as np

from .geometry import matrix_to_rvec, rvec_to_matrix

_D = np.diag([-1.0, -1.0, 1.0])


# ------------------------------------------------------------------------------------------ PLY
def write_ply(pts, clrs, filename="out.ply"):
    """Same file as ``utils.Toply(pts, clrs, filename)`` (``utils.py:334-353``)."""
    pts = np.asarray(pts)
    clrs = np.clip(np.asarray(clrs) * 255.0, 0, 255).astype(np.uint8)
    with open(filename, "w") as f:
        f.write("ply\nformat ascii 1.0\n")
        f.write("element vertex {}\n".format(pts.shape[0]))
        f.write("property float x\nproperty float y\nproperty float z\n")
        f.write("property uchar red\nproperty uchar green\nproperty uchar blue\n")
        f.write("end_header\n")
        f.write("".join("{} {} {} {} {} {}\n".format(p[0], p[1], p[2], c[0], c[1], c[2]) for p, c in zip(pts, clrs)))


def read_ply(filename):
    """ASCII PLY with ``x y z [red green blue]`` vertices -> (pts float64 (n,3), colours in 0..1 (n,3) or None)."""
    with open(filename, "r") as f:
        if f.readline().strip() != "ply":
            raise ValueError(f"{filename}: not a PLY file")
        n, props, fmt = 0, [], None
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                props.append(tok[-1])
            elif tok[0] == "end_header":
                break
        if fmt != "ascii":
            raise ValueError(f"{filename}: only ascii PLY is supported (got {fmt})")
        data = np.loadtxt(f, max_rows=n, ndmin=2) if n else np.zeros((0, len(props)))
    col = {name: i for i, name in enumerate(props)}
    pts = data[:, [col["x"], col["y"], col["z"]]].astype(np.float64)
    clrs = None
    if all(k in col for k in ("red", "green", "blue")):
        clrs = data[:, [col["red"], col["green"], col["blue"]]] / 255.0
    return pts, clrs


def points_list_to_ply(all_points_3D, filename="out.ply"):
    """The reference's ``[x, y, z, r, g, b]`` rows (``Reconstruction.py:43-47``) straight to PLY."""
    a = np.asarray(all_points_3D, dtype=np.float64).reshape(-1, 6)
    write_ply(a[:, :3], a[:, 3:], filename)


# ------------------------------------------------------------------------------------------ BAL
def _undistort(xy, f, k1, k2, iters=20):
    """observed = f r(|p|^2) p  ->  f p  (fixed-point iteration on the radial factor)."""
    pd = xy / f[:, None]
    p = pd.copy()
    for _ in range(iters):
        r2 = (p * p).sum(1)
        p = pd / (1.0 + k1 * r2 + k2 * r2 * r2)[:, None]
    return p * f[:, None]


def _distort(xy, f, k1, k2):
    p = xy / f[:, None]
    r2 = (p * p).sum(1)
    return xy * (1.0 + k1 * r2 + k2 * r2 * r2)[:, None]


def read_bal(path, n_fixed=None):
    """BAL text file -> (problem dict in the ``BaProblem`` layout, extras).

    ``extras = dict(f, k1, k2)`` per camera (needed to write the file back).  ``cam_fixed`` follows the
    reference's gauge rule (``BundleAdjustment.py:34-38``): first camera, and the second while there are
    fewer than 15 cameras; ``n_fixed`` overrides the count."""
    with open(path, "r") as fh:
        tok = fh.read().split()
    nc, npt, no = int(tok[0]), int(tok[1]), int(tok[2])
    need = 3 + 4 * no + 9 * nc + 3 * npt
    if len(tok) < need:
        raise ValueError(f"{path}: truncated BAL file ({len(tok)} tokens, {need} expected)")
    o = np.array(tok[3:3 + 4 * no], dtype=np.float64).reshape(no, 4)
    cams = np.array(tok[3 + 4 * no:3 + 4 * no + 9 * nc], dtype=np.float64).reshape(nc, 9)
    pts = np.array(tok[3 + 4 * no + 9 * nc:need], dtype=np.float64).reshape(npt, 3)
    obs_cam, obs_pt = o[:, 0].astype(np.int32), o[:, 1].astype(np.int32)
    if no and (obs_cam.min() < 0 or obs_cam.max() >= nc or obs_pt.min() < 0 or obs_pt.max() >= npt):
        raise ValueError(f"{path}: observation index out of range")
    f, k1, k2 = cams[:, 6].copy(), cams[:, 7].copy(), cams[:, 8].copy()
    R = _D @ rvec_to_matrix(cams[:, :3])
    cam_pose = np.concatenate([np.array([matrix_to_rvec(Ri) for Ri in R]).reshape(nc, 3), cams[:, 3:6] @ _D.T], axis=1)
    cam_intr = np.stack([f, f, np.zeros(nc), np.zeros(nc)], axis=1)
    obs_xy = _undistort(o[:, 2:4], f[obs_cam], k1[obs_cam], k2[obs_cam])
    cam_fixed = np.zeros(nc, dtype=np.uint8)
    nfix = n_fixed if n_fixed is not None else (2 if nc < 15 else 1)
    cam_fixed[:min(nfix, nc)] = 1
    prob = dict(cam_pose=np.ascontiguousarray(cam_pose), cam_intr=cam_intr, cam_fixed=cam_fixed, points=pts,
                obs_cam=obs_cam, obs_pt=obs_pt, obs_xy=np.ascontiguousarray(obs_xy))
    return prob, dict(f=f, k1=k1, k2=k2)


def write_bal(path, prob, extras=None):
    """Inverse of :func:`read_bal` (``fx`` is written as BAL's ``f``; ``cx, cy`` must be 0)."""
    nc, npt, no = len(prob["cam_pose"]), len(prob["points"]), len(prob["obs_cam"])
    intr = np.asarray(prob["cam_intr"], dtype=np.float64)
    if np.any(intr[:, 2:] != 0.0) or np.any(intr[:, 0] != intr[:, 1]):
        raise ValueError("BAL cameras have one focal length and the principal point at the origin")
    f = intr[:, 0]
    k1 = np.zeros(nc) if extras is None else np.asarray(extras["k1"], dtype=np.float64)
    k2 = np.zeros(nc) if extras is None else np.asarray(extras["k2"], dtype=np.float64)
    pose = np.asarray(prob["cam_pose"], dtype=np.float64)
    R = _D @ rvec_to_matrix(pose[:, :3])
    rv, t = np.array([matrix_to_rvec(Ri) for Ri in R]).reshape(nc, 3), pose[:, 3:] @ _D.T
    oc = np.asarray(prob["obs_cam"])
    xy = _distort(np.asarray(prob["obs_xy"], dtype=np.float64), f[oc], k1[oc], k2[oc])
    with open(path, "w") as fh:
        fh.write(f"{nc} {npt} {no}\n")
        fh.write("".join(f"{int(c)} {int(p)}     {x!r} {y!r}\n"
                         for c, p, (x, y) in zip(oc, prob["obs_pt"], xy.tolist())))
        cam9 = np.concatenate([rv, t, f[:, None], k1[:, None], k2[:, None]], axis=1)
        fh.write("".join(f"{v!r}\n" for v in cam9.reshape(-1).tolist()))
        fh.write("".join(f"{v!r}\n" for v in np.asarray(prob["points"], dtype=np.float64).reshape(-1).tolist()))
