"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the bundle-adjustment objective.

CPU oracle for the HIP engine (see ``oracle/__init__.py`` for who may import
this).  Every function cites the reference lines it restates; ``scipy:`` means
scipy 1.15.3 as installed in the build image
(``/usr/local/lib/python3.10/dist-packages/scipy``), the third-party solver the
reference delegates to.

Problem layout (structure of arrays, float64 / int32) -- identical to the C-ABI
``BaProblem`` of ``include/ba_hip.h``:

  cam_pose  (Nc, 6)  rvec(3), t(3)           BundleAdjustment.py:57-59, 116-119
  cam_intr  (Nc, 4)  fx, fy, cx, cy          what cv2.projectPoints reads of K (:136, :142)
  cam_fixed (Nc,)    gauge-anchored cameras  BundleAdjustment.py:34-38, :53
  points    (Np, 3)                          BundleAdjustment.py:62-63
  obs_cam   (No,)    camera *index* (position in ``list(cameras.keys())``, :30)
  obs_pt    (No,)    point index             BundleAdjustment.py:131
  obs_xy    (No, 2)  observed pixel          BundleAdjustment.py:132

The parameter vector of the reference is ``[rvec,t] x free cameras`` (camera
order) then ``xyz x all points`` (BundleAdjustment.py:46-65).

Parity status: the OpenCV arithmetic (``rodrigues`` / ``project``) is **parity
unpinned** (no OpenCV in the image, no reference fixture); the objective,
Jacobian and solver results are pinned by ``tests/golden/*.npz`` generated from
the reference's own function (``tests/golden/make_golden.py``).
"""
import numpy as np

DBL_EPS = float(np.finfo(np.float64).eps)

LOSSES = ("linear", "soft_l1", "huber", "cauchy", "arctan")  # scipy:_lsq/least_squares.py:208-209


# --------------------------------------------------------------------------
# OpenCV arithmetic (vectorised)                      -- parity unpinned
# --------------------------------------------------------------------------
def skew(v):
    v = np.asarray(v, dtype=np.float64)
    out = np.zeros(v.shape[:-1] + (3, 3))
    out[..., 0, 1] = -v[..., 2]
    out[..., 0, 2] = v[..., 1]
    out[..., 1, 0] = v[..., 2]
    out[..., 1, 2] = -v[..., 0]
    out[..., 2, 0] = -v[..., 1]
    out[..., 2, 1] = v[..., 0]
    return out


def rodrigues(rvec):
    """rvec (...,3) -> R (...,3,3); cv2.Rodrigues vector->matrix (BundleAdjustment.py:82;
    implicit inside cv2.projectPoints at :142).  theta < DBL_EPSILON -> identity."""
    r = np.asarray(rvec, dtype=np.float64)
    theta = np.sqrt((r * r).sum(-1))
    small = theta < DBL_EPS
    th = np.where(small, 1.0, theta)
    k = r / th[..., None]
    c = np.cos(th)[..., None, None]
    s = np.sin(th)[..., None, None]
    eye = np.broadcast_to(np.eye(3), r.shape[:-1] + (3, 3))
    R = c * eye + (1.0 - c) * (k[..., :, None] * k[..., None, :]) + s * skew(k)
    return np.where(small[..., None, None], eye, R)


def so3_left_jacobian(rvec):
    """J_l(r) with d(R(r) X)/dr = -[R X]x J_l(r)  (derivative w.r.t. an *additive*
    axis-angle update -- what scipy's ``x + step`` does, scipy:_lsq/trf.py:497)."""
    r = np.asarray(rvec, dtype=np.float64)
    s = (r * r).sum(-1)
    th = np.sqrt(s)
    small = s < 1e-2
    ths = np.where(small, 1.0, th)
    b_cl = (1.0 - np.cos(ths)) / (ths * ths)
    c_cl = (ths - np.sin(ths)) / (ths * ths * ths)
    b_se = 0.5 - s / 24.0 + s * s / 720.0 - s ** 3 / 40320.0 + s ** 4 / 3628800.0 - s ** 5 / 479001600.0
    c_se = 1.0 / 6.0 - s / 120.0 + s * s / 5040.0 - s ** 3 / 362880.0 + s ** 4 / 39916800.0 - s ** 5 / 6227020800.0
    b = np.where(small, b_se, b_cl)[..., None, None]
    c = np.where(small, c_se, c_cl)[..., None, None]
    K = skew(r)
    eye = np.broadcast_to(np.eye(3), r.shape[:-1] + (3, 3))
    return eye + b * K + c * (K @ K)


def project(cam_pose, cam_intr, points, obs_cam, obs_pt):
    """Pinhole projection of every observation; cv2.projectPoints with
    distCoeffs=None (BundleAdjustment.py:142): z' = 1/Z, or 1 when Z == 0."""
    R = rodrigues(cam_pose[:, :3])[obs_cam]
    t = cam_pose[obs_cam, 3:6]
    X = points[obs_pt]
    Xc = np.einsum("nij,nj->ni", R, X) + t
    z = Xc[:, 2]
    iz = np.where(z != 0, 1.0 / np.where(z != 0, z, 1.0), 1.0)
    K = cam_intr[obs_cam]
    u = K[:, 0] * (Xc[:, 0] * iz) + K[:, 2]
    v = K[:, 1] * (Xc[:, 1] * iz) + K[:, 3]
    return np.stack([u, v], axis=1)


def residuals(cam_pose, cam_intr, points, obs_cam, obs_pt, obs_xy):
    """f(x): interleaved (du, dv) in observation order (BundleAdjustment.py:129-152)."""
    return (project(cam_pose, cam_intr, points, obs_cam, obs_pt) - obs_xy).reshape(-1)


# --------------------------------------------------------------------------
# Robust loss, per scalar residual component  (scipy:_lsq/least_squares.py:169-227)
# --------------------------------------------------------------------------
def loss_rho(z, kind):
    """rho(z), rho'(z) for z = (f/f_scale)^2 (scipy:_lsq/least_squares.py:169-205)."""
    z = np.asarray(z, dtype=np.float64)
    if kind == "linear":
        return z.copy(), np.ones_like(z)
    if kind == "huber":
        m = z <= 1
        zs = np.where(m, 1.0, z)
        return np.where(m, z, 2 * np.sqrt(zs) - 1), np.where(m, 1.0, zs ** -0.5)
    if kind == "soft_l1":
        t = 1 + z
        return 2 * (np.sqrt(t) - 1), t ** -0.5
    if kind == "cauchy":
        return np.log1p(z), 1 / (1 + z)
    if kind == "arctan":
        return np.arctan(z), 1 / (1 + z * z)
    raise ValueError(kind)


def cost(f, f_scale=1.0, kind="huber"):
    """0.5 * sum_i fs^2 rho((f_i/fs)^2) over every scalar component
    (scipy:_lsq/least_squares.py:220-227 with cost_only=True)."""
    rho, _ = loss_rho((f / f_scale) ** 2, kind)
    return 0.5 * f_scale * f_scale * float(rho.sum())


def loss_curvature(z, kind):
    """rho'(z) + 2 z rho''(z): second derivative of the robustified cost w.r.t. the residual
    component (what scipy uses as the squared row scale, scipy:_lsq/common.py:720-731)."""
    z = np.asarray(z, dtype=np.float64)
    if kind == "linear":
        return np.ones_like(z)
    if kind == "huber":
        return np.where(z <= 1, 1.0, 0.0)
    if kind == "soft_l1":
        return (1 + z) ** -1.5
    if kind == "cauchy":
        return (1 - z) / (1 + z) ** 2
    if kind == "arctan":
        return (1 - 3 * z * z) / (1 + z * z) ** 2
    raise ValueError(kind)


CURV_FLOOR = 0.1  # engine default (BaOptions.curv_floor)


def hessian_weights(f, f_scale=1.0, kind="huber", floor=CURV_FLOOR):
    """Weights of the Gauss-Newton Hessian: max(rho' + 2 rho'' z, floor * rho')."""
    z = (f / f_scale) ** 2
    return np.maximum(loss_curvature(z, kind), floor * loss_rho(z, kind)[1])


def irls_weights(f, f_scale=1.0, kind="huber"):
    """rho'((f/fs)^2) per component: the IRLS / Gauss-Newton weight of the
    robustified cost (gradient = J^T (w * f), scipy:_lsq/common.py:720-731 for g)."""
    return loss_rho((f / f_scale) ** 2, kind)[1]


# --------------------------------------------------------------------------
# Analytic Jacobian blocks
# --------------------------------------------------------------------------
def jacobian_blocks(cam_pose, cam_intr, points, obs_cam, obs_pt):
    """Jc (No,2,6) = d(u,v)/d(rvec,t) and Jp (No,2,3) = d(u,v)/dX per observation.
    These are the non-zero blocks of the matrix the reference approximates by
    2-point differences over ``jac_sparsity`` (BundleAdjustment.py:185-196, :212)."""
    R = rodrigues(cam_pose[:, :3])
    Jl = so3_left_jacobian(cam_pose[:, :3])
    Ro = R[obs_cam]
    X = points[obs_pt]
    Y = np.einsum("nij,nj->ni", Ro, X)
    Xc = Y + cam_pose[obs_cam, 3:6]
    z = Xc[:, 2]
    nz = z != 0
    iz = np.where(nz, 1.0 / np.where(nz, z, 1.0), 1.0)
    fx = cam_intr[obs_cam, 0]
    fy = cam_intr[obs_cam, 1]
    n = len(obs_cam)
    P = np.zeros((n, 2, 3))
    P[:, 0, 0] = fx * iz
    P[:, 1, 1] = fy * iz
    # d(1/z)/dz = -1/z^2 ; when Z == 0 OpenCV substitutes the constant 1 -> derivative 0
    P[:, 0, 2] = np.where(nz, -fx * Xc[:, 0] * iz * iz, 0.0)
    P[:, 1, 2] = np.where(nz, -fy * Xc[:, 1] * iz * iz, 0.0)
    D = -skew(Y) @ Jl[obs_cam]
    Jc = np.concatenate([P @ D, P], axis=2)
    Jp = P @ Ro
    return Jc, Jp


def param_index(cam_fixed, n_points):
    """Column offsets of the reference parameter vector (BundleAdjustment.py:46-65,
    168-176): free cameras in camera order (6 each), then all points (3 each)."""
    cam_fixed = np.asarray(cam_fixed).astype(bool)
    free = np.flatnonzero(~cam_fixed)
    cam_off = np.full(len(cam_fixed), -1, dtype=np.int64)
    cam_off[free] = 6 * np.arange(len(free))
    return cam_off, 6 * len(free), 6 * len(free) + 3 * n_points


def pack(cam_pose, cam_fixed, points):
    """x0 of the reference (BundleAdjustment.py:46-65)."""
    free = ~np.asarray(cam_fixed).astype(bool)
    return np.concatenate([cam_pose[free].reshape(-1), points.reshape(-1)])


def unpack(x, cam_pose0, cam_fixed, n_points):
    """Inverse of ``pack`` (BundleAdjustment.py:67-100), SoA form."""
    free = ~np.asarray(cam_fixed).astype(bool)
    nf = int(free.sum())
    cam_pose = cam_pose0.copy()
    cam_pose[free] = x[:6 * nf].reshape(nf, 6)
    return cam_pose, x[6 * nf:].reshape(n_points, 3).copy()


def dense_jacobian(cam_pose, cam_intr, cam_fixed, points, obs_cam, obs_pt):
    """Full (2No x n) Jacobian in the reference's row/column order."""
    Jc, Jp = jacobian_blocks(cam_pose, cam_intr, points, obs_cam, obs_pt)
    cam_off, pt_off, n = param_index(cam_fixed, len(points))
    J = np.zeros((2 * len(obs_cam), n))
    for k in range(len(obs_cam)):
        co = cam_off[obs_cam[k]]
        if co >= 0:
            J[2 * k:2 * k + 2, co:co + 6] = Jc[k]
        po = pt_off + 3 * obs_pt[k]
        J[2 * k:2 * k + 2, po:po + 3] = Jp[k]
    return J


# --------------------------------------------------------------------------
# Normal equations, Schur complement, LM  (the engine's algorithm, dense, small sizes)
# --------------------------------------------------------------------------
D_MIN, D_MAX = 1e-6, 1e32  # clamp of the Marquardt scaling diag(J^T W J)


def normal_blocks(cam_pose, cam_intr, points, obs_cam, obs_pt, obs_xy, f_scale, kind, floor=CURV_FLOOR):
    """Per-camera U (Nc,6,6), g_c (Nc,6); per-point V (Np,3,3), g_p (Np,3);
    per-observation W = Jc^T diag(wh) Jp (No,6,3); cost F.  Gradients use w = rho', Hessian
    blocks use wh = max(rho' + 2 rho'' z, floor rho')."""
    f = residuals(cam_pose, cam_intr, points, obs_cam, obs_pt, obs_xy)
    w = irls_weights(f, f_scale, kind).reshape(-1, 2)
    wh = hessian_weights(f, f_scale, kind, floor).reshape(-1, 2)
    r = f.reshape(-1, 2)
    Jc, Jp = jacobian_blocks(cam_pose, cam_intr, points, obs_cam, obs_pt)
    Nc, Np = len(cam_pose), len(points)
    U = np.zeros((Nc, 6, 6)); gc = np.zeros((Nc, 6))
    V = np.zeros((Np, 3, 3)); gp = np.zeros((Np, 3))
    hJc = Jc * wh[:, :, None]
    hJp = Jp * wh[:, :, None]
    wr = w * r
    np.add.at(U, obs_cam, np.einsum("nki,nkj->nij", hJc, Jc))
    np.add.at(gc, obs_cam, np.einsum("nki,nk->ni", Jc, wr))
    np.add.at(V, obs_pt, np.einsum("nki,nkj->nij", hJp, Jp))
    np.add.at(gp, obs_pt, np.einsum("nki,nk->ni", Jp, wr))
    W = np.einsum("nki,nkj->nij", hJc, Jp)
    return dict(U=U, gc=gc, V=V, gp=gp, W=W, F=cost(f, f_scale, kind), f=f, w=w, wh=wh, Jc=Jc, Jp=Jp)


def reduced_system(nb, cam_fixed, obs_cam, obs_pt, lam):
    """Schur complement of the point blocks: dense S (6Nf x 6Nf), g_red (6Nf), plus the
    per-point inverse blocks needed for back-substitution."""
    cam_fixed = np.asarray(cam_fixed).astype(bool)
    cam_off, ncp, _ = param_index(cam_fixed, 0)
    U, gc, V, gp, W = nb["U"], nb["gc"], nb["V"], nb["gp"], nb["W"]
    Np = len(V)
    Dp = np.clip(np.einsum("nii->ni", V), D_MIN, D_MAX)
    Vd = V + lam * np.einsum("ni,ij->nij", Dp, np.eye(3))
    Vinv = np.linalg.inv(Vd)
    S = np.zeros((ncp, ncp)); gred = np.zeros(ncp)
    Dc = np.clip(np.einsum("nii->ni", U), D_MIN, D_MAX)
    for c in np.flatnonzero(~cam_fixed):
        o = cam_off[c]
        S[o:o + 6, o:o + 6] += U[c] + lam * np.diag(Dc[c])
        gred[o:o + 6] += gc[c]
    order = np.argsort(obs_pt, kind="stable")
    starts = np.searchsorted(obs_pt[order], np.arange(Np + 1))
    six = np.arange(6)
    for p in range(Np):
        idx = order[starts[p]:starts[p + 1]]
        idx = idx[cam_off[obs_cam[idx]] >= 0]
        if len(idx) == 0:
            continue
        # all (a, b) pairs of the point's observations at once: S[a-block, b-block] -= W_a Vinv W_b^T
        rows = (cam_off[obs_cam[idx]][:, None] + six[None, :]).reshape(-1)
        Y = W[idx] @ Vinv[p]                                   # (L, 6, 3)
        np.subtract.at(gred, rows, (Y @ gp[p]).reshape(-1))
        blocks = np.einsum("aik,bjk->aibj", Y, W[idx]).reshape(len(rows), len(rows))
        np.subtract.at(S, (rows[:, None], rows[None, :]), blocks)   # (.at: a camera may see the point twice)
    return S, gred, Vinv, Dc, Dp


def reduced_matvec(nb, cam_fixed, obs_cam, obs_pt, lam, v):
    """Matrix-free S v and g_red of the same reduced system (vectorised: usable at full BAL sizes, where the
    dense S of :func:`reduced_system` would not fit): S v = (U + lam D_c) v - W (V + lam D_p)^-1 W^T v."""
    cam_fixed = np.asarray(cam_fixed).astype(bool)
    U, gc, V, gp, W = nb["U"], nb["gc"], nb["V"], nb["gp"], nb["W"]
    free = np.flatnonzero(~cam_fixed)
    vc = np.zeros((len(cam_fixed), 6)); vc[free] = np.asarray(v).reshape(-1, 6)
    Dp = np.clip(np.einsum("nii->ni", V), D_MIN, D_MAX)
    Vd = V + lam * np.einsum("ni,ij->nij", Dp, np.eye(3))
    seen = np.zeros(len(V), dtype=bool); seen[obs_pt] = True
    Vd[~seen] = np.eye(3)                                    # unobserved points: no contribution
    Vinv = np.linalg.inv(Vd)
    Dc = np.clip(np.einsum("nii->ni", U), D_MIN, D_MAX)
    out = np.einsum("nij,nj->ni", U, vc) + lam * Dc * vc
    fo = ~cam_fixed[obs_cam]                                 # observations of free cameras
    t = np.zeros((len(V), 3)); np.add.at(t, obs_pt[fo], np.einsum("nij,ni->nj", W[fo], vc[obs_cam[fo]]))
    t = np.einsum("nij,nj->ni", Vinv, t)
    np.subtract.at(out, obs_cam[fo], np.einsum("nij,nj->ni", W[fo], t[obs_pt[fo]]))
    gred = gc.copy()
    u = np.einsum("nij,nj->ni", Vinv, gp)
    np.subtract.at(gred, obs_cam[fo], np.einsum("nij,nj->ni", W[fo], u[obs_pt[fo]]))
    return out[free].reshape(-1), gred[free].reshape(-1)


def lm_step(nb, cam_fixed, obs_cam, obs_pt, lam):
    """One damped Gauss-Newton step by Schur elimination: (delta_c (Nc,6), delta_p (Np,3))."""
    cam_fixed = np.asarray(cam_fixed).astype(bool)
    cam_off, ncp, _ = param_index(cam_fixed, 0)
    S, gred, Vinv, _, _ = reduced_system(nb, cam_fixed, obs_cam, obs_pt, lam)
    dc_flat = np.linalg.solve(S, -gred) if ncp else np.zeros(0)
    dc = np.zeros((len(cam_fixed), 6))
    free = np.flatnonzero(~cam_fixed)
    dc[free] = dc_flat.reshape(-1, 6)
    rhs = nb["gp"].copy()
    wt = np.einsum("nij,ni->nj", nb["W"], dc[obs_cam])
    np.add.at(rhs, obs_pt, wt)
    dp = -np.einsum("nij,nj->ni", Vinv, rhs)
    return dc, dp


def lm_solve(cam_pose, cam_intr, cam_fixed, points, obs_cam, obs_pt, obs_xy,
             f_scale=1.0, kind="huber", max_iter=100, ftol=1e-8, xtol=1e-8, gtol=1e-8,
             lam0=1e-4, floor_min=CURV_FLOOR, verbose=False):
    """Dense restatement of the engine's Levenberg-Marquardt control (Nielsen damping
    update; termination tests of scipy:_lsq/common.py:705-717 and trf.py:451-459).
    Returns dict(cam_pose, points, cost, status, iterations, history)."""
    cam_fixed = np.asarray(cam_fixed).astype(bool)
    cam_pose = cam_pose.copy(); points = points.copy()
    lam, nu, floor = lam0, 2.0, 1.0
    hist = []
    status, it = 0, 0
    nb = normal_blocks(cam_pose, cam_intr, points, obs_cam, obs_pt, obs_xy, f_scale, kind, floor)
    while it < max_iter:
        gfree = np.concatenate([nb["gc"][~cam_fixed].reshape(-1), nb["gp"].reshape(-1)])
        gnorm = np.abs(gfree).max() if gfree.size else 0.0
        if gnorm < gtol:
            status = 1
            break
        dc, dp = lm_step(nb, cam_fixed, obs_cam, obs_pt, lam)
        it += 1
        Jd = np.einsum("nij,nj->ni", nb["Jc"], dc[obs_cam]) + np.einsum("nij,nj->ni", nb["Jp"], dp[obs_pt])
        r = nb["f"].reshape(-1, 2)
        pred = -float((Jd * (nb["w"] * r + 0.5 * nb["wh"] * Jd)).sum())
        cp_t = cam_pose + dc; pt_t = points + dp
        F_t = cost(residuals(cp_t, cam_intr, pt_t, obs_cam, obs_pt, obs_xy), f_scale, kind)
        dF = nb["F"] - F_t
        rho = dF / pred if pred > 0 else -1.0
        step_norm = np.sqrt((dc[~cam_fixed] ** 2).sum() + (dp ** 2).sum())
        x_norm = np.sqrt((cam_pose[~cam_fixed] ** 2).sum() + (points ** 2).sum())
        hist.append((it, nb["F"], F_t, rho, lam, gnorm))
        if verbose:
            print(f"[oracle-lm] it {it} F {nb['F']:.9e} F_t {F_t:.9e} rho {rho:.3f} lam {lam:.2e} |g| {gnorm:.2e}")
        ftol_ok = dF < ftol * nb["F"] and rho > 0.25
        xtol_ok = step_norm < xtol * (xtol + x_norm)
        if rho > 0 and np.isfinite(F_t):
            cam_pose, points = cp_t, pt_t
            lam *= max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3); nu = 2.0
            if rho > 0.5 and dF < 0.05 * nb["F"]:
                floor = max(floor_min, 0.25 * floor)
            nb = normal_blocks(cam_pose, cam_intr, points, obs_cam, obs_pt, obs_xy, f_scale, kind, floor)
        else:
            lam *= nu; nu *= 2.0
            floor = min(1.0, 4.0 * floor)
            nb = normal_blocks(cam_pose, cam_intr, points, obs_cam, obs_pt, obs_xy, f_scale, kind, floor)
        lam = min(max(lam, 1e-12), 1e12)
        if ftol_ok and xtol_ok:
            status = 4; break
        if ftol_ok:
            status = 2; break
        if xtol_ok:
            status = 3; break
    gfree = np.concatenate([nb["gc"][~cam_fixed].reshape(-1), nb["gp"].reshape(-1)])
    return dict(cam_pose=cam_pose, points=points, cost=nb["F"], status=status, iterations=it,
                grad_inf=float(np.abs(gfree).max()) if gfree.size else 0.0, history=hist)


def reprojection_error_per_camera(cam_pose, K_full, points, obs_cam, obs_pt, obs_xy):
    """Mean L2 reprojection error per camera with the *full* K product
    (utils.py:271-276 via BundleAdjustment.py:388-423).  K_full: (Nc,3,3)."""
    R = rodrigues(cam_pose[:, :3])
    out = np.full(len(cam_pose), np.nan)
    for c in range(len(cam_pose)):
        m = obs_cam == c
        if not m.any():
            continue
        pc = R[c] @ points[obs_pt[m]].T + cam_pose[c, 3:6].reshape(3, 1)
        ph = K_full[c] @ pc
        uv = (ph[:2] / ph[2]).T
        out[c] = np.mean(np.linalg.norm(obs_xy[m] - uv, axis=1))
    return out
