"""Host-side (NumPy) camera arithmetic used by the marshalling layer.

The reference converts between rotation matrices and axis-angle vectors with
``cv2.Rodrigues`` when it packs / unpacks the parameter vector
(``/root/reference/src/BundleAdjustment.py:57, 82, 113``).  OpenCV is not a
dependency of this package; the two conversions are restated here from OpenCV's
documented model (float64).  This is boundary marshalling (O(#cameras) per
solve), not the hot path -- the hot path runs in HIP (``csrc/``).
"""
import numpy as np

_DBL_EPS = float(np.finfo(np.float64).eps)


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


def rvec_to_matrix(rvec):
    """Axis-angle (...,3) -> rotation matrix (...,3,3) (``cv2.Rodrigues`` vector->matrix)."""
    r = np.asarray(rvec, dtype=np.float64)
    theta = np.sqrt((r * r).sum(-1))
    small = theta < _DBL_EPS
    th = np.where(small, 1.0, theta)
    k = r / th[..., None]
    c = np.cos(th)[..., None, None]
    s = np.sin(th)[..., None, None]
    eye = np.broadcast_to(np.eye(3), r.shape[:-1] + (3, 3))
    R = c * eye + (1.0 - c) * (k[..., :, None] * k[..., None, :]) + s * skew(k)
    return np.where(small[..., None, None], eye, R)


def _rvec_from_svd(U, Vt):
    R = U @ Vt
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.sqrt(float(r @ r) * 0.25)
    c = min(1.0, max(-1.0, (np.trace(R) - 1.0) * 0.5))
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        d = np.sqrt(np.maximum((np.diag(R) + 1.0) * 0.5, 0.0))
        rx, ry, rz = d[0], d[1] * (-1.0 if R[0, 1] < 0 else 1.0), d[2] * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(rx) < abs(ry) and abs(rx) < abs(rz) and ((R[1, 2] > 0) != (ry * rz > 0)):
            rz = -rz
        v = np.array([rx, ry, rz])
        return v * (theta / np.linalg.norm(v))
    return r * (theta / (2.0 * s))


def matrix_to_rvec(Rin):
    """Rotation matrix (3,3) -> axis-angle (3,) (``cv2.Rodrigues`` matrix->vector):
    nearest rotation by SVD, then the antisymmetric part scaled by theta/(2 sin theta),
    with the separate branch near theta = pi."""
    Rin = np.asarray(Rin, dtype=np.float64).reshape(3, 3)
    U, _, Vt = np.linalg.svd(Rin)
    return _rvec_from_svd(U, Vt)


def matrices_to_rvecs(Rs):
    """:func:`matrix_to_rvec` for a stack (n, 3, 3) -> (n, 3) with ONE stacked SVD call (most of the time of the scalar
    function is the call into LAPACK, not its arithmetic); everything behind the SVD is the scalar code, per matrix,
    so the two agree bit for bit."""
    Rs = np.asarray(Rs, dtype=np.float64).reshape(-1, 3, 3)
    out = np.zeros((len(Rs), 3))
    if len(Rs):
        U, _, Vt = np.linalg.svd(Rs)
        for i in range(len(Rs)):
            out[i] = _rvec_from_svd(U[i], Vt[i])
    return out


def project_points(cam_pose, cam_intr, points, obs_cam, obs_pt):
    """Pinhole projection (no distortion, skew ignored) of each observation -> (No,2)."""
    R = rvec_to_matrix(cam_pose[:, :3])[obs_cam]
    Xc = np.einsum("nij,nj->ni", R, points[obs_pt]) + cam_pose[obs_cam, 3:6]
    z = Xc[:, 2]
    iz = np.where(z != 0, 1.0 / np.where(z != 0, z, 1.0), 1.0)
    K = cam_intr[obs_cam]
    return np.stack([K[:, 0] * (Xc[:, 0] * iz) + K[:, 2], K[:, 1] * (Xc[:, 1] * iz) + K[:, 3]], axis=1)
