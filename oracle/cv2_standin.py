"""TEST INFRASTRUCTURE ONLY -- three-symbol stand-in for ``cv2``.

The reference module imports ``cv2`` (``/root/reference/src/BundleAdjustment.py:2``)
and uses exactly three symbols of it on the bundle-adjustment path:

* ``cv2.Rodrigues``      (``BundleAdjustment.py:57, 82, 113``)
* ``cv2.projectPoints``  (``BundleAdjustment.py:142``, ``distCoeffs=None``)
* ``cv2.error``          (``BundleAdjustment.py:147``)

OpenCV (``opencv-python``, version unpinned upstream) is not installed in the
build image and there is no network, so ``tests/golden/make_golden.py``
pre-seeds ``sys.modules['cv2']`` with this module before importing the
reference unmodified.  The arithmetic below restates OpenCV's *published*
camera model (calib3d documentation): **parity with the real OpenCV binary is
unpinned** -- nothing in the reference tests it.

Model (all in float64, as OpenCV does internally):
  Rodrigues vec->mat:  theta = |r|;  theta < DBL_EPSILON -> I
                       R = cos(theta) I + (1-cos(theta)) k k^T + sin(theta) [k]x,  k = r/theta
  Rodrigues mat->vec:  SVD re-orthonormalisation, then the antisymmetric part
                       with theta = acos((tr R - 1)/2) and the theta~pi branch.
  projectPoints:       Xc = R X + t;  z' = 1/Z (1 if Z == 0);
                       u = fx X z' + cx;  v = fy Y z' + cy   (skew ignored, no distortion)
"""
import numpy as np


class error(Exception):
    """Stand-in for ``cv2.error`` (never raised by this module)."""


_DBL_EPSILON = float(np.finfo(np.float64).eps)


def rodrigues_vec_to_mat(r):
    r = np.asarray(r, dtype=np.float64).reshape(3)
    theta = float(np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]))
    if theta < _DBL_EPSILON:
        return np.eye(3)
    c = np.cos(theta)
    s = np.sin(theta)
    c1 = 1.0 - c
    k = r / theta
    kx = np.array([[0.0, -k[2], k[1]],
                   [k[2], 0.0, -k[0]],
                   [-k[1], k[0], 0.0]])
    return c * np.eye(3) + c1 * np.outer(k, k) + s * kx


def rodrigues_mat_to_vec(Rin):
    Rin = np.asarray(Rin, dtype=np.float64).reshape(3, 3)
    U, _, Vt = np.linalg.svd(Rin)
    R = U @ Vt
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.sqrt((r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * 0.25)
    c = (R[0, 0] + R[1, 1] + R[2, 2] - 1.0) * 0.5
    c = min(1.0, max(-1.0, c))
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        t = (R[0, 0] + 1.0) * 0.5
        rx = np.sqrt(max(t, 0.0))
        t = (R[1, 1] + 1.0) * 0.5
        ry = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
        t = (R[2, 2] + 1.0) * 0.5
        rz = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(rx) < abs(ry) and abs(rx) < abs(rz) and ((R[1, 2] > 0) != (ry * rz > 0)):
            rz = -rz
        v = np.array([rx, ry, rz])
        return v * (theta / np.linalg.norm(v))
    return r * (theta / (2.0 * s))


def Rodrigues(src, dst=None, jacobian=None):
    a = np.asarray(src, dtype=np.float64)
    if a.shape == (3, 3):
        return rodrigues_mat_to_vec(a).reshape(3, 1), None
    if a.size == 3:
        return rodrigues_vec_to_mat(a), None
    raise error("Rodrigues: input must be 3x3, 3x1 or 1x3")


def projectPoints(objectPoints, rvec, tvec, cameraMatrix, distCoeffs, *_a, **_k):
    X = np.asarray(objectPoints, dtype=np.float64).reshape(-1, 3)
    rv = np.asarray(rvec, dtype=np.float64)
    R = rodrigues_vec_to_mat(rv) if rv.size == 3 else rv.reshape(3, 3)
    t = np.asarray(tvec, dtype=np.float64).reshape(3)
    K = np.asarray(cameraMatrix, dtype=np.float64)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    out = np.empty((X.shape[0], 1, 2), dtype=np.float64)
    for i in range(X.shape[0]):
        x = R[0, 0] * X[i, 0] + R[0, 1] * X[i, 1] + R[0, 2] * X[i, 2] + t[0]
        y = R[1, 0] * X[i, 0] + R[1, 1] * X[i, 1] + R[1, 2] * X[i, 2] + t[1]
        z = R[2, 0] * X[i, 0] + R[2, 1] * X[i, 1] + R[2, 2] * X[i, 2] + t[2]
        z = 1.0 / z if z != 0 else 1.0
        out[i, 0, 0] = fx * (x * z) + cx
        out[i, 0, 1] = fy * (y * z) + cy
    return out, None
