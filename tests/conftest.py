import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
SMALL_CASES = ["street_clean", "street_near", "street_outliers", "street_inter", "ring16_dense",
               "ring_sfm_shape", "ring_unobserved", "street_zero_rvec"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))


def problem_of(g):
    """The golden problem at exactly the reference's starting point: free camera poses and points
    are taken from the captured x0 (the reference converts R -> rvec before packing, which e.g.
    maps |rvec| < 1e-5 to 0, BundleAdjustment.py:57)."""
    cam_pose = g["cam_pose"].copy()
    free = g["cam_fixed"] == 0
    nf = int(free.sum())
    cam_pose[free] = g["x0"][:6 * nf].reshape(nf, 6)
    points = g["x0"][6 * nf:].reshape(-1, 3).copy()
    return dict(cam_pose=cam_pose, cam_intr=g["cam_intr"], cam_fixed=g["cam_fixed"], points=points,
                obs_cam=g["obs_cam"], obs_pt=g["obs_pt"], obs_xy=g["obs_xy"])


try:   # PyTorch-ROCm brings its own HIP runtime: loaded (and asked) BEFORE the library initialises the one it links, both
    import torch   # serve the process; the other way round torch.cuda.is_available() turns False
    _TORCH_SEES_GPU = torch.cuda.is_available()
except Exception:
    _TORCH_SEES_GPU = False


def _gpu_available():
    if _TORCH_SEES_GPU:
        return True
    try:   # the library's own answer
        from structure_from_motion_amd import _capi
        return bool(_capi.warm_up())
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    if not _gpu_available():
        pytest.fail("GPU test selected but no GPU is visible (there is no CPU fallback)")
    return True
