"""CPU: the oracle (oracle/ba_oracle.py) against the golden vectors captured from the reference's
own residual closure and scipy solve (tests/golden/make_golden.py)."""
import numpy as np
import pytest
from conftest import SMALL_CASES, load_golden, problem_of

from oracle import ba_oracle as O


@pytest.mark.parametrize("name", SMALL_CASES)
def test_pack_and_residuals_match_reference_closure(name):
    g = load_golden(name)
    p = problem_of(g)
    x0 = O.pack(p["cam_pose"], p["cam_fixed"], p["points"])
    # reference x0 went R -> cv2.Rodrigues -> rvec: equal up to the round trip
    np.testing.assert_allclose(x0, g["x0"], rtol=0, atol=1e-12)
    f = O.residuals(p["cam_pose"], p["cam_intr"], p["points"], p["obs_cam"], p["obs_pt"], p["obs_xy"])
    # tolerance: 1e-8 relative (north_star); achieved ~1e-12
    np.testing.assert_allclose(f, g["f0"], rtol=1e-8, atol=1e-9)
    assert np.linalg.norm(f - g["f0"]) <= 1e-10 * np.linalg.norm(g["f0"])
    for fs in (1.0, 1.5):
        z = (g["f0"] / fs) ** 2
        ref_cost = 0.5 * fs * fs * np.where(z <= 1, z, 2 * np.sqrt(z) - 1).sum()
        assert abs(O.cost(f, fs, "huber") - ref_cost) <= 1e-10 * ref_cost


@pytest.mark.parametrize("name", SMALL_CASES)
def test_analytic_jacobian_matches_reference_differences(name):
    g = load_golden(name)
    p = problem_of(g)
    J = O.dense_jacobian(p["cam_pose"], p["cam_intr"], p["cam_fixed"], p["points"], p["obs_cam"], p["obs_pt"])
    n = J.shape[1]
    Jh = np.zeros_like(J); Jh[g["jh_row"], g["jh_col"]] = g["jh_val"]
    J2 = np.zeros_like(J); J2[g["j2_row"], g["j2_col"]] = g["j2_val"]
    assert n == len(g["x0"])
    # 4th-order differences of the reference closure: 1e-8 relative, norm-wise
    assert np.linalg.norm(J - Jh) <= 1e-8 * np.linalg.norm(Jh)
    # scipy's own 2-point Jacobian is only accurate to ~sqrt(eps): report-level agreement
    assert np.linalg.norm(J - J2) <= 1e-5 * np.linalg.norm(J2)
    # identical sparsity: nothing outside the reference pattern
    mask = np.zeros_like(J, dtype=bool); mask[g["j2_row"], g["j2_col"]] = True
    mask[g["jh_row"], g["jh_col"]] = True
    assert np.all(J[~mask] == 0.0)


@pytest.mark.parametrize("name", SMALL_CASES)
def test_oracle_lm_reaches_the_tight_minimum(name):
    g = load_golden(name)
    p = problem_of(g)
    res = O.lm_solve(p["cam_pose"], p["cam_intr"], p["cam_fixed"], p["points"], p["obs_cam"], p["obs_pt"],
                     p["obs_xy"], f_scale=float(g["f_scale"]), max_iter=300)
    tight = float(g["tight_cost"])
    assert res["status"] > 0
    assert res["cost"] <= tight * (1 + 1e-6)
    if int(g["ref_status"]) > 0:  # the reference converged: ours must not be worse (protocol (ii))
        assert res["cost"] <= float(g["ref_cost"]) * (1 + 1e-6)
    # the tight minimum really is one for the reference closure
    assert float(g["tight_grad_inf"]) < 1e-3


def test_reference_stall_is_recorded():
    """with Huber components in the linear zone the reference runs out of budget."""
    g = load_golden("street_outliers")
    assert int(g["ref_status"]) == 0 and int(g["ref_nfev"]) == int(g["max_nfev"])
    assert float(g["ref_cost"]) > float(g["tight_cost"]) * 1.01
    assert not bool(g["ref_success_flag"])


def test_reprojection_error_formula():
    g = load_golden("ring16_dense")
    p = problem_of(g)
    K = np.zeros((len(p["cam_pose"]), 3, 3))
    K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = (*p["cam_intr"].T, 1.0)
    e = O.reprojection_error_per_camera(p["cam_pose"], K, p["points"], p["obs_cam"], p["obs_pt"], p["obs_xy"])
    f = g["f0"].reshape(-1, 2)
    for c in range(len(e)):
        m = p["obs_cam"] == c
        if m.any():
            assert abs(e[c] - np.linalg.norm(f[m], axis=1).mean()) < 1e-9


def test_temple_shaped_fixture_pins_oracle():
    """Temple-shaped final BA at 15 % scale: f(x0) of the reference closure, and the cost the reference closure
    gave at the stored minimiser (tests/golden/make_golden.py --tight-for)."""
    g = load_golden("temple_final_s015")
    t = load_golden("temple_final_s015_tight")
    p = problem_of(g)
    f = O.residuals(p["cam_pose"], p["cam_intr"], p["points"], p["obs_cam"], p["obs_pt"], p["obs_xy"])
    np.testing.assert_allclose(f, g["f0"], rtol=1e-9, atol=1e-10)
    cp, pt = O.unpack(t["tight_x"], p["cam_pose"], p["cam_fixed"], len(p["points"]))
    ft = O.residuals(cp, p["cam_intr"], pt, p["obs_cam"], p["obs_pt"], p["obs_xy"])
    c = O.cost(ft, float(g["f_scale"]), "huber")
    assert abs(c - float(t["tight_cost"])) <= 1e-11 * c
    assert float(t["tight_grad_inf"]) < 1e-5 and c < float(g["ref_cost"])
    # the reference's own end state: its cost through the oracle objective
    cr, pr = O.unpack(g["ref_x"], p["cam_pose"], p["cam_fixed"], len(p["points"]))
    fr = O.residuals(cr, p["cam_intr"], pr, p["obs_cam"], p["obs_pt"], p["obs_xy"])
    np.testing.assert_allclose(fr, g["ref_fun"], rtol=1e-8, atol=1e-9)
    assert abs(O.cost(fr, float(g["f_scale"]), "huber") - float(g["ref_cost"])) <= 1e-10 * float(g["ref_cost"])


@pytest.mark.parametrize("name", ["temple_final", "fountain_final", "fountain_final_sfm", "temple_final_sfm"])
def test_config_size_fixtures_pin_oracle(name):
    """BASELINE configs 0 / 1 at full size (make_golden.py --config-size): the reference's x0 and its closure's f(x0),
    and the reference-closure cost at the stored minimiser, against the oracle."""
    g = load_golden(name)
    p = problem_of(g)
    np.testing.assert_allclose(O.pack(p["cam_pose"], p["cam_fixed"], p["points"]), g["x0"], rtol=0, atol=1e-12)
    f = O.residuals(p["cam_pose"], p["cam_intr"], p["points"], p["obs_cam"], p["obs_pt"], p["obs_xy"])
    np.testing.assert_allclose(f, g["f0"], rtol=1e-8, atol=1e-9)
    assert np.linalg.norm(f - g["f0"]) <= 1e-10 * np.linalg.norm(g["f0"])
    fs = float(g["f_scale"])
    assert abs(O.cost(f, fs, "huber") - float(g["cost0"])) <= 1e-11 * float(g["cost0"])
    cp, pt = O.unpack(g["tight_x"], p["cam_pose"], p["cam_fixed"], len(p["points"]))
    ft = O.residuals(cp, p["cam_intr"], pt, p["obs_cam"], p["obs_pt"], p["obs_xy"])
    c = O.cost(ft, fs, "huber")
    assert abs(c - float(g["tight_cost"])) <= 1e-11 * c
    assert float(g["tight_grad_inf"]) < 1e-4 and int(g["tight_status"]) > 0
    if name.endswith("_sfm"):        # tracks of 2 (seed points) and 1 (everything else)
        assert set(np.bincount(p["obs_pt"])) <= {1, 2}


def test_matrix_free_schur_product_equals_dense():
    g = load_golden("ring_unobserved")
    p = problem_of(g)
    nb = O.normal_blocks(p["cam_pose"], p["cam_intr"], p["points"], p["obs_cam"], p["obs_pt"], p["obs_xy"], 1.0, "huber")
    S, gred, _, _, _ = O.reduced_system(nb, p["cam_fixed"], p["obs_cam"], p["obs_pt"], 1e-3)
    v = np.random.default_rng(0).normal(size=S.shape[0])
    Sv, g2 = O.reduced_matvec(nb, p["cam_fixed"], p["obs_cam"], p["obs_pt"], 1e-3, v)
    np.testing.assert_allclose(Sv, S @ v, rtol=1e-10, atol=1e-10 * np.abs(S @ v).max())
    np.testing.assert_allclose(g2, gred, rtol=1e-10, atol=1e-10 * np.abs(gred).max())
