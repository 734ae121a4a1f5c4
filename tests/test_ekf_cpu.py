"""The measurement -> state step: oracle Ekf8 vs the reference's own Kalman-filter methods
(fixture O10, generated from PyGNSS' vector/ekf.py `_m5` methods), and the C-ABI dpe_ekf_* (host fp64, runs
without a GPU) vs the oracle in CUDARecv's call order (StepUpdate then StepPredict, cuekf.cu:575-588)."""
import numpy as np
import pytest

import navlab_dpe_sdr_amd as dpe


@pytest.mark.parametrize("tag,couple", [("fI", False), ("fT", True)])
def test_oracle_ekf_matches_pygnss_m5(golden, tag, couple):
    from oracle import oracle as o
    g = golden("o10_ekf")
    kf = o.Ekf8(g["x0"], None, float(g["T"]), couple_velocity=couple)
    for it in range(g[tag + "_z"].shape[0]):
        kf.predict()
        assert np.allclose(kf.x, g[tag + "_x_pred"][it], rtol=0, atol=1e-7)          # |x| ~ 4e6 m: 1e-7 m absolute
        assert np.allclose(kf.Q, g[tag + "_Q"][it], rtol=1e-13, atol=1e-18)
        assert np.allclose(kf.P, g[tag + "_P_pred"][it], rtol=1e-11, atol=1e-14)
        kf.update(g[tag + "_z"][it])
        assert np.allclose(kf.K, g[tag + "_K"][it], rtol=1e-10, atol=1e-13)
        assert np.allclose(kf.x, g[tag + "_x_upd"][it], rtol=0, atol=1e-7)
        assert np.allclose(kf.P, g[tag + "_P_upd"][it], rtol=1e-10, atol=1e-13)
    q = g[tag + "_Q"][:, 4, 4]
    assert q.min() < 4.0 and q.max() > 5.5          # the running-average speed sweeps the clamp range of _update_Q


def test_c_abi_ekf_matches_oracle_in_cudarecv_order(golden):
    from oracle import oracle as o
    g = golden("o10_ekf")
    x0, T = g["x0"], float(g["T"])
    rng = np.random.default_rng(5)
    P0 = np.eye(8) * 4.0
    R = np.diag([1.0, 1.0, 1.0, 4.0, 0.25, 0.25, 0.25, 0.01])
    ekf = dpe.cuEKF(x0, InitP=P0, SampleLength=T, EnableEKF=True)
    # oracle in the same order: the first StepUpdate uses P_k|k-1 = I (cuekf.cu:464), InitP only enters through... nothing
    ref = o.Ekf8(x0, np.eye(8), T, couple_velocity=True)
    for it in range(30):
        z = g["fT_z"][it] + rng.normal(0, 0.1, 8)
        ekf.Update(z, R)
        ref.update(z, R)
        xk1k1 = ref.x.copy()
        ref.predict()
        st = ekf.state()
        assert np.allclose(st["xk1k1"], xk1k1, rtol=0, atol=1e-7)
        assert np.allclose(st["xkk1"], ref.x, rtol=0, atol=1e-7)
        assert np.allclose(st["Pkk1"], ref.P, rtol=1e-10, atol=1e-13)
        assert np.allclose(st["Q"], ref.Q, rtol=1e-13, atol=1e-18)
        assert np.allclose(st["K"], ref.K, rtol=1e-10, atol=1e-13)
        assert np.array_equal(ekf.xCurrkk1, st["xkk1"]) and np.array_equal(ekf.xCurrk1k1, st["xk1k1"])
    ekf.Stop()


def test_pass_through_is_the_shipped_behaviour():
    """EnableEKF=false (dpeflow.cpp:90): EKF_PassMeas copies zVal to both state ports (cuekf.cu:147-159)."""
    ekf = dpe.cuEKF(np.arange(8.0), EnableEKF=False)
    z = np.linspace(1, 2, 8)
    ekf.Update(z)
    assert np.array_equal(ekf.xCurrk1k1, z) and np.array_equal(ekf.xCurrkk1, z)


def test_singular_innovation_covariance_is_reported():
    ekf = dpe.cuEKF(np.zeros(8), InitP=np.zeros((8, 8)), EnableEKF=True)
    R = -np.eye(8)                                   # S = H I H^T + R = 0
    with pytest.raises(dpe.DpeError, match="S inversion failed"):
        ekf.Update(np.ones(8), R)
