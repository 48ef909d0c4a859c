"""The oracle reproduces the committed golden fixtures (tests/golden/make_golden.py)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402

from teb_local_planner_amd import _abi  # noqa: E402


def check_against_golden(name, out, res, pos_tol, cost_rtol):
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    np.testing.assert_array_equal(out.n, g["n"])
    np.testing.assert_array_equal(res.status, g["status"])
    np.testing.assert_array_equal(res.lm_iterations, g["lm_iterations"])
    np.testing.assert_array_equal(res.lm_trials, g["lm_trials"])
    np.testing.assert_allclose(res.chi2, g["chi2"], rtol=cost_rtol)
    np.testing.assert_allclose(res.cost, g["cost"], rtol=cost_rtol)
    for b in range(out.count):
        x, y, th, dt = out.get_teb(b)
        for u, k in ((x, "x"), (y, "y"), (th, "th"), (dt, "dt")):
            assert np.abs(u - g["%s%d" % (k, b)]).max() <= pos_tol, (name, b, k)


@pytest.mark.parametrize("name", sorted(make_golden.CASES))
def test_oracle_reproduces_golden(oracle, name):
    cfg, obst, via, batch = make_golden.CASES[name]()
    cfg.jacobian_mode = _abi.JACOBIAN_ANALYTIC
    out, res = oracle.optimize_batch(cfg, obst, via, batch)
    check_against_golden(name, out, res, pos_tol=1e-12, cost_rtol=1e-12)


@pytest.mark.parametrize("name", ["c1_test_optim_node", "mixed_polygon"])
def test_faithful_mode_stays_within_reference_noise_of_golden(oracle, name):
    """g2o central differences (delta = 1e-9) vs closed-form Jacobians: SURVEY §8c T3 (1e-3 m / rad, 1e-3 rel)."""
    cfg, obst, via, batch = make_golden.CASES[name]()
    cfg.jacobian_mode = _abi.JACOBIAN_G2O_NUMERIC
    out, res = oracle.optimize_batch(cfg, obst, via, batch)
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    np.testing.assert_array_equal(out.n, g["n"])
    np.testing.assert_allclose(res.cost, g["cost"], rtol=1e-3)
    for b in range(out.count):
        x, y, th, dt = out.get_teb(b)
        for u, k in ((x, "x"), (y, "y"), (th, "th"), (dt, "dt")):
            assert np.abs(u - g["%s%d" % (k, b)]).max() <= 1e-3


def test_threaded_batch_equals_sequential(oracle):
    cfg, obst, via, batch = make_golden.CASES["mixed_point"]()
    a, ra = oracle.optimize_batch(cfg, obst, via, batch, threads=1)
    b, rb = oracle.optimize_batch(cfg, obst, via, batch, threads=3)
    np.testing.assert_array_equal(a.x, b.x)
    np.testing.assert_array_equal(ra.cost, rb.cost)
