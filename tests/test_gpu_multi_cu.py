"""Multi-CU mode (teb_local_planner_amd/csrc/teb_multicu.hpp): small batches of generic-shape scenes spread the obstacle association
and the robot <-> obstacle distances over helper workgroups on the otherwise idle CUs. The contract: the bands are BIT-IDENTICAL to the
single-CU launch (same device functions on the same inputs, rows accumulated in the same order) - for every footprint kind, with
dynamic obstacles, via-points, the velocity-obstacle-ratio edges, any number of helpers, bands longer than the workgroup; and when the
helpers do not arrive in time the launch is repeated on one CU per band with the same result."""
import numpy as np
import pytest

from teb_local_planner_amd import scenes, planner, _abi

pytestmark = pytest.mark.gpu


def _run(cfg, obst, via, batch, **opt):
    s = planner.make_solver(cfg, obst, via, batch, options=_abi.Options(**opt))
    s.optimize(cfg.optim.no_inner_iterations, cfg.optim.no_outer_iterations, True, cfg.hcp.selection_obst_cost_scale,
               cfg.hcp.selection_viapoint_cost_scale, cfg.hcp.selection_alternative_time_cost)
    res = s.results()
    out = s.download(batch.copy())
    info = s.last_launch_info()
    flags = s.debug_overflow_flags()
    ms = s.last_kernel_ms()
    s.close()
    return out, res, info, flags, ms


def _assert_identical(a, ra, b, rb):
    np.testing.assert_array_equal(a.n, b.n)
    for k in ("x", "y", "theta", "dt"):
        np.testing.assert_array_equal(getattr(a, k), getattr(b, k))
    for k in ("status", "lm_iterations", "lm_trials", "chi2", "cost", "lambda_"):
        np.testing.assert_array_equal(getattr(ra, k), getattr(rb, k))


def test_c5_full_size_is_bit_identical_and_uses_helpers():
    cfg, obst, via, batch = scenes.scene_c5(stride=320)
    one, r1, info1, f1, ms1 = _run(cfg, obst, via, batch, multi_cu=-1)
    many, rm, infom, fm, msm = _run(cfg, obst, via, batch)        # automatic: one band, 300 polygons x 300 poses
    assert info1 == (0, False)
    assert infom[0] >= 16 and not infom[1], infom
    assert not f1.any() and not fm.any()
    _assert_identical(many, rm, one, r1)
    print("C5: one CU %.3f ms, %d helpers %.3f ms" % (ms1, infom[0], msm))
    assert msm < ms1


@pytest.mark.parametrize("footprint", ["point", "circular", "two_circles", "line", "polygon"])
@pytest.mark.parametrize("helpers", [2, 7, 40])
def test_small_mixed_scenes_every_footprint(footprint, helpers):
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint=footprint)   # every obstacle type, dynamic obstacles, via-points; 3 bands
    one, r1, info1, _, _ = _run(cfg, obst, via, batch, multi_cu=-1, generic_distance_path=True)
    many, rm, infom, fm, _ = _run(cfg, obst, via, batch, multi_cu=helpers, generic_distance_path=True)
    assert info1[0] == 0 and infom == (helpers, False), (info1, infom)
    assert not fm.any()
    _assert_identical(many, rm, one, r1)


def test_velocity_obstacle_ratio_edges_read_the_records():
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint="polygon")
    cfg.optim.weight_velocity_obstacle_ratio = 3.0
    one, r1, _, _, _ = _run(cfg, obst, via, batch, multi_cu=-1)
    many, rm, infom, _, _ = _run(cfg, obst, via, batch, multi_cu=9)
    assert infom == (9, False)
    _assert_identical(many, rm, one, r1)


def test_divergence_detection_and_cost_exponent():
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint="two_circles")
    cfg.recovery.divergence_detection_enable = True
    cfg.optim.obstacle_cost_exponent = 1.5
    one, r1, _, _, _ = _run(cfg, obst, via, batch, multi_cu=-1)
    many, rm, infom, _, _ = _run(cfg, obst, via, batch, multi_cu=5)
    assert infom == (5, False)
    _assert_identical(many, rm, one, r1)


def test_long_band_beyond_the_workgroup():
    """more poses than the 256 lanes of a workgroup: the second (sliced) pass of the single-CU path against helper tiles"""
    cfg, obst, via, batch = scenes.scene_c5(n=300, M=100, stride=400)
    one, r1, _, f1, _ = _run(cfg, obst, via, batch, multi_cu=-1)
    many, rm, infom, fm, _ = _run(cfg, obst, via, batch, multi_cu=32)
    assert infom == (32, False) and not f1.any() and not fm.any()
    assert one.n[0] > 256
    _assert_identical(many, rm, one, r1)


def test_helpers_that_do_not_arrive_in_time_mean_a_repeat_on_one_cu():
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint="polygon")
    one, r1, _, _, _ = _run(cfg, obst, via, batch, multi_cu=-1)
    many, rm, infom, fm, _ = _run(cfg, obst, via, batch, multi_cu=6, multi_cu_timeout_us=1)   # 1 us: no phase can make it
    assert infom == (6, True), infom
    assert not fm.any()
    _assert_identical(many, rm, one, r1)


def test_numeric_mode_and_legacy_association_stay_on_one_cu():
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint="polygon")
    cfg.jacobian_mode = _abi.JACOBIAN_G2O_NUMERIC
    _, _, info, _, _ = _run(cfg, obst, via, batch, multi_cu=8)
    assert info == (0, False)
    cfg.jacobian_mode = _abi.JACOBIAN_ANALYTIC
    cfg.obstacles.legacy_obstacle_association = True
    _, _, info, _, _ = _run(cfg, obst, via, batch, multi_cu=8)
    assert info == (0, False)


def test_matches_the_oracle(oracle):
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint="polygon")
    many, rm, infom, _, _ = _run(cfg, obst, via, batch, multi_cu=12)
    ref, rres = oracle.optimize_batch(cfg, obst, via, batch)
    assert infom == (12, False)
    np.testing.assert_array_equal(many.n, ref.n)
    np.testing.assert_array_equal(rm.lm_trials, rres.lm_trials)
    for b in range(batch.count):
        for u, v in zip(many.get_teb(b), ref.get_teb(b)):
            assert np.abs(u - v).max() < 1e-7
    np.testing.assert_allclose(rm.cost, rres.cost, rtol=1e-8)
