"""Multi-CU mode (teb_local_planner_amd/csrc/teb_multicu.hpp): small batches use the CUs they leave idle.
  * distance helpers (generic-shape scenes): the obstacle association and the robot <-> obstacle distances of every (pose, obstacle) pair;
  * solver helpers (any scene): the damped systems of the first retries of an LM iteration, solved speculatively while the band
    evaluates its first trial.
The contract: the bands are BIT-IDENTICAL to the single-CU launch (same device functions on the same inputs, rows accumulated in the
same order, the same solve routine on the same system) - for every footprint kind, with dynamic obstacles, via-points, the
velocity-obstacle-ratio edges, any number of helpers, bands longer than the workgroup, every layout of the normal matrix; and when
the distance helpers do not arrive in time the launch is repeated on one CU per band with the same result."""
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
    one, r1, info1, f1, ms1 = _run(cfg, obst, via, batch, multi_cu=-1, speculative_trials=-1)
    many, rm, infom, fm, msm = _run(cfg, obst, via, batch)        # automatic: one band, 300 polygons x 300 poses
    assert info1 == (0, 0, False)
    assert infom[0] >= 16 and infom[1] == 3 and not infom[2], infom
    assert not f1.any() and not fm.any()
    _assert_identical(many, rm, one, r1)
    print("C5: one CU %.3f ms, %d distance + %d solver helpers %.3f ms" % (ms1, infom[0], infom[1], msm))
    assert msm < ms1


@pytest.mark.parametrize("footprint", ["point", "circular", "two_circles", "line", "polygon"])
@pytest.mark.parametrize("helpers", [2, 7, 40])
def test_small_mixed_scenes_every_footprint(footprint, helpers):
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint=footprint)   # every obstacle type, dynamic obstacles, via-points; 3 bands
    one, r1, info1, _, _ = _run(cfg, obst, via, batch, multi_cu=-1, speculative_trials=-1, generic_distance_path=True)
    many, rm, infom, fm, _ = _run(cfg, obst, via, batch, multi_cu=helpers, speculative_trials=-1, generic_distance_path=True)
    granted = min(helpers, max(2, batch.x.shape[1] // 6))   # at most one distance helper per 6 poses of the capacity (teb_amd.hip: mcu_helpers_for)
    assert info1 == (0, 0, False) and infom == (granted, 0, False), (info1, infom)
    assert not fm.any()
    _assert_identical(many, rm, one, r1)


def test_velocity_obstacle_ratio_edges_read_the_records():
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint="polygon")
    cfg.optim.weight_velocity_obstacle_ratio = 3.0
    one, r1, _, _, _ = _run(cfg, obst, via, batch, multi_cu=-1, speculative_trials=-1)
    many, rm, infom, _, _ = _run(cfg, obst, via, batch, multi_cu=9)
    assert infom == (9, 3, False)
    _assert_identical(many, rm, one, r1)


def test_divergence_detection_and_cost_exponent():
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint="two_circles")
    cfg.recovery.divergence_detection_enable = True
    cfg.optim.obstacle_cost_exponent = 1.5
    one, r1, _, _, _ = _run(cfg, obst, via, batch, multi_cu=-1, speculative_trials=-1)
    many, rm, infom, _, _ = _run(cfg, obst, via, batch, multi_cu=5, speculative_trials=2)
    assert infom == (5, 2, False)
    _assert_identical(many, rm, one, r1)


def test_long_band_beyond_the_workgroup():
    """more poses than the 256 lanes of a workgroup: the second (sliced) pass of the single-CU path against helper tiles"""
    cfg, obst, via, batch = scenes.scene_c5(n=300, M=100, stride=400)
    one, r1, _, f1, _ = _run(cfg, obst, via, batch, multi_cu=-1, speculative_trials=-1)
    many, rm, infom, fm, _ = _run(cfg, obst, via, batch, multi_cu=32)
    assert infom == (32, 3, False) and not f1.any() and not fm.any()
    assert one.n[0] > 256
    _assert_identical(many, rm, one, r1)


def test_helpers_that_do_not_arrive_in_time_mean_a_repeat_on_one_cu():
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint="polygon")
    one, r1, _, _, _ = _run(cfg, obst, via, batch, multi_cu=-1, speculative_trials=-1)
    many, rm, infom, fm, _ = _run(cfg, obst, via, batch, multi_cu=6, multi_cu_timeout_us=1)   # 1 us: no phase can make it
    assert infom == (6, 3, True), infom
    assert not fm.any()
    _assert_identical(many, rm, one, r1)


def test_numeric_mode_and_legacy_association_stay_on_one_cu():
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint="polygon")
    cfg.jacobian_mode = _abi.JACOBIAN_G2O_NUMERIC
    _, _, info, _, _ = _run(cfg, obst, via, batch, multi_cu=8)
    assert info == (0, 0, False)
    cfg.jacobian_mode = _abi.JACOBIAN_ANALYTIC
    cfg.obstacles.legacy_obstacle_association = True
    one, r1, info1, _, _ = _run(cfg, obst, via, batch, multi_cu=-1, speculative_trials=-1)
    many, rm, info, _, _ = _run(cfg, obst, via, batch, multi_cu=8)
    assert info == (0, 3, False)           # the legacy association stays with the band, the speculative trials do not depend on it
    _assert_identical(many, rm, one, r1)


def test_matches_the_oracle(oracle):
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint="polygon")
    many, rm, infom, _, _ = _run(cfg, obst, via, batch, multi_cu=12)
    ref, rres = oracle.optimize_batch(cfg, obst, via, batch)
    assert infom == (12, 3, False)
    np.testing.assert_array_equal(many.n, ref.n)
    np.testing.assert_array_equal(rm.lm_trials, rres.lm_trials)
    for b in range(batch.count):
        for u, v in zip(many.get_teb(b), ref.get_teb(b)):
            assert np.abs(u - v).max() < 1e-7
    np.testing.assert_allclose(rm.cost, rres.cost, rtol=1e-8)


# ---- speculative LM trials (solver helpers), point-like scenes included ---------------------------------------------------------------
@pytest.mark.parametrize("k", [1, 2, 3])
@pytest.mark.parametrize("layout", ["cr", "band"])
def test_speculative_trials_every_layout(layout, k):
    """C2-like single band in the blocks-in-LDS layout (in-place solve, H restored lazily) and in the hybrid layout (band copy in HBM)."""
    cfg, obst, via, batch = scenes.scene_c2(n=120, M=60, stride=208 if layout == "cr" else 288)
    one, r1, info1, _, _ = _run(cfg, obst, via, batch, speculative_trials=-1, layout=layout)
    many, rm, infom, _, _ = _run(cfg, obst, via, batch, speculative_trials=k, layout=layout)
    assert info1 == (0, 0, False) and infom == (0, k, False), (info1, infom)
    assert r1.lm_trials[0] > r1.lm_iterations[0]          # there are rejected trials to speculate on
    _assert_identical(many, rm, one, r1)


def test_c2_full_size_is_bit_identical_and_faster():
    cfg, obst, via, batch = scenes.scene_c2(stride=208)
    one, r1, info1, _, ms1 = _run(cfg, obst, via, batch, speculative_trials=-1)
    many, rm, infom, _, msm = _run(cfg, obst, via, batch)          # automatic: one band -> three solver helpers
    assert info1 == (0, 0, False) and infom == (0, 3, False), (info1, infom)
    _assert_identical(many, rm, one, r1)
    print("C2: one CU %.3f ms, with 3 solver helpers %.3f ms (%d trials for %d iterations)" % (ms1, msm, r1.lm_trials[0], r1.lm_iterations[0]))
    assert msm < ms1


def test_small_batch_of_point_like_bands_with_dynamic_obstacles():
    cfg, obst, via, batch = scenes.scene_c4(B=6, n=150, stride=288)       # hybrid layout, LDS obstacle cache, autoResize, dynamic obstacles
    one, r1, _, _, _ = _run(cfg, obst, via, batch, speculative_trials=-1)
    many, rm, infom, _, _ = _run(cfg, obst, via, batch)
    assert infom == (0, 3, False)
    _assert_identical(many, rm, one, r1)


def test_helper_count_follows_the_room_the_batch_leaves():
    """256 CUs, one workgroup per CU: B (1 + K + D) <= 256. 64 bands: all three retries; 100 bands: the first one; 200 bands: none."""
    for B, want in ((64, 3), (100, 1), (200, 0)):
        cfg, obst, via, batch = scenes.scene_c3(B=B, n=60, M=40, stride=208)
        one, r1, info1, _, _ = _run(cfg, obst, via, batch, speculative_trials=-1)
        many, rm, info, _, _ = _run(cfg, obst, via, batch)
        assert info1 == (0, 0, False) and info == (0, want, False), (B, info)
        _assert_identical(many, rm, one, r1)


def test_late_helpers_are_backed_off_and_probed_again():
    """a launch whose distance helpers came late is repeated on one CU per band; the next 4 launches then run on one
    CU per band without asking (no wait, no repeat), the 6th probes again, a second miss doubles the pause. Bands identical throughout."""
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint="polygon")
    one, r1, _, _, _ = _run(cfg, obst, via, batch, multi_cu=-1, speculative_trials=-1)
    s = planner.make_solver(cfg, obst, via, batch, options=_abi.Options(multi_cu=6, multi_cu_timeout_us=1))
    s.snapshot()
    seen = []
    for tick in range(11):
        s.restore()
        s.optimize(cfg.optim.no_inner_iterations, cfg.optim.no_outer_iterations, True, cfg.hcp.selection_obst_cost_scale,
                   cfg.hcp.selection_viapoint_cost_scale, cfg.hcp.selection_alternative_time_cost)
        seen.append((s.last_launch_info(), s.multi_cu_backoff()))
        _assert_identical(s.download(batch.copy()), s.results(), one, r1)
    s.close()
    assert seen[0] == ((6, 3, True), (4, 4)), seen          # miss -> pause of 4
    for k in range(1, 5):
        assert seen[k] == ((0, 3, False), (4 - k, 4)), seen  # paused: no distance helpers, no repeat
    assert seen[5] == ((6, 3, True), (8, 8)), seen           # probe, missed again -> pause of 8
    for k in range(6, 11):
        assert seen[k][0] == (0, 3, False), seen


def test_a_successful_probe_clears_the_pause():
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint="polygon")
    s = planner.make_solver(cfg, obst, via, batch, options=_abi.Options(multi_cu=6))
    s.snapshot()
    s.optimize(cfg.optim.no_inner_iterations, cfg.optim.no_outer_iterations, True, 1.0, 1.0, False)
    assert s.last_launch_info() == (6, 3, False) and s.multi_cu_backoff() == (0, 0)
    s.close()


def test_ticks_on_a_busy_device_fall_back_to_one_cu_per_band_without_paying_for_it_every_tick():
    """C5 ticks while another stream keeps the device busy (a queue of large fp32 GEMMs on all CUs). Whatever the
    scheduler does with the helper workgroups - arrive in time or not -, every tick must give the one-CU bands bit for bit, and once a miss
    has started the back-off the paused ticks neither wait nor repeat: each costs at most the one-CU tick under the same load + 10 %."""
    import time
    import torch
    if not torch.cuda.is_available():   # (the GEMM queue is torch's; the library's own back-off tests above need no load generator)
        pytest.skip("torch sees no GPU in this process: no load generator")
    cfg, obst, via, batch = scenes.scene_c5(stride=320)
    one, r1, _, _, _ = _run(cfg, obst, via, batch, multi_cu=-1, speculative_trials=-1)
    a = torch.randn(8192, 8192, device="cuda"); b = torch.randn(8192, 8192, device="cuda")
    side = torch.cuda.Stream()

    def load(k):
        with torch.cuda.stream(side):
            for _ in range(k):
                torch.mm(a, b)

    def ticks(s, count):
        wall, info = [], []
        for _ in range(count):
            load(6)                                   # ~ 6 x 10 ms of GEMM queued beside every tick
            s.restore()
            t0 = time.perf_counter()
            s.optimize(cfg.optim.no_inner_iterations, cfg.optim.no_outer_iterations, True, cfg.hcp.selection_obst_cost_scale,
                       cfg.hcp.selection_viapoint_cost_scale, cfg.hcp.selection_alternative_time_cost)
            s.synchronize()
            wall.append(time.perf_counter() - t0)
            info.append((s.last_launch_info(), s.multi_cu_backoff()))
            _assert_identical(s.download(batch.copy()), s.results(), one, r1)
        return wall, info

    s1 = planner.make_solver(cfg, obst, via, batch, options=_abi.Options(multi_cu=-1, speculative_trials=-1))
    s1.snapshot()
    w1, _ = ticks(s1, 6)
    s1.close()
    sm = planner.make_solver(cfg, obst, via, batch)
    sm.snapshot()
    wm, im = ticks(sm, 12)
    sm.close()
    torch.cuda.synchronize()
    one_cu = float(np.median(w1))
    paused = [w for w, (li, bo) in zip(wm, im) if li[0] == 0]          # ticks inside a pause: no distance helpers asked for
    missed = [w for w, (li, bo) in zip(wm, im) if li[2]]
    print("busy device: one CU per band %.2f ms; multi-CU ticks %s ms; %d missed, %d paused" % (
        1e3 * one_cu, ["%.2f" % (1e3 * w) for w in wm], len(missed), len(paused)))
    for w in paused:
        assert w <= 1.10 * one_cu + 1e-3, (w, one_cu)
    if missed:
        assert paused, "a miss must start a pause"
