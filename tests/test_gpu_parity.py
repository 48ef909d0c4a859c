"""Parity of the HIP path (through the C-ABI of libteb_amd.so) against the CPU oracle and the golden
fixtures. All fp64. Stated tolerances (tightened to what is actually observed):

  T0/T1  one linearisation at the same state:  chi^2 per category rel 1e-12; H, b rel 1e-12 of max|.|
         (both sides use closed-form Jacobians; only the summation order and libm ulps differ)
         association lists: bit-exact (integer work)
  T3     full optimizeTEB (4 x 5 LM iterations incl. autoResize / association / cost):
         identical pose count, status, LM iteration and trial counts; poses <= 1e-8 m / rad; chi^2 and cost rel 1e-8
         vs the vectors of the reference's own code (either Jacobian mode): per-band tolerance of tests/sensitivity.py,
         2e-5 m / rad / s on bands that damp the reference's central-difference noise (delta = 1e-9)
  T4     selectBestTeb: same index
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402

from teb_local_planner_amd import scenes, planner, _abi  # noqa: E402
from test_golden_oracle import check_against_golden  # noqa: E402

pytestmark = pytest.mark.gpu


def run_gpu(cfg, obst, via, batch, inner=None, outer=None, compute_cost=True, options=None):
    inner = cfg.optim.no_inner_iterations if inner is None else inner
    outer = cfg.optim.no_outer_iterations if outer is None else outer
    s = planner.make_solver(cfg, obst, via, batch, options=options)
    s.optimize(inner, outer, compute_cost, cfg.hcp.selection_obst_cost_scale, cfg.hcp.selection_viapoint_cost_scale,
               cfg.hcp.selection_alternative_time_cost)
    res = s.results()
    out = s.download(batch.copy())
    best = s.select_best()
    flags = s.debug_overflow_flags()
    s.close()
    assert not flags.any()
    return out, res, best


def assert_full_parity(out, res, ref, rres, pos_tol=1e-8, rtol=1e-8):
    np.testing.assert_array_equal(out.n, ref.n)
    np.testing.assert_array_equal(res.status, rres.status)
    np.testing.assert_array_equal(res.lm_iterations, rres.lm_iterations)
    np.testing.assert_array_equal(res.lm_trials, rres.lm_trials)
    np.testing.assert_allclose(res.chi2, rres.chi2, rtol=rtol)
    np.testing.assert_allclose(res.cost, rres.cost, rtol=rtol)
    for b in range(out.count):
        for u, v in zip(out.get_teb(b), ref.get_teb(b)):
            assert np.abs(u - v).max() <= pos_tol


# ---- K10: distances -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("footprint", ["point", "circular", "two_circles", "line", "polygon"])
def test_distance_library(oracle, footprint):
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint=footprint)
    s = planner.make_solver(cfg, obst, via, batch)
    rng = np.random.default_rng(11)
    nq = 400
    oi = rng.integers(0, len(obst), nq); x = rng.uniform(0, 6, nq); y = rng.uniform(-2, 2, nq)
    th = rng.uniform(-3.1, 3.1, nq); t = rng.uniform(0, 6, nq)
    for tt in (None, t):
        d, g = s.debug_distance(oi, x, y, th, tt)
        for q in range(nq):
            do, go = oracle.distance(cfg, obst, int(oi[q]), x[q], y[q], th[q], None if tt is None else tt[q])
            assert abs(d[q] - do) <= 1e-14 * max(1.0, abs(do))
            assert np.abs(g[q] - go).max() <= 1e-12
    s.close()


# ---- T0/T1 + association --------------------------------------------------------------------------------------
@pytest.mark.parametrize("footprint", ["point", "circular", "two_circles", "line", "polygon"])
def test_linearisation_and_association(oracle, footprint):
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint=footprint)
    s = planner.make_solver(cfg, obst, via, batch)
    for b in range(batch.count):
        for wm in (1.0, 8.0):
            G = s.debug_linearize(b, int(batch.n[b]), wm)
            R = oracle.linearize(cfg, obst, via, batch, b, wm)
            np.testing.assert_allclose(G["chi2"], R["chi2"], rtol=1e-12, atol=1e-14)
            assert np.abs(G["H"] - R["H"]).max() <= 1e-12 * np.abs(R["H"]).max()
            assert np.abs(G["b"] - R["b"]).max() <= 1e-12 * np.abs(R["b"]).max()
            ap, ao = oracle.associate(cfg, obst, batch, b)
            np.testing.assert_array_equal(G["assoc_pose"], ap)
            np.testing.assert_array_equal(G["assoc_obst"], ao)
    s.close()


def _variant(name):
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint="circular")
    if name == "carlike":
        cfg.robot.min_turning_radius = 0.8
    elif name == "carlike_exact_arc":
        cfg.robot.min_turning_radius = 0.8
        cfg.trajectory.exact_arc_length = True
    elif name == "exact_arc":
        cfg.trajectory.exact_arc_length = True
    elif name == "optional_edges":
        cfg.optim.weight_shortest_path = 0.7
        cfg.optim.weight_velocity_obstacle_ratio = 3.0
        cfg.obstacles.obstacle_proximity_lower_bound = 0.1
        cfg.obstacles.obstacle_proximity_upper_bound = 1.2
        cfg.obstacles.obstacle_proximity_ratio_max_vel = 0.8
        cfg.optim.obstacle_cost_exponent = 1.7
        batch.prefer_rotdir[0] = _abi.ROT_LEFT
        batch.prefer_rotdir[1] = _abi.ROT_RIGHT
    elif name == "holonomic":
        cfg.robot.max_vel_y = 0.3
        cfg.robot.max_vel_trans = 0.45
        cfg.robot.acc_lim_y = 0.4
        rng = np.random.default_rng(3)
        batch.theta += rng.uniform(-0.4, 0.4, batch.theta.shape)
        batch.dt *= 0.6
    elif name == "static_only":
        cfg.obstacles.include_dynamic_obstacles = False
    elif name == "no_inflation":
        cfg.obstacles.inflation_dist = 0.4
    elif name == "ordered_via_alt_time":
        cfg.trajectory.via_points_ordered = True
        cfg.hcp.selection_alternative_time_cost = True
        batch.via_points_enabled[2] = 0
    elif name == "divergence_stats":
        cfg.recovery.divergence_detection_enable = True
    return cfg, obst, via, batch


VARIANTS = ["carlike", "carlike_exact_arc", "exact_arc", "optional_edges", "holonomic", "static_only",
            "no_inflation", "ordered_via_alt_time", "divergence_stats"]


@pytest.mark.parametrize("name", VARIANTS)
def test_edge_family_variants_linearise_and_optimise(oracle, name):
    cfg, obst, via, batch = _variant(name)
    s = planner.make_solver(cfg, obst, via, batch)
    for b in range(batch.count):
        G = s.debug_linearize(b, int(batch.n[b]), 2.0)
        R = oracle.linearize(cfg, obst, via, batch, b, 2.0)
        np.testing.assert_allclose(G["chi2"], R["chi2"], rtol=1e-12, atol=1e-14)
        assert np.abs(G["H"] - R["H"]).max() <= 1e-12 * np.abs(R["H"]).max()
        assert np.abs(G["b"] - R["b"]).max() <= 1e-12 * np.abs(R["b"]).max()
    s.close()
    out, res, best = run_gpu(cfg, obst, via, batch)
    ref, rres = oracle.optimize_batch(cfg, obst, via, batch)
    assert_full_parity(out, res, ref, rres)
    assert best[0] == oracle.select_best(cfg, rres.cost)[0]


# ---- T3: full optimizeTEB against the golden fixtures and the live oracle ------------------------------------------
@pytest.mark.parametrize("name", sorted(make_golden.CASES))
def test_full_optimize_matches_golden(name):
    cfg, obst, via, batch = make_golden.CASES[name]()
    out, res, _ = run_gpu(cfg, obst, via, batch)
    check_against_golden(name, out, res, pos_tol=1e-8, cost_rtol=1e-8)


@pytest.mark.parametrize("name", ["c1_test_optim_node", "mixed_polygon", "c2_small"])
def test_full_optimize_within_reference_noise_of_faithful_oracle(oracle, name):
    cfg, obst, via, batch = make_golden.CASES[name]()
    out, res, _ = run_gpu(cfg, obst, via, batch)
    cfg.jacobian_mode = _abi.JACOBIAN_G2O_NUMERIC
    ref, rres = oracle.optimize_batch(cfg, obst, via, batch)
    np.testing.assert_array_equal(out.n, ref.n)
    np.testing.assert_array_equal(res.status, rres.status)
    np.testing.assert_allclose(res.cost, rres.cost, rtol=1e-3)
    for b in range(out.count):
        for u, v in zip(out.get_teb(b), ref.get_teb(b)):
            assert np.abs(u - v).max() <= 1e-3


def test_iteration_schedules_and_guards(oracle):
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint="point")
    for inner, outer in ((1, 1), (3, 2), (5, 1), (0, 1), (2, 0)):
        out, res, _ = run_gpu(cfg, obst, via, batch, inner, outer)
        ref, rres = oracle.optimize_batch(cfg, obst, via, batch, inner=inner, outer=outer)
        assert_full_parity(out, res, ref, rres)
    # guards of optimizeGraph (optimal_planner.cpp:370-382): too few samples / robot too slow / deactivated
    for mutate in ("min_samples", "max_vel_x", "deactivate"):
        cfg, obst, via, batch = scenes.scene_small_mixed(footprint="point")
        cfg.trajectory.teb_autosize = False
        if mutate == "min_samples":
            cfg.trajectory.min_samples = 30
        elif mutate == "max_vel_x":
            cfg.robot.max_vel_x = 0.005
        else:
            cfg.optim.optimization_activate = False
        out, res, _ = run_gpu(cfg, obst, via, batch)
        ref, rres = oracle.optimize_batch(cfg, obst, via, batch)
        assert (res.status == _abi.TEB_FAILED).all() and (rres.status == _abi.TEB_FAILED).all()
        np.testing.assert_array_equal(out.x, batch.x)   # state untouched


def test_autoresize_on_device_reproduces_sequential_semantics(oracle):
    """Bands with wildly uneven time differences: splits (recursive), excess shifting, merges, last-interval merge."""
    cfg, obst, via, _ = scenes.scene_small_mixed(footprint="point", with_dynamic=False)
    rng = np.random.default_rng(21)
    B, S = 6, 200
    batch = _abi.TebBatchHost(B, S)
    for b in range(B):
        n = int(rng.integers(5, 40))
        x = np.sort(rng.uniform(0, 6, n)); y = rng.uniform(-0.5, 0.5, n); th = rng.uniform(-1, 1, n)
        dt = rng.choice([0.02, 0.1, 0.3, 0.5, 0.9, 2.5], n - 1) * rng.uniform(0.8, 1.2, n - 1)
        batch.set_teb(b, x, y, th, dt)
    for fast in (True, False):
        cfg.obstacles.include_dynamic_obstacles = not fast      # fast_mode = !include_dynamic_obstacles
        out, res, _ = run_gpu(cfg, obst, via, batch, inner=1, outer=1, compute_cost=False)
        ref, rres = oracle.optimize_batch(cfg, obst, via, batch, inner=1, outer=1, compute_cost=False)
        np.testing.assert_array_equal(out.n, ref.n)
        assert_full_parity(out, res, ref, rres)


def test_autoresize_chains_and_their_fallbacks_on_device(oracle):
    """The sweep as parallel chains (csrc/teb_autoresize_chain.hpp) on long bands - both slots of a lane in use beyond 256 intervals -
    and the sweeps it has to hand to the sequential machine: a sample-count guard that may bind (max_samples / min_samples close to
    the band's size) and a chain of more than 64 rule evaluations (an excess handed on from interval to interval)."""
    cfg, obst, via, _ = scenes.scene_small_mixed(footprint="point", with_dynamic=False)
    rng = np.random.default_rng(33)
    S = 512

    def band(n, dt):
        x = np.cumsum(rng.uniform(0.02, 0.06, n)); y = rng.normal(0, 0.05, n); th = rng.uniform(-1, 1, n)
        return x, y, th, dt

    cases = []
    for n in (150, 230, 257, 300, 340):            # typical: most intervals near the dead band, a few far out
        dt = rng.normal(0.3, 0.08, n - 1).clip(0.01, None)
        dt[rng.integers(0, n - 1, 6)] = rng.choice([0.02, 0.9, 1.7], 6)
        cases.append(band(n, dt))
    n = 200
    dt = np.full(n - 1, 0.3); dt[0] = 0.41          # one chain over the whole band: the member gives up
    cases.append(band(n, dt))
    dt = np.full(n - 1, 0.3); dt[0] = 0.41; dt[50] = 0.25; dt[120] = 0.05; dt[121] = 1.3   # chains of 51 and ~70 evaluations
    cases.append(band(n, dt))
    batch = _abi.TebBatchHost(len(cases), S)
    for b, cse in enumerate(cases):
        batch.set_teb(b, *cse)
    for max_s, min_s in ((500, 3), (262, 3), (500, 228), (345, 150)):
        cfg.trajectory.max_samples, cfg.trajectory.min_samples = max_s, min_s
        for fast in (True, False):
            cfg.obstacles.include_dynamic_obstacles = not fast      # fast_mode = !include_dynamic_obstacles
            out, res, _ = run_gpu(cfg, obst, via, batch, inner=1, outer=1, compute_cost=False)
            ref, rres = oracle.optimize_batch(cfg, obst, via, batch, inner=1, outer=1, compute_cost=False, threads=8)
            np.testing.assert_array_equal(out.n, ref.n)
            np.testing.assert_array_equal(res.status, rres.status)
            for b in range(batch.count):
                if rres.status[b] != _abi.TEB_OK:
                    continue
                for u, v in zip(out.get_teb(b), ref.get_teb(b)):
                    assert np.abs(u - v).max() < 1e-6, (max_s, min_s, fast, b, np.abs(u - v).max())


# ---- T4 + BASELINE configs at full size ------------------------------------------------------------------------------
def test_c3_batch_selection_matches_oracle(oracle):
    cfg, obst, via, batch = scenes.scene_c3(B=16, n=150, M=200, stride=192)
    out, res, best = run_gpu(cfg, obst, via, batch)
    ref, rres = oracle.optimize_batch(cfg, obst, via, batch, threads=os.cpu_count() or 1)
    assert_full_parity(out, res, ref, rres, pos_tol=1e-7, rtol=1e-7)
    assert best[0] == oracle.select_best(cfg, rres.cost)[0]
    # hysteresis / prefer-initial multipliers
    s = planner.make_solver(cfg, obst, via, out)
    s.optimize(1, 1, True, 100.0, 1.0, False)
    r2 = s.results()
    for lb, ip in ((3, -1), (-1, 5), (2, 2)):
        assert s.select_best(lb, ip)[0] == oracle.select_best(cfg, r2.cost, lb, ip)[0]
    s.close()


def test_c4_full_size_properties():
    """BASELINE config C4 at full size (256 x 200 poses, 450 static + 50 dynamic obstacles): too slow for the
    oracle, so check size-independent properties: every TEB OK, 20 LM iterations each, chi^2 finite and not worse
    than the initial one, fixed start/goal untouched, re-running is bit-reproducible, and a sample of TEBs matches
    the oracle run on exactly those TEBs."""
    cfg, obst, via, batch = scenes.scene_c4()
    cfg.trajectory.teb_autosize = False
    s = planner.make_solver(cfg, obst, via, batch)
    chi0 = np.array([s.debug_linearize(b, 200, 1.0)["chi2"].sum() for b in (0, 17, 255)])
    s.optimize(5, 4, True, 100.0, 1.0, False)
    res = s.results()
    out = s.download(batch.copy())
    assert (res.status == _abi.TEB_OK).all()
    assert (res.lm_iterations == 20).all() and (out.n == 200).all()
    assert np.isfinite(res.chi2).all() and np.isfinite(res.cost).all()
    np.testing.assert_array_equal(out.x[:, 0], batch.x[:, 0]); np.testing.assert_array_equal(out.x[:, 199], batch.x[:, 199])
    np.testing.assert_array_equal(out.theta[:, 0], batch.theta[:, 0])
    s.upload(batch)
    s.optimize(5, 4, True, 100.0, 1.0, False)
    res2 = s.results()
    out2 = s.download(batch.copy())
    np.testing.assert_array_equal(out.x, out2.x); np.testing.assert_array_equal(res.cost, res2.cost)   # deterministic
    s.close()
    from oracle import oracle_py
    sub = _abi.TebBatchHost(3, 200)
    for k, b in enumerate((0, 17, 255)):
        sub.set_teb(k, *batch.get_teb(b))
        sub.has_vel_goal[k] = batch.has_vel_goal[b]
    ref, rres = oracle_py.optimize_batch(cfg, obst, via, sub)
    for k, b in enumerate((0, 17, 255)):
        assert res.lm_trials[b] == rres.lm_trials[k]
        np.testing.assert_allclose(res.cost[b], rres.cost[k], rtol=1e-7)
        for u, v in zip(out.get_teb(b), ref.get_teb(k)):
            assert np.abs(u - v).max() <= 1e-7


def test_c5_carlike_polygon_full_size(oracle):
    cfg, obst, via, batch = scenes.scene_c5(stride=320)
    out, res, _ = run_gpu(cfg, obst, via, batch)
    ref, rres = oracle.optimize_batch(cfg, obst, via, batch)
    assert_full_parity(out, res, ref, rres, pos_tol=1e-7, rtol=1e-7)


def test_capacity_errors_are_loud():
    cfg, obst, via, batch = scenes.scene_c1()
    with pytest.raises(planner.TebAmdError) as e:
        planner.TebBatchSolver(cfg, 1, 1100, 4, 1, 1)             # (600 until round 5: the band-in-HBM layout now holds ~ 950 poses)
    assert e.value.code == _abi.ERR_CAPACITY
    s = planner.make_solver(cfg, obst, via, batch)
    big = _abi.ObstacleTable()
    for k in range(10):
        big.add_point(k, 0)
    with pytest.raises(planner.TebAmdError):
        s.set_obstacles(big)
    s.close()


def test_reference_style_planner_objects(oracle):
    """TebOptimalPlanner.optimizeTEB / getCurrentCost mirror (optimal_planner.h:231-232, 437)."""
    cfg, obst, via, batch = scenes.scene_c1()
    p = planner.TebOptimalPlanner(cfg, obst, via, max_poses=128)
    p.teb().set_teb(0, *batch.get_teb(0))
    p.teb().has_vel_goal[0] = 1
    ok = p.optimizeTEB(cfg.optim.no_inner_iterations, cfg.optim.no_outer_iterations, True,
                       cfg.hcp.selection_obst_cost_scale, cfg.hcp.selection_viapoint_cost_scale, False)
    assert ok and p.isOptimized()
    ref, rres = oracle.optimize_batch(cfg, obst, via, batch)
    assert abs(p.getCurrentCost() - rres.cost[0]) <= 1e-8 * rres.cost[0]
    assert p.teb().n[0] == ref.n[0]
    cfg.optim.optimization_activate = False
    assert p.optimizeTEB(5, 4) is False


@pytest.mark.parametrize("solver", ["cr", "band", "band_ldlt", "bandg"])
def test_both_damped_solvers_match_the_oracle(oracle, solver):
    """The three damped solves - block cyclic reduction on the LDS-resident blocks ("cr"), the same reduction on HBM-resident
    blocks expanded from the LDS band (what long bands use: "band"), and the sequential banded LDL^T ("band_ldlt", kept as a
    cross-check) - solve the same system: identical accept / reject decisions, trajectories within 1e-8. Even and odd pose
    counts (block padding)."""
    opt = _abi.Options(layout="bandg" if solver == "bandg" else ("band" if solver.startswith("band") else "cr"),   # "bandg": band in HBM
                       band_ldlt=(solver == "band_ldlt"))
    for n0 in (24, 25):
        cfg, obst, via, batch = scenes.scene_small_mixed(footprint="polygon", n=n0)
        for autosize in (True, False):
            cfg.trajectory.teb_autosize = autosize
            out, res, best = run_gpu(cfg, obst, via, batch, options=opt)
            ref, rres = oracle.optimize_batch(cfg, obst, via, batch)
            assert_full_parity(out, res, ref, rres)


def test_solver_fails_like_cholesky_on_indefinite_system(oracle):
    """Negative weights make H indefinite: CSparse's cholesky (and both GPU solvers) must report failure,
    the LM loop then rejects every trial and terminates after 10 of them."""
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint="point", with_dynamic=False)
    cfg.trajectory.teb_autosize = False
    cfg.optim.weight_kinematics_nh = -1000.0
    out, res, _ = run_gpu(cfg, obst, via, batch, inner=2, outer=1)
    ref, rres = oracle.optimize_batch(cfg, obst, via, batch, inner=2, outer=1)
    assert (rres.lm_trials > rres.lm_iterations).all()      # there were failed factorisations / rejected trials
    assert_full_parity(out, res, ref, rres, pos_tol=1e-7, rtol=1e-7)


@pytest.mark.parametrize("fast", [True, False])
@pytest.mark.parametrize("footprint", ["point", "circular"])
def test_pointlike_fast_path_equals_generic_path(oracle, fast, footprint):
    """Point/Circular obstacles + Point/Circular footprint take the LDS-resident specialised distance path;
    teb_amd_options_t::generic_distance_path forces the generic one. Both must match the oracle (incl. the velocity-obstacle-ratio edge)."""
    opt = None if fast else _abi.Options(generic_distance_path=True)
    cfg, obst0, via, batch = scenes.scene_small_mixed(footprint=footprint, stride=192)
    obst = _abi.ObstacleTable()
    rng = np.random.default_rng(4)
    for k in range(30):
        p = (rng.uniform(0.5, 5.5), rng.uniform(-1.5, 1.5))
        vel = (rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2)) if k % 5 == 0 else None
        if k % 3 == 0:
            obst.add_circle(p[0], p[1], rng.uniform(0.05, 0.3), vel=vel)
        else:
            obst.add_point(p[0], p[1], vel=vel)
    cfg.optim.weight_velocity_obstacle_ratio = 2.0
    s = planner.make_solver(cfg, obst, via, batch, options=opt)
    for b in range(batch.count):
        G = s.debug_linearize(b, int(batch.n[b]), 2.0)
        R = oracle.linearize(cfg, obst, via, batch, b, 2.0)
        np.testing.assert_allclose(G["chi2"], R["chi2"], rtol=1e-12, atol=1e-14)
        assert np.abs(G["H"] - R["H"]).max() <= 1e-12 * np.abs(R["H"]).max()
        ap, ao = oracle.associate(cfg, obst, batch, b)
        np.testing.assert_array_equal(G["assoc_pose"], ap)
        np.testing.assert_array_equal(G["assoc_obst"], ao)
    s.close()
    out, res, best = run_gpu(cfg, obst, via, batch, options=opt)
    ref, rres = oracle.optimize_batch(cfg, obst, via, batch)
    assert_full_parity(out, res, ref, rres)


# ---- g2o-numeric Jacobian mode on the GPU: the reference's own linearisation scheme --------------------------------------
# Tolerances: the central differences divide residual differences by 2e-9, so last-bit differences between the device and
# host libm (sin/cos/pow; sqrt and the four operations are correctly rounded on both) appear as ~1e-7 relative noise in
# Jacobian entries - the same noise the reference has between two compilers (HISTORY.md section 5, "Compiler note").
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_ref_golden as RG  # noqa: E402

# bands compared / bands seen, per skip rule; the floors are asserted by test_skip_rules_have_a_floor at the end of this file
_SKIP_STATS = {"reference_code": [0, 0], "randomized": [0, 0]}

NUMERIC_CASES = ["edges_point", "edges_two_circles", "edges_line", "edges_polygon_carlike_arc", "edges_optional",
                 "edges_holonomic", "c1", "c1_velocities", "c2_small", "c3_small", "c5_small", "divergence_detection", "legacy_association"]


@pytest.mark.parametrize("name", NUMERIC_CASES)
def test_numeric_mode_linearisation_matches_oracle(oracle, name):
    cfg, obst, via, batch = RG.PLANNER_CASES[name]()
    cfg.jacobian_mode = _abi.JACOBIAN_G2O_NUMERIC
    s = planner.make_solver(cfg, obst, via, batch)
    for b in range(min(batch.count, 3)):
        G = s.debug_linearize(b, int(batch.n[b]), 2.0)
        R = oracle.linearize(cfg, obst, via, batch, b, 2.0)
        np.testing.assert_allclose(G["chi2"], R["chi2"], rtol=1e-12, atol=1e-14)
        assert np.abs(G["H"] - R["H"]).max() <= 2e-6 * np.abs(R["H"]).max()
        assert np.abs(G["b"] - R["b"]).max() <= 2e-6 * np.abs(R["b"]).max()
    s.close()


@pytest.mark.parametrize("mode", ["analytic", "g2o_numeric"])
@pytest.mark.parametrize("name", NUMERIC_CASES)
def test_optimizeTEB_matches_reference_code(oracle, name, mode):
    """GPU (either Jacobian mode) against the vectors produced by the REFERENCE's own src/optimal_planner.cpp
    (tests/golden/ref_opt_*.npz, see tests/test_reference_pinning.py) after the full 4 x 5 iterations: same pose count and
    success flag; poses, time differences and cost within the per-band tolerance of tests/sensitivity.py - 2e-5 for bands that
    damp the reference's own 1e-7 linearisation noise (observed 1e-8 .. 4e-6 in both modes), looser only where the CPU oracle's
    two Jacobian modes themselves drift apart (bands in collision), skipped where even their pose counts differ."""
    import sensitivity
    g = np.load(os.path.join(HERE, "golden", "ref_opt_%s.npz" % name))
    cfg, obst, via, batch = RG.PLANNER_CASES[name]()
    tols = sensitivity.band_tolerances(oracle, cfg, obst, via, batch)
    cfg.jacobian_mode = _abi.JACOBIAN_G2O_NUMERIC if mode == "g2o_numeric" else _abi.JACOBIAN_ANALYTIC
    out, res, _ = run_gpu(cfg, obst, via, batch)
    checked = 0
    for b in range(min(batch.count, RG.MAX_TEBS)):
        assert bool(g["success"][b]) == (res.status[b] == _abi.TEB_OK)
        if tols[b] is None:
            continue
        n = int(g["n"][b])
        assert int(out.n[b]) == n
        x, y, th, dt = out.get_teb(b)
        d = max(np.abs(x - g["state"][b, 0, :n]).max(), np.abs(y - g["state"][b, 1, :n]).max(),
                np.abs(th - g["state"][b, 2, :n]).max(), np.abs(dt - g["state"][b, 3, :n - 1]).max())
        assert d <= tols[b], (name, b, d, tols[b])
        assert abs(res.cost[b] - g["cost"][b]) <= tols[b] * abs(g["cost"][b]), (name, b, res.cost[b], g["cost"][b])
        checked += 1
    _SKIP_STATS["reference_code"][0] += checked
    _SKIP_STATS["reference_code"][1] += min(batch.count, RG.MAX_TEBS)
    assert checked >= 1


# ---- legacy obstacle association (AddEdgesObstaclesLegacy) on the device ---------------------------------------------------
def _legacy_case(kind, poses_affected):
    if kind == "mixed":      # generic distance path, all five obstacle types
        cfg, obst, via, batch = scenes.scene_small_mixed(footprint="circular")
    elif kind == "polygon":
        cfg, obst, via, batch = scenes.scene_small_mixed(footprint="polygon", stride=128)
    else:                    # point-like fast path (LDS obstacle cache)
        cfg, obst, via, batch = scenes.scene_c2(n=60, M=40, length=8.0, stride=256)
    cfg.obstacles.legacy_obstacle_association = True
    cfg.obstacles.obstacle_poses_affected = poses_affected
    return cfg, obst, via, batch


@pytest.mark.parametrize("kind", ["mixed", "polygon", "points"])
@pytest.mark.parametrize("poses_affected", [1, 6, 25, 1000])
def test_legacy_association_matches_oracle(oracle, kind, poses_affected):
    cfg, obst, via, batch = _legacy_case(kind, poses_affected)
    s = planner.make_solver(cfg, obst, via, batch)
    for b in range(batch.count):
        G = s.debug_linearize(b, int(batch.n[b]), 2.0)
        R = oracle.linearize(cfg, obst, via, batch, b, 2.0)
        ap, ao = oracle.associate(cfg, obst, batch, b)
        assert sorted(zip(G["assoc_pose"].tolist(), G["assoc_obst"].tolist())) == sorted(zip(ap.tolist(), ao.tolist()))
        np.testing.assert_allclose(G["chi2"], R["chi2"], rtol=1e-12, atol=1e-14)
        assert np.abs(G["H"] - R["H"]).max() <= 1e-12 * np.abs(R["H"]).max()
        assert np.abs(G["b"] - R["b"]).max() <= 1e-12 * np.abs(R["b"]).max()
    s.close()
    out, res, best = run_gpu(cfg, obst, via, batch)
    ref, rres = oracle.optimize_batch(cfg, obst, via, batch)
    assert_full_parity(out, res, ref, rres)


# ---- edge cases: empty scene, minimal / ragged bands, maximum capacities, non-finite input ------------------------------------------
def _straight_batch(cfg, ns, stride, rng):
    batch = _abi.TebBatchHost(len(ns), stride)
    for b, n in enumerate(ns):
        px, py, th, dt = scenes.sine_band(n, 0.25 * n, rng.uniform(-0.3, 0.3), 1.0, cfg.robot.max_vel_x)
        batch.set_teb(b, px, py, th, dt)
    return batch


def test_empty_scene_and_minimal_and_ragged_bands(oracle):
    cfg, _, _, _ = scenes.scene_small_mixed(footprint="point")
    rng = np.random.default_rng(4)
    obst = _abi.ObstacleTable()                               # no obstacles, no via-points
    batch = _straight_batch(cfg, [3, 4, 17, 120, 2], 320, rng)   # ragged: autoResize grows them to 8 ... 239 poses
    out, res, best = run_gpu(cfg, obst, [], batch)
    ref, rres = oracle.optimize_batch(cfg, obst, [], batch)
    assert_full_parity(out, res, ref, rres)
    assert (res.status == _abi.TEB_OK).all() and out.n[3] > 200
    assert best[0] == oracle.select_best(cfg, rres.cost)[0]
    # the same bands without resizing: the 3-pose band is optimised as it is (one free pose); the 2-pose band has no free vertex
    # and fewer poses than min_samples, so optimizeGraph refuses it (src/optimal_planner.cpp:376-381)
    cfg.trajectory.teb_autosize = False
    out, res, _ = run_gpu(cfg, obst, [], batch)
    ref, rres = oracle.optimize_batch(cfg, obst, [], batch)
    np.testing.assert_array_equal(out.n, ref.n)
    np.testing.assert_array_equal(res.status, rres.status)
    assert out.n[0] == 3 and res.status[0] == _abi.TEB_OK and res.status[4] == _abi.TEB_FAILED
    for b in range(4):
        for u, v in zip(out.get_teb(b), ref.get_teb(b)):
            assert np.abs(u - v).max() <= 1e-8
    np.testing.assert_allclose(res.cost[:4], rres.cost[:4], rtol=1e-8)


@pytest.mark.parametrize("n,solver", [(238, "cr"), (336, "band"), (337, "band"), (338, "bandg"), (500, "bandg"), (501, "bandg"), (512, "bandg")])
def test_maximum_pose_capacities(oracle, n, solver):
    """S = 238 is the largest band the block-cyclic-reduction solver holds in LDS (without the obstacle cache), S = 337 the largest for
    the band in LDS (45 doubles per pose: one padding double against bank conflicts); longer bands (the reference's max_samples default is 500) keep the band form of the normal matrix in HBM, up to
    512 poses (two per lane). 337 and 501: odd capacities (45 S is odd: the regions behind the band must still start on 16-byte
    boundaries)."""
    cfg, obst, via, _ = scenes.scene_small_mixed(footprint="point")
    cfg.trajectory.teb_autosize = False
    cfg.trajectory.max_samples = 500
    rng = np.random.default_rng(n)
    batch = _straight_batch(cfg, [n, n - 7], n, rng)
    s = planner.make_solver(cfg, obst, via, batch)
    lds, cap = s.capacity()
    assert cap >= n and _abi.MAX_POSES <= cap <= 1024        # the band-in-HBM layout: four poses per lane, LDS permitting (951 on MI355X)
    s.optimize(3, 2, True, cfg.hcp.selection_obst_cost_scale, cfg.hcp.selection_viapoint_cost_scale,
               cfg.hcp.selection_alternative_time_cost)
    res = s.results(); out = s.download(batch.copy()); s.close()
    ref, rres = oracle.optimize_batch(cfg, obst, via, batch, inner=3, outer=2)
    assert_full_parity(out, res, ref, rres, pos_tol=1e-7, rtol=1e-7)
    with pytest.raises(planner.TebAmdError) as e:             # beyond what the LDS strips of the band-in-HBM layout hold
        planner.TebBatchSolver(cfg, 1, cap + 1, 4, 4, 1)
    assert e.value.code == _abi.ERR_CAPACITY


def _long_scene(n, kind, rng, stride=None, ns=None):
    """Bands of ~ n poses (0.25 m per pose) with obstacles ALONG them: static ones every few metres and dynamic ones placed so that their
    predicted position at the time stamp of poses in every pass (0 .. 255, 256 .. 511, 512 .. ) lies next to the band."""
    cfg, mixed, via, _ = scenes.scene_small_mixed(footprint="point" if kind == "points" else "circular")
    batch = _straight_batch(cfg, ns or [n, n - 13], stride or n, rng)
    px, py, th, dt = batch.get_teb(0)
    t = np.concatenate([[0.0], np.cumsum(dt)])
    obst = _abi.ObstacleTable() if kind == "points" else mixed
    for k in range(0, len(px), 9):
        side = 1.0 if (k // 9) % 2 else -1.0
        if kind == "points" or k % 27:
            obst.add_point(px[k] + 0.1, py[k] + side * rng.uniform(0.35, 0.9))
        else:
            obst.add_polygon([(px[k], py[k] + side * 0.6), (px[k] + 0.4, py[k] + side * 0.6), (px[k] + 0.2, py[k] + side * 1.0)])
    for k in range(20, len(px), 67):                          # dynamic: at pose k's time stamp 0.45 m beside pose k
        vx, vy = rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1)
        obst.add_point(px[k] - vx * t[k] + 0.1, py[k] - vy * t[k] + 0.45, vel=(vx, vy))
    return cfg, obst, via, batch


@pytest.mark.parametrize("n,kind", [(513, "points"), (700, "points"), (769, "points"), (_abi.MAX_POSES, "points"), (640, "mixed")])
def test_bands_beyond_512_poses(oracle, n, kind):
    """trajectory.max_samples is a parameter of the reference (teb_config.h:78, 258; 500 is its default): bands of more than 512 poses run
    the band-in-HBM instantiations with three or four poses per lane. 513 / 769: the first pose of a third / fourth
    pass; 700: a leftover pass that is sliced; MAX_POSES: the capacity limit; "mixed": generic shapes (polygon distances, circular robot)."""
    rng = np.random.default_rng(n)
    cfg, obst, via, batch = _long_scene(n, kind, rng)
    cfg.trajectory.teb_autosize = False
    cfg.trajectory.max_samples = 1000
    s = planner.make_solver(cfg, obst, via, batch)
    s.optimize(2, 2, True, cfg.hcp.selection_obst_cost_scale, cfg.hcp.selection_viapoint_cost_scale,
               cfg.hcp.selection_alternative_time_cost)
    res = s.results(); out = s.download(batch.copy()); s.close()
    ref, rres = oracle.optimize_batch(cfg, obst, via, batch, inner=2, outer=2)
    assert (rres.status == _abi.TEB_OK).all()
    assert_full_parity(out, res, ref, rres, pos_tol=1e-7, rtol=1e-7)


@pytest.mark.parametrize("variant", ["holonomic", "carlike", "legacy_association", "arc_length_exponent", "g2o_numeric", "shortest_path_vel_ratio",
                                     "divergence_statistics", "generic_kernel"])
def test_bands_beyond_512_poses_off_the_defaults(oracle, variant):
    """The band-in-HBM instantiations with more than two poses per lane exist in every kind (generic, light, numeric mode ..): 600 and 587
    poses with one option off the defaults each, against the oracle."""
    rng = np.random.default_rng(600)
    cfg, obst, via, batch = _long_scene(600, "mixed" if variant == "carlike" else "points", rng)
    cfg.trajectory.teb_autosize = False
    cfg.trajectory.max_samples = 1000
    options, tol = None, 1e-7
    if variant == "holonomic":
        cfg.robot.max_vel_y = 0.3; cfg.optim.weight_kinematics_nh = 1.0
    elif variant == "carlike":
        cfg.robot.min_turning_radius = 1.0
    elif variant == "legacy_association":
        cfg.obstacles.legacy_obstacle_association = True
    elif variant == "arc_length_exponent":
        cfg.trajectory.exact_arc_length = True; cfg.optim.obstacle_cost_exponent = 1.5
    elif variant == "g2o_numeric":
        cfg.jacobian_mode = _abi.JACOBIAN_G2O_NUMERIC; tol = 1e-6   # (central differences: test_numeric_jacobians_on_long_bands)
    elif variant == "shortest_path_vel_ratio":
        cfg.optim.weight_shortest_path = 1.0; cfg.optim.weight_velocity_obstacle_ratio = 0.5
    elif variant == "divergence_statistics":
        cfg.recovery.divergence_detection_enable = True
    elif variant == "generic_kernel":
        options = _abi.Options(generic_config_path=True)
    s = planner.make_solver(cfg, obst, via, batch, options=options)
    s.optimize(2, 2, True, cfg.hcp.selection_obst_cost_scale, cfg.hcp.selection_viapoint_cost_scale,
               cfg.hcp.selection_alternative_time_cost)
    res = s.results(); out = s.download(batch.copy()); s.close()
    ref, rres = oracle.optimize_batch(cfg, obst, via, batch, inner=2, outer=2)
    assert (rres.status == _abi.TEB_OK).all()
    assert_full_parity(out, res, ref, rres, pos_tol=tol, rtol=1e-7)


def test_band_beyond_512_poses_with_autoresize_and_dynamic_obstacles(oracle):
    """A 480-pose band whose time differences call for more samples grows past 512 poses inside the kernel (max_samples 900); dynamic
    obstacles beside poses of every pass exercise the near masks of the third pass, which are not cached (two slots per lane)."""
    rng = np.random.default_rng(91)
    cfg, obst, via, batch = _long_scene(480, "points", rng, stride=_abi.MAX_POSES, ns=[480, 300])
    cfg.trajectory.max_samples = 900          # (a sweep may stop at max_samples intervals = 901 poses: the capacity must cover max_samples + 1)
    batch.dt[0, :200] *= 1.7            # above dt_ref + dt_hysteresis: these intervals are split
    s = planner.make_solver(cfg, obst, via, batch)
    s.optimize(3, 2, True, cfg.hcp.selection_obst_cost_scale, cfg.hcp.selection_viapoint_cost_scale,
               cfg.hcp.selection_alternative_time_cost)
    res = s.results(); out = s.download(batch.copy()); s.close()
    ref, rres = oracle.optimize_batch(cfg, obst, via, batch, inner=3, outer=2)
    assert int(out.n.max()) > 512
    assert_full_parity(out, res, ref, rres, pos_tol=1e-7, rtol=1e-7)


@pytest.mark.parametrize("n", [300, 400])
def test_numeric_jacobians_on_long_bands(oracle, n):
    """The g2o-numeric instantiations of the band layouts (LDS band at 300 poses, HBM band at 400) against the oracle in the same mode."""
    cfg, obst, via, _ = scenes.scene_small_mixed(footprint="point")
    cfg.trajectory.teb_autosize = False
    cfg.trajectory.max_samples = 500
    cfg.jacobian_mode = _abi.JACOBIAN_G2O_NUMERIC
    rng = np.random.default_rng(n)
    batch = _straight_batch(cfg, [n, n - 11], n, rng)
    s = planner.make_solver(cfg, obst, via, batch)
    s.optimize(2, 2, True, cfg.hcp.selection_obst_cost_scale, cfg.hcp.selection_viapoint_cost_scale,
               cfg.hcp.selection_alternative_time_cost)
    res = s.results(); out = s.download(batch.copy()); s.close()
    ref, rres = oracle.optimize_batch(cfg, obst, via, batch, inner=2, outer=2)
    # central differences with delta = 1e-9 turn the rounding of a different (equally valid) summation order of H into ~1e-7 relative
    # noise on the Jacobians (tests/sensitivity.py); on coordinates up to 100 m that is the 1e-7 seen here (observed: 1.1e-7 at n = 400)
    assert_full_parity(out, res, ref, rres, pos_tol=1e-6, rtol=1e-7)


def test_long_band_with_autoresize_grows_past_the_lds_band(oracle):
    """A 300-pose band whose time differences call for more samples: autoResize grows it past 337 poses inside the kernel (band form in
    HBM), same result as the oracle."""
    cfg, obst, via, _ = scenes.scene_small_mixed(footprint="point")
    cfg.trajectory.max_samples = 500
    rng = np.random.default_rng(77)
    batch = _straight_batch(cfg, [300, 280], 512, rng)
    for b in range(2):
        k = int(batch.n[b])
        batch.dt[b, :120] *= 1.6            # above dt_ref + dt_hysteresis: these intervals are split
    s = planner.make_solver(cfg, obst, via, batch)
    s.optimize(3, 2, True, cfg.hcp.selection_obst_cost_scale, cfg.hcp.selection_viapoint_cost_scale,
               cfg.hcp.selection_alternative_time_cost)
    res = s.results(); out = s.download(batch.copy()); s.close()
    ref, rres = oracle.optimize_batch(cfg, obst, via, batch, inner=3, outer=2)
    assert int(out.n.max()) > 337
    assert_full_parity(out, res, ref, rres, pos_tol=1e-7, rtol=1e-7)


def test_non_finite_input_is_reported_not_propagated(oracle):
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint="point")
    bad = batch.copy()
    bad.x[1, 5] = np.nan
    out, res, best = run_gpu(cfg, obst, via, bad)
    assert res.status[1] == _abi.TEB_NONFINITE
    good, gres, _ = run_gpu(cfg, obst, via, batch)
    for b in (0, 2):                                          # the other candidates are unaffected
        assert res.status[b] == _abi.TEB_OK
        np.testing.assert_array_equal(out.get_teb(b)[0], good.get_teb(b)[0])


# ---- randomized scenes: every option toggled at random, full optimizeTEB parity with the oracle ------------------------------------
from random_cases import random_case as _random_case  # noqa: E402


@pytest.mark.parametrize("seed", range(80))
def test_randomized_scenes_full_parity(oracle, seed):
    """Same closed-form mode on both sides, EVERY band (no skip rule): status, pose counts, LM iteration / trial counts identical; poses
    and time differences <= 1e-7 on bands that are well conditioned by the yardstick of tests/sensitivity.py and <= 1e-6 on the others
    (bands that start in collision amplify last-bit differences of the device's libm; observed on MI355X over the 240 bands of the 80
    seeds, tools/random_parity_probe.py: 190 below 1e-12, 231 below 1e-10, the worst 5e-8); cost rel to the same bounds."""
    import sensitivity
    cfg, obst, via, batch = _random_case(seed)
    tols = sensitivity.band_tolerances(oracle, cfg, obst, via, batch)
    out, res, best = run_gpu(cfg, obst, via, batch)
    ref, rres = oracle.optimize_batch(cfg, obst, via, batch)
    np.testing.assert_array_equal(res.status, rres.status)
    for b in range(batch.count):
        well = tols[b] is not None and tols[b] <= sensitivity.WELL_CONDITIONED_TOL
        bound = 1e-7 if well else 1e-6
        assert out.n[b] == ref.n[b], (seed, b)
        assert res.lm_iterations[b] == rres.lm_iterations[b] and res.lm_trials[b] == rres.lm_trials[b], (seed, b)
        for u, v in zip(out.get_teb(b), ref.get_teb(b)):
            assert np.abs(u - v).max() <= bound, (seed, b, np.abs(u - v).max())
        if np.isfinite(rres.cost[b]):
            assert abs(res.cost[b] - rres.cost[b]) <= bound * abs(rres.cost[b])
        _SKIP_STATS["randomized"][0] += 1
    _SKIP_STATS["randomized"][1] += batch.count
    assert best[0] == oracle.select_best(cfg, rres.cost)[0]


@pytest.mark.parametrize("layout", ["band", "bandg"])
@pytest.mark.parametrize("seed", range(0, 80, 5))
def test_randomized_scenes_do_not_depend_on_the_layout(seed, layout):
    """The three layouts of the normal matrix (8x8 blocks in LDS, band in LDS, band in HBM) run the same arithmetic up to the order of
    the block reduction: on random scenes (all options of the path toggled at random) the forced band layouts give the pose counts,
    LM iteration / trial counts and status of the default layout and poses within 1e-8."""
    cfg, obst, via, batch = _random_case(seed)
    out0, res0, _ = run_gpu(cfg, obst, via, batch)
    out1, res1, _ = run_gpu(cfg, obst, via, batch, options=_abi.Options(layout=layout))
    np.testing.assert_array_equal(res0.status, res1.status)
    np.testing.assert_array_equal(out0.n, out1.n)
    np.testing.assert_array_equal(res0.lm_iterations, res1.lm_iterations)
    for b in range(batch.count):
        if res0.status[b] != _abi.TEB_OK:
            continue
        for u, v in zip(out0.get_teb(b), out1.get_teb(b)):
            assert np.abs(u - v).max() <= 1e-8, (seed, b, np.abs(u - v).max())


def test_layout_is_chosen_per_launch_and_repeated_when_a_band_outgrows_it(oracle):
    """A handle created for 501 poses (band in HBM) that holds a 194-pose band is launched with the blocks in LDS: same result as a
    208-pose handle to 1e-8, at its speed. A band that outgrows the optimistic capacity inside the kernel (autoResize doubles it) makes
    the library repeat the launch from the saved strips in the handle's own layout: bit-identical to a handle with teb_amd_options_t::fixed_layout."""
    cfg, obst, via, batch = scenes.scene_c2(stride=208)
    def run(cap, b=batch, fixed=False):
        hb = _abi.TebBatchHost(b.count, cap)
        for k in range(b.count):
            hb.set_teb(k, *b.get_teb(k))
        s = planner.make_solver(cfg, obst, via, hb, options=_abi.Options(fixed_layout=fixed))
        s.optimize(5, 4, True, 100.0, 1.0, False); s.synchronize()
        ms = s.last_kernel_ms(); res = s.results(); out = s.download(hb.copy()); s.close()
        return out, res, ms
    small, rs, ms_small = run(208)
    big, rb, ms_big = run(501)
    fixed, rf, ms_fixed = run(501, fixed=True)
    assert small.n[0] == big.n[0] == fixed.n[0] and rs.lm_iterations[0] == rb.lm_iterations[0] == rf.lm_iterations[0]
    for u, v, w in zip(small.get_teb(0), big.get_teb(0), fixed.get_teb(0)):
        assert np.abs(u - v).max() <= 1e-8 and np.abs(u - w).max() <= 1e-8
    assert ms_big < 0.85 * ms_fixed, (ms_small, ms_big, ms_fixed)          # the per-launch choice is what makes the difference
    # outgrowing: 150 poses whose time differences call for a split of every interval -> ~300 poses > the optimistic block capacity
    rng = np.random.default_rng(3)
    grow = _abi.TebBatchHost(1, 501)
    x, y, th, dt = scenes.sine_band(150, 30.0, 0.2, 1.0, cfg.robot.max_vel_x)
    grow.set_teb(0, x, y, th, dt * 0 + 0.75)
    cfg.trajectory.max_samples = 500
    def run_grow(fixed=False):
        s = planner.make_solver(cfg, obst, via, grow, options=_abi.Options(fixed_layout=fixed))
        s.optimize(3, 2, True, 100.0, 1.0, False); s.synchronize()
        ms = s.last_kernel_ms(); res = s.results(); out = s.download(grow.copy()); s.close()
        return out, res, ms
    a, ra, ms_a = run_grow()
    b, rb2, ms_b = run_grow(fixed=True)
    assert ms_a > ms_b          # teb_amd_last_kernel_ms covers the discarded optimistic launch as well as the repeated one
    assert a.n[0] == b.n[0] > 238 and ra.status[0] == rb2.status[0] == _abi.TEB_OK
    for u, v in zip(a.get_teb(0), b.get_teb(0)):
        np.testing.assert_array_equal(u, v)


def test_skip_rules_have_a_floor():
    """The comparison with the reference-code vectors leaves out bands on which even the CPU oracle's two Jacobian modes end with
    different pose counts; the randomized comparison skips nothing. Neither rule may hide a regression: at least 90 % of the bands
    seen were actually compared (observed on MI355X: reference-code vectors 60 of 62, randomized scenes 240 of 240). Runs after them
    (file order); skipped when they were deselected."""
    for name, (checked, seen) in _SKIP_STATS.items():
        print("%s: %d of %d bands compared, %d skipped as ill conditioned" % (name, checked, seen, seen - checked))
    if all(seen == 0 for _, seen in _SKIP_STATS.values()):
        pytest.skip("the parametrised comparisons did not run in this session")
    for name, (checked, seen) in _SKIP_STATS.items():
        if seen:
            assert checked >= 0.9 * seen, (name, checked, seen)


@pytest.mark.parametrize("jmode", ["analytic", "numeric"])
def test_cached_near_masks_do_not_change_one_bit(jmode):
    """The near masks of the dynamic-obstacle edges are cached per lane across one optimize() (a superset taken at a reference position,
    exact test on the candidates); teb_amd_options_t::no_near_cache recomputes the exact mask at every pass. Both must give the same
    bits: bands, costs, chi2, iteration and trial counts - on a scene whose poses move by more than the margin of the cache (the
    recomputation path) and with fast and slow moving obstacles, in both Jacobian modes."""
    cfg, obst, via, batch = scenes.scene_c4(B=24, n=120, M_static=60, M_dyn=64, seed=77, stride=160, length=12.0)
    cfg.jacobian_mode = _abi.JACOBIAN_G2O_NUMERIC if jmode == "numeric" else _abi.JACOBIAN_ANALYTIC
    outs = []
    for off in (False, True):
        out, res, best = run_gpu(cfg, obst, via, batch, options=_abi.Options(no_near_cache=off))
        outs.append((out, res, best))
    (a, ra, ba), (b, rb, bb) = outs
    assert ba == bb
    np.testing.assert_array_equal(a.n, b.n)
    for k in ("status", "lm_iterations", "lm_trials"):
        np.testing.assert_array_equal(getattr(ra, k), getattr(rb, k))
    for u, v in ((a.x, b.x), (a.y, b.y), (a.theta, b.theta), (a.dt, b.dt), (ra.cost, rb.cost), (ra.chi2, rb.chi2)):
        assert np.array_equal(u, v, equal_nan=True)
    assert ra.lm_iterations.sum() >= batch.count and (ra.lm_trials >= ra.lm_iterations).all()   # the batch did iterate


def test_polygon_footprint_with_more_than_16_vertices(oracle):
    """Round 4: polygon footprints up to 64 vertices in the optimiser (the reference's PolygonRobotFootprint has no limit,
    robot_footprint_model.h:664-683; ABI 2 took 16). A 40-gon: distances and gradients against the oracle, then the full optimizeTEB."""
    from teb_local_planner_amd.config import RobotFootprintModel
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint="polygon")
    verts = [(0.05 + 0.32 * np.cos(2 * np.pi * k / 40), 0.22 * np.sin(2 * np.pi * k / 40)) for k in range(40)]
    cfg.robot_model = RobotFootprintModel.polygon(verts)
    s = planner.make_solver(cfg, obst, via, batch)
    rng = np.random.default_rng(12)
    nq = 200
    oi = rng.integers(0, len(obst), nq); x = rng.uniform(0, 6, nq); y = rng.uniform(-2, 2, nq); th = rng.uniform(-3.1, 3.1, nq)
    d, g = s.debug_distance(oi, x, y, th, None)
    for q in range(nq):
        do, go = oracle.distance(cfg, obst, int(oi[q]), x[q], y[q], th[q], None)
        assert abs(d[q] - do) <= 1e-14 * max(1.0, abs(do))
        assert np.abs(g[q] - go).max() <= 1e-12
    s.close()
    out, res, best = run_gpu(cfg, obst, via, batch)
    ref, rres = oracle.optimize_batch(cfg, obst, via, batch)
    assert_full_parity(out, res, ref, rres, pos_tol=1e-7, rtol=1e-7)
    with pytest.raises(ValueError):
        cfg.robot_model = RobotFootprintModel.polygon([(np.cos(k), np.sin(k)) for k in range(65)])
        cfg.to_c()
