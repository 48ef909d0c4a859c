"""The reference-side binding of INTEGRATION.md, compiled and run: teb_local_planner_amd/host/teb_amd_backend.cpp (TebOptimalPlannerAmd,
TebAmdBatch, TebConfig / ObstContainer adapters) built against the REFERENCE's own classes and linked to libteb_amd.so
(oracle/_ref/libteb_backend_check.so, built where /root/reference exists; the binary travels to the GPU box).

Each case hands the SAME reference objects (TebConfig, ObstContainer, ViaPointContainer, TimedElasticBand) to
  (1) the reference's TebOptimalPlanner::optimizeTEB — its own src/optimal_planner.cpp on the CPU (LM stand-in of shim_g2o.h), and
  (2) TebOptimalPlannerAmd::optimizeTEB / TebAmdBatch::optimizeAllTEBs + selectBestTeb — the MI355X through the C-ABI,
and compares what ends up in the planners' TimedElasticBand / getCurrentCost() / isOptimized().
Tolerances: per band, tests/sensitivity.py (2e-5 m / rad / s and relative cost for bands that damp the reference's own
linearisation noise; looser only where the CPU oracle's two Jacobian modes drift apart themselves).
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_ref_golden as RG  # noqa: E402
sys.path.insert(0, HERE)
import sensitivity  # noqa: E402

from teb_local_planner_amd import _abi  # noqa: E402

SO = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libteb_backend_check.so")
pytestmark = pytest.mark.gpu


def _lib():
    if not os.path.exists(SO):
        pytest.skip("oracle/_ref/libteb_backend_check.so not built (needs /root/reference at build time)")
    return C.CDLL(SO)


def run(cfg, obst, via, batch, jacobian_mode, mode, run_reference=True, last_best=-1, initial_plan=-1, inner=None, outer=None):
    L = _lib()
    c = cfg.to_c()
    B, S = batch.count, batch.stride
    vx = _abi.f64([v[0] for v in via]) if via else _abi.f64([0.0])
    vy = _abi.f64([v[1] for v in via]) if via else _abi.f64([0.0])
    P = lambda a: _abi._ptr(a, C.c_double)
    I = lambda a: _abi._ptr(a, C.c_int32)
    out = {}
    for side in ("ref", "amd"):
        for k in ("x", "y", "th", "dt"):
            out[side + "_" + k] = np.zeros((B, S))
        out[side + "_n"] = np.zeros(B, np.int32); out[side + "_cost"] = np.zeros(B); out[side + "_ok"] = np.zeros(B, np.int32)
    best = C.c_int32(-1); best_cost = C.c_double(0); rt = C.c_int32(0)
    bs = batch.c_struct()
    inner = cfg.optim.no_inner_iterations if inner is None else inner
    outer = cfg.optim.no_outer_iterations if outer is None else outer
    args = [C.byref(c), C.byref(obst.freeze()), len(via), P(vx), P(vy), C.byref(bs), int(inner), int(outer), 1,
            C.c_double(cfg.hcp.selection_obst_cost_scale), C.c_double(cfg.hcp.selection_viapoint_cost_scale),
            int(cfg.hcp.selection_alternative_time_cost), int(jacobian_mode), int(mode), int(run_reference), int(last_best),
            int(initial_plan)]
    for side in ("ref", "amd"):
        args += [P(out[side + "_x"]), P(out[side + "_y"]), P(out[side + "_th"]), P(out[side + "_dt"]), I(out[side + "_n"]),
                 P(out[side + "_cost"]), I(out[side + "_ok"])]
    args += [C.byref(best), C.byref(best_cost), C.byref(rt)]
    rc = L.backend_check_run(*args)
    assert rc == 0, rc
    out["best"] = best.value; out["best_cost"] = best_cost.value; out["roundtrip_ok"] = rt.value
    return out


def compare(out, tols):
    """Per-band tolerances from tests/sensitivity.py (None = the band bifurcates under the reference's own noise: flags only)."""
    checked = 0
    for b, tol in enumerate(tols):
        assert out["amd_ok"][b] == out["ref_ok"][b]
        if tol is None:
            continue
        n = int(out["ref_n"][b])
        assert int(out["amd_n"][b]) == n
        d = max(np.abs(out["amd_" + k][b, :n] - out["ref_" + k][b, :n]).max() for k in ("x", "y", "th"))
        d = max(d, np.abs(out["amd_dt"][b, :n - 1] - out["ref_dt"][b, :n - 1]).max())
        assert d <= tol, (b, d, tol)
        assert abs(out["amd_cost"][b] - out["ref_cost"][b]) <= tol * abs(out["ref_cost"][b])
        checked += 1
    assert checked >= 1


CASES = ["edges_point", "edges_two_circles", "edges_line", "edges_polygon_carlike_arc", "edges_optional", "edges_holonomic",
         "c1", "c1_velocities", "c3_small", "legacy_association"]


@pytest.mark.parametrize("name", CASES)
def test_batch_backend_on_reference_objects_numeric_mode(oracle, name):
    cfg, obst, via, batch = RG.PLANNER_CASES[name]()
    tols = sensitivity.band_tolerances(oracle, cfg, obst, via, batch)
    out = run(cfg, obst, via, batch, _abi.JACOBIAN_G2O_NUMERIC, mode=0)
    assert out["roundtrip_ok"] == 1          # TebConfig / ObstContainer adapters reproduce the inputs exactly
    compare(out, tols)
    # selectBestTeb on the device-resident costs == strict '<' arg-min over the planners' getCurrentCost()
    assert out["best"] == int(np.argmin(out["amd_cost"]))
    if all(t is not None for t in tols):
        assert out["best"] == int(np.argmin(out["ref_cost"]))


@pytest.mark.parametrize("name", ["edges_point", "c1", "edges_polygon_carlike_arc"])
def test_batch_backend_on_reference_objects_analytic_mode(oracle, name):
    cfg, obst, via, batch = RG.PLANNER_CASES[name]()
    tols = sensitivity.band_tolerances(oracle, cfg, obst, via, batch)
    out = run(cfg, obst, via, batch, _abi.JACOBIAN_ANALYTIC, mode=0)
    compare(out, tols)


@pytest.mark.parametrize("name", ["edges_point", "c1_velocities"])
def test_single_planner_optimizeTEB_override(oracle, name):
    """TebOptimalPlannerAmd::optimizeTEB (one lazily created handle per planner) instead of the batch entry point."""
    cfg, obst, via, batch = RG.PLANNER_CASES[name]()
    tols = sensitivity.band_tolerances(oracle, cfg, obst, via, batch)
    out = run(cfg, obst, via, batch, _abi.JACOBIAN_G2O_NUMERIC, mode=1)
    compare(out, tols)


def test_backend_error_behaviour_matches_reference():
    """optimization_activate = false and a too-short band: optimizeTEB returns false on both sides, bands untouched."""
    cfg, obst, via, batch = RG.PLANNER_CASES["edges_point"]()
    cfg.optim.optimization_activate = False
    out = run(cfg, obst, via, batch, _abi.JACOBIAN_ANALYTIC, mode=1)
    assert not out["amd_ok"].any() and not out["ref_ok"].any()
    for b in range(batch.count):
        n = int(batch.n[b])
        np.testing.assert_array_equal(out["amd_x"][b, :n], batch.x[b, :n])
    cfg, obst, via, batch = RG.PLANNER_CASES["edges_point"]()
    cfg.trajectory.teb_autosize = False
    cfg.trajectory.min_samples = 30
    out = run(cfg, obst, via, batch, _abi.JACOBIAN_ANALYTIC, mode=0)
    assert not out["amd_ok"].any() and not out["ref_ok"].any()


def test_selection_hysteresis_through_backend():
    cfg, obst, via, batch = RG.PLANNER_CASES["c3_small"]()
    out = run(cfg, obst, via, batch, _abi.JACOBIAN_ANALYTIC, mode=0)
    costs = out["amd_cost"].copy()
    order = np.argsort(costs)
    runner_up = int(order[1])
    cfg.hcp.selection_cost_hysteresis = float(0.5 * costs[order[0]] / costs[runner_up])   # makes the runner-up win when it was last best
    out2 = run(cfg, obst, via, batch, _abi.JACOBIAN_ANALYTIC, mode=0, run_reference=False, last_best=runner_up)
    assert out2["best"] == runner_up


# ---- rows either side of the path through the C++ backend, against the reference's own member functions --------------------------
@pytest.mark.parametrize("dynamic", [True, False])
def test_backend_rows_f1_f2_f3_on_reference_objects(dynamic):
    """TebAmdBatch::updateAllTEBs / renewAndAnalyzeOldTebs / getVelocityCommand against TimedElasticBand::updateAndPruneTEB,
    HSignature3d / HSignature (+ isEqual / isValid) and TebOptimalPlanner::getVelocityCommand of the reference, same objects."""
    L = _lib()
    cname, cfg, obst, batch = RG.h_signature_cases()[2]
    cfg.obstacles.include_dynamic_obstacles = dynamic
    c = cfg.to_c()
    B, S, M = batch.count, batch.stride, len(obst)
    W = M if dynamic else 2
    x0, y0, th0, _ = batch.get_teb(0)
    start = _abi.f64([x0[2] + 0.01, y0[2] - 0.01, th0[2]]); goal = _abi.f64([x0[-1], y0[-1], th0[-1]]); sv = _abi.f64([0.1, 0.0, 0.02])
    P = lambda a: _abi._ptr(a, C.c_double)
    I = lambda a: _abi._ptr(a, C.c_int32)
    pr = np.zeros((B, 4, S)); pa = np.zeros((B, 4, S)); nr = np.zeros(B, np.int32); na = np.zeros(B, np.int32)
    sr = np.zeros((B, W)); sa = np.zeros((B, W)); kr = np.zeros(B, np.int32); ka = np.zeros(B, np.int32); w = C.c_int32(0)
    cr = np.zeros((B, 4)); ca = np.zeros((B, 4))
    bs = batch.c_struct()
    rc = L.backend_check_rows(C.byref(c), C.byref(obst.freeze()), C.byref(bs), P(start), P(goal), P(sv), 5, 4, 1,
                              P(pr), P(pa), I(nr), I(na), P(sr), P(sa), I(kr), I(ka), C.byref(w), P(cr), P(ca))
    assert rc == 0, rc
    assert w.value == W
    # f1: pruning is IEEE-exact on both sides
    np.testing.assert_array_equal(nr, na)
    assert (nr < batch.n).any()
    for b in range(B):
        n = int(nr[b])
        np.testing.assert_array_equal(pa[b, :3, :n], pr[b, :3, :n]); np.testing.assert_array_equal(pa[b, 3, :n - 1], pr[b, 3, :n - 1])
    # f3: signatures of the optimised bands (the reference evaluates them on the bands the backend wrote back)
    if dynamic:
        assert np.abs(sa - sr).max() <= 4 * np.finfo(float).eps * max(1.0, np.abs(sr).max())
    else:
        assert np.abs(sa - sr).max() <= 1e-10 * max(np.abs(sr).max(), 1e-300)
    np.testing.assert_array_equal(ka, kr)
    assert ka[0] == 1
    # f2: velocity command from the device-resident band == the reference's on the written-back band
    np.testing.assert_array_equal(ca[:, 3], cr[:, 3])
    assert np.abs(ca[:, :3] - cr[:, :3]).max() <= 1e-12


# ---- candidate generation through the binding: TebAmdBatch::exploreEquivalenceClassesAndInitTebs vs the reference's own method ----------
@pytest.mark.parametrize("name", ["roadmap_existing_tebs_best", "detours_existing_tebs", "keypoint_mixed_3d", "keypoint_points_2d",
                                  "backwards_start_velocity_free_goal", "goal_reached_line_init", "already_full"])
def test_binding_explores_like_the_reference_planner(name):
    """Same TebConfig / ObstContainer / TimedElasticBand objects -> (1) HomotopyClassPlanner::exploreEquivalenceClassesAndInitTebs of the
    reference (CPU), (2) the binding (device). Same candidates in the same order; bands <= 1e-12 (no optimisation involved)."""
    case = RG.explore_cases()[name]
    if case.get("skip_draws"):
        pytest.skip("needs a planner whose generator has been used before")
    L = _lib()
    cfg, obst, batch = case["cfg"], case["obst"], case["batch"]
    c = cfg.to_c(); p = cfg.hcp_params()
    slots, S = 12, 256
    ref = _abi.TebBatchHost(slots, S); amd = _abi.TebBatchHost(slots, S)
    rs, as_ = ref.c_struct(), amd.c_struct()
    P = lambda a: _abi._ptr(a, C.c_double)
    I = lambda a: _abi._ptr(a, C.c_int32)
    st = _abi.f64(case["start"]); gl = _abi.f64(case["goal"])
    sv = None if case.get("start_vel") is None else _abi.f64(case["start_vel"])
    opt = None if case.get("optimized") is None else _abi.i32(case["optimized"])
    nr = C.c_int32(0); na = C.c_int32(0); ab = C.c_int32(-2)
    flags = np.zeros(2 * slots, np.int32)
    ins = batch.c_struct() if batch is not None else None
    L.backend_check_explore.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                        C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    vp = lambda x: C.cast(x, C.c_void_p) if x is not None else None
    rc = L.backend_check_explore(vp(C.pointer(c)), vp(C.pointer(p)), vp(C.pointer(obst.freeze())), vp(C.pointer(ins)) if ins is not None else None,
                                 int(case["best"]), vp(I(opt)) if opt is not None else None, vp(P(st)), vp(P(gl)),
                                 float(case.get("dist_to_obst") or cfg.obstacles.min_obstacle_dist), vp(P(sv)) if sv is not None else None,
                                 int(bool(case.get("free_goal_vel", False))), vp(C.pointer(rs)), vp(C.pointer(nr)), vp(C.pointer(as_)),
                                 vp(C.pointer(na)), vp(C.pointer(ab)), vp(I(flags)))
    assert rc == 0, rc
    assert nr.value == na.value, (nr.value, na.value)
    for k in range(nr.value):
        for a, b in zip(ref.get_teb(k), amd.get_teb(k)):
            assert len(a) == len(b) and np.abs(a - b).max(initial=0) <= 1e-12, (name, k)
    n_old = 0 if batch is None else batch.count
    if case.get("free_goal_vel") and na.value:
        assert not flags[1:2 * na.value:2].any()      # setVelocityGoalFree() on every new candidate
    assert ab.value in (-1, 0)


# ---- the drop-in class: HomotopyClassPlannerAmd vs the reference's HomotopyClassPlanner, tick by tick ---------------------------------------
def _hcp_ticks(which, case, slots=8, stride=256, jacobian_mode=_abi.JACOBIAN_ANALYTIC):
    L = _lib()
    cfg = case["cfg"]
    c = cfg.to_c(); p = cfg.hcp_params()
    st = _abi.f64(case["starts"]).reshape(-1, 3); gl = _abi.f64(case["goals"]).reshape(-1, 3)
    T = len(st)
    sv = None if case.get("start_vels") is None else _abi.f64(case["start_vels"]).reshape(T, 3)
    out = _abi.TebBatchHost(T * slots, stride)
    obs = out.c_struct()
    counts = np.zeros(T, np.int32); best = np.zeros(T, np.int32); ipt = np.zeros(T, np.int32); costs = np.zeros(T * slots); cmd = np.zeros((T, 4))
    plans = case.get("plans") or [None] * T
    off = np.zeros(T + 1, np.int32)
    for t in range(T):
        off[t + 1] = off[t] + (0 if plans[t] is None else len(plans[t][0]))
    cat = lambda k: _abi.f64(np.concatenate([np.asarray(pl[k], np.float64) for pl in plans if pl is not None] or [np.zeros(1)]))
    px, py, pyaw = cat(0), cat(1), cat(2)
    via = case.get("via") or []
    vx = _abi.f64([v[0] for v in via] or [0.0]); vy = _abi.f64([v[1] for v in via] or [0.0])
    P = lambda a: C.cast(_abi._ptr(a, C.c_double), C.c_void_p) if a is not None else None
    I = lambda a: C.cast(_abi._ptr(a, C.c_int32), C.c_void_p)
    vp = lambda x: C.cast(C.pointer(x), C.c_void_p)
    L.backend_check_hcp_ticks.argtypes = [C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_int] + [C.c_void_p] * 8 + \
        [C.c_int] + [C.c_void_p] * 4 + [C.c_int]
    rc = L.backend_check_hcp_ticks(which, vp(c), vp(p), vp(case["obst"].freeze()), T, P(st), P(gl), P(sv), 0, slots, vp(obs), I(counts), I(best),
                                   P(costs), I(off), P(px), P(py), P(pyaw), len(via), P(vx), P(vy), I(ipt), P(cmd), int(jacobian_mode))
    assert rc == 0, rc
    return [dict(bands=[out.get_teb(t * slots + k) for k in range(counts[t])], best=int(best[t]), initial=int(ipt[t]),
                 costs=costs[t * slots:t * slots + counts[t]].copy(), cmd=cmd[t].copy()) for t in range(T)]


@pytest.mark.parametrize("name", sorted(RG.hcp_tick_cases()))
def test_drop_in_planner_class_follows_the_reference_planner_tick_by_tick(name):
    """HomotopyClassPlannerAmd (teb_local_planner_amd/host/teb_amd_hcp_backend.cpp: a HomotopyClassPlanner subclass whose plan() runs on the
    MI355X) and the reference's HomotopyClassPlanner, each driven through the same sequence of plan() calls on its own planner object, in
    one process, on identical TebConfig / ObstContainer / ViaPointContainer objects. Same candidates in the same order, same best
    candidate and initial-plan candidate every tick; costs 1e-6 relative, states 2e-5, the inherited getVelocityCommand 1e-5."""
    case = RG.hcp_tick_cases()[name]
    ref = _hcp_ticks(0, case)
    amd = _hcp_ticks(1, case)
    for t, (r, a) in enumerate(zip(ref, amd)):
        assert len(a["bands"]) == len(r["bands"]) and a["best"] == r["best"] and a["initial"] == r["initial"], (t, a["best"], r["best"])
        assert np.abs(a["costs"] - r["costs"]).max() <= 1e-6 * np.abs(r["costs"]).max()
        for k, (u, v) in enumerate(zip(a["bands"], r["bands"])):
            assert len(u[0]) == len(v[0]), (t, k)
            assert max(np.abs(x - y).max() for x, y in zip(u, v)) <= 2e-5, (t, k)
        assert a["cmd"][0] == r["cmd"][0] == 1 and np.abs(a["cmd"][1:] - r["cmd"][1:]).max() <= 1e-5


@pytest.mark.parametrize("name", sorted(RG.hcp_tick_cases()))
def test_drop_in_planner_class_sharded_mode_on_a_world_of_one(name):
    """HomotopyClassPlannerAmd::setCommunicator (the C++ drop-in can shard its candidate classes over GPUs): on a
    communicator of ONE rank every RCCL call of the path runs - selection all-gather, status all-gather, broadcast of the winner's band
    (RCCL refuses two ranks on one device; the reduction over several ranks is covered by tests/test_abi.py and the gloo tests) - and the
    ticks come out exactly as without a communicator: same candidates, same best index, same bands and costs, same velocity command."""
    case = RG.hcp_tick_cases()[name]
    one = _hcp_ticks(1, case)
    sharded = _hcp_ticks(2, case)
    for t, (r, a) in enumerate(zip(one, sharded)):
        assert len(a["bands"]) == len(r["bands"]) and a["best"] == r["best"] and a["initial"] == r["initial"], (t, a["best"], r["best"])
        np.testing.assert_array_equal(a["costs"], r["costs"])
        for u, v in zip(a["bands"], r["bands"]):
            for x, y in zip(u, v):
                np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(a["cmd"], r["cmd"])


@pytest.mark.parametrize("seed", range(30))
def test_drop_in_planner_class_on_random_scenes(seed):
    """Three plan() ticks on random scenes (tests/random_explore_cases.py: every obstacle class, both graph types, random planner
    parameters, optionally an initial plan with a via-point every tick): HomotopyClassPlannerAmd (g2o-numeric Jacobians, like the
    reference) against the reference's planner.
    Candidate count, order and poses every tick; the best candidate whenever the reference's own choice is not a near-tie."""
    from random_explore_cases import random_explore_case
    base = random_explore_case(seed)
    cfg = base["cfg"]
    cfg.optim.no_inner_iterations = 3; cfg.optim.no_outer_iterations = 2
    st, gl = np.array(base["start"]), np.array(base["goal"])
    d = (gl[:2] - st[:2]) / np.linalg.norm(gl[:2] - st[:2])
    starts = [[st[0] + 0.1 * k * d[0], st[1] + 0.1 * k * d[1], st[2]] for k in range(3)]
    case = dict(cfg=cfg, obst=base["obst"], starts=starts, goals=[list(gl)] * 3, start_vels=[[0.0, 0, 0], [0.2, 0, 0], [0.25, 0, 0.02]],
                via=base.get("via"))
    if base.get("initial_plan") is not None:
        px, py, pyaw = base["initial_plan"]
        case["plans"] = []
        for k in range(3):      # the global plan re-anchored at the current start
            x = px.copy(); y = py.copy(); yaw = pyaw.copy()
            x[0], y[0] = starts[k][0], starts[k][1]
            case["plans"].append((x, y, yaw))
    ref = _hcp_ticks(0, case, slots=10)
    amd = _hcp_ticks(1, case, slots=10, jacobian_mode=_abi.JACOBIAN_G2O_NUMERIC)   # the reference's own linearisation scheme on the device
    # yardstick: the reference against itself with the start poses moved by 1e-9 m - candidates that start inside the obstacle penalty
    # zone (key points lie 0.25 m from point obstacles) amplify such noise by many orders of magnitude; a band is compared as tightly
    # as the reference reproduces itself (x 20, at least 2e-5), and not at all from the tick on where it does not reproduce itself
    moved = dict(case); moved["starts"] = [[s_[0] + 1e-9, s_[1], s_[2]] for s_ in starts]
    if case.get("plans"):
        moved["plans"] = [(np.concatenate([[x[0] + 1e-9], x[1:]]), y, yaw) for (x, y, yaw) in case["plans"]]
    ref2 = _hcp_ticks(0, moved, slots=10)
    compared = 0
    for t, (r, a, r2) in enumerate(zip(ref, amd, ref2)):
        if len(r2["bands"]) != len(r["bands"]) or any(len(u[0]) != len(v[0]) for u, v in zip(r["bands"], r2["bands"])):
            break                                   # the reference's own tick bifurcates under 1e-9 m: later ticks are not comparable
        noise = [max(np.abs(x - y).max() for x, y in zip(u, v)) for u, v in zip(r["bands"], r2["bands"])]
        if max(noise, default=0.0) > 1e-3:
            break
        assert len(a["bands"]) == len(r["bands"]) and a["initial"] == r["initial"], (t, len(a["bands"]), len(r["bands"]))
        for k, (u, v) in enumerate(zip(a["bands"], r["bands"])):
            assert len(u[0]) == len(v[0]), (t, k)
            tol = max(2e-5, 20 * noise[k])
            assert max(np.abs(x - y).max() for x, y in zip(u, v)) <= tol, (t, k, max(np.abs(x - y).max() for x, y in zip(u, v)), tol)
            compared += 1
        c = np.sort(r["costs"])
        near_tie = len(c) > 1 and (c[1] - c[0]) <= 1e-3 * abs(c[0])
        if not near_tie and r2["best"] == r["best"]:
            assert a["best"] == r["best"], (t, a["best"], r["best"], r["costs"])


# ---- the drop-in classes behind the pointer the plugin holds (PlannerInterfacePtr planner_, src/teb_local_planner_ros.cpp:120-125) --------
def _plan_ticks(which, cfg, obst, starts, goals, start_vels=None, free_goal_vel=False, overload=0, plans=None, via=None, hcp=False,
                stride=512, jacobian_mode=_abi.JACOBIAN_G2O_NUMERIC, retune_chi2=-1.0):
    """oracle/ref_shim/backend_check_plan.cpp: n ticks of plan() + hasDiverged() + getVelocityCommand() on ONE planner object, every call
    through a PlannerInterface pointer. which: 0 TebOptimalPlanner (reference), 1 TebOptimalPlannerAmd, 2 HomotopyClassPlanner, 3
    HomotopyClassPlannerAmd."""
    L = _lib()
    c = cfg.to_c(); p = cfg.hcp_params()
    st = _abi.f64(starts).reshape(-1, 3); gl = _abi.f64(goals).reshape(-1, 3)
    T = len(st)
    sv = None if start_vels is None else _abi.f64(start_vels).reshape(T, 3)
    out = _abi.TebBatchHost(T, stride)
    obs = out.c_struct()
    ok = np.zeros(T, np.int32); div = np.zeros(T + 1, np.int32); g2o = np.zeros(T, np.int32); lm = np.zeros(T, np.int32); cmd = np.zeros((T, 4))
    plans = plans or [None] * T
    off = np.zeros(T + 1, np.int32)
    for t in range(T):
        off[t + 1] = off[t] + (0 if plans[t] is None else len(plans[t][0]))
    cat = lambda k: _abi.f64(np.concatenate([np.asarray(pl[k], np.float64) for pl in plans if pl is not None] or [np.zeros(1)]))
    px, py, pyaw = cat(0), cat(1), cat(2)
    via = via or []
    vx = _abi.f64([v[0] for v in via] or [0.0]); vy = _abi.f64([v[1] for v in via] or [0.0])
    P = lambda a: C.cast(_abi._ptr(a, C.c_double), C.c_void_p) if a is not None else None
    I = lambda a: C.cast(_abi._ptr(a, C.c_int32), C.c_void_p)
    vp = lambda x: C.cast(C.pointer(x), C.c_void_p)
    f = L.backend_check_plan_ticks
    f.argtypes = [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_int] + \
        [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 6 + [C.c_double]
    rc = f(which, vp(c), vp(p) if hcp else None, vp(obst.freeze()), len(via), P(vx), P(vy), T, P(st), P(gl), P(sv), int(free_goal_vel),
           int(overload), I(off), P(px), P(py), P(pyaw), int(jacobian_mode), vp(obs), I(ok), I(div), I(g2o), I(lm), P(cmd), float(retune_chi2))
    assert rc == 0, rc
    return dict(bands=[out.get_teb(t) for t in range(T)], ok=ok, diverged=div[:T], diverged_retuned=int(div[T]), g2o_iters=g2o, lm_iters=lm,
                cmd=cmd)


def _moving_starts(x0, y0, th0, n, step=0.15):
    return [[x0 + step * k, y0 + 0.01 * k, th0] for k in range(n)]


def _single_planner_scene(name):
    """(cfg, obst, start, goal, plan) of BASELINE C1 (test_optim_node: 3 obstacles, 8 m) and C2 (single TEB, 200 poses, 100 obstacles)."""
    from teb_local_planner_amd import scenes
    cfg, obst, via, batch = scenes.scene_c1() if name == "c1" else scenes.scene_c2(stride=512)
    x, y, th, _ = batch.get_teb(0)
    return cfg, obst, [x[0], y[0], th[0]], [x[-1], y[-1], th[-1]], (x, y, th)


def _compare_ticks(ref, amd, tol, cmd_tol):
    worst = 0.0
    for t, (u, v) in enumerate(zip(amd["bands"], ref["bands"])):
        assert len(u[0]) == len(v[0]), (t, len(u[0]), len(v[0]))
        d = max(np.abs(a - b).max(initial=0) for a, b in zip(u, v))
        worst = max(worst, d)
        assert d <= tol, (t, d, tol)
    np.testing.assert_array_equal(amd["ok"], ref["ok"])
    np.testing.assert_array_equal(amd["cmd"][:, 0], ref["cmd"][:, 0])
    assert np.abs(amd["cmd"][:, 1:] - ref["cmd"][:, 1:]).max() <= cmd_tol
    return worst


@pytest.mark.parametrize("overload", [0, 1, 2])
@pytest.mark.parametrize("scene", ["c1", "c2"])
def test_single_planner_behind_the_interface_pointer_plans_on_the_device(scene, overload):
    """VERDICT r04 item 1. PlannerInterfacePtr p(new TebOptimalPlannerAmd(..)); p->plan(..) for five ticks (cold start, then warm starts
    from a moving robot pose) in each of the three overloads of planner_interface.h:99-125, against the reference's TebOptimalPlanner
    driven the same way. TebOptimalPlanner::optimizeTEB is not virtual: the override of plan() is what keeps the drop-in off g2o - the
    shim's SparseOptimizer counts the LM iterations this thread ran (zero for the drop-in, > 0 for the reference), lastLmIterations() says
    the device ran them. Bands, plan()'s return value and the inherited getVelocityCommand agree with the reference planner every tick."""
    cfg, obst, start, goal, path = _single_planner_scene(scene)
    T = 5
    starts = _moving_starts(*start, T); goals = [goal] * T
    vels = [[0.0, 0, 0], [0.2, 0, 0.0], [0.3, 0, 0.02], [0.35, 0, 0.02], [0.4, 0, 0.0]]
    plans = None
    if overload == 2:      # the global plan re-anchored at the robot pose, as the ROS adapter hands it over
        x, y, th = path
        step = max(1, len(x) // 40)
        plans = []
        for k in range(T):
            xs = x[::step].copy(); ys = y[::step].copy(); ys_th = th[::step].copy()
            xs[0], ys[0], ys_th[0] = starts[k]
            xs[-1], ys[-1], ys_th[-1] = goal
            plans.append((xs, ys, ys_th))
    kw = dict(start_vels=vels, overload=overload, plans=plans)
    ref = _plan_ticks(0, cfg, obst, starts, goals, **kw)
    amd = _plan_ticks(1, cfg, obst, starts, goals, **kw)
    assert (ref["g2o_iters"] > 0).all() and (ref["lm_iters"] == -1).all()
    assert (amd["g2o_iters"] == 0).all(), amd["g2o_iters"]            # not one LM iteration on the CPU
    assert (amd["lm_iters"] > 0).all(), amd["lm_iters"]               # .. they ran on the device
    assert amd["ok"].all()
    worst = _compare_ticks(ref, amd, tol=2e-5, cmd_tol=1e-5)
    print("plan() through PlannerInterface*, %s overload %d: worst state distance to the reference planner over %d ticks %.2e" % (scene, overload, T, worst))


def test_single_planner_free_goal_velocity_and_goal_jump():
    """free_goal_vel reaches the band (not through the tf::Pose overload: the reference drops the flag there, src/optimal_planner.cpp:283-288,
    and so does the drop-in), and a goal that jumps beyond force_reinit_new_goal_dist re-initialises the band on both sides."""
    cfg, obst, start, goal, _ = _single_planner_scene("c1")
    starts = _moving_starts(*start, 4)
    goals = [goal, goal, [goal[0] - 0.5, goal[1] + 2.5, 0.6], [goal[0] - 0.5, goal[1] + 2.5, 0.6]]
    for overload in (0, 1):
        ref = _plan_ticks(0, cfg, obst, starts, goals, start_vels=[[0.1, 0, 0]] * 4, free_goal_vel=True, overload=overload)
        amd = _plan_ticks(1, cfg, obst, starts, goals, start_vels=[[0.1, 0, 0]] * 4, free_goal_vel=True, overload=overload)
        assert (amd["g2o_iters"] == 0).all() and (amd["lm_iters"] > 0).all()
        _compare_ticks(ref, amd, tol=2e-5, cmd_tol=1e-5)
    fixed = _plan_ticks(1, cfg, obst, starts[:1], goals[:1], start_vels=[[0.1, 0, 0]], free_goal_vel=False, overload=0)
    free = _plan_ticks(1, cfg, obst, starts[:1], goals[:1], start_vels=[[0.1, 0, 0]], free_goal_vel=True, overload=0)
    dropped = _plan_ticks(1, cfg, obst, starts[:1], goals[:1], start_vels=[[0.1, 0, 0]], free_goal_vel=True, overload=1)
    assert np.abs(free["bands"][0][3][-3:] - fixed["bands"][0][3][-3:]).max() > 1e-6       # the flag changes the end of the band
    for a, b in zip(dropped["bands"][0], fixed["bands"][0]):
        np.testing.assert_array_equal(a, b)                                                 # .. and is dropped by the tf::Pose overload


@pytest.mark.parametrize("which_pair", [(0, 1), (2, 3)])
def test_has_diverged_through_the_interface_pointer(which_pair):
    """VERDICT r04 item 2. divergence_detection_enable with a threshold every optimisation exceeds: planner_->hasDiverged()
    (src/teb_local_planner_ros.cpp:374) of the drop-in classes answers like the reference's planners tick by tick - from the statistics the
    launch returned, not from the g2o object that never ran - incl. g2o's habit of sizing batchStatistics() to the requested iteration
    count (a band whose last optimize() stopped early reads chi2 = 0 and is "not diverged"); the rule reads the LIVE configuration: raising
    the threshold afterwards clears the flag on both sides. Same through HomotopyClassPlanner, whose hasDiverged forwards to best_teb_."""
    hcp = which_pair[0] == 2
    if hcp:
        case = RG.hcp_tick_cases()["keypoint_2d_4_ticks"]
        cfg, obst, starts, goals, vels = case["cfg"], case["obst"], case["starts"], case["goals"], case["start_vels"]
    else:
        cfg, obst, start, goal, _ = _single_planner_scene("c1")
        starts = _moving_starts(*start, 4); goals = [goal] * 4; vels = [[0.0, 0, 0], [0.2, 0, 0], [0.3, 0, 0], [0.3, 0, 0]]
    cfg.recovery.divergence_detection_enable = True
    seen = set()
    # (the reference's field is an int, teb_config.h:228: thresholds are whole numbers in its range)
    for thr in (0, 5, 1000000):
        cfg.recovery.divergence_detection_max_chi_squared = thr
        kw = dict(start_vels=vels, hcp=hcp, retune_chi2=2.0e9 if thr < 1000000 else 0.0)
        ref = _plan_ticks(which_pair[0], cfg, obst, starts, goals, **kw)
        amd = _plan_ticks(which_pair[1], cfg, obst, starts, goals, **kw)
        assert (amd["g2o_iters"] == 0).all() and (ref["g2o_iters"] > 0).all()
        np.testing.assert_array_equal(amd["diverged"], ref["diverged"])
        assert amd["diverged_retuned"] == ref["diverged_retuned"]
        seen.update(int(d) for d in ref["diverged"]); seen.add(10 + ref["diverged_retuned"])
        if thr == 1000000:
            assert not ref["diverged"].any()
    assert seen >= {0, 1, 10, 11}, seen      # both answers occurred, before and after the threshold moved
    cfg.recovery.divergence_detection_enable = False
    cfg.recovery.divergence_detection_max_chi_squared = 0
    off = _plan_ticks(which_pair[1], cfg, obst, starts, goals, start_vels=vels, hcp=hcp)
    assert not off["diverged"].any()


def test_has_diverged_after_an_optimisation_that_stopped_early():
    """g2o sizes batchStatistics() to the REQUESTED iteration count and fills one entry per executed iteration: when the LM loop of the last
    optimize() call terminates early, .back().chi2 is still 0 and TebOptimalPlanner::hasDiverged (src/optimal_planner.cpp:1029-1038) says
    "no" whatever chi2 was reached - with 40 inner iterations on an obstacle-free 1 m band LM stops after 20 - 30. The drop-in answers the
    same (the kernel reports the iteration count of the band's last optimize() call); with the default 5 iterations both say "yes"."""
    from teb_local_planner_amd.config import TebConfig
    obst = _abi.ObstacleTable()
    starts = [[0.02 * k, 0, 0] for k in range(4)]; goals = [[1.0, 0, 0]] * 4; vels = [[0.4, 0, 0]] * 4
    for inner, outer, expect in ((40, 1, 0), (5, 4, 1)):
        cfg = TebConfig()
        cfg.recovery.divergence_detection_enable = True
        cfg.recovery.divergence_detection_max_chi_squared = 0
        cfg.optim.no_inner_iterations = inner; cfg.optim.no_outer_iterations = outer
        ref = _plan_ticks(0, cfg, obst, starts, goals, start_vels=vels)
        amd = _plan_ticks(1, cfg, obst, starts, goals, start_vels=vels)
        assert (amd["g2o_iters"] == 0).all()
        if not expect:
            assert (ref["g2o_iters"] < inner * outer).all() and (amd["lm_iters"] < inner * outer).all()    # both stopped early
        assert (ref["diverged"] == expect).all(), ref["diverged"]
        np.testing.assert_array_equal(amd["diverged"], ref["diverged"])


@pytest.mark.parametrize("name", sorted(RG.hcp_tick_cases()))
def test_homotopy_planner_behind_the_interface_pointer(name):
    """HomotopyClassPlannerAmd held as PlannerInterfacePtr: plan(), hasDiverged(), getVelocityCommand() through the base class; no LM
    iteration on the CPU, the best band follows the reference planner's."""
    case = RG.hcp_tick_cases()[name]
    kw = dict(start_vels=case.get("start_vels"), plans=case.get("plans"), via=case.get("via"), hcp=True, overload=2 if case.get("plans") else 0,
              jacobian_mode=_abi.JACOBIAN_ANALYTIC)
    ref = _plan_ticks(2, case["cfg"], case["obst"], case["starts"], case["goals"], **kw)
    amd = _plan_ticks(3, case["cfg"], case["obst"], case["starts"], case["goals"], **kw)
    assert (amd["g2o_iters"] == 0).all() and (amd["lm_iters"] > 0).all() and (ref["g2o_iters"] > 0).all()
    _compare_ticks(ref, amd, tol=2e-5, cmd_tol=1e-5)
