"""SURVEY section 8(f) row f3, candidate generation: HomotopyClassPlanner::exploreEquivalenceClassesAndInitTebs through the C-ABI
(teb_amd_compute_h_signatures / teb_amd_filter_equivalence_classes / teb_amd_compact_bands / teb_amd_explore_candidates; kernels in
teb_graph.hpp) against the CPU oracle - which is bit-equal to the reference's own graph_search.cpp + homotopy_class_planner.hpp on
these cases (tests/test_reference_pinning.py) - and against the committed vectors of the reference code.

What is compared, and how tightly:
  * graph vertices: the same IEEE expressions evaluated on the host side of the library: <= 1e-12;
  * graph edges (N^2 * M collision tests on the device, no transcendental function): identical;
  * number of bands and every band: the device evaluates atan2 / sqrt where the reference does: <= 1e-12 absolute;
  * the order in which classes are discovered (= band order): identical."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, HERE)
import make_ref_golden as RG  # noqa: E402
from test_reference_pinning import renew_on_host, stale_signature, plan_as_seen, kept_via_flags  # noqa: E402

from teb_local_planner_amd import planner, _abi, scenes  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-12


def _make(case, max_tebs=16, stride=256):
    cfg, obst, batch = case["cfg"], case["obst"], case["batch"]
    via = case.get("via") or []
    s = planner.TebBatchSolver(cfg, max_tebs, stride, max(len(obst), 1), max(len(obst.vert_x), 1), max(len(via), 1))
    s.set_obstacles(obst)
    s.set_via_points(via)
    if case.get("stale_initial_band") is not None:   # initial_plan_eq_class_ of an earlier tick: a one-candidate exploration with that
        x, y, th, _ = case["stale_initial_band"]     # band's poses as the plan leaves its class in the handle and draws no samples
        p1 = cfg.hcp_params(); p1.max_number_classes = 1
        r0 = s.explore_candidates(case["start"], case["goal"], params=p1, initial_plan=(x, y, th))
        assert r0["n_total"] == 1 and r0["initial_plan_teb"] == 0 and r0["n_vertices"] == 0
        s.compact_bands(np.zeros(1, np.int32))
    if batch is not None:
        if case.get("via_enabled") is not None:
            batch = batch.copy(); batch.via_points_enabled[:] = case["via_enabled"]
        s.upload(batch)
        opt = case.get("optimized")
        if opt is None:
            opt = [1] * batch.count
        s.set_optimized_flags(opt)     # TebOptimalPlanner::optimized_ of bands that came from the host
    return s


def _renew_on_device(s, case):
    """renewAndAnalyzeOldTebs with the C-ABI pieces; returns the new index of the best band."""
    cfg = case["cfg"]
    if case["batch"] is None:
        return -1
    s.h_signatures(cfg.hcp.h_signature_prescaler)
    keep, _, _ = s.filter_equivalence_classes(cfg.hcp.h_signature_threshold, case["best"], cfg.hcp.max_number_plans_in_current_class)
    if cfg.hcp.delete_detours_backwards:
        keep = s.filter_detours(keep, case["best"])
    _, best = s.compact_bands(keep, case["best"])
    return best


def _explore(s, case, best, ref=None):
    plan = plan_as_seen(case, ref) if ref is not None else None
    return s.explore_candidates(case["start"], case["goal"], dist_to_obst=case.get("dist_to_obst"), start_vel=case.get("start_vel"),
                                free_goal_vel=case.get("free_goal_vel", False), best=best, initial_plan=plan)


def _bands(s, stride=256):
    b = _abi.TebBatchHost(max(s.count, 1), stride)
    if s.count:
        s.download(b)
    return [b.get_teb(k) for k in range(s.count)]


@pytest.mark.parametrize("name", sorted(RG.explore_cases()))
def test_explore_candidates_matches_oracle_and_reference_vectors(oracle, name):
    case = RG.explore_cases()[name]
    g = np.load(os.path.join(HERE, "golden", "ref_f3_explore.npz"))
    ref = {k[len(name) + 2:]: g[k] for k in g.files if k.startswith(name + "__")}
    s = _make(case)
    if case.get("stale_band") is not None:   # a best band of an earlier tick that is gone (goal jump): its class stays in the handle
        host = _abi.TebBatchHost(1, 256)
        host.set_teb(0, *case["stale_band"])
        s.upload(host)
        s.h_signatures(case["cfg"].hcp.h_signature_prescaler)
        s.filter_equivalence_classes(case["cfg"].hcp.h_signature_threshold, 0, case["cfg"].hcp.max_number_plans_in_current_class)
        s.compact_bands(np.zeros(1, np.int32))          # tebs_.clear()
        assert s.count == 0
    best = _renew_on_device(s, case)
    if case.get("skip_draws"):   # "second call": the handle's generator has produced the samples of an earlier graph already
        p = case["cfg"].hcp_params()
        assert case["skip_draws"] == 2 * p.roadmap_graph_no_samples
        _explore(s, case, best)
        s.compact_bands(np.zeros(s.count, np.int32))
        assert s.count == 0
    r = _explore(s, case, best, ref)
    b, n_tebs, obest = renew_on_host(oracle, case)
    assert obest == best
    o = oracle.explore_candidates(case["cfg"], case["obst"], b, n_tebs, obest, case["start"], case["goal"],
                                  skip_draws=case.get("skip_draws", 0), dist_to_obst=case.get("dist_to_obst"),
                                  stale_best_sig=stale_signature(oracle, case), initial_plan=plan_as_seen(case, ref),
                                  stale_initial_sig=stale_signature(oracle, case, "stale_initial_band"),
                                  via_enabled=kept_via_flags(oracle, case, b.count))
    assert r["n_total"] == o["n_total"] == int(ref["n_total"]) == s.count
    assert r["initial_plan_teb"] == o["initial_plan_teb"] == int(ref["initial_plan_teb"])
    if case.get("via"):
        np.testing.assert_array_equal(s.band_flags()[0], ref["via_enabled"])
        np.testing.assert_array_equal(o["via_enabled"][:o["n_total"]], ref["via_enabled"])
    assert r["n_vertices"] == len(o["vertices"])
    if r["n_total"] > n_tebs or name == "max_two_classes":
        assert r["n_paths"] <= o["n_paths"] or o["n_paths"] == 0   # the device may stop inside a chunk exactly where the oracle stops
    V, A = s.exploration_graph()
    if len(o["vertices"]):
        assert np.abs(V - o["vertices"]).max() <= TOL
        assert np.abs(V - ref["vertices"]).max() <= TOL
        np.testing.assert_array_equal(A, ref["adjacency"])
    got = _bands(s)
    for k in range(r["n_total"]):
        want = o["batch"].get_teb(k)
        for u, v, w in zip(got[k], want, RG.unpack(ref, k)):
            assert len(u) == len(v) == len(w)
            assert np.abs(u - v).max(initial=0) <= TOL and np.abs(u - w).max(initial=0) <= TOL
    s.close()


def test_new_bands_carry_start_velocity_and_free_goal_flags(oracle):
    """setVelocityStart / setVelocityGoalFree of addAndInitNewTeb, observed through the optimiser: the new bands optimise like oracle
    bands with the same flags, and differently from bands with the default (fixed, zero) velocities."""
    case = RG.explore_cases()["backwards_start_velocity_free_goal"]
    cfg = case["cfg"]
    s = _make(case)
    r = _explore(s, case, -1)
    n = r["n_total"]
    assert n >= 2
    host = _abi.TebBatchHost(n, 256)
    s.download(host)
    host.vel_start[:] = case["start_vel"]; host.has_vel_goal[:] = 0
    s.optimize(cfg.optim.no_inner_iterations, cfg.optim.no_outer_iterations)
    s.synchronize()
    got = _abi.TebBatchHost(n, 256)
    s.download(got)
    want, _ = oracle.optimize_batch(cfg, case["obst"], [], host)
    plain = host.copy(); plain.vel_start[:] = 0; plain.has_vel_goal[:] = 1
    other, _ = oracle.optimize_batch(cfg, case["obst"], [], plain)
    for k in range(n):
        assert got.n[k] == want.n[k]
        m = int(got.n[k])
        d_same = max(np.abs(got.x[k, :m] - want.x[k, :m]).max(), np.abs(got.dt[k, :m - 1] - want.dt[k, :m - 1]).max())
        assert d_same <= 1e-6, (k, d_same)
        if other.n[k] == want.n[k]:
            assert np.abs(other.dt[k, :m - 1] - want.dt[k, :m - 1]).max() > 100 * max(d_same, 1e-9)
    s.close()


def test_compact_bands_moves_attributes_and_keeps_reference_order(oracle):
    cfg, obst, via, batch = scenes.scene_small_mixed(B=6, n=24, seed=3, with_via=False)
    batch.vel_start[:] = np.arange(18).reshape(6, 3) * 0.01
    s = planner.make_solver(cfg, obst, [], batch, max_tebs=8)
    keep = np.array([1, 0, 1, 1, 0, 1], np.int32)
    nk, nb = s.compact_bands(keep, best=3)
    assert (nk, nb) == (4, 0) and s.count == 4
    order = [3, 2, 0, 5]          # iter_swap(first, best) then erase: [3, 1, 2, 0, 4, 5] minus the dropped ones
    got = _bands(s, batch.stride)
    for k, b in enumerate(order):
        for u, v in zip(got[k], batch.get_teb(b)):
            np.testing.assert_array_equal(u, v)
    # attributes travel with the band: optimise and compare with the oracle on the reordered host batch
    host = _abi.TebBatchHost(4, batch.stride)
    for k, b in enumerate(order):
        host.set_teb(k, *batch.get_teb(b)); host.vel_start[k] = batch.vel_start[b]
    s.optimize(cfg.optim.no_inner_iterations, cfg.optim.no_outer_iterations)
    s.synchronize()
    dev = _abi.TebBatchHost(4, batch.stride)
    s.download(dev)
    want, _ = oracle.optimize_batch(cfg, obst, [], host)
    for k in range(4):
        m = int(want.n[k])
        assert dev.n[k] == m and np.abs(dev.x[k, :m] - want.x[k, :m]).max() <= 1e-6
    nk, nb = s.compact_bands(np.array([0, 1, 0, 0], np.int32), best=0)
    assert (nk, nb) == (1, -1)
    s.close()


def test_large_keypoint_graph_all_pairs_on_the_device(oracle):
    """160 point obstacles -> up to 322 vertices (a band holds one pose per path vertex: max_poses bounds the graph), 10^5 ordered pairs
    x 160 obstacles on the device; adjacency identical to the oracle's."""
    rng = np.random.default_rng(12)
    cfg = RG.explore_cases()["keypoint_points_2d"]["cfg"]
    cfg.hcp.max_number_classes = 3
    ob = _abi.ObstacleTable()
    for _ in range(160):
        ob.add_point(rng.uniform(0.5, 19.5), rng.uniform(-4, 4))
    case = dict(cfg=cfg, obst=ob, batch=None, best=-1, start=[0, 0, 0], goal=[20, 0, 0])
    s = _make(case, max_tebs=4, stride=336)
    r = s.explore_candidates(case["start"], case["goal"], max_paths=256)
    b = _abi.TebBatchHost(4, 336)
    o = oracle.explore_candidates(cfg, ob, b, 0, -1, case["start"], case["goal"], vcap=1024, max_paths=256)
    V, A = s.exploration_graph()
    assert r["n_vertices"] == len(o["vertices"]) == len(V) > 250
    assert np.abs(V - o["vertices"]).max() <= TOL
    want = np.zeros_like(A)
    for i, row in enumerate(o["adjacency"]):
        want[i, row] = 1
    np.testing.assert_array_equal(A, want)
    assert A.sum() > 100
    s.close()


def test_optimized_flag_follows_optimizeTEB(oracle):
    """optimized_ = false at the start of optimizeTEB, true after the first completed outer iteration (src/optimal_planner.cpp:189, 220);
    uploads and new candidates start with false; optimization_activate = false leaves it untouched."""
    cfg, obst, via, batch = scenes.scene_small_mixed(B=3, n=24, seed=3, with_via=False)
    cfg.trajectory.teb_autosize = False
    batch.n[2] = 2                                   # fewer poses than min_samples: optimizeGraph fails in the first outer iteration
    s = planner.make_solver(cfg, obst, [], batch, max_tebs=6)
    np.testing.assert_array_equal(s.optimized_flags(), [0, 0, 0])
    s.optimize(cfg.optim.no_inner_iterations, cfg.optim.no_outer_iterations); s.synchronize()
    np.testing.assert_array_equal(s.optimized_flags(), [1, 1, 0])
    np.testing.assert_array_equal(s.results().status, [_abi.TEB_OK, _abi.TEB_OK, _abi.TEB_FAILED])
    cfg.optim.optimization_activate = False
    s.set_config(cfg)
    s.optimize(cfg.optim.no_inner_iterations, cfg.optim.no_outer_iterations); s.synchronize()
    np.testing.assert_array_equal(s.optimized_flags(), [1, 1, 0])      # early return before optimized_ = false
    cfg.optim.optimization_activate = True
    s.set_config(cfg)
    s.close()
    case = RG.explore_cases()["roadmap_points_3d"]
    s = _make(case)
    r = _explore(s, case, -1)
    assert r["n_total"] >= 3 and not s.optimized_flags().any()      # new TebOptimalPlanner: optimized_(false)
    s.optimize(2, 2); s.synchronize()
    assert s.optimized_flags().all()
    s.close()


@pytest.mark.parametrize("name", sorted(RG.hcp_tick_cases()))
def test_whole_plan_ticks_match_the_reference_planner(name):
    """HomotopyClassPlanner::plan() tick after tick on device-resident bands (updateAllTEBs, renewAndAnalyzeOldTebs with detour
    deletion, graph exploration, optimizeAllTEBs, selectBestTeb) against the vectors of the reference's own HomotopyClassPlanner run
    on the same inputs (oracle/ref_shim/ref_hcp_driver.cpp: reference classes, g2o's LM restated). Same number and order of bands,
    same best band in every tick; states within 2e-5 (the optimiser parity of tests/test_gpu_parity.py carried over four ticks),
    costs within 1e-6 relative."""
    case = RG.hcp_tick_cases()[name]
    g = np.load(os.path.join(HERE, "golden", "ref_f3_hcp_ticks.npz"))
    hcp = planner.HomotopyClassPlanner(case["cfg"], case["obst"], case.get("via") or [], None, max_tebs=8, max_poses=256)
    for t, (st, gl) in enumerate(zip(case["starts"], case["goals"])):
        sv = None if case["start_vels"] is None else case["start_vels"][t]
        pre = "%s__%d__" % (name, t)
        ref = {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}
        plan = None
        if case.get("plans"):            # plan(initial_plan, ...): yaw as the reference reads it back from the pose messages
            plan = (case["plans"][t][0], case["plans"][t][1], ref["plan_yaw_seen"])
        assert hcp.plan(st, gl, sv, initial_plan=plan)
        bands = hcp.bands()
        assert len(bands) == len(ref["n"]) and hcp.best_teb_ == int(ref["best"]), (t, len(bands), hcp.best_teb_)
        assert hcp.initial_plan_teb_ == int(ref["initial_plan_teb"])
        cost = np.array(hcp.results().cost[:len(bands)])
        assert np.abs(cost - ref["costs"]).max() <= 1e-6 * np.abs(ref["costs"]).max()
        for k, band in enumerate(bands):
            want = RG.unpack(ref, k)
            assert len(band[0]) == len(want[0]), (t, k)
            assert max(np.abs(a - b).max() for a, b in zip(band, want)) <= 2e-5, (t, k)
        ok, vx, vy, om = hcp.getVelocityCommand()
        assert ok and np.isfinite([vx, vy, om]).all()
    hcp.solver.close()


@pytest.mark.parametrize("seed", range(40))
def test_randomized_candidate_generation_matches_oracle(oracle, seed):
    """Random scenes (tests/random_explore_cases.py; the oracle is bit-equal to the reference's code on them,
    tests/test_reference_pinning.py): same graph, same bands in the same order, same initial-plan index and via-point flags."""
    from random_explore_cases import random_explore_case
    case = random_explore_case(seed)
    plan = case.get("initial_plan")
    b = _abi.TebBatchHost(16, 256)
    o = oracle.explore_candidates(case["cfg"], case["obst"], b, 0, -1, case["start"], case["goal"], dist_to_obst=case["dist_to_obst"],
                                  initial_plan=plan, via_enabled=np.zeros(16, np.int32) if case.get("via") else None, max_paths=20000)
    if o["n_paths"] >= 20000:
        pytest.skip("path enumeration bounded")
    s = _make(case)
    r = s.explore_candidates(case["start"], case["goal"], dist_to_obst=case["dist_to_obst"], start_vel=case.get("start_vel"),
                             free_goal_vel=case.get("free_goal_vel", False), initial_plan=plan)
    assert r["n_total"] == o["n_total"] and r["initial_plan_teb"] == o["initial_plan_teb"] and r["n_vertices"] == len(o["vertices"])
    V, A = s.exploration_graph()
    if len(V):
        assert np.abs(V - o["vertices"]).max() <= TOL
        want = np.zeros_like(A)
        for i, row in enumerate(o["adjacency"]):
            want[i, row] = 1
        np.testing.assert_array_equal(A, want)
    got = _bands(s)
    for k in range(r["n_total"]):
        for u, v in zip(got[k], o["batch"].get_teb(k)):
            assert len(u) == len(v) and np.abs(u - v).max(initial=0) <= TOL
    if r["n_total"]:
        ve, hvs, hvg = s.band_flags()
        if case.get("via"):
            np.testing.assert_array_equal(ve, o["via_enabled"][:o["n_total"]])
        assert hvs.all() and (not hvg.any() if case.get("free_goal_vel") else hvg.all())
    s.close()


@pytest.mark.parametrize("seed", range(30))
def test_whole_plan_ticks_on_random_scenes_against_the_reference_planner(seed):
    """Three device-resident plan() ticks on random scenes (every obstacle class, both graph types, random planner parameters, optionally
    an initial plan with a via-point every tick) against the reference's own HomotopyClassPlanner run live through oracle/_ref (built where
    /root/reference exists; the binary travels). Device in the reference's linearisation mode (g2o-numeric). A band is compared as tightly
    as the reference reproduces itself when its start poses move by 1e-9 m (x 20, at least 2e-5), and not at all from the tick on where it
    does not reproduce itself; the best candidate is compared unless the reference's own choice is a near-tie."""
    from random_explore_cases import random_explore_case
    from oracle import ref_py
    if not ref_py.available():
        pytest.skip("oracle/_ref/libteb_ref.so not built")
    base = random_explore_case(seed)
    cfg = base["cfg"]
    cfg.optim.no_inner_iterations = 3; cfg.optim.no_outer_iterations = 2
    cfg.jacobian_mode = _abi.JACOBIAN_G2O_NUMERIC
    st, gl = np.array(base["start"]), np.array(base["goal"])
    d = (gl[:2] - st[:2]) / np.linalg.norm(gl[:2] - st[:2])
    starts = [[st[0] + 0.1 * k * d[0], st[1] + 0.1 * k * d[1], st[2]] for k in range(3)]
    vels = [[0.0, 0, 0], [0.2, 0, 0], [0.25, 0, 0.02]]
    plans = None
    if base.get("initial_plan") is not None:
        px, py, pyaw = base["initial_plan"]
        plans = []
        for k in range(3):
            x = px.copy(); y = py.copy(); x[0], y[0] = starts[k][0], starts[k][1]
            plans.append((x, y, pyaw.copy()))
    via = base.get("via")
    ref = ref_py.hcp_plan_ticks(cfg, base["obst"], starts, [list(gl)] * 3, vels, slots=10, stride=256, plans=plans, via=via)
    moved_plans = None if plans is None else [(np.concatenate([[x[0] + 1e-9], x[1:]]), y, yaw) for (x, y, yaw) in plans]
    ref2 = ref_py.hcp_plan_ticks(cfg, base["obst"], [[s_[0] + 1e-9, s_[1], s_[2]] for s_ in starts], [list(gl)] * 3, vels, slots=10, stride=256,
                                 plans=moved_plans, via=via)
    hcp = planner.HomotopyClassPlanner(cfg, base["obst"], via or [], None, max_tebs=10, max_poses=256)
    for t in range(3):
        r, r2 = ref[t], ref2[t]
        plan = None if plans is None else (plans[t][0], plans[t][1], r["plan_yaw_seen"])
        assert hcp.plan(starts[t], list(gl), vels[t], initial_plan=plan)
        if len(r2["bands"]) != len(r["bands"]) or any(len(u[0]) != len(v[0]) for u, v in zip(r["bands"], r2["bands"])):
            break
        noise = [max(np.abs(x - y).max() for x, y in zip(u, v)) for u, v in zip(r["bands"], r2["bands"])]
        if max(noise, default=0.0) > 1e-3:
            break
        bands = hcp.bands()
        assert len(bands) == len(r["bands"]) and hcp.initial_plan_teb_ == r["initial_plan_teb"], (t, len(bands), len(r["bands"]))
        for k, (u, v) in enumerate(zip(bands, r["bands"])):
            assert len(u[0]) == len(v[0]), (t, k)
            tol = max(2e-5, 20 * noise[k])
            assert max(np.abs(x - y).max() for x, y in zip(u, v)) <= tol, (t, k, max(np.abs(x - y).max() for x, y in zip(u, v)), tol)
        c = np.sort(r["costs"])
        near_tie = len(c) > 1 and (c[1] - c[0]) <= 1e-3 * abs(c[0])
        if not near_tie and r2["best"] == r["best"]:
            assert hcp.best_teb_ == r["best"], (t, hcp.best_teb_, r["best"], r["costs"])
    hcp.solver.close()


def test_argument_errors_are_reported_not_ignored():
    case = RG.explore_cases()["keypoint_points_2d"]
    s = _make(case, max_tebs=2)
    L = planner.lib()
    p = case["cfg"].hcp_params()
    st = _abi.f64(case["start"]); gl = _abi.f64(case["goal"])
    P = lambda a: _abi._ptr(a, C.c_double)
    nt = C.c_int32(0)
    # NULL parameter block / start pose, initial plan announced but not given
    assert L.teb_amd_explore_candidates(s._h, None, P(st), P(gl), 0.5, None, 0, -1, None, 0, C.byref(nt), None, None, 0, None, None, None, None) == _abi.ERR_INVALID_ARG
    assert L.teb_amd_explore_candidates(s._h, C.byref(p), None, P(gl), 0.5, None, 0, -1, None, 0, C.byref(nt), None, None, 0, None, None, None, None) == _abi.ERR_INVALID_ARG
    assert L.teb_amd_explore_candidates(s._h, C.byref(p), P(st), P(gl), 0.5, None, 0, -1, None, 0, C.byref(nt), None, None, 5, None, None, None, None) == _abi.ERR_INVALID_ARG
    assert b"initial plan" in L.teb_amd_last_error()
    assert L.teb_amd_compact_bands(s._h, None, -1, None, None) == _abi.ERR_INVALID_ARG
    assert L.teb_amd_filter_detours(s._h, None, -1, None) == _abi.ERR_INVALID_ARG
    # max_tebs bounds the number of candidates even when max_number_classes is larger; nothing is written past the batch
    r = _explore(s, case, -1)
    assert r["n_total"] == 2 == s.count
    # a path with more vertices than a band has poses is a capacity error, not a truncated band
    t = planner.TebBatchSolver(case["cfg"], 4, 2, 4, 4, 1)
    t.set_obstacles(case["obst"]); t.set_via_points([])
    with pytest.raises(planner.TebAmdError) as e:
        t.explore_candidates(case["start"], case["goal"])
    assert e.value.code == _abi.ERR_CAPACITY
    s.close(); t.close()
