"""ctypes wrapper of oracle/_ref/libteb_ref.so — the reference's own classes behind a C interface (TEST INFRASTRUCTURE).

libteb_ref.so is built by `make -C oracle/ref_shim` from /root/reference's headers and src/{obstacles,timed_elastic_band}.cpp
against stand-in Eigen/g2o/ROS/Boost headers; it exists only where /root/reference exists (or as a prebuilt file)."""
import ctypes as C
import os
import subprocess

import numpy as np

from teb_local_planner_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "_ref", "libteb_ref.so")
_LIB = None


def available():
    return os.path.exists(SO) or os.path.isdir("/root/reference")


def build():
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "ref_shim")])
    return SO


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO):
            build()
        _LIB = C.CDLL(SO)
    return _LIB


def config_default():
    c = _abi.Config()
    assert lib().ref_config_default(C.byref(c)) == 0
    return c


def eval_edges(cfg, obst, via, batch, b, irec, drec):
    """Evaluates the edge records with the REFERENCE edge classes; returns (err [E,3], jac [E,3,11], jac_valid [E])."""
    c = cfg.to_c()
    x, y, th, dt = batch.get_teb(b)
    n = len(x)
    dtp = np.zeros(n); dtp[:n - 1] = dt
    E = len(irec)
    err = np.zeros((E, 3)); jac = np.zeros((E, 33)); jv = np.zeros(E, np.int32)
    vx = _abi.f64([v[0] for v in via]) if via else _abi.f64([0.0])
    vy = _abi.f64([v[1] for v in via]) if via else _abi.f64([0.0])
    ir = np.ascontiguousarray(irec, np.int32); dr = np.ascontiguousarray(drec, np.float64)
    vs = _abi.f64(batch.vel_start[b]); vg = _abi.f64(batch.vel_goal[b])
    P = lambda a: _abi._ptr(a, C.c_double)
    rc = lib().ref_eval_edges(C.byref(c), C.byref(obst.freeze()), len(via), P(vx), P(vy), n, P(_abi.f64(x)), P(_abi.f64(y)),
                              P(_abi.f64(th)), P(dtp), P(vs), P(vg), E, _abi._ptr(ir, C.c_int32), P(dr), P(err), P(jac),
                              _abi._ptr(jv, C.c_int32))
    assert rc == 0
    return err, jac.reshape(E, 3, 11), jv


def distance(cfg, obst, oi, x, y, th, t=None):
    c = cfg.to_c()
    oi = _abi.i32(oi); x = _abi.f64(x); y = _abi.f64(y); th = _abi.f64(th)
    nq = len(oi)
    st = _abi.i32(np.zeros(nq) if t is None else np.ones(nq)); tt = _abi.f64(np.zeros(nq) if t is None else t)
    d = np.zeros(nq); cx = np.zeros(nq); cy = np.zeros(nq)
    P = lambda a: _abi._ptr(a, C.c_double)
    rc = lib().ref_distance(C.byref(c), C.byref(obst.freeze()), nq, _abi._ptr(oi, C.c_int32), P(x), P(y), P(th),
                            _abi._ptr(st, C.c_int32), P(tt), P(d), P(cx), P(cy))
    assert rc == 0
    return d, cx, cy


def autoresize(x, y, theta, dt, dt_ref, dt_hysteresis, min_samples, max_samples, fast_mode, cap=4096):
    n = len(x)
    X = np.zeros(cap); Y = np.zeros(cap); T = np.zeros(cap); D = np.zeros(cap)
    X[:n] = x; Y[:n] = y; T[:n] = theta; D[:n - 1] = dt
    nn = C.c_int32(n)
    P = lambda a: _abi._ptr(a, C.c_double)
    lib().ref_autoresize.argtypes = [_abi.p_f64] * 4 + [_abi.p_i32, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]
    rc = lib().ref_autoresize(P(X), P(Y), P(T), P(D), C.byref(nn), cap, dt_ref, dt_hysteresis, min_samples, max_samples,
                              int(fast_mode))
    assert rc == 0, rc
    n = nn.value
    return X[:n].copy(), Y[:n].copy(), T[:n].copy(), D[:n - 1].copy()


def _via_xy(via):
    vx = _abi.f64([v[0] for v in via]) if via else _abi.f64([0.0])
    vy = _abi.f64([v[1] for v in via]) if via else _abi.f64([0.0])
    return vx, vy


def _teb_flags(batch, b):
    hs = int(batch.has_vel_start[b]); hg = int(batch.has_vel_goal[b])
    rot = int(batch.prefer_rotdir[b]); ve = int(batch.via_points_enabled[b])
    return hs, _abi.f64(batch.vel_start[b]), hg, _abi.f64(batch.vel_goal[b]), rot, ve


def optimize_teb(cfg, obst, via, batch, b=0, inner=None, outer=None, compute_cost=True, obst_cost_scale=None,
                 viapoint_cost_scale=None, alternative_time_cost=None, cap=4096):
    """The reference's TebOptimalPlanner::optimizeTEB (src/optimal_planner.cpp, compiled in place) on TEB b.
    Returns dict(success, x, y, theta, dt, cost). The LM iteration inside is the stand-in of shim_g2o.h."""
    c = cfg.to_c()
    x, y, th, dt = batch.get_teb(b)
    n = len(x)
    X = np.zeros(cap); Y = np.zeros(cap); T = np.zeros(cap); D = np.zeros(cap)
    X[:n] = x; Y[:n] = y; T[:n] = th; D[:n - 1] = dt
    nn = C.c_int32(n)
    vx, vy = _via_xy(via)
    hs, vs, hg, vg, rot, ve = _teb_flags(batch, b)
    inner = cfg.optim.no_inner_iterations if inner is None else inner
    outer = cfg.optim.no_outer_iterations if outer is None else outer
    osc = cfg.hcp.selection_obst_cost_scale if obst_cost_scale is None else obst_cost_scale
    vsc = cfg.hcp.selection_viapoint_cost_scale if viapoint_cost_scale is None else viapoint_cost_scale
    atc = cfg.hcp.selection_alternative_time_cost if alternative_time_cost is None else alternative_time_cost
    ok = C.c_int32(0); cost = C.c_double(0)
    P = lambda a: _abi._ptr(a, C.c_double)
    rc = lib().ref_optimize_teb(C.byref(c), C.byref(obst.freeze()), len(via), P(vx), P(vy), C.byref(nn), cap, P(X), P(Y),
                                P(T), P(D), hs, P(vs), hg, P(vg), rot, ve, int(inner), int(outer), int(compute_cost),
                                C.c_double(osc), C.c_double(vsc), int(atc), C.byref(ok), C.byref(cost))
    assert rc == 0, rc
    n = nn.value
    return dict(success=bool(ok.value), x=X[:n].copy(), y=Y[:n].copy(), theta=T[:n].copy(), dt=D[:n - 1].copy(),
                cost=cost.value)


def build_graph(cfg, obst, via, batch, b=0, weight_multiplier=1.0, cap=1 << 16):
    """The hyper-graph the reference's buildGraph creates for TEB b: (irec [E,8], drec [E,48]), see ref_driver.cpp."""
    c = cfg.to_c()
    x, y, th, dt = batch.get_teb(b)
    n = len(x)
    dtp = np.zeros(n); dtp[:n - 1] = dt
    vx, vy = _via_xy(via)
    hs, vs, hg, vg, rot, ve = _teb_flags(batch, b)
    ir = np.zeros((cap, 8), np.int32); dr = np.zeros((cap, 48))
    cnt = C.c_int32(0)
    P = lambda a: _abi._ptr(a, C.c_double)
    rc = lib().ref_build_graph(C.byref(c), C.byref(obst.freeze()), len(via), P(vx), P(vy), n, P(_abi.f64(x)),
                               P(_abi.f64(y)), P(_abi.f64(th)), P(dtp), hs, P(vs), hg, P(vg), rot, ve,
                               C.c_double(weight_multiplier), _abi._ptr(ir, C.c_int32), P(dr), cap, C.byref(cnt))
    assert rc == 0, rc
    k = min(cnt.value, cap)
    return ir[:k].copy(), dr[:k].copy()


# ---- SURVEY section 8(f) rows f1 / f2 on the reference's own TimedElasticBand / TebOptimalPlanner -------------------------
_P = lambda a: _abi._ptr(a, C.c_double)


def _bufs(cap):
    return np.zeros(cap), np.zeros(cap), np.zeros(cap), np.zeros(cap), C.c_int32(0)


def _res(X, Y, T, D, nn):
    n = nn.value
    return X[:n].copy(), Y[:n].copy(), T[:n].copy(), D[:max(n - 1, 0)].copy()


def init_trajectory_line(start, goal, diststep, max_vel_x, min_samples, guess_backwards_motion, cap=4096):
    X, Y, T, D, nn = _bufs(cap)
    s = _abi.f64(start); g = _abi.f64(goal)
    assert lib().ref_init_trajectory_line(_P(s), _P(g), C.c_double(diststep), C.c_double(max_vel_x), int(min_samples),
                                          int(guess_backwards_motion), _P(X), _P(Y), _P(T), _P(D), C.byref(nn), cap) == 0
    return _res(X, Y, T, D, nn)


def init_trajectory_plan(px, py, pyaw, max_vel_x, max_vel_theta, estimate_orient, min_samples, guess_backwards_motion, cap=4096):
    """Returns (band, yaw_seen): the plan travels as quaternions (geometry_msgs::PoseStamped); yaw_seen = tf::getYaw of them."""
    X, Y, T, D, nn = _bufs(cap)
    px = _abi.f64(px); py = _abi.f64(py); pyaw = _abi.f64(pyaw)
    seen = np.zeros(len(px))
    assert lib().ref_init_trajectory_plan(len(px), _P(px), _P(py), _P(pyaw), C.c_double(max_vel_x), C.c_double(max_vel_theta),
                                          int(estimate_orient), int(min_samples), int(guess_backwards_motion), _P(seen), _P(X), _P(Y),
                                          _P(T), _P(D), C.byref(nn), cap) == 0
    return _res(X, Y, T, D, nn), seen


def init_trajectory_path(px, py, max_vel_x, max_vel_theta, max_acc_x, start_orient, goal_orient, min_samples,
                         guess_backwards_motion, cap=4096):
    X, Y, T, D, nn = _bufs(cap)
    px = _abi.f64(px); py = _abi.f64(py)
    opt = lambda v: (int(v is not None), C.c_double(0.0 if v is None else v))
    a, so, go = opt(max_acc_x), opt(start_orient), opt(goal_orient)
    assert lib().ref_init_trajectory_path(len(px), _P(px), _P(py), C.c_double(max_vel_x), C.c_double(max_vel_theta), a[0], a[1], so[0],
                                          so[1], go[0], go[1], int(min_samples), int(guess_backwards_motion), _P(X), _P(Y), _P(T),
                                          _P(D), C.byref(nn), cap) == 0
    return _res(X, Y, T, D, nn)


def update_and_prune(x, y, theta, dt, new_start, new_goal, min_samples):
    n = len(x)
    X = _abi.f64(x).copy(); Y = _abi.f64(y).copy(); T = _abi.f64(theta).copy(); D = np.zeros(n); D[:n - 1] = dt
    nn = C.c_int32(n)
    s = None if new_start is None else _abi.f64(new_start); g = None if new_goal is None else _abi.f64(new_goal)
    assert lib().ref_update_and_prune(_P(X), _P(Y), _P(T), _P(D), C.byref(nn), None if s is None else _P(s),
                                      None if g is None else _P(g), int(min_samples)) == 0
    return _res(X, Y, T, D, nn)


def consumers(cfg, batch, b, look_ahead_poses=1, prevent_look_ahead_poses_near_goal=0):
    c = cfg.to_c()
    x, y, th, dt = batch.get_teb(b)
    n = len(x)
    dtp = np.zeros(n); dtp[:n - 1] = dt
    cmd = np.zeros(3); ok = C.c_int32(0); prof = np.zeros((n + 1, 3)); traj = np.zeros((n, 7))
    vs = _abi.f64(batch.vel_start[b]); vg = _abi.f64(batch.vel_goal[b])
    assert lib().ref_consumers(C.byref(c), n, _P(_abi.f64(x)), _P(_abi.f64(y)), _P(_abi.f64(th)), _P(dtp), int(batch.has_vel_start[b]),
                               _P(vs), int(batch.has_vel_goal[b]), _P(vg), int(look_ahead_poses), int(prevent_look_ahead_poses_near_goal),
                               _P(cmd), C.byref(ok), _P(prof), _P(traj)) == 0
    return dict(cmd=cmd, ok=bool(ok.value), profile=prof, trajectory=traj)


def is_trajectory_feasible(batch, b, costmap, footprint, inscribed_radius, min_resolution_collision_check_angular=3.141592653589793,
                           look_ahead_idx=-1, feasibility_check_lookahead_distance=-1.0):
    """The reference's own TebOptimalPlanner::isTrajectoryFeasible (src/optimal_planner.cpp:1250-1308) over the grid CostmapModel of
    oracle/grid_costmap.h: (feasible, number of footprint tests passed before the failing one or -1)."""
    x, y, th, dt = batch.get_teb(b)
    n = len(x)
    dtp = np.zeros(n); dtp[:n - 1] = dt
    fx = _abi.f64([p[0] for p in footprint]); fy = _abi.f64([p[1] for p in footprint])
    ok = C.c_int32(0); first = C.c_int32(-1)
    assert lib().ref_is_trajectory_feasible(n, _P(_abi.f64(x)), _P(_abi.f64(y)), _P(_abi.f64(th)), _P(dtp),
                                            costmap.cells.ctypes.data_as(C.c_void_p), costmap.size_x, costmap.size_y,
                                            C.c_double(costmap.resolution), C.c_double(costmap.origin_x), C.c_double(costmap.origin_y),
                                            len(footprint), _P(fx), _P(fy), C.c_double(inscribed_radius),
                                            C.c_double(min_resolution_collision_check_angular), int(look_ahead_idx),
                                            C.c_double(feasibility_check_lookahead_distance), C.byref(ok), C.byref(first)) == 0
    return bool(ok.value), first.value


def h_signatures(cfg, obst, batch, mode, prescaler=1.0, threshold=0.1):
    """The reference's HSignature (mode 2) / HSignature3d (mode 3) on every band: dict(sig, equal [B,B], valid, reasonable)."""
    c = cfg.to_c()
    bs = batch.c_struct()
    B, M = batch.count, len(obst)
    sig = np.zeros((B, 2 if mode == 2 else M)); eq = np.zeros((B, B), np.int32); valid = np.zeros(B, np.int32); reas = np.zeros(B, np.int32)
    I = lambda a: _abi._ptr(a, C.c_int32)
    assert lib().ref_h_signatures(C.byref(c), C.byref(obst.freeze()), C.byref(bs), int(mode), C.c_double(prescaler), C.c_double(threshold),
                                  _P(sig), I(eq), I(valid), I(reas)) == 0
    return dict(sig=sig, equal=eq, valid=valid, reasonable=reas)


def optimize_batch(cfg, obst, via, batch, inner=None, outer=None, compute_cost=True, threads=1, trace=False):
    """B x the reference's TebOptimalPlanner::optimizeTEB (thread per band, capped): (new_batch, ok [B], cost [B], lm_iterations [B]);
    with trace=True also the per-band LM traces of the stand-in optimiser (list of [iterations, 4]: chi2, lambda, trials, pose count)."""
    c = cfg.to_c()
    out = batch.copy()
    bs = out.c_struct()
    vx, vy = _via_xy(via)
    inner = cfg.optim.no_inner_iterations if inner is None else inner
    outer = cfg.optim.no_outer_iterations if outer is None else outer
    B = batch.count
    ok = np.zeros(B, np.int32); cost = np.zeros(B); it = np.zeros(B, np.int32)
    I = lambda a: _abi._ptr(a, C.c_int32)
    cap = max(1, int(inner) * int(outer))
    if trace:
        tbuf = np.zeros((B, cap, 4)); trows = np.zeros(B, np.int32)
        lib().ref_set_trace(_P(tbuf), cap, I(trows))
    try:
        rc = lib().ref_optimize_batch(C.byref(c), C.byref(obst.freeze()), len(via), _P(vx), _P(vy), C.byref(bs), int(inner), int(outer),
                                      int(compute_cost), C.c_double(cfg.hcp.selection_obst_cost_scale),
                                      C.c_double(cfg.hcp.selection_viapoint_cost_scale), int(cfg.hcp.selection_alternative_time_cost),
                                      int(threads), I(ok), _P(cost), I(it))
    finally:
        if trace:
            lib().ref_set_trace(None, 0, None)
    assert rc == 0, rc
    if trace:
        return out, ok, cost, it, [tbuf[b, :trows[b]].copy() for b in range(B)]
    return out, ok, cost, it


def explore_candidates(cfg, obst, batch, best, start, goal, dist_to_obst=None, start_vel=None, free_goal_vel=False, skip_draws=0,
                       slots=None, vcap=4096, acap=1 << 20, optimized=None, stale_band=None, initial_plan=None, stale_initial_band=None, via=None,
                       via_enabled=None):
    """The reference's HomotopyClassPlanner::exploreEquivalenceClassesAndInitTebs (renewAndAnalyzeOldTebs without detour deletion,
    then createGraph / DepthFirst / addAndInitNewTeb) with tebs_ = the bands of `batch` (may be None) and best_teb_ = band `best`.
    dict(batch, n_total, vertices, adjacency, has_vel_start, vel_start, has_vel_goal)."""
    from teb_local_planner_amd import _abi as A
    c = cfg.to_c()
    p = cfg.hcp_params()
    stride = batch.stride if batch is not None else 2048
    slots = slots or max(cfg.hcp.max_number_classes + (batch.count if batch is not None else 0), 1)
    out = A.TebBatchHost(slots, stride)
    obs = out.c_struct()
    dist_to_obst = cfg.obstacles.min_obstacle_dist if dist_to_obst is None else dist_to_obst
    st = np.ascontiguousarray(start, np.float64); gl = np.ascontiguousarray(goal, np.float64)
    sv = None if start_vel is None else np.ascontiguousarray(start_vel, np.float64)
    vx = np.zeros(vcap); vy = np.zeros(vcap); off = np.zeros(vcap + 1, np.int32); adj = np.zeros(acap, np.int32)
    nt = C.c_int32(0); nv = C.c_int32(0)
    hvs = np.zeros(slots, np.int32); vs = np.zeros((slots, 3)); hvg = np.zeros(slots, np.int32)
    I = lambda a: _abi._ptr(a, C.c_int32)
    f = lib().ref_explore_candidates
    f.restype = C.c_int
    f.argtypes = [C.POINTER(A.Config), C.POINTER(A.HcpParams), C.POINTER(A.Obstacles), C.POINTER(A.TebBatch), C.c_int, A.p_f64, A.p_f64,
                  C.c_double, A.p_f64, C.c_int, C.c_long, A.p_i32, C.c_int, A.p_f64, A.p_f64, A.p_f64, A.p_f64, C.POINTER(A.TebBatch), A.p_i32, A.p_i32,
                  A.p_f64, A.p_i32, C.c_int, A.p_f64,
                  A.p_f64, A.p_i32, C.c_int, A.p_i32, A.p_i32, C.c_int, A.p_f64, A.p_f64, A.p_f64, C.c_int, A.p_f64, A.p_f64, A.p_f64, A.p_f64,
                  C.c_int, A.p_f64, A.p_f64, A.p_i32, A.p_i32, A.p_i32, A.p_f64]
    ins = batch.c_struct() if batch is not None else None
    sb = [np.ascontiguousarray(a, np.float64) for a in stale_band] if stale_band is not None else [np.zeros(0)] * 4
    sb[3] = np.append(sb[3], 0.0)
    plan = [np.ascontiguousarray(a, np.float64) for a in initial_plan] if initial_plan is not None else [np.zeros(0)] * 3
    si = [np.ascontiguousarray(a, np.float64) for a in stale_initial_band] if stale_initial_band is not None else [np.zeros(0)] * 4
    si[3] = np.append(si[3], 0.0)
    vxs = np.ascontiguousarray([v[0] for v in (via or [])], np.float64); vys = np.ascontiguousarray([v[1] for v in (via or [])], np.float64)
    ipt = C.c_int32(-1); vout = np.zeros(slots, np.int32); seen = np.zeros(max(len(plan[0]), 1))
    rc = f(C.byref(c), C.byref(p), C.byref(obst.freeze()), C.byref(ins) if ins is not None else None, int(best), _P(st), _P(gl),
           float(dist_to_obst), _abi._ptr(sv, C.c_double), int(bool(free_goal_vel)), int(skip_draws),
           _abi._ptr(None if optimized is None else np.ascontiguousarray(optimized, np.int32), C.c_int32), len(sb[0]), _P(sb[0]), _P(sb[1]),
           _P(sb[2]), _P(sb[3]), C.byref(obs), C.byref(nt), I(hvs),
           _P(vs), I(hvg), vcap, _P(vx), _P(vy), C.byref(nv), acap, I(off), I(adj), len(plan[0]), _P(plan[0]), _P(plan[1]), _P(plan[2]),
           len(si[0]), _P(si[0]), _P(si[1]), _P(si[2]), _P(si[3]), len(vxs), _P(vxs), _P(vys),
           _abi._ptr(None if via_enabled is None else np.ascontiguousarray(via_enabled, np.int32), C.c_int32), C.byref(ipt), I(vout), _P(seen))
    assert rc == 0, rc
    N = nv.value
    return dict(batch=out, n_total=nt.value, vertices=np.stack([vx[:N], vy[:N]], 1),
                adjacency=[adj[off[v]:off[v + 1]].tolist() for v in range(N)], has_vel_start=hvs, vel_start=vs, has_vel_goal=hvg,
                initial_plan_teb=ipt.value, via_enabled=vout[:nt.value].copy(), plan_yaw_seen=seen[:len(plan[0])].copy())


def hcp_plan_ticks(cfg, obst, starts, goals, start_vels=None, free_goal_vel=False, slots=8, stride=512, plans=None, via=None):
    """n_ticks x the reference's HomotopyClassPlanner::plan() on one planner object: list of dict(bands, best, costs) per tick."""
    from teb_local_planner_amd import _abi as A
    c = cfg.to_c()
    p = cfg.hcp_params()
    st = np.ascontiguousarray(starts, np.float64).reshape(-1, 3); gl = np.ascontiguousarray(goals, np.float64).reshape(-1, 3)
    T = len(st)
    sv = None if start_vels is None else np.ascontiguousarray(start_vels, np.float64).reshape(T, 3)
    out = A.TebBatchHost(T * slots, stride)
    obs = out.c_struct()
    counts = np.zeros(T, np.int32); best = np.zeros(T, np.int32); costs = np.zeros(T * slots)
    f = lib().ref_hcp_plan_ticks
    f.restype = C.c_int
    f.argtypes = [C.POINTER(A.Config), C.POINTER(A.HcpParams), C.POINTER(A.Obstacles), C.c_int, A.p_f64, A.p_f64, A.p_f64, C.c_int, C.c_int,
                  C.POINTER(A.TebBatch), A.p_i32, A.p_i32, A.p_f64, A.p_i32, A.p_f64, A.p_f64, A.p_f64, A.p_f64, C.c_int, A.p_f64, A.p_f64, A.p_i32]
    plans = plans or [None] * T
    off = np.zeros(T + 1, np.int32)
    for t in range(T):
        off[t + 1] = off[t] + (0 if plans[t] is None else len(plans[t][0]))
    cat = lambda k: np.ascontiguousarray(np.concatenate([np.asarray(pl[k], np.float64) for pl in plans if pl is not None] or [np.zeros(1)]))
    px, py, pyaw = cat(0), cat(1), cat(2)
    seen = np.zeros(len(pyaw)); ipt = np.zeros(T, np.int32)
    vxs = np.ascontiguousarray([v[0] for v in (via or [])] or [0.0], np.float64); vys = np.ascontiguousarray([v[1] for v in (via or [])] or [0.0], np.float64)
    rc = f(C.byref(c), C.byref(p), C.byref(obst.freeze()), T, _P(st), _P(gl), _abi._ptr(sv, C.c_double), int(bool(free_goal_vel)), slots,
           C.byref(obs), _abi._ptr(counts, C.c_int32), _abi._ptr(best, C.c_int32), _P(costs), _abi._ptr(off, C.c_int32), _P(px), _P(py), _P(pyaw),
           _P(seen), len(via or []), _P(vxs), _P(vys), _abi._ptr(ipt, C.c_int32))
    assert rc == 0, rc
    res = []
    for t in range(T):
        assert counts[t] <= slots
        res.append(dict(bands=[out.get_teb(t * slots + k) for k in range(counts[t])], best=int(best[t]),
                        costs=costs[t * slots:t * slots + counts[t]].copy(), initial_plan_teb=int(ipt[t]),
                        plan_yaw_seen=seen[off[t]:off[t + 1]].copy()))
    return res
