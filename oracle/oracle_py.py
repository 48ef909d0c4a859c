"""ctypes wrapper of libteb_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from teb_local_planner_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

COST_REFERENCE, COST_FRESH = 0, 1


def build(force=False):
    so = os.path.join(_HERE, "libteb_oracle.so")
    # every file the Makefile's rule depends on: a header that changes the layout of teb_amd_config_t (include/teb_amd.h) must rebuild the
    # oracle too, or it reads the caller's structs at the old offsets (found in round 4: TEB_AMD_MAX_FOOTPRINT_VERTICES 16 -> 64)
    deps = [os.path.join(_HERE, f) for f in ("teb_oracle.cpp", "teb_oracle.h", "grid_costmap.h")] + [os.path.join(_HERE, "..", "include", "teb_amd.h")]
    stale = not os.path.exists(so) or any(os.path.exists(d) and os.path.getmtime(d) > os.path.getmtime(so) for d in deps)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "libteb_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libteb_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.teb_oracle_optimize_batch.restype = C.c_int
        _LIB.teb_oracle_optimize_batch.argtypes = [
            C.POINTER(_abi.Config), C.POINTER(_abi.Obstacles), C.c_int32, _abi.p_f64, _abi.p_f64,
            C.POINTER(_abi.TebBatch), C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int32,
            C.c_int32, C.c_int32, C.POINTER(_abi.Results)]
        _LIB.teb_oracle_select_best.restype = C.c_int
        _LIB.teb_oracle_select_best.argtypes = [
            C.POINTER(_abi.Config), C.c_int32, _abi.p_f64, C.c_int32, C.c_int32, _abi.p_i32, _abi.p_f64]
        _LIB.teb_oracle_autoresize.restype = C.c_int
        _LIB.teb_oracle_autoresize.argtypes = [
            _abi.p_f64, _abi.p_f64, _abi.p_f64, _abi.p_f64, _abi.p_i32, C.c_int32, C.c_double, C.c_double,
            C.c_int32, C.c_int32, C.c_int32]
        _LIB.teb_oracle_linearize.restype = C.c_int
        _LIB.teb_oracle_linearize.argtypes = [
            C.POINTER(_abi.Config), C.POINTER(_abi.Obstacles), C.c_int32, _abi.p_f64, _abi.p_f64,
            C.POINTER(_abi.TebBatch), C.c_int32, C.c_double, _abi.p_f64, _abi.p_f64, _abi.p_f64,
            _abi.p_i32, _abi.p_i32]
        _LIB.teb_oracle_edges.restype = C.c_int
        _LIB.teb_oracle_edges.argtypes = [
            C.POINTER(_abi.Config), C.POINTER(_abi.Obstacles), C.c_int32, _abi.p_f64, _abi.p_f64,
            C.POINTER(_abi.TebBatch), C.c_int32, C.c_double, _abi.p_i32, _abi.p_f64, C.c_int32, _abi.p_i32]
        _LIB.teb_oracle_associate.restype = C.c_int
        _LIB.teb_oracle_associate.argtypes = [
            C.POINTER(_abi.Config), C.POINTER(_abi.Obstacles), C.POINTER(_abi.TebBatch), C.c_int32,
            _abi.p_i32, _abi.p_i32, C.c_int32, _abi.p_i32]
        _LIB.teb_oracle_distance.restype = C.c_int
        _LIB.teb_oracle_distance.argtypes = [
            C.POINTER(_abi.Config), C.POINTER(_abi.Obstacles), C.c_int32, C.c_double, C.c_double, C.c_double,
            C.c_int32, C.c_double, _abi.p_f64, _abi.p_f64]
        _LIB.teb_oracle_centroid.restype = C.c_int
        _LIB.teb_oracle_centroid.argtypes = [C.POINTER(_abi.Obstacles), C.c_int32, _abi.p_f64, _abi.p_f64]
    return _LIB


def _via_arrays(via):
    vx = _abi.f64([v[0] for v in via]) if via else _abi.f64([0.0])
    vy = _abi.f64([v[1] for v in via]) if via else _abi.f64([0.0])
    return vx, vy


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with status %d" % (what, rc))


def optimize_batch(cfg, obst, via, batch, inner=None, outer=None, compute_cost=True, obst_cost_scale=None,
                   viapoint_cost_scale=None, alternative_time_cost=None, cost_mode=COST_REFERENCE, threads=1, trace=False):
    """Runs B x optimizeTEB on a COPY of batch; returns (new_batch, ResultsHost), with trace=True also the per-band LM traces
    (list of [iterations, 4] arrays: chi2, lambda, damping trials, pose count per LM iteration)."""
    c = cfg.to_c()
    out = batch.copy()
    res = _abi.ResultsHost(batch.count)
    vx, vy = _via_arrays(via)
    inner = cfg.optim.no_inner_iterations if inner is None else inner
    outer = cfg.optim.no_outer_iterations if outer is None else outer
    osc = cfg.hcp.selection_obst_cost_scale if obst_cost_scale is None else obst_cost_scale
    vsc = cfg.hcp.selection_viapoint_cost_scale if viapoint_cost_scale is None else viapoint_cost_scale
    atc = cfg.hcp.selection_alternative_time_cost if alternative_time_cost is None else alternative_time_cost
    bs = out.c_struct()
    rs = res.c_struct()
    cap = max(1, int(inner) * int(outer))
    if trace:
        tbuf = np.zeros((batch.count, cap, 4)); trows = np.zeros(batch.count, np.int32)
        lib().teb_oracle_set_trace(_abi._ptr(tbuf, C.c_double), cap, _abi._ptr(trows, C.c_int32))
    try:
        rc = lib().teb_oracle_optimize_batch(
            C.byref(c), C.byref(obst.freeze()), len(via), _abi._ptr(vx, C.c_double), _abi._ptr(vy, C.c_double),
            C.byref(bs), inner, outer, int(compute_cost), float(osc), float(vsc), int(atc), cost_mode, threads,
            C.byref(rs))
    finally:
        if trace:
            lib().teb_oracle_set_trace(None, 0, None)
    _check(rc, "teb_oracle_optimize_batch")
    if trace:
        return out, res, [tbuf[b, :trows[b]].copy() for b in range(batch.count)]
    return out, res


def select_best(cfg, cost, last_best=-1, initial_plan=-1):
    c = cfg.to_c()
    cost = _abi.f64(cost)
    best = C.c_int32(-1)
    bc = C.c_double(0)
    _check(lib().teb_oracle_select_best(C.byref(c), len(cost), _abi._ptr(cost, C.c_double), last_best,
                                        initial_plan, C.byref(best), C.byref(bc)), "select_best")
    return best.value, bc.value


def autoresize(x, y, theta, dt, dt_ref, dt_hysteresis, min_samples, max_samples, fast_mode, cap=2048):
    n = len(x)
    X = np.zeros(cap); Y = np.zeros(cap); T = np.zeros(cap); D = np.zeros(cap)
    X[:n] = x; Y[:n] = y; T[:n] = theta; D[:n - 1] = dt
    nn = C.c_int32(n)
    _check(lib().teb_oracle_autoresize(_abi._ptr(X, C.c_double), _abi._ptr(Y, C.c_double), _abi._ptr(T, C.c_double),
                                       _abi._ptr(D, C.c_double), C.byref(nn), cap, dt_ref, dt_hysteresis,
                                       min_samples, max_samples, int(fast_mode)), "autoresize")
    n = nn.value
    return X[:n].copy(), Y[:n].copy(), T[:n].copy(), D[:n - 1].copy()


def linearize(cfg, obst, via, batch, b=0, weight_multiplier=1.0):
    """Returns dict(H [4n,4n], b [4n], chi2 [4], n_edges, n_rows) in the canonical index space."""
    c = cfg.to_c()
    n = int(batch.n[b])
    D = 4 * n
    H = np.zeros((D, D)); bv = np.zeros(D); chi2 = np.zeros(4)
    ne = C.c_int32(0); nr = C.c_int32(0)
    vx, vy = _via_arrays(via)
    bs = batch.c_struct()
    _check(lib().teb_oracle_linearize(C.byref(c), C.byref(obst.freeze()), len(via), _abi._ptr(vx, C.c_double),
                                      _abi._ptr(vy, C.c_double), C.byref(bs), b, weight_multiplier,
                                      _abi._ptr(H, C.c_double), _abi._ptr(bv, C.c_double),
                                      _abi._ptr(chi2, C.c_double), C.byref(ne), C.byref(nr)), "linearize")
    return dict(H=H, b=bv, chi2=chi2, n_edges=ne.value, n_rows=nr.value)


def edges(cfg, obst, via, batch, b=0, weight_multiplier=1.0, cap=1 << 16):
    """The oracle's hyper-graph of TEB b: (irec [E,16] int32, drec [E,56] float64), see teb_oracle.h."""
    c = cfg.to_c()
    ir = np.zeros((cap, 16), np.int32); dr = np.zeros((cap, 56))
    cnt = C.c_int32(0)
    vx, vy = _via_arrays(via)
    bs = batch.c_struct()
    _check(lib().teb_oracle_edges(C.byref(c), C.byref(obst.freeze()), len(via), _abi._ptr(vx, C.c_double),
                                  _abi._ptr(vy, C.c_double), C.byref(bs), b, weight_multiplier,
                                  _abi._ptr(ir, C.c_int32), _abi._ptr(dr, C.c_double), cap, C.byref(cnt)), "edges")
    k = min(cnt.value, cap)
    return ir[:k].copy(), dr[:k].copy()


def associate(cfg, obst, batch, b=0, cap=1 << 16):
    c = cfg.to_c()
    ap = np.zeros(cap, np.int32); ao = np.zeros(cap, np.int32)
    cnt = C.c_int32(0)
    bs = batch.c_struct()
    _check(lib().teb_oracle_associate(C.byref(c), C.byref(obst.freeze()), C.byref(bs), b,
                                      _abi._ptr(ap, C.c_int32), _abi._ptr(ao, C.c_int32), cap, C.byref(cnt)),
           "associate")
    k = min(cnt.value, cap)
    return ap[:k].copy(), ao[:k].copy()


def distance(cfg, obst, index, x, y, theta, t=None):
    c = cfg.to_c()
    d = C.c_double(0)
    g = np.zeros(3)
    _check(lib().teb_oracle_distance(C.byref(c), C.byref(obst.freeze()), index, x, y, theta,
                                     0 if t is None else 1, 0.0 if t is None else float(t), C.byref(d),
                                     _abi._ptr(g, C.c_double)), "distance")
    return d.value, g


def centroid(obst, index):
    cx = C.c_double(0); cy = C.c_double(0)
    _check(lib().teb_oracle_centroid(C.byref(obst.freeze()), index, C.byref(cx), C.byref(cy)), "centroid")
    return cx.value, cy.value


# ---- SURVEY section 8(f) rows f1 / f2 ------------------------------------------------------------------------------------
def _band_buffers(cap):
    return np.zeros(cap), np.zeros(cap), np.zeros(cap), np.zeros(cap), C.c_int32(0)


def _band_result(X, Y, T, D, nn):
    n = nn.value
    return X[:n].copy(), Y[:n].copy(), T[:n].copy(), D[:max(n - 1, 0)].copy()


_P = lambda a: _abi._ptr(a, C.c_double)


def init_trajectory_line(start, goal, diststep, max_vel_x, min_samples, guess_backwards_motion, cap=4096):
    X, Y, T, D, nn = _band_buffers(cap)
    s = _abi.f64(start); g = _abi.f64(goal)
    _check(lib().teb_oracle_init_trajectory_line(_P(s), _P(g), C.c_double(diststep), C.c_double(max_vel_x), int(min_samples),
                                                 int(guess_backwards_motion), _P(X), _P(Y), _P(T), _P(D), C.byref(nn), cap),
           "init_trajectory_line")
    return _band_result(X, Y, T, D, nn)


def init_trajectory_plan(px, py, pyaw, max_vel_x, max_vel_theta, estimate_orient, min_samples, guess_backwards_motion, cap=4096):
    X, Y, T, D, nn = _band_buffers(cap)
    px = _abi.f64(px); py = _abi.f64(py); pyaw = _abi.f64(pyaw)
    _check(lib().teb_oracle_init_trajectory_plan(len(px), _P(px), _P(py), _P(pyaw), C.c_double(max_vel_x), C.c_double(max_vel_theta),
                                                 int(estimate_orient), int(min_samples), int(guess_backwards_motion), _P(X), _P(Y),
                                                 _P(T), _P(D), C.byref(nn), cap), "init_trajectory_plan")
    return _band_result(X, Y, T, D, nn)


def init_trajectory_path(px, py, max_vel_x, max_vel_theta, max_acc_x, start_orient, goal_orient, min_samples,
                         guess_backwards_motion, cap=4096):
    """max_acc_x / start_orient / goal_orient: None = boost::none."""
    X, Y, T, D, nn = _band_buffers(cap)
    px = _abi.f64(px); py = _abi.f64(py)
    opt = lambda v: (int(v is not None), C.c_double(0.0 if v is None else v))
    a, so, go = opt(max_acc_x), opt(start_orient), opt(goal_orient)
    _check(lib().teb_oracle_init_trajectory_path(len(px), _P(px), _P(py), C.c_double(max_vel_x), C.c_double(max_vel_theta), a[0], a[1],
                                                 so[0], so[1], go[0], go[1], int(min_samples), int(guess_backwards_motion), _P(X), _P(Y),
                                                 _P(T), _P(D), C.byref(nn), cap), "init_trajectory_path")
    return _band_result(X, Y, T, D, nn)


def update_and_prune(x, y, theta, dt, new_start, new_goal, min_samples):
    n = len(x)
    X = _abi.f64(x).copy(); Y = _abi.f64(y).copy(); T = _abi.f64(theta).copy(); D = np.zeros(n); D[:n - 1] = dt
    nn = C.c_int32(n)
    s = None if new_start is None else _abi.f64(new_start); g = None if new_goal is None else _abi.f64(new_goal)
    _check(lib().teb_oracle_update_and_prune(_P(X), _P(Y), _P(T), _P(D), C.byref(nn), None if s is None else _P(s),
                                             None if g is None else _P(g), int(min_samples)), "update_and_prune")
    return _band_result(X, Y, T, D, nn)


def consumers(cfg, batch, b, look_ahead_poses=1, prevent_look_ahead_poses_near_goal=0):
    """getVelocityCommand / getVelocityProfile / getFullTrajectory on TEB b: dict(cmd [3], ok, profile [n+1,3], trajectory [n,7])."""
    c = cfg.to_c()
    bs = batch.c_struct()
    n = int(batch.n[b])
    cmd = np.zeros(3); ok = C.c_int32(0); prof = np.zeros((n + 1, 3)); traj = np.zeros((n, 7))
    _check(lib().teb_oracle_velocity_command(C.byref(c), C.byref(bs), b, int(look_ahead_poses), int(prevent_look_ahead_poses_near_goal),
                                             _P(cmd), C.byref(ok)), "velocity_command")
    _check(lib().teb_oracle_velocity_profile(C.byref(c), C.byref(bs), b, _P(prof)), "velocity_profile")
    _check(lib().teb_oracle_full_trajectory(C.byref(c), C.byref(bs), b, _P(traj)), "full_trajectory")
    return dict(cmd=cmd, ok=bool(ok.value), profile=prof, trajectory=traj)


# ---- row f3 (arithmetic core): H-signatures and equivalence classes ------------------------------------------------------------
class Costmap:
    """uint8 grid of a costmap_2d::Costmap2D: cells[my, mx]; 254 lethal, 253 inscribed, 255 no information."""

    def __init__(self, cells, resolution, origin_x, origin_y):
        self.cells = np.ascontiguousarray(cells, dtype=np.uint8)
        self.size_y, self.size_x = self.cells.shape
        self.resolution, self.origin_x, self.origin_y = float(resolution), float(origin_x), float(origin_y)


def footprint_cost(costmap, x, y, theta, footprint):
    """base_local_planner::CostmapModel::footprintCost restated on the grid (oracle/grid_costmap.h)."""
    L = lib()
    L.teb_oracle_footprint_cost.restype = C.c_double
    fx = _abi.f64([p[0] for p in footprint]); fy = _abi.f64([p[1] for p in footprint])
    return L.teb_oracle_footprint_cost(costmap.cells.ctypes.data_as(C.c_void_p), costmap.size_x, costmap.size_y, C.c_double(costmap.resolution),
                                       C.c_double(costmap.origin_x), C.c_double(costmap.origin_y), C.c_double(x), C.c_double(y), C.c_double(theta),
                                       len(footprint), _abi._ptr(fx, C.c_double), _abi._ptr(fy, C.c_double))


def is_trajectory_feasible(batch, b, costmap, footprint, inscribed_radius, min_resolution_collision_check_angular=3.141592653589793,
                           look_ahead_idx=-1, feasibility_check_lookahead_distance=-1.0):
    """TebOptimalPlanner::isTrajectoryFeasible (src/optimal_planner.cpp:1250-1308): (feasible, number of footprint tests passed before the
    failing one or -1)."""
    bs = batch.c_struct()
    fx = _abi.f64([p[0] for p in footprint]); fy = _abi.f64([p[1] for p in footprint])
    ok = C.c_int32(0); first = C.c_int32(-1)
    _check(lib().teb_oracle_is_trajectory_feasible(C.byref(bs), int(b), costmap.cells.ctypes.data_as(C.c_void_p), costmap.size_x, costmap.size_y,
                                                   C.c_double(costmap.resolution), C.c_double(costmap.origin_x), C.c_double(costmap.origin_y),
                                                   len(footprint), _abi._ptr(fx, C.c_double), _abi._ptr(fy, C.c_double),
                                                   C.c_double(inscribed_radius), C.c_double(min_resolution_collision_check_angular),
                                                   int(look_ahead_idx), C.c_double(feasibility_check_lookahead_distance), C.byref(ok),
                                                   C.byref(first)), "teb_oracle_is_trajectory_feasible")
    return bool(ok.value), first.value


def h_signatures(cfg, obst, batch, mode, prescaler=1.0):
    """mode 2: HSignature -> [B, 2]; mode 3: HSignature3d -> [B, M]."""
    c = cfg.to_c()
    bs = batch.c_struct()
    M = len(obst)
    out = np.zeros((batch.count, 2 if mode == 2 else M))
    for b in range(batch.count):
        row = np.zeros(out.shape[1])
        if mode == 2:
            _check(lib().teb_oracle_h_signature_2d(C.byref(c), C.byref(obst.freeze()), C.byref(bs), b, C.c_double(prescaler), _P(row)),
                   "h_signature_2d")
        else:
            _check(lib().teb_oracle_h_signature_3d(C.byref(c), C.byref(obst.freeze()), C.byref(bs), b, _P(row)), "h_signature_3d")
        out[b] = row
    return out


def filter_equivalence_classes(mode, sig, threshold=0.1, best=-1, max_number_plans_in_current_class=1, stale_best_sig=None):
    sig = np.ascontiguousarray(sig, np.float64)
    B = sig.shape[0]
    M = sig.shape[1]
    keep = np.zeros(B, np.int32); valid = np.zeros(B, np.int32); reas = np.zeros(B, np.int32)
    I = lambda a: _abi._ptr(a, C.c_int32)
    f = lib().teb_oracle_filter_equivalence_classes_stale
    f.restype = C.c_int
    f.argtypes = [C.c_int32, C.c_int32, C.c_int32, _abi.p_f64, C.c_double, C.c_int32, C.c_int32, _abi.p_f64, _abi.p_i32, _abi.p_i32, _abi.p_i32]
    st = None if stale_best_sig is None else np.ascontiguousarray(stale_best_sig, np.float64)
    _check(f(int(mode), B, M, _P(sig), C.c_double(threshold), int(best), int(max_number_plans_in_current_class),
             _abi._ptr(st, C.c_double), I(keep), I(valid), I(reas)), "filter_equivalence_classes")
    return keep, valid, reas


# ---- row f3 (candidate generation): createGraph + DepthFirst + addAndInitNewTeb --------------------------------------------------
def explore_candidates(cfg, obst, batch, n_tebs, best, start, goal, dist_to_obst=None, unit_samples=None, skip_draws=0,
                       vcap=4096, acap=1 << 20, max_paths=0, stale_best_sig=None, initial_plan=None, stale_initial_sig=None, via_enabled=None):
    """Bands 0..n_tebs-1 of `batch` are tebs_ after renewAndAnalyzeOldTebs; returns dict(batch (copy, candidates appended), n_total,
    vertices [nv, 2], adjacency (list of lists, insertion order), n_paths)."""
    c = cfg.to_c()
    p = cfg.hcp_params()
    out = batch.copy()
    bs = out.c_struct()
    dist_to_obst = cfg.obstacles.min_obstacle_dist if dist_to_obst is None else dist_to_obst
    st = np.ascontiguousarray(start, np.float64); gl = np.ascontiguousarray(goal, np.float64)
    us = None if unit_samples is None else np.ascontiguousarray(unit_samples, np.float64).ravel()
    vx = np.zeros(vcap); vy = np.zeros(vcap); off = np.zeros(vcap + 1, np.int32); adj = np.zeros(acap, np.int32)
    nt = C.c_int32(0); nv = C.c_int32(0); npth = C.c_int32(0)
    I = lambda a: _abi._ptr(a, C.c_int32)
    f = lib().teb_oracle_explore_candidates
    f.restype = C.c_int
    f.argtypes = [C.POINTER(_abi.Config), C.POINTER(_abi.HcpParams), C.POINTER(_abi.Obstacles), C.POINTER(_abi.TebBatch), C.c_int32,
                  C.c_int32, _abi.p_f64, _abi.p_f64, C.c_double, _abi.p_f64, C.c_int64, C.c_int64, _abi.p_f64, _abi.p_i32, C.c_int32, _abi.p_f64, _abi.p_f64,
                  _abi.p_i32, C.c_int32, _abi.p_i32, _abi.p_i32, _abi.p_i32, C.c_int32, _abi.p_f64, _abi.p_f64, _abi.p_f64, _abi.p_f64,
                  _abi.p_i32, _abi.p_i32]
    plan = [None, None, None] if initial_plan is None else [np.ascontiguousarray(a, np.float64) for a in initial_plan]
    sis = None if stale_initial_sig is None else np.ascontiguousarray(stale_initial_sig, np.float64)
    ve = None if via_enabled is None else np.ascontiguousarray(via_enabled, np.int32).copy()
    ipt = C.c_int32(-1)
    _check(f(C.byref(c), C.byref(p), C.byref(obst.freeze()), C.byref(bs), int(n_tebs), int(best), _P(st), _P(gl),
             float(dist_to_obst), _abi._ptr(us, C.c_double), int(skip_draws), int(max_paths),
             _abi._ptr(None if stale_best_sig is None else np.ascontiguousarray(stale_best_sig, np.float64), C.c_double), C.byref(nt), vcap, _P(vx), _P(vy), C.byref(nv), acap,
             I(off), I(adj), C.byref(npth), 0 if initial_plan is None else len(plan[0]), _abi._ptr(plan[0], C.c_double),
             _abi._ptr(plan[1], C.c_double), _abi._ptr(plan[2], C.c_double), _abi._ptr(sis, C.c_double), C.byref(ipt), _abi._ptr(ve, C.c_int32)),
           "explore_candidates")
    N = nv.value
    assert N <= vcap and off[N] <= acap
    return dict(batch=out, n_total=nt.value, vertices=np.stack([vx[:N], vy[:N]], 1),
                adjacency=[adj[off[v]:off[v + 1]].tolist() for v in range(N)], n_paths=npth.value, initial_plan_teb=ipt.value,
                via_enabled=ve)


def filter_detours(cfg, batch, keep, best, optimized):
    """deletePlansDetouringBackwards on the bands with keep != 0: returns the new keep array."""
    p = cfg.hcp_params()
    bs = batch.c_struct()
    keep = np.ascontiguousarray(keep, np.int32).copy()
    opt = np.ascontiguousarray(optimized, np.int32)
    f = lib().teb_oracle_filter_detours
    f.restype = C.c_int
    f.argtypes = [C.POINTER(_abi.TebBatch), C.POINTER(_abi.HcpParams), C.c_int32, _abi.p_i32, _abi.p_i32]
    _check(f(C.byref(bs), C.byref(p), int(best), _abi._ptr(opt, C.c_int32), _abi._ptr(keep, C.c_int32)), "filter_detours")
    return keep
