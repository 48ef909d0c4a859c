"""Host-side mirror of the reference's optimiser interface on top of the C-ABI (libteb_amd.so).

  TebOptimalPlanner.optimizeTEB(...)         <-> reference optimal_planner.h:231-232
  HomotopyClassPlanner.optimizeAllTEBs(...)  <-> reference src/homotopy_class_planner.cpp:466-493
  HomotopyClassPlanner.selectBestTeb()       <-> reference src/homotopy_class_planner.cpp:564-667

Same names, argument meaning and error behaviour (bool returns, no exceptions for optimiser failures).
The library is loaded with ctypes; there is NO CPU fallback: constructing a solver without a gfx950 GPU
raises TebAmdError.
"""
import ctypes as C
import os

import numpy as np

from . import _abi
from .config import TebConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class TebAmdError(RuntimeError):
    def __init__(self, code, what, msg):
        super().__init__("%s failed: status %d (%s)" % (what, code, msg))
        self.code = code


def lib():
    """Loads libteb_amd.so (built in-tree by teb_local_planner_amd.build)."""
    global _LIB
    if _LIB is None:
        so = os.environ.get("TEB_AMD_LIB") or os.path.join(_HERE, "libteb_amd.so")   # TEB_AMD_LIB: kernel experiments (tools/)
        if not os.path.exists(so):
            raise TebAmdError(-1, "load", "libteb_amd.so missing: run `python -m teb_local_planner_amd.build`")
        L = C.CDLL(so)
        vp = C.c_void_p
        L.teb_amd_last_error.restype = C.c_char_p
        L.teb_amd_config_default.argtypes = [C.POINTER(_abi.Config)]
        L.teb_amd_create.argtypes = [C.POINTER(_abi.Config), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, vp, C.POINTER(vp)]
        L.teb_amd_create_ex.argtypes = [C.POINTER(_abi.Config), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, vp, C.POINTER(_abi.Options), C.POINTER(vp)]
        L.teb_amd_options_default.argtypes = [C.POINTER(_abi.Options)]
        L.teb_amd_options_default.restype = None
        L.teb_amd_destroy.argtypes = [vp]
        L.teb_amd_destroy.restype = None
        L.teb_amd_set_config.argtypes = [vp, C.POINTER(_abi.Config)]
        L.teb_amd_set_obstacles.argtypes = [vp, C.POINTER(_abi.Obstacles)]
        L.teb_amd_set_via_points.argtypes = [vp, C.c_int32, _abi.p_f64, _abi.p_f64]
        L.teb_amd_upload_tebs.argtypes = [vp, C.POINTER(_abi.TebBatch)]
        L.teb_amd_download_tebs.argtypes = [vp, C.POINTER(_abi.TebBatch)]
        L.teb_amd_optimize_batch.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int32]
        L.teb_amd_synchronize.argtypes = [vp]
        L.teb_amd_get_results.argtypes = [vp, C.POINTER(_abi.Results)]
        L.teb_amd_select_best.argtypes = [vp, C.c_int32, C.c_int32, _abi.p_i32, _abi.p_f64]
        if hasattr(L, "teb_amd_last_launch_info"):
            L.teb_amd_last_launch_info.argtypes = [vp, _abi.p_i32, _abi.p_i32, _abi.p_i32]
        if hasattr(L, "teb_amd_multi_cu_backoff"):
            L.teb_amd_multi_cu_backoff.argtypes = [vp, _abi.p_i32, _abi.p_i32]
        if hasattr(L, "teb_amd_set_iteration_log"):   # (absent from the older builds tools/ compares against through TEB_AMD_LIB)
            L.teb_amd_set_iteration_log.argtypes = [vp, C.c_int32]
            L.teb_amd_get_iteration_log.argtypes = [vp, C.c_int32, _abi.p_f64, C.c_int32, _abi.p_i32]
            L.teb_amd_debug_world_argmin.argtypes = [_abi.p_f64, C.c_int32, _abi.p_i32, _abi.p_f64, _abi.p_i32]
        L.teb_amd_device_state.argtypes = [vp] + [C.POINTER(vp)] * 5 + [_abi.p_i32]
        L.teb_amd_snapshot_state.argtypes = [vp]
        L.teb_amd_restore_state.argtypes = [vp]
        L.teb_amd_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.teb_amd_capacity.argtypes = [vp, _abi.p_i32, _abi.p_i32]
        L.teb_amd_debug_linearize.argtypes = [vp, C.c_int32, C.c_double, _abi.p_f64, _abi.p_f64, _abi.p_f64,
                                              _abi.p_i32, _abi.p_i32, C.c_int32, _abi.p_i32]
        L.teb_amd_debug_distance.argtypes = [vp, C.c_int32, _abi.p_i32, _abi.p_f64, _abi.p_f64, _abi.p_f64,
                                             _abi.p_i32, _abi.p_f64, _abi.p_f64, _abi.p_f64]
        L.teb_amd_debug_assoc_overflow.argtypes = [vp, _abi.p_i32]
        L.teb_amd_debug_stream.argtypes = [vp, C.c_int64, C.c_int32]
        L.teb_amd_debug_profile.argtypes = [vp, _abi.p_f64]
        # producers / consumers of the device-resident strips
        d, i32 = C.c_double, C.c_int32
        L.teb_amd_init_trajectory_line.argtypes = [vp, i32, _abi.p_f64, _abi.p_f64, d, d, i32, i32]
        L.teb_amd_init_trajectory_plan.argtypes = [vp, i32, i32, _abi.p_f64, _abi.p_f64, _abi.p_f64, d, d, i32, i32, i32]
        L.teb_amd_init_trajectory_path.argtypes = [vp, i32, i32, _abi.p_f64, _abi.p_f64, d, d, _abi.p_f64, _abi.p_f64, _abi.p_f64,
                                                   i32, i32]
        L.teb_amd_update_and_prune.argtypes = [vp, i32, _abi.p_f64, _abi.p_f64, i32]
        L.teb_amd_set_velocity_start.argtypes = [vp, i32, i32, _abi.p_f64]
        L.teb_amd_set_velocity_goal.argtypes = [vp, i32, i32, _abi.p_f64]
        L.teb_amd_get_pose_counts.argtypes = [vp, _abi.p_i32, _abi.p_i32]
        L.teb_amd_get_velocity_command.argtypes = [vp, i32, i32, i32, _abi.p_f64, _abi.p_f64, _abi.p_f64, _abi.p_i32]
        L.teb_amd_get_velocity_profile.argtypes = [vp, i32, _abi.p_f64, i32, _abi.p_i32]
        L.teb_amd_get_full_trajectory.argtypes = [vp, i32, _abi.p_f64, i32, _abi.p_i32]
        L.teb_amd_has_diverged.argtypes = [vp, i32, _abi.p_i32]
        L.teb_amd_get_batch_statistics.argtypes = [vp, _abi.p_i32, _abi.p_f64]
        L.teb_amd_compute_h_signatures.argtypes = [vp, d, _abi.p_f64, _abi.p_i32]
        L.teb_amd_filter_equivalence_classes.argtypes = [vp, d, i32, i32, _abi.p_i32, _abi.p_i32, _abi.p_i32]
        L.teb_amd_explore_candidates.argtypes = [vp, C.POINTER(_abi.HcpParams), _abi.p_f64, _abi.p_f64, d, _abi.p_f64, i32, i32,
                                                 _abi.p_f64, C.c_int64, _abi.p_i32, _abi.p_i32, _abi.p_i32, i32, _abi.p_f64, _abi.p_f64,
                                                 _abi.p_f64, _abi.p_i32]
        L.teb_amd_get_exploration_graph.argtypes = [vp, _abi.p_f64, _abi.p_f64, C.POINTER(C.c_ubyte), i32, _abi.p_i32]
        L.teb_amd_compact_bands.argtypes = [vp, _abi.p_i32, i32, _abi.p_i32, _abi.p_i32]
        L.teb_amd_filter_detours.argtypes = [vp, C.POINTER(_abi.HcpParams), i32, _abi.p_i32]
        L.teb_amd_set_optimized_flags.argtypes = [vp, _abi.p_i32]
        L.teb_amd_get_optimized_flags.argtypes = [vp, _abi.p_i32]
        L.teb_amd_get_band_flags.argtypes = [vp, _abi.p_i32, _abi.p_i32, _abi.p_i32]
        L.teb_amd_hcp_params_default.argtypes = [C.POINTER(_abi.HcpParams)]
        L.teb_amd_hcp_params_default.restype = None
        L.teb_amd_set_costmap.argtypes = [vp, C.c_void_p, i32, i32, d, d, d]
        L.teb_amd_is_trajectory_feasible.argtypes = [vp, i32, i32, _abi.p_f64, _abi.p_f64, d, d, i32, d, _abi.p_i32, _abi.p_i32]
        # multi-GPU exchange
        L.teb_amd_comm_unique_id.argtypes = [C.c_char_p]
        L.teb_amd_comm_create.argtypes = [C.c_char_p, i32, i32, i32, C.POINTER(vp)]
        L.teb_amd_comm_destroy.argtypes = [vp]
        L.teb_amd_comm_destroy.restype = None
        L.teb_amd_select_best_distributed.argtypes = [vp, vp, i32, i32, i32, _abi.p_i32, _abi.p_f64, _abi.p_i32]
        L.teb_amd_broadcast_band.argtypes = [vp, vp, i32, i32, i32, _abi.p_i32, _abi.p_f64, _abi.p_f64, _abi.p_f64, _abi.p_f64]
        _LIB = L
    return _LIB


def _chk(rc, what):
    if rc != 0:
        raise TebAmdError(rc, what, lib().teb_amd_last_error().decode())


class TebBatchSolver:
    """Thin RAII wrapper of a teb_amd_handle_t (one per GPU / host thread)."""

    def __init__(self, cfg, max_tebs, max_poses, max_obstacles=0, max_obstacle_vertices=0, max_via_points=0,
                 device=0, stream=None, options=None):
        self._h = C.c_void_p(None)
        self.cfg = cfg
        c = cfg.to_c()
        _chk(lib().teb_amd_create_ex(C.byref(c), max_tebs, max_poses, max_obstacles, max_obstacle_vertices,
                                     max_via_points, device, C.c_void_p(stream) if stream else None,
                                     C.byref(options) if options is not None else None, C.byref(self._h)), "teb_amd_create_ex")
        self.max_tebs, self.max_poses = max_tebs, max_poses
        self.count = 0

    def close(self):
        if self._h:
            lib().teb_amd_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- scene ---------------------------------------------------------------------------------------
    def set_config(self, cfg):
        self.cfg = cfg
        c = cfg.to_c()
        _chk(lib().teb_amd_set_config(self._h, C.byref(c)), "teb_amd_set_config")

    def set_obstacles(self, obst):
        _chk(lib().teb_amd_set_obstacles(self._h, C.byref(obst.freeze())), "teb_amd_set_obstacles")
        self._n_obst = len(obst)

    def set_via_points(self, via):
        vx = _abi.f64([v[0] for v in via]) if via else _abi.f64([0.0])
        vy = _abi.f64([v[1] for v in via]) if via else _abi.f64([0.0])
        _chk(lib().teb_amd_set_via_points(self._h, len(via), _abi._ptr(vx, C.c_double), _abi._ptr(vy, C.c_double)),
             "teb_amd_set_via_points")

    # -- state ---------------------------------------------------------------------------------------
    def upload(self, batch):
        bs = batch.c_struct()
        _chk(lib().teb_amd_upload_tebs(self._h, C.byref(bs)), "teb_amd_upload_tebs")
        self.count = batch.count

    def download(self, batch):
        bs = batch.c_struct()
        _chk(lib().teb_amd_download_tebs(self._h, C.byref(bs)), "teb_amd_download_tebs")
        return batch

    def snapshot(self):
        _chk(lib().teb_amd_snapshot_state(self._h), "teb_amd_snapshot_state")

    def restore(self):
        _chk(lib().teb_amd_restore_state(self._h), "teb_amd_restore_state")

    # -- hot path --------------------------------------------------------------------------------------
    def optimize(self, inner, outer, compute_cost=False, obst_cost_scale=1.0, viapoint_cost_scale=1.0,
                 alternative_time_cost=False):
        _chk(lib().teb_amd_optimize_batch(self._h, inner, outer, int(compute_cost), float(obst_cost_scale),
                                          float(viapoint_cost_scale), int(alternative_time_cost)),
             "teb_amd_optimize_batch")

    def synchronize(self):
        _chk(lib().teb_amd_synchronize(self._h), "teb_amd_synchronize")

    def results(self):
        res = _abi.ResultsHost(self.count)
        rs = res.c_struct()
        _chk(lib().teb_amd_get_results(self._h, C.byref(rs)), "teb_amd_get_results")
        return res

    def set_iteration_log(self, enable=True):
        """Opt-in per-iteration log of the LM loop (g2o's verbose line as data, src/optimal_planner.cpp:384)."""
        _chk(lib().teb_amd_set_iteration_log(self._h, int(bool(enable))), "teb_amd_set_iteration_log")

    PHASES = ("autoresize", "graph side data", "linearize", "H backup", "solve", "update + evaluate", "accept / restore")

    def set_phase_log(self, enable=True):
        """Phase split of the product kernel (include/teb_amd_debug.h): shader cycles per phase of every band's workgroup."""
        L = lib()
        L.teb_amd_set_phase_log.argtypes = [C.c_void_p, C.c_int32]
        _chk(L.teb_amd_set_phase_log(self._h, int(bool(enable))), "teb_amd_set_phase_log")

    def phase_log(self):
        """[B, 9] shader cycles of the last optimize(): the seven PHASES, a spare slot, the whole workgroup."""
        L = lib()
        L.teb_amd_get_phase_log.argtypes = [C.c_void_p, _abi.p_f64, C.c_int32, _abi.p_i32]
        self._sync_count()
        out = np.zeros((max(self.count, 1), 9)); nb = C.c_int32(0)
        _chk(L.teb_amd_get_phase_log(self._h, _abi._ptr(out, C.c_double), self.count, C.byref(nb)), "teb_amd_get_phase_log")
        return out[:nb.value]

    def iteration_log(self, b, capacity_rows=256):
        """[iterations, 4] of band b after the last optimize(): chi2 after the iteration, lambda after it, damping trials, pose count."""
        rows = np.zeros((int(capacity_rows), 4))
        n = C.c_int32(0)
        _chk(lib().teb_amd_get_iteration_log(self._h, int(b), _abi._ptr(rows, C.c_double), int(capacity_rows), C.byref(n)),
             "teb_amd_get_iteration_log")
        return rows[:n.value].copy()

    def select_best(self, last_best=-1, initial_plan=-1):
        best = C.c_int32(-1)
        cost = C.c_double(0)
        _chk(lib().teb_amd_select_best(self._h, last_best, initial_plan, C.byref(best), C.byref(cost)),
             "teb_amd_select_best")
        return best.value, cost.value

    def select_best_distributed(self, comm, global_offset, last_best_global=-1, initial_plan_global=-1):
        """selectBestTeb over the candidates of all ranks of `comm` (parallel.RcclComm): (best_global, scaled cost, owner_rank), the
        same on every rank. One 16-byte-per-rank ncclAllGather inside libteb_amd.so."""
        best = C.c_int32(-1); cost = C.c_double(0); owner = C.c_int32(-1)
        _chk(lib().teb_amd_select_best_distributed(self._h, comm._c, int(global_offset), int(last_best_global), int(initial_plan_global),
                                                   C.byref(best), C.byref(cost), C.byref(owner)), "teb_amd_select_best_distributed")
        return best.value, cost.value, owner.value

    def broadcast_band(self, comm, owner_rank, local_index, capacity=None):
        """The winner's strip from its owner to every rank: (x, y, theta, dt) as host arrays."""
        cap = int(capacity or self.max_poses)
        n = C.c_int32(0)
        x = np.zeros(cap); y = np.zeros(cap); th = np.zeros(cap); dt = np.zeros(cap)
        P = lambda a: _abi._ptr(a, C.c_double)
        _chk(lib().teb_amd_broadcast_band(self._h, comm._c, int(owner_rank), int(local_index), cap, C.byref(n), P(x), P(y), P(th), P(dt)),
             "teb_amd_broadcast_band")
        return x[:n.value].copy(), y[:n.value].copy(), th[:n.value].copy(), dt[:max(n.value - 1, 0)].copy()

    def last_kernel_ms(self):
        ms = C.c_float(0)
        _chk(lib().teb_amd_last_kernel_ms(self._h, C.byref(ms)), "teb_amd_last_kernel_ms")
        return ms.value

    def last_shader_clock_mhz(self):
        """shader clock of the last optimise kernel (cycle counter / real-time counter of its first workgroup)"""
        L = lib()
        L.teb_amd_last_shader_clock_mhz.argtypes = [C.c_void_p, _abi.p_f64]
        v = C.c_double(0)
        _chk(L.teb_amd_last_shader_clock_mhz(self._h, C.byref(v)), "teb_amd_last_shader_clock_mhz")
        return v.value

    def last_launch_info(self):
        """(distance helpers per band of the last optimize(), solver helpers per band, repeated on one CU per band after a timeout)"""
        a = C.c_int32(0); k = C.c_int32(0); b = C.c_int32(0)
        _chk(lib().teb_amd_last_launch_info(self._h, C.byref(a), C.byref(k), C.byref(b)), "teb_amd_last_launch_info")
        return a.value, k.value, bool(b.value)

    def multi_cu_backoff(self):
        """(launches still paused, current pause length) of the distance helpers' back-off (teb_amd_multi_cu_backoff)"""
        a = C.c_int32(0); k = C.c_int32(0)
        _chk(lib().teb_amd_multi_cu_backoff(self._h, C.byref(a), C.byref(k)), "teb_amd_multi_cu_backoff")
        return a.value, k.value

    def last_instantiation(self):
        """(layout, Jacobian mode, scene kind) of the kernel instantiation the last optimize() launched (teb_amd_debug_last_instantiation)"""
        a = C.c_int32(-1); j = C.c_int32(-1); k = C.c_int32(-1)
        _chk(lib().teb_amd_debug_last_instantiation(self._h, C.byref(a), C.byref(j), C.byref(k)), "teb_amd_debug_last_instantiation")
        return a.value, j.value, k.value

    def last_config_profile(self):
        """Which kernel the last optimize() ran: 1 = specialised on the TebConfig defaults, 2 = the same folds except via-points and the
        holonomic choice (*_WIDE kinds), 3 = every cost-term flag at run time (*_LIGHT kinds), 0 = the generic instantiation (teb_amd_options_t::generic_config_path forces it)"""
        L = lib()
        L.teb_amd_debug_last_config_profile.argtypes = [C.c_void_p, _abi.p_i32]
        v = C.c_int32(0)
        _chk(L.teb_amd_debug_last_config_profile(self._h, C.byref(v)), "teb_amd_debug_last_config_profile")
        return int(v.value)

    @staticmethod
    def build_info():
        """What the loaded binary was built from: (kernel hash, source hash, variant defines, threads per workgroup of the optimise
        kernel) - teb_amd_debug_build_info, include/teb_amd_debug.h"""
        L = lib()
        L.teb_amd_debug_build_info.argtypes = [C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, _abi.p_i32]
        kb = C.create_string_buffer(64); hb = C.create_string_buffer(64); db = C.create_string_buffer(1024); t = C.c_int32(0)
        _chk(L.teb_amd_debug_build_info(kb, 64, hb, 64, db, 1024, C.byref(t)), "teb_amd_debug_build_info")
        return kb.value.decode(), hb.value.decode(), db.value.decode(), t.value

    @staticmethod
    def rtc_stats():
        """run-time compiled instantiations of this process: (ready, compiling, failed, compile seconds of the last one, last error)"""
        L = lib()
        L.teb_amd_debug_rtc_stats.argtypes = [_abi.p_i32, _abi.p_i32, _abi.p_i32, _abi.p_f64, C.c_char_p, C.c_int32]
        r = C.c_int32(0); w = C.c_int32(0); f = C.c_int32(0); s = C.c_double(0); buf = C.create_string_buffer(4096)
        _chk(L.teb_amd_debug_rtc_stats(C.byref(r), C.byref(w), C.byref(f), C.byref(s), buf, 4096), "teb_amd_debug_rtc_stats")
        return r.value, w.value, f.value, s.value, buf.value.decode(errors="replace")

    def capacity(self):
        a = C.c_int32(0)
        b = C.c_int32(0)
        _chk(lib().teb_amd_capacity(self._h, C.byref(a), C.byref(b)), "teb_amd_capacity")
        return a.value, b.value

    # -- producers / consumers of the device-resident strips ----------------------
    @staticmethod
    def _opt3(v):
        return None if v is None else _abi._ptr(_abi.f64(v), C.c_double)

    @staticmethod
    def _opt1(v):
        return None if v is None else _abi._ptr(_abi.f64([v]), C.c_double)

    def _sync_count(self):
        c = C.c_int32(0)
        _chk(lib().teb_amd_get_pose_counts(self._h, None, C.byref(c)), "teb_amd_get_pose_counts")
        self.count = c.value

    def init_trajectory_line(self, b, start, goal, diststep, max_vel_x, min_samples, guess_backwards_motion=False):
        _chk(lib().teb_amd_init_trajectory_line(self._h, b, self._opt3(start), self._opt3(goal), diststep, max_vel_x, min_samples,
                                                int(guess_backwards_motion)), "teb_amd_init_trajectory_line")
        self._sync_count()

    def init_trajectory_plan(self, b, px, py, pyaw, max_vel_x, max_vel_theta, estimate_orient, min_samples,
                             guess_backwards_motion=False):
        px = _abi.f64(px); py = _abi.f64(py); pyaw = _abi.f64(pyaw)
        P = lambda a: _abi._ptr(a, C.c_double)
        _chk(lib().teb_amd_init_trajectory_plan(self._h, b, len(px), P(px), P(py), P(pyaw), max_vel_x, max_vel_theta,
                                                int(estimate_orient), min_samples, int(guess_backwards_motion)),
             "teb_amd_init_trajectory_plan")
        self._sync_count()

    def init_trajectory_path(self, b, px, py, max_vel_x, max_vel_theta, max_acc_x=None, start_orientation=None,
                             goal_orientation=None, min_samples=3, guess_backwards_motion=False):
        px = _abi.f64(px); py = _abi.f64(py)
        P = lambda a: _abi._ptr(a, C.c_double)
        _chk(lib().teb_amd_init_trajectory_path(self._h, b, len(px), P(px), P(py), max_vel_x, max_vel_theta, self._opt1(max_acc_x),
                                                self._opt1(start_orientation), self._opt1(goal_orientation), min_samples,
                                                int(guess_backwards_motion)), "teb_amd_init_trajectory_path")
        self._sync_count()

    def update_and_prune(self, new_start=None, new_goal=None, min_samples=3, b=-1):
        _chk(lib().teb_amd_update_and_prune(self._h, b, self._opt3(new_start), self._opt3(new_goal), min_samples),
             "teb_amd_update_and_prune")

    def set_velocity_start(self, v, fixed=True, b=-1):
        _chk(lib().teb_amd_set_velocity_start(self._h, b, int(fixed), self._opt3(v)), "teb_amd_set_velocity_start")

    def set_velocity_goal(self, v=None, fixed=True, b=-1):
        _chk(lib().teb_amd_set_velocity_goal(self._h, b, int(fixed), self._opt3(v)), "teb_amd_set_velocity_goal")

    def pose_counts(self):
        self._sync_count()
        n = np.zeros(max(self.count, 1), np.int32)
        c = C.c_int32(0)
        _chk(lib().teb_amd_get_pose_counts(self._h, _abi._ptr(n, C.c_int32), C.byref(c)), "teb_amd_get_pose_counts")
        return n[:c.value].copy()

    def velocity_command(self, b, look_ahead_poses=1, prevent_look_ahead_poses_near_goal=0):
        vx = C.c_double(0); vy = C.c_double(0); om = C.c_double(0); ok = C.c_int32(0)
        _chk(lib().teb_amd_get_velocity_command(self._h, b, look_ahead_poses, prevent_look_ahead_poses_near_goal, C.byref(vx),
                                                C.byref(vy), C.byref(om), C.byref(ok)), "teb_amd_get_velocity_command")
        return bool(ok.value), np.array([vx.value, vy.value, om.value])

    def velocity_profile(self, b):
        rows = C.c_int32(0)
        _chk(lib().teb_amd_get_velocity_profile(self._h, b, None, 0, C.byref(rows)), "teb_amd_get_velocity_profile")
        out = np.zeros((rows.value, 3))
        _chk(lib().teb_amd_get_velocity_profile(self._h, b, _abi._ptr(out, C.c_double), rows.value, C.byref(rows)),
             "teb_amd_get_velocity_profile")
        return out

    def full_trajectory(self, b):
        rows = C.c_int32(0)
        _chk(lib().teb_amd_get_full_trajectory(self._h, b, None, 0, C.byref(rows)), "teb_amd_get_full_trajectory")
        out = np.zeros((rows.value, 7))
        _chk(lib().teb_amd_get_full_trajectory(self._h, b, _abi._ptr(out, C.c_double), rows.value, C.byref(rows)),
             "teb_amd_get_full_trajectory")
        return out

    def has_diverged(self, b):
        d = C.c_int32(0)
        _chk(lib().teb_amd_has_diverged(self._h, b, C.byref(d)), "teb_amd_has_diverged")
        return bool(d.value)

    def batch_statistics(self):
        """(available [B] bool, back_chi2 [B]): what optimizer_->batchStatistics() of every band would hold after the last optimize() -
        not empty / .back().chi2 (zero when the band's last optimize() call stopped before its last requested iteration)."""
        self._sync_count()
        B = self.count
        av = np.zeros(max(B, 1), np.int32); ch = np.zeros(max(B, 1))
        _chk(lib().teb_amd_get_batch_statistics(self._h, _abi._ptr(av, C.c_int32), _abi._ptr(ch, C.c_double)), "teb_amd_get_batch_statistics")
        return av[:B].astype(bool), ch[:B]

    # -- feasibility of the resident bands against a costmap grid (arithmetic part) -----------
    def set_costmap(self, cells, resolution, origin_x, origin_y):
        cells = np.ascontiguousarray(cells, dtype=np.uint8)
        _chk(lib().teb_amd_set_costmap(self._h, cells.ctypes.data_as(C.c_void_p), cells.shape[1], cells.shape[0], float(resolution),
                                       float(origin_x), float(origin_y)), "teb_amd_set_costmap")

    def is_trajectory_feasible(self, b, footprint, inscribed_radius, min_resolution_collision_check_angular=3.141592653589793,
                               look_ahead_idx=-1, feasibility_check_lookahead_distance=-1.0):
        """isTrajectoryFeasible of band b (-1: every band): (feasible, first_infeasible) - bools / ints, arrays for b = -1."""
        cnt = self.count if b < 0 else 1
        fx = _abi.f64([p[0] for p in footprint]); fy = _abi.f64([p[1] for p in footprint])
        ok = np.zeros(cnt, np.int32); first = np.zeros(cnt, np.int32)
        _chk(lib().teb_amd_is_trajectory_feasible(self._h, int(b), len(footprint), _abi._ptr(fx, C.c_double), _abi._ptr(fy, C.c_double),
                                                  float(inscribed_radius), float(min_resolution_collision_check_angular), int(look_ahead_idx),
                                                  float(feasibility_check_lookahead_distance), _abi._ptr(ok, C.c_int32), _abi._ptr(first, C.c_int32)),
             "teb_amd_is_trajectory_feasible")
        if b < 0:
            return ok.astype(bool), first
        return bool(ok[0]), int(first[0])

    # -- equivalence classes of the resident bands (arithmetic core) ----------------------
    def h_signatures(self, prescaler=1.0, values=True):
        """[B, M] (HSignature3d, include_dynamic_obstacles) or [B, 2] (HSignature: re, im); values=False: compute only (the
        signatures stay in the handle for filter_equivalence_classes / explore_candidates)."""
        self._sync_count()
        w = C.c_int32(0)
        if not values:
            _chk(lib().teb_amd_compute_h_signatures(self._h, prescaler, None, C.byref(w)), "teb_amd_compute_h_signatures")
            return None
        width = max(getattr(self, "_n_obst", 0), 2)
        out = np.zeros((max(self.count, 1), width))
        _chk(lib().teb_amd_compute_h_signatures(self._h, prescaler, _abi._ptr(out, C.c_double), C.byref(w)),
             "teb_amd_compute_h_signatures")
        return out.ravel()[:self.count * w.value].reshape(self.count, w.value).copy()

    def filter_equivalence_classes(self, threshold=0.1, best=-1, max_number_plans_in_current_class=1):
        keep = np.zeros(self.count, np.int32); valid = np.zeros(self.count, np.int32); reas = np.zeros(self.count, np.int32)
        I = lambda a: _abi._ptr(a, C.c_int32)
        _chk(lib().teb_amd_filter_equivalence_classes(self._h, threshold, best, max_number_plans_in_current_class, I(keep), I(valid),
                                                      I(reas)), "teb_amd_filter_equivalence_classes")
        return keep, valid, reas

    # -- candidate generation: createGraph + DepthFirst + addAndInitNewTeb -----------------
    def set_optimized_flags(self, flags):
        f = _abi.i32(flags)
        _chk(lib().teb_amd_set_optimized_flags(self._h, _abi._ptr(f, C.c_int32)), "teb_amd_set_optimized_flags")

    def optimized_flags(self):
        self._sync_count()
        f = np.zeros(max(self.count, 1), np.int32)
        _chk(lib().teb_amd_get_optimized_flags(self._h, _abi._ptr(f, C.c_int32)), "teb_amd_get_optimized_flags")
        return f[:self.count]

    def band_flags(self):
        """(via_points_enabled, has_vel_start, has_vel_goal), each [B]."""
        self._sync_count()
        v = np.zeros(max(self.count, 1), np.int32); a = v.copy(); g = v.copy()
        I = lambda x: _abi._ptr(x, C.c_int32)
        _chk(lib().teb_amd_get_band_flags(self._h, I(v), I(a), I(g)), "teb_amd_get_band_flags")
        return v[:self.count], a[:self.count], g[:self.count]

    def filter_detours(self, keep, best, params=None):
        """deletePlansDetouringBackwards on the bands with keep != 0: returns the new keep array."""
        p = params if params is not None else self.cfg.hcp_params()
        keep = _abi.i32(keep).copy()
        _chk(lib().teb_amd_filter_detours(self._h, C.byref(p), int(best), _abi._ptr(keep, C.c_int32)), "teb_amd_filter_detours")
        return keep

    def compact_bands(self, keep, best=-1):
        """Keeps the bands with keep[b] != 0, last best band first (renewAndAnalyzeOldTebs): (n_kept, new_best)."""
        keep = _abi.i32(keep)
        nk = C.c_int32(0); nb = C.c_int32(-1)
        _chk(lib().teb_amd_compact_bands(self._h, _abi._ptr(keep, C.c_int32), int(best), C.byref(nk), C.byref(nb)),
             "teb_amd_compact_bands")
        self.count = nk.value
        return nk.value, nb.value

    def explore_candidates(self, start, goal, dist_to_obst=None, start_vel=None, free_goal_vel=False, best=-1, unit_samples=None,
                           max_paths=0, params=None, initial_plan=None):
        """exploreEquivalenceClassesAndInitTebs after renewAndAnalyzeOldTebs on the resident batch: dict(n_total, n_vertices, n_paths)."""
        p = params if params is not None else self.cfg.hcp_params()
        st = _abi.f64(start); gl = _abi.f64(goal)
        sv = None if start_vel is None else _abi.f64(start_vel)
        us = None if unit_samples is None else _abi.f64(np.asarray(unit_samples).ravel())
        dist_to_obst = self.cfg.obstacles.min_obstacle_dist if dist_to_obst is None else dist_to_obst
        nt = C.c_int32(0); nv = C.c_int32(0); npth = C.c_int32(0); ipt = C.c_int32(-1)
        plan = [None, None, None] if initial_plan is None else [_abi.f64(a) for a in initial_plan]      # (x, y, yaw) of the poses
        _chk(lib().teb_amd_explore_candidates(self._h, C.byref(p), _abi._ptr(st, C.c_double), _abi._ptr(gl, C.c_double),
                                              float(dist_to_obst), _abi._ptr(sv, C.c_double), int(bool(free_goal_vel)), int(best),
                                              _abi._ptr(us, C.c_double), int(max_paths), C.byref(nt), C.byref(nv), C.byref(npth),
                                              0 if initial_plan is None else len(plan[0]), _abi._ptr(plan[0], C.c_double),
                                              _abi._ptr(plan[1], C.c_double), _abi._ptr(plan[2], C.c_double), C.byref(ipt)),
             "teb_amd_explore_candidates")
        self.count = nt.value
        return dict(n_total=nt.value, n_vertices=nv.value, n_paths=npth.value, initial_plan_teb=ipt.value)

    def exploration_graph(self):
        """(vertices [N, 2], adjacency [N, N] uint8) of the last explore_candidates call."""
        nv = C.c_int32(0)
        _chk(lib().teb_amd_get_exploration_graph(self._h, None, None, None, 0, C.byref(nv)), "teb_amd_get_exploration_graph")
        N = nv.value
        vx = np.zeros(max(N, 1)); vy = np.zeros(max(N, 1)); adj = np.zeros((max(N, 1), max(N, 1)), np.uint8)
        if N:
            _chk(lib().teb_amd_get_exploration_graph(self._h, _abi._ptr(vx, C.c_double), _abi._ptr(vy, C.c_double),
                                                     adj.ctypes.data_as(C.POINTER(C.c_ubyte)), N, C.byref(nv)),
                 "teb_amd_get_exploration_graph")
        return np.stack([vx[:N], vy[:N]], 1), adj[:N, :N]

    # -- test hooks -----------------------------------------------------------------------------------
    def debug_linearize(self, b, n, weight_multiplier=1.0, assoc_cap=1 << 16):
        D = 4 * n
        H = np.zeros((D, D)); bv = np.zeros(D); chi2 = np.zeros(4)
        ap = np.zeros(assoc_cap, np.int32); ao = np.zeros(assoc_cap, np.int32)
        cnt = C.c_int32(0)
        _chk(lib().teb_amd_debug_linearize(self._h, b, weight_multiplier, _abi._ptr(H, C.c_double),
                                           _abi._ptr(bv, C.c_double), _abi._ptr(chi2, C.c_double),
                                           _abi._ptr(ap, C.c_int32), _abi._ptr(ao, C.c_int32), assoc_cap,
                                           C.byref(cnt)), "teb_amd_debug_linearize")
        k = min(cnt.value, assoc_cap)
        return dict(H=H, b=bv, chi2=chi2, assoc_pose=ap[:k].copy(), assoc_obst=ao[:k].copy())

    def debug_distance(self, obst_index, x, y, theta, t=None):
        oi = _abi.i32(obst_index); x = _abi.f64(x); y = _abi.f64(y); th = _abi.f64(theta)
        nq = len(oi)
        st = _abi.i32(np.zeros(nq) if t is None else np.ones(nq))
        tt = _abi.f64(np.zeros(nq) if t is None else t)
        d = np.zeros(nq); g = np.zeros((nq, 3))
        _chk(lib().teb_amd_debug_distance(self._h, nq, _abi._ptr(oi, C.c_int32), _abi._ptr(x, C.c_double),
                                          _abi._ptr(y, C.c_double), _abi._ptr(th, C.c_double),
                                          _abi._ptr(st, C.c_int32), _abi._ptr(tt, C.c_double),
                                          _abi._ptr(d, C.c_double), _abi._ptr(g, C.c_double)),
             "teb_amd_debug_distance")
        return d, g

    def debug_stream(self, n_doubles, repeats=1):
        _chk(lib().teb_amd_debug_stream(self._h, int(n_doubles), int(repeats)), "teb_amd_debug_stream")

    def debug_overflow_flags(self):
        f = np.zeros(self.count, np.int32)
        _chk(lib().teb_amd_debug_assoc_overflow(self._h, _abi._ptr(f, C.c_int32)), "debug_assoc_overflow")
        return f


def make_solver(cfg, obst, via, batch, device=0, stream=None, max_tebs=None, max_poses=None, options=None):
    """Creates a solver sized for the scene, uploads scene + batch. options: _abi.Options (layout pins etc.) or None."""
    nverts = len(obst.vert_x)
    s = TebBatchSolver(cfg, max_tebs or batch.count, max_poses or batch.stride, max(len(obst), 1), max(nverts, 1),
                       max(len(via), 1), device=device, stream=stream, options=options)
    s.set_obstacles(obst)
    s.set_via_points(via)
    s.upload(batch)
    return s


class TebOptimalPlanner:
    """Single-candidate view (reference include/teb_local_planner/optimal_planner.h:100-696)."""

    def __init__(self, cfg, obstacles=None, via_points=None, max_poses=None, device=0):
        self.cfg_ = cfg
        self.obstacles_ = obstacles if obstacles is not None else _abi.ObstacleTable()
        self.via_points_ = list(via_points or [])
        # capacity = whatever autoResize may produce (max_samples + 1, at most TEB_AMD_MAX_POSES); pass a smaller max_poses for the faster LDS layouts
        self.max_poses = max_poses or min(cfg.trajectory.max_samples + 1, _abi.MAX_POSES)
        self.teb_ = _abi.TebBatchHost(1, self.max_poses)
        self.cost_ = float("nan")
        self.optimized_ = False
        self.device = device
        self._solver = None
        self.last_results = None

    def teb(self):
        return self.teb_

    def setVelocityStart(self, vx, vy, omega):
        self.teb_.has_vel_start[0] = 1
        self.teb_.vel_start[0] = (vx, vy, omega)

    def setVelocityGoal(self, vx, vy, omega):
        self.teb_.has_vel_goal[0] = 1
        self.teb_.vel_goal[0] = (vx, vy, omega)

    def setVelocityGoalFree(self):
        self.teb_.has_vel_goal[0] = 0

    def getCurrentCost(self):
        return self.cost_

    def isOptimized(self):
        return self.optimized_

    def optimizeTEB(self, iterations_innerloop, iterations_outerloop, compute_cost_afterwards=False,
                    obst_cost_scale=1.0, viapoint_cost_scale=1.0, alternative_time_cost=False):
        if not self.cfg_.optim.optimization_activate:
            return False
        self.optimized_ = False
        if self._solver is None:
            self._solver = TebBatchSolver(self.cfg_, 1, self.max_poses, max(len(self.obstacles_), 1),
                                          max(len(self.obstacles_.vert_x), 1), max(len(self.via_points_), 1),
                                          device=self.device)
        s = self._solver
        s.set_config(self.cfg_)
        s.set_obstacles(self.obstacles_)
        s.set_via_points(self.via_points_)
        s.upload(self.teb_)
        s.optimize(iterations_innerloop, iterations_outerloop, compute_cost_afterwards, obst_cost_scale,
                   viapoint_cost_scale, alternative_time_cost)
        res = s.results()
        s.download(self.teb_)
        self.last_results = res
        if res.status[0] != _abi.TEB_OK:
            return False
        self.optimized_ = True
        if compute_cost_afterwards:
            self.cost_ = float(res.cost[0])
        return True


    # ---- plan(): warm start on the device-resident band, then optimizeTEB (src/optimal_planner.cpp:247-320) -------------------
    def _ensure_solver(self):
        if self._solver is None:
            self._solver = TebBatchSolver(self.cfg_, 1, self.max_poses, max(len(self.obstacles_), 1),
                                          max(len(self.obstacles_.vert_x), 1), max(len(self.via_points_), 1), device=self.device)
            self._resident = False
        return self._solver

    def clearPlanner(self):
        """clearPlanner(): the next plan() initialises a new band (optimal_planner.h:315-320)."""
        self.teb_.n[0] = 0
        self._resident = False
        self.optimized_ = False

    def _tick(self, init, start, goal, start_vel, free_goal_vel):
        import math
        s = self._ensure_solver()
        t = self.cfg_.trajectory
        n = int(self.teb_.n[0])
        warm = False
        if getattr(self, "_resident", False) and n > 0:
            gx, gy, gth = self.teb_.x[0, n - 1], self.teb_.y[0, n - 1], self.teb_.theta[0, n - 1]
            d = math.hypot(goal[0] - gx, goal[1] - gy)
            a = abs((goal[2] - gth + math.pi) % (2 * math.pi) - math.pi)      # fabs(g2o::normalize_theta(...))
            warm = d < t.force_reinit_new_goal_dist and a < t.force_reinit_new_goal_angular
        if warm:
            s.update_and_prune(start, goal, t.min_samples)                          # updateAndPruneTEB on the device
        else:
            init(s)                                                                 # initTrajectoryToGoal on the device
        self._resident = True
        if start_vel is not None:
            self.setVelocityStart(*start_vel)
        self.teb_.has_vel_goal[0] = 0 if free_goal_vel else 1                       # setVelocityGoalFree() / vel_goal_.first = true
        s.set_velocity_start(self.teb_.vel_start[0], bool(self.teb_.has_vel_start[0]))
        s.set_velocity_goal(self.teb_.vel_goal[0], bool(self.teb_.has_vel_goal[0]))
        s.set_config(self.cfg_)
        s.set_obstacles(self.obstacles_)
        s.set_via_points(self.via_points_)
        o = self.cfg_.optim
        if not o.optimization_activate:
            return False
        self.optimized_ = False
        s.optimize(o.no_inner_iterations, o.no_outer_iterations)
        res = s.results()
        s.download(self.teb_)               # host copy for teb() / the next warm-start test; the band itself stays on the device
        self.last_results = res
        self.optimized_ = res.status[0] == _abi.TEB_OK
        return bool(self.optimized_)

    def plan(self, start, goal, start_vel=None, free_goal_vel=False):
        """plan(const PoseSE2& start, const PoseSE2& goal, start_vel, free_goal_vel), src/optimal_planner.cpp:292-320.
        start / goal = (x, y, theta), start_vel = (vx, vy, omega) or None."""
        t, r = self.cfg_.trajectory, self.cfg_.robot
        return self._tick(lambda s: s.init_trajectory_line(0, start, goal, 0, r.max_vel_x, t.min_samples,
                                                           t.allow_init_with_backwards_motion), start, goal, start_vel, free_goal_vel)

    def planFromPath(self, plan_x, plan_y, plan_yaw, start_vel=None, free_goal_vel=False):
        """plan(const std::vector<geometry_msgs::PoseStamped>& initial_plan, ...), src/optimal_planner.cpp:247-283."""
        t, r = self.cfg_.trajectory, self.cfg_.robot
        start = (plan_x[0], plan_y[0], plan_yaw[0]); goal = (plan_x[-1], plan_y[-1], plan_yaw[-1])
        return self._tick(lambda s: s.init_trajectory_plan(0, plan_x, plan_y, plan_yaw, r.max_vel_x, r.max_vel_theta,
                                                           t.global_plan_overwrite_orientation, t.min_samples,
                                                           t.allow_init_with_backwards_motion), start, goal, start_vel, free_goal_vel)

    # ---- consumers (src/optimal_planner.cpp:1023-1247) on the device-resident band ---------------------------------------------
    def getVelocityCommand(self, look_ahead_poses=None):
        t = self.cfg_.trajectory
        ok, v = self._ensure_solver().velocity_command(0, t.control_look_ahead_poses if look_ahead_poses is None else look_ahead_poses,
                                                       t.prevent_look_ahead_poses_near_goal)
        return ok, float(v[0]), float(v[1]), float(v[2])

    def getVelocityProfile(self):
        return self._ensure_solver().velocity_profile(0)

    def getFullTrajectory(self):
        return self._ensure_solver().full_trajectory(0)

    def hasDiverged(self):
        return self._ensure_solver().has_diverged(0)

    def isTrajectoryFeasible(self, costmap, footprint_spec, inscribed_radius, circumscribed_radius=0.0, look_ahead_idx=-1,
                             feasibility_check_lookahead_distance=-1.0):
        """isTrajectoryFeasible (optimal_planner.h:500, src/optimal_planner.cpp:1250-1308) of the resident band. costmap: an object with
        cells [size_y, size_x] uint8, resolution, origin_x, origin_y (the grid of costmap_2d::Costmap2D) standing in for the
        base_local_planner::CostmapModel* of the reference; footprint_spec: [(x, y), ...]. circumscribed_radius is accepted and unused,
        as in CostmapModel::footprintCost."""
        s = self._ensure_solver()
        s.set_costmap(costmap.cells, costmap.resolution, costmap.origin_x, costmap.origin_y)
        return s.is_trajectory_feasible(0, footprint_spec, inscribed_radius, self.cfg_.trajectory.min_resolution_collision_check_angular,
                                        look_ahead_idx, feasibility_check_lookahead_distance)[0]


class HomotopyClassPlanner:
    """Batch view: owns the candidates resident on one GPU (reference homotopy_class_planner.h). Either constructed around a host
    batch (optimizeAllTEBs / selectBestTeb on given bands) or empty (batch=None): plan() then runs the reference's whole tick on the
    device-resident bands - updateAllTEBs, exploreEquivalenceClassesAndInitTebs (incl. the initial-plan candidate), optimizeAllTEBs,
    selectBestTeb (src/homotopy_class_planner.cpp:84-125), incl. randomlyDropTebs (off by default; own random stream - the reference
    seeds from std::random_device) and switching_blocking_period."""

    def __init__(self, cfg, obstacles, via_points, batch=None, device=0, stream=None, max_tebs=None, max_poses=None):
        self.cfg_ = cfg
        self.tebs_ = batch
        self.obstacles_ = obstacles
        self.via_points_ = list(via_points or [])
        if batch is not None:
            self.solver = make_solver(cfg, obstacles, via_points, batch, device=device, stream=stream, max_tebs=max_tebs,
                                      max_poses=max_poses)
        else:
            # pose capacity: 224 keeps the normal matrix as 8x8 blocks in LDS (the fastest layout, <= 238 poses); pass up to _abi.MAX_POSES for longer bands
            self.solver = TebBatchSolver(cfg, max_tebs or max(cfg.hcp.max_number_classes, 1), max_poses or 224, max(len(obstacles), 1),
                                         max(len(obstacles.vert_x), 1), max(len(self.via_points_), 1), device=device, stream=stream)
            self.solver.set_obstacles(obstacles)
            self.solver.set_via_points(self.via_points_)
        self.best_teb_ = -1
        self.initial_plan_teb_ = -1
        self.last_results = None
        self.last_exploration = None
        self._goal = None

    # ---- src/homotopy_class_planner.cpp:539-562 ---------------------------------------------------------------------------------
    def updateAllTEBs(self, start, goal, start_velocity=None):
        import math
        s, t = self.solver, self.cfg_.trajectory
        if s.count > 0 and self._goal is not None:
            d = math.hypot(goal[0] - self._goal[0], goal[1] - self._goal[1])
            a = abs((goal[2] - self._goal[2] + math.pi) % (2 * math.pi) - math.pi)
            if d >= t.force_reinit_new_goal_dist or a >= t.force_reinit_new_goal_angular:
                s.compact_bands(np.zeros(s.count, np.int32))      # tebs_.clear(); equivalence_classes_.clear()
                self.best_teb_ = -1
        if s.count > 0:
            s.update_and_prune(start, goal, t.min_samples)        # every band, one launch
            if start_velocity is not None:
                s.set_velocity_start(start_velocity, True)
        self._goal = tuple(goal)

    # ---- :318-340 (renewAndAnalyzeOldTebs :214-254, deletePlansDetouringBackwards :766-817, createGraph) -------------------------
    def exploreEquivalenceClassesAndInitTebs(self, start, goal, dist_to_obst, start_vel=None, free_goal_vel=False, initial_plan=None):
        s, h = self.solver, self.cfg_.hcp
        if s.count > 0:
            s.h_signatures(h.h_signature_prescaler, values=False)
            keep, _, _ = s.filter_equivalence_classes(h.h_signature_threshold, self.best_teb_, h.max_number_plans_in_current_class)
            if h.delete_detours_backwards:
                keep = s.filter_detours(keep, self.best_teb_)
            if h.selection_dropping_probability > 0:       # randomlyDropTebs (:539-560): every band but the best one, with that probability
                if not hasattr(self, "_rng"):
                    self._rng = np.random.default_rng()
                drop = self._rng.random(len(keep)) <= h.selection_dropping_probability
                drop[self.best_teb_] = False if self.best_teb_ >= 0 else drop[self.best_teb_]
                keep = np.where(drop, 0, keep).astype(np.int32)
            _, self.best_teb_ = s.compact_bands(keep, self.best_teb_)
        self.last_exploration = s.explore_candidates(start, goal, dist_to_obst, start_vel, free_goal_vel, self.best_teb_,
                                                     initial_plan=initial_plan)
        self.initial_plan_teb_ = self.last_exploration["initial_plan_teb"]      # getInitialPlanTEB() of selectBestTeb
        return self.last_exploration["n_total"]

    def plan(self, start, goal, start_vel=None, free_goal_vel=False, initial_plan=None):
        """plan(const PoseSE2& start, const PoseSE2& goal, const geometry_msgs::Twist* start_vel, bool free_goal_vel), :107-125;
        initial_plan = (x, y, yaw) arrays: plan(const std::vector<geometry_msgs::PoseStamped>& initial_plan, ...), :84-96 (start / goal
        are then its first / last pose)."""
        if initial_plan is not None:
            px, py, pyaw = initial_plan
            start = (px[0], py[0], pyaw[0]); goal = (px[-1], py[-1], pyaw[-1])
        o = self.cfg_.optim
        self.solver.set_config(self.cfg_)
        self.updateAllTEBs(start, goal, start_vel)
        self.exploreEquivalenceClassesAndInitTebs(start, goal, self.cfg_.obstacles.min_obstacle_dist, start_vel, free_goal_vel, initial_plan)
        if self.solver.count == 0:
            self.best_teb_ = -1
            return True
        self.optimizeAllTEBs(o.no_inner_iterations, o.no_outer_iterations)
        self.selectBestTeb()
        return True

    def getVelocityCommand(self, look_ahead_poses=None):
        """(ok, vx, vy, omega) of the best band (:127-139)."""
        if self.best_teb_ < 0:
            return False, 0.0, 0.0, 0.0
        t = self.cfg_.trajectory
        ok, v = self.solver.velocity_command(self.best_teb_, t.control_look_ahead_poses if look_ahead_poses is None else look_ahead_poses,
                                             t.prevent_look_ahead_poses_near_goal)
        return ok, float(v[0]), float(v[1]), float(v[2])

    def isTrajectoryFeasible(self, costmap, footprint_spec, inscribed_radius, circumscribed_radius=0.0, look_ahead_idx=-1,
                             feasibility_check_lookahead_distance=-1.0):
        """HomotopyClassPlanner::isTrajectoryFeasible (src/homotopy_class_planner.cpp:686-709): the best band is checked; an infeasible
        one is removed and the next best tried, unless it was already the best band of the previous tick (then False: "not failing could
        result in oscillations between trajectories")."""
        s = self.solver
        if s.count == 0:
            return False
        s.set_costmap(costmap.cells, costmap.resolution, costmap.origin_x, costmap.origin_y)
        feasible = False
        while not feasible and s.count > 0:
            if self.best_teb_ < 0:                       # findBestTeb (:711-723): re-select when the best band is gone
                self.selectBestTeb()
            best = self.best_teb_
            if best < 0:
                return False
            feasible = s.is_trajectory_feasible(best, footprint_spec, inscribed_radius, self.cfg_.trajectory.min_resolution_collision_check_angular,
                                                look_ahead_idx, feasibility_check_lookahead_distance)[0]
            if not feasible:
                same_as_before = getattr(self, "_last_best_teb", -1) == best
                keep = np.ones(s.count, np.int32)
                keep[best] = 0
                s.compact_bands(keep, -1)                # removeTeb (:725-744)
                lb = getattr(self, "_last_best_teb", -1)
                self._last_best_teb = -1 if lb == best else (lb - 1 if lb > best else lb)
                if 0 <= self.initial_plan_teb_:
                    self.initial_plan_teb_ = -1 if self.initial_plan_teb_ == best else self.initial_plan_teb_ - (self.initial_plan_teb_ > best)
                self.best_teb_ = -1
                if same_as_before:
                    return False
        return feasible

    def bands(self, stride=None):
        """Host copy of the resident bands: list of (x, y, theta, dt)."""
        s = self.solver
        if s.count == 0:
            return []
        b = _abi.TebBatchHost(s.count, stride or s.max_poses)
        s.download(b)
        return [b.get_teb(k) for k in range(s.count)]

    def optimizeAllTEBs(self, iter_innerloop, iter_outerloop):
        h = self.cfg_.hcp
        self.solver.optimize(iter_innerloop, iter_outerloop, True, h.selection_obst_cost_scale,
                             h.selection_viapoint_cost_scale, h.selection_alternative_time_cost)

    def selectBestTeb(self, now=None):
        """selectBestTeb (:564-667). now [s]: wall clock for hcp.switching_blocking_period (:648-663: a switch to another candidate is
        only allowed when more than that period has passed since the last switch); None = time.monotonic()."""
        import time
        last = self.best_teb_
        self._last_best_teb = last                       # last_best_teb_ (:566)
        best, _ = self.solver.select_best(self.best_teb_, self.initial_plan_teb_)
        if last >= 0 and best != last:
            now = time.monotonic() if now is None else now
            if now - getattr(self, "_last_switch", 0.0) > self.cfg_.hcp.switching_blocking_period:
                self._last_switch = now
            else:
                best = last        # switching blocked
        self.best_teb_ = best
        return best

    def results(self):
        self.last_results = self.solver.results()
        return self.last_results

    def download(self):
        return self.solver.download(self.tebs_)
