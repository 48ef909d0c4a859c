"""ctypes mirror of include/teb_amd.h (POD structs only).

Field order and types must stay identical to the header; tests/test_abi.py checks sizeof() against the
value compiled into libteb_amd.so (teb_amd_sizeof_*), so a drift fails loudly instead of corrupting memory.
"""
import ctypes as C

import numpy as np

MAX_FOOTPRINT_VERTICES = 64

# status codes
OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_CAPACITY, ERR_UNSUPPORTED = range(6)
MAX_POSES = 944   # TEB_AMD_MAX_POSES of include/teb_amd.h (tests/test_abi.py keeps the two equal)
TEB_OK, TEB_FAILED, TEB_NONFINITE = range(3)
FOOTPRINT_POINT, FOOTPRINT_CIRCULAR, FOOTPRINT_TWO_CIRCLES, FOOTPRINT_LINE, FOOTPRINT_POLYGON = range(5)
OBST_POINT, OBST_CIRCULAR, OBST_LINE, OBST_PILL, OBST_POLYGON = range(5)
ROT_LEFT, ROT_NONE, ROT_RIGHT = range(3)
JACOBIAN_ANALYTIC, JACOBIAN_G2O_NUMERIC = range(2)

c_i32 = C.c_int32
c_f64 = C.c_double
p_i32 = C.POINTER(C.c_int32)
p_f64 = C.POINTER(C.c_double)


class Config(C.Structure):
    _fields_ = [
        ("teb_autosize", c_i32), ("dt_ref", c_f64), ("dt_hysteresis", c_f64),
        ("min_samples", c_i32), ("max_samples", c_i32), ("exact_arc_length", c_i32),
        ("via_points_ordered", c_i32),
        ("max_vel_x", c_f64), ("max_vel_x_backwards", c_f64), ("max_vel_y", c_f64),
        ("max_vel_trans", c_f64), ("max_vel_theta", c_f64), ("acc_lim_x", c_f64),
        ("acc_lim_y", c_f64), ("acc_lim_theta", c_f64), ("min_turning_radius", c_f64),
        ("min_obstacle_dist", c_f64), ("inflation_dist", c_f64),
        ("dynamic_obstacle_inflation_dist", c_f64), ("include_dynamic_obstacles", c_i32),
        ("obstacle_poses_affected", c_i32), ("legacy_obstacle_association", c_i32),
        ("obstacle_association_force_inclusion_factor", c_f64),
        ("obstacle_association_cutoff_factor", c_f64),
        ("obstacle_proximity_ratio_max_vel", c_f64), ("obstacle_proximity_lower_bound", c_f64),
        ("obstacle_proximity_upper_bound", c_f64),
        ("no_inner_iterations", c_i32), ("no_outer_iterations", c_i32),
        ("optimization_activate", c_i32), ("penalty_epsilon", c_f64),
        ("weight_max_vel_x", c_f64), ("weight_max_vel_y", c_f64), ("weight_max_vel_theta", c_f64),
        ("weight_acc_lim_x", c_f64), ("weight_acc_lim_y", c_f64), ("weight_acc_lim_theta", c_f64),
        ("weight_kinematics_nh", c_f64), ("weight_kinematics_forward_drive", c_f64),
        ("weight_kinematics_turning_radius", c_f64), ("weight_optimaltime", c_f64),
        ("weight_shortest_path", c_f64), ("weight_obstacle", c_f64), ("weight_inflation", c_f64),
        ("weight_dynamic_obstacle", c_f64), ("weight_dynamic_obstacle_inflation", c_f64),
        ("weight_velocity_obstacle_ratio", c_f64), ("weight_viapoint", c_f64),
        ("weight_prefer_rotdir", c_f64), ("weight_adapt_factor", c_f64),
        ("obstacle_cost_exponent", c_f64),
        ("selection_cost_hysteresis", c_f64), ("selection_prefer_initial_plan", c_f64),
        ("selection_obst_cost_scale", c_f64), ("selection_viapoint_cost_scale", c_f64),
        ("selection_alternative_time_cost", c_i32),
        ("divergence_detection_enable", c_i32), ("divergence_detection_max_chi_squared", c_f64),
        ("footprint_type", c_i32), ("footprint_radius", c_f64), ("footprint_front_offset", c_f64),
        ("footprint_front_radius", c_f64), ("footprint_rear_offset", c_f64),
        ("footprint_rear_radius", c_f64), ("footprint_n_vertices", c_i32),
        ("footprint_vx", c_f64 * MAX_FOOTPRINT_VERTICES),
        ("footprint_vy", c_f64 * MAX_FOOTPRINT_VERTICES),
        ("jacobian_mode", c_i32),
    ]


class Obstacles(C.Structure):
    _fields_ = [
        ("count", c_i32), ("type", p_i32), ("ax", p_f64), ("ay", p_f64), ("bx", p_f64), ("by", p_f64),
        ("radius", p_f64), ("vx", p_f64), ("vy", p_f64), ("dynamic", p_i32),
        ("vert_offset", p_i32), ("vert_x", p_f64), ("vert_y", p_f64),
    ]


class TebBatch(C.Structure):
    _fields_ = [
        ("count", c_i32), ("stride", c_i32), ("n", p_i32), ("x", p_f64), ("y", p_f64), ("theta", p_f64),
        ("dt", p_f64), ("has_vel_start", p_i32), ("vel_start", p_f64), ("has_vel_goal", p_i32),
        ("vel_goal", p_f64), ("prefer_rotdir", p_i32), ("via_points_enabled", p_i32),
    ]


class HcpParams(C.Structure):
    _fields_ = [
        ("simple_exploration", c_i32), ("roadmap_graph_no_samples", c_i32), ("roadmap_graph_area_width", c_f64),
        ("roadmap_graph_area_length_scale", c_f64), ("obstacle_heading_threshold", c_f64), ("xy_goal_tolerance", c_f64),
        ("max_number_classes", c_i32), ("max_number_plans_in_current_class", c_i32), ("h_signature_prescaler", c_f64),
        ("h_signature_threshold", c_f64), ("allow_init_with_backwards_motion", c_i32), ("delete_detours_backwards", c_i32),
        ("detours_orientation_tolerance", c_f64), ("length_start_orientation_vector", c_f64),
        ("max_ratio_detours_duration_best_duration", c_f64), ("global_plan_overwrite_orientation", c_i32),
        ("viapoints_all_candidates", c_i32),
    ]


LAYOUT_AUTO, LAYOUT_BLOCKS_LDS, LAYOUT_BAND_LDS, LAYOUT_BAND_HBM = 0, 1, 2, 3
HSIG3D_AUTO, HSIG3D_WIDE, HSIG3D_SMALL = 0, 1, 2


class Options(C.Structure):
    """teb_amd_options_t (include/teb_amd.h): behaviour switches of a handle, fixed at create."""
    _fields_ = [("struct_size", c_i32), ("layout", c_i32), ("fixed_layout", c_i32), ("band_ldlt", c_i32),
                ("generic_distance_path", c_i32), ("hsig3d_kernel", c_i32), ("no_near_cache", c_i32), ("multi_cu", c_i32),
                ("multi_cu_timeout_us", c_i32), ("speculative_trials", c_i32), ("generic_config_path", c_i32), ("compile_for_config", c_i32), ("reserved", c_i32 * 4)]

    def __init__(self, layout=LAYOUT_AUTO, fixed_layout=False, band_ldlt=False, generic_distance_path=False, hsig3d_kernel=HSIG3D_AUTO,
                 no_near_cache=False, multi_cu=0, multi_cu_timeout_us=0, speculative_trials=0, generic_config_path=False, compile_for_config=0):
        super().__init__()
        self.struct_size = C.sizeof(Options)
        self.layout = {"auto": LAYOUT_AUTO, "cr": LAYOUT_BLOCKS_LDS, "band": LAYOUT_BAND_LDS, "bandg": LAYOUT_BAND_HBM}.get(layout, layout)
        self.fixed_layout = int(fixed_layout)
        self.band_ldlt = int(band_ldlt)
        self.generic_distance_path = int(generic_distance_path)
        self.hsig3d_kernel = {"auto": HSIG3D_AUTO, "wide": HSIG3D_WIDE, "small": HSIG3D_SMALL}.get(hsig3d_kernel, hsig3d_kernel)
        self.no_near_cache = int(no_near_cache)
        self.multi_cu = int(multi_cu)   # 0 auto, -1 never, n > 0: at most n helper workgroups per band (teb_amd.h)
        self.multi_cu_timeout_us = int(multi_cu_timeout_us)
        self.speculative_trials = int(speculative_trials)   # 0 auto, -1 never, 1 .. 3 solver workgroups per band
        self.generic_config_path = int(generic_config_path)   # 1: never the kernels specialised on the TebConfig defaults
        self.compile_for_config = int(compile_for_config)     # 0 off, 1 compile for this configuration in the background (hipRTC), 2 wait for it


class Results(C.Structure):
    _fields_ = [
        ("status", p_i32), ("lm_iterations", p_i32), ("lm_trials", p_i32), ("chi2", p_f64),
        ("cost", p_f64), ("lambda_", p_f64),
    ]


def _ptr(a, ty):
    if a is None:
        return C.cast(None, C.POINTER(ty))
    return a.ctypes.data_as(C.POINTER(ty))


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class ObstacleTable:
    """Host-side SoA obstacle table (ObstContainer, obstacles.h:262). Keeps numpy arrays alive."""

    def __init__(self):
        self.type, self.ax, self.ay, self.bx, self.by = [], [], [], [], []
        self.radius, self.vx, self.vy, self.dynamic = [], [], [], []
        self.vert_offset, self.vert_x, self.vert_y = [0], [], []
        self._frozen = None

    def _add(self, ty, ax=0.0, ay=0.0, bx=0.0, by=0.0, r=0.0, verts=None, vel=None):
        self._frozen = None
        self.type.append(ty); self.ax.append(ax); self.ay.append(ay); self.bx.append(bx); self.by.append(by)
        self.radius.append(r)
        if vel is None:
            self.vx.append(0.0); self.vy.append(0.0); self.dynamic.append(0)
        else:  # Obstacle::setCentroidVelocity sets dynamic_ = true (obstacles.h:205)
            self.vx.append(float(vel[0])); self.vy.append(float(vel[1])); self.dynamic.append(1)
        if verts is not None:
            for (px, py) in verts:
                self.vert_x.append(float(px)); self.vert_y.append(float(py))
        self.vert_offset.append(len(self.vert_x))
        return len(self.type) - 1

    def add_point(self, x, y, vel=None):
        return self._add(OBST_POINT, x, y, vel=vel)

    def add_circle(self, x, y, radius, vel=None):
        return self._add(OBST_CIRCULAR, x, y, r=radius, vel=vel)

    def add_line(self, x1, y1, x2, y2, vel=None):
        return self._add(OBST_LINE, x1, y1, x2, y2, vel=vel)

    def add_pill(self, x1, y1, x2, y2, radius, vel=None):
        return self._add(OBST_PILL, x1, y1, x2, y2, r=radius, vel=vel)

    def add_polygon(self, verts, vel=None):
        verts = [tuple(map(float, v)) for v in verts]
        # PolygonObstacle::fixPolygonClosure (src/obstacles.cpp:48-54): drop a duplicated closing vertex
        if len(verts) >= 2 and np.allclose(verts[0], verts[-1], rtol=1e-12, atol=0):
            verts = verts[:-1]
        return self._add(OBST_POLYGON, verts=verts, vel=vel)

    def __len__(self):
        return len(self.type)

    def freeze(self):
        if self._frozen is None:
            arrs = dict(
                type=i32(self.type), ax=f64(self.ax), ay=f64(self.ay), bx=f64(self.bx), by=f64(self.by),
                radius=f64(self.radius), vx=f64(self.vx), vy=f64(self.vy), dynamic=i32(self.dynamic),
                vert_offset=i32(self.vert_offset), vert_x=f64(self.vert_x if self.vert_x else [0.0]),
                vert_y=f64(self.vert_y if self.vert_y else [0.0]))
            st = Obstacles()
            st.count = len(self.type)
            for k in ("type", "dynamic", "vert_offset"):
                setattr(st, k, _ptr(arrs[k], C.c_int32))
            for k in ("ax", "ay", "bx", "by", "radius", "vx", "vy", "vert_x", "vert_y"):
                setattr(st, k, _ptr(arrs[k], C.c_double))
            self._frozen = (st, arrs)
        return self._frozen[0]


class TebBatchHost:
    """Padded SoA batch of candidate TEBs on the host (mirrors teb_amd_teb_batch_t)."""

    def __init__(self, count, stride):
        self.count, self.stride = int(count), int(stride)
        B, S = self.count, self.stride
        self.n = np.zeros(B, np.int32)
        self.x = np.zeros((B, S)); self.y = np.zeros((B, S)); self.theta = np.zeros((B, S)); self.dt = np.zeros((B, S))
        # TebOptimalPlanner::initialize(): start and goal velocity fixed, at zero (src/optimal_planner.cpp:94-102)
        self.has_vel_start = np.ones(B, np.int32); self.vel_start = np.zeros((B, 3))
        self.has_vel_goal = np.ones(B, np.int32); self.vel_goal = np.zeros((B, 3))
        self.prefer_rotdir = np.full(B, ROT_NONE, np.int32)
        self.via_points_enabled = np.ones(B, np.int32)

    def set_teb(self, b, x, y, theta, dt):
        n = len(x)
        assert len(dt) == n - 1 and n <= self.stride
        self.n[b] = n
        self.x[b, :n] = x; self.y[b, :n] = y; self.theta[b, :n] = theta; self.dt[b, :n - 1] = dt
        self.dt[b, n - 1:] = 0

    def get_teb(self, b):
        n = int(self.n[b])
        return (self.x[b, :n].copy(), self.y[b, :n].copy(), self.theta[b, :n].copy(), self.dt[b, :n - 1].copy())

    def copy(self):
        o = TebBatchHost(self.count, self.stride)
        for k in ("n", "x", "y", "theta", "dt", "has_vel_start", "vel_start", "has_vel_goal", "vel_goal",
                  "prefer_rotdir", "via_points_enabled"):
            setattr(o, k, getattr(self, k).copy())
        return o

    def c_struct(self):
        st = TebBatch()
        st.count, st.stride = self.count, self.stride
        st.n = _ptr(self.n, C.c_int32)
        for k in ("x", "y", "theta", "dt", "vel_start", "vel_goal"):
            a = getattr(self, k)
            assert a.flags["C_CONTIGUOUS"] and a.dtype == np.float64
            setattr(st, k, _ptr(a, C.c_double))
        for k in ("has_vel_start", "has_vel_goal", "prefer_rotdir", "via_points_enabled"):
            setattr(st, k, _ptr(getattr(self, k), C.c_int32))
        return st


class ResultsHost:
    def __init__(self, count):
        self.status = np.zeros(count, np.int32)
        self.lm_iterations = np.zeros(count, np.int32)
        self.lm_trials = np.zeros(count, np.int32)
        self.chi2 = np.zeros(count)
        self.cost = np.full(count, np.nan)
        self.lambda_ = np.zeros(count)

    def c_struct(self):
        st = Results()
        st.status = _ptr(self.status, C.c_int32)
        st.lm_iterations = _ptr(self.lm_iterations, C.c_int32)
        st.lm_trials = _ptr(self.lm_trials, C.c_int32)
        st.chi2 = _ptr(self.chi2, C.c_double)
        st.cost = _ptr(self.cost, C.c_double)
        st.lambda_ = _ptr(self.lambda_, C.c_double)
        return st
