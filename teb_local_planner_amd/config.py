"""TebConfig mirror (include/teb_local_planner/teb_config.h:62-430 of the reference).

Same group / field names and the same constructor defaults (teb_config.h:245-390) for every field the
optimiser path reads. ROS-only fields (topics, costmap converter, visualisation,
goal tolerance, ...) are not carried: they are outside the hot path.
"""
import math
from types import SimpleNamespace

from . import _abi


class RobotFootprintModel:
    """Flattened robot_model (robot_footprint_model.h:58-770)."""

    def __init__(self, type=_abi.FOOTPRINT_POINT, radius=0.0, front_offset=0.0, front_radius=0.0,
                 rear_offset=0.0, rear_radius=0.0, vertices=()):
        self.type = type
        self.radius = radius
        self.front_offset, self.front_radius = front_offset, front_radius
        self.rear_offset, self.rear_radius = rear_offset, rear_radius
        self.vertices = [tuple(map(float, v)) for v in vertices]

    @staticmethod
    def point():
        return RobotFootprintModel(_abi.FOOTPRINT_POINT)

    @staticmethod
    def circular(radius):
        return RobotFootprintModel(_abi.FOOTPRINT_CIRCULAR, radius=radius)

    @staticmethod
    def two_circles(front_offset, front_radius, rear_offset, rear_radius):
        return RobotFootprintModel(_abi.FOOTPRINT_TWO_CIRCLES, front_offset=front_offset,
                                   front_radius=front_radius, rear_offset=rear_offset, rear_radius=rear_radius)

    @staticmethod
    def line(start, end):
        return RobotFootprintModel(_abi.FOOTPRINT_LINE, vertices=[start, end])

    @staticmethod
    def polygon(vertices):
        return RobotFootprintModel(_abi.FOOTPRINT_POLYGON, vertices=vertices)


class TebConfig:
    def __init__(self):
        self.robot_model = RobotFootprintModel.point()
        self.trajectory = SimpleNamespace(
            teb_autosize=True, dt_ref=0.3, dt_hysteresis=0.1, min_samples=3, max_samples=500,
            exact_arc_length=False, via_points_ordered=False,
            # warm start / read-out parameters of plan() and getVelocityCommand() (teb_config.h:259-273); they are arguments of the
            # C-ABI entry points, not fields of teb_amd_config_t
            global_plan_overwrite_orientation=True, allow_init_with_backwards_motion=False, force_reinit_new_goal_dist=1.0,
            force_reinit_new_goal_angular=0.5 * 3.141592653589793, control_look_ahead_poses=1,
            prevent_look_ahead_poses_near_goal=0, feasibility_check_no_poses=5, feasibility_check_lookahead_distance=-1.0, min_resolution_collision_check_angular=3.141592653589793)
        self.robot = SimpleNamespace(
            max_vel_x=0.4, max_vel_x_backwards=0.2, max_vel_y=0.0, max_vel_trans=0.0, max_vel_theta=0.3,
            acc_lim_x=0.5, acc_lim_y=0.5, acc_lim_theta=0.5, min_turning_radius=0.0)
        self.obstacles = SimpleNamespace(
            min_obstacle_dist=0.5, inflation_dist=0.6, dynamic_obstacle_inflation_dist=0.6,
            include_dynamic_obstacles=True, obstacle_poses_affected=25, legacy_obstacle_association=False,
            obstacle_association_force_inclusion_factor=1.5, obstacle_association_cutoff_factor=5.0,
            obstacle_proximity_ratio_max_vel=1.0, obstacle_proximity_lower_bound=0.0,
            obstacle_proximity_upper_bound=0.5)
        self.optim = SimpleNamespace(
            no_inner_iterations=5, no_outer_iterations=4, optimization_activate=True, penalty_epsilon=0.05,
            weight_max_vel_x=2.0, weight_max_vel_y=2.0, weight_max_vel_theta=1.0, weight_acc_lim_x=1.0,
            weight_acc_lim_y=1.0, weight_acc_lim_theta=1.0, weight_kinematics_nh=1000.0,
            weight_kinematics_forward_drive=1.0, weight_kinematics_turning_radius=1.0,
            weight_optimaltime=1.0, weight_shortest_path=0.0, weight_obstacle=50.0, weight_inflation=0.1,
            weight_dynamic_obstacle=50.0, weight_dynamic_obstacle_inflation=0.1,
            weight_velocity_obstacle_ratio=0.0, weight_viapoint=1.0, weight_prefer_rotdir=50.0,
            weight_adapt_factor=2.0, obstacle_cost_exponent=1.0)
        self.hcp = SimpleNamespace(
            enable_multithreading=True, max_number_classes=5, selection_cost_hysteresis=1.0,
            selection_prefer_initial_plan=0.95, selection_obst_cost_scale=100.0,
            selection_viapoint_cost_scale=1.0, selection_alternative_time_cost=False,
            viapoints_all_candidates=True,
            # candidate generation (teb_config.h:352-367; max_number_plans_in_current_class has no constructor default in the
            # reference, its dynamic-reconfigure default is 1): fields of teb_amd_hcp_params_t
            simple_exploration=False, max_number_plans_in_current_class=1, obstacle_heading_threshold=0.45,
            roadmap_graph_no_samples=15, roadmap_graph_area_width=6.0, roadmap_graph_area_length_scale=1.0,
            h_signature_prescaler=1.0, h_signature_threshold=0.1, delete_detours_backwards=True,
            detours_orientation_tolerance=0.5 * 3.141592653589793, length_start_orientation_vector=0.4,
            max_ratio_detours_duration_best_duration=3.0, selection_dropping_probability=0.0, switching_blocking_period=0.0)
        self.goal_tolerance = SimpleNamespace(xy_goal_tolerance=0.2, yaw_goal_tolerance=0.2)
        self.recovery = SimpleNamespace(divergence_detection_enable=False,
                                        divergence_detection_max_chi_squared=10.0)
        # extension (not in the reference): Jacobian mode of the GPU path / oracle
        self.jacobian_mode = _abi.JACOBIAN_ANALYTIC

    def hcp_params(self):
        """teb_amd_hcp_params_t of this configuration (candidate generation, include/teb_amd.h)."""
        p = _abi.HcpParams()
        for k, _ in _abi.HcpParams._fields_:
            if hasattr(self.hcp, k):
                v = getattr(self.hcp, k)
                setattr(p, k, int(v) if isinstance(v, bool) else v)
        p.xy_goal_tolerance = self.goal_tolerance.xy_goal_tolerance
        p.allow_init_with_backwards_motion = int(self.trajectory.allow_init_with_backwards_motion)
        p.global_plan_overwrite_orientation = int(self.trajectory.global_plan_overwrite_orientation)
        return p

    def to_c(self):
        c = _abi.Config()
        for grp in (self.trajectory, self.robot, self.obstacles, self.optim, self.hcp, self.recovery):
            for k, v in vars(grp).items():
                if hasattr(c, k):
                    setattr(c, k, int(v) if isinstance(v, bool) else v)
        m = self.robot_model
        c.footprint_type = m.type
        c.footprint_radius = m.radius
        c.footprint_front_offset, c.footprint_front_radius = m.front_offset, m.front_radius
        c.footprint_rear_offset, c.footprint_rear_radius = m.rear_offset, m.rear_radius
        if len(m.vertices) > _abi.MAX_FOOTPRINT_VERTICES:
            raise ValueError("footprint has more than %d vertices" % _abi.MAX_FOOTPRINT_VERTICES)
        c.footprint_n_vertices = len(m.vertices)
        for i, (vx, vy) in enumerate(m.vertices):
            c.footprint_vx[i] = vx
            c.footprint_vy[i] = vy
        c.jacobian_mode = self.jacobian_mode
        return c


def normalize_theta(theta):
    """g2o::normalize_theta -> [-pi, pi)."""
    if -math.pi <= theta < math.pi:
        return theta
    m = math.floor(theta / (2 * math.pi))
    theta = theta - m * 2 * math.pi
    if theta >= math.pi:
        theta -= 2 * math.pi
    if theta < -math.pi:
        theta += 2 * math.pi
    return theta
