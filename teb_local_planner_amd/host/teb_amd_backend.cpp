// teb_amd_backend.cpp — see teb_amd_backend.h. Host C++ above the C-ABI, written against the reference's classes.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <sstream>
#include <stack>
#include <string>
#include <vector>

// The five footprint classes keep their parameters private and offer no getters (robot_footprint_model.h:227, 306, 424,
// 592, 768). host/patches/footprint_and_signature_accessors.patch adds them (build with TEB_AMD_BACKEND_HAVE_ACCESSORS, see
// host/CMakeLists.txt); without the patch this out-of-tree build reads the members directly instead.
#ifdef TEB_AMD_BACKEND_HAVE_ACCESSORS
#define FP_RADIUS(m) (m)->getRadius()
#define FP_FRONT_OFFSET(m) (m)->getFrontOffset()
#define FP_FRONT_RADIUS(m) (m)->getFrontRadius()
#define FP_REAR_OFFSET(m) (m)->getRearOffset()
#define FP_REAR_RADIUS(m) (m)->getRearRadius()
#define FP_LINE_START(m) (m)->getLineStart()
#define FP_LINE_END(m) (m)->getLineEnd()
#define FP_VERTICES(m) (m)->getVertices()
#else
#define private public
#include <teb_local_planner/robot_footprint_model.h>
#undef private
#define FP_RADIUS(m) (m)->radius_
#define FP_FRONT_OFFSET(m) (m)->front_offset_
#define FP_FRONT_RADIUS(m) (m)->front_radius_
#define FP_REAR_OFFSET(m) (m)->rear_offset_
#define FP_REAR_RADIUS(m) (m)->rear_radius_
#define FP_LINE_START(m) (m)->line_start_
#define FP_LINE_END(m) (m)->line_end_
#define FP_VERTICES(m) (m)->vertices_
#endif

#include "teb_amd_backend.h"

namespace teb_local_planner {

// TimedElasticBand::addPoseAndTimeDiff asserts dt > 0 (src/timed_elastic_band.cpp:100-105); the optimiser itself is not bound by that
// (g2o updates the vertices in place, a time difference may end up <= 0 exactly as in the reference), so a band that comes back
// from the device is rebuilt vertex by vertex without the assertion.
static void appendPoseAndTimeDiff(TimedElasticBand& t, double x, double y, double theta, double dt)
{
  t.addPose(x, y, theta, false);
  t.timediffs().push_back(new VertexTimeDiff(dt, false));
}


namespace {
int g_jacobian_mode = TEB_AMD_JACOBIAN_ANALYTIC;
}

void setAmdJacobianMode(int mode) { g_jacobian_mode = mode; }

bool toAmdFootprint(const BaseRobotFootprintModel& model, teb_amd_config_t& a)
{
  a.footprint_radius = 0; a.footprint_front_offset = a.footprint_front_radius = 0;
  a.footprint_rear_offset = a.footprint_rear_radius = 0; a.footprint_n_vertices = 0;
  if (dynamic_cast<const PointRobotFootprint*>(&model)) { a.footprint_type = TEB_AMD_FOOTPRINT_POINT; return true; }
  if (const CircularRobotFootprint* m = dynamic_cast<const CircularRobotFootprint*>(&model))
  {
    a.footprint_type = TEB_AMD_FOOTPRINT_CIRCULAR;
    a.footprint_radius = FP_RADIUS(m);
    return true;
  }
  if (const TwoCirclesRobotFootprint* m = dynamic_cast<const TwoCirclesRobotFootprint*>(&model))
  {
    a.footprint_type = TEB_AMD_FOOTPRINT_TWO_CIRCLES;
    a.footprint_front_offset = FP_FRONT_OFFSET(m); a.footprint_front_radius = FP_FRONT_RADIUS(m);
    a.footprint_rear_offset = FP_REAR_OFFSET(m);   a.footprint_rear_radius = FP_REAR_RADIUS(m);
    return true;
  }
  if (const LineRobotFootprint* m = dynamic_cast<const LineRobotFootprint*>(&model))
  {
    a.footprint_type = TEB_AMD_FOOTPRINT_LINE;
    a.footprint_n_vertices = 2;
    a.footprint_vx[0] = FP_LINE_START(m).x(); a.footprint_vy[0] = FP_LINE_START(m).y();
    a.footprint_vx[1] = FP_LINE_END(m).x();   a.footprint_vy[1] = FP_LINE_END(m).y();
    return true;
  }
  if (const PolygonRobotFootprint* m = dynamic_cast<const PolygonRobotFootprint*>(&model))
  {
    if ((int)FP_VERTICES(m).size() > TEB_AMD_MAX_FOOTPRINT_VERTICES) return false;
    a.footprint_type = TEB_AMD_FOOTPRINT_POLYGON;
    a.footprint_n_vertices = (int32_t)FP_VERTICES(m).size();
    for (int i = 0; i < a.footprint_n_vertices; ++i) { a.footprint_vx[i] = FP_VERTICES(m)[i].x(); a.footprint_vy[i] = FP_VERTICES(m)[i].y(); }
    return true;
  }
  return false;
}

bool toAmdConfig(const TebConfig& c, teb_amd_config_t& a)
{
  teb_amd_config_default(&a);
  a.teb_autosize = c.trajectory.teb_autosize ? 1 : 0;   a.dt_ref = c.trajectory.dt_ref;
  a.dt_hysteresis = c.trajectory.dt_hysteresis;          a.min_samples = c.trajectory.min_samples;
  a.max_samples = c.trajectory.max_samples;              a.exact_arc_length = c.trajectory.exact_arc_length;
  a.via_points_ordered = c.trajectory.via_points_ordered;
  a.max_vel_x = c.robot.max_vel_x;                       a.max_vel_x_backwards = c.robot.max_vel_x_backwards;
  a.max_vel_y = c.robot.max_vel_y;                       a.max_vel_trans = c.robot.max_vel_trans;
  a.max_vel_theta = c.robot.max_vel_theta;               a.acc_lim_x = c.robot.acc_lim_x;
  a.acc_lim_y = c.robot.acc_lim_y;                       a.acc_lim_theta = c.robot.acc_lim_theta;
  a.min_turning_radius = c.robot.min_turning_radius;
  a.min_obstacle_dist = c.obstacles.min_obstacle_dist;   a.inflation_dist = c.obstacles.inflation_dist;
  a.dynamic_obstacle_inflation_dist = c.obstacles.dynamic_obstacle_inflation_dist;
  a.include_dynamic_obstacles = c.obstacles.include_dynamic_obstacles;
  a.obstacle_poses_affected = c.obstacles.obstacle_poses_affected;
  a.legacy_obstacle_association = c.obstacles.legacy_obstacle_association;
  a.obstacle_association_force_inclusion_factor = c.obstacles.obstacle_association_force_inclusion_factor;
  a.obstacle_association_cutoff_factor = c.obstacles.obstacle_association_cutoff_factor;
  a.obstacle_proximity_ratio_max_vel = c.obstacles.obstacle_proximity_ratio_max_vel;
  a.obstacle_proximity_lower_bound = c.obstacles.obstacle_proximity_lower_bound;
  a.obstacle_proximity_upper_bound = c.obstacles.obstacle_proximity_upper_bound;
  a.no_inner_iterations = c.optim.no_inner_iterations;   a.no_outer_iterations = c.optim.no_outer_iterations;
  a.optimization_activate = c.optim.optimization_activate; a.penalty_epsilon = c.optim.penalty_epsilon;
  a.weight_max_vel_x = c.optim.weight_max_vel_x;         a.weight_max_vel_y = c.optim.weight_max_vel_y;
  a.weight_max_vel_theta = c.optim.weight_max_vel_theta; a.weight_acc_lim_x = c.optim.weight_acc_lim_x;
  a.weight_acc_lim_y = c.optim.weight_acc_lim_y;         a.weight_acc_lim_theta = c.optim.weight_acc_lim_theta;
  a.weight_kinematics_nh = c.optim.weight_kinematics_nh;
  a.weight_kinematics_forward_drive = c.optim.weight_kinematics_forward_drive;
  a.weight_kinematics_turning_radius = c.optim.weight_kinematics_turning_radius;
  a.weight_optimaltime = c.optim.weight_optimaltime;     a.weight_shortest_path = c.optim.weight_shortest_path;
  a.weight_obstacle = c.optim.weight_obstacle;           a.weight_inflation = c.optim.weight_inflation;
  a.weight_dynamic_obstacle = c.optim.weight_dynamic_obstacle;
  a.weight_dynamic_obstacle_inflation = c.optim.weight_dynamic_obstacle_inflation;
  a.weight_velocity_obstacle_ratio = c.optim.weight_velocity_obstacle_ratio;
  a.weight_viapoint = c.optim.weight_viapoint;           a.weight_prefer_rotdir = c.optim.weight_prefer_rotdir;
  a.weight_adapt_factor = c.optim.weight_adapt_factor;   a.obstacle_cost_exponent = c.optim.obstacle_cost_exponent;
  a.selection_cost_hysteresis = c.hcp.selection_cost_hysteresis;
  a.selection_prefer_initial_plan = c.hcp.selection_prefer_initial_plan;
  a.selection_obst_cost_scale = c.hcp.selection_obst_cost_scale;
  a.selection_viapoint_cost_scale = c.hcp.selection_viapoint_cost_scale;
  a.selection_alternative_time_cost = c.hcp.selection_alternative_time_cost;
  a.divergence_detection_enable = c.recovery.divergence_detection_enable;
  a.divergence_detection_max_chi_squared = c.recovery.divergence_detection_max_chi_squared;
  a.jacobian_mode = g_jacobian_mode;
  // Fail closed: a footprint the device cannot represent (unknown class, polygon with more than TEB_AMD_MAX_FOOTPRINT_VERTICES
  // vertices) must never be replaced by a smaller one - obstacle clearance would be optimised for the wrong robot.
  if (!c.robot_model || !toAmdFootprint(*c.robot_model, a))
  {
    ROS_ERROR("toAmdConfig(): the robot footprint model cannot be represented on the device (unknown class or more than %d polygon vertices); "
              "optimizeTEB() will fail instead of planning with a different footprint", TEB_AMD_MAX_FOOTPRINT_VERTICES);
    return false;
  }
  return true;
}

void AmdObstacleTable::assign(const ObstContainer* obstacles)
{
  type.clear(); dynamic.clear(); vert_offset.assign(1, 0);
  ax.clear(); ay.clear(); bx.clear(); by.clear(); radius.clear(); vx.clear(); vy.clear(); vert_x.clear(); vert_y.clear();
  if (!obstacles) return;
  for (const ObstaclePtr& p : *obstacles)
  {
    double x0 = 0, y0 = 0, x1 = 0, y1 = 0, r = 0;
    int32_t t;
    if (const PointObstacle* q = dynamic_cast<const PointObstacle*>(p.get())) { t = TEB_AMD_OBST_POINT; x0 = q->x(); y0 = q->y(); }
    else if (const CircularObstacle* q = dynamic_cast<const CircularObstacle*>(p.get())) { t = TEB_AMD_OBST_CIRCULAR; x0 = q->x(); y0 = q->y(); r = q->radius(); }
    else if (const LineObstacle* q = dynamic_cast<const LineObstacle*>(p.get()))
    { t = TEB_AMD_OBST_LINE; x0 = q->start().x(); y0 = q->start().y(); x1 = q->end().x(); y1 = q->end().y(); }
    else if (const PillObstacle* q = dynamic_cast<const PillObstacle*>(p.get()))
    {
      t = TEB_AMD_OBST_PILL; x0 = q->start().x(); y0 = q->start().y(); x1 = q->end().x(); y1 = q->end().y();
      r = -q->getMinimumDistance(q->start());   // no radius getter: distance of the start point is 0 - radius_ (obstacles.h:800-803)
    }
    else if (const PolygonObstacle* q = dynamic_cast<const PolygonObstacle*>(p.get()))
    {
      t = TEB_AMD_OBST_POLYGON;
      for (const Eigen::Vector2d& v : q->vertices()) { vert_x.push_back(v.x()); vert_y.push_back(v.y()); }
    }
    else { ROS_ERROR("AmdObstacleTable: unknown obstacle class skipped"); continue; }
    type.push_back(t); ax.push_back(x0); ay.push_back(y0); bx.push_back(x1); by.push_back(y1); radius.push_back(r);
    vx.push_back(p->getCentroidVelocity().x()); vy.push_back(p->getCentroidVelocity().y());
    dynamic.push_back(p->isDynamic() ? 1 : 0);
    vert_offset.push_back((int32_t)vert_x.size());
  }
}

teb_amd_obstacles_t AmdObstacleTable::view() const
{
  teb_amd_obstacles_t o;
  std::memset(&o, 0, sizeof(o));
  o.count = (int32_t)type.size();
  o.type = type.data(); o.ax = ax.data(); o.ay = ay.data(); o.bx = bx.data(); o.by = by.data(); o.radius = radius.data();
  o.vx = vx.data(); o.vy = vy.data(); o.dynamic = dynamic.data();
  o.vert_offset = vert_offset.data(); o.vert_x = vert_x.data(); o.vert_y = vert_y.data();
  return o;
}

// ---- TebOptimalPlannerAmd --------------------------------------------------------------------------------------------

TebOptimalPlannerAmd::TebOptimalPlannerAmd(const TebConfig& cfg, ObstContainer* obstacles, TebVisualizationPtr visual,
                                           const ViaPointContainer* via_points)
    : TebOptimalPlanner(cfg, obstacles, visual, via_points)
{
}

bool TebOptimalPlannerAmd::optimizeTEB(int iterations_innerloop, int iterations_outerloop, bool compute_cost_afterwards,
                                       double obst_cost_scale, double viapoint_cost_scale, bool alternative_time_cost)
{
  if (cfg_->optim.optimization_activate == false) return false;   // src/optimal_planner.cpp:186-187
  AmdObstacleTable probe;
  probe.assign(obstacles_);
  // Capacities: whatever autoResize can produce (max_samples + 1 poses; the layout is chosen per launch, so a large capacity does not
  // slow down short bands), and head-room over the obstacle / vertex / via-point counts of THIS tick - the costmap converter changes
  // them every tick, so the handle is rebuilt with room to spare whenever a count outgrows it.
  const int need_poses = std::max(teb_.sizePoses(), std::min(cfg_->trajectory.max_samples + 1, TEB_AMD_MAX_POSES));
  const int n_obst = (int)probe.type.size(), n_vert = probe.vertices(), n_via = via_points_ ? (int)via_points_->size() : 0;
  if (!single_ || !single_->valid() || teb_.sizePoses() > single_->maxPoses() || n_obst > single_->maxObstacles() ||
      n_vert > single_->maxObstacleVertices() || n_via > single_->maxViaPoints())
    single_.reset(new TebAmdBatch(*cfg_, 1, need_poses, std::max(16, 2 * n_obst), std::max(64, 2 * n_vert), std::max(8, 2 * n_via)));
  if (!single_->valid()) return false;
  std::vector<TebOptimalPlannerAmd*> one(1, this);
  return single_->optimizeAllTEBs(one, iterations_innerloop, iterations_outerloop, compute_cost_afterwards, obst_cost_scale,
                                  viapoint_cost_scale, alternative_time_cost) == 1;
}

// The sequence of TebOptimalPlanner::plan (src/optimal_planner.cpp:247-320) on the host band: a band that exists and whose goal moved by
// less than force_reinit_new_goal_dist / _angular is warm-started (updateAndPruneTEB), anything else is (re)built by `init`; then the
// start velocity, the goal-velocity flag and the optimisation - this class's, on the device.
template <class Init>
bool TebOptimalPlannerAmd::planOnBand(const PoseSE2& start, const PoseSE2& goal, Init init, const geometry_msgs::Twist* start_vel, bool free_goal_vel)
{
  ROS_ASSERT_MSG(initialized_, "Call initialize() first.");
  bool keep = false;
  if (teb_.isInit())
  {
    keep = teb_.sizePoses() > 0
        && (goal.position() - teb_.BackPose().position()).norm() < cfg_->trajectory.force_reinit_new_goal_dist
        && fabs(g2o::normalize_theta(goal.theta() - teb_.BackPose().theta())) < cfg_->trajectory.force_reinit_new_goal_angular;
    if (keep) teb_.updateAndPruneTEB(start, goal, cfg_->trajectory.min_samples);
    else teb_.clearTimedElasticBand();   // the goal jumped: start over
  }
  if (!keep) init();
  if (start_vel) setVelocityStart(*start_vel);
  if (free_goal_vel) setVelocityGoalFree();
  else vel_goal_.first = true;           // the previously set goal velocity counts again (:274-275, :315-316)
  return optimizeTEB(cfg_->optim.no_inner_iterations, cfg_->optim.no_outer_iterations);
}

bool TebOptimalPlannerAmd::plan(const std::vector<geometry_msgs::PoseStamped>& initial_plan, const geometry_msgs::Twist* start_vel, bool free_goal_vel)
{
  const PoseSE2 start(initial_plan.front().pose), goal(initial_plan.back().pose);
  return planOnBand(start, goal, [&]() {
    teb_.initTrajectoryToGoal(initial_plan, cfg_->robot.max_vel_x, cfg_->robot.max_vel_theta, cfg_->trajectory.global_plan_overwrite_orientation,
                              cfg_->trajectory.min_samples, cfg_->trajectory.allow_init_with_backwards_motion); }, start_vel, free_goal_vel);
}

bool TebOptimalPlannerAmd::plan(const tf::Pose& start, const tf::Pose& goal, const geometry_msgs::Twist* start_vel, bool free_goal_vel)
{
  (void)free_goal_vel;   // the reference's overload forwards start_vel only (src/optimal_planner.cpp:283-288): the goal velocity stays fixed
  return plan(PoseSE2(start), PoseSE2(goal), start_vel);
}

bool TebOptimalPlannerAmd::plan(const PoseSE2& start, const PoseSE2& goal, const geometry_msgs::Twist* start_vel, bool free_goal_vel)
{
  return planOnBand(start, goal, [&]() {   // no intermediate samples, dt = 1: autoResize inserts poses before the first optimisation
    teb_.initTrajectoryToGoal(start, goal, 0, cfg_->robot.max_vel_x, cfg_->trajectory.min_samples, cfg_->trajectory.allow_init_with_backwards_motion); },
    start_vel, free_goal_vel);
}

bool TebOptimalPlannerAmd::hasDiverged() const
{
  if (!cfg_->recovery.divergence_detection_enable) return false;                         // src/optimal_planner.cpp:1026-1027
  if (!stats_available_) return false;                                                   // "no statistics yet", :1031-1033
  return stats_back_chi2_ > cfg_->recovery.divergence_detection_max_chi_squared;         // :1036-1038
}

// ---- TebAmdBatch -------------------------------------------------------------------------------------------------

TebAmdBatch::TebAmdBatch(const TebConfig& cfg, int max_tebs, int max_poses, int max_obstacles, int max_obstacle_vertices,
                         int max_via_points, int device)
    : max_tebs_(max_tebs), max_poses_(max_poses), max_obstacles_(max_obstacles), max_obstacle_vertices_(max_obstacle_vertices),
      max_via_points_(max_via_points)
{
  teb_amd_config_t a;
  if (!toAmdConfig(cfg, a)) { check(TEB_AMD_ERR_UNSUPPORTED, "toAmdConfig (robot footprint not representable)"); return; }   // h_ stays NULL
  check(teb_amd_create(&a, max_tebs, max_poses, max_obstacles, max_obstacle_vertices, max_via_points, device, NULL, &h_),
        "teb_amd_create");
}

TebAmdBatch::~TebAmdBatch() { if (h_) teb_amd_destroy(h_); }

bool TebAmdBatch::check(int rc, const char* what)
{
  if (rc == TEB_AMD_OK) return true;
  std::ostringstream s;
  s << what << " failed (" << rc << "): " << teb_amd_last_error();
  error_ = s.str();
  ROS_ERROR("TebAmdBatch: %s", error_.c_str());   // no CPU fallback: the caller sees optimizeTEB() == false
  return false;
}

bool TebAmdBatch::setCostmap(const unsigned char* cells, unsigned int size_x, unsigned int size_y, double resolution, double origin_x,
                             double origin_y)
{
  if (!h_) return false;
  return check(teb_amd_set_costmap(h_, cells, (int32_t)size_x, (int32_t)size_y, resolution, origin_x, origin_y), "teb_amd_set_costmap");
}

bool TebAmdBatch::isTrajectoryFeasible(int index, const std::vector<geometry_msgs::Point>& footprint_spec, double inscribed_radius,
                                       double min_resolution_collision_check_angular, int look_ahead_idx,
                                       double feasibility_check_lookahead_distance)
{
  if (!h_) return false;
  std::vector<double> fx, fy;
  for (const geometry_msgs::Point& p : footprint_spec) { fx.push_back(p.x); fy.push_back(p.y); }
  int32_t feasible = 0;
  if (!check(teb_amd_is_trajectory_feasible(h_, index, (int32_t)fx.size(), fx.data(), fy.data(), inscribed_radius,
                                            min_resolution_collision_check_angular, look_ahead_idx, feasibility_check_lookahead_distance,
                                            &feasible, NULL), "teb_amd_is_trajectory_feasible")) return false;
  return feasible != 0;
}

float TebAmdBatch::lastKernelMs() const
{
  float ms = 0;
  if (h_) teb_amd_last_kernel_ms(h_, &ms);
  return ms;
}

int TebAmdBatch::optimizeAllTEBs(const std::vector<TebOptimalPlannerAmd*>& tebs, int iter_innerloop, int iter_outerloop,
                                 bool compute_cost_afterwards, double obst_cost_scale, double viapoint_cost_scale,
                                 bool alternative_time_cost)
{
  const int B = (int)tebs.size(), S = max_poses_;
  last_call_ok_ = h_ != NULL;
  if (!h_ || B == 0) return 0;
  last_call_ok_ = false;   // until the results are back
  if (B > max_tebs_) { check(TEB_AMD_ERR_CAPACITY, "optimizeAllTEBs (more candidates than max_tebs)"); return 0; }
  TebOptimalPlannerAmd& first = *tebs.front();
  if (first.cfg_->optim.optimization_activate == false) { last_call_ok_ = true; return 0; }

  // scene: TebConfig (dynamic_reconfigure may have changed it), obstacles, via-points
  teb_amd_config_t a;
  if (!toAmdConfig(*first.cfg_, a)) { check(TEB_AMD_ERR_UNSUPPORTED, "toAmdConfig (robot footprint not representable)"); return 0; }
  AmdObstacleTable table;
  table.assign(first.obstacles_);
  teb_amd_obstacles_t ov = table.view();
  std::vector<double> viax, viay;
  // the via-point container is shared (HomotopyClassPlanner hands the same pointer to every candidate that has via-points at all:
  // updateReferenceTrajectoryViaPoints, :286-315); candidates without via-points carry NULL and get their flag cleared below
  const ViaPointContainer* shared_via = NULL;
  for (TebOptimalPlannerAmd* t : tebs) if (t->via_points_) { shared_via = t->via_points_; break; }
  if (shared_via)
    for (const Eigen::Vector2d& v : *shared_via) { viax.push_back(v.x()); viay.push_back(v.y()); }
  // every error of set_config is a real one (ABI 2: it re-derives the obstacle lists itself when include_dynamic_obstacles / the
  // footprint type change, and keeps the previous configuration when the new one is invalid)
  if (!check(teb_amd_set_config(h_, &a), "teb_amd_set_config")) return 0;
  if (!check(teb_amd_set_obstacles(h_, &ov), "teb_amd_set_obstacles")) return 0;
  if (!check(teb_amd_set_via_points(h_, (int32_t)viax.size(), viax.data(), viay.data()), "teb_amd_set_via_points")) return 0;

  // bands: TimedElasticBand -> SoA strips
  std::vector<int32_t> n(B), hvs(B), hvg(B), rot(B), via(B);
  std::vector<double> x((size_t)B * S, 0.0), y((size_t)B * S, 0.0), th((size_t)B * S, 0.0), dt((size_t)B * S, 0.0), vs(3 * (size_t)B), vg(3 * (size_t)B);
  for (int b = 0; b < B; ++b)
  {
    TebOptimalPlannerAmd& p = *tebs[b];
    p.optimized_ = false;
    const TimedElasticBand& t = p.teb_;
    if (t.sizePoses() > S) { check(TEB_AMD_ERR_CAPACITY, "optimizeAllTEBs (band longer than max_poses)"); return 0; }
    n[b] = t.sizePoses();
    for (int i = 0; i < t.sizePoses(); ++i) { x[(size_t)b * S + i] = t.Pose(i).x(); y[(size_t)b * S + i] = t.Pose(i).y(); th[(size_t)b * S + i] = t.Pose(i).theta(); }
    for (int i = 0; i < t.sizeTimeDiffs(); ++i) dt[(size_t)b * S + i] = t.TimeDiff(i);
    hvs[b] = p.vel_start_.first; hvg[b] = p.vel_goal_.first;
    vs[3 * b] = p.vel_start_.second.linear.x; vs[3 * b + 1] = p.vel_start_.second.linear.y; vs[3 * b + 2] = p.vel_start_.second.angular.z;
    vg[3 * b] = p.vel_goal_.second.linear.x;  vg[3 * b + 1] = p.vel_goal_.second.linear.y;  vg[3 * b + 2] = p.vel_goal_.second.angular.z;
    rot[b] = p.prefer_rotdir_ == RotType::left ? TEB_AMD_ROT_LEFT : (p.prefer_rotdir_ == RotType::right ? TEB_AMD_ROT_RIGHT : TEB_AMD_ROT_NONE);
    via[b] = p.via_points_ != NULL;
  }
  teb_amd_teb_batch_t batch;
  std::memset(&batch, 0, sizeof(batch));
  batch.count = B; batch.stride = S; batch.n = n.data(); batch.x = x.data(); batch.y = y.data(); batch.theta = th.data(); batch.dt = dt.data();
  batch.has_vel_start = hvs.data(); batch.vel_start = vs.data(); batch.has_vel_goal = hvg.data(); batch.vel_goal = vg.data();
  batch.prefer_rotdir = rot.data(); batch.via_points_enabled = via.data();
  if (!check(teb_amd_upload_tebs(h_, &batch), "teb_amd_upload_tebs")) return 0;

  // the whole outer x inner loop of every candidate: one launch
  if (!check(teb_amd_optimize_batch(h_, iter_innerloop, iter_outerloop, compute_cost_afterwards ? 1 : 0, obst_cost_scale,
                                    viapoint_cost_scale, alternative_time_cost ? 1 : 0), "teb_amd_optimize_batch")) return 0;
  std::vector<int32_t> status(B), iters(B), trials(B);
  std::vector<double> chi2(B), cost(B), lambda(B);
  teb_amd_results_t res = { status.data(), iters.data(), trials.data(), chi2.data(), cost.data(), lambda.data() };
  if (!check(teb_amd_get_results(h_, &res), "teb_amd_get_results")) return 0;
  std::vector<int32_t> stats(B);
  std::vector<double> back_chi2(B);
  if (!check(teb_amd_get_batch_statistics(h_, stats.data(), back_chi2.data()), "teb_amd_get_batch_statistics")) return 0;
  if (!check(teb_amd_download_tebs(h_, &batch), "teb_amd_download_tebs")) return 0;
  last_call_ok_ = true;

  // write back: bands (autoResize may have changed the pose count), cost_, optimized_
  int ok = 0;
  for (int b = 0; b < B; ++b)
  {
    TebOptimalPlannerAmd& p = *tebs[b];
    TimedElasticBand& t = p.teb_;
    const size_t o = (size_t)b * S;
    if (n[b] == t.sizePoses())
    {
      for (int i = 0; i < n[b]; ++i) { t.Pose(i).x() = x[o + i]; t.Pose(i).y() = y[o + i]; t.Pose(i).theta() = th[o + i]; }
      for (int i = 0; i < n[b] - 1; ++i) t.TimeDiff(i) = dt[o + i];
    }
    else
    {
      t.clearTimedElasticBand();
      t.addPose(x[o], y[o], th[o], true);                       // start pose is fixed
      for (int i = 1; i < n[b]; ++i) appendPoseAndTimeDiff(t, x[o + i], y[o + i], th[o + i], dt[o + i - 1]);
      t.setPoseVertexFixed(n[b] - 1, true);                     // goal pose is fixed
    }
    p.lm_iterations_ = iters[b]; p.lm_trials_ = trials[b];
    if (iters[b] > 0) { p.stats_available_ = stats[b] != 0; p.stats_back_chi2_ = back_chi2[b]; }   // optimize() ran: it replaced the statistics
    if (compute_cost_afterwards && status[b] == TEB_AMD_TEB_OK) p.cost_ = cost[b];   // computeCurrentCost ran in the last outer iteration
    if (status[b] == TEB_AMD_TEB_OK) { p.optimized_ = true; ++ok; }
  }
  return ok;
}

bool TebAmdBatch::uploadBands(const std::vector<TebOptimalPlannerAmd*>& tebs)
{
  const int B = (int)tebs.size(), S = max_poses_;
  if (!h_ || B == 0 || B > max_tebs_) return false;
  std::vector<int32_t> n(B), hvs(B), hvg(B), rot(B), via(B);
  std::vector<double> x((size_t)B * S, 0.0), y((size_t)B * S, 0.0), th((size_t)B * S, 0.0), dt((size_t)B * S, 0.0), vs(3 * (size_t)B), vg(3 * (size_t)B);
  for (int b = 0; b < B; ++b)
  {
    TebOptimalPlannerAmd& p = *tebs[b];
    const TimedElasticBand& t = p.teb_;
    if (t.sizePoses() > S) return check(TEB_AMD_ERR_CAPACITY, "uploadBands (band longer than max_poses)");
    n[b] = t.sizePoses();
    for (int i = 0; i < t.sizePoses(); ++i) { x[(size_t)b * S + i] = t.Pose(i).x(); y[(size_t)b * S + i] = t.Pose(i).y(); th[(size_t)b * S + i] = t.Pose(i).theta(); }
    for (int i = 0; i < t.sizeTimeDiffs(); ++i) dt[(size_t)b * S + i] = t.TimeDiff(i);
    hvs[b] = p.vel_start_.first; hvg[b] = p.vel_goal_.first;
    vs[3 * b] = p.vel_start_.second.linear.x; vs[3 * b + 1] = p.vel_start_.second.linear.y; vs[3 * b + 2] = p.vel_start_.second.angular.z;
    vg[3 * b] = p.vel_goal_.second.linear.x;  vg[3 * b + 1] = p.vel_goal_.second.linear.y;  vg[3 * b + 2] = p.vel_goal_.second.angular.z;
    rot[b] = p.prefer_rotdir_ == RotType::left ? TEB_AMD_ROT_LEFT : (p.prefer_rotdir_ == RotType::right ? TEB_AMD_ROT_RIGHT : TEB_AMD_ROT_NONE);
    via[b] = p.via_points_ != NULL;
  }
  teb_amd_teb_batch_t batch;
  std::memset(&batch, 0, sizeof(batch));
  batch.count = B; batch.stride = S; batch.n = n.data(); batch.x = x.data(); batch.y = y.data(); batch.theta = th.data(); batch.dt = dt.data();
  batch.has_vel_start = hvs.data(); batch.vel_start = vs.data(); batch.has_vel_goal = hvg.data(); batch.vel_goal = vg.data();
  batch.prefer_rotdir = rot.data(); batch.via_points_enabled = via.data();
  return check(teb_amd_upload_tebs(h_, &batch), "teb_amd_upload_tebs");
}

bool TebAmdBatch::downloadBands(const std::vector<TebOptimalPlannerAmd*>& tebs)
{
  const int B = (int)tebs.size(), S = max_poses_;
  std::vector<int32_t> n(B);
  std::vector<double> x((size_t)B * S), y((size_t)B * S), th((size_t)B * S), dt((size_t)B * S);
  teb_amd_teb_batch_t batch;
  std::memset(&batch, 0, sizeof(batch));
  batch.count = B; batch.stride = S; batch.n = n.data(); batch.x = x.data(); batch.y = y.data(); batch.theta = th.data(); batch.dt = dt.data();
  if (!check(teb_amd_download_tebs(h_, &batch), "teb_amd_download_tebs")) return false;
  for (int b = 0; b < B; ++b)
  {
    TimedElasticBand& t = tebs[b]->teb_;
    const size_t o = (size_t)b * S;
    t.clearTimedElasticBand();
    t.addPose(x[o], y[o], th[o], true);
    for (int i = 1; i < n[b]; ++i) appendPoseAndTimeDiff(t, x[o + i], y[o + i], th[o + i], dt[o + i - 1]);
    t.setPoseVertexFixed(n[b] - 1, true);
  }
  return true;
}

bool TebAmdBatch::updateAllTEBs(const std::vector<TebOptimalPlannerAmd*>& tebs, const PoseSE2* start, const PoseSE2* goal,
                                const geometry_msgs::Twist* start_velocity)
{
  if (tebs.empty() || !uploadBands(tebs)) return false;
  double s[3], g[3];
  if (start) { s[0] = start->x(); s[1] = start->y(); s[2] = start->theta(); }
  if (goal) { g[0] = goal->x(); g[1] = goal->y(); g[2] = goal->theta(); }
  if (!check(teb_amd_update_and_prune(h_, -1, start ? s : NULL, goal ? g : NULL, tebs.front()->cfg_->trajectory.min_samples),
             "teb_amd_update_and_prune")) return false;
  if (start_velocity)
  {
    const double v[3] = { start_velocity->linear.x, start_velocity->linear.y, start_velocity->angular.z };
    if (!check(teb_amd_set_velocity_start(h_, -1, 1, v), "teb_amd_set_velocity_start")) return false;
    for (TebOptimalPlannerAmd* p : tebs) p->setVelocityStart(*start_velocity);
  }
  return downloadBands(tebs);
}

bool TebAmdBatch::renewAndAnalyzeOldTebs(const TebConfig& cfg, int best_index, std::vector<bool>& keep, std::vector<double>* values, int* width)
{
  int32_t w = 0, count = 0;
  if (!check(teb_amd_get_pose_counts(h_, NULL, &count), "teb_amd_get_pose_counts") || count <= 0) return false;
  if (!check(teb_amd_compute_h_signatures(h_, cfg.hcp.h_signature_prescaler, NULL, &w), "teb_amd_compute_h_signatures")) return false;
  if (values)
  {
    values->assign((size_t)count * (w > 0 ? w : 1), 0.0);
    if (!check(teb_amd_compute_h_signatures(h_, cfg.hcp.h_signature_prescaler, values->data(), &w), "teb_amd_compute_h_signatures")) return false;
  }
  if (width) *width = w;
  std::vector<int32_t> k(count), valid(count), reasonable(count);
  if (!check(teb_amd_filter_equivalence_classes(h_, cfg.hcp.h_signature_threshold, best_index, cfg.hcp.max_number_plans_in_current_class,
                                                k.data(), valid.data(), reasonable.data()), "teb_amd_filter_equivalence_classes")) return false;
  keep.assign(count, false);
  for (int b = 0; b < count; ++b) keep[b] = k[b] != 0;
  return true;
}

void toAmdHcpParams(const TebConfig& cfg, teb_amd_hcp_params_t& p)
{
  teb_amd_hcp_params_default(&p);
  p.simple_exploration = cfg.hcp.simple_exploration;
  p.roadmap_graph_no_samples = cfg.hcp.roadmap_graph_no_samples;
  p.roadmap_graph_area_width = cfg.hcp.roadmap_graph_area_width;
  p.roadmap_graph_area_length_scale = cfg.hcp.roadmap_graph_area_length_scale;
  p.obstacle_heading_threshold = cfg.hcp.obstacle_heading_threshold;
  p.xy_goal_tolerance = cfg.goal_tolerance.xy_goal_tolerance;
  p.max_number_classes = cfg.hcp.max_number_classes;
  p.max_number_plans_in_current_class = cfg.hcp.max_number_plans_in_current_class;
  p.h_signature_prescaler = cfg.hcp.h_signature_prescaler;
  p.h_signature_threshold = cfg.hcp.h_signature_threshold;
  p.allow_init_with_backwards_motion = cfg.trajectory.allow_init_with_backwards_motion;
  p.delete_detours_backwards = cfg.hcp.delete_detours_backwards;
  p.detours_orientation_tolerance = cfg.hcp.detours_orientation_tolerance;
  p.length_start_orientation_vector = cfg.hcp.length_start_orientation_vector;
  p.max_ratio_detours_duration_best_duration = cfg.hcp.max_ratio_detours_duration_best_duration;
  p.global_plan_overwrite_orientation = cfg.trajectory.global_plan_overwrite_orientation;
  p.viapoints_all_candidates = cfg.hcp.viapoints_all_candidates;
}

bool TebAmdBatch::exploreEquivalenceClassesAndInitTebs(const TebConfig& cfg, ObstContainer* obstacles, const ViaPointContainer* via_points,
                                                       std::vector<TebOptimalPlannerAmdPtr>& tebs, int& best_index, const PoseSE2& start,
                                                       const PoseSE2& goal, double dist_to_obst, const geometry_msgs::Twist* start_vel,
                                                       bool free_goal_vel, const std::vector<geometry_msgs::PoseStamped>* initial_plan,
                                                       int* initial_plan_index, const std::function<bool()>* random_drop)
{
  if (!h_) return false;
  teb_amd_hcp_params_t hp;
  toAmdHcpParams(cfg, hp);
  // scene
  teb_amd_config_t a;
  if (!toAmdConfig(cfg, a)) return check(TEB_AMD_ERR_UNSUPPORTED, "toAmdConfig (robot footprint not representable)");
  AmdObstacleTable table;
  table.assign(obstacles);
  teb_amd_obstacles_t ov = table.view();
  if (!check(teb_amd_set_config(h_, &a), "teb_amd_set_config")) return false;
  if (!check(teb_amd_set_obstacles(h_, &ov), "teb_amd_set_obstacles")) return false;
  std::vector<double> viax, viay;
  if (via_points)
    for (const Eigen::Vector2d& v : *via_points) { viax.push_back(v.x()); viay.push_back(v.y()); }
  if (!check(teb_amd_set_via_points(h_, (int32_t)viax.size(), viax.data(), viay.data()), "teb_amd_set_via_points")) return false;
  // ---- renewAndAnalyzeOldTebs on the existing candidates
  int32_t n_kept = 0, new_best = -1;
  if (!tebs.empty())
  {
    std::vector<TebOptimalPlannerAmd*> raw;
    std::vector<int32_t> optimized;
    for (const TebOptimalPlannerAmdPtr& t : tebs) { raw.push_back(t.get()); optimized.push_back(t->isOptimized()); }
    if (!uploadBands(raw)) return false;
    if (!check(teb_amd_set_optimized_flags(h_, optimized.data()), "teb_amd_set_optimized_flags")) return false;
    int32_t w = 0;
    if (!check(teb_amd_compute_h_signatures(h_, hp.h_signature_prescaler, NULL, &w), "teb_amd_compute_h_signatures")) return false;
    std::vector<int32_t> keep(tebs.size());
    if (!check(teb_amd_filter_equivalence_classes(h_, hp.h_signature_threshold, best_index, hp.max_number_plans_in_current_class, keep.data(),
                                                  NULL, NULL), "teb_amd_filter_equivalence_classes")) return false;
    if (hp.delete_detours_backwards && !check(teb_amd_filter_detours(h_, &hp, best_index, keep.data()), "teb_amd_filter_detours")) return false;
    // randomlyDropTebs (:539-562): every surviving candidate but the best one is dropped when the caller's generator says so; visited in
    // the order the reference's vector has after renewAndAnalyzeOldTebs (best first, then the others)
    if (random_drop && *random_drop)
    {
      std::vector<int> visit(tebs.size());
      for (size_t b = 0; b < tebs.size(); ++b) visit[b] = (int)b;
      if (best_index >= 0 && best_index < (int)tebs.size()) std::swap(visit[0], visit[best_index]);
      for (int b : visit) if (keep[b] && b != best_index && (*random_drop)()) keep[b] = 0;
    }
    if (!check(teb_amd_compact_bands(h_, keep.data(), best_index, &n_kept, &new_best), "teb_amd_compact_bands")) return false;
    std::vector<int> order(tebs.size());
    for (size_t b = 0; b < tebs.size(); ++b) order[b] = (int)b;
    if (best_index >= 0 && best_index < (int)tebs.size()) std::swap(order[0], order[best_index]);
    std::vector<TebOptimalPlannerAmdPtr> kept;
    for (int b : order) if (keep[b]) kept.push_back(tebs[b]);
    tebs.swap(kept);
  }
  else
  {   // tebs_.clear() (new planner, clearPlanner(), goal jump in updateAllTEBs): whatever the device still holds goes as well
    int32_t count = 0;
    if (!check(teb_amd_get_pose_counts(h_, NULL, &count), "teb_amd_get_pose_counts")) return false;
    if (count > 0)
    {
      std::vector<int32_t> none(count, 0);
      if (!check(teb_amd_compact_bands(h_, none.data(), -1, &n_kept, &new_best), "teb_amd_compact_bands")) return false;
    }
  }
  best_index = new_best;
  // ---- createGraph + DepthFirst + addAndInitNewTeb
  const double s[3] = { start.x(), start.y(), start.theta() }, g[3] = { goal.x(), goal.y(), goal.theta() };
  double sv[3] = { 0, 0, 0 };
  if (start_vel) { sv[0] = start_vel->linear.x; sv[1] = start_vel->linear.y; sv[2] = start_vel->angular.z; }
  int32_t n_total = 0, ip_index = -1;
  std::vector<double> px, py, pyaw;
  if (initial_plan)
    for (const geometry_msgs::PoseStamped& ps : *initial_plan)
    { px.push_back(ps.pose.position.x); py.push_back(ps.pose.position.y); pyaw.push_back(tf::getYaw(ps.pose.orientation)); }
  if (!check(teb_amd_explore_candidates(h_, &hp, s, g, dist_to_obst, start_vel ? sv : NULL, free_goal_vel ? 1 : 0, best_index, NULL, 0,
                                        &n_total, NULL, NULL, (int32_t)px.size(), px.data(), py.data(), pyaw.data(), &ip_index),
             "teb_amd_explore_candidates")) return false;
  if (initial_plan_index) *initial_plan_index = ip_index;
  const size_t old = tebs.size();
  for (int b = (int)old; b < n_total; ++b)
  {
    TebOptimalPlannerAmdPtr c(new TebOptimalPlannerAmd(cfg, obstacles, TebVisualizationPtr(), via_points));
    if (start_vel) c->setVelocityStart(*start_vel);
    if (free_goal_vel) c->setVelocityGoalFree();
    tebs.push_back(c);
  }
  if (n_total > 0)   // updateReferenceTrajectoryViaPoints: the flags the device applied, onto the planner objects
  {
    std::vector<int32_t> ve(n_total);
    if (!check(teb_amd_get_band_flags(h_, ve.data(), NULL, NULL), "teb_amd_get_band_flags")) return false;
    for (int b = 0; b < n_total; ++b) tebs[b]->setViaPoints(ve[b] ? via_points : NULL);
  }
  if (n_total > (int)old)   // bands of the new candidates (the old ones are unchanged)
  {
    const int B = n_total, S = max_poses_;
    std::vector<int32_t> n(B);
    std::vector<double> x((size_t)B * S), y((size_t)B * S), th((size_t)B * S), dt((size_t)B * S);
    teb_amd_teb_batch_t batch;
    std::memset(&batch, 0, sizeof(batch));
    batch.count = B; batch.stride = S; batch.n = n.data(); batch.x = x.data(); batch.y = y.data(); batch.theta = th.data(); batch.dt = dt.data();
    if (!check(teb_amd_download_tebs(h_, &batch), "teb_amd_download_tebs")) return false;
    for (int b = (int)old; b < B; ++b)
    {
      TimedElasticBand& t = tebs[b]->teb_;
      const size_t o = (size_t)b * S;
      t.clearTimedElasticBand();
      t.addPose(x[o], y[o], th[o], true);
      for (int i = 1; i < n[b]; ++i) appendPoseAndTimeDiff(t, x[o + i], y[o + i], th[o + i], dt[o + i - 1]);
      t.setPoseVertexFixed(n[b] - 1, true);
    }
  }
  return true;
}

bool TebAmdBatch::signatures(const TebConfig& cfg, std::vector<double>& values, int& width)
{
  int32_t w = 0, count = 0;
  values.clear(); width = 0;
  if (!check(teb_amd_get_pose_counts(h_, NULL, &count), "teb_amd_get_pose_counts")) return false;
  if (count <= 0) return true;
  if (!check(teb_amd_compute_h_signatures(h_, cfg.hcp.h_signature_prescaler, NULL, &w), "teb_amd_compute_h_signatures")) return false;
  values.assign((size_t)count * (w > 0 ? w : 1), 0.0);
  if (!check(teb_amd_compute_h_signatures(h_, cfg.hcp.h_signature_prescaler, values.data(), &w), "teb_amd_compute_h_signatures")) return false;
  width = w;
  return true;
}

bool TebAmdBatch::getVelocityCommand(const TebConfig& cfg, int index, double& vx, double& vy, double& omega, int look_ahead_poses)
{
  int32_t ok = 0;
  vx = vy = omega = 0;
  if (!check(teb_amd_get_velocity_command(h_, index, look_ahead_poses, cfg.trajectory.prevent_look_ahead_poses_near_goal, &vx, &vy, &omega, &ok),
             "teb_amd_get_velocity_command")) return false;
  return ok != 0;
}

int TebAmdBatch::selectBestTeb(int last_best, int initial_plan, double* best_cost)
{
  int32_t best = -1;
  double bc = 0;
  if (!h_ || !check(teb_amd_select_best(h_, last_best, initial_plan, &best, &bc), "teb_amd_select_best")) return -1;
  if (best_cost) *best_cost = bc;
  return best;
}

int TebAmdBatch::selectBestTebDistributed(int last_best_global, int initial_plan_global, double* best_cost, int* owner_rank, bool* local_ok)
{
  int32_t best = -1, owner = -1;
  double bc = 0;
  if (local_ok) *local_ok = false;
  if (!comm_) { error_ = "selectBestTebDistributed: no communicator (setCommunicator)"; return -1; }
  // (a NULL handle still enters the exchange: the C call sends the unusable record for it)
  const bool ok = check(teb_amd_select_best_distributed(h_, comm_, global_offset_, last_best_global, initial_plan_global, &best, &bc, &owner),
                        "teb_amd_select_best_distributed");
  // On a local error the C call has still been through the all-gather and reports the PEERS' choice: a caller that can follow them
  // (local_ok given) gets it - it must enter broadcastBand with them -, any other caller gets -1 as before.
  if (!ok && !local_ok) return -1;
  if (local_ok) *local_ok = ok;
  if (best_cost) *best_cost = bc;
  if (owner_rank) *owner_rank = owner;
  return best;
}

// The same collective entered by a rank whose tick failed locally (exploration, signatures, upload, optimise): it contributes the
// unusable record - no peer can pick it, no peer waits for it - and learns the peers' choice so that it can follow them into
// broadcastBand. Returns the peers' best global index (-1: nobody has a candidate).
int TebAmdBatch::selectBestTebDistributedAsFailedRank(int* owner_rank)
{
  int32_t best = -1, owner = -1;
  double bc = 0;
  if (!comm_) { error_ = "selectBestTebDistributedAsFailedRank: no communicator (setCommunicator)"; return -1; }
  (void)teb_amd_select_best_distributed(NULL, comm_, global_offset_, -1, -1, &best, &bc, &owner);   // returns this rank's error by design
  if (owner_rank) *owner_rank = owner;
  return best;
}

void TebAmdBatch::adoptBroadcastStatistics(TebOptimalPlannerAmd& mirror) const
{
  int32_t available = 0;
  double back = 0;
  if (comm_ && teb_amd_comm_last_band_statistics(comm_, &available, &back) == TEB_AMD_OK)
  { mirror.stats_available_ = available != 0; mirror.stats_back_chi2_ = back; }
}

bool TebAmdBatch::broadcastBand(int owner_rank, int local_index, TimedElasticBand& teb)
{
  if (!h_ || !comm_) { error_ = "broadcastBand: no communicator (setCommunicator)"; return false; }
  const int S = max_poses_;
  int32_t n = 0;
  std::vector<double> x(S), y(S), th(S), dt(S);
  if (!check(teb_amd_broadcast_band(h_, comm_, owner_rank, local_index, S, &n, x.data(), y.data(), th.data(), dt.data()), "teb_amd_broadcast_band"))
    return false;
  teb.clearTimedElasticBand();
  if (n < 1) return true;
  teb.addPose(x[0], y[0], th[0], true);
  for (int i = 1; i < n; ++i) appendPoseAndTimeDiff(teb, x[i], y[i], th[i], dt[i - 1]);
  teb.setPoseVertexFixed(n - 1, true);
  return true;
}

} // namespace teb_local_planner
