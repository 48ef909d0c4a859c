// ref_driver.cpp — C interface over the REFERENCE's own classes (compiled from /root/reference where they lie, against
// the shim headers in ./include). TEST INFRASTRUCTURE: used only by tests/test_reference_pinning.py to pin the oracle.
// What is reference code here: every computeError(), penalties.h, misc.h, pose_se2.h, distance_calculations.h,
// obstacles.h + src/obstacles.cpp, robot_footprint_model.h, TimedElasticBand::autoResize (src/timed_elastic_band.cpp),
// EdgeKinematicsDiffDrive::linearizeOplus / EdgeTimeOptimal::linearizeOplus, TebConfig's constructor defaults.
// + src/optimal_planner.cpp (buildGraph / AddEdges* / optimizeGraph / computeCurrentCost / optimizeTEB) driven through the
// recording SparseOptimizer stand-in of shim_g2o.h (the LM iteration itself is restated there, libg2o being absent).
// What is NOT (external, absent): libg2o, Eigen, ROS, Boost -> ./shim_*.h.
#include "ref_common.h"
#include "../grid_costmap.h"

#include <atomic>
#include <thread>

using namespace teb_local_planner;
using namespace refshim;

extern "C" {

// TebOptimalPlanner::optimizeTEB of the reference (src/optimal_planner.cpp:183-233) on one strip. Arrays have capacity cap.
// out: success flag, n, state, cost (getCurrentCost()).
int ref_optimize_teb(const teb_amd_config_t* acfg, const teb_amd_obstacles_t* o, int n_via, const double* via_x,
                     const double* via_y, int32_t* n_io, int cap, double* x, double* y, double* th, double* dt, int has_vs,
                     const double* vs, int has_vg, const double* vg, int rotdir, int via_enabled, int inner, int outer,
                     int compute_cost, double obst_cost_scale, double viapoint_cost_scale, int alternative_time_cost,
                     int32_t* success, double* cost) {
  TebConfig cfg;
  to_ref_config(*acfg, cfg);
  ObstContainer obst;
  to_ref_obstacles(o, obst);
  ViaPointContainer via;
  for (int i = 0; i < n_via; ++i) via.push_back(Eigen::Vector2d(via_x[i], via_y[i]));
  PlannerProbe pl(cfg, &obst, TebVisualizationPtr(), via_enabled ? &via : nullptr);
  fill_planner(pl, *n_io, x, y, th, dt, has_vs, vs, has_vg, vg, rotdir);
  const bool ok = pl.optimizeTEB(inner, outer, compute_cost != 0, obst_cost_scale, viapoint_cost_scale, alternative_time_cost != 0);
  *success = ok ? 1 : 0;
  *cost = pl.getCurrentCost();
  const TimedElasticBand& teb = pl.teb();
  const int n = teb.sizePoses();
  if (n > cap) return 4;
  *n_io = n;
  for (int i = 0; i < n; ++i) { x[i] = teb.Pose(i).x(); y[i] = teb.Pose(i).y(); th[i] = teb.Pose(i).theta(); }
  for (int i = 0; i < teb.sizeTimeDiffs(); ++i) dt[i] = teb.TimeDiff(i);
  return 0;
}

// The hyper-graph TebOptimalPlanner::buildGraph (src/optimal_planner.cpp:323-366) creates for one strip, in insertion order,
// errors computed at the given state. Per edge: irec[8] = {type, dim, n_vertices, v0, v1, v2, v3, v4} with vertex codes
// 2*i for pose i and 2*i+1 for time difference i (the ids AddTEBVertices assigns, :423-441); drec[48] = {err0..2, info0..2, J(3 x 14 row-major: up to 3 pose
// vertices x 3 columns, then up to 2 time-difference vertices, then 3 spare) = what e->linearizeOplus() leaves (the reference's
// two analytic Jacobians, else the central differences of shim_g2o.h over the reference's computeError)}.
int ref_build_graph(const teb_amd_config_t* acfg, const teb_amd_obstacles_t* o, int n_via, const double* via_x,
                    const double* via_y, int n, const double* x, const double* y, const double* th, const double* dt,
                    int has_vs, const double* vs, int has_vg, const double* vg, int rotdir, int via_enabled,
                    double weight_multiplier, int32_t* irec, double* drec, int cap, int32_t* count) {
  TebConfig cfg;
  to_ref_config(*acfg, cfg);
  ObstContainer obst;
  to_ref_obstacles(o, obst);
  ViaPointContainer via;
  for (int i = 0; i < n_via; ++i) via.push_back(Eigen::Vector2d(via_x[i], via_y[i]));
  PlannerProbe pl(cfg, &obst, TebVisualizationPtr(), via_enabled ? &via : nullptr);
  fill_planner(pl, n, x, y, th, dt, has_vs, vs, has_vg, vg, rotdir);
  if (!pl.buildGraph(weight_multiplier)) return 1;
  pl.optimizer()->initializeOptimization();
  pl.optimizer()->computeActiveErrors();
  int k = 0;
  for (g2o::OptimizableGraph::Edge* e : pl.optimizer()->activeEdges()) {
    if (k < cap) {
      int32_t* ir = irec + (size_t)k * 8;
      double* dr = drec + (size_t)k * 48;
      for (int q = 0; q < 48; ++q) dr[q] = 0;
      ir[0] = edge_type_code(e); ir[1] = e->dimension(); ir[2] = e->nVertices();
      for (int q = 0; q < 5; ++q) ir[3 + q] = q < e->nVertices() ? e->vertexAt(q)->id() : -1;
      for (int q = 0; q < 3; ++q) { dr[q] = q < e->dimension() ? e->errorAt(q) : 0.0; dr[3 + q] = q < e->dimension() ? e->infoAt(q, q) : 0.0; }
      e->linearizeOplus();
      int np = 0, nd = 0;
      for (int v = 0; v < e->nVertices(); ++v) {
        g2o::OptimizableGraph::Vertex* vv = e->vertexAt(v);
        const bool is_pose = vv->dimension() == 3;
        const int col0 = is_pose ? 3 * np : 9 + nd;
        if (!vv->fixed())
          for (int r = 0; r < e->dimension(); ++r)
            for (int cc = 0; cc < vv->dimension(); ++cc) dr[6 + r * 14 + col0 + cc] = e->jacAt(v, r, cc);
        if (is_pose) ++np; else ++nd;
      }
    }
    ++k;
  }
  *count = k;
  pl.clearGraph();
  return 0;
}


// TebConfig() constructor defaults of the reference (teb_config.h:245-390), flattened like teb_amd_config_default
int ref_config_default(teb_amd_config_t* a) {
  TebConfig c;
  std::memset(a, 0, sizeof(*a));
  a->teb_autosize = c.trajectory.teb_autosize; a->dt_ref = c.trajectory.dt_ref; a->dt_hysteresis = c.trajectory.dt_hysteresis;
  a->min_samples = c.trajectory.min_samples; a->max_samples = c.trajectory.max_samples;
  a->exact_arc_length = c.trajectory.exact_arc_length; a->via_points_ordered = c.trajectory.via_points_ordered;
  a->max_vel_x = c.robot.max_vel_x; a->max_vel_x_backwards = c.robot.max_vel_x_backwards; a->max_vel_y = c.robot.max_vel_y;
  a->max_vel_trans = c.robot.max_vel_trans; a->max_vel_theta = c.robot.max_vel_theta; a->acc_lim_x = c.robot.acc_lim_x;
  a->acc_lim_y = c.robot.acc_lim_y; a->acc_lim_theta = c.robot.acc_lim_theta; a->min_turning_radius = c.robot.min_turning_radius;
  a->min_obstacle_dist = c.obstacles.min_obstacle_dist; a->inflation_dist = c.obstacles.inflation_dist;
  a->dynamic_obstacle_inflation_dist = c.obstacles.dynamic_obstacle_inflation_dist;
  a->include_dynamic_obstacles = c.obstacles.include_dynamic_obstacles;
  a->obstacle_poses_affected = c.obstacles.obstacle_poses_affected;
  a->legacy_obstacle_association = c.obstacles.legacy_obstacle_association;
  a->obstacle_association_force_inclusion_factor = c.obstacles.obstacle_association_force_inclusion_factor;
  a->obstacle_association_cutoff_factor = c.obstacles.obstacle_association_cutoff_factor;
  a->obstacle_proximity_ratio_max_vel = c.obstacles.obstacle_proximity_ratio_max_vel;
  a->obstacle_proximity_lower_bound = c.obstacles.obstacle_proximity_lower_bound;
  a->obstacle_proximity_upper_bound = c.obstacles.obstacle_proximity_upper_bound;
  a->no_inner_iterations = c.optim.no_inner_iterations; a->no_outer_iterations = c.optim.no_outer_iterations;
  a->optimization_activate = c.optim.optimization_activate; a->penalty_epsilon = c.optim.penalty_epsilon;
  a->weight_max_vel_x = c.optim.weight_max_vel_x; a->weight_max_vel_y = c.optim.weight_max_vel_y;
  a->weight_max_vel_theta = c.optim.weight_max_vel_theta; a->weight_acc_lim_x = c.optim.weight_acc_lim_x;
  a->weight_acc_lim_y = c.optim.weight_acc_lim_y; a->weight_acc_lim_theta = c.optim.weight_acc_lim_theta;
  a->weight_kinematics_nh = c.optim.weight_kinematics_nh; a->weight_kinematics_forward_drive = c.optim.weight_kinematics_forward_drive;
  a->weight_kinematics_turning_radius = c.optim.weight_kinematics_turning_radius; a->weight_optimaltime = c.optim.weight_optimaltime;
  a->weight_shortest_path = c.optim.weight_shortest_path; a->weight_obstacle = c.optim.weight_obstacle;
  a->weight_inflation = c.optim.weight_inflation; a->weight_dynamic_obstacle = c.optim.weight_dynamic_obstacle;
  a->weight_dynamic_obstacle_inflation = c.optim.weight_dynamic_obstacle_inflation;
  a->weight_velocity_obstacle_ratio = c.optim.weight_velocity_obstacle_ratio; a->weight_viapoint = c.optim.weight_viapoint;
  a->weight_prefer_rotdir = c.optim.weight_prefer_rotdir; a->weight_adapt_factor = c.optim.weight_adapt_factor;
  a->obstacle_cost_exponent = c.optim.obstacle_cost_exponent;
  a->selection_cost_hysteresis = c.hcp.selection_cost_hysteresis; a->selection_prefer_initial_plan = c.hcp.selection_prefer_initial_plan;
  a->selection_obst_cost_scale = c.hcp.selection_obst_cost_scale; a->selection_viapoint_cost_scale = c.hcp.selection_viapoint_cost_scale;
  a->selection_alternative_time_cost = c.hcp.selection_alternative_time_cost;
  return 0;
}

// Evaluate E edges (records in the layout of teb_oracle_edges) with the reference edge classes on the given state.
// err_out [E*3]; jac_out [E*33] is filled only for the two edges with a live analytic Jacobian in the reference
// (EdgeKinematicsDiffDrive, EdgeTimeOptimal); jac_valid [E] says which.
int ref_eval_edges(const teb_amd_config_t* acfg, const teb_amd_obstacles_t* o, int n_via, const double* via_x,
                   const double* via_y, int n, const double* x, const double* y, const double* th, const double* dt,
                   const double* vel_start, const double* vel_goal, int E, const int32_t* irec, const double* drec,
                   double* err_out, double* jac_out, int32_t* jac_valid) {
  TebConfig cfg;
  to_ref_config(*acfg, cfg);
  ObstContainer obst;
  to_ref_obstacles(o, obst);
  std::vector<Eigen::Vector2d> via;
  for (int i = 0; i < n_via; ++i) via.push_back(Eigen::Vector2d(via_x[i], via_y[i]));
  std::vector<std::unique_ptr<VertexPose>> poses;
  std::vector<std::unique_ptr<VertexTimeDiff>> dts;
  for (int i = 0; i < n; ++i) poses.emplace_back(new VertexPose(x[i], y[i], th[i], i == 0 || i == n - 1));
  for (int i = 0; i < n - 1; ++i) dts.emplace_back(new VertexTimeDiff(dt[i]));
  geometry_msgs::Twist vs, vg;
  vs.linear.x = vel_start[0]; vs.linear.y = vel_start[1]; vs.angular.z = vel_start[2];
  vg.linear.x = vel_goal[0]; vg.linear.y = vel_goal[1]; vg.angular.z = vel_goal[2];
  for (int k = 0; k < E; ++k) {
    const int32_t* ir = irec + (size_t)k * 16;
    const double* dr = drec + (size_t)k * 56;
    double* eo = err_out + (size_t)k * 3;
    double* jo = jac_out + (size_t)k * 33;
    eo[0] = eo[1] = eo[2] = 0;
    for (int q = 0; q < 33; ++q) jo[q] = 0;
    jac_valid[k] = 0;
    const int type = ir[0], p0 = ir[2], p1 = ir[3], p2 = ir[4], d0 = ir[6], d1 = ir[7], ob = ir[9], vi = ir[10];
    switch (type) {
      case 0: { EdgeObstacle e; e.setVertex(0, poses[p0].get()); e.setParameters(cfg, obst[ob].get()); e.computeError(); eo[0] = e.error()[0]; break; }
      case 1: { EdgeInflatedObstacle e; e.setVertex(0, poses[p0].get()); e.setParameters(cfg, obst[ob].get()); e.computeError(); eo[0] = e.error()[0]; eo[1] = e.error()[1]; break; }
      case 2: { EdgeDynamicObstacle e(dr[6]); e.setVertex(0, poses[p0].get()); e.setParameters(cfg, obst[ob].get()); e.computeError(); eo[0] = e.error()[0]; eo[1] = e.error()[1]; break; }
      case 3: { EdgeViaPoint e; e.setVertex(0, poses[p0].get()); e.setParameters(cfg, &via[vi]); e.computeError(); eo[0] = e.error()[0]; break; }
      case 4: { EdgeVelocity e; e.setVertex(0, poses[p0].get()); e.setVertex(1, poses[p1].get()); e.setVertex(2, dts[d0].get()); e.setTebConfig(cfg); e.computeError(); eo[0] = e.error()[0]; eo[1] = e.error()[1]; break; }
      case 5: { EdgeVelocityHolonomic e; e.setVertex(0, poses[p0].get()); e.setVertex(1, poses[p1].get()); e.setVertex(2, dts[d0].get()); e.setTebConfig(cfg); e.computeError(); for (int q = 0; q < 3; ++q) eo[q] = e.error()[q]; break; }
      case 6: { EdgeAcceleration e; e.setVertex(0, poses[p0].get()); e.setVertex(1, poses[p1].get()); e.setVertex(2, poses[p2].get()); e.setVertex(3, dts[d0].get()); e.setVertex(4, dts[d1].get()); e.setTebConfig(cfg); e.computeError(); eo[0] = e.error()[0]; eo[1] = e.error()[1]; break; }
      case 7: { EdgeAccelerationStart e; e.setVertex(0, poses[p0].get()); e.setVertex(1, poses[p1].get()); e.setVertex(2, dts[d0].get()); e.setInitialVelocity(vs); e.setTebConfig(cfg); e.computeError(); eo[0] = e.error()[0]; eo[1] = e.error()[1]; break; }
      case 8: { EdgeAccelerationGoal e; e.setVertex(0, poses[p0].get()); e.setVertex(1, poses[p1].get()); e.setVertex(2, dts[d0].get()); e.setGoalVelocity(vg); e.setTebConfig(cfg); e.computeError(); eo[0] = e.error()[0]; eo[1] = e.error()[1]; break; }
      case 9: { EdgeAccelerationHolonomic e; e.setVertex(0, poses[p0].get()); e.setVertex(1, poses[p1].get()); e.setVertex(2, poses[p2].get()); e.setVertex(3, dts[d0].get()); e.setVertex(4, dts[d1].get()); e.setTebConfig(cfg); e.computeError(); for (int q = 0; q < 3; ++q) eo[q] = e.error()[q]; break; }
      case 10: { EdgeAccelerationHolonomicStart e; e.setVertex(0, poses[p0].get()); e.setVertex(1, poses[p1].get()); e.setVertex(2, dts[d0].get()); e.setInitialVelocity(vs); e.setTebConfig(cfg); e.computeError(); for (int q = 0; q < 3; ++q) eo[q] = e.error()[q]; break; }
      case 11: { EdgeAccelerationHolonomicGoal e; e.setVertex(0, poses[p0].get()); e.setVertex(1, poses[p1].get()); e.setVertex(2, dts[d0].get()); e.setGoalVelocity(vg); e.setTebConfig(cfg); e.computeError(); for (int q = 0; q < 3; ++q) eo[q] = e.error()[q]; break; }
      case 12: {
        EdgeTimeOptimal e; e.setVertex(0, dts[d0].get()); e.setTebConfig(cfg); e.computeError(); eo[0] = e.error()[0];
        e.linearizeOplus(); jo[9] = e.jacobianOplusXi()(0, 0); jac_valid[k] = 1;
        break;
      }
      case 13: { EdgeShortestPath e; e.setVertex(0, poses[p0].get()); e.setVertex(1, poses[p1].get()); e.setTebConfig(cfg); e.computeError(); eo[0] = e.error()[0]; break; }
      case 14: {
        EdgeKinematicsDiffDrive e; e.setVertex(0, poses[p0].get()); e.setVertex(1, poses[p1].get()); e.setTebConfig(cfg); e.computeError();
        eo[0] = e.error()[0]; eo[1] = e.error()[1];
        e.linearizeOplus();
        for (int r = 0; r < 2; ++r) for (int q = 0; q < 3; ++q) { jo[r * 11 + q] = e.jacobianOplusXi()(r, q); jo[r * 11 + 3 + q] = e.jacobianOplusXj()(r, q); }
        jac_valid[k] = 1;
        break;
      }
      case 15: { EdgeKinematicsCarlike e; e.setVertex(0, poses[p0].get()); e.setVertex(1, poses[p1].get()); e.setTebConfig(cfg); e.computeError(); eo[0] = e.error()[0]; eo[1] = e.error()[1]; break; }
      case 16: { EdgePreferRotDir e; e.setVertex(0, poses[p0].get()); e.setVertex(1, poses[p1].get()); e.setRotDir(dr[7]); e.computeError(); eo[0] = e.error()[0]; break; }
      case 17: { EdgeVelocityObstacleRatio e; e.setVertex(0, poses[p0].get()); e.setVertex(1, poses[p1].get()); e.setVertex(2, dts[d0].get()); e.setParameters(cfg, obst[ob].get()); e.computeError(); eo[0] = e.error()[0]; eo[1] = e.error()[1]; break; }
      default: return 1;
    }
  }
  return 0;
}

// robot_model->calculateDistance / estimateSpatioTemporalDistance and Obstacle::getCentroid of the reference
int ref_distance(const teb_amd_config_t* acfg, const teb_amd_obstacles_t* o, int nq, const int32_t* oi, const double* x,
                 const double* y, const double* th, const int32_t* st, const double* t, double* dist, double* cx, double* cy) {
  TebConfig cfg;
  to_ref_config(*acfg, cfg);
  ObstContainer obst;
  to_ref_obstacles(o, obst);
  for (int q = 0; q < nq; ++q) {
    PoseSE2 p(x[q], y[q], th[q]);
    const Obstacle* ob = obst[oi[q]].get();
    dist[q] = st[q] ? cfg.robot_model->estimateSpatioTemporalDistance(p, ob, t[q]) : cfg.robot_model->calculateDistance(p, ob);
    cx[q] = ob->getCentroid().x(); cy[q] = ob->getCentroid().y();
  }
  return 0;
}

// TimedElasticBand::autoResize of the reference (src/timed_elastic_band.cpp:227-286)
int ref_autoresize(double* x, double* y, double* th, double* dt, int32_t* n, int cap, double dt_ref, double dt_hyst,
                   int min_samples, int max_samples, int fast_mode) {
  TimedElasticBand teb;
  teb.addPose(PoseSE2(x[0], y[0], th[0]), true);
  for (int i = 1; i < *n; ++i) teb.addPoseAndTimeDiff(PoseSE2(x[i], y[i], th[i]), dt[i - 1]);
  teb.setPoseVertexFixed(*n - 1, true);
  teb.autoResize(dt_ref, dt_hyst, min_samples, max_samples, fast_mode != 0);
  if (teb.sizePoses() > cap) return 4;
  *n = teb.sizePoses();
  for (int i = 0; i < teb.sizePoses(); ++i) { x[i] = teb.Pose(i).x(); y[i] = teb.Pose(i).y(); th[i] = teb.Pose(i).theta(); }
  for (int i = 0; i < teb.sizeTimeDiffs(); ++i) dt[i] = teb.TimeDiff(i);
  return 0;
}

// penalties.h / misc.h scalar helpers
int ref_scalar_helpers(int nq, const double* v, const double* a, const double* b, const double* eps, double* out /* nq*8 */) {
  for (int q = 0; q < nq; ++q) {
    double* o = out + (size_t)q * 8;
    o[0] = penaltyBoundToInterval(v[q], a[q], eps[q]);
    o[1] = penaltyBoundToInterval(v[q], a[q], b[q], eps[q]);
    o[2] = penaltyBoundFromBelow(v[q], a[q], eps[q]);
    o[3] = penaltyBoundToIntervalDerivative(v[q], a[q], eps[q]);
    o[4] = penaltyBoundToIntervalDerivative(v[q], a[q], b[q], eps[q]);
    o[5] = penaltyBoundFromBelowDerivative(v[q], a[q], eps[q]);
    o[6] = fast_sigmoid(v[q]);
    o[7] = PoseSE2::average(PoseSE2(0, 0, v[q]), PoseSE2(0, 0, a[q])).theta();
  }
  return 0;
}

// ---- SURVEY section 8(f) rows f1 / f2, executed by the reference's own TimedElasticBand / TebOptimalPlanner ---------------------
namespace {
int band_out(const TimedElasticBand& teb, double* x, double* y, double* th, double* dt, int32_t* n, int cap) {
  const int k = teb.sizePoses();
  if (k > cap) return 4;
  *n = k;
  for (int i = 0; i < k; ++i) { x[i] = teb.Pose(i).x(); y[i] = teb.Pose(i).y(); th[i] = teb.Pose(i).theta(); }
  for (int i = 0; i < teb.sizeTimeDiffs(); ++i) dt[i] = teb.TimeDiff(i);
  return 0;
}
void band_in(TimedElasticBand& teb, int n, const double* x, const double* y, const double* th, const double* dt) {
  teb.addPose(x[0], y[0], th[0], true);
  for (int i = 1; i < n; ++i) teb.addPoseAndTimeDiff(x[i], y[i], th[i], dt[i - 1]);
  teb.setPoseVertexFixed(n - 1, true);
}
}  // namespace

// TimedElasticBand::initTrajectoryToGoal(start, goal, diststep, ...) src/timed_elastic_band.cpp:325-377
int ref_init_trajectory_line(const double* start, const double* goal, double diststep, double max_vel_x, int min_samples, int guess_backwards,
                             double* x, double* y, double* th, double* dt, int32_t* n, int cap) {
  TimedElasticBand teb;
  if (!teb.initTrajectoryToGoal(PoseSE2(start[0], start[1], start[2]), PoseSE2(goal[0], goal[1], goal[2]), diststep, max_vel_x, min_samples,
                                guess_backwards != 0)) return 1;
  return band_out(teb, x, y, th, dt, n, cap);
}

// initTrajectoryToGoal(plan (PoseStamped), ...) :380-452. yaw_seen[np] = tf::getYaw of the quaternions the plan was given as.
int ref_init_trajectory_plan(int np, const double* px, const double* py, const double* pyaw, double max_vel_x, double max_vel_theta,
                             int estimate_orient, int min_samples, int guess_backwards, double* yaw_seen, double* x, double* y, double* th,
                             double* dt, int32_t* n, int cap) {
  std::vector<geometry_msgs::PoseStamped> plan(np);
  for (int i = 0; i < np; ++i) {
    plan[i].pose.position.x = px[i]; plan[i].pose.position.y = py[i];
    plan[i].pose.orientation = tf::createQuaternionMsgFromYaw(pyaw[i]);
    yaw_seen[i] = tf::getYaw(plan[i].pose.orientation);
  }
  TimedElasticBand teb;
  if (!teb.initTrajectoryToGoal(plan, max_vel_x, max_vel_theta, estimate_orient != 0, min_samples, guess_backwards != 0)) return 1;
  return band_out(teb, x, y, th, dt, n, cap);
}

// template initTrajectoryToGoal(path_start, path_end, fun_position, ...) timed_elastic_band.hpp:46-183
int ref_init_trajectory_path(int np, const double* px, const double* py, double max_vel_x, double max_vel_theta, int has_max_acc_x,
                             double max_acc_x, int has_start_orient, double start_orient, int has_goal_orient, double goal_orient,
                             int min_samples, int guess_backwards, double* x, double* y, double* th, double* dt, int32_t* n, int cap) {
  std::vector<Eigen::Vector2d> path;
  for (int i = 0; i < np; ++i) path.push_back(Eigen::Vector2d(px[i], py[i]));
  TimedElasticBand teb;
  auto fun = [](const Eigen::Vector2d& p) -> const Eigen::Vector2d& { return p; };
  boost::optional<double> acc = has_max_acc_x ? boost::optional<double>(max_acc_x) : boost::optional<double>(boost::none);
  boost::optional<double> so = has_start_orient ? boost::optional<double>(start_orient) : boost::optional<double>(boost::none);
  boost::optional<double> go = has_goal_orient ? boost::optional<double>(goal_orient) : boost::optional<double>(boost::none);
  if (!teb.initTrajectoryToGoal(path.begin(), path.end(), fun, max_vel_x, max_vel_theta, acc, boost::optional<double>(boost::none), so, go,
                                min_samples, guess_backwards != 0)) return 1;
  return band_out(teb, x, y, th, dt, n, cap);
}

// TimedElasticBand::updateAndPruneTEB :555-597
int ref_update_and_prune(double* x, double* y, double* th, double* dt, int32_t* n, const double* new_start, const double* new_goal,
                         int min_samples) {
  TimedElasticBand teb;
  band_in(teb, *n, x, y, th, dt);
  PoseSE2 s, g;
  if (new_start) s = PoseSE2(new_start[0], new_start[1], new_start[2]);
  if (new_goal) g = PoseSE2(new_goal[0], new_goal[1], new_goal[2]);
  teb.updateAndPruneTEB(new_start ? boost::optional<const PoseSE2&>(s) : boost::optional<const PoseSE2&>(boost::none),
                        new_goal ? boost::optional<const PoseSE2&>(g) : boost::optional<const PoseSE2&>(boost::none), min_samples);
  return band_out(teb, x, y, th, dt, n, *n);
}

// getVelocityCommand / getVelocityProfile / getFullTrajectory of the reference, src/optimal_planner.cpp:1135-1247
int ref_consumers(const teb_amd_config_t* acfg, int n, const double* x, const double* y, const double* th, const double* dt, int has_vs,
                  const double* vs, int has_vg, const double* vg, int look_ahead_poses, int prevent_look_ahead_poses_near_goal,
                  double* cmd /*3*/, int32_t* cmd_ok, double* profile /*(n+1)*3*/, double* traj /*n*7*/) {
  TebConfig cfg;
  to_ref_config(*acfg, cfg);
  cfg.trajectory.prevent_look_ahead_poses_near_goal = prevent_look_ahead_poses_near_goal;
  PlannerProbe pl(cfg, nullptr, TebVisualizationPtr(), nullptr);
  fill_planner(pl, n, x, y, th, dt, has_vs, vs, has_vg, vg, TEB_AMD_ROT_NONE);
  if (!has_vg) {   // fill_planner called setVelocityGoalFree(); the stored twist stays zero
  }
  *cmd_ok = pl.getVelocityCommand(cmd[0], cmd[1], cmd[2], look_ahead_poses) ? 1 : 0;
  std::vector<geometry_msgs::Twist> prof;
  pl.getVelocityProfile(prof);
  for (size_t i = 0; i < prof.size(); ++i) { profile[3 * i] = prof[i].linear.x; profile[3 * i + 1] = prof[i].linear.y; profile[3 * i + 2] = prof[i].angular.z; }
  std::vector<TrajectoryPointMsg> tr;
  pl.getFullTrajectory(tr);
  for (size_t i = 0; i < tr.size(); ++i) {
    double* o = traj + 7 * i;
    o[0] = tr[i].pose.position.x; o[1] = tr[i].pose.position.y; o[2] = tf::getYaw(tr[i].pose.orientation);
    o[3] = tr[i].velocity.linear.x; o[4] = tr[i].velocity.linear.y; o[5] = tr[i].velocity.angular.z; o[6] = tr[i].time_from_start.toSec();
  }
  return 0;
}

// ---- row f4: the reference's own TebOptimalPlanner::isTrajectoryFeasible (src/optimal_planner.cpp:1250-1308) with a CostmapModel
// whose footprintCost is the grid restatement of oracle/grid_costmap.h (base_local_planner is absent from this image)
namespace {
struct GridCostmapModel : public base_local_planner::CostmapModel {
  gridcostmap::Grid g;
  int tests = 0, failed_at = -1;
  double footprintCost(double x, double y, double theta, const std::vector<geometry_msgs::Point>& spec, double, double) override {
    std::vector<double> fx, fy;
    for (const geometry_msgs::Point& p : spec) { fx.push_back(p.x); fy.push_back(p.y); }
    const double c = gridcostmap::footprint_cost(g, x, y, theta, (int)spec.size(), fx.data(), fy.data());
    if (c == -1 && failed_at < 0) failed_at = tests;
    ++tests;
    return c;
  }
};
}  // namespace
int ref_is_trajectory_feasible(int n, const double* x, const double* y, const double* th, const double* dt, const uint8_t* cells, int size_x,
                               int size_y, double resolution, double origin_x, double origin_y, int nf, const double* fx, const double* fy,
                               double inscribed_radius, double min_resolution_collision_check_angular, int look_ahead_idx,
                               double feasibility_check_lookahead_distance, int32_t* feasible, int32_t* first_infeasible) {
  TebConfig cfg;
  cfg.trajectory.min_resolution_collision_check_angular = min_resolution_collision_check_angular;
  PlannerProbe pl(cfg, nullptr, TebVisualizationPtr(), nullptr);
  const double zero[3] = {0, 0, 0};
  fill_planner(pl, n, x, y, th, dt, 1, zero, 1, zero, TEB_AMD_ROT_NONE);
  GridCostmapModel model;
  model.g = gridcostmap::Grid{cells, size_x, size_y, resolution, origin_x, origin_y};
  std::vector<geometry_msgs::Point> spec(nf);
  for (int i = 0; i < nf; ++i) { spec[i].x = fx[i]; spec[i].y = fy[i]; spec[i].z = 0; }
  *feasible = pl.isTrajectoryFeasible(&model, spec, inscribed_radius, 0.0, look_ahead_idx, feasibility_check_lookahead_distance) ? 1 : 0;
  if (first_infeasible) *first_infeasible = model.failed_at;
  return 0;
}

// ---- row f3, arithmetic core: the reference's HSignature / HSignature3d (h_signature.h) on a batch of bands ---------------------
// mode 2: HSignature (sig [B*2] = re, im); mode 3: HSignature3d (sig [B*M]). equal [B*B] = a.isEqual(b); valid, reasonable [B].
int ref_h_signatures(const teb_amd_config_t* acfg, const teb_amd_obstacles_t* o, const teb_amd_teb_batch_t* bt, int mode, double prescaler,
                     double threshold, double* sig, int32_t* equal, int32_t* valid, int32_t* reasonable) {
  TebConfig cfg;
  to_ref_config(*acfg, cfg);
  cfg.hcp.h_signature_prescaler = prescaler;
  cfg.hcp.h_signature_threshold = threshold;
  ObstContainer obst;
  to_ref_obstacles(o, obst);
  const int B = bt->count, S = bt->stride, M = (int)obst.size();
  auto fun = [](const VertexPose* pose) { return std::complex<long double>(pose->x(), pose->y()); };   // getCplxFromVertexPosePtr, homotopy_class_planner.h:72-75
  std::vector<std::unique_ptr<TimedElasticBand>> bands;
  std::vector<std::unique_ptr<EquivalenceClass>> cls;
  for (int b = 0; b < B; ++b) {
    bands.emplace_back(new TimedElasticBand());
    const size_t so = (size_t)b * S;
    band_in(*bands.back(), bt->n[b], bt->x + so, bt->y + so, bt->theta + so, bt->dt + so);
    TimedElasticBand& teb = *bands.back();
    if (mode == 2) {
      HSignature* H = new HSignature(cfg);
      H->calculateHSignature(teb.poses().begin(), teb.poses().end(), fun, &obst);
      sig[2 * b] = (double)H->value().real(); sig[2 * b + 1] = (double)H->value().imag();
      cls.emplace_back(H);
    } else {
      HSignature3d* H = new HSignature3d(cfg);
      H->calculateHSignature(teb.poses().begin(), teb.poses().end(), fun, &obst, teb.timediffs().begin(), teb.timediffs().end());
      for (int l = 0; l < M; ++l) sig[(size_t)b * M + l] = H->values()[l];
      cls.emplace_back(H);
    }
    valid[b] = cls.back()->isValid();
    reasonable[b] = cls.back()->isReasonable();
  }
  for (int a = 0; a < B; ++a)
    for (int b = 0; b < B; ++b) equal[a * B + b] = cls[a]->isEqual(*cls[b]);
  return 0;
}

// trace of the LM loop (the stand-in's, shim_g2o.h) of the following ref_optimize_batch calls: buf [B][cap_rows][4], rows [B]; NULL = off
static double* g_trace_buf = nullptr;
static int g_trace_cap = 0;
static int32_t* g_trace_rows = nullptr;
int ref_set_trace(double* buf, int cap_rows, int32_t* rows) {
  g_trace_buf = buf; g_trace_cap = cap_rows; g_trace_rows = rows;
  return 0;
}

// B x TebOptimalPlanner::optimizeTEB of the reference, one std::thread per band capped at `threads` - the reference's own
// optimizeAllTEBs uses one boost::thread per candidate (src/homotopy_class_planner.cpp:476-483). Used as the CPU baseline of bench.py.
int ref_optimize_batch(const teb_amd_config_t* acfg, const teb_amd_obstacles_t* o, int n_via, const double* via_x, const double* via_y,
                       teb_amd_teb_batch_t* bt, int inner, int outer, int compute_cost, double osc, double vsc, int atc, int threads,
                       int32_t* ok_out, double* cost_out, int32_t* lm_iterations) {
  TebConfig cfg;
  to_ref_config(*acfg, cfg);
  ObstContainer obst;
  to_ref_obstacles(o, obst);
  ViaPointContainer via;
  for (int i = 0; i < n_via; ++i) via.push_back(Eigen::Vector2d(via_x[i], via_y[i]));
  const int B = bt->count, S = bt->stride;
  { PlannerProbe warm(cfg, &obst, TebVisualizationPtr(), nullptr); }   // registerG2OTypes once, before the threads start
  std::atomic<int> next(0);
  std::atomic<int> err(0);
  auto work = [&]() {
    for (;;) {
      const int b = next.fetch_add(1);
      if (b >= B) break;
      const size_t so = (size_t)b * S;
      PlannerProbe pl(cfg, &obst, TebVisualizationPtr(), (!bt->via_points_enabled || bt->via_points_enabled[b]) ? &via : nullptr);
      fill_planner(pl, bt->n[b], bt->x + so, bt->y + so, bt->theta + so, bt->dt + so, bt->has_vel_start ? bt->has_vel_start[b] : 1,
                   bt->vel_start ? bt->vel_start + 3 * b : nullptr, bt->has_vel_goal ? bt->has_vel_goal[b] : 1,
                   bt->vel_goal ? bt->vel_goal + 3 * b : nullptr, bt->prefer_rotdir ? bt->prefer_rotdir[b] : TEB_AMD_ROT_NONE);
      g2o::SparseOptimizer::iterationCounter() = 0;
      g2o::SparseOptimizer::LmTrace& tr = g2o::SparseOptimizer::lmTrace();
      if (g_trace_buf && g_trace_rows) { tr.buf = g_trace_buf + (size_t)b * g_trace_cap * 4; tr.cap = g_trace_cap; tr.rows = g_trace_rows + b; g_trace_rows[b] = 0; }
      const bool ok = pl.optimizeTEB(inner, outer, compute_cost != 0, osc, vsc, atc != 0);
      tr = g2o::SparseOptimizer::LmTrace();
      if (ok_out) ok_out[b] = ok;
      if (cost_out) cost_out[b] = pl.getCurrentCost();
      if (lm_iterations) lm_iterations[b] = (int32_t)g2o::SparseOptimizer::iterationCounter();
      if (band_out(pl.teb(), bt->x + so, bt->y + so, bt->theta + so, bt->dt + so, bt->n + b, S)) err = 4;
    }
  };
  std::vector<std::thread> pool;
  const int T = std::max(1, std::min(threads, B));
  for (int t = 1; t < T; ++t) pool.emplace_back(work);
  work();
  for (std::thread& t : pool) t.join();
  return err.load();
}

}  // extern "C"
