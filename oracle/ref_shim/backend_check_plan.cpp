// backend_check_plan.cpp — the drop-in classes driven THROUGH THE POINTER THE PLUGIN HOLDS: PlannerInterfacePtr planner_
// (src/teb_local_planner_ros.cpp:120-125 creates a HomotopyClassPlanner or a TebOptimalPlanner behind it, :357 calls planner_->plan(),
// :374 planner_->hasDiverged(), :413 planner_->getVelocityCommand()). Every call below goes through that base-class pointer, never
// through the derived type: TebOptimalPlanner::optimizeTEB is not virtual (optimal_planner.h:231), so a subclass that merely
// re-declares it would run g2o on the CPU from the inherited plan(). The shim's SparseOptimizer counts the LM iterations the calling
// thread ran (iterationCounter): a drop-in object must leave that counter untouched.
// TEST INFRASTRUCTURE, built into oracle/_ref/libteb_backend_check.so; used by tests/test_reference_backend.py.
#include <algorithm>
#include <memory>
#include <vector>
#include "shim_boost_graph.h"
#include <teb_local_planner/homotopy_class_planner.h>
#include <teb_local_planner/planner_interface.h>
#include "ref_common.h"
#include "../../teb_local_planner_amd/host/teb_amd_hcp_backend.h"

using namespace teb_local_planner;
using namespace refshim;

namespace {
void hcp_params_onto(const teb_amd_hcp_params_t& p, TebConfig& cfg) {
  cfg.hcp.simple_exploration = p.simple_exploration;
  cfg.hcp.roadmap_graph_no_samples = p.roadmap_graph_no_samples;
  cfg.hcp.roadmap_graph_area_width = p.roadmap_graph_area_width;
  cfg.hcp.roadmap_graph_area_length_scale = p.roadmap_graph_area_length_scale;
  cfg.hcp.obstacle_heading_threshold = p.obstacle_heading_threshold;
  cfg.goal_tolerance.xy_goal_tolerance = p.xy_goal_tolerance;
  cfg.hcp.max_number_classes = p.max_number_classes;
  cfg.hcp.max_number_plans_in_current_class = p.max_number_plans_in_current_class;
  cfg.hcp.h_signature_prescaler = p.h_signature_prescaler;
  cfg.hcp.h_signature_threshold = p.h_signature_threshold;
  cfg.trajectory.allow_init_with_backwards_motion = p.allow_init_with_backwards_motion;
  cfg.hcp.delete_detours_backwards = p.delete_detours_backwards;
  cfg.hcp.detours_orientation_tolerance = p.detours_orientation_tolerance;
  cfg.hcp.length_start_orientation_vector = p.length_start_orientation_vector;
  cfg.hcp.max_ratio_detours_duration_best_duration = p.max_ratio_detours_duration_best_duration;
  cfg.hcp.viapoints_all_candidates = p.viapoints_all_candidates;
  cfg.trajectory.global_plan_overwrite_orientation = p.global_plan_overwrite_orientation;
  cfg.hcp.selection_dropping_probability = 0.0;
  cfg.hcp.switching_blocking_period = 0.0;
  cfg.hcp.enable_multithreading = false;
}
void band_out(const TimedElasticBand& teb, teb_amd_teb_batch_t* out, int slot) {
  const size_t o = (size_t)slot * out->stride;
  const int k = teb.sizePoses();
  out->n[slot] = k;
  for (int i = 0; i < k && i < out->stride; ++i) { out->x[o + i] = teb.Pose(i).x(); out->y[o + i] = teb.Pose(i).y(); out->theta[o + i] = teb.Pose(i).theta(); }
  for (int i = 0; i < teb.sizeTimeDiffs() && i < out->stride; ++i) out->dt[o + i] = teb.TimeDiff(i);
}
}  // namespace

// which: 0 = TebOptimalPlanner (reference, g2o stand-in on the CPU), 1 = TebOptimalPlannerAmd, 2 = HomotopyClassPlanner (reference),
//        3 = HomotopyClassPlannerAmd - all four held and driven as PlannerInterfacePtr.
// overload: 0 = plan(PoseSE2, PoseSE2, ..), 1 = plan(tf::Pose, tf::Pose, ..), 2 = plan(initial_plan, ..) with the plan of tick t in
//           plan_x/y/yaw[plan_off[t] .. plan_off[t+1]) (a two-pose plan start -> goal when the tick has none).
// Per tick: ok = plan()'s return value, diverged = hasDiverged(), g2o_iters = LM iterations the shim's SparseOptimizer ran inside plan()
// on this thread, lm_iters = lastLmIterations() of the planner that holds the plan (Amd classes; -1 for the reference), the band of the
// (best) planner in out slot t, cmd [4] = getVelocityCommand return value, vx, vy, omega.
// retune_chi2 >= 0: after the LAST tick divergence_detection_max_chi_squared of the live TebConfig is set to it and hasDiverged() is
// asked again (diverged[n_ticks]): the rule reads the configuration at the time of the question (src/optimal_planner.cpp:1026, 1038).
extern "C" int backend_check_plan_ticks(int which, const teb_amd_config_t* acfg, const teb_amd_hcp_params_t* hp, const teb_amd_obstacles_t* o,
                                        int n_via, const double* via_x, const double* via_y, int n_ticks, const double* starts,
                                        const double* goals, const double* start_vels, int free_goal_vel, int overload,
                                        const int32_t* plan_off, const double* plan_x, const double* plan_y, const double* plan_yaw,
                                        int jacobian_mode, teb_amd_teb_batch_t* out, int32_t* ok, int32_t* diverged, int32_t* g2o_iters,
                                        int32_t* lm_iters, double* cmd, double retune_chi2) {
  setAmdJacobianMode(jacobian_mode);
  TebConfig cfg;
  to_ref_config(*acfg, cfg);
  if (hp) hcp_params_onto(*hp, cfg);
  ObstContainer obst;
  to_ref_obstacles(o, obst);
  ViaPointContainer via;
  for (int k = 0; k < n_via; ++k) via.push_back(Eigen::Vector2d(via_x[k], via_y[k]));
  const ViaPointContainer* vp = n_via > 0 ? &via : NULL;
  const int M = std::max<int>((int)obst.size(), 1), V = o && o->vert_offset ? std::max(o->vert_offset[o->count], 1) : 1;
  PlannerInterfacePtr planner;   // the plugin's member type
  switch (which) {
    case 0: planner = PlannerInterfacePtr(new TebOptimalPlanner(cfg, &obst, TebVisualizationPtr(), vp)); break;
    case 1: planner = PlannerInterfacePtr(new TebOptimalPlannerAmd(cfg, &obst, TebVisualizationPtr(), vp)); break;
    case 2: planner = PlannerInterfacePtr(new HomotopyClassPlanner(cfg, &obst, TebVisualizationPtr(), vp)); break;
    case 3: planner = PlannerInterfacePtr(new HomotopyClassPlannerAmd(cfg, &obst, TebVisualizationPtr(), vp, 16, out->stride, M, V, std::max(n_via, 1))); break;
    default: return 1;
  }
  std::vector<geometry_msgs::PoseStamped> plan;
  for (int t = 0; t < n_ticks; ++t) {
    const double* s = starts + 3 * t;
    const double* g = goals + 3 * t;
    geometry_msgs::Twist tw;
    if (start_vels) { tw.linear.x = start_vels[3 * t]; tw.linear.y = start_vels[3 * t + 1]; tw.angular.z = start_vels[3 * t + 2]; }
    const geometry_msgs::Twist* sv = start_vels ? &tw : NULL;
    const long before = g2o::SparseOptimizer::iterationCounter();
    bool r;
    if (overload == 0) {
      r = planner->plan(PoseSE2(s[0], s[1], s[2]), PoseSE2(g[0], g[1], g[2]), sv, free_goal_vel != 0);
    } else if (overload == 1) {
      tf::Pose a, b;
      a.o.v[0] = s[0]; a.o.v[1] = s[1]; a.r.q = tf::createQuaternionMsgFromYaw(s[2]);
      b.o.v[0] = g[0]; b.o.v[1] = g[1]; b.r.q = tf::createQuaternionMsgFromYaw(g[2]);
      r = planner->plan(a, b, sv, free_goal_vel != 0);
    } else {
      const int np = plan_off ? plan_off[t + 1] - plan_off[t] : 0;
      plan.clear();
      if (np >= 2) {
        for (int k = 0; k < np; ++k) {
          const int q = plan_off[t] + k;
          geometry_msgs::PoseStamped ps;
          ps.pose.position.x = plan_x[q]; ps.pose.position.y = plan_y[q]; ps.pose.orientation = tf::createQuaternionMsgFromYaw(plan_yaw[q]);
          plan.push_back(ps);
        }
      } else {
        geometry_msgs::PoseStamped ps;
        ps.pose.position.x = s[0]; ps.pose.position.y = s[1]; ps.pose.orientation = tf::createQuaternionMsgFromYaw(s[2]);
        plan.push_back(ps);
        ps.pose.position.x = g[0]; ps.pose.position.y = g[1]; ps.pose.orientation = tf::createQuaternionMsgFromYaw(g[2]);
        plan.push_back(ps);
      }
      r = planner->plan(plan, sv, free_goal_vel != 0);
    }
    ok[t] = r;
    g2o_iters[t] = (int32_t)(g2o::SparseOptimizer::iterationCounter() - before);
    diverged[t] = planner->hasDiverged();
    double vx = 0, vy = 0, om = 0;
    cmd[4 * t] = planner->getVelocityCommand(vx, vy, om, 1);
    cmd[4 * t + 1] = vx; cmd[4 * t + 2] = vy; cmd[4 * t + 3] = om;
    // read-out only: which object holds the plan
    TebOptimalPlannerPtr holder;
    if (which <= 1) holder = boost::dynamic_pointer_cast<TebOptimalPlanner>(planner);
    else holder = boost::dynamic_pointer_cast<HomotopyClassPlanner>(planner)->bestTeb();
    lm_iters[t] = -1;
    out->n[t] = 0;
    if (holder) {
      if (holder->teb().sizePoses() > out->stride) return 4;
      band_out(holder->teb(), out, t);
      if (TebOptimalPlannerAmd* a = dynamic_cast<TebOptimalPlannerAmd*>(holder.get())) lm_iters[t] = a->lastLmIterations();
    }
  }
  if (retune_chi2 >= 0) {
    cfg.recovery.divergence_detection_max_chi_squared = retune_chi2;   // the planners hold a pointer to this object
    diverged[n_ticks] = planner->hasDiverged();
  }
  return 0;
}
