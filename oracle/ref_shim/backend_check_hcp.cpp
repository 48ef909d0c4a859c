// backend_check_hcp.cpp — the candidate-generation half of the reference-side binding (TebAmdBatch::exploreEquivalenceClassesAndInitTebs,
// teb_local_planner_amd/host/teb_amd_backend.cpp) against the REFERENCE's HomotopyClassPlanner::exploreEquivalenceClassesAndInitTebs
// on identical TebConfig / ObstContainer / TimedElasticBand objects, in one process. TEST INFRASTRUCTURE, built into
// oracle/_ref/libteb_backend_check.so; used by tests/test_reference_backend.py.
#include <algorithm>
#include <complex>
#include <cstring>
#include <functional>
#include <memory>
#include <random>
#include <vector>
#include "shim_boost_graph.h"
#define private public
#define protected public
#include <teb_local_planner/homotopy_class_planner.h>
#include <teb_local_planner/graph_search.h>
#undef private
#undef protected
#include "ref_common.h"
#include "../../teb_local_planner_amd/host/teb_amd_hcp_backend.h"

using namespace teb_local_planner;
using namespace refshim;

namespace {
void put_band(TimedElasticBand& teb, int n, const double* x, const double* y, const double* th, const double* dt) {
  teb.addPose(x[0], y[0], th[0], true);
  for (int i = 1; i < n; ++i) teb.addPoseAndTimeDiff(x[i], y[i], th[i], dt[i - 1]);
  teb.setPoseVertexFixed(n - 1, true);
}
void get_band(const TimedElasticBand& teb, teb_amd_teb_batch_t* out, int slot) {
  const size_t o = (size_t)slot * out->stride;
  const int k = teb.sizePoses();
  out->n[slot] = k;
  for (int i = 0; i < k && i < out->stride; ++i) { out->x[o + i] = teb.Pose(i).x(); out->y[o + i] = teb.Pose(i).y(); out->theta[o + i] = teb.Pose(i).theta(); }
  for (int i = 0; i < teb.sizeTimeDiffs() && i < out->stride; ++i) out->dt[o + i] = teb.TimeDiff(i);
}
}  // namespace

extern "C" int backend_check_explore(const teb_amd_config_t* acfg, const teb_amd_hcp_params_t* p, const teb_amd_obstacles_t* o,
                                     const teb_amd_teb_batch_t* in, int best, const int32_t* optimized, const double* start,
                                     const double* goal, double dist_to_obst, const double* start_vel, int free_goal_vel,
                                     teb_amd_teb_batch_t* ref_out, int32_t* ref_n, teb_amd_teb_batch_t* amd_out, int32_t* amd_n,
                                     int32_t* amd_best, int32_t* amd_flags /* [slots*2]: vel_start_.first / vel_goal_.first */) {
  TebConfig cfg;
  to_ref_config(*acfg, cfg);
  cfg.hcp.simple_exploration = p->simple_exploration;
  cfg.hcp.roadmap_graph_no_samples = p->roadmap_graph_no_samples;
  cfg.hcp.roadmap_graph_area_width = p->roadmap_graph_area_width;
  cfg.hcp.roadmap_graph_area_length_scale = p->roadmap_graph_area_length_scale;
  cfg.hcp.obstacle_heading_threshold = p->obstacle_heading_threshold;
  cfg.goal_tolerance.xy_goal_tolerance = p->xy_goal_tolerance;
  cfg.hcp.max_number_classes = p->max_number_classes;
  cfg.hcp.max_number_plans_in_current_class = p->max_number_plans_in_current_class;
  cfg.hcp.h_signature_prescaler = p->h_signature_prescaler;
  cfg.hcp.h_signature_threshold = p->h_signature_threshold;
  cfg.trajectory.allow_init_with_backwards_motion = p->allow_init_with_backwards_motion;
  cfg.hcp.delete_detours_backwards = p->delete_detours_backwards;
  cfg.hcp.detours_orientation_tolerance = p->detours_orientation_tolerance;
  cfg.hcp.length_start_orientation_vector = p->length_start_orientation_vector;
  cfg.hcp.max_ratio_detours_duration_best_duration = p->max_ratio_detours_duration_best_duration;
  cfg.hcp.selection_dropping_probability = 0.0;
  cfg.hcp.enable_multithreading = false;
  ObstContainer obst;
  to_ref_obstacles(o, obst);
  PoseSE2 s(start[0], start[1], start[2]), g(goal[0], goal[1], goal[2]);
  geometry_msgs::Twist tw;
  if (start_vel) { tw.linear.x = start_vel[0]; tw.linear.y = start_vel[1]; tw.angular.z = start_vel[2]; }
  // (1) the reference
  {
    HomotopyClassPlanner hcp;
    hcp.initialize(cfg, &obst, TebVisualizationPtr(), NULL);
    for (int b = 0; in && b < in->count; ++b) {
      TebOptimalPlannerPtr t(new TebOptimalPlanner(cfg, &obst));
      const size_t so = (size_t)b * in->stride;
      put_band(t->teb(), in->n[b], in->x + so, in->y + so, in->theta + so, in->dt + so);
      t->optimized_ = optimized ? optimized[b] != 0 : true;
      hcp.tebs_.push_back(t);
    }
    if (in && best >= 0 && best < in->count) hcp.best_teb_ = hcp.tebs_[best];
    hcp.exploreEquivalenceClassesAndInitTebs(s, g, dist_to_obst, start_vel ? &tw : NULL, free_goal_vel != 0);
    *ref_n = (int)hcp.tebs_.size();
    for (int b = 0; b < *ref_n && b < ref_out->count; ++b) get_band(hcp.tebs_[b]->teb(), ref_out, b);
  }
  // (2) the binding on identical objects
  {
    std::vector<TebOptimalPlannerAmdPtr> tebs;
    for (int b = 0; in && b < in->count; ++b) {
      TebOptimalPlannerAmdPtr t(new TebOptimalPlannerAmd(cfg, &obst));
      const size_t so = (size_t)b * in->stride;
      put_band(t->teb(), in->n[b], in->x + so, in->y + so, in->theta + so, in->dt + so);
      t->optimized_ = optimized ? optimized[b] != 0 : true;
      tebs.push_back(t);
    }
    TebAmdBatch batch(cfg, std::max(amd_out->count, 1), amd_out->stride, std::max<int>((int)obst.size(), 1), o && o->vert_offset ? std::max(o->vert_offset[o->count], 1) : 1, 1);
    int bi = best;
    if (!batch.exploreEquivalenceClassesAndInitTebs(cfg, &obst, NULL, tebs, bi, s, g, dist_to_obst, start_vel ? &tw : NULL, free_goal_vel != 0)) return 2;
    *amd_n = (int)tebs.size();
    *amd_best = bi;
    for (int b = 0; b < *amd_n && b < amd_out->count; ++b) {
      get_band(tebs[b]->teb(), amd_out, b);
      amd_flags[2 * b] = tebs[b]->vel_start_.first; amd_flags[2 * b + 1] = tebs[b]->vel_goal_.first;
    }
  }
  return 0;
}

// n_ticks x plan() on ONE planner object, once with the reference's HomotopyClassPlanner (which = 0) and once with the drop-in class
// HomotopyClassPlannerAmd (which = 1): same TebConfig / ObstContainer / ViaPointContainer objects, same call sequence a ROS adapter
// makes (plan(initial_plan, ...) when plans are given). out: slots bands per tick, counts / best / initial-plan index per tick, costs.
extern "C" int backend_check_hcp_ticks(int which, const teb_amd_config_t* acfg, const teb_amd_hcp_params_t* p, const teb_amd_obstacles_t* o,
                                       int n_ticks, const double* starts, const double* goals, const double* start_vels, int free_goal_vel,
                                       int slots, teb_amd_teb_batch_t* out, int32_t* counts, int32_t* best, double* costs,
                                       const int32_t* plan_off, const double* plan_x, const double* plan_y, const double* plan_yaw, int n_via,
                                       const double* via_x, const double* via_y, int32_t* initial_plan_teb, double* cmd /* [n_ticks*4] */,
                                       int jacobian_mode) {
  setAmdJacobianMode(jacobian_mode);
  TebConfig cfg;
  to_ref_config(*acfg, cfg);
  cfg.hcp.simple_exploration = p->simple_exploration;
  cfg.hcp.roadmap_graph_no_samples = p->roadmap_graph_no_samples;
  cfg.hcp.roadmap_graph_area_width = p->roadmap_graph_area_width;
  cfg.hcp.roadmap_graph_area_length_scale = p->roadmap_graph_area_length_scale;
  cfg.hcp.obstacle_heading_threshold = p->obstacle_heading_threshold;
  cfg.goal_tolerance.xy_goal_tolerance = p->xy_goal_tolerance;
  cfg.hcp.max_number_classes = p->max_number_classes;
  cfg.hcp.max_number_plans_in_current_class = p->max_number_plans_in_current_class;
  cfg.hcp.h_signature_prescaler = p->h_signature_prescaler;
  cfg.hcp.h_signature_threshold = p->h_signature_threshold;
  cfg.trajectory.allow_init_with_backwards_motion = p->allow_init_with_backwards_motion;
  cfg.hcp.delete_detours_backwards = p->delete_detours_backwards;
  cfg.hcp.detours_orientation_tolerance = p->detours_orientation_tolerance;
  cfg.hcp.length_start_orientation_vector = p->length_start_orientation_vector;
  cfg.hcp.max_ratio_detours_duration_best_duration = p->max_ratio_detours_duration_best_duration;
  cfg.hcp.viapoints_all_candidates = p->viapoints_all_candidates;
  cfg.trajectory.global_plan_overwrite_orientation = p->global_plan_overwrite_orientation;
  cfg.hcp.selection_dropping_probability = 0.0;
  cfg.hcp.switching_blocking_period = 0.0;
  cfg.hcp.enable_multithreading = false;
  ObstContainer obst;
  to_ref_obstacles(o, obst);
  ViaPointContainer via;
  for (int k = 0; k < n_via; ++k) via.push_back(Eigen::Vector2d(via_x[k], via_y[k]));
  boost::shared_ptr<HomotopyClassPlanner> hcp;
  if (which == 0) { hcp.reset(new HomotopyClassPlanner()); hcp->initialize(cfg, &obst, TebVisualizationPtr(), n_via > 0 ? &via : NULL); }
  else hcp.reset(new HomotopyClassPlannerAmd(cfg, &obst, TebVisualizationPtr(), n_via > 0 ? &via : NULL, std::max(slots, 1), out->stride,
                                             std::max<int>((int)obst.size(), 1), o && o->vert_offset ? std::max(o->vert_offset[o->count], 1) : 1,
                                             std::max(n_via, 1)));
  // which = 2: the drop-in class in its SHARDED mode on a communicator of one rank (RCCL refuses two ranks on one device): every RCCL
  // call of the path - the selection all-gather, the status all-gather and the broadcast of the winner's band - runs, and the ticks have
  // to come out exactly as with which = 1
  teb_amd_comm_t* comm = NULL;
  if (which == 2) {
    char id[TEB_AMD_COMM_ID_BYTES];
    if (teb_amd_comm_unique_id(id) != TEB_AMD_OK || teb_amd_comm_create(id, 0, 1, 0, &comm) != TEB_AMD_OK) return 3;
    static_cast<HomotopyClassPlannerAmd*>(hcp.get())->setCommunicator(comm, 0);
  }
  struct CommGuard { teb_amd_comm_t*& c; ~CommGuard() { if (c) teb_amd_comm_destroy(c); } } guard{comm};
  std::vector<geometry_msgs::PoseStamped> plan;
  for (int t = 0; t < n_ticks; ++t) {
    PoseSE2 s(starts[3 * t], starts[3 * t + 1], starts[3 * t + 2]), g(goals[3 * t], goals[3 * t + 1], goals[3 * t + 2]);
    geometry_msgs::Twist tw;
    if (start_vels) { tw.linear.x = start_vels[3 * t]; tw.linear.y = start_vels[3 * t + 1]; tw.angular.z = start_vels[3 * t + 2]; }
    const int np = plan_off ? plan_off[t + 1] - plan_off[t] : 0;
    bool ok;
    if (np > 0) {
      plan.assign(np, geometry_msgs::PoseStamped());
      for (int k = 0; k < np; ++k) {
        const int q = plan_off[t] + k;
        plan[k].pose.position.x = plan_x[q]; plan[k].pose.position.y = plan_y[q];
        plan[k].pose.orientation = tf::createQuaternionMsgFromYaw(plan_yaw[q]);
      }
      ok = hcp->plan(plan, start_vels ? &tw : NULL, free_goal_vel != 0);
    } else {
      ok = hcp->plan(s, g, start_vels ? &tw : NULL, free_goal_vel != 0);
    }
    if (!ok) return 2;
    const int nt = (int)hcp->tebs_.size();
    counts[t] = nt;
    best[t] = hcp->bestTebIdx();
    TebOptimalPlannerPtr ip = hcp->getInitialPlanTEB();
    initial_plan_teb[t] = -1;
    for (int b = 0; b < nt; ++b) if (hcp->tebs_[b] == ip) initial_plan_teb[t] = b;
    for (int b = 0; b < nt && b < slots; ++b) {
      get_band(hcp->tebs_[b]->teb(), out, t * slots + b);
      costs[t * slots + b] = hcp->tebs_[b]->getCurrentCost();
    }
    double vx = 0, vy = 0, om = 0;
    cmd[4 * t] = hcp->getVelocityCommand(vx, vy, om, 1);   // inherited, reads best_teb_
    cmd[4 * t + 1] = vx; cmd[4 * t + 2] = vy; cmd[4 * t + 3] = om;
  }
  return 0;
}
