// ref_hcp_driver.cpp — C interface over the REFERENCE's own HomotopyClassPlanner / graph_search.cpp (compiled from /root/reference
// where they lie, against the stand-in headers in ./include). TEST INFRASTRUCTURE: used only by tests/test_reference_pinning.py to
// pin the oracle's candidate generation (row f3). Reference code here: src/graph_search.cpp (createGraph x2, DepthFirst),
// src/homotopy_class_planner.cpp (exploreEquivalenceClassesAndInitTebs, renewAndAnalyzeOldTebs, addEquivalenceClassIfNew, ...),
// homotopy_class_planner.hpp (addAndInitNewTeb, calculateEquivalenceClass), h_signature.h, timed_elastic_band.hpp, obstacles.h.
// NOT reference code (absent from the image): Boost.Graph / Boost.Random -> shim_boost_graph.h restates adjacency_list's iteration
// order, mt19937 and uniform_real_distribution.
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

using namespace teb_local_planner;
using namespace refshim;

extern "C" {
void band_in(TimedElasticBand& teb, int n, const double* x, const double* y, const double* th, const double* dt);
int band_out(const TimedElasticBand& teb, double* x, double* y, double* th, double* dt, int32_t* n, int cap);

// in: bands 0..in->count-1 = tebs_ (before renewAndAnalyzeOldTebs), best = index of best_teb_ or -1, optimized [count] = optimized_
// of every band (NULL = all true). stale_*: a band whose class is installed as best_teb_eq_class_ while best_teb_ is not in tebs_.
// out: bands (capacity out->count slots of out->stride), *n_out = tebs_.size() afterwards; graph: vertices vx, vy [vcap], *nv,
// adjacency in insertion order as CSR (adj_off [nv+1], adj [acap]). skip_draws: engine draws discarded first (PRM only).
int ref_explore_candidates(const teb_amd_config_t* acfg, const teb_amd_hcp_params_t* p, const teb_amd_obstacles_t* o,
                           const teb_amd_teb_batch_t* in, int best, const double* start, const double* goal, double dist_to_obst,
                           const double* start_vel, int free_goal_vel, long skip_draws, const int32_t* optimized, int stale_n,
                           const double* stale_x, const double* stale_y, const double* stale_th, const double* stale_dt, teb_amd_teb_batch_t* out, int32_t* n_out,
                           int32_t* has_vs_out, double* vs_out, int32_t* has_vg_out, int vcap, double* vx, double* vy, int32_t* nv,
                           int acap, int32_t* adj_off, int32_t* adj, int n_plan, const double* plan_x, const double* plan_y,
                           const double* plan_yaw, int stale_initial_n, const double* si_x, const double* si_y, const double* si_th,
                           const double* si_dt, int n_via, const double* via_x, const double* via_y, const int32_t* via_in,
                           int32_t* initial_plan_teb, int32_t* via_out, double* plan_yaw_seen) {
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
  cfg.hcp.selection_dropping_probability = 0.0;   // randomlyDropTebs off (it draws from std::random_device)
  cfg.hcp.enable_multithreading = false;
  ObstContainer obst;
  to_ref_obstacles(o, obst);
  HomotopyClassPlanner hcp;
  cfg.trajectory.global_plan_overwrite_orientation = p->global_plan_overwrite_orientation;
  cfg.hcp.viapoints_all_candidates = p->viapoints_all_candidates;
  ViaPointContainer via;
  for (int k = 0; k < n_via; ++k) via.push_back(Eigen::Vector2d(via_x[k], via_y[k]));
  hcp.initialize(cfg, &obst, TebVisualizationPtr(), n_via > 0 ? &via : NULL);
  for (int b = 0; in && b < in->count; ++b) {
    TebOptimalPlannerPtr t(new TebOptimalPlanner(cfg, &obst, TebVisualizationPtr(), (via_in && via_in[b]) ? &via : NULL));
    const size_t so = (size_t)b * in->stride;
    band_in(t->teb(), in->n[b], in->x + so, in->y + so, in->theta + so, in->dt + so);
    t->optimized_ = optimized ? optimized[b] != 0 : true;
    hcp.tebs_.push_back(t);
  }
  if (in && best >= 0 && best < in->count) hcp.best_teb_ = hcp.tebs_[best];
  TimedElasticBand stale;   // a band that was the best one in an earlier tick and is gone: only its class (best_teb_eq_class_) is left
  if (stale_n > 0) {
    band_in(stale, stale_n, stale_x, stale_y, stale_th, stale_dt);
    hcp.best_teb_eq_class_ = hcp.calculateEquivalenceClass(stale.poses().begin(), stale.poses().end(), getCplxFromVertexPosePtr, &obst,
                                                           stale.timediffs().begin(), stale.timediffs().end());
  }
  if (skip_draws > 0) {
    ProbRoadmapGraph* g = dynamic_cast<ProbRoadmapGraph*>(hcp.graph_search_.get());
    if (g) g->rnd_generator_.discard(skip_draws);
  }
  TimedElasticBand stale_initial;   // initial_plan_eq_class_ left over from an earlier tick's initial plan
  if (stale_initial_n > 0) {
    band_in(stale_initial, stale_initial_n, si_x, si_y, si_th, si_dt);
    hcp.initial_plan_eq_class_ = hcp.calculateEquivalenceClass(stale_initial.poses().begin(), stale_initial.poses().end(), getCplxFromVertexPosePtr,
                                                               &obst, stale_initial.timediffs().begin(), stale_initial.timediffs().end());
  }
  std::vector<geometry_msgs::PoseStamped> plan(n_plan > 0 ? n_plan : 0);
  for (int k = 0; k < n_plan; ++k) {
    plan[k].pose.position.x = plan_x[k]; plan[k].pose.position.y = plan_y[k];
    plan[k].pose.orientation = tf::createQuaternionMsgFromYaw(plan_yaw[k]);
    plan_yaw_seen[k] = tf::getYaw(plan[k].pose.orientation);   // what initTrajectoryToGoal(plan, ...) reads back from the message
  }
  if (n_plan > 0) hcp.initial_plan_ = &plan;
  PoseSE2 s(start[0], start[1], start[2]), g(goal[0], goal[1], goal[2]);
  geometry_msgs::Twist tw;
  if (start_vel) { tw.linear.x = start_vel[0]; tw.linear.y = start_vel[1]; tw.angular.z = start_vel[2]; }
  hcp.exploreEquivalenceClassesAndInitTebs(s, g, dist_to_obst, start_vel ? &tw : NULL, free_goal_vel != 0);
  hcp.updateReferenceTrajectoryViaPoints(cfg.hcp.viapoints_all_candidates);     // the next statement of plan(), :117
  {
    TebOptimalPlannerPtr ip = hcp.getInitialPlanTEB();                            // what selectBestTeb will see, :569
    *initial_plan_teb = -1;
    for (size_t b = 0; b < hcp.tebs_.size(); ++b) if (hcp.tebs_[b] == ip) *initial_plan_teb = (int)b;
    for (size_t b = 0; b < hcp.tebs_.size() && (int)b < out->count; ++b) via_out[b] = hcp.tebs_[b]->via_points_ != NULL;
  }
  const int nt = (int)hcp.tebs_.size();
  *n_out = nt;
  for (int b = 0; b < nt && b < out->count; ++b) {
    const size_t so = (size_t)b * out->stride;
    band_out(hcp.tebs_[b]->teb(), out->x + so, out->y + so, out->theta + so, out->dt + so, &out->n[b], out->stride);
    const TebOptimalPlanner& t = *hcp.tebs_[b];
    has_vs_out[b] = t.vel_start_.first; has_vg_out[b] = t.vel_goal_.first;
    vs_out[3 * b] = t.vel_start_.second.linear.x; vs_out[3 * b + 1] = t.vel_start_.second.linear.y; vs_out[3 * b + 2] = t.vel_start_.second.angular.z;
  }
  const HcGraph& G = hcp.graph_search_->graph_;
  const int N = (int)boost::num_vertices(G);
  *nv = N;
  int e = 0;
  for (int v = 0; v < N && v < vcap; ++v) {
    vx[v] = G[v].pos.x(); vy[v] = G[v].pos.y();
    adj_off[v] = e;
    for (auto w : G.out[v]) { if (e < acap) adj[e] = (int)w; ++e; }
  }
  if (N <= vcap) adj_off[N] = e;
  return (N > vcap || e > acap || nt > out->count) ? 1 : 0;
}

// n_ticks x HomotopyClassPlanner::plan(start_k, goal_k, start_vel_k, free_goal_vel) of the reference on ONE planner object
// (src/homotopy_class_planner.cpp:107-125): updateAllTEBs, exploreEquivalenceClassesAndInitTebs, optimizeAllTEBs (the reference's
// optimizeTEB through the recording g2o stand-in), selectBestTeb. starts / goals [n_ticks*3], start_vels [n_ticks*3] or NULL.
// out: slots bands per tick (band k of tick t at out slot t*slots+k), counts [n_ticks], best [n_ticks], costs [n_ticks*slots].
int ref_hcp_plan_ticks(const teb_amd_config_t* acfg, const teb_amd_hcp_params_t* p, const teb_amd_obstacles_t* o, int n_ticks,
                       const double* starts, const double* goals, const double* start_vels, int free_goal_vel, int slots,
                       teb_amd_teb_batch_t* out, int32_t* counts, int32_t* best, double* costs, const int32_t* plan_off,
                       const double* plan_x, const double* plan_y, const double* plan_yaw, double* plan_yaw_seen, int n_via,
                       const double* via_x, const double* via_y, int32_t* initial_plan_teb) {
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
  cfg.hcp.switching_blocking_period = 0.0;
  cfg.hcp.enable_multithreading = false;
  cfg.hcp.viapoints_all_candidates = p->viapoints_all_candidates;
  cfg.trajectory.global_plan_overwrite_orientation = p->global_plan_overwrite_orientation;
  ObstContainer obst;
  to_ref_obstacles(o, obst);
  ViaPointContainer via;
  for (int k = 0; k < n_via; ++k) via.push_back(Eigen::Vector2d(via_x[k], via_y[k]));
  HomotopyClassPlanner hcp;
  hcp.initialize(cfg, &obst, TebVisualizationPtr(), n_via > 0 ? &via : NULL);
  std::vector<geometry_msgs::PoseStamped> plan;
  for (int t = 0; t < n_ticks; ++t) {
    PoseSE2 s(starts[3 * t], starts[3 * t + 1], starts[3 * t + 2]), g(goals[3 * t], goals[3 * t + 1], goals[3 * t + 2]);
    geometry_msgs::Twist tw;
    if (start_vels) { tw.linear.x = start_vels[3 * t]; tw.linear.y = start_vels[3 * t + 1]; tw.angular.z = start_vels[3 * t + 2]; }
    const int np = plan_off ? plan_off[t + 1] - plan_off[t] : 0;
    if (np > 0) {   // plan(const std::vector<geometry_msgs::PoseStamped>& initial_plan, ...), :84-96
      plan.assign(np, geometry_msgs::PoseStamped());
      for (int k = 0; k < np; ++k) {
        const int q = plan_off[t] + k;
        plan[k].pose.position.x = plan_x[q]; plan[k].pose.position.y = plan_y[q];
        plan[k].pose.orientation = tf::createQuaternionMsgFromYaw(plan_yaw[q]);
        plan_yaw_seen[q] = tf::getYaw(plan[k].pose.orientation);
      }
      hcp.plan(plan, start_vels ? &tw : NULL, free_goal_vel != 0);
    } else {
      hcp.plan(s, g, start_vels ? &tw : NULL, free_goal_vel != 0);
    }
    {
      TebOptimalPlannerPtr ip = hcp.getInitialPlanTEB();
      initial_plan_teb[t] = -1;
      for (size_t b = 0; b < hcp.tebs_.size(); ++b) if (hcp.tebs_[b] == ip) initial_plan_teb[t] = (int)b;
    }
    const int nt = (int)hcp.tebs_.size();
    counts[t] = nt;
    best[t] = hcp.bestTebIdx();
    for (int b = 0; b < nt && b < slots; ++b) {
      const size_t slot = (size_t)t * slots + b, so = slot * out->stride;
      band_out(hcp.tebs_[b]->teb(), out->x + so, out->y + so, out->theta + so, out->dt + so, &out->n[slot], out->stride);
      costs[slot] = hcp.tebs_[b]->getCurrentCost();
    }
  }
  return 0;
}
}
