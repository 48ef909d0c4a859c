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

// in: bands 0..in->count-1 = tebs_ (before renewAndAnalyzeOldTebs), best = index of best_teb_ or -1.
// out: bands (capacity out->count slots of out->stride), *n_out = tebs_.size() afterwards; graph: vertices vx, vy [vcap], *nv,
// adjacency in insertion order as CSR (adj_off [nv+1], adj [acap]). skip_draws: engine draws discarded first (PRM only).
int ref_explore_candidates(const teb_amd_config_t* acfg, const teb_amd_hcp_params_t* p, const teb_amd_obstacles_t* o,
                           const teb_amd_teb_batch_t* in, int best, const double* start, const double* goal, double dist_to_obst,
                           const double* start_vel, int free_goal_vel, long skip_draws, teb_amd_teb_batch_t* out, int32_t* n_out,
                           int32_t* has_vs_out, double* vs_out, int32_t* has_vg_out, int vcap, double* vx, double* vy, int32_t* nv,
                           int acap, int32_t* adj_off, int32_t* adj) {
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
  cfg.hcp.delete_detours_backwards = false;       // deletePlansDetouringBackwards is not part of this row
  cfg.hcp.selection_dropping_probability = 0.0;   // randomlyDropTebs off (it draws from std::random_device)
  cfg.hcp.enable_multithreading = false;
  ObstContainer obst;
  to_ref_obstacles(o, obst);
  HomotopyClassPlanner hcp;
  hcp.initialize(cfg, &obst, TebVisualizationPtr(), NULL);
  for (int b = 0; in && b < in->count; ++b) {
    TebOptimalPlannerPtr t(new TebOptimalPlanner(cfg, &obst));
    const size_t so = (size_t)b * in->stride;
    band_in(t->teb(), in->n[b], in->x + so, in->y + so, in->theta + so, in->dt + so);
    hcp.tebs_.push_back(t);
  }
  if (in && best >= 0 && best < in->count) hcp.best_teb_ = hcp.tebs_[best];
  if (skip_draws > 0) {
    ProbRoadmapGraph* g = dynamic_cast<ProbRoadmapGraph*>(hcp.graph_search_.get());
    if (g) g->rnd_generator_.discard(skip_draws);
  }
  PoseSE2 s(start[0], start[1], start[2]), g(goal[0], goal[1], goal[2]);
  geometry_msgs::Twist tw;
  if (start_vel) { tw.linear.x = start_vel[0]; tw.linear.y = start_vel[1]; tw.angular.z = start_vel[2]; }
  hcp.exploreEquivalenceClassesAndInitTebs(s, g, dist_to_obst, start_vel ? &tw : NULL, free_goal_vel != 0);
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
}
