// backend_check.cpp — drives teb_local_planner_amd/host/teb_amd_backend.cpp (the reference-side binding of INTEGRATION.md)
// with the REFERENCE's own objects: the same TebConfig / ObstContainer / ViaPointContainer / TimedElasticBand instances are
// given (1) to the reference's TebOptimalPlanner::optimizeTEB (CPU, src/optimal_planner.cpp compiled in place, LM stand-in of
// shim_g2o.h) and (2) to TebOptimalPlannerAmd / TebAmdBatch (libteb_amd.so, MI355X). TEST INFRASTRUCTURE, built into
// oracle/_ref/libteb_backend_check.so; used by tests/test_reference_backend.py.
#include "ref_common.h"

#include "../../teb_local_planner_amd/host/teb_amd_backend.h"

using namespace teb_local_planner;
using namespace refshim;

namespace {
void band_in_common(TimedElasticBand& teb, int n, const double* x, const double* y, const double* th, const double* dt) {
  teb.addPose(x[0], y[0], th[0], true);
  for (int i = 1; i < n; ++i) teb.addPoseAndTimeDiff(x[i], y[i], th[i], dt[i - 1]);
  teb.setPoseVertexFixed(n - 1, true);
}
void read_band(const TimedElasticBand& teb, int S, double* x, double* y, double* th, double* dt, int32_t* n) {
  const int k = teb.sizePoses();
  *n = k;
  for (int i = 0; i < k && i < S; ++i) { x[i] = teb.Pose(i).x(); y[i] = teb.Pose(i).y(); th[i] = teb.Pose(i).theta(); }
  for (int i = 0; i < teb.sizeTimeDiffs() && i < S; ++i) dt[i] = teb.TimeDiff(i);
}
}  // namespace

extern "C" {

// mode: 0 = one TebAmdBatch::optimizeAllTEBs over all B candidates (+ selectBestTeb), 1 = B separate TebOptimalPlannerAmd::optimizeTEB
// calls. run_reference: also run the reference's CPU optimizeTEB on identical objects. Output arrays: [B*stride] / [B].
int backend_check_run(const teb_amd_config_t* acfg, const teb_amd_obstacles_t* o, int n_via, const double* via_x,
                      const double* via_y, const teb_amd_teb_batch_t* bt, int inner, int outer, int compute_cost, double osc,
                      double vsc, int atc, int jacobian_mode, int mode, int run_reference, int last_best, int initial_plan,
                      double* ref_x, double* ref_y, double* ref_th, double* ref_dt, int32_t* ref_n, double* ref_cost, int32_t* ref_ok,
                      double* amd_x, double* amd_y, double* amd_th, double* amd_dt, int32_t* amd_n, double* amd_cost, int32_t* amd_ok,
                      int32_t* amd_best, double* amd_best_cost, int32_t* roundtrip_ok) {
  TebConfig cfg;
  to_ref_config(*acfg, cfg);
  ObstContainer obst;
  to_ref_obstacles(o, obst);
  ViaPointContainer via;
  for (int i = 0; i < n_via; ++i) via.push_back(Eigen::Vector2d(via_x[i], via_y[i]));
  const int B = bt->count, S = bt->stride;
  setAmdJacobianMode(jacobian_mode);

  // adapters must give back what the reference objects were made from: TebConfig -> teb_amd_config_t, ObstContainer -> table
  {
    teb_amd_config_t back;
    toAmdConfig(cfg, back);
    back.jacobian_mode = acfg->jacobian_mode;
    int ok = 1;
#define SAME(f) if (back.f != acfg->f) ok = 0
    SAME(teb_autosize); SAME(dt_ref); SAME(dt_hysteresis); SAME(min_samples); SAME(max_samples); SAME(exact_arc_length);
    SAME(via_points_ordered); SAME(max_vel_x); SAME(max_vel_x_backwards); SAME(max_vel_y); SAME(max_vel_trans); SAME(max_vel_theta);
    SAME(acc_lim_x); SAME(acc_lim_y); SAME(acc_lim_theta); SAME(min_turning_radius); SAME(min_obstacle_dist); SAME(inflation_dist);
    SAME(dynamic_obstacle_inflation_dist); SAME(include_dynamic_obstacles); SAME(obstacle_poses_affected);
    SAME(legacy_obstacle_association); SAME(obstacle_association_force_inclusion_factor); SAME(obstacle_association_cutoff_factor);
    SAME(obstacle_proximity_ratio_max_vel); SAME(obstacle_proximity_lower_bound); SAME(obstacle_proximity_upper_bound);
    SAME(no_inner_iterations); SAME(no_outer_iterations); SAME(optimization_activate); SAME(penalty_epsilon);
    SAME(weight_max_vel_x); SAME(weight_max_vel_y); SAME(weight_max_vel_theta); SAME(weight_acc_lim_x); SAME(weight_acc_lim_y);
    SAME(weight_acc_lim_theta); SAME(weight_kinematics_nh); SAME(weight_kinematics_forward_drive);
    SAME(weight_kinematics_turning_radius); SAME(weight_optimaltime); SAME(weight_shortest_path); SAME(weight_obstacle);
    SAME(weight_inflation); SAME(weight_dynamic_obstacle); SAME(weight_dynamic_obstacle_inflation);
    SAME(weight_velocity_obstacle_ratio); SAME(weight_viapoint); SAME(weight_prefer_rotdir); SAME(weight_adapt_factor);
    SAME(obstacle_cost_exponent); SAME(selection_cost_hysteresis); SAME(selection_prefer_initial_plan);
    SAME(selection_obst_cost_scale); SAME(selection_viapoint_cost_scale); SAME(selection_alternative_time_cost);
    SAME(divergence_detection_enable); SAME(divergence_detection_max_chi_squared); SAME(footprint_type); SAME(footprint_radius);
    SAME(footprint_front_offset); SAME(footprint_front_radius); SAME(footprint_rear_offset); SAME(footprint_rear_radius);
#undef SAME
    if (acfg->footprint_type == TEB_AMD_FOOTPRINT_LINE || acfg->footprint_type == TEB_AMD_FOOTPRINT_POLYGON) {
      if (back.footprint_n_vertices != acfg->footprint_n_vertices) ok = 0;
      for (int i = 0; i < acfg->footprint_n_vertices && ok; ++i)
        if (back.footprint_vx[i] != acfg->footprint_vx[i] || back.footprint_vy[i] != acfg->footprint_vy[i]) ok = 0;
    }
    AmdObstacleTable tb;
    tb.assign(&obst);
    if ((int)tb.type.size() != o->count) ok = 0;
    for (int i = 0; i < o->count && ok; ++i) {
      if (tb.type[i] != o->type[i] || tb.ax[i] != o->ax[i] || tb.ay[i] != o->ay[i]) ok = 0;
      if ((o->type[i] == TEB_AMD_OBST_LINE || o->type[i] == TEB_AMD_OBST_PILL) && (tb.bx[i] != o->bx[i] || tb.by[i] != o->by[i])) ok = 0;
      if ((o->type[i] == TEB_AMD_OBST_CIRCULAR || o->type[i] == TEB_AMD_OBST_PILL) && tb.radius[i] != o->radius[i]) ok = 0;
      if (tb.dynamic[i] != (o->dynamic && o->dynamic[i] ? 1 : 0)) ok = 0;
      if (tb.dynamic[i] && (tb.vx[i] != o->vx[i] || tb.vy[i] != o->vy[i])) ok = 0;
      if (o->type[i] == TEB_AMD_OBST_POLYGON) {
        const int k0 = o->vert_offset[i], k1 = o->vert_offset[i + 1];
        if (tb.vert_offset[i + 1] - tb.vert_offset[i] != k1 - k0) ok = 0;
        for (int k = 0; k < k1 - k0 && ok; ++k)
          if (tb.vert_x[tb.vert_offset[i] + k] != o->vert_x[k0 + k] || tb.vert_y[tb.vert_offset[i] + k] != o->vert_y[k0 + k]) ok = 0;
      }
    }
    *roundtrip_ok = ok;
  }

  auto fill = [&](TebOptimalPlanner& pl, int b) {
    const size_t so = (size_t)b * S;
    fill_planner(pl, bt->n[b], bt->x + so, bt->y + so, bt->theta + so, bt->dt + so, bt->has_vel_start ? bt->has_vel_start[b] : 1,
                 bt->vel_start ? bt->vel_start + 3 * b : nullptr, bt->has_vel_goal ? bt->has_vel_goal[b] : 1,
                 bt->vel_goal ? bt->vel_goal + 3 * b : nullptr, bt->prefer_rotdir ? bt->prefer_rotdir[b] : TEB_AMD_ROT_NONE);
  };
  auto via_of = [&](int b) -> const ViaPointContainer* { return (!bt->via_points_enabled || bt->via_points_enabled[b]) ? &via : nullptr; };

  if (run_reference) {
    for (int b = 0; b < B; ++b) {
      TebOptimalPlanner pl(cfg, &obst, TebVisualizationPtr(), via_of(b));
      fill(pl, b);
      const bool ok = pl.optimizeTEB(inner, outer, compute_cost != 0, osc, vsc, atc != 0);
      ref_ok[b] = ok; ref_cost[b] = pl.getCurrentCost();
      if (pl.teb().sizePoses() > S) return 4;
      const size_t so = (size_t)b * S;
      read_band(pl.teb(), S, ref_x + so, ref_y + so, ref_th + so, ref_dt + so, ref_n + b);
    }
  }

  std::vector<std::unique_ptr<TebOptimalPlannerAmd>> own;
  std::vector<TebOptimalPlannerAmd*> tebs;
  for (int b = 0; b < B; ++b) {
    own.emplace_back(new TebOptimalPlannerAmd(cfg, &obst, TebVisualizationPtr(), via_of(b)));
    fill(*own.back(), b);
    tebs.push_back(own.back().get());
  }
  *amd_best = -1; *amd_best_cost = 0;
  if (mode == 0) {
    TebAmdBatch batch(cfg, B, S, o->count > 0 ? o->count : 1, o->vert_offset ? std::max(1, (int)o->vert_offset[o->count]) : 1,
                      n_via > 0 ? n_via : 1);
    batch.optimizeAllTEBs(tebs, inner, outer, compute_cost != 0, osc, vsc, atc != 0);
    if (!batch.lastError().empty()) return 3;
    *amd_best = batch.selectBestTeb(last_best, initial_plan, amd_best_cost);
    for (int b = 0; b < B; ++b) amd_ok[b] = tebs[b]->isOptimized();
  } else {
    for (int b = 0; b < B; ++b) amd_ok[b] = tebs[b]->optimizeTEB(inner, outer, compute_cost != 0, osc, vsc, atc != 0);
  }
  for (int b = 0; b < B; ++b) {
    amd_cost[b] = tebs[b]->getCurrentCost();
    if (tebs[b]->teb().sizePoses() > S) return 4;
    const size_t so = (size_t)b * S;
    read_band(tebs[b]->teb(), S, amd_x + so, amd_y + so, amd_th + so, amd_dt + so, amd_n + b);
  }
  return 0;
}

// The rows either side of the path through the backend, against the reference's own member functions on the same objects:
//   updateAllTEBs  vs TimedElasticBand::updateAndPruneTEB                    -> prune_* arrays (band after pruning, both sides)
//   renewAndAnalyzeOldTebs vs HSignature3d / HSignature + isEqual / isValid  -> sig_*, keep_*
//   getVelocityCommand (device band) vs TebOptimalPlanner::getVelocityCommand on the written-back band -> cmd_*
// Sequence: updateAllTEBs(start, goal, start_vel) -> optimizeAllTEBs -> renewAndAnalyzeOldTebs(best = -1) -> getVelocityCommand.
int backend_check_rows(const teb_amd_config_t* acfg, const teb_amd_obstacles_t* o, const teb_amd_teb_batch_t* bt, const double* start,
                       const double* goal, const double* start_vel, int inner, int outer, int look_ahead,
                       double* prune_ref /*[B*S*4]*/, double* prune_amd, int32_t* prune_n_ref, int32_t* prune_n_amd,
                       double* sig_ref /*[B*W]*/, double* sig_amd, int32_t* keep_ref, int32_t* keep_amd, int32_t* width,
                       double* cmd_ref /*[B*4]*/, double* cmd_amd) {
  TebConfig cfg;
  to_ref_config(*acfg, cfg);
  ObstContainer obst;
  to_ref_obstacles(o, obst);
  const int B = bt->count, S = bt->stride, M = (int)obst.size();
  setAmdJacobianMode(TEB_AMD_JACOBIAN_ANALYTIC);
  PoseSE2 ps(start[0], start[1], start[2]), pg(goal[0], goal[1], goal[2]);
  geometry_msgs::Twist sv; sv.linear.x = start_vel[0]; sv.linear.y = start_vel[1]; sv.angular.z = start_vel[2];
  std::vector<std::unique_ptr<TebOptimalPlannerAmd>> own;
  std::vector<TebOptimalPlannerAmd*> tebs;
  for (int b = 0; b < B; ++b) {
    own.emplace_back(new TebOptimalPlannerAmd(cfg, &obst, TebVisualizationPtr(), nullptr));
    const size_t so = (size_t)b * S;
    const double zero3[3] = {0, 0, 0};
    fill_planner(*own.back(), bt->n[b], bt->x + so, bt->y + so, bt->theta + so, bt->dt + so, 1, zero3, 1, zero3, TEB_AMD_ROT_NONE);
    tebs.push_back(own.back().get());
    // reference side of the pruning: the same band, the reference's own updateAndPruneTEB
    TimedElasticBand rt;
    band_in_common(rt, bt->n[b], bt->x + so, bt->y + so, bt->theta + so, bt->dt + so);
    rt.updateAndPruneTEB(boost::optional<const PoseSE2&>(ps), boost::optional<const PoseSE2&>(pg), cfg.trajectory.min_samples);
    read_band(rt, S, prune_ref + 4 * so, prune_ref + 4 * so + S, prune_ref + 4 * so + 2 * S, prune_ref + 4 * so + 3 * S, prune_n_ref + b);
  }
  TebAmdBatch batch(cfg, B, S, M > 0 ? M : 1, o->vert_offset ? std::max(1, (int)o->vert_offset[o->count]) : 1, 1);
  if (!batch.updateAllTEBs(tebs, &ps, &pg, &sv)) return 3;
  for (int b = 0; b < B; ++b) {
    const size_t so = (size_t)b * S;
    read_band(tebs[b]->teb(), S, prune_amd + 4 * so, prune_amd + 4 * so + S, prune_amd + 4 * so + 2 * S, prune_amd + 4 * so + 3 * S, prune_n_amd + b);
  }
  batch.optimizeAllTEBs(tebs, inner, outer, true, cfg.hcp.selection_obst_cost_scale, cfg.hcp.selection_viapoint_cost_scale,
                        cfg.hcp.selection_alternative_time_cost);
  if (!batch.lastError().empty()) return 3;
  // equivalence classes: backend (device) vs the reference's classes on the planners' bands (which now hold the optimised bands)
  std::vector<bool> keep;
  std::vector<double> values;
  int W = 0;
  if (!batch.renewAndAnalyzeOldTebs(cfg, -1, keep, &values, &W)) return 3;
  *width = W;
  auto fun = [](const VertexPose* pose) { return std::complex<long double>(pose->x(), pose->y()); };
  std::vector<std::unique_ptr<EquivalenceClass>> cls;
  std::vector<int> classes;
  for (int b = 0; b < B; ++b) {
    TimedElasticBand& teb = tebs[b]->teb();
    if (cfg.obstacles.include_dynamic_obstacles) {
      HSignature3d* H = new HSignature3d(cfg);
      H->calculateHSignature(teb.poses().begin(), teb.poses().end(), fun, &obst, teb.timediffs().begin(), teb.timediffs().end());
      for (int l = 0; l < M; ++l) sig_ref[(size_t)b * W + l] = H->values()[l];
      cls.emplace_back(H);
    } else {
      HSignature* H = new HSignature(cfg);
      H->calculateHSignature(teb.poses().begin(), teb.poses().end(), fun, &obst);
      sig_ref[2 * b] = (double)H->value().real(); sig_ref[2 * b + 1] = (double)H->value().imag();
      cls.emplace_back(H);
    }
    for (int l = 0; l < W; ++l) sig_amd[(size_t)b * W + l] = values[(size_t)b * W + l];
    bool is_new = cls.back()->isValid();
    for (int cidx : classes) if (is_new && cls.back()->isEqual(*cls[cidx])) is_new = false;   // hasEquivalenceClass
    if (is_new) classes.push_back(b);
    keep_ref[b] = is_new; keep_amd[b] = keep[b];
    // velocity command: device-resident band vs the reference's getVelocityCommand on the planner object
    double vx, vy, om;
    const bool okr = tebs[b]->TebOptimalPlanner::getVelocityCommand(vx, vy, om, look_ahead);
    cmd_ref[4 * b] = vx; cmd_ref[4 * b + 1] = vy; cmd_ref[4 * b + 2] = om; cmd_ref[4 * b + 3] = okr;
    const bool oka = batch.getVelocityCommand(cfg, b, vx, vy, om, look_ahead);
    cmd_amd[4 * b] = vx; cmd_amd[4 * b + 1] = vy; cmd_amd[4 * b + 2] = om; cmd_amd[4 * b + 3] = oka;
  }
  return 0;
}

}  // extern "C"
