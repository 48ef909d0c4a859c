/*
 * teb_oracle.h — C interface of the CPU ORACLE (test infrastructure, NOT product code).
 *
 * The oracle is a dependency-free fp64 restatement of the reference hot path
 * (TebOptimalPlanner::optimizeTEB and its g2o/CSparse back end). Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load libteb_oracle.so; libteb_amd.so never links or calls it.
 *
 * PINNED bit-for-bit against the reference's own sources compiled in place (oracle/_ref/libteb_ref.so: every edge
 * class, geometry, autoResize, src/optimal_planner.cpp's buildGraph / optimizeTEB / computeCurrentCost) by
 * tests/test_reference_pinning.py and the committed tests/golden/ref_*.npz. PARITY UNPINNED only inside the external
 * libg2o (LM iteration, central differences, linear solver: absent from /root/reference, restated). See DESIGN.md section 5.
 */
#ifndef TEB_ORACLE_H_
#define TEB_ORACLE_H_

#include "../include/teb_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* cost_mode for computeCurrentCost */
enum {
  TEB_ORACLE_COST_REFERENCE = 0, /* sum the edges' stored _error exactly like src/optimal_planner.cpp:1070-1072
                                    (stale after a rejected final LM trial unless divergence_detection_enable) */
  TEB_ORACLE_COST_FRESH = 1      /* recompute all errors at the final state first */
};

/* B x optimizeTEB + per-TEB results. batch is updated in place (state and n). threads<=1: sequential;
 * else one std::thread per TEB capped at `threads` (src/homotopy_class_planner.cpp:476-483). */
int teb_oracle_optimize_batch(const teb_amd_config_t* cfg, const teb_amd_obstacles_t* obst,
                              int32_t n_via, const double* via_x, const double* via_y,
                              teb_amd_teb_batch_t* batch,
                              int32_t iterations_innerloop, int32_t iterations_outerloop,
                              int32_t compute_cost_afterwards, double obst_cost_scale,
                              double viapoint_cost_scale, int32_t alternative_time_cost,
                              int32_t cost_mode, int32_t threads, teb_amd_results_t* out);

/* Opt-in trace of the LM loop of the following teb_oracle_optimize_batch calls: trace [B][cap_rows][4] = per LM iteration
 * {chi2 after it, lambda after it, damping trials, pose count}, rows [B] = rows written. trace = NULL: off. */
int teb_oracle_set_trace(double* trace, int32_t cap_rows, int32_t* rows);

/* selectBestTeb (src/homotopy_class_planner.cpp:564-667) on a cost array. */
int teb_oracle_select_best(const teb_amd_config_t* cfg, int32_t count, const double* cost,
                           int32_t last_best, int32_t initial_plan, int32_t* best, double* best_cost);

/* TimedElasticBand::autoResize (src/timed_elastic_band.cpp:227-286) on one strip; arrays have capacity cap. */
int teb_oracle_autoresize(double* x, double* y, double* theta, double* dt, int32_t* n, int32_t cap,
                          double dt_ref, double dt_hysteresis, int32_t min_samples, int32_t max_samples,
                          int32_t fast_mode);

/*
 * Test hook: build the graph of TEB `b` (buildGraph, src/optimal_planner.cpp:323-366) with the given
 * weight_multiplier, compute all errors and linearise once. Outputs in the CANONICAL index space
 * var(i,c) = 4*i + c, c in {x,y,theta,dt}, dimension 4*n (rows/cols of fixed vertices and of the
 * non-existing dt_{n-1} are zero):
 *   H_dense [4n*4n] row-major (full symmetric), b [4n], chi2[4] = {obstacle-type, via-point, time-optimal, other}
 *   n_edges, n_rows (residual rows). Any output pointer may be NULL.
 */
int teb_oracle_linearize(const teb_amd_config_t* cfg, const teb_amd_obstacles_t* obst,
                         int32_t n_via, const double* via_x, const double* via_y,
                         const teb_amd_teb_batch_t* batch, int32_t b, double weight_multiplier,
                         double* H_dense, double* bvec, double* chi2, int32_t* n_edges, int32_t* n_rows);

/* Test hook: the hyper-graph of TEB b in insertion order (buildGraph with the given weight_multiplier), one record
 * per edge, 16 int32 and 56 doubles:
 *   irec: [type, np, pose0, pose1, pose2, nd, dt0, dt1, dim, obst, via, 0...]   (type = EType of teb_oracle.cpp:
 *         0 obstacle, 1 inflated obstacle, 2 dynamic obstacle, 3 via-point, 4 velocity, 5 velocity holonomic, 6 acceleration,
 *         7 acc. start, 8 acc. goal, 9 acc. holonomic, 10 acc. holonomic start, 11 acc. holonomic goal, 12 time-optimal,
 *         13 shortest path, 14 kinematics diff-drive, 15 kinematics car-like, 16 prefer-rotdir, 17 velocity-obstacle-ratio)
 *   drec: [err0..2, info0..2, t, dir, J(3x11 row-major, columns pose0 xyz, pose1 xyz, pose2 xyz, dt0, dt1) in the
 *          configured jacobian_mode, 15 spare]
 * Returns the number of edges in *count (records beyond cap are not written). */
int teb_oracle_edges(const teb_amd_config_t* cfg, const teb_amd_obstacles_t* obst, int32_t n_via, const double* via_x,
                     const double* via_y, const teb_amd_teb_batch_t* batch, int32_t b, double weight_multiplier,
                     int32_t* irec, double* drec, int32_t cap, int32_t* count);

/* Test hook: obstacle association of TEB b (AddEdgesObstacles, src/optimal_planner.cpp:444-548, or the
 * legacy variant :551-643). assoc_pose/assoc_obst receive up to cap (pose, obstacle) pairs in edge
 * insertion order; returns the pair count in *count. */
int teb_oracle_associate(const teb_amd_config_t* cfg, const teb_amd_obstacles_t* obst,
                         const teb_amd_teb_batch_t* batch, int32_t b,
                         int32_t* assoc_pose, int32_t* assoc_obst, int32_t cap, int32_t* count);

/* Test hook: footprint<->obstacle distance (robot_footprint_model.h calculateDistance /
 * estimateSpatioTemporalDistance): spatio_temporal=0 ignores t. Also returns the analytic gradient
 * (d/dx, d/dy, d/dtheta) used by the analytic Jacobian mode in grad[3] (may be NULL). */
int teb_oracle_distance(const teb_amd_config_t* cfg, const teb_amd_obstacles_t* obst, int32_t obst_index,
                        double x, double y, double theta, int32_t spatio_temporal, double t,
                        double* dist, double* grad);

/* Polygon centroid as PolygonObstacle::calcCentroid (src/obstacles.cpp:56-121), and the centroid of any obstacle. */
int teb_oracle_centroid(const teb_amd_obstacles_t* obst, int32_t obst_index, double* cx, double* cy);

/* ---- SURVEY section 8(f) rows f1 (producers of the state strip) and f2 (consumers of the optimised strip) ---------------- */
/* TimedElasticBand::initTrajectoryToGoal(start, goal, diststep, max_vel_x, min_samples, guess_backwards_motion),
 * src/timed_elastic_band.cpp:325-377. start/goal = (x, y, theta). Output arrays have capacity cap. */
int teb_oracle_init_trajectory_line(const double* start, const double* goal, double diststep, double max_vel_x, int32_t min_samples,
                                    int32_t guess_backwards_motion, double* x, double* y, double* th, double* dt, int32_t* n,
                                    int32_t cap);
/* initTrajectoryToGoal(plan, max_vel_x, max_vel_theta, estimate_orient, min_samples, guess_backwards_motion), :380-452;
 * the plan as positions + yaw (the caller applies tf::getYaw). */
int teb_oracle_init_trajectory_plan(int32_t np, const double* px, const double* py, const double* pyaw, double max_vel_x,
                                    double max_vel_theta, int32_t estimate_orient, int32_t min_samples,
                                    int32_t guess_backwards_motion, double* x, double* y, double* th, double* dt, int32_t* n,
                                    int32_t cap);
/* template initTrajectoryToGoal(path_start, path_end, fun_position, ...), timed_elastic_band.hpp:46-183 (what
 * HomotopyClassPlanner::addAndInitNewTeb feeds with graph vertices). */
int teb_oracle_init_trajectory_path(int32_t np, const double* px, const double* py, double max_vel_x, double max_vel_theta,
                                    int32_t has_max_acc_x, double max_acc_x, int32_t has_start_orient, double start_orient,
                                    int32_t has_goal_orient, double goal_orient, int32_t min_samples,
                                    int32_t guess_backwards_motion, double* x, double* y, double* th, double* dt, int32_t* n,
                                    int32_t cap);
/* TimedElasticBand::updateAndPruneTEB (src/timed_elastic_band.cpp:555-597) in place; new_start / new_goal may be NULL. */
int teb_oracle_update_and_prune(double* x, double* y, double* th, double* dt, int32_t* n, const double* new_start,
                                const double* new_goal, int32_t min_samples);
/* TebOptimalPlanner::getVelocityCommand (src/optimal_planner.cpp:1135-1168): v = (vx, vy, omega), *ok = return value. */
int teb_oracle_velocity_command(const teb_amd_config_t* cfg, const teb_amd_teb_batch_t* batch, int32_t b, int32_t look_ahead_poses,
                                int32_t prevent_look_ahead_poses_near_goal, double* v, int32_t* ok);
/* getVelocityProfile (:1170-1196): out[(n+1)*3] = (linear.x, linear.y, angular.z). */
int teb_oracle_velocity_profile(const teb_amd_config_t* cfg, const teb_amd_teb_batch_t* batch, int32_t b, double* out);
/* getFullTrajectory (:1198-1247): out[n*7] = (x, y, theta, vx, vy, omega, time_from_start). */
int teb_oracle_full_trajectory(const teb_amd_config_t* cfg, const teb_amd_teb_batch_t* batch, int32_t b, double* out);

/* ---- row f4, arithmetic part: TebOptimalPlanner::isTrajectoryFeasible (src/optimal_planner.cpp:1250-1308) ------------------------
 * on a uint8 costmap grid. The footprint test is base_local_planner::CostmapModel::footprintCost of the ROS navigation stack
 * (base_local_planner/src/costmap_model.cpp, an un-vendored dependency of the reference: package.xml <depend>base_local_planner</depend>,
 * no version pin; restated from the published noetic sources): footprint vertices rotated / translated to the pose, every edge
 * rasterised with base_local_planner::LineIterator (Bresenham), cell costs LETHAL_OBSTACLE 254 -> -1, NO_INFORMATION 255 -> -2,
 * off the map -> -3, fewer than 3 vertices -> the centre cell alone (253 counts as lethal there). The reference treats ONLY -1 as a
 * collision (:1269, :1292). cells[my * size_x + mx]; world (wx, wy) -> cell ((int)((wx - origin_x) / resolution), ...).
 * *first_infeasible (may be NULL) = number of footprint tests that passed before the failing one, -1 if feasible. */
double teb_oracle_footprint_cost(const uint8_t* cells, int32_t size_x, int32_t size_y, double resolution, double origin_x, double origin_y,
                                 double x, double y, double theta, int32_t n_footprint, const double* fx, const double* fy);
int teb_oracle_is_trajectory_feasible(const teb_amd_teb_batch_t* batch, int32_t b, const uint8_t* cells, int32_t size_x, int32_t size_y,
                                      double resolution, double origin_x, double origin_y, int32_t n_footprint, const double* fx,
                                      const double* fy, double inscribed_radius, double min_resolution_collision_check_angular,
                                      int32_t look_ahead_idx, double feasibility_check_lookahead_distance, int32_t* feasible,
                                      int32_t* first_infeasible);

/* ---- row f3, arithmetic core: equivalence classes (h_signature.h) ---------------------------------------------------------- */
/* HSignature::calculateHSignature, h_signature.h:96-188 -> re_im[2] (long double inside, like the reference) */
int teb_oracle_h_signature_2d(const teb_amd_config_t* cfg, const teb_amd_obstacles_t* obst, const teb_amd_teb_batch_t* batch, int32_t b,
                              double prescaler, double* re_im);
/* HSignature3d::calculateHSignature, h_signature.h:281-347 -> values[M] */
int teb_oracle_h_signature_3d(const teb_amd_config_t* cfg, const teb_amd_obstacles_t* obst, const teb_amd_teb_batch_t* batch, int32_t b,
                              double* values);
/* isValid / isReasonable / isEqual + the class list of renewAndAnalyzeOldTebs / addEquivalenceClassIfNew
 * (src/homotopy_class_planner.cpp:178-254). mode 2: sig [B*2]; mode 3: sig [B*M]. */
int teb_oracle_filter_equivalence_classes(int32_t mode, int32_t B, int32_t M, const double* sig, double threshold, int32_t best,
                                          int32_t max_number_plans_in_current_class, int32_t* keep, int32_t* valid,
                                          int32_t* reasonable);

/* as teb_oracle_filter_equivalence_classes, with best_teb_eq_class_ left over from an earlier tick (used when best < 0; NULL = none) */
int teb_oracle_filter_equivalence_classes_stale(int32_t mode, int32_t B, int32_t M, const double* sig, double threshold, int32_t best,
                                                int32_t max_number_plans_in_current_class, const double* stale_best_sig, int32_t* keep,
                                                int32_t* valid, int32_t* reasonable);
/* deletePlansDetouringBackwards (src/homotopy_class_planner.cpp:766-838): keep [B] in/out, optimized [B] = isOptimized() per band */
int teb_oracle_filter_detours(const teb_amd_teb_batch_t* batch, const teb_amd_hcp_params_t* p, int32_t best, const int32_t* optimized,
                              int32_t* keep);
/* ---- row f3, candidate generation: GraphSearchInterface::createGraph (src/graph_search.cpp:95-340), DepthFirst (:45-91),
 * addAndInitNewTeb (homotopy_class_planner.hpp:66-93) on bands 0..n_tebs-1 (= tebs_ after renewAndAnalyzeOldTebs); batch->count = slots.
 * n_plan > 0: the initial plan, tried first (addAndInitNewTeb(*initial_plan_, ...), :326-329, 412-440); stale_initial_sig =
 * initial_plan_eq_class_ of an earlier tick or NULL; *initial_plan_teb = getInitialPlanTEB(); via_enabled [slots] in/out (may be
 * NULL) = updateReferenceTrajectoryViaPoints given via-points exist.
 * max_paths > 0 bounds the number of start-goal paths examined (test guard; the reference has no bound).
 * Graph out: vertices vx, vy [vcap], adjacency in insertion order as CSR (adj_off [nv+1], adj [acap]). */
int teb_oracle_explore_candidates(const teb_amd_config_t* cfg, const teb_amd_hcp_params_t* p, const teb_amd_obstacles_t* obst,
                                  teb_amd_teb_batch_t* batch, int32_t n_tebs, int32_t best, const double* start, const double* goal,
                                  double dist_to_obst, const double* unit_samples, int64_t skip_draws, int64_t max_paths,
                                  const double* stale_best_sig, int32_t* n_total, int32_t vcap, double* vx, double* vy, int32_t* nv,
                                  int32_t acap, int32_t* adj_off, int32_t* adj, int32_t* n_paths, int32_t n_plan, const double* plan_x,
                                  const double* plan_y, const double* plan_yaw, const double* stale_initial_sig, int32_t* initial_plan_teb,
                                  int32_t* via_enabled);

#ifdef __cplusplus
}
#endif
#endif
