// teb_amd_backend.h — host side of the drop-in, written against the REFERENCE's own classes.
//
// This is the translation unit a teb_local_planner maintainer adds: adapters from TebConfig /
// ObstContainer / ViaPointContainer / TimedElasticBand to the C-ABI of libteb_amd.so (include/teb_amd.h), a
// TebOptimalPlanner subclass whose optimizeTEB() runs on the MI355X, and the batched replacement of
// HomotopyClassPlanner::optimizeAllTEBs / selectBestTeb. Same method names, argument meaning and return values as
//   TebOptimalPlanner::optimizeTEB                 include/teb_local_planner/optimal_planner.h:231, src/optimal_planner.cpp:183-233
//   HomotopyClassPlanner::optimizeAllTEBs          src/homotopy_class_planner.cpp:466-493
//   HomotopyClassPlanner::selectBestTeb (arg-min)  src/homotopy_class_planner.cpp:564-667
// It needs the reference's headers to compile, so in this repository it is built only where /root/reference exists
// (oracle/ref_shim/Makefile -> oracle/_ref/libteb_backend_check.so) and exercised by tests/test_reference_backend.py.
#ifndef TEB_AMD_BACKEND_H_
#define TEB_AMD_BACKEND_H_

#include <teb_amd.h>

#include <teb_local_planner/optimal_planner.h>

#include <functional>
#include <string>
#include <vector>

namespace teb_local_planner {

//! TebConfig (teb_config.h:62-430) -> flat teb_amd_config_t; robot_model is flattened by toAmdFootprint. Returns false (fails closed:
//! every planner built on it then reports optimizeTEB() == false) when the footprint model cannot be represented on the device.
bool toAmdConfig(const TebConfig& cfg, teb_amd_config_t& out);

//! One of the five footprint classes (robot_footprint_model.h:134-770) -> footprint_* fields. Returns false for an unknown class.
bool toAmdFootprint(const BaseRobotFootprintModel& model, teb_amd_config_t& out);

//! ObstContainer (obstacles.h:262) as the SoA table of the C-ABI; owns the arrays the view points into.
struct AmdObstacleTable
{
  std::vector<int32_t> type, dynamic, vert_offset;
  std::vector<double> ax, ay, bx, by, radius, vx, vy, vert_x, vert_y;
  void assign(const ObstContainer* obstacles);
  teb_amd_obstacles_t view() const;
  int vertices() const { return (int)vert_x.size(); }
};

//! Extension of the C-ABI that TebConfig has no field for (TEB_AMD_JACOBIAN_*); process-wide, default analytic.
void setAmdJacobianMode(int mode);

class TebAmdBatch;

//! teb_amd_hcp_params_t of a TebConfig (hcp.*, goal_tolerance.xy_goal_tolerance, trajectory.allow_init_with_backwards_motion)
void toAmdHcpParams(const TebConfig& cfg, teb_amd_hcp_params_t& p);

/**
 * TebOptimalPlanner whose optimisation is executed by libteb_amd.so, usable through the pointer the plugin holds (PlannerInterfacePtr
 * planner_, src/teb_local_planner_ros.cpp:122). TebOptimalPlanner::optimizeTEB is NOT virtual (optimal_planner.h:231) and the three
 * inherited plan() overloads call it (src/optimal_planner.cpp:279, 319): a subclass that only re-declared optimizeTEB would plan on
 * the CPU with g2o whenever it is driven through the interface. So this class overrides every virtual that ends in the optimiser or
 * reads its state:
 *   plan() x 3      the reference's sequence (warm start / re-initialisation on the host TimedElasticBand, velocity flags), then
 *                   THIS class's optimizeTEB
 *   hasDiverged()   optimizer_->batchStatistics() is never filled here; the rule of src/optimal_planner.cpp:1023-1039 is applied to
 *                   the statistics the launch returned (teb_amd_get_batch_statistics)
 * Inherited unchanged: velocity extraction, isTrajectoryFeasible, visualisation, clearPlanner. computeCurrentCost(...) as a separate
 * call still evaluates the reference's edge classes on the host (no LM iteration runs there); the cost of a plan comes back from the
 * launch (compute_cost_afterwards). With host/patches/virtual_optimizeTEB.patch applied to the reference the plan() overrides become
 * redundant, not wrong.
 */
class TebOptimalPlannerAmd : public TebOptimalPlanner
{
public:
  TebOptimalPlannerAmd(const TebConfig& cfg, ObstContainer* obstacles = NULL,
                       TebVisualizationPtr visual = TebVisualizationPtr(), const ViaPointContainer* via_points = NULL);

  //! Same contract as TebOptimalPlanner::optimizeTEB (optimal_planner.h:204-232).
  bool optimizeTEB(int iterations_innerloop, int iterations_outerloop, bool compute_cost_afterwards = false,
                   double obst_cost_scale = 1.0, double viapoint_cost_scale = 1.0, bool alternative_time_cost = false);

  //! PlannerInterface::plan (planner_interface.h:99-125); same contracts as TebOptimalPlanner::plan (src/optimal_planner.cpp:247-320).
  virtual bool plan(const std::vector<geometry_msgs::PoseStamped>& initial_plan, const geometry_msgs::Twist* start_vel = NULL, bool free_goal_vel = false);
  virtual bool plan(const tf::Pose& start, const tf::Pose& goal, const geometry_msgs::Twist* start_vel = NULL, bool free_goal_vel = false);
  virtual bool plan(const PoseSE2& start, const PoseSE2& goal, const geometry_msgs::Twist* start_vel = NULL, bool free_goal_vel = false);

  //! PlannerInterface::hasDiverged (planner_interface.h:198) on the statistics of the last launch that optimised this band.
  bool hasDiverged() const override;

  //! Statistics of the last run (not available from g2o in the reference): LM iterations / damped-solve trials.
  int lastLmIterations() const { return lm_iterations_; }
  int lastLmTrials() const { return lm_trials_; }

private:
  friend class TebAmdBatch;
  //! what both band-producing plan() overloads share: keep or rebuild the band, velocity flags, optimise
  template <class Init> bool planOnBand(const PoseSE2& start, const PoseSE2& goal, Init init, const geometry_msgs::Twist* start_vel, bool free_goal_vel);
  boost::shared_ptr<TebAmdBatch> single_;   //!< lazily created batch of one
  int lm_iterations_ = 0, lm_trials_ = 0;
  bool stats_available_ = false;            //!< optimizer_->batchStatistics() would not be empty
  double stats_back_chi2_ = 0;              //!< its .back().chi2
};
typedef boost::shared_ptr<TebOptimalPlannerAmd> TebOptimalPlannerAmdPtr;

/**
 * One libteb_amd.so handle = device buffers for up to max_tebs candidates. optimizeAllTEBs() packs the candidates'
 * bands into one batch, runs ONE kernel launch for the whole outer x inner loop of every candidate and writes bands,
 * cost and the optimized flag back into the planner objects.
 */
class TebAmdBatch
{
public:
  TebAmdBatch(const TebConfig& cfg, int max_tebs, int max_poses, int max_obstacles, int max_obstacle_vertices,
              int max_via_points, int device = 0);
  ~TebAmdBatch();
  TebAmdBatch(const TebAmdBatch&) = delete;
  TebAmdBatch& operator=(const TebAmdBatch&) = delete;

  /**
   * Replaces the body of HomotopyClassPlanner::optimizeAllTEBs. All candidates share cfg / obstacles / via-points of
   * the first one (as in the reference, where they are constructed from the same pointers, homotopy_class_planner.cpp:434-449).
   * Cost parameters as passed by the reference: cfg.hcp.selection_obst_cost_scale, selection_viapoint_cost_scale,
   * selection_alternative_time_cost. Returns the number of candidates whose optimizeTEB "returned true"; lastCallOk() tells a
   * library error (lastError()) from candidates that legitimately failed.
   */
  int optimizeAllTEBs(const std::vector<TebOptimalPlannerAmd*>& tebs, int iter_innerloop, int iter_outerloop,
                      bool compute_cost_afterwards, double obst_cost_scale, double viapoint_cost_scale,
                      bool alternative_time_cost);

  /**
   * The arg-min of selectBestTeb over the costs of the last optimizeAllTEBs (strict '<', hysteresis on last_best,
   * selection_prefer_initial_plan on initial_plan; -1 = none). Returns the index into the vector passed to optimizeAllTEBs, -1 if empty.
   */
  int selectBestTeb(int last_best, int initial_plan, double* best_cost = NULL);

  /**
   * A candidate batch SHARDED over several GPUs (one process per GPU): the ranks that share it exchange nothing but the
   * selection. comm = the communicator of those ranks (teb_amd_comm_create; it stays the caller's), global_offset = the global
   * index of this rank's first candidate. NULL = this rank alone.
   */
  void setCommunicator(teb_amd_comm_t* comm, int global_offset) { comm_ = comm; global_offset_ = global_offset; }
  bool sharded() const { return comm_ != NULL; }
  int globalOffset() const { return global_offset_; }
  /**
   * selectBestTeb over the candidates of ALL ranks (collective; one 16-byte record per rank through RCCL): indices are GLOBAL
   * (offset of the owning rank + local index, -1 = none). Returns the global index of the winner (-1: no rank holds a candidate), its
   * owner in *owner_rank. Same arithmetic as selectBestTeb: the rank that owns last_best / initial_plan applies the multipliers.
   * A rank on which the call fails locally has still taken part in the exchange: the peers' choice is returned with *local_ok = false
   * (without local_ok: -1), so that the caller can follow them into broadcastBand, a collective too.
   */
  int selectBestTebDistributed(int last_best_global, int initial_plan_global, double* best_cost = NULL, int* owner_rank = NULL, bool* local_ok = NULL);
  int selectBestTebDistributedAsFailedRank(int* owner_rank = NULL);   // a rank whose tick failed still enters the collective (unusable record)
  /** The winner's band from its owner to every rank (collective): `teb` is rebuilt from it on every rank. */
  bool broadcastBand(int owner_rank, int local_index, TimedElasticBand& teb);
  /** The winner's batch statistics came with its band: a mirror object answers hasDiverged() like the owner's planner. */
  void adoptBroadcastStatistics(TebOptimalPlannerAmd& mirror) const;

  int maxTebs() const { return max_tebs_; }
  int maxPoses() const { return max_poses_; }
  int maxObstacles() const { return max_obstacles_; }
  int maxObstacleVertices() const { return max_obstacle_vertices_; }
  int maxViaPoints() const { return max_via_points_; }
  bool valid() const { return h_ != NULL; }   //!< false: the device / footprint / capacity was refused at construction (lastError())
  /**
   * HomotopyClassPlanner::updateAllTEBs (src/homotopy_class_planner.cpp:539-562): TimedElasticBand::updateAndPruneTEB(start, goal,
   * cfg.trajectory.min_samples) on every candidate - one launch on the device - and setVelocityStart(*start_velocity).
   * NULL = boost::none / no new start velocity. The pruned bands are written back into the planner objects.
   */
  bool updateAllTEBs(const std::vector<TebOptimalPlannerAmd*>& tebs, const PoseSE2* start, const PoseSE2* goal,
                     const geometry_msgs::Twist* start_velocity);

  /**
   * The equivalence classes of renewAndAnalyzeOldTebs (src/homotopy_class_planner.cpp:214-254) for the candidates of the last
   * optimizeAllTEBs / updateAllTEBs call: HSignature3d or HSignature of every band in one launch (values(): [B * width]), then the
   * first-come-first-served class list with the last best candidate first. keep[b] == false -> the reference erases candidate b.
   */
  bool renewAndAnalyzeOldTebs(const TebConfig& cfg, int best_index, std::vector<bool>& keep, std::vector<double>* values = NULL,
                              int* width = NULL);

  /**
   * Replaces the body of HomotopyClassPlanner::exploreEquivalenceClassesAndInitTebs (src/homotopy_class_planner.cpp:318-340; not
   * randomlyDropTebs), with initial_plan = initial_plan_ (or NULL) and *initial_plan_index = the index of getInitialPlanTEB() in `tebs`
   * afterwards (for selectBestTeb); random_drop = the draw of randomlyDropTebs (:539-562; NULL or empty: no dropping), incl. updateReferenceTrajectoryViaPoints on the device-side attributes: renewAndAnalyzeOldTebs incl. deletePlansDetouringBackwards on the
   * candidates in `tebs` (erased ones leave the vector; the last best candidate, best_index, moves to the front), then
   * createGraph / DepthFirst / addAndInitNewTeb on the device; every new band arrives as a new TebOptimalPlannerAmd appended to
   * `tebs` (constructed like the reference's candidates: same cfg, obstacles, via-points; setVelocityStart / setVelocityGoalFree
   * applied). best_index is updated (0 or -1). Returns false on a library error (lastError()).
   */
  bool exploreEquivalenceClassesAndInitTebs(const TebConfig& cfg, ObstContainer* obstacles, const ViaPointContainer* via_points,
                                            std::vector<TebOptimalPlannerAmdPtr>& tebs, int& best_index, const PoseSE2& start,
                                            const PoseSE2& goal, double dist_to_obst, const geometry_msgs::Twist* start_vel,
                                            bool free_goal_vel, const std::vector<geometry_msgs::PoseStamped>* initial_plan = NULL,
                                            int* initial_plan_index = NULL, const std::function<bool()>* random_drop = NULL);

  //! H-signatures of the bands currently on the device (after exploreEquivalenceClassesAndInitTebs: the candidates in `tebs` order):
  //! values [B * width], width = #obstacles (HSignature3d) or 2 (HSignature: re, im).
  bool signatures(const TebConfig& cfg, std::vector<double>& values, int& width);

  //! TebOptimalPlanner::getVelocityCommand (src/optimal_planner.cpp:1135-1168) of candidate `index`, from the device-resident band.
  bool getVelocityCommand(const TebConfig& cfg, int index, double& vx, double& vy, double& omega, int look_ahead_poses);

  /**
   * TebOptimalPlanner::isTrajectoryFeasible (src/optimal_planner.cpp:1250-1308) of candidate `index` on the device, from the resident
   * band. The reference takes a base_local_planner::CostmapModel*; the adapter owns the costmap behind it (costmap_ros->getCostmap()):
   * hand its grid to setCostmap() once per tick - Costmap2D::getCharMap(), getSizeInCellsX/Y(), getResolution(), getOriginX/Y() - then
   * call this with footprint_spec_, robot_inscribed_radius_ and cfg.trajectory.{min_resolution_collision_check_angular,
   * feasibility_check_no_poses, feasibility_check_lookahead_distance} as src/teb_local_planner_ros.cpp:396 does. Returns false on a
   * library error as well (lastError()).
   */
  bool setCostmap(const unsigned char* cells, unsigned int size_x, unsigned int size_y, double resolution, double origin_x, double origin_y);
  bool isTrajectoryFeasible(int index, const std::vector<geometry_msgs::Point>& footprint_spec, double inscribed_radius,
                            double min_resolution_collision_check_angular, int look_ahead_idx, double feasibility_check_lookahead_distance);

  const std::string& lastError() const { return error_; }
  bool lastCallOk() const { return last_call_ok_; }   //!< the last optimizeAllTEBs reached the device and read its results back
  float lastKernelMs() const;

private:
  bool check(int rc, const char* what);
  bool uploadBands(const std::vector<TebOptimalPlannerAmd*>& tebs);
  bool downloadBands(const std::vector<TebOptimalPlannerAmd*>& tebs);
  teb_amd_handle_t* h_ = NULL;
  teb_amd_comm_t* comm_ = NULL;
  int global_offset_ = 0;
  int max_tebs_, max_poses_, max_obstacles_ = 0, max_obstacle_vertices_ = 0, max_via_points_ = 0;
  std::string error_;
  bool last_call_ok_ = true;
};

} // namespace teb_local_planner

#endif
