// teb_amd_hcp_backend.h — HomotopyClassPlanner whose planning tick runs on the MI355X: the class a teb_local_planner maintainer
// instantiates instead of HomotopyClassPlanner (src/teb_local_planner_ros.cpp creates it when enable_homotopy_class_planning is set).
// plan() keeps the reference's sequence (src/homotopy_class_planner.cpp:107-125) and its members (tebs_, best_teb_, equivalence_classes_,
// initial_plan_teb_, ...) stay meaningful for everything the class inherits unchanged (getVelocityCommand, isTrajectoryFeasible,
// visualize, ...). hasDiverged() is inherited too: it forwards to best_teb_ (:749-755), a TebOptimalPlannerAmd, which answers from the
// statistics the launch returned (in the sharded mode a mirror of the winner carries the statistics that came with its band):
//   updateAllTEBs                         inherited (O(n) per candidate on the host objects)
//   exploreEquivalenceClassesAndInitTebs  TebAmdBatch::exploreEquivalenceClassesAndInitTebs (signatures, class list, detours, graph,
//                                         candidate bands, via-point flags on the device)
//   optimizeAllTEBs                       TebAmdBatch::optimizeAllTEBs (one kernel launch for all candidates)
//   selectBestTeb                         TebAmdBatch::selectBestTeb + the reference's switching_blocking_period rule
//   randomlyDropTebs (:539-562)           taken over inside the exploration call with the planner's own generator `random_` (the reference's
//                                         draw, same sequence; selection_dropping_probability is 0 by default)
// Built where the reference's headers exist (oracle/ref_shim/Makefile -> libteb_backend_check.so); exercised against the reference's
// own HomotopyClassPlanner tick by tick in tests/test_reference_backend.py.
#ifndef TEB_AMD_HCP_BACKEND_H_
#define TEB_AMD_HCP_BACKEND_H_

#include <teb_local_planner/homotopy_class_planner.h>

#include "teb_amd_backend.h"

namespace teb_local_planner {

class HomotopyClassPlannerAmd : public HomotopyClassPlanner
{
public:
  /** max_tebs >= hcp.max_number_classes; max_poses = pose capacity per candidate (trajectory.max_samples + 1 <= TEB_AMD_MAX_POSES covers autoResize). */
  HomotopyClassPlannerAmd(const TebConfig& cfg, ObstContainer* obstacles = NULL, TebVisualizationPtr visualization = TebVisualizationPtr(),
                          const ViaPointContainer* via_points = NULL, int max_tebs = 16, int max_poses = 512, int max_obstacles = 512,
                          int max_obstacle_vertices = 4096, int max_via_points = 256, int device = 0);

  using HomotopyClassPlanner::plan;   // plan(initial_plan, ...) and plan(tf::Pose, ...) forward to the virtual overload below
  virtual bool plan(const PoseSE2& start, const PoseSE2& goal, const geometry_msgs::Twist* start_vel = NULL, bool free_goal_vel = false);

  const std::string& lastError() const { return batch_->lastError(); }

  /**
   * Candidate classes SHARDED over several GPUs (one process per GPU; the batch of HomotopyClassPlanner is the data-parallel axis,
   * src/homotopy_class_planner.cpp:466-493): every rank explores, keeps and optimises its own candidates; selectBestTeb (:564-667)
   * runs over all of them - one 16-byte record per rank through the communicator - and the winner's band reaches every rank, so that
   * bestTeb() / getVelocityCommand work on every rank (on the ranks that do not own it best_teb_ is a mirror object that is not part
   * of tebs_). Global candidate index = rank * max_tebs + local index. comm stays the caller's (teb_amd_comm_create); NULL switches
   * the sharding off. plan() becomes a collective: every rank of the communicator has to call it each tick.
   */
  void setCommunicator(teb_amd_comm_t* comm, int rank);
  int bestTebGlobalIndex() const { return best_global_; }   //!< of the last plan(), -1 = none / not sharded
  int bestTebOwnerRank() const { return best_owner_; }

private:
  boost::shared_ptr<TebAmdBatch> batch_;
  int rank_ = 0;
  int best_global_ = -1, best_owner_ = -1;
  TebOptimalPlannerAmdPtr remote_best_;   //!< mirror of a winner another rank owns
};

} // namespace teb_local_planner

#endif
