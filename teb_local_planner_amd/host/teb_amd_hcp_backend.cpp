// teb_amd_hcp_backend.cpp — see teb_amd_hcp_backend.h. Written against the reference's HomotopyClassPlanner; cites refer to
// src/homotopy_class_planner.cpp.
#include <complex>
#include <vector>
// the two equivalence-class types keep their values private and offer no setter: the classes computed on the device are written into
// objects of the reference's own types, so that everything that reads equivalence_classes_ keeps working
#ifdef TEB_AMD_BACKEND_HAVE_ACCESSORS   // host/patches/footprint_and_signature_accessors.patch applied
#include <teb_local_planner/h_signature.h>
#define HSIG3D_SET(H, first, last) (H)->setValues(std::vector<double>(first, last))
#define HSIG2D_SET(H, v) (H)->setValue(v)
#else
#define private public
#include <teb_local_planner/h_signature.h>
#undef private
#define HSIG3D_SET(H, first, last) (H)->hsignature3d_.assign(first, last)
#define HSIG2D_SET(H, v) (H)->hsignature_ = (v)
#endif
#include "teb_amd_hcp_backend.h"

namespace teb_local_planner {

HomotopyClassPlannerAmd::HomotopyClassPlannerAmd(const TebConfig& cfg, ObstContainer* obstacles, TebVisualizationPtr visualization,
                                                 const ViaPointContainer* via_points, int max_tebs, int max_poses, int max_obstacles,
                                                 int max_obstacle_vertices, int max_via_points, int device)
  : HomotopyClassPlanner(cfg, obstacles, visualization, via_points),
    batch_(new TebAmdBatch(cfg, max_tebs, max_poses, max_obstacles, max_obstacle_vertices, max_via_points, device))
{
}

void HomotopyClassPlannerAmd::setCommunicator(teb_amd_comm_t* comm, int rank)
{
  rank_ = rank;
  batch_->setCommunicator(comm, rank * batch_->maxTebs());
  best_global_ = -1; best_owner_ = -1;
  remote_best_.reset();
}

bool HomotopyClassPlannerAmd::plan(const PoseSE2& start, const PoseSE2& goal, const geometry_msgs::Twist* start_vel, bool free_goal_vel)
{
  ROS_ASSERT_MSG(initialized_, "Call initialize() first.");
  // Update old TEBs with new start, goal and velocity (:539-562)
  updateAllTEBs(&start, &goal, start_vel);

  // ---- exploreEquivalenceClassesAndInitTebs (:318-340) on the device ------------------------------------------------------------------
  std::vector<TebOptimalPlannerAmdPtr> cand;
  int best_index = -1;
  for (std::size_t i = 0; i < tebs_.size(); ++i)
  {
    TebOptimalPlannerAmdPtr p = boost::dynamic_pointer_cast<TebOptimalPlannerAmd>(tebs_[i]);
    if (!p) continue;   // a candidate created by foreign code: cannot be moved to the device
    if (tebs_[i] == best_teb_) best_index = (int)cand.size();
    cand.push_back(p);
  }
  int initial_index = -1;
  // randomlyDropTebs (:539-562) with the planner's own generator, exactly the reference's draw
  std::function<bool()> drop;
  if (cfg_->hcp.selection_dropping_probability != 0.0)
    drop = [this]() { return random_() <= cfg_->hcp.selection_dropping_probability * random_.max(); };
  // In the sharded mode plan() is a collective: a rank on which a step fails must not return before the selection exchange, or its peers
  // wait in the all-gather for ever. It records the failure, skips the rest of its own work, enters the exchange with the
  // unusable record, follows the peers through the winner broadcast and returns false afterwards.
  bool local_ok = true;
  if (!batch_->exploreEquivalenceClassesAndInitTebs(*cfg_, obstacles_, via_points_, cand, best_index, start, goal,
                                                    cfg_->obstacles.min_obstacle_dist, start_vel, free_goal_vel, initial_plan_, &initial_index,
                                                    &drop))
  {
    if (!batch_->sharded()) return false;
    local_ok = false; cand.clear(); initial_index = -1; best_index = -1;
  }
  tebs_.clear();
  for (const TebOptimalPlannerAmdPtr& p : cand) tebs_.push_back(p);
  initial_plan_teb_ = initial_index >= 0 ? tebs_[initial_index] : TebOptimalPlannerPtr();
  // equivalence_classes_: the reference's own class objects, filled with the signatures computed on the device
  std::vector<double> values;
  int width = 0;
  equivalence_classes_.clear();
  if (!tebs_.empty())
  {
    if (!batch_->signatures(*cfg_, values, width))
    {
      if (!batch_->sharded()) return false;
      local_ok = false;
    }
    for (std::size_t b = 0; local_ok && b < tebs_.size(); ++b)
    {
      if (cfg_->obstacles.include_dynamic_obstacles)
      {
        HSignature3d* H = new HSignature3d(*cfg_);
        HSIG3D_SET(H, values.begin() + b * width, values.begin() + (b + 1) * width);
        equivalence_classes_.push_back(std::make_pair(EquivalenceClassPtr(H), false));
      }
      else
      {
        HSignature* H = new HSignature(*cfg_);
        HSIG2D_SET(H, std::complex<long double>(values[b * width], values[b * width + 1]));
        equivalence_classes_.push_back(std::make_pair(EquivalenceClassPtr(H), false));
      }
    }
    if (local_ok && best_index >= 0) best_teb_eq_class_ = equivalence_classes_[best_index].first;
    if (local_ok && initial_index >= 0 && initial_plan_) initial_plan_eq_class_ = equivalence_classes_[initial_index].first;
  }
  // update via-points if activated: done on the device-side flags and mirrored onto the candidates by the call above (:286-315)

  if (tebs_.empty() && !batch_->sharded())
  {
    best_teb_.reset();
    initial_plan_ = nullptr;
    return true;
  }
  // ---- optimizeAllTEBs (:466-493): one launch ------------------------------------------------------------------------------------------
  if (local_ok && !tebs_.empty())
  {
    std::vector<TebOptimalPlannerAmd*> raw;
    for (const TebOptimalPlannerAmdPtr& p : cand) raw.push_back(p.get());
    batch_->optimizeAllTEBs(raw, cfg_->optim.no_inner_iterations, cfg_->optim.no_outer_iterations, true, cfg_->hcp.selection_obst_cost_scale,
                            cfg_->hcp.selection_viapoint_cost_scale, cfg_->hcp.selection_alternative_time_cost);
    // a library error (not: candidates whose optimisation legitimately returned false) leaves stale costs on the device: this rank
    // must not offer them to the selection
    if (!batch_->lastCallOk())
    {
      if (!batch_->sharded()) return false;
      local_ok = false;
    }
  }
  // ---- selectBestTeb (:564-667) over the candidates of every rank that shares the batch ----------------------------------------------------
  if (batch_->sharded())
  {
    const int off = batch_->globalOffset();
    int initial_global = initial_index >= 0 ? off + initial_index : -1;
    int last_global = -1;   // the last winner survives as a candidate of its owner (moved to the front by the exploration, :766-838)
    if (best_owner_ == rank_ && best_teb_)
      for (std::size_t i = 0; i < tebs_.size(); ++i) if (tebs_[i] == best_teb_) last_global = off + (int)i;
    int owner = -1;
    int sel;
    if (local_ok)
    {   // a local failure INSIDE the exchange (after this rank's record went out) still yields the peers' choice: follow them
      bool sel_ok = true;
      sel = batch_->selectBestTebDistributed(last_global, initial_global, NULL, &owner, &sel_ok);
      if (!sel_ok) local_ok = false;
    }
    else
      sel = batch_->selectBestTebDistributedAsFailedRank(&owner);
    TebOptimalPlannerPtr previous = best_teb_;
    // sel < 0 is the WORLD's verdict (no rank holds a candidate): every rank sees it and none enters the broadcast
    if (sel < 0) { best_teb_.reset(); best_global_ = -1; best_owner_ = -1; initial_plan_ = nullptr; return local_ok; }
    if (owner == rank_ && local_ok)
      best_teb_ = tebs_[sel - off];
    else
    {
      if (!remote_best_) remote_best_.reset(new TebOptimalPlannerAmd(*cfg_, obstacles_, visualization_, via_points_));
      best_teb_ = remote_best_;
    }
    // the winner's band on every rank (a collective: the owner takes part as well); non-owners mirror it
    // Where the broadcast unpacks: the mirror whenever best_teb_ IS the mirror - also on the owner itself when its own work failed this
    // tick (local_ok false): its strip still went out to the peers, and hasDiverged() / getVelocityCommand() on this rank then read the
    // band the world agreed on, not an empty mirror. The healthy owner keeps its candidate and unpacks into a scratch band.
    TimedElasticBand scratch;
    TimedElasticBand& unpack_into = (best_teb_ == remote_best_) ? remote_best_->teb() : scratch;
    if (!batch_->broadcastBand(owner, owner == rank_ ? sel - off : 0, unpack_into)) return false;
    if (best_teb_ == remote_best_) batch_->adoptBroadcastStatistics(*remote_best_);   // hasDiverged() forwards to best_teb_ (:749-755)
    (void)previous;   // the switching_blocking_period rule needs one clock for all ranks: it stays with the caller in the sharded mode
    best_global_ = sel; best_owner_ = owner;
    initial_plan_ = nullptr;
    return local_ok;
  }
  {
    int last_best = -1;
    for (std::size_t i = 0; i < tebs_.size(); ++i) if (tebs_[i] == best_teb_) last_best = (int)i;
    last_best_teb_ = last_best >= 0 ? best_teb_ : TebOptimalPlannerPtr();
    const int sel = batch_->selectBestTeb(last_best, initial_index);
    best_teb_ = sel >= 0 ? tebs_[sel] : TebOptimalPlannerPtr();
    if (last_best_teb_ && best_teb_ != last_best_teb_)   // check if we are allowed to change (:648-663)
    {
      ros::Time now = ros::Time::now();
      if ((now - last_eq_class_switching_time_).toSec() > cfg_->hcp.switching_blocking_period)
        last_eq_class_switching_time_ = now;
      else
        best_teb_ = last_best_teb_;   // block switching
    }
  }
  initial_plan_ = nullptr;   // any previous plan is useless regarding the h-signature (:123)
  return true;
}

} // namespace teb_local_planner
