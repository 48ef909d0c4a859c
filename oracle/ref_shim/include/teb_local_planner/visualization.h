// Shadows the reference's visualization.h (rviz publishers; out of scope): no-op TebVisualization
// with the members src/optimal_planner.cpp calls. TEST INFRASTRUCTURE (oracle/_ref build only).
#ifndef VISUALIZATION_H_
#define VISUALIZATION_H_
#include <teb_local_planner/teb_config.h>
#include <teb_local_planner/timed_elastic_band.h>
#include <teb_local_planner/robot_footprint_model.h>
namespace teb_local_planner {
class TebOptimalPlanner;
class TebVisualization {
 public:
  void publishLocalPlanAndPoses(const TimedElasticBand&) const {}
  void publishRobotFootprintModel(const PoseSE2&, const BaseRobotFootprintModel&, const std::string& = "RobotFootprintModel",
                                  const std_msgs::ColorRGBA& = std_msgs::ColorRGBA()) {}
  void publishInfeasibleRobotPose(const PoseSE2&, const BaseRobotFootprintModel&, const std::vector<geometry_msgs::Point>& = {}) {}
  void publishFeedbackMessage(const TebOptimalPlanner&, const ObstContainer&) {}
  template <class TebContainer> void publishFeedbackMessage(const TebContainer&, unsigned int, const ObstContainer&) {}
  template <class Graph> void publishGraph(const Graph&, const std::string& = "Graph") {}
  template <class TebContainer> void publishTebContainer(const TebContainer&, const std::string& = "TebContainer") {}
};
typedef boost::shared_ptr<TebVisualization> TebVisualizationPtr;
typedef boost::shared_ptr<const TebVisualization> TebVisualizationConstPtr;
}  // namespace teb_local_planner
#endif
