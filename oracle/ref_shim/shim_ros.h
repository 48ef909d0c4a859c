// shim_ros.h — stand-ins for the ROS / Boost / message types that appear in the signatures of the reference headers on
// the optimiser path. TEST INFRASTRUCTURE (oracle/_ref build only). No behaviour of the path lives in these types.
#pragma once
#include <cfloat>
#include <algorithm>
#include <array>
#include <cassert>
#include <iterator>
#include <cmath>
#include <cstdio>
#include <memory>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

// ---- ros/console.h, ros/assert.h
#define ROS_DEPRECATED
#define ROS_ASSERT(cond) assert(cond)
#define ROS_ASSERT_MSG(cond, ...) do { if (!(cond)) { std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); assert(cond); } } while (0)
#define ROS_DEBUG(...) do {} while (0)
#define ROS_DEBUG_COND(c, ...) do {} while (0)
#define ROS_DEBUG_ONCE(...) do {} while (0)
#define ROS_INFO(...) do {} while (0)
#define ROS_INFO_ONCE(...) do {} while (0)
#define ROS_WARN(...) do {} while (0)
#define ROS_WARN_ONCE(...) do {} while (0)
#define ROS_WARN_COND(c, ...) do {} while (0)
#define ROS_ERROR(...) do { std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define ROS_ERROR_COND(c, ...) do {} while (0)

namespace ros {
class NodeHandle {};
struct Duration { double d = 0; explicit Duration(double x = 0) : d(x) {} double toSec() const { return d; } Duration& fromSec(double x) { d = x; return *this; } };
struct Time { double t = 0; static Time now() { static double clock = 0; Time r; r.t = (clock += 1.0); return r; }   // a clock that advances: 1 s per call
  double toSec() const { return t; } Duration operator-(const Time& o) const { return Duration(t - o.t); } };
inline bool ok() { return true; }
}  // namespace ros

namespace boost {
template <typename T> using shared_ptr = std::shared_ptr<T>;
template <typename T, typename... A> shared_ptr<T> make_shared(A&&... a) { return std::make_shared<T>(std::forward<A>(a)...); }
template <typename T, typename U> shared_ptr<T> dynamic_pointer_cast(const shared_ptr<U>& p) { return std::dynamic_pointer_cast<T>(p); }
template <typename T, typename U> shared_ptr<T> static_pointer_cast(const shared_ptr<U>& p) { return std::static_pointer_cast<T>(p); }
using mutex = std::mutex;
using once_flag = std::once_flag;
template <typename F> void call_once(F f, once_flag& fl) { std::call_once(fl, f); }
#define BOOST_ONCE_INIT {}
template <typename T> struct is_pointer : std::is_pointer<T> {};
template <typename C, typename T = void> struct disable_if : std::enable_if<!C::value, T> {};
template <typename C, typename T = void> struct enable_if : std::enable_if<C::value, T> {};
struct none_t {};
static const none_t none = {};
template <typename It> It prior(It it) { return std::prev(it); }
template <typename It> It next(It it) { return std::next(it); }
// boost::optional<T> / optional<const T&> as used by TimedElasticBand
template <typename T>
class optional {
  bool has_ = false;
  T v_{};
 public:
  optional() {}
  optional(none_t) {}
  optional(const T& v) : has_(true), v_(v) {}
  explicit operator bool() const { return has_; }
  const T& operator*() const { return v_; }
  const T& get() const { return v_; }
};
template <typename T>
class optional<const T&> {
  const T* p_ = nullptr;
 public:
  optional() {}
  optional(none_t) {}
  optional(const T& v) : p_(&v) {}
  explicit operator bool() const { return p_ != nullptr; }
  const T& operator*() const { return *p_; }
  const T* operator->() const { return p_; }
};
template <typename T> bool operator==(const optional<T>& o, none_t) { return !static_cast<bool>(o); }
template <typename T> bool operator!=(const optional<T>& o, none_t) { return static_cast<bool>(o); }
namespace math { template <typename T> inline int sign(const T& z) { return (z == 0) ? 0 : (z < 0 ? -1 : 1); } }
}  // namespace boost

namespace std_msgs {
struct Header { std::string frame_id; unsigned seq = 0; ros::Time stamp; };
struct ColorRGBA { float r = 0, g = 0, b = 0, a = 0; };
}  // namespace std_msgs

namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Point32 { float x = 0, y = 0, z = 0; };
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { std_msgs::Header header; Pose pose; };
struct PoseArray { std_msgs::Header header; std::vector<Pose> poses; };
struct Polygon { std::vector<Point32> points; };
struct Twist { Vector3 linear, angular; };
struct TwistWithCovariance { Twist twist; double covariance[36] = {0}; };
struct QuaternionStamped { std_msgs::Header header; Quaternion quaternion; };
}  // namespace geometry_msgs

namespace tf {
inline double getYaw(const geometry_msgs::Quaternion& q) {
  return std::atan2(2.0 * (q.w * q.z + q.x * q.y), 1.0 - 2.0 * (q.y * q.y + q.z * q.z));
}
inline geometry_msgs::Quaternion createQuaternionMsgFromYaw(double yaw) {
  geometry_msgs::Quaternion q; q.x = 0; q.y = 0; q.z = std::sin(yaw / 2); q.w = std::cos(yaw / 2); return q;
}
struct Vector3 { double v[3] = {0, 0, 0}; double getX() const { return v[0]; } double getY() const { return v[1]; } double x() const { return v[0]; } double y() const { return v[1]; } };
struct Quaternion { geometry_msgs::Quaternion q; };
inline double getYaw(const Quaternion& q) { return getYaw(q.q); }
struct Pose { Vector3 o; Quaternion r; const Vector3& getOrigin() const { return o; } const Quaternion& getRotation() const { return r; } };
}  // namespace tf

namespace visualization_msgs {
struct Marker {
  enum { ARROW = 0, CUBE = 1, SPHERE = 2, CYLINDER = 3, LINE_STRIP = 4, LINE_LIST = 5, CUBE_LIST = 6, SPHERE_LIST = 7, POINTS = 8 };
  enum { ADD = 0 };
  std_msgs::Header header;
  std::string ns;
  int id = 0, type = 0, action = 0;
  geometry_msgs::Pose pose;
  geometry_msgs::Vector3 scale;
  std_msgs::ColorRGBA color;
  ros::Duration lifetime;
  std::vector<geometry_msgs::Point> points;
};
}  // namespace visualization_msgs

namespace geometry_msgs {
struct TwistStamped { std_msgs::Header header; Twist twist; };
struct Accel { Vector3 linear, angular; };
}  // namespace geometry_msgs
namespace nav_msgs {
struct Path { std_msgs::Header header; std::vector<geometry_msgs::PoseStamped> poses; };
struct Odometry { std_msgs::Header header; };
}  // namespace nav_msgs
namespace base_local_planner {
class CostmapModel {
 public:
  virtual ~CostmapModel() {}
  virtual double footprintCost(double, double, double, const std::vector<geometry_msgs::Point>&, double = 0.0, double = 0.0) { return 0.0; }
};
}  // namespace base_local_planner

namespace teb_local_planner {
class TebLocalPlannerReconfigureConfig {};
struct TrajectoryPointMsg { geometry_msgs::Pose pose; geometry_msgs::Twist velocity; geometry_msgs::Twist acceleration; ros::Duration time_from_start; };
struct TrajectoryMsg { std_msgs::Header header; std::vector<TrajectoryPointMsg> trajectory; };
}
