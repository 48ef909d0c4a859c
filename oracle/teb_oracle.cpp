/*
 * teb_oracle.cpp — CPU ORACLE for the TEB hot path. TEST INFRASTRUCTURE ONLY (see teb_oracle.h).
 *
 * Restates, in dependency-free C++17 / fp64, the reference path
 *   TebOptimalPlanner::optimizeTEB            /root/reference/src/optimal_planner.cpp:182-231
 *   buildGraph / AddEdges*                    src/optimal_planner.cpp:323-366, 422-1021
 *   optimizeGraph / computeCurrentCost        src/optimal_planner.cpp:368-402, 1041-1094
 *   TimedElasticBand::autoResize              src/timed_elastic_band.cpp:227-286
 *   the 15 edge classes                       include/teb_local_planner/g2o_types/edge_*.h
 *   penalties / fast_sigmoid                  g2o_types/penalties.h:57-187, misc.h:95-98
 *   footprints / obstacles / distances        robot_footprint_model.h, obstacles.h, distance_calculations.h
 *   selectBestTeb                             src/homotopy_class_planner.cpp:564-667
 * plus the behaviour of the EXTERNAL libg2o (unpinned version, package.xml:41) that the reference drives:
 *   SparseOptimizer::optimize, OptimizationAlgorithmLevenberg::solve (tau=1e-5, <=10 trials),
 *   Base{Unary,Binary,Multi}Edge::linearizeOplus (central differences, delta=1e-9),
 *   constructQuadraticForm, BlockSolver + LinearSolverCSparse (exact sparse Cholesky).
 * g2o is not vendored in /root/reference; its published algorithm is restated here from the upstream
 * sources of that API generation. PARITY UNPINNED inside that library; everything of the reference's
 * own code is pinned bit-for-bit by tests/test_reference_pinning.py (oracle/_ref).
 *
 * Jacobian modes: TEB_AMD_JACOBIAN_G2O_NUMERIC = what the reference really does ("faithful");
 *                 TEB_AMD_JACOBIAN_ANALYTIC    = closed forms with the one-sided conventions of
 *                                                penalties.h:127-187 ("clean"), validated against the
 *                                                numeric mode in tests/test_oracle_jacobians.py.
 */
#include "teb_oracle.h"
#include "grid_costmap.h"

#include <cfloat>
#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------------
// small helpers (g2o/stuff/misc.h semantics)
// ------------------------------------------------------------------------------------------------
struct V2 { double x, y; };
inline V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
inline V2 operator-(V2 a, V2 b) { return {a.x - b.x, a.y - b.y}; }
inline V2 operator*(double s, V2 a) { return {s * a.x, s * a.y}; }
inline double dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
// Eigen cross product: (a1 b2 - a2 b1, a2 b0 - a0 b2, a0 b1 - a1 b0)
inline void cross3(const double* a, const double* b, double* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }
inline double sqnorm(V2 a) { return a.x * a.x + a.y * a.y; }
inline double norm(V2 a) { return std::sqrt(a.x * a.x + a.y * a.y); }

inline double normalize_theta(double theta) {
  if (theta >= -M_PI && theta < M_PI) return theta;
  double multiplier = std::floor(theta / (2 * M_PI));
  theta = theta - multiplier * 2 * M_PI;
  if (theta >= M_PI) theta -= 2 * M_PI;
  if (theta < -M_PI) theta += 2 * M_PI;
  return theta;
}
inline double sign(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0); }
inline double average_angle(double t1, double t2) {
  double x = std::cos(t1) + std::cos(t2);
  double y = std::sin(t1) + std::sin(t2);
  if (x == 0 && y == 0) return 0;
  return std::atan2(y, x);
}
// misc.h:95-98
inline double fast_sigmoid(double x) { return x / (1 + std::fabs(x)); }

// penalties.h:57-117
inline double penaltyBoundToInterval(double var, double a, double eps) {
  if (var < -a + eps) return (-var - (a - eps));
  if (var <= a - eps) return 0.;
  return (var - (a - eps));
}
inline double penaltyBoundToInterval(double var, double a, double b, double eps) {
  if (var < a + eps) return (-var + (a + eps));
  if (var <= b - eps) return 0.;
  return (var - (b - eps));
}
inline double penaltyBoundFromBelow(double var, double a, double eps) {
  if (var >= a + eps) return 0.;
  return (-var + (a + eps));
}
// penalties.h:127-187
inline double penaltyBoundToIntervalDerivative(double var, double a, double eps) {
  if (var < -a + eps) return -1;
  if (var <= a - eps) return 0.;
  return 1;
}
inline double penaltyBoundToIntervalDerivative(double var, double a, double b, double eps) {
  if (var < a + eps) return -1;
  if (var <= b - eps) return 0.;
  return 1;
}
inline double penaltyBoundFromBelowDerivative(double var, double a, double eps) {
  if (var >= a + eps) return 0.;
  return -1;
}

// ------------------------------------------------------------------------------------------------
// scene data
// ------------------------------------------------------------------------------------------------
struct Obst {
  int type;
  V2 a, b;
  double r;
  V2 vel;
  bool dyn;
  V2 c;  // centroid
  std::vector<V2> verts;
};

// PolygonObstacle::calcCentroid, src/obstacles.cpp:56-121
V2 polygon_centroid(const std::vector<V2>& v) {
  const int n = (int)v.size();
  if (n == 0) return {NAN, NAN};
  if (n == 1) return v[0];
  if (n == 2) return 0.5 * (v[0] + v[1]);
  V2 c{0, 0};
  double A = 0;
  for (int i = 0; i < n - 1; ++i) A += v[i].x * v[i + 1].y - v[i + 1].x * v[i].y;
  A += v[n - 1].x * v[0].y - v[0].x * v[n - 1].y;
  A *= 0.5;
  if (A != 0) {
    for (int i = 0; i < n - 1; ++i) {
      double aux = (v[i].x * v[i + 1].y - v[i + 1].x * v[i].y);
      c = c + aux * (v[i] + v[i + 1]);
    }
    double aux = (v[n - 1].x * v[0].y - v[0].x * v[n - 1].y);
    c = c + aux * (v[n - 1] + v[0]);
    c.x /= (6 * A);  // centroid_ /= (6*A): a division, not a multiplication by the reciprocal
    c.y /= (6 * A);
    return c;
  }
  int i_cand = 0, j_cand = 0;
  double max_dist = 0;
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) {
      double d = norm(v[j] - v[i]);
      if (d > max_dist) { max_dist = d; i_cand = i; j_cand = j; }
    }
  return 0.5 * (v[i_cand] + v[j_cand]);
}

struct Scene {
  teb_amd_config_t cfg;
  std::vector<Obst> obst;
  std::vector<V2> via;
  std::vector<V2> footprint;  // body-frame vertices (line: 2, polygon: k)
};

int load_scene(Scene& s, const teb_amd_config_t* cfg, const teb_amd_obstacles_t* o, int n_via,
               const double* vx, const double* vy) {
  if (!cfg) return TEB_AMD_ERR_INVALID_ARG;
  s.cfg = *cfg;
  s.obst.clear();
  if (o) {
    for (int i = 0; i < o->count; ++i) {
      Obst ob;
      ob.type = o->type[i];
      ob.a = {o->ax ? o->ax[i] : 0.0, o->ay ? o->ay[i] : 0.0};
      ob.b = {o->bx ? o->bx[i] : 0.0, o->by ? o->by[i] : 0.0};
      ob.r = o->radius ? o->radius[i] : 0.0;
      ob.vel = {o->vx ? o->vx[i] : 0.0, o->vy ? o->vy[i] : 0.0};
      ob.dyn = o->dynamic ? (o->dynamic[i] != 0) : false;
      switch (ob.type) {
        case TEB_AMD_OBST_POINT:
        case TEB_AMD_OBST_CIRCULAR: ob.c = ob.a; break;
        case TEB_AMD_OBST_LINE:
        case TEB_AMD_OBST_PILL: ob.c = 0.5 * (ob.a + ob.b); break;  // calcCentroid obstacles.h:742,892
        case TEB_AMD_OBST_POLYGON: {
          if (!o->vert_offset || !o->vert_x || !o->vert_y) return TEB_AMD_ERR_INVALID_ARG;
          for (int k = o->vert_offset[i]; k < o->vert_offset[i + 1]; ++k)
            ob.verts.push_back({o->vert_x[k], o->vert_y[k]});
          if (ob.verts.empty()) return TEB_AMD_ERR_INVALID_ARG;
          ob.c = polygon_centroid(ob.verts);
          break;
        }
        default: return TEB_AMD_ERR_INVALID_ARG;
      }
      s.obst.push_back(std::move(ob));
    }
  }
  s.via.clear();
  for (int i = 0; i < n_via; ++i) s.via.push_back({vx[i], vy[i]});
  s.footprint.clear();
  for (int i = 0; i < cfg->footprint_n_vertices && i < TEB_AMD_MAX_FOOTPRINT_VERTICES; ++i)
    s.footprint.push_back({cfg->footprint_vx[i], cfg->footprint_vy[i]});
  return TEB_AMD_OK;
}

// ------------------------------------------------------------------------------------------------
// distance_calculations.h:60-262, with witness points for the analytic gradients.
// A Witness holds the pair of closest points (p1 on the first argument, p2 on the second).
// ------------------------------------------------------------------------------------------------
struct Wit { double d; V2 p1, p2; };

// distance_calculations.h:60-76
inline V2 closest_point_on_line_segment_2d(V2 point, V2 ls, V2 le) {
  V2 diff = le - ls;
  double sq_norm = sqnorm(diff);
  if (sq_norm == 0) return ls;
  double u = ((point.x - ls.x) * diff.x + (point.y - ls.y) * diff.y) / sq_norm;
  if (u <= 0) return ls;
  else if (u >= 1) return le;
  return ls + u * diff;
}
// :85-88
inline Wit distance_point_to_segment_2d(V2 point, V2 ls, V2 le) {
  V2 c = closest_point_on_line_segment_2d(point, ls, le);
  return {norm(point - c), point, c};
}
// :99-130
inline bool check_line_segments_intersection_2d(V2 l1s, V2 l1e, V2 l2s, V2 l2e) {
  V2 line1 = l1e - l1s;
  V2 line2 = l2e - l2s;
  double denom = line1.x * line2.y - line2.x * line1.y;
  if (denom == 0) return false;
  bool denomPositive = denom > 0;
  V2 aux = l1s - l2s;
  double s_numer = line1.x * aux.y - line1.y * aux.x;
  if ((s_numer < 0) == denomPositive) return false;
  double t_numer = line2.x * aux.y - line2.y * aux.x;
  if ((t_numer < 0) == denomPositive) return false;
  if (((s_numer > denom) == denomPositive) || ((t_numer > denom) == denomPositive)) return false;
  return true;
}
// :140-157 (p1 on segment 1, p2 on segment 2)
inline Wit distance_segment_to_segment_2d(V2 l1s, V2 l1e, V2 l2s, V2 l2e) {
  if (check_line_segments_intersection_2d(l1s, l1e, l2s, l2e)) return {0.0, l1s, l1s};
  Wit w0 = distance_point_to_segment_2d(l1s, l2s, l2e);  // p1 = l1s (on seg1), p2 on seg2
  Wit w1 = distance_point_to_segment_2d(l1e, l2s, l2e);
  Wit w2 = distance_point_to_segment_2d(l2s, l1s, l1e);  // p1 = l2s (on seg2!), p2 on seg1
  Wit w3 = distance_point_to_segment_2d(l2e, l1s, l1e);
  // std::min_element: first smallest
  Wit best = w0;
  int bi = 0;
  if (w1.d < best.d) { best = w1; bi = 1; }
  if (w2.d < best.d) { best = w2; bi = 2; }
  if (w3.d < best.d) { best = w3; bi = 3; }
  if (bi >= 2) std::swap(best.p1, best.p2);  // re-order so p1 is on seg1, p2 on seg2
  return best;
}
// :168-193 (p1 = point, p2 on polygon)
inline Wit distance_point_to_polygon_2d(V2 point, const std::vector<V2>& v) {
  Wit best{HUGE_VAL, point, point};
  if (v.size() == 1) return {norm(point - v.front()), point, v.front()};
  for (int i = 0; i < (int)v.size() - 1; ++i) {
    Wit w = distance_point_to_segment_2d(point, v[i], v[i + 1]);
    if (w.d < best.d) best = w;
  }
  if (v.size() > 2) {
    Wit w = distance_point_to_segment_2d(point, v.back(), v.front());
    if (w.d < best.d) return w;
  }
  return best;
}
// :203-229 (p1 on segment, p2 on polygon)
inline Wit distance_segment_to_polygon_2d(V2 ls, V2 le, const std::vector<V2>& v) {
  Wit best{HUGE_VAL, ls, ls};
  if (v.size() == 1) {
    Wit w = distance_point_to_segment_2d(v.front(), ls, le);  // p1 = vertex, p2 on segment
    std::swap(w.p1, w.p2);
    return w;
  }
  for (int i = 0; i < (int)v.size() - 1; ++i) {
    Wit w = distance_segment_to_segment_2d(ls, le, v[i], v[i + 1]);
    if (w.d < best.d) best = w;
  }
  if (v.size() > 2) {
    Wit w = distance_segment_to_segment_2d(ls, le, v.back(), v.front());
    if (w.d < best.d) return w;
  }
  return best;
}
// :237-262 (p1 on polygon 1, p2 on polygon 2)
inline Wit distance_polygon_to_polygon_2d(const std::vector<V2>& v1, const std::vector<V2>& v2) {
  Wit best{HUGE_VAL, {0, 0}, {0, 0}};
  if (v1.size() == 1) return distance_point_to_polygon_2d(v1.front(), v2);
  for (int i = 0; i < (int)v1.size() - 1; ++i) {
    Wit w = distance_segment_to_polygon_2d(v1[i], v1[i + 1], v2);
    if (w.d < best.d) best = w;
  }
  if (v1.size() > 2) {
    Wit w = distance_segment_to_polygon_2d(v1.back(), v1.front(), v2);
    if (w.d < best.d) return w;
  }
  return best;
}

// ------------------------------------------------------------------------------------------------
// Obstacle::getMinimumDistance / getMinimumSpatioTemporalDistance (obstacles.h:358-397, 502-541,
// 653-695, 800-843, 968-1021). `off` = t * centroid_velocity (zero for the static variants).
// Returned witness: p1 on the ROBOT shape, p2 on the obstacle (before subtracting the radius).
// ------------------------------------------------------------------------------------------------
inline void shifted_verts(const Obst& o, V2 off, std::vector<V2>& out) {
  out.resize(o.verts.size());
  for (size_t i = 0; i < o.verts.size(); ++i) out[i] = o.verts[i] + off;  // predictVertices :1023-1031
}

Wit obst_dist_point(const Obst& o, V2 p, V2 off) {
  switch (o.type) {
    case TEB_AMD_OBST_POINT: {
      V2 q = o.a + off;
      // static: (position-pos_).norm(); spatio-temporal: (pos_ + t*v - position).norm(): same value
      return {norm(p - q), p, q};
    }
    case TEB_AMD_OBST_CIRCULAR: {
      V2 q = o.a + off;
      return {norm(p - q) - o.r, p, q};
    }
    case TEB_AMD_OBST_LINE: return distance_point_to_segment_2d(p, o.a + off, o.b + off);
    case TEB_AMD_OBST_PILL: {
      Wit w = distance_point_to_segment_2d(p, o.a + off, o.b + off);
      w.d -= o.r;
      return w;
    }
    default: {
      std::vector<V2> v;
      shifted_verts(o, off, v);
      return distance_point_to_polygon_2d(p, v);
    }
  }
}
Wit obst_dist_segment(const Obst& o, V2 ls, V2 le, V2 off) {
  switch (o.type) {
    case TEB_AMD_OBST_POINT: {
      Wit w = distance_point_to_segment_2d(o.a + off, ls, le);  // p1 = obstacle, p2 on robot segment
      std::swap(w.p1, w.p2);
      return w;
    }
    case TEB_AMD_OBST_CIRCULAR: {
      Wit w = distance_point_to_segment_2d(o.a + off, ls, le);
      std::swap(w.p1, w.p2);
      w.d -= o.r;
      return w;
    }
    case TEB_AMD_OBST_LINE: {
      Wit w = distance_segment_to_segment_2d(o.a + off, o.b + off, ls, le);  // seg1 = obstacle
      std::swap(w.p1, w.p2);
      return w;
    }
    case TEB_AMD_OBST_PILL: {
      Wit w = distance_segment_to_segment_2d(o.a + off, o.b + off, ls, le);
      std::swap(w.p1, w.p2);
      w.d -= o.r;
      return w;
    }
    default: {
      std::vector<V2> v;
      shifted_verts(o, off, v);
      return distance_segment_to_polygon_2d(ls, le, v);  // p1 on robot segment
    }
  }
}
Wit obst_dist_polygon(const Obst& o, const std::vector<V2>& poly, V2 off) {
  switch (o.type) {
    case TEB_AMD_OBST_POINT: {
      Wit w = distance_point_to_polygon_2d(o.a + off, poly);  // p1 = obstacle point
      std::swap(w.p1, w.p2);
      return w;
    }
    case TEB_AMD_OBST_CIRCULAR: {
      Wit w = distance_point_to_polygon_2d(o.a + off, poly);
      std::swap(w.p1, w.p2);
      w.d -= o.r;
      return w;
    }
    case TEB_AMD_OBST_LINE: {
      Wit w = distance_segment_to_polygon_2d(o.a + off, o.b + off, poly);  // p1 on obstacle segment
      std::swap(w.p1, w.p2);
      return w;
    }
    case TEB_AMD_OBST_PILL: {
      Wit w = distance_segment_to_polygon_2d(o.a + off, o.b + off, poly);
      std::swap(w.p1, w.p2);
      w.d -= o.r;
      return w;
    }
    default: {
      std::vector<V2> v;
      shifted_verts(o, off, v);
      return distance_polygon_to_polygon_2d(poly, v);  // robot polygon first (obstacles.h:979-982)
    }
  }
}

// ------------------------------------------------------------------------------------------------
// BaseRobotFootprintModel::calculateDistance / estimateSpatioTemporalDistance
// (robot_footprint_model.h:160-175, 263-278, 351-372, 496-517, 664-683).
// grad (optional) = analytic d(dist)/d(x,y,theta): unit vector from the obstacle witness to the robot
// witness, theta through the lever arm of the robot witness; 0 where the distance is not differentiable
// (coincident witnesses / intersecting shapes).
// ------------------------------------------------------------------------------------------------
double footprint_distance(const Scene& s, double x, double y, double th, const Obst& o, bool st, double t,
                          double* grad) {
  const teb_amd_config_t& c = s.cfg;
  V2 off = st ? t * o.vel : V2{0, 0};
  V2 pos{x, y};
  Wit w;
  double sub = 0;
  switch (c.footprint_type) {
    case TEB_AMD_FOOTPRINT_POINT: w = obst_dist_point(o, pos, off); break;
    case TEB_AMD_FOOTPRINT_CIRCULAR:
      w = obst_dist_point(o, pos, off);
      sub = c.footprint_radius;
      break;
    case TEB_AMD_FOOTPRINT_TWO_CIRCLES: {
      V2 dir{std::cos(th), std::sin(th)};  // orientationUnitVec pose_se2.h:215-218
      Wit wf = obst_dist_point(o, pos + c.footprint_front_offset * dir, off);
      Wit wr = obst_dist_point(o, pos - c.footprint_rear_offset * dir, off);
      double df = wf.d - c.footprint_front_radius;
      double dr = wr.d - c.footprint_rear_radius;
      // std::min(dist_front, dist_rear): returns rear only if rear < front
      if (dr < df) { w = wr; sub = c.footprint_rear_radius; }
      else { w = wf; sub = c.footprint_front_radius; }
      break;
    }
    case TEB_AMD_FOOTPRINT_LINE: {
      double cs = std::cos(th), sn = std::sin(th);
      V2 a = s.footprint[0], b = s.footprint[1];
      V2 ws{x + cs * a.x - sn * a.y, y + sn * a.x + cs * a.y};
      V2 we{x + cs * b.x - sn * b.y, y + sn * b.x + cs * b.y};
      w = obst_dist_segment(o, ws, we, off);
      break;
    }
    default: {
      double cs = std::cos(th), sn = std::sin(th);
      std::vector<V2> poly(s.footprint.size());
      for (size_t i = 0; i < poly.size(); ++i) {
        poly[i].x = x + cs * s.footprint[i].x - sn * s.footprint[i].y;
        poly[i].y = y + sn * s.footprint[i].x + cs * s.footprint[i].y;
      }
      w = obst_dist_polygon(o, poly, off);
      break;
    }
  }
  if (grad) {
    V2 dv = w.p1 - w.p2;
    double dn = norm(dv);
    if (dn > 0) {
      V2 nrm{dv.x / dn, dv.y / dn};
      V2 lever = w.p1 - pos;
      grad[0] = nrm.x;
      grad[1] = nrm.y;
      grad[2] = -nrm.x * lever.y + nrm.y * lever.x;
    } else {
      grad[0] = grad[1] = grad[2] = 0;
    }
  }
  return w.d - sub;
}

// ------------------------------------------------------------------------------------------------
// TimedElasticBand state strip
// ------------------------------------------------------------------------------------------------
struct Teb {
  std::vector<double> x, y, th, dt;
  bool has_vs = false, has_vg = false;
  double vs[3] = {0, 0, 0}, vg[3] = {0, 0, 0};
  int rotdir = TEB_AMD_ROT_NONE;
  bool via_enabled = true;
  int n() const { return (int)x.size(); }
};

// src/timed_elastic_band.cpp:227-286 (literal: vector insert/erase, i-- re-checks)
void auto_resize(Teb& t, double dt_ref, double dt_hyst, int min_samples, int max_samples, bool fast_mode) {
  bool modified = true;
  for (int rep = 0; rep < 100 && modified; ++rep) {
    modified = false;
    for (int i = 0; i < (int)t.dt.size(); ++i) {
      if (t.dt[i] > dt_ref + dt_hyst && (int)t.dt.size() < max_samples) {
        if (t.dt[i] > 2 * dt_ref) {
          double newtime = 0.5 * t.dt[i];
          t.dt[i] = newtime;
          // PoseSE2::average pose_se2.h:266-269
          double ax = (t.x[i] + t.x[i + 1]) / 2, ay = (t.y[i] + t.y[i + 1]) / 2;
          double ath = average_angle(t.th[i], t.th[i + 1]);
          t.x.insert(t.x.begin() + i + 1, ax);
          t.y.insert(t.y.begin() + i + 1, ay);
          t.th.insert(t.th.begin() + i + 1, ath);
          t.dt.insert(t.dt.begin() + i + 1, newtime);
          i--;
          modified = true;
        } else {
          if (i < (int)t.dt.size() - 1) t.dt[i + 1] += t.dt[i] - dt_ref;
          t.dt[i] = dt_ref;
        }
      } else if (t.dt[i] < dt_ref - dt_hyst && (int)t.dt.size() > min_samples) {
        if (i < ((int)t.dt.size() - 1)) {
          t.dt[i + 1] = t.dt[i + 1] + t.dt[i];
          t.dt.erase(t.dt.begin() + i);
          t.x.erase(t.x.begin() + i + 1);
          t.y.erase(t.y.begin() + i + 1);
          t.th.erase(t.th.begin() + i + 1);
          i--;
        } else {
          t.dt[i - 1] += t.dt[i];
          t.dt.erase(t.dt.begin() + i);
          t.x.erase(t.x.begin() + i);
          t.y.erase(t.y.begin() + i);
          t.th.erase(t.th.begin() + i);
        }
        modified = true;
      }
    }
    if (fast_mode) break;
  }
}

// ------------------------------------------------------------------------------------------------
// hyper-graph: edges in g2o insertion order
// ------------------------------------------------------------------------------------------------
enum EType {
  E_OBST, E_INFL, E_DYN, E_VIA, E_VEL, E_VEL_HOLO, E_ACC, E_ACC_START, E_ACC_GOAL, E_ACC_HOLO,
  E_ACC_HOLO_START, E_ACC_HOLO_GOAL, E_TIME, E_SHORTEST, E_KIN_DD, E_KIN_CL, E_ROTDIR, E_VOR
};
enum Cat { CAT_OBST = 0, CAT_VIA = 1, CAT_TIME = 2, CAT_OTHER = 3 };

struct Edge {
  int type;
  int np = 0, nd = 0;   // #pose vertices, #timediff vertices (g2o order: poses first, then timediffs)
  int pose[3] = {0, 0, 0};
  int dts[2] = {0, 0};
  int dim = 1;
  double info[3] = {0, 0, 0};  // diagonal information
  int obst = -1;
  int via = -1;
  double t = 0;          // EdgeDynamicObstacle::t_
  double dir = 1;        // EdgePreferRotDir::_measurement
  double err[3] = {0, 0, 0};
  int cat() const {
    if (type == E_OBST || type == E_INFL || type == E_DYN) return CAT_OBST;
    if (type == E_VIA) return CAT_VIA;
    if (type == E_TIME) return CAT_TIME;
    return CAT_OTHER;
  }
};

// columns of the edge Jacobian: pose0 xyz (0-2), pose1 (3-5), pose2 (6-8), dt0 (9), dt1 (10)
struct Jac { double j[3][11]; };

struct SignedVel {  // v = dist/dt * fast_sigmoid(100 * deltaS . (cos th_a, sin th_a)), omega = angle_diff/dt
  double v, omega, dist, angle_diff;
  // derivatives of v wrt (xa, ya, tha, xb, yb, thb, dt)
  double dv[7];
};

inline void signed_velocity(const teb_amd_config_t& c, double xa, double ya, double tha, double xb, double yb,
                            double thb, double dt, SignedVel& o, bool want_grad) {
  double dx = xb - xa, dy = yb - ya;
  double dist = std::sqrt(dx * dx + dy * dy);  // deltaS.norm()
  const double eucl = dist;
  const double angle_diff = normalize_theta(thb - tha);
  double k = 1.0, kp = 0.0;
  bool arc = false;
  if (c.exact_arc_length && angle_diff != 0) {
    double radius = dist / (2 * std::sin(angle_diff / 2));
    dist = std::fabs(angle_diff * radius);
    arc = true;
  }
  double ca = std::cos(tha), sa = std::sin(tha);
  double p = dx * ca + dy * sa;
  double sg = fast_sigmoid(100 * p);
  double vel = dist / dt;
  vel *= sg;
  o.v = vel;
  o.omega = angle_diff / dt;
  o.dist = dist;
  o.angle_diff = angle_diff;
  if (!want_grad) return;
  if (arc) {
    double h = angle_diff / 2, sh = std::sin(h), ch = std::cos(h);
    k = angle_diff / (2 * sh);
    kp = 1.0 / (2 * sh) - angle_diff * ch / (4 * sh * sh);
    if (k < 0) { k = -k; kp = -kp; }  // fabs
  }
  // d(dist)/d(...)
  double ddx = 0, ddy = 0;
  if (eucl > 0) { ddx = dx / eucl; ddy = dy / eucl; }
  double dD[7] = {-k * ddx, -k * ddy, -eucl * kp, k * ddx, k * ddy, eucl * kp, 0};
  double a100 = 1 + std::fabs(100 * p);
  double sp = 100.0 / (a100 * a100);  // d sigmoid / dp
  double dP[7] = {-ca, -sa, -dx * sa + dy * ca, ca, sa, 0, 0};
  for (int q = 0; q < 6; ++q) o.dv[q] = (dD[q] * sg + dist * sp * dP[q]) / dt;
  o.dv[6] = -vel / dt;
}

struct Graph {
  const Scene* s = nullptr;
  Teb* teb = nullptr;
  std::vector<Edge> edges;
  int n = 0;  // #poses
  int N = 0;  // #free scalars (g2o index space)
  bool pose_fixed(int i) const { return i == 0 || i == n - 1; }
  int idx_pose(int i, int c) const { return pose_fixed(i) ? -1 : 4 * i - 3 + c; }
  int idx_dt(int i) const { return 4 * i; }
};

// ---- computeError of every edge class (file:line in each case) ----------------
void compute_error(const Graph& g, Edge& e) {
  const Scene& s = *g.s;
  const teb_amd_config_t& c = s.cfg;
  const Teb& t = *g.teb;
  switch (e.type) {
    case E_OBST:
    case E_INFL: {  // edge_obstacle.h:90-103, 212-229
      int i = e.pose[0];
      double dist = footprint_distance(s, t.x[i], t.y[i], t.th[i], s.obst[e.obst], false, 0, nullptr);
      e.err[0] = penaltyBoundFromBelow(dist, c.min_obstacle_dist, c.penalty_epsilon);
      if (c.obstacle_cost_exponent != 1.0 && c.min_obstacle_dist > 0.0)
        e.err[0] = c.min_obstacle_dist * std::pow(e.err[0] / c.min_obstacle_dist, c.obstacle_cost_exponent);
      if (e.type == E_INFL) e.err[1] = penaltyBoundFromBelow(dist, c.inflation_dist, 0.0);
      break;
    }
    case E_DYN: {  // edge_dynamic_obstacle.h:93-104
      int i = e.pose[0];
      double dist = footprint_distance(s, t.x[i], t.y[i], t.th[i], s.obst[e.obst], true, e.t, nullptr);
      e.err[0] = penaltyBoundFromBelow(dist, c.min_obstacle_dist, c.penalty_epsilon);
      e.err[1] = penaltyBoundFromBelow(dist, c.dynamic_obstacle_inflation_dist, 0.0);
      break;
    }
    case E_VIA: {  // edge_via_point.h:81-89
      int i = e.pose[0];
      e.err[0] = norm(V2{t.x[i], t.y[i]} - s.via[e.via]);
      break;
    }
    case E_VEL: {  // edge_velocity.h:85-117
      int a = e.pose[0], b = e.pose[1];
      SignedVel sv;
      signed_velocity(c, t.x[a], t.y[a], t.th[a], t.x[b], t.y[b], t.th[b], t.dt[e.dts[0]], sv, false);
      e.err[0] = penaltyBoundToInterval(sv.v, -c.max_vel_x_backwards, c.max_vel_x, c.penalty_epsilon);
      e.err[1] = penaltyBoundToInterval(sv.omega, c.max_vel_theta, c.penalty_epsilon);
      break;
    }
    case E_VEL_HOLO: {  // edge_velocity.h:232-271
      int a = e.pose[0], b = e.pose[1];
      double dtv = t.dt[e.dts[0]];
      double dx = t.x[b] - t.x[a], dy = t.y[b] - t.y[a];
      double c1 = std::cos(t.th[a]), s1 = std::sin(t.th[a]);
      double r_dx = c1 * dx + s1 * dy;
      double r_dy = -s1 * dx + c1 * dy;
      double vx = r_dx / dtv, vy = r_dy / dtv;
      double omega = normalize_theta(t.th[b] - t.th[a]) / dtv;
      double rem_y = std::sqrt(std::max(0.0, c.max_vel_trans * c.max_vel_trans - vx * vx));
      double rem_x = std::sqrt(std::max(0.0, c.max_vel_trans * c.max_vel_trans - vy * vy));
      double max_vel_y = std::min(rem_y, c.max_vel_y);
      double max_vel_x = std::min(rem_x, c.max_vel_x);
      double max_vel_x_backwards = std::min(rem_x, c.max_vel_x_backwards);
      e.err[0] = penaltyBoundToInterval(vx, -max_vel_x_backwards, max_vel_x, 0.0);
      e.err[1] = penaltyBoundToInterval(vy, max_vel_y, 0.0);
      e.err[2] = penaltyBoundToInterval(omega, c.max_vel_theta, c.penalty_epsilon);
      break;
    }
    case E_ACC: {  // edge_acceleration.h:88-149
      int p1 = e.pose[0], p2 = e.pose[1], p3 = e.pose[2];
      double dt1 = t.dt[e.dts[0]], dt2 = t.dt[e.dts[1]];
      SignedVel v1, v2;
      signed_velocity(c, t.x[p1], t.y[p1], t.th[p1], t.x[p2], t.y[p2], t.th[p2], dt1, v1, false);
      signed_velocity(c, t.x[p2], t.y[p2], t.th[p2], t.x[p3], t.y[p3], t.th[p3], dt2, v2, false);
      const double acc_lin = (v2.v - v1.v) * 2 / (dt1 + dt2);
      e.err[0] = penaltyBoundToInterval(acc_lin, c.acc_lim_x, c.penalty_epsilon);
      const double acc_rot = (v2.omega - v1.omega) * 2 / (dt1 + dt2);
      e.err[1] = penaltyBoundToInterval(acc_rot, c.acc_lim_theta, c.penalty_epsilon);
      break;
    }
    case E_ACC_START: {  // edge_acceleration.h:303-345
      int p1 = e.pose[0], p2 = e.pose[1];
      double dtv = t.dt[e.dts[0]];
      SignedVel v2;
      signed_velocity(c, t.x[p1], t.y[p1], t.th[p1], t.x[p2], t.y[p2], t.th[p2], dtv, v2, false);
      const double acc_lin = (v2.v - t.vs[0]) / dtv;
      e.err[0] = penaltyBoundToInterval(acc_lin, c.acc_lim_x, c.penalty_epsilon);
      const double acc_rot = (v2.omega - t.vs[2]) / dtv;
      e.err[1] = penaltyBoundToInterval(acc_rot, c.acc_lim_theta, c.penalty_epsilon);
      break;
    }
    case E_ACC_GOAL: {  // edge_acceleration.h:394-437
      int p1 = e.pose[0], p2 = e.pose[1];
      double dtv = t.dt[e.dts[0]];
      SignedVel v1;
      signed_velocity(c, t.x[p1], t.y[p1], t.th[p1], t.x[p2], t.y[p2], t.th[p2], dtv, v1, false);
      const double acc_lin = (t.vg[0] - v1.v) / dtv;
      e.err[0] = penaltyBoundToInterval(acc_lin, c.acc_lim_x, c.penalty_epsilon);
      const double acc_rot = (t.vg[2] - v1.omega) / dtv;
      e.err[1] = penaltyBoundToInterval(acc_rot, c.acc_lim_theta, c.penalty_epsilon);
      break;
    }
    case E_ACC_HOLO: {  // edge_acceleration.h:484-534
      int p1 = e.pose[0], p2 = e.pose[1], p3 = e.pose[2];
      double dt1 = t.dt[e.dts[0]], dt2 = t.dt[e.dts[1]];
      double d1x = t.x[p2] - t.x[p1], d1y = t.y[p2] - t.y[p1];
      double d2x = t.x[p3] - t.x[p2], d2y = t.y[p3] - t.y[p2];
      double c1 = std::cos(t.th[p1]), s1 = std::sin(t.th[p1]);
      double c2 = std::cos(t.th[p2]), s2 = std::sin(t.th[p2]);
      double p1_dx = c1 * d1x + s1 * d1y, p1_dy = -s1 * d1x + c1 * d1y;
      double p2_dx = c2 * d2x + s2 * d2y, p2_dy = -s2 * d2x + c2 * d2y;
      double vel1_x = p1_dx / dt1, vel1_y = p1_dy / dt1;
      double vel2_x = p2_dx / dt2, vel2_y = p2_dy / dt2;
      double dt12 = dt1 + dt2;
      double acc_x = (vel2_x - vel1_x) * 2 / dt12;
      double acc_y = (vel2_y - vel1_y) * 2 / dt12;
      e.err[0] = penaltyBoundToInterval(acc_x, c.acc_lim_x, c.penalty_epsilon);
      e.err[1] = penaltyBoundToInterval(acc_y, c.acc_lim_y, c.penalty_epsilon);
      double omega1 = normalize_theta(t.th[p2] - t.th[p1]) / dt1;
      double omega2 = normalize_theta(t.th[p3] - t.th[p2]) / dt2;
      double acc_rot = (omega2 - omega1) * 2 / dt12;
      e.err[2] = penaltyBoundToInterval(acc_rot, c.acc_lim_theta, c.penalty_epsilon);
      break;
    }
    case E_ACC_HOLO_START:
    case E_ACC_HOLO_GOAL: {  // edge_acceleration.h:577-620, 668-712
      int p1 = e.pose[0], p2 = e.pose[1];
      double dtv = t.dt[e.dts[0]];
      double dx = t.x[p2] - t.x[p1], dy = t.y[p2] - t.y[p1];
      double c1 = std::cos(t.th[p1]), s1 = std::sin(t.th[p1]);
      double pdx = c1 * dx + s1 * dy, pdy = -s1 * dx + c1 * dy;
      double om = normalize_theta(t.th[p2] - t.th[p1]) / dtv;
      double ax, ay, ar;
      if (e.type == E_ACC_HOLO_START) {
        ax = (pdx / dtv - t.vs[0]) / dtv;
        ay = (pdy / dtv - t.vs[1]) / dtv;
        ar = (om - t.vs[2]) / dtv;
      } else {
        ax = (t.vg[0] - pdx / dtv) / dtv;
        ay = (t.vg[1] - pdy / dtv) / dtv;
        ar = (t.vg[2] - om) / dtv;
      }
      e.err[0] = penaltyBoundToInterval(ax, c.acc_lim_x, c.penalty_epsilon);
      e.err[1] = penaltyBoundToInterval(ay, c.acc_lim_y, c.penalty_epsilon);
      e.err[2] = penaltyBoundToInterval(ar, c.acc_lim_theta, c.penalty_epsilon);
      break;
    }
    case E_TIME: e.err[0] = t.dt[e.dts[0]]; break;  // edge_time_optimal.h:93
    case E_SHORTEST: {  // edge_shortest_path.h:78
      int a = e.pose[0], b = e.pose[1];
      e.err[0] = norm(V2{t.x[b], t.y[b]} - V2{t.x[a], t.y[a]});
      break;
    }
    case E_KIN_DD: {  // edge_kinematics.h:87-103
      int a = e.pose[0], b = e.pose[1];
      double dx = t.x[b] - t.x[a], dy = t.y[b] - t.y[a];
      e.err[0] = std::fabs((std::cos(t.th[a]) + std::cos(t.th[b])) * dy - (std::sin(t.th[a]) + std::sin(t.th[b])) * dx);
      double ax = std::cos(t.th[a]), ay = std::sin(t.th[a]);
      e.err[1] = penaltyBoundFromBelow(dx * ax + dy * ay, 0, 0);
      break;
    }
    case E_KIN_CL: {  // edge_kinematics.h:196-216
      int a = e.pose[0], b = e.pose[1];
      double dx = t.x[b] - t.x[a], dy = t.y[b] - t.y[a];
      e.err[0] = std::fabs((std::cos(t.th[a]) + std::cos(t.th[b])) * dy - (std::sin(t.th[a]) + std::sin(t.th[b])) * dx);
      double angle_diff = normalize_theta(t.th[b] - t.th[a]);
      double nrm = std::sqrt(dx * dx + dy * dy);
      if (angle_diff == 0) e.err[1] = 0;
      else if (c.exact_arc_length)
        e.err[1] = penaltyBoundFromBelow(std::fabs(nrm / (2 * std::sin(angle_diff / 2))), c.min_turning_radius, 0.0);
      else
        e.err[1] = penaltyBoundFromBelow(nrm / std::fabs(angle_diff), c.min_turning_radius, 0.0);
      break;
    }
    case E_ROTDIR: {  // edge_prefer_rotdir.h:95-103
      int a = e.pose[0], b = e.pose[1];
      e.err[0] = penaltyBoundFromBelow(e.dir * normalize_theta(t.th[b] - t.th[a]), 0, 0);
      break;
    }
    case E_VOR: {  // edge_velocity_obstacle_ratio.h:79-121
      int a = e.pose[0], b = e.pose[1];
      SignedVel sv;
      signed_velocity(c, t.x[a], t.y[a], t.th[a], t.x[b], t.y[b], t.th[b], t.dt[e.dts[0]], sv, false);
      double dobs = footprint_distance(s, t.x[a], t.y[a], t.th[a], s.obst[e.obst], false, 0, nullptr);
      double ratio;
      if (dobs < c.obstacle_proximity_lower_bound) ratio = 0;
      else if (dobs > c.obstacle_proximity_upper_bound) ratio = 1;
      else ratio = (dobs - c.obstacle_proximity_lower_bound) / (c.obstacle_proximity_upper_bound - c.obstacle_proximity_lower_bound);
      ratio *= c.obstacle_proximity_ratio_max_vel;
      const double max_vel_fwd = ratio * c.max_vel_x;
      const double max_omega = ratio * c.max_vel_theta;
      e.err[0] = penaltyBoundToInterval(sv.v, max_vel_fwd, 0);
      e.err[1] = penaltyBoundToInterval(sv.omega, max_omega, 0);
      break;
    }
  }
}

// ---- numeric Jacobian: g2o Base{Unary,Binary,Multi}Edge::linearizeOplus ------
void linearize_numeric(Graph& g, Edge& e, Jac& J) {
  Teb& t = *g.teb;
  const double delta = 1e-9;
  const double scalar = 1.0 / (2 * delta);
  double errorBeforeNumeric[3] = {e.err[0], e.err[1], e.err[2]};
  std::memset(&J, 0, sizeof(J));
  for (int v = 0; v < e.np; ++v) {
    int i = e.pose[v];
    if (g.pose_fixed(i)) continue;
    for (int d = 0; d < 3; ++d) {
      double bx = t.x[i], by = t.y[i], bth = t.th[i];  // push
      double add[3] = {0, 0, 0};
      add[d] = delta;
      t.x[i] += add[0]; t.y[i] += add[1]; t.th[i] = normalize_theta(t.th[i] + add[2]);  // oplus, pose_se2.h:238-243
      compute_error(g, e);
      double e1[3] = {e.err[0], e.err[1], e.err[2]};
      t.x[i] = bx; t.y[i] = by; t.th[i] = bth;  // pop
      add[d] = -delta;
      t.x[i] += add[0]; t.y[i] += add[1]; t.th[i] = normalize_theta(t.th[i] + add[2]);
      compute_error(g, e);
      t.x[i] = bx; t.y[i] = by; t.th[i] = bth;
      for (int k = 0; k < e.dim; ++k) J.j[k][3 * v + d] = scalar * (e1[k] - e.err[k]);
    }
  }
  for (int v = 0; v < e.nd; ++v) {
    int i = e.dts[v];
    double b = t.dt[i];
    t.dt[i] += delta;
    compute_error(g, e);
    double e1[3] = {e.err[0], e.err[1], e.err[2]};
    t.dt[i] = b;
    t.dt[i] += -delta;
    compute_error(g, e);
    t.dt[i] = b;
    for (int k = 0; k < e.dim; ++k) J.j[k][9 + v] = scalar * (e1[k] - e.err[k]);
  }
  e.err[0] = errorBeforeNumeric[0]; e.err[1] = errorBeforeNumeric[1]; e.err[2] = errorBeforeNumeric[2];
}

// ---- analytic Jacobians ("clean" mode) -------------------------------------------------------------
// Conventions: penalty derivatives exactly as penalties.h:127-187; d||v||/dv := 0 at v = 0;
// normalize_theta treated as identity (derivative 1); fabs'(0) := g2o::sign(0) = 0.
// central_difference_kink (car-like edge only): the reference has a live analytic Jacobian for EdgeKinematicsDiffDrive only;
// EdgeKinematicsCarlike (edge_kinematics.h:182-230) is differentiated by g2o's central differences, delta = 1e-9. For f = |x| those return
// sign(x) g only while |x| >= |g| delta; closer to the kink the two samples straddle it and the quotient is sign(g) x / delta - next to
// nothing. That is not a corner case: a straight stretch of the initial band (a plan initialised from a line, the inflection points of a
// curve) has x = 0 up to rounding, 1e-17, whose sign is noise - closed forms then carry the full nonholonomic constraint of that segment
// where the reference's linearisation has none, and the two part at the first LM step (profiles/analytic_margin_r06.txt: 7 of 20 car-like
// scenes ended 4 .. 150 x T3 apart). The closed-form mode therefore reproduces the quotient of the central differences at this one kink.
void kin_nh_row(const Teb& t, int a, int b, double* row /*6: pose a xyz, pose b xyz*/, bool central_difference_kink = false) {
  // edge_kinematics.h:112-149 (live analytic Jacobian of the reference)
  double dx = t.x[b] - t.x[a], dy = t.y[b] - t.y[a];
  double cos1 = std::cos(t.th[a]), cos2 = std::cos(t.th[b]);
  double sin1 = std::sin(t.th[a]), sin2 = std::sin(t.th[b]);
  double aux1 = sin1 + sin2, aux2 = cos1 + cos2;
  double dd_error_1 = dx * cos1, dd_error_2 = dy * sin1;
  double dev_nh_abs = sign((std::cos(t.th[a]) + std::cos(t.th[b])) * dy - (std::sin(t.th[a]) + std::sin(t.th[b])) * dx);
  row[0] = aux1 * dev_nh_abs;
  row[1] = -aux2 * dev_nh_abs;
  row[2] = (-dd_error_2 - dd_error_1) * dev_nh_abs;
  row[3] = -aux1 * dev_nh_abs;
  row[4] = aux2 * dev_nh_abs;
  row[5] = (-sin2 * dy - cos2 * dx) * dev_nh_abs;
  if (central_difference_kink) {
    const double val = aux2 * dy - aux1 * dx;
    const double g[6] = {aux1, -aux2, -dd_error_2 - dd_error_1, -aux1, aux2, -sin2 * dy - cos2 * dx};
    for (int q = 0; q < 6; ++q)
      row[q] = std::fabs(val) >= std::fabs(g[q]) * 1e-9 ? g[q] * sign(val) : sign(g[q]) * val * 1e9;
  }
}

void linearize_analytic(Graph& g, Edge& e, Jac& J) {
  const Scene& s = *g.s;
  const teb_amd_config_t& c = s.cfg;
  const Teb& t = *g.teb;
  std::memset(&J, 0, sizeof(J));
  switch (e.type) {
    case E_OBST:
    case E_INFL:
    case E_DYN: {
      int i = e.pose[0];
      double gr[3];
      bool st = (e.type == E_DYN);
      double dist = footprint_distance(s, t.x[i], t.y[i], t.th[i], s.obst[e.obst], st, e.t, gr);
      double d0 = penaltyBoundFromBelowDerivative(dist, c.min_obstacle_dist, c.penalty_epsilon);
      if (e.type != E_DYN && c.obstacle_cost_exponent != 1.0 && c.min_obstacle_dist > 0.0) {
        double lin = penaltyBoundFromBelow(dist, c.min_obstacle_dist, c.penalty_epsilon);
        if (lin > 0)
          d0 *= c.obstacle_cost_exponent * std::pow(lin / c.min_obstacle_dist, c.obstacle_cost_exponent - 1.0);
        else
          d0 = 0;
      }
      for (int q = 0; q < 3; ++q) J.j[0][q] = d0 * gr[q];
      if (e.dim == 2) {
        double bound = (e.type == E_DYN) ? c.dynamic_obstacle_inflation_dist : c.inflation_dist;
        double d1 = penaltyBoundFromBelowDerivative(dist, bound, 0.0);
        for (int q = 0; q < 3; ++q) J.j[1][q] = d1 * gr[q];
      }
      break;
    }
    case E_VIA: {
      int i = e.pose[0];
      V2 d = V2{t.x[i], t.y[i]} - s.via[e.via];
      double nn = norm(d);
      if (nn > 0) { J.j[0][0] = d.x / nn; J.j[0][1] = d.y / nn; }
      break;
    }
    case E_VEL: {
      int a = e.pose[0], b = e.pose[1];
      double dtv = t.dt[e.dts[0]];
      SignedVel sv;
      signed_velocity(c, t.x[a], t.y[a], t.th[a], t.x[b], t.y[b], t.th[b], dtv, sv, true);
      double dv = penaltyBoundToIntervalDerivative(sv.v, -c.max_vel_x_backwards, c.max_vel_x, c.penalty_epsilon);
      double dw = penaltyBoundToIntervalDerivative(sv.omega, c.max_vel_theta, c.penalty_epsilon);
      for (int q = 0; q < 6; ++q) J.j[0][q] = dv * sv.dv[q];
      J.j[0][9] = dv * sv.dv[6];
      J.j[1][2] = -dw / dtv;
      J.j[1][5] = dw / dtv;
      J.j[1][9] = -dw * sv.omega / dtv;
      break;
    }
    case E_ACC: {
      int p1 = e.pose[0], p2 = e.pose[1], p3 = e.pose[2];
      double dt1 = t.dt[e.dts[0]], dt2 = t.dt[e.dts[1]];
      SignedVel v1, v2;
      signed_velocity(c, t.x[p1], t.y[p1], t.th[p1], t.x[p2], t.y[p2], t.th[p2], dt1, v1, true);
      signed_velocity(c, t.x[p2], t.y[p2], t.th[p2], t.x[p3], t.y[p3], t.th[p3], dt2, v2, true);
      double T = dt1 + dt2;
      double acc_lin = (v2.v - v1.v) * 2 / T;
      double acc_rot = (v2.omega - v1.omega) * 2 / T;
      double da = penaltyBoundToIntervalDerivative(acc_lin, c.acc_lim_x, c.penalty_epsilon);
      double dr = penaltyBoundToIntervalDerivative(acc_rot, c.acc_lim_theta, c.penalty_epsilon);
      double f = 2 / T;
      // v1 depends on pose1 (cols 0-2), pose2 (3-5), dt1 (9); v2 on pose2 (3-5), pose3 (6-8), dt2 (10)
      for (int q = 0; q < 6; ++q) J.j[0][q] += da * f * (-v1.dv[q]);
      for (int q = 0; q < 6; ++q) J.j[0][3 + q] += da * f * (v2.dv[q]);
      J.j[0][9] = da * (f * (-v1.dv[6]) - acc_lin / T);
      J.j[0][10] = da * (f * (v2.dv[6]) - acc_lin / T);
      J.j[1][2] = dr * f * (1.0 / dt1);
      J.j[1][5] = dr * f * (-1.0 / dt2 - 1.0 / dt1);
      J.j[1][8] = dr * f * (1.0 / dt2);
      J.j[1][9] = dr * (f * (v1.omega / dt1) - acc_rot / T);
      J.j[1][10] = dr * (f * (-v2.omega / dt2) - acc_rot / T);
      break;
    }
    case E_ACC_START: {
      int p1 = e.pose[0], p2 = e.pose[1];
      double dtv = t.dt[e.dts[0]];
      SignedVel v2;
      signed_velocity(c, t.x[p1], t.y[p1], t.th[p1], t.x[p2], t.y[p2], t.th[p2], dtv, v2, true);
      double acc_lin = (v2.v - t.vs[0]) / dtv;
      double acc_rot = (v2.omega - t.vs[2]) / dtv;
      double da = penaltyBoundToIntervalDerivative(acc_lin, c.acc_lim_x, c.penalty_epsilon);
      double dr = penaltyBoundToIntervalDerivative(acc_rot, c.acc_lim_theta, c.penalty_epsilon);
      for (int q = 0; q < 6; ++q) J.j[0][q] = da * v2.dv[q] / dtv;
      J.j[0][9] = da * (v2.dv[6] / dtv - acc_lin / dtv);
      J.j[1][2] = dr * (-1.0 / (dtv * dtv));
      J.j[1][5] = dr * (1.0 / (dtv * dtv));
      J.j[1][9] = dr * ((-v2.omega / dtv) / dtv - acc_rot / dtv);
      break;
    }
    case E_ACC_GOAL: {
      int p1 = e.pose[0], p2 = e.pose[1];
      double dtv = t.dt[e.dts[0]];
      SignedVel v1;
      signed_velocity(c, t.x[p1], t.y[p1], t.th[p1], t.x[p2], t.y[p2], t.th[p2], dtv, v1, true);
      double acc_lin = (t.vg[0] - v1.v) / dtv;
      double acc_rot = (t.vg[2] - v1.omega) / dtv;
      double da = penaltyBoundToIntervalDerivative(acc_lin, c.acc_lim_x, c.penalty_epsilon);
      double dr = penaltyBoundToIntervalDerivative(acc_rot, c.acc_lim_theta, c.penalty_epsilon);
      for (int q = 0; q < 6; ++q) J.j[0][q] = -da * v1.dv[q] / dtv;
      J.j[0][9] = da * (-v1.dv[6] / dtv - acc_lin / dtv);
      J.j[1][2] = dr * (1.0 / (dtv * dtv));
      J.j[1][5] = dr * (-1.0 / (dtv * dtv));
      J.j[1][9] = dr * ((v1.omega / dtv) / dtv - acc_rot / dtv);
      break;
    }
    case E_TIME: J.j[0][9] = 1; break;  // edge_time_optimal.h:102-106
    case E_SHORTEST: {
      int a = e.pose[0], b = e.pose[1];
      double dx = t.x[b] - t.x[a], dy = t.y[b] - t.y[a];
      double nn = std::sqrt(dx * dx + dy * dy);
      if (nn > 0) {
        J.j[0][0] = -dx / nn; J.j[0][1] = -dy / nn;
        J.j[0][3] = dx / nn;  J.j[0][4] = dy / nn;
      }
      break;
    }
    case E_KIN_DD: {  // edge_kinematics.h:112-149
      int a = e.pose[0], b = e.pose[1];
      kin_nh_row(t, a, b, J.j[0]);
      double dx = t.x[b] - t.x[a], dy = t.y[b] - t.y[a];
      double cos1 = std::cos(t.th[a]), sin1 = std::sin(t.th[a]);
      double dd_dev = penaltyBoundFromBelowDerivative(dx * cos1 + dy * sin1, 0, 0);
      J.j[1][0] = -cos1 * dd_dev;
      J.j[1][1] = -sin1 * dd_dev;
      J.j[1][2] = (-sin1 * dx + cos1 * dy) * dd_dev;
      J.j[1][3] = cos1 * dd_dev;
      J.j[1][4] = sin1 * dd_dev;
      J.j[1][5] = 0;
      break;
    }
    case E_KIN_CL: {
      int a = e.pose[0], b = e.pose[1];
      kin_nh_row(t, a, b, J.j[0], true);
      double dx = t.x[b] - t.x[a], dy = t.y[b] - t.y[a];
      double angle_diff = normalize_theta(t.th[b] - t.th[a]);
      double nn = std::sqrt(dx * dx + dy * dy);
      if (angle_diff != 0) {
        double rho, drho_dn, drho_dth2;  // rho = radius; d/d(norm), d/d(theta_b)
        if (c.exact_arc_length) {
          double h = angle_diff / 2, sh = std::sin(h), ch = std::cos(h);
          rho = std::fabs(nn / (2 * sh));
          drho_dn = 1.0 / (2 * std::fabs(sh));
          drho_dth2 = -nn * ch * sign(sh) / (4 * sh * sh);
        } else {
          rho = nn / std::fabs(angle_diff);
          drho_dn = 1.0 / std::fabs(angle_diff);
          drho_dth2 = -nn * sign(angle_diff) / (angle_diff * angle_diff);
        }
        double dev = penaltyBoundFromBelowDerivative(rho, c.min_turning_radius, 0.0);
        double ux = 0, uy = 0;
        if (nn > 0) { ux = dx / nn; uy = dy / nn; }
        J.j[1][0] = dev * drho_dn * (-ux);
        J.j[1][1] = dev * drho_dn * (-uy);
        J.j[1][2] = dev * (-drho_dth2);
        J.j[1][3] = dev * drho_dn * ux;
        J.j[1][4] = dev * drho_dn * uy;
        J.j[1][5] = dev * drho_dth2;
      }
      break;
    }
    case E_ROTDIR: {
      int a = e.pose[0], b = e.pose[1];
      double dev = penaltyBoundFromBelowDerivative(e.dir * normalize_theta(t.th[b] - t.th[a]), 0, 0);
      J.j[0][2] = dev * (-e.dir);
      J.j[0][5] = dev * (e.dir);
      break;
    }
    case E_VOR: {
      int a = e.pose[0], b = e.pose[1];
      double dtv = t.dt[e.dts[0]];
      SignedVel sv;
      signed_velocity(c, t.x[a], t.y[a], t.th[a], t.x[b], t.y[b], t.th[b], dtv, sv, true);
      double gr[3];
      double dobs = footprint_distance(s, t.x[a], t.y[a], t.th[a], s.obst[e.obst], false, 0, gr);
      double ratio, dratio;
      if (dobs < c.obstacle_proximity_lower_bound) { ratio = 0; dratio = 0; }
      else if (dobs > c.obstacle_proximity_upper_bound) { ratio = 1; dratio = 0; }
      else {
        ratio = (dobs - c.obstacle_proximity_lower_bound) / (c.obstacle_proximity_upper_bound - c.obstacle_proximity_lower_bound);
        dratio = 1.0 / (c.obstacle_proximity_upper_bound - c.obstacle_proximity_lower_bound);
      }
      ratio *= c.obstacle_proximity_ratio_max_vel;
      dratio *= c.obstacle_proximity_ratio_max_vel;
      double max_vel_fwd = ratio * c.max_vel_x, max_omega = ratio * c.max_vel_theta;
      double dv = penaltyBoundToIntervalDerivative(sv.v, max_vel_fwd, 0);
      double dw = penaltyBoundToIntervalDerivative(sv.omega, max_omega, 0);
      // e = |var| - a when active: d/da = -1 on both active branches
      for (int q = 0; q < 6; ++q) J.j[0][q] = dv * sv.dv[q];
      J.j[0][9] = dv * sv.dv[6];
      if (dv != 0) for (int q = 0; q < 3; ++q) J.j[0][q] -= c.max_vel_x * dratio * gr[q];
      J.j[1][2] = -dw / dtv;
      J.j[1][5] = dw / dtv;
      J.j[1][9] = -dw * sv.omega / dtv;
      if (dw != 0) for (int q = 0; q < 3; ++q) J.j[1][q] -= c.max_vel_theta * dratio * gr[q];
      break;
    }
    case E_VEL_HOLO: {
      int a = e.pose[0], b = e.pose[1];
      double dtv = t.dt[e.dts[0]];
      double dx = t.x[b] - t.x[a], dy = t.y[b] - t.y[a];
      double c1 = std::cos(t.th[a]), s1 = std::sin(t.th[a]);
      double r_dx = c1 * dx + s1 * dy, r_dy = -s1 * dx + c1 * dy;
      double vx = r_dx / dtv, vy = r_dy / dtv;
      double omega = normalize_theta(t.th[b] - t.th[a]) / dtv;
      // gradients of vx, vy wrt (xa,ya,tha,xb,yb,thb,dt) -> cols 0..5, 9
      double gvx[7] = {-c1 / dtv, -s1 / dtv, r_dy / dtv, c1 / dtv, s1 / dtv, 0, -vx / dtv};
      double gvy[7] = {s1 / dtv, -c1 / dtv, -r_dx / dtv, -s1 / dtv, c1 / dtv, 0, -vy / dtv};
      double vt2 = c.max_vel_trans * c.max_vel_trans;
      double rem_y = std::sqrt(std::max(0.0, vt2 - vx * vx));
      double rem_x = std::sqrt(std::max(0.0, vt2 - vy * vy));
      // d rem_y / d vx, d rem_x / d vy
      double drem_y = (vt2 - vx * vx > 0 && rem_y > 0) ? -vx / rem_y : 0.0;
      double drem_x = (vt2 - vy * vy > 0 && rem_x > 0) ? -vy / rem_x : 0.0;
      // std::min(a,b) returns b only if b < a
      bool y_uses_rem = !(c.max_vel_y < rem_y);
      bool x_uses_rem = !(c.max_vel_x < rem_x);
      bool xb_uses_rem = !(c.max_vel_x_backwards < rem_x);
      double max_vel_y = std::min(rem_y, c.max_vel_y);
      double max_vel_x = std::min(rem_x, c.max_vel_x);
      double max_vel_x_backwards = std::min(rem_x, c.max_vel_x_backwards);
      int cols[7] = {0, 1, 2, 3, 4, 5, 9};
      // e0 = pI(vx, -max_bw, max_x, 0)
      {
        double a_ = -max_vel_x_backwards, b_ = max_vel_x;
        if (vx < a_) {  // -vx + a ; a = -max_bw -> da/dvy = -drem_x (if rem used)
          for (int q = 0; q < 7; ++q) J.j[0][cols[q]] = -gvx[q] + (xb_uses_rem ? -drem_x * gvy[q] : 0.0);
        } else if (vx <= b_) {
        } else {  // vx - b
          for (int q = 0; q < 7; ++q) J.j[0][cols[q]] = gvx[q] - (x_uses_rem ? drem_x * gvy[q] : 0.0);
        }
      }
      // e1 = pI(vy, max_vel_y, 0): var < -a -> -var - a ; var <= a -> 0 ; else var - a
      {
        if (vy < -max_vel_y) {
          for (int q = 0; q < 7; ++q) J.j[1][cols[q]] = -gvy[q] - (y_uses_rem ? drem_y * gvx[q] : 0.0);
        } else if (vy <= max_vel_y) {
        } else {
          for (int q = 0; q < 7; ++q) J.j[1][cols[q]] = gvy[q] - (y_uses_rem ? drem_y * gvx[q] : 0.0);
        }
      }
      double dw = penaltyBoundToIntervalDerivative(omega, c.max_vel_theta, c.penalty_epsilon);
      J.j[2][2] = -dw / dtv;
      J.j[2][5] = dw / dtv;
      J.j[2][9] = -dw * omega / dtv;
      break;
    }
    case E_ACC_HOLO: {
      int p1 = e.pose[0], p2 = e.pose[1], p3 = e.pose[2];
      double dt1 = t.dt[e.dts[0]], dt2 = t.dt[e.dts[1]];
      double d1x = t.x[p2] - t.x[p1], d1y = t.y[p2] - t.y[p1];
      double d2x = t.x[p3] - t.x[p2], d2y = t.y[p3] - t.y[p2];
      double c1 = std::cos(t.th[p1]), s1 = std::sin(t.th[p1]);
      double c2 = std::cos(t.th[p2]), s2 = std::sin(t.th[p2]);
      double p1_dx = c1 * d1x + s1 * d1y, p1_dy = -s1 * d1x + c1 * d1y;
      double p2_dx = c2 * d2x + s2 * d2y, p2_dy = -s2 * d2x + c2 * d2y;
      double v1x = p1_dx / dt1, v1y = p1_dy / dt1, v2x = p2_dx / dt2, v2y = p2_dy / dt2;
      double T = dt1 + dt2, f = 2 / T;
      double acc_x = (v2x - v1x) * f, acc_y = (v2y - v1y) * f;
      double om1 = normalize_theta(t.th[p2] - t.th[p1]) / dt1, om2 = normalize_theta(t.th[p3] - t.th[p2]) / dt2;
      double acc_rot = (om2 - om1) * f;
      double dxx = penaltyBoundToIntervalDerivative(acc_x, c.acc_lim_x, c.penalty_epsilon);
      double dyy = penaltyBoundToIntervalDerivative(acc_y, c.acc_lim_y, c.penalty_epsilon);
      double drr = penaltyBoundToIntervalDerivative(acc_rot, c.acc_lim_theta, c.penalty_epsilon);
      // v1x: cols pose1 (0-2), pose2 xy (3,4), dt1 (9); v2x: pose2 (3-5), pose3 xy (6,7), dt2 (10)
      double g1x[11] = {0}, g1y[11] = {0}, g2x[11] = {0}, g2y[11] = {0};
      g1x[0] = -c1 / dt1; g1x[1] = -s1 / dt1; g1x[2] = p1_dy / dt1; g1x[3] = c1 / dt1; g1x[4] = s1 / dt1; g1x[9] = -v1x / dt1;
      g1y[0] = s1 / dt1; g1y[1] = -c1 / dt1; g1y[2] = -p1_dx / dt1; g1y[3] = -s1 / dt1; g1y[4] = c1 / dt1; g1y[9] = -v1y / dt1;
      g2x[3] = -c2 / dt2; g2x[4] = -s2 / dt2; g2x[5] = p2_dy / dt2; g2x[6] = c2 / dt2; g2x[7] = s2 / dt2; g2x[10] = -v2x / dt2;
      g2y[3] = s2 / dt2; g2y[4] = -c2 / dt2; g2y[5] = -p2_dx / dt2; g2y[6] = -s2 / dt2; g2y[7] = c2 / dt2; g2y[10] = -v2y / dt2;
      for (int q = 0; q < 11; ++q) {
        double extra = (q == 9 || q == 10) ? 1.0 : 0.0;
        J.j[0][q] = dxx * (f * (g2x[q] - g1x[q]) - extra * acc_x / T);
        J.j[1][q] = dyy * (f * (g2y[q] - g1y[q]) - extra * acc_y / T);
      }
      J.j[2][2] = drr * f * (1.0 / dt1);
      J.j[2][5] = drr * f * (-1.0 / dt2 - 1.0 / dt1);
      J.j[2][8] = drr * f * (1.0 / dt2);
      J.j[2][9] = drr * (f * (om1 / dt1) - acc_rot / T);
      J.j[2][10] = drr * (f * (-om2 / dt2) - acc_rot / T);
      break;
    }
    case E_ACC_HOLO_START:
    case E_ACC_HOLO_GOAL: {
      int p1 = e.pose[0], p2 = e.pose[1];
      double dtv = t.dt[e.dts[0]];
      double dx = t.x[p2] - t.x[p1], dy = t.y[p2] - t.y[p1];
      double c1 = std::cos(t.th[p1]), s1 = std::sin(t.th[p1]);
      double pdx = c1 * dx + s1 * dy, pdy = -s1 * dx + c1 * dy;
      double vx = pdx / dtv, vy = pdy / dtv;
      double om = normalize_theta(t.th[p2] - t.th[p1]) / dtv;
      double sgn = (e.type == E_ACC_HOLO_START) ? 1.0 : -1.0;
      double ax, ay, ar;
      if (e.type == E_ACC_HOLO_START) { ax = (vx - t.vs[0]) / dtv; ay = (vy - t.vs[1]) / dtv; ar = (om - t.vs[2]) / dtv; }
      else { ax = (t.vg[0] - vx) / dtv; ay = (t.vg[1] - vy) / dtv; ar = (t.vg[2] - om) / dtv; }
      double dxx = penaltyBoundToIntervalDerivative(ax, c.acc_lim_x, c.penalty_epsilon);
      double dyy = penaltyBoundToIntervalDerivative(ay, c.acc_lim_y, c.penalty_epsilon);
      double drr = penaltyBoundToIntervalDerivative(ar, c.acc_lim_theta, c.penalty_epsilon);
      double gvx[7] = {-c1 / dtv, -s1 / dtv, pdy / dtv, c1 / dtv, s1 / dtv, 0, -vx / dtv};
      double gvy[7] = {s1 / dtv, -c1 / dtv, -pdx / dtv, -s1 / dtv, c1 / dtv, 0, -vy / dtv};
      double gom[7] = {0, 0, -1 / dtv, 0, 0, 1 / dtv, -om / dtv};
      int cols[7] = {0, 1, 2, 3, 4, 5, 9};
      for (int q = 0; q < 7; ++q) {
        double extra = (q == 6) ? 1.0 : 0.0;
        J.j[0][cols[q]] = dxx * (sgn * gvx[q] / dtv - extra * ax / dtv);
        J.j[1][cols[q]] = dyy * (sgn * gvy[q] / dtv - extra * ay / dtv);
        J.j[2][cols[q]] = drr * (sgn * gom[q] / dtv - extra * ar / dtv);
      }
      break;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// buildGraph (src/optimal_planner.cpp:323-366) and the AddEdges* family
// ------------------------------------------------------------------------------------------------
inline double cross2d(V2 a, V2 b) { return a.x * b.y - b.x * a.y; }  // misc.h:119-123

// obstacles_per_vertex_ lists in visiting order
void associate_obstacles(const Scene& s, const Teb& t, std::vector<std::vector<int>>& per_vertex, int& first_vertex) {
  const teb_amd_config_t& c = s.cfg;
  const int n = t.n();
  per_vertex.assign(n, {});
  first_vertex = c.weight_velocity_obstacle_ratio == 0 ? 1 : 0;
  int slot = 0;
  for (int i = first_vertex; i < n - 1; ++i) {
    double left_min_dist = std::numeric_limits<double>::max();
    double right_min_dist = std::numeric_limits<double>::max();
    int left_obstacle = -1, right_obstacle = -1;
    V2 pose_orient{std::cos(t.th[i]), std::sin(t.th[i])};
    std::vector<int>& lst = per_vertex[slot];
    for (int k = 0; k < (int)s.obst.size(); ++k) {
      const Obst& ob = s.obst[k];
      if (c.include_dynamic_obstacles && ob.dyn) continue;
      double dist = footprint_distance(s, t.x[i], t.y[i], t.th[i], ob, false, 0, nullptr);
      if (dist < c.min_obstacle_dist * c.obstacle_association_force_inclusion_factor) {
        lst.push_back(k);
        continue;
      }
      if (dist > c.min_obstacle_dist * c.obstacle_association_cutoff_factor) continue;
      if (cross2d(pose_orient, ob.c - V2{t.x[i], t.y[i]}) > 0) {
        if (dist < left_min_dist) { left_min_dist = dist; left_obstacle = k; }
      } else {
        if (dist < right_min_dist) { right_min_dist = dist; right_obstacle = k; }
      }
    }
    if (left_obstacle >= 0) lst.push_back(left_obstacle);
    if (right_obstacle >= 0) lst.push_back(right_obstacle);
    ++slot;
  }
}

int find_closest_pose_point(const Teb& t, V2 p, int begin_idx) {  // src/timed_elastic_band.cpp:455-478
  int n = t.n();
  if (begin_idx < 0 || begin_idx >= n) return -1;
  double min_dist_sq = std::numeric_limits<double>::max();
  int min_idx = -1;
  for (int i = begin_idx; i < n; i++) {
    double dist_sq = sqnorm(p - V2{t.x[i], t.y[i]});
    if (dist_sq < min_dist_sq) { min_dist_sq = dist_sq; min_idx = i; }
  }
  return min_idx;
}
int find_closest_pose_obstacle(const Teb& t, const Obst& o) {  // :481-552
  int n = t.n();
  if (o.type == TEB_AMD_OBST_POINT) return find_closest_pose_point(t, o.a, 0);
  if (o.type == TEB_AMD_OBST_LINE) {
    double min_dist = std::numeric_limits<double>::max();
    int min_idx = -1;
    for (int i = 0; i < n; i++) {
      double d = distance_point_to_segment_2d(V2{t.x[i], t.y[i]}, o.a, o.b).d;
      if (d < min_dist) { min_dist = d; min_idx = i; }
    }
    return min_idx;
  }
  if (o.type == TEB_AMD_OBST_POLYGON) {
    const auto& v = o.verts;
    if (v.empty()) return 0;
    if (v.size() == 1) return find_closest_pose_point(t, v.front(), 0);
    if (v.size() == 2) {
      double min_dist = std::numeric_limits<double>::max();
      int min_idx = -1;
      for (int i = 0; i < n; i++) {
        double d = distance_point_to_segment_2d(V2{t.x[i], t.y[i]}, v.front(), v.back()).d;
        if (d < min_dist) { min_dist = d; min_idx = i; }
      }
      return min_idx;
    }
    double min_dist = std::numeric_limits<double>::max();
    int min_idx = -1;
    for (int i = 0; i < n; i++) {
      V2 point{t.x[i], t.y[i]};
      double dp = std::numeric_limits<double>::max();
      for (int j = 0; j < (int)v.size() - 1; ++j) dp = std::min(dp, distance_point_to_segment_2d(point, v[j], v[j + 1]).d);
      dp = std::min(dp, distance_point_to_segment_2d(point, v.back(), v.front()).d);
      if (dp < min_dist) { min_dist = dp; min_idx = i; }
    }
    return min_idx;
  }
  return find_closest_pose_point(t, o.c, 0);
}

void add_obstacle_edge(Graph& g, int pose, int obst, double weight_multiplier) {
  const teb_amd_config_t& c = g.s->cfg;
  if (g.pose_fixed(pose)) return;  // g2o: edges whose vertices are all fixed are never activated
  bool inflated = c.inflation_dist > c.min_obstacle_dist;
  Edge e;
  e.type = inflated ? E_INFL : E_OBST;
  e.np = 1; e.pose[0] = pose; e.obst = obst;
  e.dim = inflated ? 2 : 1;
  e.info[0] = c.weight_obstacle * weight_multiplier;
  e.info[1] = c.weight_inflation;
  g.edges.push_back(e);
}

void build_graph(Graph& g, double weight_multiplier, std::vector<std::vector<int>>* per_vertex_out = nullptr) {
  const Scene& s = *g.s;
  const teb_amd_config_t& c = s.cfg;
  Teb& t = *g.teb;
  g.edges.clear();
  g.n = t.n();
  g.N = 4 * g.n - 7;
  const int n = g.n;
  std::vector<std::vector<int>> per_vertex;
  int first_vertex = 1;

  // --- AddEdgesObstacles / Legacy (src/optimal_planner.cpp:444-643)
  if (!(c.weight_obstacle == 0 || weight_multiplier == 0)) {
    if (c.legacy_obstacle_association) {
      for (int k = 0; k < (int)s.obst.size(); ++k) {
        const Obst& ob = s.obst[k];
        if (c.include_dynamic_obstacles && ob.dyn) continue;
        int index;
        if (c.obstacle_poses_affected >= n) index = n / 2;
        else index = find_closest_pose_obstacle(t, ob);
        if ((index <= 1) || (index > n - 2)) continue;
        add_obstacle_edge(g, index, k, weight_multiplier);
        for (int nb = 0; nb < std::floor(c.obstacle_poses_affected / 2); nb++) {
          if (index + nb < n) add_obstacle_edge(g, index + nb, k, weight_multiplier);
          if (index - nb >= 0) add_obstacle_edge(g, index - nb, k, weight_multiplier);
        }
      }
    } else {
      associate_obstacles(s, t, per_vertex, first_vertex);
      int slot = 0;
      for (int i = first_vertex; i < n - 1; ++i, ++slot) {
        if (i == 0) continue;
        for (int k : per_vertex[slot]) add_obstacle_edge(g, i, k, weight_multiplier);
      }
    }
  }
  if (per_vertex_out) *per_vertex_out = per_vertex;

  // --- AddEdgesDynamicObstacles (:646-673), called with weight_multiplier = 1 (:343)
  if (c.include_dynamic_obstacles && !(c.weight_obstacle == 0)) {
    for (int k = 0; k < (int)s.obst.size(); ++k) {
      if (!s.obst[k].dyn) continue;
      double time = t.dt[0];
      for (int i = 1; i < n - 1; ++i) {
        Edge e;
        e.type = E_DYN; e.np = 1; e.pose[0] = i; e.obst = k; e.t = time; e.dim = 2;
        e.info[0] = c.weight_dynamic_obstacle * 1.0;
        e.info[1] = c.weight_dynamic_obstacle_inflation;
        g.edges.push_back(e);
        time += t.dt[i];
      }
    }
  }

  // --- AddEdgesViaPoints (:675-718)
  if (!(c.weight_viapoint == 0 || !t.via_enabled || s.via.empty()) && n >= 3) {
    int start_pose_idx = 0;
    for (int v = 0; v < (int)s.via.size(); ++v) {
      int index = find_closest_pose_point(t, s.via[v], start_pose_idx);
      if (c.via_points_ordered) start_pose_idx = index + 2;
      if (index > n - 2) index = n - 2;
      if (index < 1) {
        if (c.via_points_ordered) index = 1;
        else continue;
      }
      Edge e;
      e.type = E_VIA; e.np = 1; e.pose[0] = index; e.via = v; e.dim = 1;
      e.info[0] = c.weight_viapoint;
      if (!g.pose_fixed(index)) g.edges.push_back(e);
    }
  }

  // --- AddEdgesVelocity (:720-769)
  if (c.max_vel_y == 0) {
    if (!(c.weight_max_vel_x == 0 && c.weight_max_vel_theta == 0)) {
      for (int i = 0; i < n - 1; ++i) {
        Edge e;
        e.type = E_VEL; e.np = 2; e.pose[0] = i; e.pose[1] = i + 1; e.nd = 1; e.dts[0] = i; e.dim = 2;
        e.info[0] = c.weight_max_vel_x; e.info[1] = c.weight_max_vel_theta;
        g.edges.push_back(e);
      }
    }
  } else {
    if (!(c.weight_max_vel_x == 0 && c.weight_max_vel_y == 0 && c.weight_max_vel_theta == 0)) {
      for (int i = 0; i < n - 1; ++i) {
        Edge e;
        e.type = E_VEL_HOLO; e.np = 2; e.pose[0] = i; e.pose[1] = i + 1; e.nd = 1; e.dts[0] = i; e.dim = 3;
        e.info[0] = c.weight_max_vel_x; e.info[1] = c.weight_max_vel_y; e.info[2] = c.weight_max_vel_theta;
        g.edges.push_back(e);
      }
    }
  }

  // --- AddEdgesAcceleration (:771-873)
  if (!(c.weight_acc_lim_x == 0 && c.weight_acc_lim_theta == 0)) {
    bool nonholo = (c.max_vel_y == 0 || c.acc_lim_y == 0);
    double i0 = c.weight_acc_lim_x, i1 = nonholo ? c.weight_acc_lim_theta : c.weight_acc_lim_y, i2 = c.weight_acc_lim_theta;
    int dim = nonholo ? 2 : 3;
    if (t.has_vs) {
      Edge e;
      e.type = nonholo ? E_ACC_START : E_ACC_HOLO_START;
      e.np = 2; e.pose[0] = 0; e.pose[1] = 1; e.nd = 1; e.dts[0] = 0; e.dim = dim;
      e.info[0] = i0; e.info[1] = i1; e.info[2] = i2;
      g.edges.push_back(e);
    }
    for (int i = 0; i < n - 2; ++i) {
      Edge e;
      e.type = nonholo ? E_ACC : E_ACC_HOLO;
      e.np = 3; e.pose[0] = i; e.pose[1] = i + 1; e.pose[2] = i + 2; e.nd = 2; e.dts[0] = i; e.dts[1] = i + 1; e.dim = dim;
      e.info[0] = i0; e.info[1] = i1; e.info[2] = i2;
      g.edges.push_back(e);
    }
    if (t.has_vg) {
      Edge e;
      e.type = nonholo ? E_ACC_GOAL : E_ACC_HOLO_GOAL;
      e.np = 2; e.pose[0] = n - 2; e.pose[1] = n - 1; e.nd = 1; e.dts[0] = (int)t.dt.size() - 1; e.dim = dim;
      e.info[0] = i0; e.info[1] = i1; e.info[2] = i2;
      g.edges.push_back(e);
    }
  }

  // --- AddEdgesTimeOptimal (:877-893)
  if (c.weight_optimaltime != 0) {
    for (int i = 0; i < (int)t.dt.size(); ++i) {
      Edge e;
      e.type = E_TIME; e.nd = 1; e.dts[0] = i; e.dim = 1; e.info[0] = c.weight_optimaltime;
      g.edges.push_back(e);
    }
  }
  // --- AddEdgesShortestPath (:895-912)
  if (c.weight_shortest_path != 0) {
    for (int i = 0; i < n - 1; ++i) {
      Edge e;
      e.type = E_SHORTEST; e.np = 2; e.pose[0] = i; e.pose[1] = i + 1; e.dim = 1; e.info[0] = c.weight_shortest_path;
      if (!(g.pose_fixed(i) && g.pose_fixed(i + 1))) g.edges.push_back(e);
    }
  }
  // --- kinematics (:355-358, 916-958)
  if (c.min_turning_radius == 0 || c.weight_kinematics_turning_radius == 0) {
    if (!(c.weight_kinematics_nh == 0 && c.weight_kinematics_forward_drive == 0)) {
      for (int i = 0; i < n - 1; i++) {
        Edge e;
        e.type = E_KIN_DD; e.np = 2; e.pose[0] = i; e.pose[1] = i + 1; e.dim = 2;
        e.info[0] = c.weight_kinematics_nh; e.info[1] = c.weight_kinematics_forward_drive;
        if (!(g.pose_fixed(i) && g.pose_fixed(i + 1))) g.edges.push_back(e);
      }
    }
  } else {
    if (!(c.weight_kinematics_nh == 0 && c.weight_kinematics_turning_radius == 0)) {
      for (int i = 0; i < n - 1; i++) {
        Edge e;
        e.type = E_KIN_CL; e.np = 2; e.pose[0] = i; e.pose[1] = i + 1; e.dim = 2;
        e.info[0] = c.weight_kinematics_nh; e.info[1] = c.weight_kinematics_turning_radius;
        if (!(g.pose_fixed(i) && g.pose_fixed(i + 1))) g.edges.push_back(e);
      }
    }
  }
  // --- AddEdgesPreferRotDir (:961-997)
  if (!(t.rotdir == TEB_AMD_ROT_NONE || c.weight_prefer_rotdir == 0) &&
      (t.rotdir == TEB_AMD_ROT_LEFT || t.rotdir == TEB_AMD_ROT_RIGHT)) {
    for (int i = 0; i < n - 1 && i < 3; ++i) {
      Edge e;
      e.type = E_ROTDIR; e.np = 2; e.pose[0] = i; e.pose[1] = i + 1; e.dim = 1; e.info[0] = c.weight_prefer_rotdir;
      e.dir = (t.rotdir == TEB_AMD_ROT_LEFT) ? 1 : -1;
      if (!(g.pose_fixed(i) && g.pose_fixed(i + 1))) g.edges.push_back(e);
    }
  }
  // --- AddEdgesVelocityObstacleRatio (:999-1021)
  if (c.weight_velocity_obstacle_ratio > 0 && !c.legacy_obstacle_association &&
      !(c.weight_obstacle == 0 || weight_multiplier == 0)) {
    for (int index = 0; index < n - 1; ++index) {
      for (int k : per_vertex[index]) {
        Edge e;
        e.type = E_VOR; e.np = 2; e.pose[0] = index; e.pose[1] = index + 1; e.nd = 1; e.dts[0] = index; e.dim = 2;
        e.obst = k;
        e.info[0] = c.weight_velocity_obstacle_ratio; e.info[1] = c.weight_velocity_obstacle_ratio;
        g.edges.push_back(e);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// g2o: computeActiveErrors / activeRobustChi2 / buildSystem / LM  
// ------------------------------------------------------------------------------------------------
constexpr int KD = 10;  // scalar half-bandwidth of H in g2o's vertex-id ordering

struct System {
  int N = 0;
  std::vector<double> Hb;  // lower band, Hb[r*(KD+1) + (r-c)]
  std::vector<double> b, x, L;
  void resize(int n_) {
    N = n_;
    Hb.assign((size_t)N * (KD + 1), 0.0);
    b.assign(N, 0.0);
    x.assign(N, 0.0);
    L.assign((size_t)N * (KD + 1), 0.0);
  }
  double& H(int r, int c) { return Hb[(size_t)r * (KD + 1) + (r - c)]; }
};

void compute_active_errors(Graph& g) {
  for (Edge& e : g.edges) compute_error(g, e);
}
double active_chi2(const Graph& g, double* cats = nullptr) {
  double chi = 0;
  if (cats) cats[0] = cats[1] = cats[2] = cats[3] = 0;
  for (const Edge& e : g.edges) {
    double c2 = 0;
    for (int k = 0; k < e.dim; ++k) c2 += e.err[k] * (e.info[k] * e.err[k]);  // _error.dot(information()*_error)
    chi += c2;
    if (cats) cats[e.cat()] += c2;
  }
  return chi;
}

inline int edge_col_index(const Graph& g, const Edge& e, int col) {
  if (col < 9) {
    int v = col / 3;
    if (v >= e.np) return -1;
    return g.idx_pose(e.pose[v], col % 3);
  }
  int v = col - 9;
  if (v >= e.nd) return -1;
  return g.idx_dt(e.dts[v]);
}

void build_system(Graph& g, System& sys, int jac_mode) {
  std::fill(sys.Hb.begin(), sys.Hb.end(), 0.0);
  std::fill(sys.b.begin(), sys.b.end(), 0.0);
  Jac J;
  for (Edge& e : g.edges) {
    bool analytic_in_ref = (e.type == E_KIN_DD || e.type == E_TIME);  // live analytic Jacobians of the reference
    if (jac_mode == TEB_AMD_JACOBIAN_ANALYTIC || analytic_in_ref) linearize_analytic(g, e, J);
    else linearize_numeric(g, e, J);
    // constructQuadraticForm: H_ij += J_i^T Omega J_j, b_i += J_i^T (-Omega e)
    int idx[11];
    for (int q = 0; q < 11; ++q) idx[q] = edge_col_index(g, e, q);
    for (int qa = 0; qa < 11; ++qa) {
      int ia = idx[qa];
      if (ia < 0) continue;
      double bb = 0;
      for (int k = 0; k < e.dim; ++k) bb += J.j[k][qa] * (-(e.info[k] * e.err[k]));
      sys.b[ia] += bb;
      for (int qb = 0; qb < 11; ++qb) {
        int ib = idx[qb];
        if (ib < 0 || ib > ia) continue;  // lower triangle only
        double h = 0;
        for (int k = 0; k < e.dim; ++k) h += (J.j[k][qa] * e.info[k]) * J.j[k][qb];
        sys.H(ia, ib) += h;
      }
    }
  }
}

// banded Cholesky (stand-in for CSparse cs_chol: fails iff a pivot d <= 0); solves (H + lambda I) x = b
bool solve_damped(System& sys, double lambda) {
  const int N = sys.N, W = KD + 1;
  std::vector<double>& L = sys.L;
  for (int j = 0; j < N; ++j) {
    // row j of L: columns max(0,j-KD)..j
    int c0 = std::max(0, j - KD);
    for (int c = c0; c <= j; ++c) {
      double sum = sys.Hb[(size_t)j * W + (j - c)];
      if (c == j) sum += lambda;
      int k0 = std::max(c0, std::max(0, c - KD));
      for (int k = k0; k < c; ++k) sum -= L[(size_t)j * W + (j - k)] * L[(size_t)c * W + (c - k)];
      if (c == j) {
        if (sum <= 0) return false;
        L[(size_t)j * W] = std::sqrt(sum);
      } else {
        L[(size_t)j * W + (j - c)] = sum / L[(size_t)c * W];
      }
    }
  }
  std::vector<double>& x = sys.x;
  for (int i = 0; i < N; ++i) {
    double sum = sys.b[i];
    for (int k = std::max(0, i - KD); k < i; ++k) sum -= L[(size_t)i * W + (i - k)] * x[k];
    x[i] = sum / L[(size_t)i * W];
  }
  for (int i = N - 1; i >= 0; --i) {
    double sum = x[i];
    for (int k = i + 1; k <= std::min(N - 1, i + KD); ++k) sum -= L[(size_t)k * W + (k - i)] * x[k];
    x[i] = sum / L[(size_t)i * W];
  }
  return true;
}

void apply_update(Graph& g, const std::vector<double>& x) {  // SparseOptimizer::update -> oplusImpl
  Teb& t = *g.teb;
  for (int i = 1; i < g.n - 1; ++i) {
    int b = 4 * i - 3;
    t.x[i] += x[b];
    t.y[i] += x[b + 1];
    t.th[i] = normalize_theta(t.th[i] + x[b + 2]);  // vertex_pose.h:195-198 / pose_se2.h:238-243
  }
  for (int i = 0; i < g.n - 1; ++i) t.dt[i] += x[4 * i];  // vertex_timediff.h:113-116
}

struct LMState { double lambda = 0, ni = 2; };
struct OptStats {
  int iterations = 0, trials = 0; double chi2 = 0; double lambda = 0; bool nonfinite = false;
  // opt-in trace (teb_oracle_set_trace): one row per LM iteration = {chi2 after it, lambda after it, damping trials, pose count} - the
  // data of g2o's verbose line (src/optimal_planner.cpp:384), used by the tests to say WHERE two implementations part
  double* trace = nullptr; int trace_cap = 0; int32_t* trace_rows = nullptr;
};
struct TraceCfg { double* buf = nullptr; int cap = 0; int32_t* rows = nullptr; };
TraceCfg g_trace;

// OptimizationAlgorithmLevenberg::solve; returns true for OK, false for Terminate
bool lm_solve(Graph& g, System& sys, LMState& lm, int iteration, int jac_mode, OptStats& st, double* cats_last) {
  compute_active_errors(g);
  double currentChi = active_chi2(g);
  double tempChi = currentChi;
  build_system(g, sys, jac_mode);
  if (iteration == 0) {
    double maxDiagonal = 0;
    for (int k = 0; k < sys.N; ++k) maxDiagonal = std::max(std::fabs(sys.Hb[(size_t)k * (KD + 1)]), maxDiagonal);
    lm.lambda = 1e-5 * maxDiagonal;  // _tau
    lm.ni = 2;
  }
  double rho = 0;
  int qmax = 0;
  Teb backup;
  do {
    backup = *g.teb;  // push
    bool ok2 = solve_damped(sys, lm.lambda);
    if (!ok2) sys.x = sys.b;  // LinearSolverCSparse: x was memcpy'd from b before the failed factorisation
    apply_update(g, sys.x);
    compute_active_errors(g);
    tempChi = active_chi2(g, cats_last);
    if (!ok2) tempChi = std::numeric_limits<double>::max();
    rho = (currentChi - tempChi);
    double scale = 0;
    for (int j = 0; j < sys.N; ++j) scale += sys.x[j] * (lm.lambda * sys.x[j] + sys.b[j]);
    scale += 1e-3;
    rho /= scale;
    st.trials++;
    if (rho > 0 && std::isfinite(tempChi)) {
      double alpha = 1. - std::pow((2 * rho - 1), 3);
      alpha = std::min(alpha, 2. / 3.);
      double scaleFactor = std::max(1. / 3., alpha);
      lm.lambda *= scaleFactor;
      lm.ni = 2;
      currentChi = tempChi;
    } else {
      lm.lambda *= lm.ni;
      lm.ni *= 2;
      *g.teb = backup;  // pop
      if (!std::isfinite(lm.lambda)) break;
    }
    qmax++;
  } while (rho < 0 && qmax < 10);
  st.chi2 = currentChi;
  st.lambda = lm.lambda;
  if (st.trace && st.trace_rows && *st.trace_rows < st.trace_cap) {
    double* row = st.trace + (size_t)(*st.trace_rows) * 4;
    row[0] = currentChi; row[1] = lm.lambda; row[2] = (double)qmax; row[3] = (double)g.n;
    ++*st.trace_rows;
  }
  if (qmax == 10 || rho == 0 || !std::isfinite(lm.lambda)) return false;
  return true;
}

// SparseOptimizer::optimize; returns #iterations executed
int g2o_optimize(Graph& g, System& sys, int iterations, int jac_mode, bool batch_stats, OptStats& st, double* cats_last) {
  if (g.N <= 0) return -1;
  sys.resize(g.N);
  LMState lm;
  int cj = 0;
  bool ok = true;
  for (int i = 0; i < iterations && ok; i++) {
    ok = lm_solve(g, sys, lm, i, jac_mode, st, cats_last);
    if (batch_stats) {  // computeActiveErrors again -> stored _error are fresh
      compute_active_errors(g);
      st.chi2 = active_chi2(g, cats_last);
    }
    ++cj;
  }
  return cj;
}

bool state_finite(const Teb& t) {
  for (int i = 0; i < t.n(); ++i)
    if (!std::isfinite(t.x[i]) || !std::isfinite(t.y[i]) || !std::isfinite(t.th[i])) return false;
  for (double d : t.dt) if (!std::isfinite(d)) return false;
  return true;
}

struct TebResult { int status = TEB_AMD_TEB_OK; int iterations = 0; int trials = 0; double chi2 = 0; double cost = NAN; double lambda = 0; };

// TebOptimalPlanner::optimizeTEB, src/optimal_planner.cpp:182-231
bool optimize_teb(const Scene& s, Teb& t, int inner, int outer, bool compute_cost, double obst_cost_scale,
                  double viapoint_cost_scale, bool alternative_time_cost, int cost_mode, TebResult& res, int band = -1) {
  const teb_amd_config_t& c = s.cfg;
  if (band >= 0 && g_trace.rows) g_trace.rows[band] = 0;
  if (!c.optimization_activate) return false;
  double weight_multiplier = 1.0;
  bool fast_mode = !c.include_dynamic_obstacles;
  Graph g;
  g.s = &s;
  g.teb = &t;
  System sys;
  OptStats st;
  if (band >= 0 && g_trace.buf && g_trace.rows) {
    st.trace = g_trace.buf + (size_t)band * g_trace.cap * 4; st.trace_cap = g_trace.cap; st.trace_rows = g_trace.rows + band;
  }
  for (int i = 0; i < outer; ++i) {
    if (c.teb_autosize) auto_resize(t, c.dt_ref, c.dt_hysteresis, c.min_samples, c.max_samples, fast_mode);
    build_graph(g, weight_multiplier);
    // optimizeGraph, :368-402
    if (c.max_vel_x < 0.01) return false;
    if (t.dt.empty() || t.x.empty() || t.n() < c.min_samples) return false;
    double cats[4] = {0, 0, 0, 0};
    int iter = g2o_optimize(g, sys, inner, c.jacobian_mode, c.divergence_detection_enable != 0, st, cats);
    if (iter > 0) res.iterations += iter;
    res.trials = st.trials;
    res.chi2 = st.chi2;
    res.lambda = st.lambda;
    if (!iter) return false;
    if (compute_cost && i == outer - 1) {
      // computeCurrentCost, :1041-1094
      if (cost_mode == TEB_ORACLE_COST_FRESH) {
        compute_active_errors(g);
        active_chi2(g, cats);
      }
      double cost = 0;
      if (alternative_time_cost) {
        double sum = 0;
        for (double d : t.dt) sum += d;  // getSumOfAllTimeDiffs
        cost += sum;
      }
      // Sum per edge in activeEdges order with per-type scaling (:1070-1089)
      for (const Edge& e : g.edges) {
        double cur = 0;
        for (int k = 0; k < e.dim; ++k) cur += e.err[k] * (e.info[k] * e.err[k]);
        int cat = e.cat();
        if (cat == CAT_OBST) cur *= obst_cost_scale;
        else if (cat == CAT_VIA) cur *= viapoint_cost_scale;
        else if (cat == CAT_TIME && alternative_time_cost) continue;
        cost += cur;
      }
      res.cost = cost;
    }
    weight_multiplier *= c.weight_adapt_factor;
  }
  return true;
}

void teb_from_batch(const teb_amd_teb_batch_t* bt, int b, Teb& t) {
  int n = bt->n[b];
  size_t o = (size_t)b * bt->stride;
  t.x.assign(bt->x + o, bt->x + o + n);
  t.y.assign(bt->y + o, bt->y + o + n);
  t.th.assign(bt->theta + o, bt->theta + o + n);
  t.dt.assign(bt->dt + o, bt->dt + o + std::max(0, n - 1));
  t.has_vs = !bt->has_vel_start || bt->has_vel_start[b];
  t.has_vg = !bt->has_vel_goal || bt->has_vel_goal[b];
  for (int k = 0; k < 3; ++k) {
    t.vs[k] = (bt->vel_start) ? bt->vel_start[3 * b + k] : 0.0;
    t.vg[k] = (bt->vel_goal) ? bt->vel_goal[3 * b + k] : 0.0;
  }
  t.rotdir = bt->prefer_rotdir ? bt->prefer_rotdir[b] : TEB_AMD_ROT_NONE;
  t.via_enabled = bt->via_points_enabled ? (bt->via_points_enabled[b] != 0) : true;
}
int teb_to_batch(const Teb& t, teb_amd_teb_batch_t* bt, int b) {
  int n = t.n();
  if (n > bt->stride) return TEB_AMD_ERR_CAPACITY;
  size_t o = (size_t)b * bt->stride;
  bt->n[b] = n;
  std::copy(t.x.begin(), t.x.end(), bt->x + o);
  std::copy(t.y.begin(), t.y.end(), bt->y + o);
  std::copy(t.th.begin(), t.th.end(), bt->theta + o);
  std::copy(t.dt.begin(), t.dt.end(), bt->dt + o);
  return TEB_AMD_OK;
}

}  // namespace

// ==================================================================================================
// C interface
// ==================================================================================================
extern "C" {

int teb_oracle_optimize_batch(const teb_amd_config_t* cfg, const teb_amd_obstacles_t* obst, int32_t n_via,
                              const double* via_x, const double* via_y, teb_amd_teb_batch_t* batch,
                              int32_t inner, int32_t outer, int32_t compute_cost, double obst_cost_scale,
                              double viapoint_cost_scale, int32_t alternative_time_cost, int32_t cost_mode,
                              int32_t threads, teb_amd_results_t* out) {
  if (!cfg || !batch) return TEB_AMD_ERR_INVALID_ARG;
  Scene s;
  int rc = load_scene(s, cfg, obst, n_via, via_x, via_y);
  if (rc) return rc;
  const int B = batch->count;
  std::vector<TebResult> results(B);
  std::vector<int> rcs(B, 0);
  auto work = [&](int b) {
    Teb t;
    teb_from_batch(batch, b, t);
    TebResult& r = results[b];
    bool ok = optimize_teb(s, t, inner, outer, compute_cost != 0, obst_cost_scale, viapoint_cost_scale,
                           alternative_time_cost != 0, cost_mode, r, b);
    r.status = ok ? TEB_AMD_TEB_OK : TEB_AMD_TEB_FAILED;
    if (!state_finite(t)) r.status = TEB_AMD_TEB_NONFINITE;
    rcs[b] = teb_to_batch(t, batch, b);
  };
  if (threads <= 1 || B <= 1) {
    for (int b = 0; b < B; ++b) work(b);
  } else {
    std::atomic<int> next{0};
    std::vector<std::thread> pool;
    int nt = std::min<int>(threads, B);
    for (int k = 0; k < nt; ++k)
      pool.emplace_back([&]() {
        for (;;) {
          int b = next.fetch_add(1);
          if (b >= B) break;
          work(b);
        }
      });
    for (auto& th : pool) th.join();
  }
  for (int b = 0; b < B; ++b) if (rcs[b]) return rcs[b];
  if (out) {
    for (int b = 0; b < B; ++b) {
      if (out->status) out->status[b] = results[b].status;
      if (out->lm_iterations) out->lm_iterations[b] = results[b].iterations;
      if (out->lm_trials) out->lm_trials[b] = results[b].trials;
      if (out->chi2) out->chi2[b] = results[b].chi2;
      if (out->cost) out->cost[b] = results[b].cost;
      if (out->lambda) out->lambda[b] = results[b].lambda;
    }
  }
  return TEB_AMD_OK;
}

// trace of the following teb_oracle_optimize_batch calls: buf [B][cap_rows][4], rows [B]; buf = NULL switches it off. Not thread-safe
// against concurrent optimize_batch calls (one process-wide setting; the bands of one call write disjoint slices).
int teb_oracle_set_trace(double* buf, int32_t cap_rows, int32_t* rows) {
  g_trace.buf = buf; g_trace.cap = cap_rows; g_trace.rows = rows;
  return TEB_AMD_OK;
}

// src/homotopy_class_planner.cpp:564-667 (without the ros::Time switching block)
int teb_oracle_select_best(const teb_amd_config_t* cfg, int32_t count, const double* cost, int32_t last_best,
                           int32_t initial_plan, int32_t* best, double* best_cost) {
  if (!cfg || !cost || !best) return TEB_AMD_ERR_INVALID_ARG;
  double min_cost = std::numeric_limits<double>::max();
  double min_cost_last_best = std::numeric_limits<double>::max();
  double min_cost_initial_plan_teb = std::numeric_limits<double>::max();
  bool have_last = last_best >= 0 && last_best < count;
  bool have_init = initial_plan >= 0 && initial_plan < count;
  if (have_last) min_cost_last_best = cost[last_best] * cfg->selection_cost_hysteresis;
  if (have_init) min_cost_initial_plan_teb = cost[initial_plan] * cfg->selection_prefer_initial_plan;
  int b = -1;
  for (int i = 0; i < count; ++i) {
    double teb_cost;
    if (have_last && i == last_best) teb_cost = min_cost_last_best;
    else if (have_init && i == initial_plan) teb_cost = min_cost_initial_plan_teb;
    else teb_cost = cost[i];
    if (teb_cost < min_cost) { b = i; min_cost = teb_cost; }
  }
  *best = b;
  if (best_cost) *best_cost = min_cost;
  return TEB_AMD_OK;
}

int teb_oracle_autoresize(double* x, double* y, double* theta, double* dt, int32_t* n, int32_t cap, double dt_ref,
                          double dt_hysteresis, int32_t min_samples, int32_t max_samples, int32_t fast_mode) {
  Teb t;
  t.x.assign(x, x + *n); t.y.assign(y, y + *n); t.th.assign(theta, theta + *n);
  t.dt.assign(dt, dt + std::max(0, *n - 1));
  auto_resize(t, dt_ref, dt_hysteresis, min_samples, max_samples, fast_mode != 0);
  if (t.n() > cap) return TEB_AMD_ERR_CAPACITY;
  *n = t.n();
  std::copy(t.x.begin(), t.x.end(), x); std::copy(t.y.begin(), t.y.end(), y);
  std::copy(t.th.begin(), t.th.end(), theta); std::copy(t.dt.begin(), t.dt.end(), dt);
  return TEB_AMD_OK;
}

// ================================================================================================================
// the producers and consumers either side of optimizeTEB.
// ================================================================================================================
namespace {

// estimateDeltaT, src/timed_elastic_band.cpp:52-65
double estimate_delta_t(double sx, double sy, double sth, double ex, double ey, double eth, double max_vel_x, double max_vel_theta) {
  double dt_constant_motion = 0.1;
  if (max_vel_x > 0) {
    double trans_dist = norm(V2{ex, ey} - V2{sx, sy});
    dt_constant_motion = trans_dist / max_vel_x;
  }
  if (max_vel_theta > 0) {
    double rot_dist = std::abs(normalize_theta(eth - sth));
    dt_constant_motion = std::max(dt_constant_motion, rot_dist / max_vel_theta);
  }
  return dt_constant_motion;
}

void push_pose(Teb& t, double x, double y, double th) { t.x.push_back(x); t.y.push_back(y); t.th.push_back(th); }
void push_pose_dt(Teb& t, double x, double y, double th, double dt) { push_pose(t, x, y, th); t.dt.push_back(dt); }

// PoseSE2::average(BackPose(), goal), pose_se2.h:266-269
void back_goal_average(const Teb& t, double gx, double gy, double gth, double& ax, double& ay, double& ath) {
  ax = (t.x.back() + gx) / 2; ay = (t.y.back() + gy) / 2;
  ath = average_angle(t.th.back(), gth);
}

// TimedElasticBand::initTrajectoryToGoal(start, goal, diststep, max_vel_x, min_samples, guess_backwards_motion), :325-377
bool init_trajectory_line(Teb& t, const double* start, const double* goal, double diststep, double max_vel_x, int min_samples,
                          bool guess_backwards_motion) {
  if (t.n() != 0 || !t.dt.empty()) return false;   // isInit()
  push_pose(t, start[0], start[1], start[2]);
  double timestep = 0.1;
  if (diststep != 0) {
    V2 point_to_goal = V2{goal[0], goal[1]} - V2{start[0], start[1]};
    double dir_to_goal = std::atan2(point_to_goal.y, point_to_goal.x);
    double dx = diststep * std::cos(dir_to_goal);
    double dy = diststep * std::sin(dir_to_goal);
    double orient_init = dir_to_goal;
    if (guess_backwards_motion && dot(point_to_goal, V2{std::cos(start[2]), std::sin(start[2])}) < 0)
      orient_init = normalize_theta(orient_init + M_PI);
    double dist_to_goal = norm(point_to_goal);
    double no_steps_d = dist_to_goal / std::abs(diststep);
    unsigned int no_steps = (unsigned int)std::floor(no_steps_d);
    if (max_vel_x > 0) timestep = diststep / max_vel_x;
    for (unsigned int i = 1; i <= no_steps; i++) {
      if (i == no_steps && no_steps_d == (float)no_steps) break;
      push_pose_dt(t, start[0] + i * dx, start[1] + i * dy, orient_init, timestep);
    }
  }
  if (t.n() < min_samples - 1) {
    while (t.n() < min_samples - 1) {
      double ax, ay, ath;
      back_goal_average(t, goal[0], goal[1], goal[2], ax, ay, ath);
      if (max_vel_x > 0) timestep = norm(V2{ax, ay} - V2{t.x.back(), t.y.back()}) / max_vel_x;
      push_pose_dt(t, ax, ay, ath, timestep);
    }
  }
  if (max_vel_x > 0) timestep = norm(V2{goal[0], goal[1]} - V2{t.x.back(), t.y.back()}) / max_vel_x;
  push_pose_dt(t, goal[0], goal[1], goal[2], timestep);
  return true;
}

// initTrajectoryToGoal(plan, max_vel_x, max_vel_theta, estimate_orient, min_samples, guess_backwards_motion), :380-452
// plan given as positions + yaw (tf::getYaw of the pose orientation is taken by the caller)
bool init_trajectory_plan(Teb& t, int np, const double* px, const double* py, const double* pyaw, double max_vel_x,
                          double max_vel_theta, bool estimate_orient, int min_samples, bool guess_backwards_motion) {
  if (t.n() != 0 || !t.dt.empty()) return false;
  const double sx = px[0], sy = py[0], sth = pyaw[0];
  const double gx = px[np - 1], gy = py[np - 1], gth = pyaw[np - 1];
  push_pose(t, sx, sy, sth);
  bool backwards = false;
  if (guess_backwards_motion && dot(V2{gx, gy} - V2{sx, sy}, V2{std::cos(sth), std::sin(sth)}) < 0) backwards = true;
  for (int i = 1; i < np - 1; ++i) {
    double yaw;
    if (estimate_orient) {
      double dx = px[i + 1] - px[i];
      double dy = py[i + 1] - py[i];
      yaw = std::atan2(dy, dx);
      if (backwards) yaw = normalize_theta(yaw + M_PI);
    } else {
      yaw = pyaw[i];
    }
    double dt = estimate_delta_t(t.x.back(), t.y.back(), t.th.back(), px[i], py[i], yaw, max_vel_x, max_vel_theta);
    push_pose_dt(t, px[i], py[i], yaw, dt);
  }
  if (t.n() < min_samples - 1) {
    while (t.n() < min_samples - 1) {
      double ax, ay, ath;
      back_goal_average(t, gx, gy, gth, ax, ay, ath);
      double dt = estimate_delta_t(t.x.back(), t.y.back(), t.th.back(), ax, ay, ath, max_vel_x, max_vel_theta);
      push_pose_dt(t, ax, ay, ath, dt);
    }
  }
  double dt = estimate_delta_t(t.x.back(), t.y.back(), t.th.back(), gx, gy, gth, max_vel_x, max_vel_theta);
  push_pose_dt(t, gx, gy, gth, dt);
  return true;
}

// template initTrajectoryToGoal(path_start, path_end, fun_position, ...), timed_elastic_band.hpp:46-183 (positions only)
bool init_trajectory_path(Teb& t, int np, const double* px, const double* py, double max_vel_x, double max_vel_theta,
                          bool has_max_acc_x, double max_acc_x, bool has_start_orient, double start_orientation,
                          bool has_goal_orient, double goal_orientation, int min_samples, bool guess_backwards_motion) {
  (void)max_vel_theta;
  V2 start_position{px[0], py[0]}, goal_position{px[np - 1], py[np - 1]};
  bool backwards = false;
  double start_orient, goal_orient;
  if (has_start_orient) {
    start_orient = start_orientation;
    if (guess_backwards_motion && dot(goal_position - start_position, V2{std::cos(start_orient), std::sin(start_orient)}) < 0)
      backwards = true;
  } else {
    V2 start2goal = goal_position - start_position;
    start_orient = std::atan2(start2goal.y, start2goal.x);
  }
  double timestep = 1;
  goal_orient = has_goal_orient ? goal_orientation : start_orient;
  if (t.n() != 0 || !t.dt.empty()) return false;
  push_pose(t, start_position.x, start_position.y, start_orient);
  int idx = 0;
  for (int k = 1; k < np - 1; ++k) {
    V2 curr_point{px[k], py[k]};
    V2 diff_last = curr_point - V2{t.x[idx], t.y[idx]};
    double diff_norm = norm(diff_last);
    double timestep_vel = diff_norm / max_vel_x;
    double timestep_acc;
    if (has_max_acc_x) {
      timestep_acc = std::sqrt(2 * diff_norm / max_acc_x);
      if (timestep_vel < timestep_acc && has_max_acc_x) timestep = timestep_acc;
      else timestep = timestep_vel;
    } else timestep = timestep_vel;
    if (timestep <= 0) timestep = 0.2;
    double yaw = std::atan2(diff_last.y, diff_last.x);
    if (backwards) yaw = normalize_theta(yaw + M_PI);
    push_pose_dt(t, curr_point.x, curr_point.y, yaw, timestep);
    ++idx;
  }
  V2 diff = goal_position - V2{t.x[idx], t.y[idx]};
  double diff_norm = norm(diff);
  double timestep_vel = diff_norm / max_vel_x;
  if (has_max_acc_x) {
    double timestep_acc = std::sqrt(2 * diff_norm / max_acc_x);
    if (timestep_vel < timestep_acc) timestep = timestep_acc;
    else timestep = timestep_vel;
  } else timestep = timestep_vel;
  if (t.n() < min_samples - 1) {
    while (t.n() < min_samples - 1) {
      timestep /= 2;
      double ax, ay, ath;
      back_goal_average(t, goal_position.x, goal_position.y, goal_orient, ax, ay, ath);
      push_pose_dt(t, ax, ay, ath, timestep);
    }
  }
  push_pose_dt(t, goal_position.x, goal_position.y, goal_orient, timestep);
  return true;
}

// TimedElasticBand::updateAndPruneTEB, src/timed_elastic_band.cpp:555-597
void update_and_prune(Teb& t, bool has_start, const double* ns, bool has_goal, const double* ng, int min_samples) {
  if (has_start && t.n() > 0) {
    V2 p{ns[0], ns[1]};
    double dist_cache = norm(p - V2{t.x[0], t.y[0]});
    double dist;
    int lookahead = std::min<int>(t.n() - min_samples, 10);
    int nearest_idx = 0;
    for (int i = 1; i <= lookahead; ++i) {
      dist = norm(p - V2{t.x[i], t.y[i]});
      if (dist < dist_cache) { dist_cache = dist; nearest_idx = i; }
      else break;
    }
    if (nearest_idx > 0) {   // deletePoses(1, nearest_idx); deleteTimeDiffs(1, nearest_idx)
      t.x.erase(t.x.begin() + 1, t.x.begin() + 1 + nearest_idx);
      t.y.erase(t.y.begin() + 1, t.y.begin() + 1 + nearest_idx);
      t.th.erase(t.th.begin() + 1, t.th.begin() + 1 + nearest_idx);
      t.dt.erase(t.dt.begin() + 1, t.dt.begin() + 1 + nearest_idx);
    }
    t.x[0] = ns[0]; t.y[0] = ns[1]; t.th[0] = ns[2];
  }
  if (has_goal && t.n() > 0) { t.x.back() = ng[0]; t.y.back() = ng[1]; t.th.back() = ng[2]; }
}

// TebOptimalPlanner::extractVelocity, src/optimal_planner.cpp:1097-1133
void extract_velocity(const teb_amd_config_t& c, double x1, double y1, double th1, double x2, double y2, double th2, double dt,
                      double& vx, double& vy, double& omega) {
  if (dt == 0) { vx = 0; vy = 0; omega = 0; return; }
  V2 deltaS = V2{x2, y2} - V2{x1, y1};
  if (c.max_vel_y == 0) {
    V2 conf1dir{std::cos(th1), std::sin(th1)};
    double dir = dot(deltaS, conf1dir);
    vx = sign(dir) * norm(deltaS) / dt;   // (double) g2o::sign(dir)
    vy = 0;
  } else {
    double cos_theta1 = std::cos(th1);
    double sin_theta1 = std::sin(th1);
    double p1_dx = cos_theta1 * deltaS.x + sin_theta1 * deltaS.y;
    double p1_dy = -sin_theta1 * deltaS.x + cos_theta1 * deltaS.y;
    vx = p1_dx / dt;
    vy = p1_dy / dt;
  }
  double orientdiff = normalize_theta(th2 - th1);
  omega = orientdiff / dt;
}

int teb_out(const Teb& t, double* x, double* y, double* th, double* dt, int32_t* n, int cap) {
  if (t.n() > cap) return TEB_AMD_ERR_CAPACITY;
  *n = t.n();
  std::copy(t.x.begin(), t.x.end(), x); std::copy(t.y.begin(), t.y.end(), y);
  std::copy(t.th.begin(), t.th.end(), th); std::copy(t.dt.begin(), t.dt.end(), dt);
  return TEB_AMD_OK;
}

}  // namespace

int teb_oracle_init_trajectory_line(const double* start, const double* goal, double diststep, double max_vel_x, int32_t min_samples,
                                    int32_t guess_backwards_motion, double* x, double* y, double* th, double* dt, int32_t* n,
                                    int32_t cap) {
  Teb t;
  if (!init_trajectory_line(t, start, goal, diststep, max_vel_x, min_samples, guess_backwards_motion != 0)) return TEB_AMD_ERR_INVALID_ARG;
  return teb_out(t, x, y, th, dt, n, cap);
}

int teb_oracle_init_trajectory_plan(int32_t np, const double* px, const double* py, const double* pyaw, double max_vel_x,
                                    double max_vel_theta, int32_t estimate_orient, int32_t min_samples,
                                    int32_t guess_backwards_motion, double* x, double* y, double* th, double* dt, int32_t* n,
                                    int32_t cap) {
  if (np < 1) return TEB_AMD_ERR_INVALID_ARG;
  Teb t;
  if (!init_trajectory_plan(t, np, px, py, pyaw, max_vel_x, max_vel_theta, estimate_orient != 0, min_samples, guess_backwards_motion != 0))
    return TEB_AMD_ERR_INVALID_ARG;
  return teb_out(t, x, y, th, dt, n, cap);
}

int teb_oracle_init_trajectory_path(int32_t np, const double* px, const double* py, double max_vel_x, double max_vel_theta,
                                    int32_t has_max_acc_x, double max_acc_x, int32_t has_start_orient, double start_orient,
                                    int32_t has_goal_orient, double goal_orient, int32_t min_samples,
                                    int32_t guess_backwards_motion, double* x, double* y, double* th, double* dt, int32_t* n,
                                    int32_t cap) {
  if (np < 1) return TEB_AMD_ERR_INVALID_ARG;
  Teb t;
  if (!init_trajectory_path(t, np, px, py, max_vel_x, max_vel_theta, has_max_acc_x != 0, max_acc_x, has_start_orient != 0, start_orient,
                            has_goal_orient != 0, goal_orient, min_samples, guess_backwards_motion != 0))
    return TEB_AMD_ERR_INVALID_ARG;
  return teb_out(t, x, y, th, dt, n, cap);
}

int teb_oracle_update_and_prune(double* x, double* y, double* th, double* dt, int32_t* n, const double* new_start,
                                const double* new_goal, int32_t min_samples) {
  Teb t;
  t.x.assign(x, x + *n); t.y.assign(y, y + *n); t.th.assign(th, th + *n); t.dt.assign(dt, dt + std::max(0, *n - 1));
  update_and_prune(t, new_start != nullptr, new_start, new_goal != nullptr, new_goal, min_samples);
  return teb_out(t, x, y, th, dt, n, *n);
}

int teb_oracle_velocity_command(const teb_amd_config_t* cfg, const teb_amd_teb_batch_t* batch, int32_t b, int32_t look_ahead_poses,
                                int32_t prevent_look_ahead_poses_near_goal, double* v, int32_t* ok) {
  Teb t;
  teb_from_batch(batch, b, t);
  v[0] = v[1] = v[2] = 0; *ok = 0;
  if (t.n() < 2) return TEB_AMD_OK;                                   // :1137-1144
  look_ahead_poses = std::max(1, std::min(look_ahead_poses, t.n() - 1 - prevent_look_ahead_poses_near_goal));
  double dt = 0.0;
  for (int counter = 0; counter < look_ahead_poses; ++counter) {
    dt += t.dt[counter];
    if (dt >= cfg->dt_ref * look_ahead_poses) { look_ahead_poses = counter + 1; break; }
  }
  if (dt <= 0) return TEB_AMD_OK;                                     // :1156-1163
  extract_velocity(*cfg, t.x[0], t.y[0], t.th[0], t.x[look_ahead_poses], t.y[look_ahead_poses], t.th[look_ahead_poses], dt, v[0], v[1], v[2]);
  *ok = 1;
  return TEB_AMD_OK;
}

// getVelocityProfile :1170-1196 -> out[(n+1)*3] = (linear.x, linear.y, angular.z)
int teb_oracle_velocity_profile(const teb_amd_config_t* cfg, const teb_amd_teb_batch_t* batch, int32_t b, double* out) {
  Teb t;
  teb_from_batch(batch, b, t);
  const int n = t.n();
  for (int k = 0; k < 3; ++k) { out[k] = t.vs[k]; }
  for (int i = 1; i < n; ++i)
    extract_velocity(*cfg, t.x[i - 1], t.y[i - 1], t.th[i - 1], t.x[i], t.y[i], t.th[i], t.dt[i - 1], out[3 * i], out[3 * i + 1], out[3 * i + 2]);
  for (int k = 0; k < 3; ++k) out[3 * n + k] = t.vg[k];
  return TEB_AMD_OK;
}

// getFullTrajectory :1198-1247 -> out[n*7] = (x, y, theta, vx, vy, omega, time_from_start)
int teb_oracle_full_trajectory(const teb_amd_config_t* cfg, const teb_amd_teb_batch_t* batch, int32_t b, double* out) {
  Teb t;
  teb_from_batch(batch, b, t);
  const int n = t.n();
  if (n == 0) return TEB_AMD_OK;
  double curr_time = 0;
  auto put = [&](int i, double vx, double vy, double om, double tt) {
    double* o = out + 7 * i;
    o[0] = t.x[i]; o[1] = t.y[i]; o[2] = t.th[i]; o[3] = vx; o[4] = vy; o[5] = om; o[6] = tt;
  };
  put(0, t.vs[0], t.vs[1], t.vs[2], curr_time);
  curr_time += t.dt[0];
  for (int i = 1; i < n - 1; ++i) {
    double vel1_x, vel1_y, vel2_x, vel2_y, omega1, omega2;
    extract_velocity(*cfg, t.x[i - 1], t.y[i - 1], t.th[i - 1], t.x[i], t.y[i], t.th[i], t.dt[i - 1], vel1_x, vel1_y, omega1);
    extract_velocity(*cfg, t.x[i], t.y[i], t.th[i], t.x[i + 1], t.y[i + 1], t.th[i + 1], t.dt[i], vel2_x, vel2_y, omega2);
    put(i, 0.5 * (vel1_x + vel2_x), 0.5 * (vel1_y + vel2_y), 0.5 * (omega1 + omega2), curr_time);
    curr_time += t.dt[i];
  }
  put(n - 1, t.vg[0], t.vg[1], t.vg[2], curr_time);
  return TEB_AMD_OK;
}

// ================================================================================================================
// SURVEY section 8(f) row f4, arithmetic part: isTrajectoryFeasible on a costmap grid.
// ================================================================================================================
double teb_oracle_footprint_cost(const uint8_t* cells, int32_t size_x, int32_t size_y, double resolution, double origin_x, double origin_y,
                                 double x, double y, double theta, int32_t nf, const double* fx, const double* fy) {
  return gridcostmap::footprint_cost(gridcostmap::Grid{cells, size_x, size_y, resolution, origin_x, origin_y}, x, y, theta, nf, fx, fy);
}

// TebOptimalPlanner::isTrajectoryFeasible, src/optimal_planner.cpp:1250-1308
int teb_oracle_is_trajectory_feasible(const teb_amd_teb_batch_t* batch, int32_t b, const uint8_t* cells, int32_t size_x, int32_t size_y,
                                      double resolution, double origin_x, double origin_y, int32_t nf, const double* fx, const double* fy,
                                      double inscribed_radius, double min_resolution_collision_check_angular, int32_t look_ahead_idx,
                                      double feasibility_check_lookahead_distance, int32_t* feasible, int32_t* first_infeasible) {
  Teb t;
  teb_from_batch(batch, b, t);
  const int n = t.n();
  int tests = 0;
  auto cost = [&](double x, double y, double th) {
    return teb_oracle_footprint_cost(cells, size_x, size_y, resolution, origin_x, origin_y, x, y, th, nf, fx, fy);
  };
  auto fail = [&]() { *feasible = 0; if (first_infeasible) *first_infeasible = tests; return TEB_AMD_OK; };
  if (look_ahead_idx < 0 || look_ahead_idx >= n) look_ahead_idx = n - 1;
  if (feasibility_check_lookahead_distance > 0) {
    for (int i = 1; i < n; ++i) {
      const double pose_distance = std::hypot(t.x[i] - t.x[0], t.y[i] - t.y[0]);
      if (pose_distance > feasibility_check_lookahead_distance) { look_ahead_idx = i - 1; break; }
    }
  }
  for (int i = 0; i <= look_ahead_idx; ++i) {
    if (cost(t.x[i], t.y[i], t.th[i]) == -1) return fail();
    ++tests;
    if (i < look_ahead_idx) {
      const double delta_rot = normalize_theta(normalize_theta(t.th[i + 1]) - normalize_theta(t.th[i]));
      const double ddx = t.x[i + 1] - t.x[i], ddy = t.y[i + 1] - t.y[i];
      const double dnorm = std::sqrt(ddx * ddx + ddy * ddy);   // Eigen::Vector2d::norm()
      if (fabs(delta_rot) > min_resolution_collision_check_angular || dnorm > inscribed_radius) {
        const int n_additional_samples = (int)std::max(std::ceil(fabs(delta_rot) / min_resolution_collision_check_angular),
                                                       std::ceil(dnorm / inscribed_radius)) - 1;
        double px = t.x[i], py = t.y[i], pth = t.th[i];
        for (int step = 0; step < n_additional_samples; ++step) {
          px = px + ddx / (n_additional_samples + 1.0);
          py = py + ddy / (n_additional_samples + 1.0);
          pth = normalize_theta(pth + delta_rot / (n_additional_samples + 1.0));
          if (cost(px, py, pth) == -1) return fail();
          ++tests;
        }
      }
    }
  }
  *feasible = 1;
  if (first_infeasible) *first_infeasible = -1;
  return TEB_AMD_OK;
}

// ================================================================================================================
// SURVEY section 8(f) row f3, arithmetic core: equivalence classes of candidate bands (h_signature.h).
// ================================================================================================================
namespace {

typedef std::complex<long double> cplx;

// HSignature::calculateHSignature, h_signature.h:96-188 (2-D complex-log signature; long double like the reference)
cplx h_signature_2d(const Scene& s, const Teb& t, double prescaler) {
  const int M = (int)s.obst.size();
  if (M == 0) return cplx(0, 0);
  int m = std::max(M - 1, 5);
  int a = (int)std::ceil(double(m) / 2.0);
  int b = m - a;
  const int last = t.n() - 1;
  cplx start((long double)t.x[0], (long double)t.y[0]);
  cplx end((long double)t.x[last], (long double)t.y[last]);
  cplx delta = end - start;
  cplx normal(-delta.imag(), delta.real());
  cplx map_bottom_left, map_top_right;
  if (std::abs(delta) < 3.0) {
    map_bottom_left = start + cplx(0, -3);
    map_top_right = start + cplx(3, 3);
  } else {
    map_bottom_left = start - normal;
    map_top_right = start + delta + normal;
  }
  cplx hsig = 0;
  std::vector<double> imag_proposals(5);
  for (int i = 0; i < last; ++i) {
    cplx z1((long double)t.x[i], (long double)t.y[i]);
    cplx z2((long double)t.x[i + 1], (long double)t.y[i + 1]);
    for (int l = 0; l < M; ++l) {
      cplx obst_l((long double)s.obst[l].c.x, (long double)s.obst[l].c.y);   // getCentroidCplx
      cplx f0 = (long double)prescaler * (long double)a * (obst_l - map_bottom_left) * (long double)b * (obst_l - map_top_right);
      cplx Al = f0;
      for (int j = 0; j < M; ++j) {
        if (j == l) continue;
        cplx obst_j((long double)s.obst[j].c.x, (long double)s.obst[j].c.y);
        cplx diff = obst_l - obst_j;
        if (std::abs(diff) < 0.05) continue;
        else Al /= diff;
      }
      double diff2 = std::abs(z2 - obst_l);
      double diff1 = std::abs(z1 - obst_l);
      if (diff2 == 0 || diff1 == 0) continue;
      double log_real = std::log(diff2) - std::log(diff1);
      double arg_diff = std::arg(z2 - obst_l) - std::arg(z1 - obst_l);
      imag_proposals[0] = arg_diff;
      imag_proposals[1] = arg_diff + 2 * M_PI;
      imag_proposals[2] = arg_diff - 2 * M_PI;
      imag_proposals[3] = arg_diff + 4 * M_PI;
      imag_proposals[4] = arg_diff - 4 * M_PI;
      double log_imag = *std::min_element(imag_proposals.begin(), imag_proposals.end(),
                                          [](double x, double y) { return std::abs(x) < std::abs(y); });   // smaller_than_abs, misc.h
      cplx log_value(log_real, log_imag);
      hsig += Al * log_value;
    }
  }
  return hsig;
}

// HSignature3d::calculateHSignature, h_signature.h:281-347 (x-y-t Biot-Savart signature, one value per obstacle)
void h_signature_3d(const Scene& s, const Teb& t, std::vector<double>& out) {
  const int M = (int)s.obst.size();
  out.assign(M, 0.0);
  const int last = t.n() - 1;
  constexpr int num_int_steps_per_segment = 10;
  for (int l = 0; l < M; ++l) {
    double H = 0;
    double transition_time = 0;
    double next_transition_time = 0;
    const double s1[3] = {s.obst[l].c.x, s.obst[l].c.y, 0};
    double tt = 120;
    const double s2[3] = {s.obst[l].c.x + tt * s.obst[l].vel.x, s.obst[l].c.y + tt * s.obst[l].vel.y, tt};   // predictCentroidConstantVelocity
    const double ds[3] = {s2[0] - s1[0], s2[1] - s1[1], s2[2] - s1[2]};
    const double ds_sq_norm = ((0.0 + ds[0] * ds[0]) + ds[1] * ds[1]) + ds[2] * ds[2];
    for (int i = 0; i < last; ++i) {
      cplx z1((long double)t.x[i], (long double)t.y[i]);
      cplx z2((long double)t.x[i + 1], (long double)t.y[i + 1]);
      transition_time = next_transition_time;
      next_transition_time += t.dt[i];
      double dir[3];
      dir[0] = (double)(z2.real() - z1.real());
      dir[1] = (double)(z2.imag() - z1.imag());
      dir[2] = next_transition_time - transition_time;
      if (std::sqrt(((0.0 + dir[0] * dir[0]) + dir[1] * dir[1]) + dir[2] * dir[2]) < 1e-15) continue;
      double r[3] = {(double)z1.real(), (double)z1.imag(), transition_time};
      const double sc = 1.0 / static_cast<double>(num_int_steps_per_segment);
      const double dl[3] = {dir[0] * sc, dir[1] * sc, dir[2] * sc};
      for (int k = 0; k < num_int_steps_per_segment; ++k) {
        double p1[3], p2[3], c12[3], d[3], c2[3], c1[3], phi[3];
        for (int q = 0; q < 3; ++q) { p1[q] = s1[q] - r[q]; p2[q] = s2[q] - r[q]; }
        cross3(p1, p2, c12);
        cross3(ds, c12, d);
        for (int q = 0; q < 3; ++q) d[q] = d[q] / ds_sq_norm;
        cross3(d, p2, c2);
        cross3(d, p1, c1);
        const double n2 = std::sqrt(((0.0 + p2[0] * p2[0]) + p2[1] * p2[1]) + p2[2] * p2[2]);
        const double n1 = std::sqrt(((0.0 + p1[0] * p1[0]) + p1[1] * p1[1]) + p1[2] * p1[2]);
        const double f = 1.0 / (((0.0 + d[0] * d[0]) + d[1] * d[1]) + d[2] * d[2]);
        for (int q = 0; q < 3; ++q) phi[q] = (c2[q] / n2 - c1[q] / n1) * f;   // scalar * vector: coefficient * scalar in the shim / Eigen
        H += ((0.0 + phi[0] * dl[0]) + phi[1] * dl[1]) + phi[2] * dl[2];
        for (int q = 0; q < 3; ++q) r[q] += dl[q];
      }
    }
    out[l] = H / (4.0 * M_PI);
  }
}

}  // namespace

int teb_oracle_h_signature_2d(const teb_amd_config_t* cfg, const teb_amd_obstacles_t* obst, const teb_amd_teb_batch_t* batch, int32_t b,
                              double prescaler, double* re_im) {
  Scene s;
  int rc = load_scene(s, cfg, obst, 0, nullptr, nullptr);
  if (rc) return rc;
  Teb t;
  teb_from_batch(batch, b, t);
  cplx h = h_signature_2d(s, t, prescaler);
  re_im[0] = (double)h.real(); re_im[1] = (double)h.imag();
  return TEB_AMD_OK;
}

int teb_oracle_h_signature_3d(const teb_amd_config_t* cfg, const teb_amd_obstacles_t* obst, const teb_amd_teb_batch_t* batch, int32_t b,
                              double* values) {
  Scene s;
  int rc = load_scene(s, cfg, obst, 0, nullptr, nullptr);
  if (rc) return rc;
  Teb t;
  teb_from_batch(batch, b, t);
  std::vector<double> v;
  h_signature_3d(s, t, v);
  std::copy(v.begin(), v.end(), values);
  return TEB_AMD_OK;
}

// isValid / isReasonable / isEqual of both classes (h_signature.h:190-226, 349-409) and the "first come first serve" class list of
// HomotopyClassPlanner::renewAndAnalyzeOldTebs / addEquivalenceClassIfNew (src/homotopy_class_planner.cpp:178-254).
// sig: mode 2 -> [B*2] (re, im); mode 3 -> [B*M]. best = index of the last best TEB or -1. keep[b] = 1 iff the band survives.
int teb_oracle_filter_equivalence_classes_stale(int32_t mode, int32_t B, int32_t M, const double* sig, double threshold, int32_t best,
                                                int32_t max_number_plans_in_current_class, const double* stale_best_sig, int32_t* keep,
                                                int32_t* valid, int32_t* reasonable);
int teb_oracle_filter_equivalence_classes(int32_t mode, int32_t B, int32_t M, const double* sig, double threshold, int32_t best,
                                          int32_t max_number_plans_in_current_class, int32_t* keep, int32_t* valid,
                                          int32_t* reasonable) {
  return teb_oracle_filter_equivalence_classes_stale(mode, B, M, sig, threshold, best, max_number_plans_in_current_class, nullptr, keep,
                                                     valid, reasonable);
}
// stale_best_sig: best_teb_eq_class_ left over from an earlier call (the member outlives the band it was computed from and is only
// replaced when a best band exists, src/homotopy_class_planner.cpp:220-228); NULL = none. Used when best < 0.
int teb_oracle_filter_equivalence_classes_stale(int32_t mode, int32_t B, int32_t M, const double* sig, double threshold, int32_t best,
                                                int32_t max_number_plans_in_current_class, const double* stale_best_sig, int32_t* keep,
                                                int32_t* valid, int32_t* reasonable) {
  const int W = mode == 2 ? 2 : M;
  auto is_valid = [&](int b) { for (int k = 0; k < W; ++k) if (!std::isfinite(sig[(size_t)b * W + k])) return false; return true; };
  auto is_reasonable = [&](int b) { if (mode == 2) return true; for (int k = 0; k < W; ++k) if (sig[(size_t)b * W + k] > 1.0) return false; return true; };
  auto sgn = [](double z) { return (z == 0) ? 0 : (z < 0 ? -1 : 1); };   // boost::math::sign
  auto rows_equal = [&](const double* x, const double* y) {   // x.isEqual(y)
    if (mode == 2) {
      double diff_real = std::abs(y[0] - x[0]);
      double diff_imag = std::abs(y[1] - x[1]);
      return diff_real <= threshold && diff_imag <= threshold;
    }
    for (int i = 0; i < W; ++i) {
      if (std::abs(y[i]) < threshold || std::abs(x[i]) < threshold) continue;
      if (sgn(y[i]) != sgn(x[i])) return false;
    }
    return true;
  };
  auto is_equal = [&](int a, int b) {   // a.isEqual(b)
    const double* x = sig + (size_t)a * W; const double* y = sig + (size_t)b * W;
    if (mode == 2) {
      double diff_real = std::abs(y[0] - x[0]);
      double diff_imag = std::abs(y[1] - x[1]);
      return diff_real <= threshold && diff_imag <= threshold;
    }
    for (int i = 0; i < W; ++i) {
      if (std::abs(y[i]) < threshold || std::abs(x[i]) < threshold) continue;
      if (sgn(y[i]) != sgn(x[i])) return false;
    }
    return true;
  };
  std::vector<int> order(B);
  for (int b = 0; b < B; ++b) order[b] = b;
  const bool has_best = best >= 0 && best < B;
  if (has_best) std::swap(order[0], order[best]);   // std::iter_swap(tebs_.begin(), it_best_teb)
  std::vector<int> classes;   // equivalence_classes_ (band indices)
  for (int b = 0; b < B; ++b) { keep[b] = 0; valid[b] = is_valid(b); reasonable[b] = is_reasonable(b); }
  for (int k = 0; k < B; ++k) {
    const int b = order[k];
    bool add;
    if (!valid[b]) add = false;
    else {
      bool has = false;
      for (int c : classes) if (is_equal(b, c)) { has = true; break; }
      add = true;
      if (has) {
        const double* best_sig = has_best ? sig + (size_t)order[0] * W : stale_best_sig;   // best_teb_eq_class_
        bool in_best = best_sig && rows_equal(best_sig, sig + (size_t)b * W);             // best_teb_eq_class_->isEqual(*eq_class)
        int count = 0;
        if (best_sig) for (int c : classes) if (rows_equal(best_sig, sig + (size_t)c * W)) ++count;   // numTebsInBestTebClass
        if (!in_best || count >= max_number_plans_in_current_class) add = false;
      }
    }
    if (add) { classes.push_back(b); keep[b] = 1; }
  }
  return TEB_AMD_OK;
}

// ---- row f3, candidate generation: graph_search.cpp + addAndInitNewTeb --------------------------------------------------------
namespace {

// Obstacle::checkLineIntersection per class (obstacles.h:339-354, 483-498, 647-650, 794-797; src/obstacles.cpp:178-191)
bool segments_intersect(V2 l1s, V2 l1e, V2 l2s, V2 l2e) {   // check_line_segments_intersection_2d, distance_calculations.h:97-127
  V2 line1 = l1e - l1s;
  V2 line2 = l2e - l2s;
  double denom = line1.x * line2.y - line2.x * line1.y;
  if (denom == 0) return false;
  bool denomPositive = denom > 0;
  V2 aux = l1s - l2s;
  double s_numer = line1.x * aux.y - line1.y * aux.x;
  if ((s_numer < 0) == denomPositive) return false;
  double t_numer = line2.x * aux.y - line2.y * aux.x;
  if ((t_numer < 0) == denomPositive) return false;
  if (((s_numer > denom) == denomPositive) || ((t_numer > denom) == denomPositive)) return false;
  return true;
}
bool check_line_intersection(const Obst& o, V2 ls, V2 le, double min_dist) {
  switch (o.type) {
    case TEB_AMD_OBST_POINT:
    case TEB_AMD_OBST_CIRCULAR: {
      V2 a = le - ls;
      V2 b = o.a - ls;
      double t = dot(a, b) / dot(a, a);
      if (t < 0) t = 0; else if (t > 1) t = 1;
      V2 nearest = ls + t * a;   // line_start + a*t
      double d = norm(nearest - o.a);
      if (o.type == TEB_AMD_OBST_CIRCULAR) d = d - o.r;
      return d < min_dist;
    }
    case TEB_AMD_OBST_LINE:
    case TEB_AMD_OBST_PILL:
      return segments_intersect(ls, le, o.a, o.b);
    default: {
      const int nv = (int)o.verts.size();
      for (int i = 0; i < nv - 1; ++i)
        if (segments_intersect(ls, le, o.verts[i], o.verts[i + 1])) return true;
      if (nv == 2) return false;
      return segments_intersect(ls, le, o.verts[nv - 1], o.verts[0]);
    }
  }
}

V2 normalized_in_place(V2 v) {   // Eigen normalize(): v /= sqrt(squaredNorm) when squaredNorm > 0
  double z = v.x * v.x + v.y * v.y;
  if (z > 0) { double n = std::sqrt(z); v.x = v.x / n; v.y = v.y / n; }
  return v;
}

struct HcGraph {
  std::vector<V2> pos;
  std::vector<std::vector<int>> adj;   // out-edges in insertion order
  int add_vertex(V2 p) { pos.push_back(p); adj.emplace_back(); return (int)pos.size() - 1; }
};

struct ClassSig { std::vector<double> v; };

struct Explorer {
  const Scene* s;
  const teb_amd_hcp_params_t* p;
  int mode, W;
  std::vector<Teb> tebs;
  std::vector<ClassSig> classes;     // equivalence_classes_ (one per teb, same order)
  bool has_best = false;
  ClassSig best_class;
  int cap = 0;                       // slots available (the reference has no such bound; max_number_classes <= cap is required)
  int n_paths = 0;
  int64_t max_paths = 0;             // > 0: stop after this many start-goal paths (guard of the tests; the reference has none)
  int64_t expansions = 0;

  ClassSig signature(const Teb& t) const {
    ClassSig c;
    if (mode == 2) { cplx h = h_signature_2d(*s, t, p->h_signature_prescaler); c.v = {(double)h.real(), (double)h.imag()}; }
    else h_signature_3d(*s, t, c.v);
    return c;
  }
  bool is_valid(const ClassSig& c) const { for (double z : c.v) if (!std::isfinite(z)) return false; return true; }
  bool is_equal(const ClassSig& a, const ClassSig& b) const {   // a.isEqual(b)
    const double thr = p->h_signature_threshold;
    if (mode == 2) return std::abs(b.v[0] - a.v[0]) <= thr && std::abs(b.v[1] - a.v[1]) <= thr;
    auto sgn = [](double z) { return (z == 0) ? 0 : (z < 0 ? -1 : 1); };
    for (size_t i = 0; i < a.v.size(); ++i) {
      if (std::abs(b.v[i]) < thr || std::abs(a.v[i]) < thr) continue;
      if (sgn(b.v[i]) != sgn(a.v[i])) return false;
    }
    return true;
  }
  // addEquivalenceClassIfNew, src/homotopy_class_planner.cpp:189-212
  bool add_class_if_new(const ClassSig& c) {
    if (!is_valid(c)) return false;
    bool has = false;
    for (const ClassSig& e : classes) if (is_equal(c, e)) { has = true; break; }
    if (has) {
      bool in_best = has_best && is_equal(best_class, c);
      int count = 0;
      if (has_best) for (const ClassSig& e : classes) if (is_equal(best_class, e)) ++count;
      if (!in_best || count >= p->max_number_plans_in_current_class) return false;
    }
    classes.push_back(c);
    return true;
  }
  // addAndInitNewTeb(path_start, path_end, fun_position, start_orientation, goal_orientation, ...), homotopy_class_planner.hpp:66-93
  void add_and_init_path(const HcGraph& g, const std::vector<int>& path, double start_orientation, double goal_orientation) {
    ++n_paths;
    std::vector<double> px, py;
    for (int v : path) { px.push_back(g.pos[v].x); py.push_back(g.pos[v].y); }
    Teb t;
    const teb_amd_config_t& c = s->cfg;
    init_trajectory_path(t, (int)path.size(), px.data(), py.data(), c.max_vel_x, c.max_vel_theta, true, c.acc_lim_x, true, start_orientation,
                         true, goal_orientation, c.min_samples, p->allow_init_with_backwards_motion != 0);
    ClassSig H = signature(t);
    if (add_class_if_new(H)) tebs.push_back(t);
  }
  // addAndInitNewTeb(start, goal, ...), src/homotopy_class_planner.cpp:358-384
  void add_and_init_line(const double* start, const double* goal) {
    if ((int)tebs.size() >= p->max_number_classes) return;
    Teb t;
    const teb_amd_config_t& c = s->cfg;
    init_trajectory_line(t, start, goal, 0, c.max_vel_x, c.min_samples, p->allow_init_with_backwards_motion != 0);
    ClassSig H = signature(t);
    if (add_class_if_new(H)) tebs.push_back(t);
  }
  // GraphSearchInterface::DepthFirst, src/graph_search.cpp:45-91
  void depth_first(const HcGraph& g, std::vector<int>& visited, int goal, double start_orientation, double goal_orientation) {
    if ((int)tebs.size() >= p->max_number_classes || (int)tebs.size() >= cap) return;
    if (max_paths > 0 && (n_paths >= max_paths || ++expansions > max_paths * 10000)) return;   // test guard (dead-end subtrees can be exponential)
    const int back = visited.back();
    for (int v : g.adj[back]) {
      if (std::find(visited.begin(), visited.end(), v) != visited.end()) continue;
      if (v == goal) {
        visited.push_back(v);
        add_and_init_path(g, visited, start_orientation, goal_orientation);
        visited.pop_back();
        break;
      }
    }
    for (int v : g.adj[back]) {
      if (std::find(visited.begin(), visited.end(), v) != visited.end() || v == goal) continue;
      visited.push_back(v);
      depth_first(g, visited, goal, start_orientation, goal_orientation);
      visited.pop_back();
    }
  }
};

// boost::random::uniform_real_distribution<double>(a, b) on mt19937 (one 32-bit draw per value), boost/random/uniform_real_distribution.hpp
struct Mt19937 {   // the 32-bit Mersenne twister (Matsumoto & Nishimura 1998), default seed 5489 as boost::random::mt19937()
  uint32_t mt[624]; int idx = 624;
  explicit Mt19937(uint32_t seed = 5489u) { mt[0] = seed; for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i; }
  uint32_t next() {
    if (idx >= 624) {
      for (int i = 0; i < 624; ++i) {
        uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
        mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
  }
};

}  // namespace

// createGraph + DepthFirst + addAndInitNewTeb on bands 0..n_tebs-1 (= tebs_ after renewAndAnalyzeOldTebs). batch->count = slots.
// unit_samples [2*no_samples] (u in [0,1): value = u * (b - a) + a) or NULL = boost mt19937 default stream after skip_draws draws.
int teb_oracle_explore_candidates(const teb_amd_config_t* cfg, const teb_amd_hcp_params_t* p, const teb_amd_obstacles_t* obst,
                                  teb_amd_teb_batch_t* batch, int32_t n_tebs, int32_t best, const double* start, const double* goal,
                                  double dist_to_obst, const double* unit_samples, int64_t skip_draws, int64_t max_paths,
                                  const double* stale_best_sig, int32_t* n_total, int32_t vcap, double* vx, double* vy, int32_t* nv,
                                  int32_t acap, int32_t* adj_off, int32_t* adj, int32_t* n_paths, int32_t n_plan, const double* plan_x,
                                  const double* plan_y, const double* plan_yaw, const double* stale_initial_sig, int32_t* initial_plan_teb,
                                  int32_t* via_enabled) {
  Scene s;
  int rc = load_scene(s, cfg, obst, 0, nullptr, nullptr);
  if (rc) return rc;
  Explorer ex;
  ex.s = &s; ex.p = p; ex.mode = cfg->include_dynamic_obstacles ? 3 : 2; ex.W = ex.mode == 2 ? 2 : (int)s.obst.size();
  ex.cap = batch->count;
  ex.max_paths = max_paths;
  for (int b = 0; b < n_tebs; ++b) {
    Teb t;
    teb_from_batch(batch, b, t);
    ex.classes.push_back(ex.signature(t));
    ex.tebs.push_back(t);
  }
  if (best >= 0 && best < n_tebs) { ex.has_best = true; ex.best_class = ex.classes[best]; }
  else if (stale_best_sig) { ex.has_best = true; ex.best_class.v.assign(stale_best_sig, stale_best_sig + ex.W); }   // stale best_teb_eq_class_
  // ---- addAndInitNewTeb(*initial_plan_, ...), src/homotopy_class_planner.cpp:326-329, 412-440
  int initial_idx = -1;
  bool have_initial_class = false;
  ClassSig initial_class;
  if (stale_initial_sig) { have_initial_class = true; initial_class.v.assign(stale_initial_sig, stale_initial_sig + ex.W); }
  if (n_plan > 0 && (int)ex.tebs.size() < p->max_number_classes) {
    Teb t;
    init_trajectory_plan(t, n_plan, plan_x, plan_y, plan_yaw, cfg->max_vel_x, cfg->max_vel_theta, p->global_plan_overwrite_orientation != 0,
                         cfg->min_samples, p->allow_init_with_backwards_motion != 0);
    initial_class = ex.signature(t); have_initial_class = true;   // initial_plan_eq_class_
    if (ex.add_class_if_new(initial_class)) { ex.tebs.push_back(t); initial_idx = (int)ex.tebs.size() - 1; }
  }
  HcGraph g;
  const double thr = p->obstacle_heading_threshold;
  const V2 sp{start[0], start[1]}, gp{goal[0], goal[1]};
  auto finish = [&]() {
    *n_total = (int)ex.tebs.size();
    for (int b = n_tebs; b < (int)ex.tebs.size(); ++b) { int r = teb_to_batch(ex.tebs[b], batch, b); if (r) return r; }
    for (int b = (int)ex.tebs.size(); b < batch->count; ++b) batch->n[b] = 0;
    const int N = (int)g.pos.size();
    if (nv) *nv = N;
    if (n_paths) *n_paths = ex.n_paths;
    int e = 0;
    for (int v = 0; v < N && v < vcap; ++v) {
      vx[v] = g.pos[v].x; vy[v] = g.pos[v].y; adj_off[v] = e;
      for (int w : g.adj[v]) { if (e < acap) adj[e] = w; ++e; }
    }
    if (N <= vcap && adj_off) adj_off[N] = e;
    // getInitialPlanTEB (:495-536): the band made from the plan, else the first band whose class equals initial_plan_eq_class_
    if (initial_idx < 0 && have_initial_class && ex.is_valid(initial_class))
      for (int b = 0; b < (int)ex.classes.size(); ++b)
        if (ex.is_equal(ex.classes[b], initial_class)) { initial_idx = b; break; }
    if (initial_plan_teb) *initial_plan_teb = initial_idx;
    // updateReferenceTrajectoryViaPoints (:286-315) given that via-points exist and weight_viapoint > 0: in = flags of the existing
    // bands, new bands start at 0 (constructed without via-points)
    if (via_enabled) {
      for (int b = n_tebs; b < (int)ex.tebs.size(); ++b) via_enabled[b] = 0;
      if (p->viapoints_all_candidates) { for (int b = 0; b < (int)ex.tebs.size(); ++b) via_enabled[b] = 1; }
      else if (n_plan > 0)
        for (int b = 0; b < (int)ex.tebs.size(); ++b) via_enabled[b] = have_initial_class && ex.is_equal(initial_class, ex.classes[b]);
    }
    return (int)TEB_AMD_OK;
  };
  if ((int)ex.tebs.size() >= p->max_number_classes) return finish();   // src/graph_search.cpp:99-100, 231-232
  V2 diff = gp - sp;
  const double start_goal_dist = norm(diff);
  if (start_goal_dist < p->xy_goal_tolerance) {                          // :104-113, :237-246
    if (ex.tebs.empty()) ex.add_and_init_line(start, goal);
    return finish();
  }
  int start_vtx, goal_vtx;
  if (p->simple_exploration) {                                           // lrKeyPointGraph::createGraph, :95-223
    V2 normal{-diff.y, diff.x};
    normal = normalized_in_place(normal);
    normal = dist_to_obst * normal;
    start_vtx = g.add_vertex(sp);
    diff = normalized_in_place(diff);
    int near_u = -1, near_v = -1;
    double min_dist = DBL_MAX;
    for (const Obst& o : s.obst) {
      V2 start2obst = o.c - sp;
      double dist = norm(start2obst);
      if (dot(start2obst, diff) / dist < 0.1) continue;
      int u = g.add_vertex(o.c + normal);
      int v = g.add_vertex(o.c - normal);
      if (thr && dist < min_dist) { min_dist = dist; near_u = u; near_v = v; }
    }
    goal_vtx = g.add_vertex(gp);
    const int N = (int)g.pos.size();
    for (int i = 0; i < N - 1; ++i)
      for (int j = 0; j < N; ++j) {
        if (i == j) continue;
        V2 distij = normalized_in_place(g.pos[j] - g.pos[i]);
        if (dot(distij, diff) <= thr) continue;
        if (thr && i == start_vtx && min_dist != DBL_MAX) {
          if (j == near_u || j == near_v) {
            V2 keypoint_dist = normalized_in_place(g.pos[j] - sp);
            V2 start_orient_vec{std::cos(start[2]), std::sin(start[2])};
            if (dot(start_orient_vec, keypoint_dist) <= thr) continue;
          }
        }
        bool collision = false;
        for (const Obst& o : s.obst)
          if (check_line_intersection(o, g.pos[i], g.pos[j], 0.5 * dist_to_obst)) { collision = true; break; }
        if (collision) continue;
        g.adj[i].push_back(j);
      }
  } else {                                                               // ProbRoadmapGraph::createGraph, :227-340
    V2 normal = normalized_in_place(V2{-diff.y, diff.x});
    const double area_width = p->roadmap_graph_area_width;
    const double bx = start_goal_dist * p->roadmap_graph_area_length_scale;   // distribution_x(0, bx), distribution_y(0, area_width)
    const double phi = std::atan2(diff.y, diff.x);
    V2 area_origin;
    if (p->roadmap_graph_area_length_scale != 1.0) {
      V2 dn = normalized_in_place(diff);   // diff.normalized(): same quotients
      area_origin = (sp + (0.5 * (1.0 - p->roadmap_graph_area_length_scale) * start_goal_dist) * dn) - (0.5 * area_width) * normal;
    } else {
      area_origin = sp - (0.5 * area_width) * normal;
    }
    start_vtx = g.add_vertex(sp);
    diff = normalized_in_place(diff);
    Mt19937 eng;
    for (int64_t k = 0; k < skip_draws; ++k) eng.next();
    int drawn = 0;
    auto draw = [&](double a, double b) {
      if (unit_samples) { double u = unit_samples[drawn++]; return u * (b - a) + a; }
      for (;;) {
        double numerator = (double)eng.next();
        double divisor = 4294967295.0 + 1;
        double result = numerator / divisor * (b - a) + a;
        if (result < b) return result;
      }
    };
    for (int i = 0; i < p->roadmap_graph_no_samples; ++i) {
      // Eigen::Vector2d(distribution_x(rnd), distribution_y(rnd)), :274: the order of the two draws is unspecified in C++; GCC
      // (the compiler of every ROS distribution the reference targets) evaluates the arguments right to left: y is drawn first
      double uy = draw(0, area_width);
      double ux = draw(0, bx);
      V2 rot{std::cos(phi) * ux - std::sin(phi) * uy, std::sin(phi) * ux + std::cos(phi) * uy};
      g.add_vertex(area_origin + rot);
    }
    goal_vtx = g.add_vertex(gp);
    const int N = (int)g.pos.size();
    for (int i = 0; i < N - 1; ++i)
      for (int j = 0; j < N; ++j) {
        if (i == j) continue;
        V2 distij = normalized_in_place(g.pos[j] - g.pos[i]);
        if (dot(distij, diff) <= thr) continue;
        bool collision = false;
        for (const Obst& o : s.obst)
          if (check_line_intersection(o, g.pos[i], g.pos[j], dist_to_obst)) { collision = true; break; }
        if (collision) continue;
        g.adj[i].push_back(j);
      }
  }
  std::vector<int> visited{start_vtx};
  ex.depth_first(g, visited, goal_vtx, start[2], goal[2]);
  return finish();
}

// HomotopyClassPlanner::deletePlansDetouringBackwards + computeStartOrientation (src/homotopy_class_planner.cpp:766-838) on the bands
// with keep[b] != 0 (keep in/out); optimized [B] = TebOptimalPlanner::isOptimized() of every band.
int teb_oracle_filter_detours(const teb_amd_teb_batch_t* batch, const teb_amd_hcp_params_t* p, int32_t best, const int32_t* optimized,
                              int32_t* keep) {
  const int B = batch->count;
  int kept = 0;
  for (int b = 0; b < B; ++b) kept += keep[b] != 0;
  if (kept < 2 || best < 0 || best >= B || !keep[best] || batch->n[best] < 2) return TEB_AMD_OK;
  auto start_orientation = [&](int b, double& orientation) {
    Teb t;
    teb_from_batch(batch, b, t);
    bool second_pose_found = false;
    V2 start_vector{0, 0};
    for (int i = 0; i < t.n(); ++i) {
      start_vector = V2{t.x[0], t.y[0]} - V2{t.x[i], t.y[i]};
      if (norm(start_vector) > p->length_start_orientation_vector) { second_pose_found = true; break; }
    }
    if (!second_pose_found) return false;
    orientation = std::atan2(start_vector.y, start_vector.x);
    return true;
  };
  auto sum_dt = [&](int b) {   // getSumOfAllTimeDiffs, src/timed_elastic_band.cpp:227-236
    double time = 0;
    const size_t o = (size_t)b * batch->stride;
    for (int i = 0; i < batch->n[b] - 1; ++i) time += batch->dt[o + i];
    return time;
  };
  double current_movement_orientation;
  const double best_plan_duration = std::max(sum_dt(best), 1.0);
  if (!start_orientation(best, current_movement_orientation)) return TEB_AMD_OK;
  for (int b = 0; b < B; ++b) {
    if (!keep[b] || b == best) continue;
    if (batch->n[b] < 2) { keep[b] = 0; continue; }
    double plan_orientation;
    if (!start_orientation(b, plan_orientation)) { keep[b] = 0; continue; }
    if (std::fabs(normalize_theta(plan_orientation - current_movement_orientation)) > p->detours_orientation_tolerance) { keep[b] = 0; continue; }
    if (!optimized[b]) { keep[b] = 0; continue; }
    if (sum_dt(b) / best_plan_duration > p->max_ratio_detours_duration_best_duration) { keep[b] = 0; continue; }
  }
  return TEB_AMD_OK;
}

int teb_oracle_linearize(const teb_amd_config_t* cfg, const teb_amd_obstacles_t* obst, int32_t n_via,
                         const double* via_x, const double* via_y, const teb_amd_teb_batch_t* batch, int32_t b,
                         double weight_multiplier, double* H_dense, double* bvec, double* chi2, int32_t* n_edges,
                         int32_t* n_rows) {
  if (!cfg || !batch) return TEB_AMD_ERR_INVALID_ARG;
  Scene s;
  int rc = load_scene(s, cfg, obst, n_via, via_x, via_y);
  if (rc) return rc;
  Teb t;
  teb_from_batch(batch, b, t);
  Graph g;
  g.s = &s; g.teb = &t;
  build_graph(g, weight_multiplier);
  System sys;
  sys.resize(g.N);
  compute_active_errors(g);
  double cats[4];
  active_chi2(g, cats);
  build_system(g, sys, cfg->jacobian_mode);
  const int n = g.n, D = 4 * n;
  // g2o index -> canonical: dt_i (4i) -> 4i+3 ; pose i comp c (4i-3+c) -> 4i+c
  auto canon = [&](int gi) {
    if (gi % 4 == 0) return gi + 3;        // dt_i at 4i -> 4i+3
    int i = (gi + 3) / 4, c = (gi + 3) % 4;  // 4i-3+c -> i, c
    return 4 * i + c;
  };
  if (H_dense) {
    std::fill(H_dense, H_dense + (size_t)D * D, 0.0);
    for (int r = 0; r < sys.N; ++r)
      for (int c = std::max(0, r - KD); c <= r; ++c) {
        double v = sys.H(r, c);
        int cr = canon(r), cc = canon(c);
        H_dense[(size_t)cr * D + cc] = v;
        H_dense[(size_t)cc * D + cr] = v;
      }
  }
  if (bvec) {
    std::fill(bvec, bvec + D, 0.0);
    for (int r = 0; r < sys.N; ++r) bvec[canon(r)] = sys.b[r];
  }
  if (chi2) for (int k = 0; k < 4; ++k) chi2[k] = cats[k];
  if (n_edges) *n_edges = (int)g.edges.size();
  if (n_rows) { int r = 0; for (const Edge& e : g.edges) r += e.dim; *n_rows = r; }
  return TEB_AMD_OK;
}

int teb_oracle_edges(const teb_amd_config_t* cfg, const teb_amd_obstacles_t* obst, int32_t n_via, const double* via_x,
                     const double* via_y, const teb_amd_teb_batch_t* batch, int32_t b, double weight_multiplier,
                     int32_t* irec, double* drec, int32_t cap, int32_t* count) {
  if (!cfg || !batch || !count) return TEB_AMD_ERR_INVALID_ARG;
  Scene s;
  int rc = load_scene(s, cfg, obst, n_via, via_x, via_y);
  if (rc) return rc;
  Teb t;
  teb_from_batch(batch, b, t);
  Graph g;
  g.s = &s; g.teb = &t;
  build_graph(g, weight_multiplier);
  compute_active_errors(g);
  int k = 0;
  for (Edge& e : g.edges) {
    if (k < cap) {
      Jac J;
      bool analytic_in_ref = (e.type == E_KIN_DD || e.type == E_TIME);
      if (cfg->jacobian_mode == TEB_AMD_JACOBIAN_ANALYTIC || analytic_in_ref) linearize_analytic(g, e, J);
      else linearize_numeric(g, e, J);
      int32_t* ir = irec + (size_t)k * 16;
      double* dr = drec + (size_t)k * 56;
      for (int q = 0; q < 16; ++q) ir[q] = 0;
      for (int q = 0; q < 56; ++q) dr[q] = 0;
      ir[0] = e.type; ir[1] = e.np; ir[2] = e.pose[0]; ir[3] = e.pose[1]; ir[4] = e.pose[2];
      ir[5] = e.nd; ir[6] = e.dts[0]; ir[7] = e.dts[1]; ir[8] = e.dim; ir[9] = e.obst; ir[10] = e.via;
      for (int q = 0; q < 3; ++q) { dr[q] = e.err[q]; dr[3 + q] = e.info[q]; }
      dr[6] = e.t; dr[7] = e.dir;
      for (int r = 0; r < 3; ++r) for (int q = 0; q < 11; ++q) dr[8 + r * 11 + q] = (r < e.dim) ? J.j[r][q] : 0.0;
    }
    ++k;
  }
  *count = k;
  return TEB_AMD_OK;
}

int teb_oracle_associate(const teb_amd_config_t* cfg, const teb_amd_obstacles_t* obst, const teb_amd_teb_batch_t* batch,
                         int32_t b, int32_t* assoc_pose, int32_t* assoc_obst, int32_t cap, int32_t* count) {
  if (!cfg || !batch || !count) return TEB_AMD_ERR_INVALID_ARG;
  Scene s;
  int rc = load_scene(s, cfg, obst, 0, nullptr, nullptr);
  if (rc) return rc;
  Teb t;
  teb_from_batch(batch, b, t);
  Graph g;
  g.s = &s; g.teb = &t;
  build_graph(g, 1.0);
  int k = 0;
  for (const Edge& e : g.edges) {
    if (e.type != E_OBST && e.type != E_INFL) continue;
    if (k < cap) { if (assoc_pose) assoc_pose[k] = e.pose[0]; if (assoc_obst) assoc_obst[k] = e.obst; }
    ++k;
  }
  *count = k;
  return TEB_AMD_OK;
}

int teb_oracle_distance(const teb_amd_config_t* cfg, const teb_amd_obstacles_t* obst, int32_t obst_index, double x,
                        double y, double theta, int32_t spatio_temporal, double t, double* dist, double* grad) {
  if (!cfg || !obst || !dist) return TEB_AMD_ERR_INVALID_ARG;
  Scene s;
  int rc = load_scene(s, cfg, obst, 0, nullptr, nullptr);
  if (rc) return rc;
  if (obst_index < 0 || obst_index >= (int)s.obst.size()) return TEB_AMD_ERR_INVALID_ARG;
  *dist = footprint_distance(s, x, y, theta, s.obst[obst_index], spatio_temporal != 0, t, grad);
  return TEB_AMD_OK;
}

int teb_oracle_centroid(const teb_amd_obstacles_t* obst, int32_t obst_index, double* cx, double* cy) {
  teb_amd_config_t cfg;
  std::memset(&cfg, 0, sizeof(cfg));
  Scene s;
  int rc = load_scene(s, &cfg, obst, 0, nullptr, nullptr);
  if (rc) return rc;
  if (obst_index < 0 || obst_index >= (int)s.obst.size()) return TEB_AMD_ERR_INVALID_ARG;
  *cx = s.obst[obst_index].c.x;
  *cy = s.obst[obst_index].c.y;
  return TEB_AMD_OK;
}

}  // extern "C"
