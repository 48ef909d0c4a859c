// teb_geometry.hpp — device library: 2-D distances between the robot footprint and obstacles, with the
// witness points needed for closed-form gradients (kernel K10 of SURVEY.md §7).
//
// Behaviour follows (argument order, strict '<' argmin order, early return on the closing edge):
//   distance_calculations.h:60-262, obstacles.h:358-397 / 502-541 / 653-695 / 800-843 / 968-1021,
//   robot_footprint_model.h:160-175 / 263-278 / 351-372 / 496-517 / 664-683 of the reference.
// Independent of oracle/ (the oracle is the checker, never a dependency).
#pragma once
#include "teb_device.hpp"

namespace tebamd {

struct Wit {   // distance and the pair of closest points: (p1x,p1y) on the ROBOT, (p2x,p2y) on the OBSTACLE
  double d, p1x, p1y, p2x, p2y;
};

__device__ __forceinline__ double nrm2(double x, double y) { return sqrt(x * x + y * y); }

// distance_calculations.h:60-76
__device__ __forceinline__ void closest_on_segment(double px, double py, double ax, double ay, double bx, double by,
                                                   double& cx, double& cy) {
  double dx = bx - ax, dy = by - ay;
  double sq = dx * dx + dy * dy;
  if (sq == 0) { cx = ax; cy = ay; return; }
  double u = ((px - ax) * dx + (py - ay) * dy) / sq;
  if (u <= 0) { cx = ax; cy = ay; }
  else if (u >= 1) { cx = bx; cy = by; }
  else { cx = ax + u * dx; cy = ay + u * dy; }
}

// :85-88 ; returns distance, closest point in (cx,cy)
__device__ __forceinline__ double point_segment(double px, double py, double ax, double ay, double bx, double by,
                                                double& cx, double& cy) {
  closest_on_segment(px, py, ax, ay, bx, by, cx, cy);
  return nrm2(px - cx, py - cy);
}

// :99-130
__device__ __forceinline__ bool segments_intersect(double s1x, double s1y, double e1x, double e1y, double s2x,
                                                   double s2y, double e2x, double e2y) {
  double l1x = e1x - s1x, l1y = e1y - s1y;
  double l2x = e2x - s2x, l2y = e2y - s2y;
  double denom = l1x * l2y - l2x * l1y;
  if (denom == 0) return false;
  bool dp = denom > 0;
  double auxx = s1x - s2x, auxy = s1y - s2y;
  double s_numer = l1x * auxy - l1y * auxx;
  if ((s_numer < 0) == dp) return false;
  double t_numer = l2x * auxy - l2y * auxx;
  if ((t_numer < 0) == dp) return false;
  if (((s_numer > denom) == dp) || ((t_numer > denom) == dp)) return false;
  return true;
}

// :140-157. Witness (q1 on segment 1, q2 on segment 2). std::min_element -> first smallest.
__device__ __forceinline__ double segment_segment(double s1x, double s1y, double e1x, double e1y, double s2x,
                                                  double s2y, double e2x, double e2y, double& q1x, double& q1y,
                                                  double& q2x, double& q2y) {
  if (segments_intersect(s1x, s1y, e1x, e1y, s2x, s2y, e2x, e2y)) {
    q1x = q2x = s1x; q1y = q2y = s1y;
    return 0.0;
  }
  double cx, cy;
  double best = point_segment(s1x, s1y, s2x, s2y, e2x, e2y, cx, cy);
  q1x = s1x; q1y = s1y; q2x = cx; q2y = cy;
  double d = point_segment(e1x, e1y, s2x, s2y, e2x, e2y, cx, cy);
  if (d < best) { best = d; q1x = e1x; q1y = e1y; q2x = cx; q2y = cy; }
  d = point_segment(s2x, s2y, s1x, s1y, e1x, e1y, cx, cy);
  if (d < best) { best = d; q2x = s2x; q2y = s2y; q1x = cx; q1y = cy; }
  d = point_segment(e2x, e2y, s1x, s1y, e1x, e1y, cx, cy);
  if (d < best) { best = d; q2x = e2x; q2y = e2y; q1x = cx; q1y = cy; }
  return best;
}

// ------------------------------------------------------------------------------------------------------
// Generic vertex-list shapes. k = 1 point, k = 2 segment (open), k > 2 closed polygon.
// ------------------------------------------------------------------------------------------------------
struct RobotShape {   // world vertex i = (px,py) + R(theta) * body_i ; k==0 means "the point (px,py)" itself
  int k;
  double px, py, cs, sn;
  const double* bvx;   // body-frame vertices (kernarg / constant memory)
  const double* bvy;
  __device__ __forceinline__ void vertex(int i, double& x, double& y) const {
    if (k == 0) { x = px; y = py; return; }
    double bx = bvx[i], by = bvy[i];
    x = px + cs * bx - sn * by;     // robot_footprint_model.h:610-618, 757-766
    y = py + sn * bx + cs * by;
  }
  __device__ __forceinline__ int count() const { return k == 0 ? 1 : k; }
};

struct ObstShape {    // vertex i = stored vertex + (ox,oy) (constant-velocity prediction, obstacles.h:1023-1031)
  int k;
  double ax, ay, bx, by;   // used when verts == nullptr (point / circle / line / pill)
  const double* vx;
  const double* vy;
  double ox, oy;
  __device__ __forceinline__ void vertex(int i, double& x, double& y) const {
    if (vx) { x = vx[i] + ox; y = vy[i] + oy; }
    else if (i == 0) { x = ax + ox; y = ay + oy; }
    else { x = bx + ox; y = by + oy; }
  }
};

// distance_point_to_polygon_2d (:168-193) with p = robot point
__device__ inline Wit robot_point_to_obst(double px, double py, const ObstShape& O) {
  Wit w;
  w.p1x = px; w.p1y = py;
  double ax, ay, bx, by, cx, cy;
  if (O.k == 1) {
    O.vertex(0, ax, ay);
    w.d = nrm2(px - ax, py - ay);
    w.p2x = ax; w.p2y = ay;
    return w;
  }
  w.d = HUGE_VAL; w.p2x = px; w.p2y = py;
  O.vertex(0, ax, ay);
  double fx = ax, fy = ay;
  for (int i = 0; i < O.k - 1; ++i) {
    O.vertex(i + 1, bx, by);
    double d = point_segment(px, py, ax, ay, bx, by, cx, cy);
    if (d < w.d) { w.d = d; w.p2x = cx; w.p2y = cy; }
    ax = bx; ay = by;
  }
  if (O.k > 2) {
    double d = point_segment(px, py, ax, ay, fx, fy, cx, cy);   // back -> front
    if (d < w.d) { w.d = d; w.p2x = cx; w.p2y = cy; }
  }
  return w;
}

// one robot edge (r1 -> r2) against the obstacle shape: distance_segment_to_polygon_2d (:203-229).
// obst_first: the reference passes the obstacle segment as the FIRST argument of
// distance_segment_to_segment_2d for Line/Pill obstacles (obstacles.h:658-661, 805-808, 663-666, 810-813).
__device__ inline void robot_edge_to_obst(double r1x, double r1y, double r2x, double r2y, const ObstShape& O,
                                          bool obst_first, Wit& best /* in: best.d = HUGE_VAL */) {
  double ax, ay, bx, by, q1x, q1y, q2x, q2y;
  if (O.k == 1) {
    O.vertex(0, ax, ay);
    double cx, cy;
    best.d = point_segment(ax, ay, r1x, r1y, r2x, r2y, cx, cy);
    best.p1x = cx; best.p1y = cy; best.p2x = ax; best.p2y = ay;
    return;
  }
  O.vertex(0, ax, ay);
  double fx = ax, fy = ay;
  for (int j = 0; j < O.k - 1; ++j) {
    O.vertex(j + 1, bx, by);
    double d;
    if (obst_first) d = segment_segment(ax, ay, bx, by, r1x, r1y, r2x, r2y, q2x, q2y, q1x, q1y);
    else d = segment_segment(r1x, r1y, r2x, r2y, ax, ay, bx, by, q1x, q1y, q2x, q2y);
    if (d < best.d) { best.d = d; best.p1x = q1x; best.p1y = q1y; best.p2x = q2x; best.p2y = q2y; }
    ax = bx; ay = by;
  }
  if (O.k > 2) {
    double d;
    if (obst_first) d = segment_segment(ax, ay, fx, fy, r1x, r1y, r2x, r2y, q2x, q2y, q1x, q1y);
    else d = segment_segment(r1x, r1y, r2x, r2y, ax, ay, fx, fy, q1x, q1y, q2x, q2y);
    if (d < best.d) { best.d = d; best.p1x = q1x; best.p1y = q1y; best.p2x = q2x; best.p2y = q2y; }
  }
}

// distance_polygon_to_polygon_2d (:237-262) with polygon 1 = robot. For Line/Pill obstacles against a
// robot polygon the reference loops the ROBOT edges too (distance_segment_to_polygon_2d(obst, robot)),
// so one loop structure covers every case.
__device__ inline Wit robot_shape_to_obst(const RobotShape& R, const ObstShape& O, bool obst_first) {
  const int kr = R.count();
  double r1x, r1y, r2x, r2y;
  if (kr == 1) {
    R.vertex(0, r1x, r1y);
    return robot_point_to_obst(r1x, r1y, O);
  }
  Wit best;
  best.d = HUGE_VAL; best.p1x = best.p1y = best.p2x = best.p2y = 0;
  R.vertex(0, r1x, r1y);
  double fx = r1x, fy = r1y;
  for (int i = 0; i < kr - 1; ++i) {
    R.vertex(i + 1, r2x, r2y);
    Wit w; w.d = HUGE_VAL; w.p1x = w.p1y = w.p2x = w.p2y = 0;
    robot_edge_to_obst(r1x, r1y, r2x, r2y, O, obst_first, w);
    if (w.d < best.d) best = w;
    r1x = r2x; r1y = r2y;
  }
  if (kr > 2) {
    Wit w; w.d = HUGE_VAL; w.p1x = w.p1y = w.p2x = w.p2y = 0;
    robot_edge_to_obst(r1x, r1y, fx, fy, O, obst_first, w);
    if (w.d < best.d) best = w;
  }
  return best;
}

// ------------------------------------------------------------------------------------------------------
// BaseRobotFootprintModel::calculateDistance / estimateSpatioTemporalDistance for every
// footprint x obstacle pair. Returns the distance; if grad != nullptr also d(dist)/d(x,y,theta).
// ------------------------------------------------------------------------------------------------------
__device__ inline double footprint_distance(const teb_amd_config_t& c, const SceneDev& sc, int oi, double x, double y,
                                            double cth, double sth, bool spatio_temporal, double t, double* grad) {
  // cth, sth = cos(theta), sin(theta) of the pose (orientationUnitVec / transformToWorld)
  const int ty = sc.type[oi];
  ObstShape O;
  O.ox = 0; O.oy = 0;
  if (spatio_temporal) { O.ox = t * sc.vx[oi]; O.oy = t * sc.vy[oi]; }
  O.vx = nullptr; O.vy = nullptr;
  O.ax = sc.ax[oi]; O.ay = sc.ay[oi]; O.bx = 0; O.by = 0;
  double obst_r = 0;
  bool obst_first = false;
  if (ty == TEB_AMD_OBST_POINT) O.k = 1;
  else if (ty == TEB_AMD_OBST_CIRCULAR) { O.k = 1; obst_r = sc.rad[oi]; }
  else if (ty == TEB_AMD_OBST_LINE || ty == TEB_AMD_OBST_PILL) {
    O.k = 2; O.bx = sc.bx[oi]; O.by = sc.by[oi];
    obst_first = true;
    if (ty == TEB_AMD_OBST_PILL) obst_r = sc.rad[oi];
  } else {
    int o0 = sc.voff[oi];
    O.k = sc.voff[oi + 1] - o0;
    O.vx = sc.pvx + o0; O.vy = sc.pvy + o0;
  }
  Wit w;
  double dist;   // same subtraction order as the reference: (geometric distance - obstacle radius) - robot radius
  const int fp = c.footprint_type;
  if (fp == TEB_AMD_FOOTPRINT_POINT || fp == TEB_AMD_FOOTPRINT_CIRCULAR) {
    w = robot_point_to_obst(x, y, O);
    dist = w.d - obst_r;
    if (fp == TEB_AMD_FOOTPRINT_CIRCULAR) dist = dist - c.footprint_radius;
  } else if (fp == TEB_AMD_FOOTPRINT_TWO_CIRCLES) {
    double dx = cth, dy = sth;
    Wit wf = robot_point_to_obst(x + c.footprint_front_offset * dx, y + c.footprint_front_offset * dy, O);
    Wit wr = robot_point_to_obst(x - c.footprint_rear_offset * dx, y - c.footprint_rear_offset * dy, O);
    double df = (wf.d - obst_r) - c.footprint_front_radius;
    double dr = (wr.d - obst_r) - c.footprint_rear_radius;
    // std::min(dist_front, dist_rear): rear only if strictly smaller
    if (dr < df) { w = wr; dist = dr; } else { w = wf; dist = df; }
  } else {
    RobotShape R;
    R.k = c.footprint_n_vertices;
    R.px = x; R.py = y; R.cs = cth; R.sn = sth;
    R.bvx = c.footprint_vx; R.bvy = c.footprint_vy;
    w = robot_shape_to_obst(R, O, obst_first);
    dist = w.d - obst_r;
  }
  if (grad) {
    double vx_ = w.p1x - w.p2x, vy_ = w.p1y - w.p2y;
    double dn = nrm2(vx_, vy_);
    if (dn > 0) {
      double nx = vx_ / dn, ny = vy_ / dn;
      double lx = w.p1x - x, ly = w.p1y - y;
      grad[0] = nx; grad[1] = ny; grad[2] = -nx * ly + ny * lx;
    } else {
      grad[0] = 0; grad[1] = 0; grad[2] = 0;
    }
  }
  return dist;
}

}  // namespace tebamd
