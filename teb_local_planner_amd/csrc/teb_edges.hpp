// teb_edges.hpp — device library: residuals and closed-form Jacobian rows of the TEB cost terms.
//
// Residuals follow the reference edge classes exactly:
//   g2o_types/edge_velocity.h:85-117, 232-271   edge_acceleration.h:88-149, 303-345, 394-437, 484-534, 577-620, 668-712
//   edge_kinematics.h:87-103, 196-216           edge_obstacle.h:90-103, 212-229   edge_dynamic_obstacle.h:93-104
//   edge_time_optimal.h:93   edge_via_point.h:86   edge_shortest_path.h:78   edge_prefer_rotdir.h:95-103
//   edge_velocity_obstacle_ratio.h:79-121       penalties.h:57-187   misc.h:95-98
// Jacobian conventions (TEB_AMD_JACOBIAN_ANALYTIC): penalty derivatives exactly as penalties.h:127-187
// (-1 / 0 / +1 with the same branch tests), d||v||/dv := 0 at v = 0, normalize_theta has derivative 1,
// fabs'(0) := g2o::sign(0) = 0 (as the reference's live analytic EdgeKinematicsDiffDrive Jacobian).
//
// A "row" is one residual component with its gradient w.r.t. the thread-local window of 11 scalars:
//   [x_i y_i th_i dt_i | x_{i+1} y_{i+1} th_{i+1} dt_{i+1} | x_{i+2} y_{i+2} th_{i+2}]   (local index 0..10)
#pragma once
#include "teb_geometry.hpp"

namespace tebamd {

__device__ __forceinline__ double normalize_theta(double theta) {   // g2o/stuff/misc.h
  if (theta >= -M_PI && theta < M_PI) return theta;
  double multiplier = floor(theta / (2 * M_PI));
  theta = theta - multiplier * 2 * M_PI;
  if (theta >= M_PI) theta -= 2 * M_PI;
  if (theta < -M_PI) theta += 2 * M_PI;
  return theta;
}
__device__ __forceinline__ double sgn(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0); }
__device__ __forceinline__ double fast_sigmoid(double x) { return x / (1 + fabs(x)); }   // misc.h:95-98

// penalties.h:57-117 and their derivatives :127-187
__device__ __forceinline__ double pen_interval(double var, double a, double eps, double& dev) {
  if (var < -a + eps) { dev = -1; return (-var - (a - eps)); }
  if (var <= a - eps) { dev = 0; return 0.; }
  dev = 1;
  return (var - (a - eps));
}
__device__ __forceinline__ double pen_interval2(double var, double a, double b, double eps, double& dev) {
  if (var < a + eps) { dev = -1; return (-var + (a + eps)); }
  if (var <= b - eps) { dev = 0; return 0.; }
  dev = 1;
  return (var - (b - eps));
}
__device__ __forceinline__ double pen_below(double var, double a, double eps, double& dev) {
  if (var >= a + eps) { dev = 0; return 0.; }
  dev = -1;
  return (-var + (a + eps));
}

// v = dist/dt * fast_sigmoid(100 * deltaS.(cos th_a, sin th_a)), omega = normalize(th_b - th_a)/dt.
// dv[7] = dv/d(xa, ya, tha, xb, yb, thb, dt) when JAC.
template <bool JAC>
__device__ __forceinline__ void signed_velocity(const teb_amd_config_t& c, double xa, double ya, double tha, double ca,
                                                double sa, double xb, double yb, double thb, double dt, double& v,
                                                double& omega, double* dv) {
  double dx = xb - xa, dy = yb - ya;
  double eucl = sqrt(dx * dx + dy * dy);
  double dist = eucl;
  const double angle_diff = normalize_theta(thb - tha);
  bool arc = false;
  if (TEB_CFGI(EXACT_ARC) && angle_diff != 0) {
    double radius = dist / (2 * sin(angle_diff / 2));
    dist = fabs(angle_diff * radius);
    arc = true;
  }
  double p = dx * ca + dy * sa;   // ca, sa = cos(tha), sin(tha) (per-pose LDS cache)
  double sg = fast_sigmoid(100 * p);
  double vel = dist / dt;
  vel *= sg;
  v = vel;
  omega = angle_diff / dt;
  if (JAC) {
    double k = 1.0, kp = 0.0;
    if (arc) {
      double h = angle_diff / 2, sh = sin(h), ch = cos(h);
      k = angle_diff / (2 * sh);
      kp = 1.0 / (2 * sh) - angle_diff * ch / (4 * sh * sh);
      if (k < 0) { k = -k; kp = -kp; }
    }
    double ddx = 0, ddy = 0;
    if (eucl > 0) { ddx = dx / eucl; ddy = dy / eucl; }
    double a100 = 1 + fabs(100 * p);
    double sp = 100.0 / (a100 * a100);
    // d(dist) and d(p) w.r.t. (xa, ya, tha, xb, yb, thb)
    dv[0] = ((-k * ddx) * sg + dist * sp * (-ca)) / dt;
    dv[1] = ((-k * ddy) * sg + dist * sp * (-sa)) / dt;
    dv[2] = ((-eucl * kp) * sg + dist * sp * (-dx * sa + dy * ca)) / dt;
    dv[3] = ((k * ddx) * sg + dist * sp * (ca)) / dt;
    dv[4] = ((k * ddy) * sg + dist * sp * (sa)) / dt;
    dv[5] = ((eucl * kp) * sg) / dt;
    dv[6] = -vel / dt;
  }
}

// The per-thread accumulator of J^T Omega J (lower triangle of the 11x11 window) and J^T Omega e,
// plus chi^2 per cost category. Everything is fully unrolled so it lives in VGPRs.
enum { CAT_OBST = 0, CAT_VIA = 1, CAT_TIME = 2, CAT_OTHER = 3 };

struct Accum {
  double H[66];   // H[a*(a+1)/2 + b], b <= a
  double g[11];   // gradient accumulator: sum J^T Omega e   (b = -g)
  double chi[4];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int q = 0; q < 66; ++q) H[q] = 0;
#pragma unroll
    for (int q = 0; q < 11; ++q) g[q] = 0;
  }
  __device__ __forceinline__ void clear_chi() { chi[0] = chi[1] = chi[2] = chi[3] = 0; }
  // one residual row: e, information w, gradient row r (only columns with bit set in MASK can be non-zero)
  template <unsigned MASK, bool JAC>
  __device__ __forceinline__ void row(int cat, double e, double w, const double* r) {
    // The accumulation (and only it) uses fused multiply-adds: the residual e and the Jacobian row r - everything that decides on
    // which side of a penalty kink a value falls - are computed by the caller under -ffp-contract=off; what is summed here has no
    // bit-level counterpart in the reference (g2o adds the edges in another order). The fusion is spelled out (fma(), not a contraction
    // pragma that leaves the choice to the optimiser): every instantiation that replays a row - single-CU, multi-CU, any layout - rounds
    // it the same way. -DTEB_AMD_NO_ROW_FMA: separate mul / add (C4 with 200 fixed poses + 5 %, C2 / C3 + 3 %, headline + 2 %).
    chi[cat] = fma(e, w * e, chi[cat]);
    if (JAC) {
#pragma unroll
      for (int a = 0; a < 11; ++a) {
        if (!((MASK >> a) & 1u)) continue;
        const double ra = r[a] * w;
        g[a] = fma(ra, e, g[a]);
#pragma unroll
        for (int b = 0; b <= a; ++b) {
          if (!((MASK >> b) & 1u)) continue;
          H[a * (a + 1) / 2 + b] = fma(ra, r[b], H[a * (a + 1) / 2 + b]);
        }
      }
    }
  }
};

// window column masks
constexpr unsigned M_POSE0 = 0x007;            // x_i y_i th_i
constexpr unsigned M_DT0 = 0x008;              // dt_i
constexpr unsigned M_POSE1 = 0x070;            // pose i+1
constexpr unsigned M_DT1 = 0x080;              // dt_{i+1}
constexpr unsigned M_POSE2 = 0x700;            // pose i+2
constexpr unsigned M_SEG = M_POSE0 | M_DT0 | M_POSE1;   // velocity-type edges
constexpr unsigned M_ALL = 0x7FF;

struct Win {   // thread-local copy of the window state (+ cached cos/sin of theta_i, theta_{i+1})
  double x0, y0, t0, d0, x1, y1, t1, d1, x2, y2, t2;
  double c0, s0, c1, s1;
};

// ---- g2o numeric differentiation (TEB_AMD_JACOBIAN_G2O_NUMERIC) ----------------------------------------------------
// Base{Unary,Binary,Multi}Edge::linearizeOplus of libg2o: per vertex dimension d, push; oplus(+delta e_d); computeError;
// pop; push; oplus(-delta e_d); computeError; pop; column = (1 / (2 delta)) * (e+ - e-), delta = 1e-9.
// RowRec stands in for Accum while an edge is only evaluated: it records the residual rows in order.
struct RowRec {
  double e[3], wgt[3];
  int n;
  __device__ __forceinline__ RowRec() : n(0) {}
  template <unsigned MASK, bool JAC>
  __device__ __forceinline__ void row(int, double e_, double w_, const double*) {
    e[n] = e_; wgt[n] = w_; ++n;
  }
};

// VertexPose::oplusImpl / VertexTimeDiff::oplusImpl on window column Q (x += d, y += d, theta = normalize_theta(theta + d),
// dt += d; vertex_pose.h:195-198, vertex_timediff.h:113-116); the cached cos/sin follow theta.
template <int Q>
__device__ __forceinline__ Win oplus_column(const Win& w, double d) {
  Win p = w;
  if (Q == 0) p.x0 += d;
  if (Q == 1) p.y0 += d;
  if (Q == 2) { p.t0 = normalize_theta(p.t0 + d); p.c0 = cos(p.t0); p.s0 = sin(p.t0); }
  if (Q == 3) p.d0 += d;
  if (Q == 4) p.x1 += d;
  if (Q == 5) p.y1 += d;
  if (Q == 6) { p.t1 = normalize_theta(p.t1 + d); p.c1 = cos(p.t1); p.s1 = sin(p.t1); }
  if (Q == 7) p.d1 += d;
  if (Q == 8) p.x2 += d;
  if (Q == 9) p.y2 += d;
  if (Q == 10) p.t2 = normalize_theta(p.t2 + d);
  return p;
}

template <unsigned VMASK, int Q, class F>
__device__ __forceinline__ void numeric_column(const Win& w, F& f, double (*J)[11]) {
  if ((VMASK >> Q) & 1u) {
    const double delta = 1e-9;
    const double scalar = 1.0 / (2 * delta);
    RowRec ep, em;
    f(oplus_column<Q>(w, delta), ep);
    f(oplus_column<Q>(w, -delta), em);
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (k < ep.n) J[k][Q] = scalar * (ep.e[k] - em.e[k]);
  }
}

// One edge in numeric mode: f(window, accumulator) evaluates the edge's residual rows (error only). VMASK = the window
// columns of the edge's vertices, CAT its cost category. The rows then enter the accumulator exactly like analytic rows.
template <unsigned VMASK, int CAT, class F>
__device__ __forceinline__ void numeric_edge(const Win& w, Accum& A, F f) {
  RowRec base;
  f(w, base);
  double J[3][11];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int q = 0; q < 11; ++q) J[k][q] = 0;
  numeric_column<VMASK, 0>(w, f, J); numeric_column<VMASK, 1>(w, f, J); numeric_column<VMASK, 2>(w, f, J);
  numeric_column<VMASK, 3>(w, f, J); numeric_column<VMASK, 4>(w, f, J); numeric_column<VMASK, 5>(w, f, J);
  numeric_column<VMASK, 6>(w, f, J); numeric_column<VMASK, 7>(w, f, J); numeric_column<VMASK, 8>(w, f, J);
  numeric_column<VMASK, 9>(w, f, J); numeric_column<VMASK, 10>(w, f, J);
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (k < base.n) A.template row<VMASK, true>(CAT, base.e[k], base.wgt[k], J[k]);
}

// ---- EdgeVelocity --------------------------------------------------------------------------------------
template <bool JAC, class ACC>
__device__ __forceinline__ void edge_velocity(const teb_amd_config_t& c, const Win& w, ACC& A) {
  double v, om, dv[7];
  signed_velocity<JAC>(c, w.x0, w.y0, w.t0, w.c0, w.s0, w.x1, w.y1, w.t1, w.d0, v, om, dv);
  double sv, sw;
  double e0 = pen_interval2(v, -c.max_vel_x_backwards, c.max_vel_x, c.penalty_epsilon, sv);
  double e1 = pen_interval(om, c.max_vel_theta, c.penalty_epsilon, sw);
  double r[11];
  if (JAC) {
#pragma unroll
    for (int q = 0; q < 11; ++q) r[q] = 0;
    r[0] = sv * dv[0]; r[1] = sv * dv[1]; r[2] = sv * dv[2]; r[4] = sv * dv[3]; r[5] = sv * dv[4]; r[6] = sv * dv[5];
    r[3] = sv * dv[6];
  }
  A.template row<M_SEG, JAC>(CAT_OTHER, e0, c.weight_max_vel_x, r);
  if (JAC) {
#pragma unroll
    for (int q = 0; q < 11; ++q) r[q] = 0;
    r[2] = -sw / w.d0; r[6] = sw / w.d0; r[3] = -sw * om / w.d0;
  }
  A.template row<0x04C, JAC>(CAT_OTHER, e1, c.weight_max_vel_theta, r);
}

// ---- EdgeVelocityHolonomic -----------------------------------------------------------------------------
template <bool JAC, class ACC>
__device__ __forceinline__ void edge_velocity_holonomic(const teb_amd_config_t& c, const Win& w, ACC& A) {
  double dtv = w.d0;
  double dx = w.x1 - w.x0, dy = w.y1 - w.y0;
  double c1 = w.c0, s1 = w.s0;
  double r_dx = c1 * dx + s1 * dy, r_dy = -s1 * dx + c1 * dy;
  double vx = r_dx / dtv, vy = r_dy / dtv;
  double omega = normalize_theta(w.t1 - w.t0) / dtv;
  double vt2 = c.max_vel_trans * c.max_vel_trans;
  double rem_y = sqrt(fmax(0.0, vt2 - vx * vx));
  double rem_x = sqrt(fmax(0.0, vt2 - vy * vy));
  // std::min(a, b) returns b only if b < a
  bool y_cfg = c.max_vel_y < rem_y, x_cfg = c.max_vel_x < rem_x, xb_cfg = c.max_vel_x_backwards < rem_x;
  double max_vel_y = y_cfg ? c.max_vel_y : rem_y;
  double max_vel_x = x_cfg ? c.max_vel_x : rem_x;
  double max_vel_xb = xb_cfg ? c.max_vel_x_backwards : rem_x;
  double s0, s1_, s2;
  double e0 = pen_interval2(vx, -max_vel_xb, max_vel_x, 0.0, s0);
  double e1 = pen_interval(vy, max_vel_y, 0.0, s1_);
  double e2 = pen_interval(omega, c.max_vel_theta, c.penalty_epsilon, s2);
  double r[11];
  // local columns of (xa, ya, tha, xb, yb, thb, dt) = 0,1,2,4,5,6,3
  double gvx[7] = {-c1 / dtv, -s1 / dtv, r_dy / dtv, c1 / dtv, s1 / dtv, 0, -vx / dtv};
  double gvy[7] = {s1 / dtv, -c1 / dtv, -r_dx / dtv, -s1 / dtv, c1 / dtv, 0, -vy / dtv};
  const int col[7] = {0, 1, 2, 4, 5, 6, 3};
  if (JAC) {
    double drem_y = (vt2 - vx * vx > 0 && rem_y > 0) ? -vx / rem_y : 0.0;
    double drem_x = (vt2 - vy * vy > 0 && rem_x > 0) ? -vy / rem_x : 0.0;
#pragma unroll
    for (int q = 0; q < 11; ++q) r[q] = 0;
    if (s0 < 0) {
#pragma unroll
      for (int q = 0; q < 7; ++q) r[col[q]] = -gvx[q] + (!xb_cfg ? -drem_x * gvy[q] : 0.0);
    } else if (s0 > 0) {
#pragma unroll
      for (int q = 0; q < 7; ++q) r[col[q]] = gvx[q] - (!x_cfg ? drem_x * gvy[q] : 0.0);
    }
    A.template row<M_SEG, JAC>(CAT_OTHER, e0, c.weight_max_vel_x, r);
#pragma unroll
    for (int q = 0; q < 11; ++q) r[q] = 0;
    if (s1_ < 0) {
#pragma unroll
      for (int q = 0; q < 7; ++q) r[col[q]] = -gvy[q] - (!y_cfg ? drem_y * gvx[q] : 0.0);
    } else if (s1_ > 0) {
#pragma unroll
      for (int q = 0; q < 7; ++q) r[col[q]] = gvy[q] - (!y_cfg ? drem_y * gvx[q] : 0.0);
    }
    A.template row<M_SEG, JAC>(CAT_OTHER, e1, c.weight_max_vel_y, r);
#pragma unroll
    for (int q = 0; q < 11; ++q) r[q] = 0;
    r[2] = -s2 / dtv; r[6] = s2 / dtv; r[3] = -s2 * omega / dtv;
    A.template row<0x04C, JAC>(CAT_OTHER, e2, c.weight_max_vel_theta, r);
  } else {
    A.template row<M_SEG, false>(CAT_OTHER, e0, c.weight_max_vel_x, r);
    A.template row<M_SEG, false>(CAT_OTHER, e1, c.weight_max_vel_y, r);
    A.template row<M_SEG, false>(CAT_OTHER, e2, c.weight_max_vel_theta, r);
  }
}

// ---- EdgeAcceleration (poses i, i+1, i+2; dt_i, dt_{i+1}) -------------------------------------------------
template <bool JAC, class ACC>
__device__ __forceinline__ void edge_acceleration(const teb_amd_config_t& c, const Win& w, ACC& A) {
  double v1, o1, v2, o2, d1[7], d2[7];
  signed_velocity<JAC>(c, w.x0, w.y0, w.t0, w.c0, w.s0, w.x1, w.y1, w.t1, w.d0, v1, o1, d1);
  signed_velocity<JAC>(c, w.x1, w.y1, w.t1, w.c1, w.s1, w.x2, w.y2, w.t2, w.d1, v2, o2, d2);
  double T = w.d0 + w.d1;
  const double acc_lin = (v2 - v1) * 2 / T;
  const double acc_rot = (o2 - o1) * 2 / T;
  double sa, sr;
  double e0 = pen_interval(acc_lin, c.acc_lim_x, c.penalty_epsilon, sa);
  double e1 = pen_interval(acc_rot, c.acc_lim_theta, c.penalty_epsilon, sr);
  double r[11];
  if (JAC) {
    double f = 2 / T;
    // v1: (x0,y0,t0,x1,y1,t1,dt0) -> local 0,1,2,4,5,6,3 ; v2: (x1,y1,t1,x2,y2,t2,dt1) -> 4,5,6,8,9,10,7
    r[0] = sa * f * (-d1[0]); r[1] = sa * f * (-d1[1]); r[2] = sa * f * (-d1[2]);
    r[4] = sa * f * (d2[0] - d1[3]); r[5] = sa * f * (d2[1] - d1[4]); r[6] = sa * f * (d2[2] - d1[5]);
    r[8] = sa * f * (d2[3]); r[9] = sa * f * (d2[4]); r[10] = sa * f * (d2[5]);
    r[3] = sa * (f * (-d1[6]) - acc_lin / T);
    r[7] = sa * (f * (d2[6]) - acc_lin / T);
  }
  A.template row<M_ALL, JAC>(CAT_OTHER, e0, c.weight_acc_lim_x, r);
  if (JAC) {
    double f = 2 / T;
#pragma unroll
    for (int q = 0; q < 11; ++q) r[q] = 0;
    r[2] = sr * f * (1.0 / w.d0);
    r[6] = sr * f * (-1.0 / w.d1 - 1.0 / w.d0);
    r[10] = sr * f * (1.0 / w.d1);
    r[3] = sr * (f * (o1 / w.d0) - acc_rot / T);
    r[7] = sr * (f * (-o2 / w.d1) - acc_rot / T);
  }
  A.template row<0x4CC, JAC>(CAT_OTHER, e1, c.weight_acc_lim_theta, r);
}

// ---- EdgeAccelerationStart / Goal: segment (pose a = window pose 0, pose b = window pose 1, dt0) ---------
template <bool JAC, bool START, class ACC>
__device__ __forceinline__ void edge_acceleration_se(const teb_amd_config_t& c, const Win& w, double vlin,
                                                     double vang, ACC& A) {
  double v, om, dv[7];
  double dtv = w.d0;
  signed_velocity<JAC>(c, w.x0, w.y0, w.t0, w.c0, w.s0, w.x1, w.y1, w.t1, dtv, v, om, dv);
  double acc_lin, acc_rot;
  if (START) { acc_lin = (v - vlin) / dtv; acc_rot = (om - vang) / dtv; }
  else { acc_lin = (vlin - v) / dtv; acc_rot = (vang - om) / dtv; }
  double sa, sr;
  double e0 = pen_interval(acc_lin, c.acc_lim_x, c.penalty_epsilon, sa);
  double e1 = pen_interval(acc_rot, c.acc_lim_theta, c.penalty_epsilon, sr);
  double r[11];
  const double sg = START ? 1.0 : -1.0;
  if (JAC) {
#pragma unroll
    for (int q = 0; q < 11; ++q) r[q] = 0;
    r[0] = sa * sg * dv[0] / dtv; r[1] = sa * sg * dv[1] / dtv; r[2] = sa * sg * dv[2] / dtv;
    r[4] = sa * sg * dv[3] / dtv; r[5] = sa * sg * dv[4] / dtv; r[6] = sa * sg * dv[5] / dtv;
    r[3] = sa * (sg * dv[6] / dtv - acc_lin / dtv);
  }
  A.template row<M_SEG, JAC>(CAT_OTHER, e0, c.weight_acc_lim_x, r);
  if (JAC) {
#pragma unroll
    for (int q = 0; q < 11; ++q) r[q] = 0;
    r[2] = sr * sg * (-1.0 / (dtv * dtv));
    r[6] = sr * sg * (1.0 / (dtv * dtv));
    r[3] = sr * (sg * (-om / dtv) / dtv - acc_rot / dtv);
  }
  A.template row<0x04C, JAC>(CAT_OTHER, e1, c.weight_acc_lim_theta, r);
}

// ---- holonomic accelerations -----------------------------------------------------------------------------
template <bool JAC, class ACC>
__device__ __forceinline__ void edge_acceleration_holonomic(const teb_amd_config_t& c, const Win& w, ACC& A) {
  double dt1 = w.d0, dt2 = w.d1;
  double d1x = w.x1 - w.x0, d1y = w.y1 - w.y0, d2x = w.x2 - w.x1, d2y = w.y2 - w.y1;
  double c1 = w.c0, s1 = w.s0, c2 = w.c1, s2 = w.s1;
  double p1_dx = c1 * d1x + s1 * d1y, p1_dy = -s1 * d1x + c1 * d1y;
  double p2_dx = c2 * d2x + s2 * d2y, p2_dy = -s2 * d2x + c2 * d2y;
  double v1x = p1_dx / dt1, v1y = p1_dy / dt1, v2x = p2_dx / dt2, v2y = p2_dy / dt2;
  double T = dt1 + dt2;
  double acc_x = (v2x - v1x) * 2 / T, acc_y = (v2y - v1y) * 2 / T;
  double om1 = normalize_theta(w.t1 - w.t0) / dt1, om2 = normalize_theta(w.t2 - w.t1) / dt2;
  double acc_rot = (om2 - om1) * 2 / T;
  double sx, sy, sr;
  double e0 = pen_interval(acc_x, c.acc_lim_x, c.penalty_epsilon, sx);
  double e1 = pen_interval(acc_y, c.acc_lim_y, c.penalty_epsilon, sy);
  double e2 = pen_interval(acc_rot, c.acc_lim_theta, c.penalty_epsilon, sr);
  double r[11];
  if (JAC) {
    double f = 2 / T;
    double g1x[11], g1y[11], g2x[11], g2y[11];
#pragma unroll
    for (int q = 0; q < 11; ++q) { g1x[q] = 0; g1y[q] = 0; g2x[q] = 0; g2y[q] = 0; }
    g1x[0] = -c1 / dt1; g1x[1] = -s1 / dt1; g1x[2] = p1_dy / dt1; g1x[4] = c1 / dt1; g1x[5] = s1 / dt1; g1x[3] = -v1x / dt1;
    g1y[0] = s1 / dt1; g1y[1] = -c1 / dt1; g1y[2] = -p1_dx / dt1; g1y[4] = -s1 / dt1; g1y[5] = c1 / dt1; g1y[3] = -v1y / dt1;
    g2x[4] = -c2 / dt2; g2x[5] = -s2 / dt2; g2x[6] = p2_dy / dt2; g2x[8] = c2 / dt2; g2x[9] = s2 / dt2; g2x[7] = -v2x / dt2;
    g2y[4] = s2 / dt2; g2y[5] = -c2 / dt2; g2y[6] = -p2_dx / dt2; g2y[8] = -s2 / dt2; g2y[9] = c2 / dt2; g2y[7] = -v2y / dt2;
#pragma unroll
    for (int q = 0; q < 11; ++q) {
      double extra = (q == 3 || q == 7) ? 1.0 : 0.0;
      r[q] = sx * (f * (g2x[q] - g1x[q]) - extra * acc_x / T);
    }
    A.template row<M_ALL, JAC>(CAT_OTHER, e0, c.weight_acc_lim_x, r);
#pragma unroll
    for (int q = 0; q < 11; ++q) {
      double extra = (q == 3 || q == 7) ? 1.0 : 0.0;
      r[q] = sy * (f * (g2y[q] - g1y[q]) - extra * acc_y / T);
    }
    A.template row<M_ALL, JAC>(CAT_OTHER, e1, c.weight_acc_lim_y, r);
#pragma unroll
    for (int q = 0; q < 11; ++q) r[q] = 0;
    r[2] = sr * f * (1.0 / dt1);
    r[6] = sr * f * (-1.0 / dt2 - 1.0 / dt1);
    r[10] = sr * f * (1.0 / dt2);
    r[3] = sr * (f * (om1 / dt1) - acc_rot / T);
    r[7] = sr * (f * (-om2 / dt2) - acc_rot / T);
    A.template row<0x4CC, JAC>(CAT_OTHER, e2, c.weight_acc_lim_theta, r);
  } else {
    A.template row<M_ALL, false>(CAT_OTHER, e0, c.weight_acc_lim_x, r);
    A.template row<M_ALL, false>(CAT_OTHER, e1, c.weight_acc_lim_y, r);
    A.template row<M_ALL, false>(CAT_OTHER, e2, c.weight_acc_lim_theta, r);
  }
}

template <bool JAC, bool START, class ACC>
__device__ __forceinline__ void edge_acceleration_holonomic_se(const teb_amd_config_t& c, const Win& w,
                                                               const double* vel /* vx, vy, omega */, ACC& A) {
  double dtv = w.d0;
  double dx = w.x1 - w.x0, dy = w.y1 - w.y0;
  double c1 = w.c0, s1 = w.s0;
  double pdx = c1 * dx + s1 * dy, pdy = -s1 * dx + c1 * dy;
  double vx = pdx / dtv, vy = pdy / dtv;
  double om = normalize_theta(w.t1 - w.t0) / dtv;
  double ax, ay, ar;
  if (START) { ax = (vx - vel[0]) / dtv; ay = (vy - vel[1]) / dtv; ar = (om - vel[2]) / dtv; }
  else { ax = (vel[0] - vx) / dtv; ay = (vel[1] - vy) / dtv; ar = (vel[2] - om) / dtv; }
  double sx, sy, sr;
  double e0 = pen_interval(ax, c.acc_lim_x, c.penalty_epsilon, sx);
  double e1 = pen_interval(ay, c.acc_lim_y, c.penalty_epsilon, sy);
  double e2 = pen_interval(ar, c.acc_lim_theta, c.penalty_epsilon, sr);
  double r[11];
  if (JAC) {
    const double sg = START ? 1.0 : -1.0;
    double gvx[7] = {-c1 / dtv, -s1 / dtv, pdy / dtv, c1 / dtv, s1 / dtv, 0, -vx / dtv};
    double gvy[7] = {s1 / dtv, -c1 / dtv, -pdx / dtv, -s1 / dtv, c1 / dtv, 0, -vy / dtv};
    double gom[7] = {0, 0, -1 / dtv, 0, 0, 1 / dtv, -om / dtv};
    const int col[7] = {0, 1, 2, 4, 5, 6, 3};
#pragma unroll
    for (int q = 0; q < 11; ++q) r[q] = 0;
#pragma unroll
    for (int q = 0; q < 7; ++q) r[col[q]] = sx * (sg * gvx[q] / dtv - (q == 6 ? ax / dtv : 0.0));
    A.template row<M_SEG, JAC>(CAT_OTHER, e0, c.weight_acc_lim_x, r);
#pragma unroll
    for (int q = 0; q < 7; ++q) r[col[q]] = sy * (sg * gvy[q] / dtv - (q == 6 ? ay / dtv : 0.0));
    A.template row<M_SEG, JAC>(CAT_OTHER, e1, c.weight_acc_lim_y, r);
#pragma unroll
    for (int q = 0; q < 7; ++q) r[col[q]] = sr * (sg * gom[q] / dtv - (q == 6 ? ar / dtv : 0.0));
    A.template row<M_SEG, JAC>(CAT_OTHER, e2, c.weight_acc_lim_theta, r);
  } else {
    A.template row<M_SEG, false>(CAT_OTHER, e0, c.weight_acc_lim_x, r);
    A.template row<M_SEG, false>(CAT_OTHER, e1, c.weight_acc_lim_y, r);
    A.template row<M_SEG, false>(CAT_OTHER, e2, c.weight_acc_lim_theta, r);
  }
}

// ---- kinematics -------------------------------------------------------------------------------------------
// CDK ("central-difference kink", the car-like edge only): the reference has a live analytic Jacobian for EdgeKinematicsDiffDrive
// (edge_kinematics.h:112-149, g2o::sign) but differentiates EdgeKinematicsCarlike (:182-230) by g2o's central differences, delta = 1e-9.
// For f = |x| those give sign(x) g only while |x| >= |g| delta; closer to the kink the two samples straddle it and the quotient is
// sign(g) x / delta - next to nothing. A straight stretch of the initial band (a plan initialised from a line, the inflection points of a
// curve) has x = 0 up to rounding, and sign(1e-17) is noise: closed forms then carry the full nonholonomic constraint of the segment
// where the reference's linearisation has none, and the two runs part at the first LM step - 7 of 20 seeded car-like scenes ended 4 ..
// 150 x T3 from the reference (profiles/analytic_margin_r06.txt). The closed-form mode reproduces the central differences' quotient at
// this one kink (oracle/teb_oracle.cpp: kin_nh_row does the same); with it all 20 are within 0.2 T3.
template <bool JAC, bool CDK = false>
__device__ __forceinline__ double kin_nh(const Win& w, double* r /* cols 0,1,2,4,5,6 */) {
  double dx = w.x1 - w.x0, dy = w.y1 - w.y0;
  double cos1 = w.c0, cos2 = w.c1, sin1 = w.s0, sin2 = w.s1;
  double aux1 = sin1 + sin2, aux2 = cos1 + cos2;
  double val = aux2 * dy - aux1 * dx;
  if (JAC) {
    double dev = sgn(val);
    if (!CDK) {
      r[0] = aux1 * dev;
      r[1] = -aux2 * dev;
      r[2] = (-dy * sin1 - dx * cos1) * dev;
      r[4] = -aux1 * dev;
      r[5] = aux2 * dev;
      r[6] = (-sin2 * dy - cos2 * dx) * dev;
    } else {
      const double g[6] = {aux1, -aux2, -dy * sin1 - dx * cos1, -aux1, aux2, -sin2 * dy - cos2 * dx};
      const double av = fabs(val);
#pragma unroll
      for (int q = 0; q < 6; ++q) r[q < 3 ? q : q + 1] = av >= fabs(g[q]) * 1e-9 ? g[q] * dev : sgn(g[q]) * val * 1e9;
    }
  }
  return fabs(val);
}

template <bool JAC, class ACC>
__device__ __forceinline__ void edge_kinematics_diffdrive(const teb_amd_config_t& c, const Win& w, ACC& A) {
  double r[11];
  if (JAC) {
#pragma unroll
    for (int q = 0; q < 11; ++q) r[q] = 0;
  }
  double e0 = kin_nh<JAC>(w, r);
  A.template row<0x077, JAC>(CAT_OTHER, e0, c.weight_kinematics_nh, r);
  double dx = w.x1 - w.x0, dy = w.y1 - w.y0;
  double cos1 = w.c0, sin1 = w.s0;
  double dd;
  double e1 = pen_below(dx * cos1 + dy * sin1, 0, 0, dd);
  if (JAC) {
    r[0] = -cos1 * dd; r[1] = -sin1 * dd; r[2] = (-sin1 * dx + cos1 * dy) * dd;
    r[4] = cos1 * dd; r[5] = sin1 * dd; r[6] = 0;
  }
  A.template row<0x037, JAC>(CAT_OTHER, e1, c.weight_kinematics_forward_drive, r);
}

template <bool JAC, class ACC>
__device__ __forceinline__ void edge_kinematics_carlike(const teb_amd_config_t& c, const Win& w, ACC& A) {
  double r[11];
  if (JAC) {
#pragma unroll
    for (int q = 0; q < 11; ++q) r[q] = 0;
  }
  double e0 = kin_nh<JAC, true>(w, r);
  A.template row<0x077, JAC>(CAT_OTHER, e0, c.weight_kinematics_nh, r);
  double dx = w.x1 - w.x0, dy = w.y1 - w.y0;
  double angle_diff = normalize_theta(w.t1 - w.t0);
  double nn = sqrt(dx * dx + dy * dy);
  double e1 = 0;
  if (JAC) {
#pragma unroll
    for (int q = 0; q < 11; ++q) r[q] = 0;
  }
  if (angle_diff != 0) {
    double rho, drho_dn, drho_dth2, dev;
    if (TEB_CFGI(EXACT_ARC)) {
      double h = angle_diff / 2, sh = sin(h), ch = cos(h);
      rho = fabs(nn / (2 * sh));
      drho_dn = 1.0 / (2 * fabs(sh));
      drho_dth2 = -nn * ch * sgn(sh) / (4 * sh * sh);
    } else {
      rho = nn / fabs(angle_diff);
      drho_dn = 1.0 / fabs(angle_diff);
      drho_dth2 = -nn * sgn(angle_diff) / (angle_diff * angle_diff);
    }
    e1 = pen_below(rho, c.min_turning_radius, 0.0, dev);
    if (JAC) {
      double ux = 0, uy = 0;
      if (nn > 0) { ux = dx / nn; uy = dy / nn; }
      r[0] = dev * drho_dn * (-ux); r[1] = dev * drho_dn * (-uy); r[2] = dev * (-drho_dth2);
      r[4] = dev * drho_dn * ux; r[5] = dev * drho_dn * uy; r[6] = dev * drho_dth2;
    }
  }
  A.template row<0x077, JAC>(CAT_OTHER, e1, c.weight_kinematics_turning_radius, r);
}

// ---- EdgeShortestPath / EdgePreferRotDir / EdgeTimeOptimal ------------------------------------------------
template <bool JAC, class ACC>
__device__ __forceinline__ void edge_shortest_path(const teb_amd_config_t& c, const Win& w, ACC& A) {
  double dx = w.x1 - w.x0, dy = w.y1 - w.y0;
  double nn = sqrt(dx * dx + dy * dy);
  double r[11];
  if (JAC) {
#pragma unroll
    for (int q = 0; q < 11; ++q) r[q] = 0;
    if (nn > 0) { r[0] = -dx / nn; r[1] = -dy / nn; r[4] = dx / nn; r[5] = dy / nn; }
  }
  A.template row<0x033, JAC>(CAT_OTHER, nn, c.weight_shortest_path, r);
}

template <bool JAC, class ACC>
__device__ __forceinline__ void edge_prefer_rotdir(const teb_amd_config_t& c, const Win& w, double dir, ACC& A) {
  double dev;
  double e = pen_below(dir * normalize_theta(w.t1 - w.t0), 0, 0, dev);
  double r[11];
  if (JAC) {
#pragma unroll
    for (int q = 0; q < 11; ++q) r[q] = 0;
    r[2] = dev * (-dir); r[6] = dev * dir;
  }
  A.template row<0x044, JAC>(CAT_OTHER, e, c.weight_prefer_rotdir, r);
}

template <bool JAC, class ACC>
__device__ __forceinline__ void edge_time_optimal(const teb_amd_config_t& c, const Win& w, ACC& A) {
  double r[11];
  if (JAC) {
#pragma unroll
    for (int q = 0; q < 11; ++q) r[q] = 0;
    r[3] = 1;
  }
  A.template row<M_DT0, JAC>(CAT_TIME, w.d0, c.weight_optimaltime, r);
}

// ---- unary pose edges (pose i = window pose 0) --------------------------------------------------------------
// EdgeObstacle / EdgeInflatedObstacle (static, weight_obstacle * multiplier) and EdgeDynamicObstacle
template <bool JAC, class ACC>
__device__ __forceinline__ void edge_obstacle(const teb_amd_config_t& c, const SceneDev& sc, int oi, const Win& w,
                                              double w_obst, bool inflated, ACC& A) {
  double gr[3];
  double dist = footprint_distance(c, sc, oi, w.x0, w.y0, w.c0, w.s0, false, 0.0, JAC ? gr : nullptr);
  double d0;
  double e0 = pen_below(dist, c.min_obstacle_dist, c.penalty_epsilon, d0);
  if (TEB_CFGI(COST_EXPONENT)) {
    double lin = e0;
    e0 = c.min_obstacle_dist * pow(lin / c.min_obstacle_dist, c.obstacle_cost_exponent);
    if (JAC) {
      if (lin > 0) d0 *= c.obstacle_cost_exponent * pow(lin / c.min_obstacle_dist, c.obstacle_cost_exponent - 1.0);
      else d0 = 0;
    }
  }
  double r[11];
  if (JAC) { r[0] = d0 * gr[0]; r[1] = d0 * gr[1]; r[2] = d0 * gr[2]; }
  A.template row<M_POSE0, JAC>(CAT_OBST, e0, w_obst, r);
  if (inflated) {
    double d1;
    double e1 = pen_below(dist, c.inflation_dist, 0.0, d1);
    if (JAC) { r[0] = d1 * gr[0]; r[1] = d1 * gr[1]; r[2] = d1 * gr[2]; }
    A.template row<M_POSE0, JAC>(CAT_OBST, e1, c.weight_inflation, r);
  }
}

// residual rows of EdgeObstacle / EdgeInflatedObstacle and of EdgeDynamicObstacle for the generic shapes, given the distance and its
// gradient (d/dx, d/dy, d/dtheta): exactly what edge_obstacle / edge_dynamic_obstacle do after footprint_distance. The multi-CU mode
// (teb_multicu.hpp) replays the rows from distances other workgroups computed.
template <bool JAC, class ACC>
__device__ __forceinline__ void obstacle_rows_g(const teb_amd_config_t& c, double dist, const double* gr, double w_obst, bool inflated, ACC& A) {
  double d0;
  double e0 = pen_below(dist, c.min_obstacle_dist, c.penalty_epsilon, d0);
  if (TEB_CFGI(COST_EXPONENT)) {
    double lin = e0;
    e0 = c.min_obstacle_dist * pow(lin / c.min_obstacle_dist, c.obstacle_cost_exponent);
    if (JAC) {
      if (lin > 0) d0 *= c.obstacle_cost_exponent * pow(lin / c.min_obstacle_dist, c.obstacle_cost_exponent - 1.0);
      else d0 = 0;
    }
  }
  double r[11];
  if (JAC) { r[0] = d0 * gr[0]; r[1] = d0 * gr[1]; r[2] = d0 * gr[2]; }
  A.template row<M_POSE0, JAC>(CAT_OBST, e0, w_obst, r);
  if (inflated) {
    double d1;
    double e1 = pen_below(dist, c.inflation_dist, 0.0, d1);
    if (JAC) { r[0] = d1 * gr[0]; r[1] = d1 * gr[1]; r[2] = d1 * gr[2]; }
    A.template row<M_POSE0, JAC>(CAT_OBST, e1, c.weight_inflation, r);
  }
}
template <bool JAC, class ACC>
__device__ __forceinline__ void dynamic_obstacle_rows_g(const teb_amd_config_t& c, double dist, const double* gr, ACC& A) {
  double d0, d1;
  double e0 = pen_below(dist, c.min_obstacle_dist, c.penalty_epsilon, d0);
  double e1 = pen_below(dist, c.dynamic_obstacle_inflation_dist, 0.0, d1);
  double r[11];
  if (JAC) { r[0] = d0 * gr[0]; r[1] = d0 * gr[1]; r[2] = d0 * gr[2]; }
  A.template row<M_POSE0, JAC>(CAT_OBST, e0, c.weight_dynamic_obstacle, r);
  if (JAC) { r[0] = d1 * gr[0]; r[1] = d1 * gr[1]; r[2] = d1 * gr[2]; }
  A.template row<M_POSE0, JAC>(CAT_OBST, e1, c.weight_dynamic_obstacle_inflation, r);
}

template <bool JAC, class ACC>
__device__ __forceinline__ void edge_dynamic_obstacle(const teb_amd_config_t& c, const SceneDev& sc, int oi,
                                                      const Win& w, double t, ACC& A) {
  double gr[3];
  double dist = footprint_distance(c, sc, oi, w.x0, w.y0, w.c0, w.s0, true, t, JAC ? gr : nullptr);
  double d0, d1;
  double e0 = pen_below(dist, c.min_obstacle_dist, c.penalty_epsilon, d0);
  double e1 = pen_below(dist, c.dynamic_obstacle_inflation_dist, 0.0, d1);
  double r[11];
  if (JAC) { r[0] = d0 * gr[0]; r[1] = d0 * gr[1]; r[2] = d0 * gr[2]; }
  A.template row<M_POSE0, JAC>(CAT_OBST, e0, c.weight_dynamic_obstacle, r);
  if (JAC) { r[0] = d1 * gr[0]; r[1] = d1 * gr[1]; r[2] = d1 * gr[2]; }
  A.template row<M_POSE0, JAC>(CAT_OBST, e1, c.weight_dynamic_obstacle_inflation, r);
}

template <bool JAC, class ACC>
__device__ __forceinline__ void edge_via_point(const teb_amd_config_t& c, double vx, double vy, const Win& w,
                                               ACC& A) {
  double dx = w.x0 - vx, dy = w.y0 - vy;
  double nn = sqrt(dx * dx + dy * dy);
  double r[11];
  if (JAC) {
    r[0] = 0; r[1] = 0; r[2] = 0;
    if (nn > 0) { r[0] = dx / nn; r[1] = dy / nn; }
  }
  A.template row<0x003, JAC>(CAT_VIA, nn, c.weight_viapoint, r);
}


// ---- point-like fast path ---------------------------------------------------------------------------------------
// Point/Circular footprint against Point/Circular obstacles whose (x, y, vx, vy, radius) live in LDS. Same
// arithmetic (and the same order of operations) as footprint_distance() for these types:
//   PointObstacle/CircularObstacle::getMinimumDistance / getMinimumSpatioTemporalDistance (obstacles.h:358-397,
//   502-541) and Point/CircularRobotFootprint (robot_footprint_model.h:160-175, 263-278).
template <bool GRAD>
__device__ __forceinline__ double pointlike_distance(const teb_amd_config_t& c, double px, double py, double ox, double oy,
                                                     double orad, double* grad) {
  const double vx_ = px - ox, vy_ = py - oy;
  const double dn = sqrt(vx_ * vx_ + vy_ * vy_);
  double dist = dn - orad;
#ifdef TEB_AMD_DEFAULTS_PROFILE
  // (specialised kernels: point and circular footprints without a branch - subtracting 0.0 changes no bit; folding the footprint to
  // "point" would be 1.2 % faster on the headline and shut circular robots out of these kernels)
  dist = dist - (c.footprint_type == TEB_AMD_FOOTPRINT_CIRCULAR ? c.footprint_radius : 0.0);
#else
  if (c.footprint_type == TEB_AMD_FOOTPRINT_CIRCULAR) dist = dist - c.footprint_radius;
#endif
  if (GRAD) {
    if (dn > 0) { grad[0] = vx_ / dn; grad[1] = vy_ / dn; } else { grad[0] = 0; grad[1] = 0; }
  }
  return dist;
}

// residual rows of EdgeObstacle / EdgeInflatedObstacle given the distance and its gradient (split from the distance so that the
// sqrt / divide chains of several obstacles can be in flight at once)
template <bool JAC, class ACC>
__device__ __forceinline__ void obstacle_rows(const teb_amd_config_t& c, double dist, const double* gr, double w_obst, bool inflated, ACC& A) {
  double d0;
  double e0 = pen_below(dist, c.min_obstacle_dist, c.penalty_epsilon, d0);
  if (TEB_CFGI(COST_EXPONENT)) {
    double lin = e0;
    e0 = c.min_obstacle_dist * pow(lin / c.min_obstacle_dist, c.obstacle_cost_exponent);
    if (JAC) {
      if (lin > 0) d0 *= c.obstacle_cost_exponent * pow(lin / c.min_obstacle_dist, c.obstacle_cost_exponent - 1.0);
      else d0 = 0;
    }
  }
  double r[11];
  if (JAC) { r[0] = d0 * gr[0]; r[1] = d0 * gr[1]; }
  A.template row<0x003, JAC>(CAT_OBST, e0, w_obst, r);
  if (inflated) {
    double d1;
    double e1 = pen_below(dist, c.inflation_dist, 0.0, d1);
    if (JAC) { r[0] = d1 * gr[0]; r[1] = d1 * gr[1]; }
    A.template row<0x003, JAC>(CAT_OBST, e1, c.weight_inflation, r);
  }
}
template <bool JAC, class ACC>
__device__ __forceinline__ void edge_obstacle_fast(const teb_amd_config_t& c, double ox, double oy, double orad, const Win& w,
                                                   double w_obst, bool inflated, ACC& A) {
  double gr[2];
  double dist = pointlike_distance<JAC>(c, w.x0, w.y0, ox, oy, orad, gr);
  obstacle_rows<JAC>(c, dist, gr, w_obst, inflated, A);
}

// residual rows of EdgeDynamicObstacle given the distance and its gradient (split from the distance so that
// several obstacles can be in flight at once: the sqrt / divide chains of independent obstacles interleave)
template <bool JAC, class ACC>
__device__ __forceinline__ void dynamic_obstacle_rows(const teb_amd_config_t& c, double dist, const double* gr, ACC& A) {
  double d0, d1;
  double e0 = pen_below(dist, c.min_obstacle_dist, c.penalty_epsilon, d0);
  double e1 = pen_below(dist, c.dynamic_obstacle_inflation_dist, 0.0, d1);
  double r[11];
  if (JAC) { r[0] = d0 * gr[0]; r[1] = d0 * gr[1]; }
  A.template row<0x003, JAC>(CAT_OBST, e0, c.weight_dynamic_obstacle, r);
  if (JAC) { r[0] = d1 * gr[0]; r[1] = d1 * gr[1]; }
  A.template row<0x003, JAC>(CAT_OBST, e1, c.weight_dynamic_obstacle_inflation, r);
}

template <bool JAC, class ACC>
__device__ __forceinline__ void edge_dynamic_obstacle_fast(const teb_amd_config_t& c, double ox, double oy, double orad,
                                                           const Win& w, ACC& A) {
  double gr[2];
  double dist = pointlike_distance<JAC>(c, w.x0, w.y0, ox, oy, orad, gr);
  dynamic_obstacle_rows<JAC>(c, dist, gr, A);
}

// ---- EdgeVelocityObstacleRatio (pose i, pose i+1, dt_i, obstacle) ------------------------------------------
template <bool JAC, class ACC>
__device__ __forceinline__ void edge_velocity_obstacle_ratio(const teb_amd_config_t& c, double dobs, const double* gr,
                                                             const Win& w, ACC& A) {
  // dobs, gr: calculateDistance(conf1->pose(), obstacle) and its gradient w.r.t. pose i
  double v, om, dv[7];
  signed_velocity<JAC>(c, w.x0, w.y0, w.t0, w.c0, w.s0, w.x1, w.y1, w.t1, w.d0, v, om, dv);
  double ratio, dratio;
  if (dobs < c.obstacle_proximity_lower_bound) { ratio = 0; dratio = 0; }
  else if (dobs > c.obstacle_proximity_upper_bound) { ratio = 1; dratio = 0; }
  else {
    ratio = (dobs - c.obstacle_proximity_lower_bound) / (c.obstacle_proximity_upper_bound - c.obstacle_proximity_lower_bound);
    dratio = 1.0 / (c.obstacle_proximity_upper_bound - c.obstacle_proximity_lower_bound);
  }
  ratio *= c.obstacle_proximity_ratio_max_vel;
  dratio *= c.obstacle_proximity_ratio_max_vel;
  const double max_vel_fwd = ratio * c.max_vel_x;
  const double max_omega = ratio * c.max_vel_theta;
  double sv, sw;
  double e0 = pen_interval(v, max_vel_fwd, 0, sv);
  double e1 = pen_interval(om, max_omega, 0, sw);
  double r[11];
  if (JAC) {
#pragma unroll
    for (int q = 0; q < 11; ++q) r[q] = 0;
    r[0] = sv * dv[0]; r[1] = sv * dv[1]; r[2] = sv * dv[2]; r[4] = sv * dv[3]; r[5] = sv * dv[4]; r[6] = sv * dv[5];
    r[3] = sv * dv[6];
    if (sv != 0) { r[0] -= c.max_vel_x * dratio * gr[0]; r[1] -= c.max_vel_x * dratio * gr[1]; r[2] -= c.max_vel_x * dratio * gr[2]; }
  }
  A.template row<M_SEG, JAC>(CAT_OTHER, e0, c.weight_velocity_obstacle_ratio, r);
  if (JAC) {
#pragma unroll
    for (int q = 0; q < 11; ++q) r[q] = 0;
    r[2] = -sw / w.d0; r[6] = sw / w.d0; r[3] = -sw * om / w.d0;
    if (sw != 0) { r[0] -= c.max_vel_theta * dratio * gr[0]; r[1] -= c.max_vel_theta * dratio * gr[1]; r[2] -= c.max_vel_theta * dratio * gr[2]; }
  }
  A.template row<M_SEG, JAC>(CAT_OTHER, e1, c.weight_velocity_obstacle_ratio, r);
}

}  // namespace tebamd
