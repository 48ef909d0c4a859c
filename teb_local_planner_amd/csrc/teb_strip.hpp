// teb_strip.hpp — kernels for the rows either side of optimizeTEB: they create, prune and
// read out the DEVICE-RESIDENT state strips, so that a planning tick needs no full re-upload / download of the bands.
//   f1  TimedElasticBand::initTrajectoryToGoal x3   src/timed_elastic_band.cpp:325-452, timed_elastic_band.hpp:46-183
//       TimedElasticBand::updateAndPruneTEB         src/timed_elastic_band.cpp:555-597
//   f2  TebOptimalPlanner::getVelocityCommand / extractVelocity / getVelocityProfile / getFullTrajectory
//                                                   src/optimal_planner.cpp:1097-1247
// All of it is O(n) fp64 with the same order of operations as the reference (the sequential pieces - running time sums, the
// "insert until min_samples" loops, the early-break nearest-pose search - run on one lane; the per-pose pieces are lane-parallel).
#pragma once
#include "teb_edges.hpp"

namespace tebamd {

// PoseSE2::average (pose_se2.h:266-269) with g2o::average_angle
__device__ __forceinline__ void pose_average(double x1, double y1, double t1, double x2, double y2, double t2, double& ax,
                                             double& ay, double& at) {
  ax = (x1 + x2) / 2; ay = (y1 + y2) / 2;
  const double sx = cos(t1) + cos(t2), sy = sin(t1) + sin(t2);
  at = (sx == 0 && sy == 0) ? 0.0 : atan2(sy, sx);
}

// estimateDeltaT, src/timed_elastic_band.cpp:52-65
__device__ __forceinline__ double estimate_delta_t(double sx, double sy, double sth, double ex, double ey, double eth,
                                                   double max_vel_x, double max_vel_theta) {
  double dt_constant_motion = 0.1;
  if (max_vel_x > 0) dt_constant_motion = nrm2(ex - sx, ey - sy) / max_vel_x;
  if (max_vel_theta > 0) {
    const double rot_dist = fabs(normalize_theta(eth - sth));
    const double r = rot_dist / max_vel_theta;
    if (dt_constant_motion < r) dt_constant_motion = r;   // std::max(a, b)
  }
  return dt_constant_motion;
}

struct StripDev {   // one band of the batch
  double *x, *y, *th, *dt;
  int* n;
  int cap;
};
__device__ __forceinline__ StripDev strip_of(const BatchDev& bt, int b) {
  const size_t o = (size_t)b * bt.stride;
  return StripDev{bt.x + o, bt.y + o, bt.th + o, bt.dt + o, bt.n + b, bt.stride};
}

// the common tail of the three initTrajectoryToGoal variants, on one lane: insert averages until min_samples-1, then the goal.
// MODE 0: timestep = |step| / max_vel_x if max_vel_x > 0 (:361-372); 1: estimateDeltaT (:427-441); 2: timestep halves (hpp:160-172)
template <int MODE>
__device__ inline int init_tail(StripDev s, int n, double gx, double gy, double gth, double max_vel_x, double max_vel_theta,
                                double timestep, int min_samples) {
  while (n < min_samples - 1) {
    if (n + 1 > s.cap) return -1;
    double ax, ay, at;
    pose_average(s.x[n - 1], s.y[n - 1], s.th[n - 1], gx, gy, gth, ax, ay, at);
    if (MODE == 0) { if (max_vel_x > 0) timestep = nrm2(ax - s.x[n - 1], ay - s.y[n - 1]) / max_vel_x; }
    else if (MODE == 1) timestep = estimate_delta_t(s.x[n - 1], s.y[n - 1], s.th[n - 1], ax, ay, at, max_vel_x, max_vel_theta);
    else timestep /= 2;
    s.x[n] = ax; s.y[n] = ay; s.th[n] = at; s.dt[n - 1] = timestep;
    ++n;
  }
  if (n + 1 > s.cap) return -1;
  if (MODE == 0) { if (max_vel_x > 0) timestep = nrm2(gx - s.x[n - 1], gy - s.y[n - 1]) / max_vel_x; }
  else if (MODE == 1) timestep = estimate_delta_t(s.x[n - 1], s.y[n - 1], s.th[n - 1], gx, gy, gth, max_vel_x, max_vel_theta);
  s.x[n] = gx; s.y[n] = gy; s.th[n] = gth; s.dt[n - 1] = timestep;
  return n + 1;
}

// initTrajectoryToGoal(start, goal, diststep, max_vel_x, min_samples, guess_backwards_motion), :325-377. One workgroup.
__global__ void init_line_kernel(BatchDev bt, int b, double sx, double sy, double sth, double gx, double gy, double gth,
                                 double diststep, double max_vel_x, int min_samples, int guess_backwards, int* err) {
  StripDev s = strip_of(bt, b);
  double timestep = 0.1;
  int n = 1;
  if (threadIdx.x == 0) { s.x[0] = sx; s.y[0] = sy; s.th[0] = sth; }
  if (diststep != 0) {
    const double px = gx - sx, py = gy - sy;
    const double dir_to_goal = atan2(py, px);
    const double dx = diststep * cos(dir_to_goal);
    const double dy = diststep * sin(dir_to_goal);
    double orient_init = dir_to_goal;
    if (guess_backwards && (px * cos(sth) + py * sin(sth)) < 0) orient_init = normalize_theta(orient_init + M_PI);
    const double dist_to_goal = nrm2(px, py);
    const double no_steps_d = dist_to_goal / fabs(diststep);
    const unsigned int no_steps = (unsigned int)floor(no_steps_d);
    if (max_vel_x > 0) timestep = diststep / max_vel_x;
    // poses 1 .. last, where the final sample is dropped when it coincides with the goal (:351-353)
    unsigned int last = no_steps;
    if (no_steps >= 1 && no_steps_d == (double)(float)no_steps) last = no_steps - 1;
    if ((int)last + 1 > s.cap) { if (threadIdx.x == 0) { *err = 1; *s.n = 0; } return; }
    for (unsigned int i = 1 + threadIdx.x; i <= last; i += blockDim.x) {
      s.x[i] = sx + i * dx; s.y[i] = sy + i * dy; s.th[i] = orient_init; s.dt[i - 1] = timestep;
    }
    n = (int)last + 1;
  }
  __threadfence_block();
  __syncthreads();
  if (threadIdx.x == 0) {
    const int r = init_tail<0>(s, n, gx, gy, gth, max_vel_x, 0.0, timestep, min_samples);
    if (r < 0) { *err = 1; *s.n = 0; } else *s.n = r;
  }
}

// initTrajectoryToGoal(plan, max_vel_x, max_vel_theta, estimate_orient, min_samples, guess_backwards_motion), :380-452.
// plan = np positions + yaw staged in px/py/pyaw (device). One workgroup.
__global__ void init_plan_kernel(BatchDev bt, int b, int np, const double* px, const double* py, const double* pyaw,
                                 double max_vel_x, double max_vel_theta, int estimate_orient, int min_samples,
                                 int guess_backwards, int* err) {
  StripDev s = strip_of(bt, b);
  const double sx = px[0], sy = py[0], sth = pyaw[0];
  const double gx = px[np - 1], gy = py[np - 1], gth = pyaw[np - 1];
  const bool backwards = guess_backwards && ((gx - sx) * cos(sth) + (gy - sy) * sin(sth)) < 0;
  const int n = np >= 2 ? np - 1 : 1;   // start + intermediate plan poses
  if (n > s.cap) { if (threadIdx.x == 0) { *err = 1; *s.n = 0; } return; }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    double yaw;
    if (i == 0) yaw = sth;
    else if (estimate_orient) {
      yaw = atan2(py[i + 1] - py[i], px[i + 1] - px[i]);
      if (backwards) yaw = normalize_theta(yaw + M_PI);
    } else yaw = pyaw[i];
    s.x[i] = px[i]; s.y[i] = py[i]; s.th[i] = yaw;
  }
  __threadfence_block();
  __syncthreads();
  for (int i = 1 + threadIdx.x; i < n; i += blockDim.x)   // dt between BackPose() (= pose i-1) and pose i
    s.dt[i - 1] = estimate_delta_t(s.x[i - 1], s.y[i - 1], s.th[i - 1], s.x[i], s.y[i], s.th[i], max_vel_x, max_vel_theta);
  __threadfence_block();
  __syncthreads();
  if (threadIdx.x == 0) {
    const int r = init_tail<1>(s, n, gx, gy, gth, max_vel_x, max_vel_theta, 0.0, min_samples);
    if (r < 0) { *err = 1; *s.n = 0; } else *s.n = r;
  }
}

// updateAndPruneTEB, :555-597. One workgroup per band (blockIdx.x + b0). Dynamic LDS: 4 * stride doubles.
__global__ void prune_kernel(BatchDev bt, int b0, int has_start, double sx, double sy, double sth, int has_goal, double gx,
                             double gy, double gth, int min_samples) {
  extern __shared__ __attribute__((aligned(16))) double sbuf[];
  __shared__ int sh_k;
  StripDev s = strip_of(bt, b0 + blockIdx.x);
  const int n = *s.n;
  if (n <= 0) return;
  if (has_start) {
    if (threadIdx.x == 0) {
      double dist_cache = nrm2(sx - s.x[0], sy - s.y[0]);
      int lookahead = n - min_samples; if (lookahead > 10) lookahead = 10;
      int nearest_idx = 0;
      for (int i = 1; i <= lookahead; ++i) {
        const double dist = nrm2(sx - s.x[i], sy - s.y[i]);
        if (dist < dist_cache) { dist_cache = dist; nearest_idx = i; }
        else break;
      }
      sh_k = nearest_idx;
    }
    __syncthreads();
    const int k = sh_k;
    if (k > 0) {   // deletePoses(1, k); deleteTimeDiffs(1, k): pose j (j >= 1) <- pose j+k, dt j <- dt j+k
      const int S = s.cap;
      for (int i = threadIdx.x; i < n; i += blockDim.x) { sbuf[i] = s.x[i]; sbuf[S + i] = s.y[i]; sbuf[2 * S + i] = s.th[i]; sbuf[3 * S + i] = s.dt[i]; }
      __syncthreads();
      for (int j = 1 + threadIdx.x; j < n - k; j += blockDim.x) { s.x[j] = sbuf[j + k]; s.y[j] = sbuf[S + j + k]; s.th[j] = sbuf[2 * S + j + k]; }
      for (int j = 1 + threadIdx.x; j < n - 1 - k; j += blockDim.x) s.dt[j] = sbuf[3 * S + j + k];
      __syncthreads();
    }
    if (threadIdx.x == 0) { s.x[0] = sx; s.y[0] = sy; s.th[0] = sth; *s.n = n - k; }
    __threadfence_block();
    __syncthreads();
  }
  if (has_goal && threadIdx.x == 0) {
    const int m = *s.n;
    if (m > 0) { s.x[m - 1] = gx; s.y[m - 1] = gy; s.th[m - 1] = gth; }
  }
}

// extractVelocity, src/optimal_planner.cpp:1097-1133
__device__ __forceinline__ void extract_velocity(const teb_amd_config_t& c, double x1, double y1, double th1, double x2, double y2,
                                                 double th2, double dt, double& vx, double& vy, double& omega) {
  if (dt == 0) { vx = 0; vy = 0; omega = 0; return; }
  const double dx = x2 - x1, dy = y2 - y1;
  if (c.max_vel_y == 0) {
    const double dir = dx * cos(th1) + dy * sin(th1);
    vx = sgn(dir) * nrm2(dx, dy) / dt;
    vy = 0;
  } else {
    const double cos_theta1 = cos(th1), sin_theta1 = sin(th1);
    const double p1_dx = cos_theta1 * dx + sin_theta1 * dy;
    const double p1_dy = -sin_theta1 * dx + cos_theta1 * dy;
    vx = p1_dx / dt;
    vy = p1_dy / dt;
  }
  omega = normalize_theta(th2 - th1) / dt;
}

// getVelocityCommand (:1135-1168), getVelocityProfile (:1170-1196), getFullTrajectory (:1198-1247) of every band in one launch.
// cmd [B][4] = (vx, vy, omega, ok); prof [B][(S+1)*3]; traj [B][S*7] = (x, y, theta, vx, vy, omega, time_from_start)
__global__ void consumers_kernel(const teb_amd_config_t c, BatchDev bt, int look_ahead_poses, int prevent_near_goal, double* cmd,
                                 double* prof, double* traj) {
  const int b = blockIdx.x, S = bt.stride;
  StripDev s = strip_of(bt, b);
  const int n = *s.n;
  double* pr = prof + (size_t)b * (S + 1) * 3;
  double* tr = traj + (size_t)b * S * 7;
  double* cm = cmd + (size_t)b * 4;
  const double* vs = bt.vs + 3 * b;
  const double* vg = bt.vg + 3 * b;
  if (threadIdx.x == 0) {
    double vx = 0, vy = 0, om = 0, ok = 0;
    if (n >= 2) {
      int la = look_ahead_poses;
      const int lim = n - 1 - prevent_near_goal;
      if (la > lim) la = lim;
      if (la < 1) la = 1;
      double dt = 0.0;
      for (int counter = 0; counter < la; ++counter) {
        dt += s.dt[counter];
        if (dt >= c.dt_ref * la) { la = counter + 1; break; }
      }
      if (dt > 0) { extract_velocity(c, s.x[0], s.y[0], s.th[0], s.x[la], s.y[la], s.th[la], dt, vx, vy, om); ok = 1; }
    }
    cm[0] = vx; cm[1] = vy; cm[2] = om; cm[3] = ok;
    // running time stamps, summed left to right like the reference
    double curr_time = 0;
    for (int i = 0; i < n; ++i) { tr[7 * i + 6] = curr_time; if (i < n - 1) curr_time += s.dt[i]; }
  }
  if (n <= 0) return;
  for (int i = threadIdx.x; i <= n; i += blockDim.x) {
    double vx, vy, om;
    if (i == 0) { vx = vs[0]; vy = vs[1]; om = vs[2]; }
    else if (i == n) { vx = vg[0]; vy = vg[1]; om = vg[2]; }
    else extract_velocity(c, s.x[i - 1], s.y[i - 1], s.th[i - 1], s.x[i], s.y[i], s.th[i], s.dt[i - 1], vx, vy, om);
    pr[3 * i] = vx; pr[3 * i + 1] = vy; pr[3 * i + 2] = om;
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    double vx, vy, om;
    if (i == 0) { vx = vs[0]; vy = vs[1]; om = vs[2]; }
    else if (i == n - 1) { vx = vg[0]; vy = vg[1]; om = vg[2]; }
    else {
      double v1x, v1y, o1, v2x, v2y, o2;
      extract_velocity(c, s.x[i - 1], s.y[i - 1], s.th[i - 1], s.x[i], s.y[i], s.th[i], s.dt[i - 1], v1x, v1y, o1);
      extract_velocity(c, s.x[i], s.y[i], s.th[i], s.x[i + 1], s.y[i + 1], s.th[i + 1], s.dt[i], v2x, v2y, o2);
      vx = 0.5 * (v1x + v2x); vy = 0.5 * (v1y + v2y); om = 0.5 * (o1 + o2);
    }
    tr[7 * i] = s.x[i]; tr[7 * i + 1] = s.y[i]; tr[7 * i + 2] = s.th[i]; tr[7 * i + 3] = vx; tr[7 * i + 4] = vy; tr[7 * i + 5] = om;
  }
}

}  // namespace tebamd
