// teb_graph.hpp — kernels of the candidate generation (GraphSearchInterface::createGraph,
// src/graph_search.cpp:95-340, and the band initialisation of addAndInitNewTeb, homotopy_class_planner.hpp:66-93).
//   graph_edges_kernel     every ordered vertex pair (i, j): forward-direction test, the start-heading test of the keypoint graph,
//                          then Obstacle::checkLineIntersection against every obstacle -> adjacency byte matrix [N * N].
//                          One lane per pair; the obstacle index is wave-uniform, so the obstacle table arrives through scalar loads.
//                          N^2 * M segment tests: 272 * M for the 17-vertex roadmap, 10^6 * M for a 500-obstacle keypoint graph.
//   init_path_batch_kernel the template initTrajectoryToGoal(path...) for a chunk of candidate paths, one workgroup per path.
//   move_bands_kernel      gathers bands (and their per-band attributes) between strip sets: candidate -> batch, compaction.
// fp64, one IEEE operation per source operation of the reference (the library is built with -ffp-contract=off).
#pragma once
#include "teb_strip.hpp"

namespace tebamd {

// segments_intersect = check_line_segments_intersection_2d (distance_calculations.h:97-127) lives in teb_geometry.hpp

// Obstacle::checkLineIntersection(line_start, line_end, min_dist) of obstacle o (obstacles.h:339-354, 483-498, 647-650, 794-797;
// src/obstacles.cpp:178-191): point / circle obstacles keep min_dist from the segment, line / pill / polygon obstacles only test
// for a proper crossing (min_dist unused there, as in the reference)
__device__ __forceinline__ bool line_hits_obstacle(const SceneDev& sc, int o, double sx, double sy, double ex, double ey,
                                                   double min_dist) {
  const int ty = sc.type[o];
  if (ty == TEB_AMD_OBST_POINT || ty == TEB_AMD_OBST_CIRCULAR) {
    const double ax = ex - sx, ay = ey - sy;
    const double px = sc.ax[o], py = sc.ay[o];
    const double bx = px - sx, by = py - sy;
    double t = (ax * bx + ay * by) / (ax * ax + ay * ay);
    if (t < 0) t = 0; else if (t > 1) t = 1;
    const double nx = sx + ax * t, ny = sy + ay * t;
    double d = nrm2(nx - px, ny - py);
    if (ty == TEB_AMD_OBST_CIRCULAR) d = d - sc.rad[o];
    return d < min_dist;
  }
  if (ty == TEB_AMD_OBST_LINE || ty == TEB_AMD_OBST_PILL) return segments_intersect(sx, sy, ex, ey, sc.ax[o], sc.ay[o], sc.bx[o], sc.by[o]);
  const int k0 = sc.voff[o], nv = sc.voff[o + 1] - k0;
  for (int i = 0; i < nv - 1; ++i)
    if (segments_intersect(sx, sy, ex, ey, sc.pvx[k0 + i], sc.pvy[k0 + i], sc.pvx[k0 + i + 1], sc.pvy[k0 + i + 1])) return true;
  if (nv == 2) return false;
  if (nv < 1) return false;
  return segments_intersect(sx, sy, ex, ey, sc.pvx[k0 + nv - 1], sc.pvy[k0 + nv - 1], sc.pvx[k0], sc.pvy[k0]);
}

struct GraphArgs {
  int N;                 // vertices: 0 = start, N-1 = goal
  const double *gx, *gy;
  double dnx, dny;       // normalised start->goal direction
  double thr;            // obstacle_heading_threshold
  int keypoint;          // 1: lrKeyPointGraph (start-heading test on the nearest obstacle's key points), 0: ProbRoadmapGraph
  int near_u, near_v;    // key points of the obstacle nearest to the start, or -1
  double sox, soy;       // (cos, sin) of the start orientation
  double min_dist;       // 0.5 * dist_to_obst (keypoint graph) or dist_to_obst (roadmap)
  unsigned char* adj;    // [N * N], adj[i * N + j] = 1 iff edge i -> j
};

// Eigen normalize(): v /= sqrt(squaredNorm) when squaredNorm > 0
__device__ __forceinline__ void normalize2(double& x, double& y) {
  const double z = x * x + y * y;
  if (z > 0) { const double n = sqrt(z); x = x / n; y = y / n; }
}

// src/graph_search.cpp:156-213 (keypoint graph) and :301-333 (roadmap): the edge insertion double loop
__global__ void __launch_bounds__(kThreads) graph_edges_kernel(const SceneDev sc, const GraphArgs g) {
  const long long idx = (long long)blockIdx.x * kThreads + threadIdx.x;
  const int N = g.N;
  if (idx >= (long long)N * N) return;
  const int i = (int)(idx / N), j = (int)(idx % N);
  bool edge = (i != j) && (i != N - 1);   // the goal vertex has no outgoing edges
  double xi = 0, yi = 0, xj = 0, yj = 0;
  if (edge) {
    xi = g.gx[i]; yi = g.gy[i]; xj = g.gx[j]; yj = g.gy[j];
    double dx = xj - xi, dy = yj - yi;
    normalize2(dx, dy);
    if (dx * g.dnx + dy * g.dny <= g.thr) edge = false;   // backwards (or sideways) connection
  }
  if (edge && g.keypoint && i == 0 && (j == g.near_u || j == g.near_v)) {   // start angle to the nearest obstacle, :174-188
    double kx = xj - g.gx[0], ky = yj - g.gy[0];
    normalize2(kx, ky);
    if (g.sox * kx + g.soy * ky <= g.thr) edge = false;
  }
  // the obstacle loop is wave-uniform in o (scalar loads of the table); a lane leaves it at its first hit
  if (__builtin_amdgcn_ballot_w64(edge) != 0) {
    for (int o = 0; o < sc.M; ++o) {
      if (edge && line_hits_obstacle(sc, o, xi, yi, xj, yj, g.min_dist)) edge = false;
      if (__builtin_amdgcn_ballot_w64(edge) == 0) break;
    }
  }
  g.adj[idx] = edge ? 1 : 0;
}

// template initTrajectoryToGoal(path_start, path_end, fun_position, ...), timed_elastic_band.hpp:46-183, for one band; called by
// every lane of the workgroup that owns the band
__device__ inline void init_path_band(StripDev s, int np, const double* px, const double* py, double max_vel_x, int has_max_acc_x,
                                      double max_acc_x, int has_start_orient, double start_orientation, int has_goal_orient,
                                      double goal_orientation, int min_samples, int guess_backwards, int* err) {
  const double sx = px[0], sy = py[0], gx = px[np - 1], gy = py[np - 1];
  bool backwards = false;
  double start_orient;
  if (has_start_orient) {
    start_orient = start_orientation;
    if (guess_backwards && ((gx - sx) * cos(start_orient) + (gy - sy) * sin(start_orient)) < 0) backwards = true;
  } else start_orient = atan2(gy - sy, gx - sx);
  const double goal_orient = has_goal_orient ? goal_orientation : start_orient;
  const int n = np >= 2 ? np - 1 : 1;
  if (n > s.cap) { if (threadIdx.x == 0) { *err = 1; *s.n = 0; } return; }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    if (i == 0) { s.x[0] = sx; s.y[0] = sy; s.th[0] = start_orient; continue; }
    const double dlx = px[i] - px[i - 1], dly = py[i] - py[i - 1];   // curr_point - Pose(idx).position(): the previous path point
    const double diff_norm = nrm2(dlx, dly);
    const double timestep_vel = diff_norm / max_vel_x;
    double timestep;
    if (has_max_acc_x) {
      const double timestep_acc = sqrt(2 * diff_norm / max_acc_x);
      timestep = (timestep_vel < timestep_acc) ? timestep_acc : timestep_vel;
    } else timestep = timestep_vel;
    if (timestep <= 0) timestep = 0.2;
    double yaw = atan2(dly, dlx);
    if (backwards) yaw = normalize_theta(yaw + M_PI);
    s.x[i] = px[i]; s.y[i] = py[i]; s.th[i] = yaw; s.dt[i - 1] = timestep;
  }
  __threadfence_block();
  __syncthreads();
  if (threadIdx.x == 0) {
    const double diff_norm = nrm2(gx - s.x[n - 1], gy - s.y[n - 1]);
    const double timestep_vel = diff_norm / max_vel_x;
    double timestep;
    if (has_max_acc_x) {
      const double timestep_acc = sqrt(2 * diff_norm / max_acc_x);
      timestep = (timestep_vel < timestep_acc) ? timestep_acc : timestep_vel;
    } else timestep = timestep_vel;
    const int r = init_tail<2>(s, n, gx, gy, goal_orient, max_vel_x, 0.0, timestep, min_samples);
    if (r < 0) { *err = 1; *s.n = 0; } else *s.n = r;
  }
}

// the single-band entry point (teb_amd_init_trajectory_path). One workgroup.
__global__ void init_path_kernel(BatchDev bt, int b, int np, const double* px, const double* py, double max_vel_x,
                                 int has_max_acc_x, double max_acc_x, int has_start_orient, double start_orientation,
                                 int has_goal_orient, double goal_orientation, int min_samples, int guess_backwards, int* err) {
  init_path_band(strip_of(bt, b), np, px, py, max_vel_x, has_max_acc_x, max_acc_x, has_start_orient, start_orientation, has_goal_orient,
                 goal_orientation, min_samples, guess_backwards, err);
}

// one workgroup per candidate path k: path points px/py [off[k], off[k+1]) -> band k of bt
__global__ void init_path_batch_kernel(BatchDev bt, const int* off, const double* px, const double* py, double max_vel_x,
                                       double max_acc_x, double start_orientation, double goal_orientation, int min_samples,
                                       int guess_backwards, int* err) {
  const int k = blockIdx.x;
  const int o = off[k];
  init_path_band(strip_of(bt, k), off[k + 1] - o, px + o, py + o, max_vel_x, 1, max_acc_x, 1, start_orientation, 1, goal_orientation,
                 min_samples, guess_backwards, err);
}

// dst band blockIdx.x <- src band map[blockIdx.x], with the per-band attributes when attrs != 0
__global__ void move_bands_kernel(BatchDev src, BatchDev dst, const int* map, int dst0, int attrs) {
  const int d = dst0 + blockIdx.x, s = map[blockIdx.x];
  const int n = src.n[s];
  const size_t so = (size_t)s * src.stride, dofs = (size_t)d * dst.stride;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    dst.x[dofs + i] = src.x[so + i]; dst.y[dofs + i] = src.y[so + i]; dst.th[dofs + i] = src.th[so + i];
    if (i < n - 1) dst.dt[dofs + i] = src.dt[so + i];
  }
  if (threadIdx.x == 0) {
    dst.n[d] = n;
    if (attrs) {
      const_cast<int*>(dst.has_vs)[d] = src.has_vs[s]; const_cast<int*>(dst.has_vg)[d] = src.has_vg[s];
      const_cast<int*>(dst.rotdir)[d] = src.rotdir[s]; const_cast<int*>(dst.via_en)[d] = src.via_en[s];
      for (int k = 0; k < 3; ++k) { const_cast<double*>(dst.vs)[3 * d + k] = src.vs[3 * s + k]; const_cast<double*>(dst.vg)[3 * d + k] = src.vg[3 * s + k]; }
      dst.status[d] = src.status[s]; dst.iters[d] = src.iters[s]; dst.last_iters[d] = src.last_iters[s]; dst.trials[d] = src.trials[s]; dst.optimized[d] = src.optimized[s];
      dst.chi2[d] = src.chi2[s]; dst.cost[d] = src.cost[s]; dst.lambda[d] = src.lambda[s];
    }
  }
}

// per band: computeStartOrientation (src/homotopy_class_planner.cpp:819-838) and getSumOfAllTimeDiffs, the inputs of
// deletePlansDetouringBackwards (:766-817). out [B * 4] = found (0/1), orientation, sum of time differences, number of poses
__global__ void __launch_bounds__(kThreads) detour_stats_kernel(BatchDev bt, double len_orientation_vector, double* out) {
  __shared__ int first;
  const int b = blockIdx.x;
  const int n = bt.n[b];
  const size_t so = (size_t)b * bt.stride;
  if (threadIdx.x == 0) first = n;
  __syncthreads();
  if (n > 0) {
    const double x0 = bt.x[so], y0 = bt.y[so];
    int mine = n;
    for (int i = threadIdx.x; i < n; i += kThreads)
      if (nrm2(x0 - bt.x[so + i], y0 - bt.y[so + i]) > len_orientation_vector) { mine = i; break; }   // ascending i per lane
    if (mine < n) atomicMin(&first, mine);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int f = first;
    double orient = 0, sum = 0;
    if (f < n) orient = atan2(bt.y[so] - bt.y[so + f], bt.x[so] - bt.x[so + f]);
    for (int i = 0; i < n - 1; ++i) sum += bt.dt[so + i];   // the reference's running sum, in order
    out[4 * b] = f < n ? 1.0 : 0.0; out[4 * b + 1] = orient; out[4 * b + 2] = sum; out[4 * b + 3] = (double)n;
  }
}

}  // namespace tebamd
