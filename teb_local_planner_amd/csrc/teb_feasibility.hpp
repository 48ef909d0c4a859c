// teb_feasibility.hpp — SURVEY section 8(f) row f4, the arithmetic part: TebOptimalPlanner::isTrajectoryFeasible
// (src/optimal_planner.cpp:1250-1308) on the device-resident bands against a uint8 costmap grid in HBM.
//
// The footprint test the reference calls is base_local_planner::CostmapModel::footprintCost of the ROS navigation stack (an un-vendored
// dependency): footprint vertices rotated / translated to the pose, every outline edge rasterised with Bresenham
// (base_local_planner::LineIterator), LETHAL_OBSTACLE 254 -> -1, NO_INFORMATION 255 -> -2, a vertex or the centre off the map -> -3,
// fewer than 3 vertices -> the centre cell alone (where INSCRIBED_INFLATED_OBSTACLE 253 is lethal as well). The reference treats ONLY
// -1 as a collision (:1269, :1292).
//
// The reference walks the poses 0 .. look_ahead_idx one after the other and, where two consecutive poses are more than the inscribed
// radius apart or turn by more than min_resolution_collision_check_angular, n additional samples in between, stopping at the first
// collision. All these footprint tests are independent: here every test is one work item (index = its position in the reference's
// order), 256 lanes per band, and the verdict is the minimum index of a colliding test. The additional samples are produced by the
// reference's own recurrence (position += delta / (n + 1), theta = normalize(theta + delta_rot / (n + 1)), step by step), replayed by
// the lane from the segment's first pose so that every sample has the reference's bits.
#pragma once
#include "teb_edges.hpp"

namespace tebamd {

struct GridDev {
  const unsigned char* cells;
  int sx, sy;
  double res, ox, oy;
};

// costmap_2d::Costmap2D::worldToMap
__device__ __forceinline__ bool world_to_map(const GridDev& g, double wx, double wy, int& mx, int& my) {
  if (wx < g.ox || wy < g.oy) return false;
  mx = (int)((wx - g.ox) / g.res);
  my = (int)((wy - g.oy) / g.res);
  return (unsigned)mx < (unsigned)g.sx && (unsigned)my < (unsigned)g.sy;
}
// CostmapModel::lineCost over LineIterator: < 0 at the first lethal (-1) / unknown (-2) cell, else the highest cost on the line
__device__ __forceinline__ int line_cost(const GridDev& g, int x0, int x1, int y0, int y1) {
  int best = 0;
  const int deltax = abs(x1 - x0), deltay = abs(y1 - y0);
  int x = x0, y = y0;
  int xinc1, xinc2, yinc1, yinc2;
  if (x1 >= x0) { xinc1 = 1; xinc2 = 1; } else { xinc1 = -1; xinc2 = -1; }
  if (y1 >= y0) { yinc1 = 1; yinc2 = 1; } else { yinc1 = -1; yinc2 = -1; }
  int den, num, numadd, numpixels;
  if (deltax >= deltay) { xinc1 = 0; yinc2 = 0; den = deltax; num = deltax / 2; numadd = deltay; numpixels = deltax; }
  else { xinc2 = 0; yinc1 = 0; den = deltay; num = deltay / 2; numadd = deltax; numpixels = deltay; }
  for (int cur = 0; cur <= numpixels; ++cur) {
    const int cost = g.cells[(size_t)y * g.sx + x];
    if (cost == 255) return -2;
    if (cost == 254) return -1;
    best = cost > best ? cost : best;
    num += numadd;
    if (num >= den) { num -= den; x += xinc1; y += yinc1; }
    x += xinc2; y += yinc2;
  }
  return best;
}
// WorldModel::footprintCost(x, y, theta, spec) -> CostmapModel::footprintCost(position, oriented footprint); cell costs are integers
__device__ __forceinline__ int footprint_cost(const GridDev& g, double x, double y, double theta, int nf, const double* fx, const double* fy) {
  const double cos_th = cos(theta), sin_th = sin(theta);
  int cx, cy;
  if (!world_to_map(g, x, y, cx, cy)) return -3;
  if (nf < 3) {
    const int cost = g.cells[(size_t)cy * g.sx + cx];
    if (cost == 255) return -2;
    if (cost == 254 || cost == 253) return -1;
    return cost;
  }
  int best = 0;
  for (int i = 0; i < nf; ++i) {   // edges 0-1, 1-2, ..., then last-first
    const int j = (i + 1 < nf) ? i + 1 : 0;
    const double ax = x + (fx[i] * cos_th - fy[i] * sin_th), ay = y + (fx[i] * sin_th + fy[i] * cos_th);
    const double bx = x + (fx[j] * cos_th - fy[j] * sin_th), by = y + (fx[j] * sin_th + fy[j] * cos_th);
    int x0, y0, x1, y1;
    if (!world_to_map(g, ax, ay, x0, y0)) return -3;
    if (!world_to_map(g, bx, by, x1, y1)) return -3;
    const int lc = line_cost(g, x0, x1, y0, y1);
    best = lc > best ? lc : best;
    if (lc < 0) return lc;
  }
  return best;
}

constexpr int kMaxFeasFootprint = 64;
constexpr int kFeasThreads = 256;

// grid = number of bands checked (band = first + blockIdx.x); out_feasible / out_first [gridDim.x]
__global__ void __launch_bounds__(kFeasThreads)
feasibility_kernel(const int* __restrict__ n_arr, const double* __restrict__ X, const double* __restrict__ Y, const double* __restrict__ TH,
                   int stride, int first, GridDev g, int nf, const double* __restrict__ fpx, const double* __restrict__ fpy,
                   double inscribed_radius, double min_res_angular, int look_ahead_idx, double lookahead_distance, int max_tests,
                   int* __restrict__ out_feasible, int* __restrict__ out_first, int* __restrict__ out_overflow) {
  __shared__ int s_look, s_fail, s_total;
  __shared__ int s_base[1024 + 1];          // index of pose i's own test in the reference's order (i <= look_ahead)
  __shared__ double s_fx[kMaxFeasFootprint], s_fy[kMaxFeasFootprint];
  const int b = first + blockIdx.x, tid = threadIdx.x;
  const int n = n_arr[b];
  const double* x = X + (size_t)b * stride; const double* y = Y + (size_t)b * stride; const double* th = TH + (size_t)b * stride;
  if (tid < nf) { s_fx[tid] = fpx[tid]; s_fy[tid] = fpy[tid]; }
  if (tid == 0) {
    int look = look_ahead_idx;
    if (look < 0 || look >= n) look = n - 1;                                       // :1253-1254
    s_look = look; s_fail = 0x7fffffff;
  }
  __syncthreads();
  if (lookahead_distance > 0) {                                                       // :1256-1264: first pose farther than that from pose 0
    int cand = 0x7fffffff;
    for (int i = 1 + tid; i < n; i += kFeasThreads)
      if (hypot(x[i] - x[0], y[i] - y[0]) > lookahead_distance) { cand = i; break; }
    if (cand != 0x7fffffff) atomicMin(&s_fail, cand);
    __syncthreads();
    if (tid == 0) { if (s_fail != 0x7fffffff) s_look = s_fail - 1; s_fail = 0x7fffffff; }
    __syncthreads();
  }
  const int look = s_look;
  // additional samples per segment i -> i + 1 (:1278-1284), then the position of every pose's own test (sequential prefix: look <= 1024)
  for (int i = tid; i <= look; i += kFeasThreads) {
    int extra = 0;
    if (i < look) {
      const double delta_rot = normalize_theta(normalize_theta(th[i + 1]) - normalize_theta(th[i]));
      const double ddx = x[i + 1] - x[i], ddy = y[i + 1] - y[i];
      const double dnorm = sqrt(ddx * ddx + ddy * ddy);
      if (fabs(delta_rot) > min_res_angular || dnorm > inscribed_radius) {
        const double a = ceil(fabs(delta_rot) / min_res_angular), d = ceil(dnorm / inscribed_radius);
        const double m = fmax(a, d);
        extra = (m >= 2147483647.0 || !(m == m)) ? 0x3fffffff : (int)m - 1;
      }
    }
    s_base[i + 1] = extra;   // temporarily: samples after pose i
  }
  __syncthreads();
  if (tid == 0) {
    long long acc = 0;
    int ovf = 0;
    for (int i = 0; i <= look; ++i) {
      const int extra = s_base[i + 1];
      s_base[i] = (int)acc;
      acc += 1 + (long long)extra;
      if (acc > max_tests) { ovf = 1; break; }
    }
    s_total = ovf ? -1 : (int)acc;
    if (!ovf) s_base[look + 1] = (int)acc;
  }
  __syncthreads();
  const int total = s_total;
  if (total < 0) {   // an absurd number of samples (inscribed radius / angular resolution ~ 0): refuse rather than spin
    if (tid == 0) { out_feasible[blockIdx.x] = 0; out_first[blockIdx.x] = -1; *out_overflow = 1; }
    return;
  }
  for (int q = tid; q < total; q += kFeasThreads) {
    if (q > s_fail) break;                         // a test earlier in the reference's order already failed
    int lo = 0, hi = look;                         // the segment of test q: largest i with base[i] <= q
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_base[mid] <= q) lo = mid; else hi = mid - 1; }
    const int i = lo, step = q - s_base[i];
    double px = x[i], py = y[i], pth = th[i];
    if (step > 0) {
      const double delta_rot = normalize_theta(normalize_theta(th[i + 1]) - normalize_theta(th[i]));
      const double ddx = x[i + 1] - x[i], ddy = y[i + 1] - y[i];
      const int nadd = s_base[i + 1] - s_base[i] - 1;
      for (int k = 0; k < step; ++k) {
        px = px + ddx / (nadd + 1.0);
        py = py + ddy / (nadd + 1.0);
        pth = normalize_theta(pth + delta_rot / (nadd + 1.0));
      }
    }
    if (footprint_cost(g, px, py, pth, nf, s_fx, s_fy) == -1) atomicMin(&s_fail, q);
  }
  __syncthreads();
  if (tid == 0) {
    const int f = s_fail;
    out_feasible[blockIdx.x] = f == 0x7fffffff ? 1 : 0;
    out_first[blockIdx.x] = f == 0x7fffffff ? -1 : f;
  }
}

}  // namespace tebamd
