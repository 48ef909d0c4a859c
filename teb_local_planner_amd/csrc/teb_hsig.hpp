// teb_hsig.hpp — equivalence classes of the candidate bands (arithmetic core):
//   HSignature3d::calculateHSignature  include/teb_local_planner/h_signature.h:281-347  (x-y-t, Biot-Savart line integral per obstacle)
//   HSignature::calculateHSignature    include/teb_local_planner/h_signature.h:96-188   (2-D complex-log signature)
// as HomotopyClassPlanner::calculateEquivalenceClass (homotopy_class_planner.hpp:46-62) evaluates them for every candidate in
// renewAndAnalyzeOldTebs (src/homotopy_class_planner.cpp:214-254). The reference does this one candidate after the other on the
// CPU: O(n * M * 10) fp64 operations per candidate in 3-D. Here: one launch over the device-resident strips of the whole batch.
#pragma once
#include "teb_strip.hpp"

namespace tebamd {

__device__ __forceinline__ void cross3(const double* a, const double* b, double* o) {   // Eigen: (a1 b2 - a2 b1, a2 b0 - a0 b2, a0 b1 - a1 b0)
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ double sqn3(const double* a) { return ((0.0 + a[0] * a[0]) + a[1] * a[1]) + a[2] * a[2]; }

// 3-D: one lane per (band, obstacle); the lane walks the band's segments and the 10 integration steps per segment in the
// reference's order with the reference's operations (only + - * / sqrt: IEEE-exact, so the result is bit-equal to the CPU's).
// grid = (ceil(M / kThreads), B). Dynamic LDS: 3 * stride doubles (x, y, transition time of every pose; read as broadcasts).
__global__ void __launch_bounds__(kThreads) hsig3d_kernel(const SceneDev sc, const BatchDev bt, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) double hs_lds[];
  const int b = blockIdx.y, S = bt.stride, M = sc.M;
  const int n = bt.n[b];
  double* lx = hs_lds; double* ly = hs_lds + S; double* lt = hs_lds + 2 * S;
  const size_t so = (size_t)b * S;
  for (int i = threadIdx.x; i < n; i += kThreads) { lx[i] = bt.x[so + i]; ly[i] = bt.y[so + i]; lt[i] = (i < n - 1) ? bt.dt[so + i] : 0.0; }
  __syncthreads();
  if (threadIdx.x == 0) {   // next_transition_time += dt, left to right (:313-320); the time differences were staged in parallel
    double t = 0;
    for (int i = 0; i < n; ++i) { const double d = lt[i]; lt[i] = t; t += d; }
  }
  __syncthreads();
  const int l = blockIdx.x * kThreads + threadIdx.x;
  if (l >= M) return;
  const double s1[3] = {sc.cx[l], sc.cy[l], 0.0};
  const double tt = 120;   // "some large value for defining the end point of the obstacle/conductor model"
  const double s2[3] = {sc.cx[l] + tt * sc.vx[l], sc.cy[l] + tt * sc.vy[l], tt};   // predictCentroidConstantVelocity (obstacles.h:187-193)
  const double ds[3] = {s2[0] - s1[0], s2[1] - s1[1], s2[2] - s1[2]};
  const double ds_sq_norm = sqn3(ds);
  double H = 0;
  for (int i = 0; i < n - 1; ++i) {
    const double dir[3] = {lx[i + 1] - lx[i], ly[i + 1] - ly[i], lt[i + 1] - lt[i]};
    if (sqrt(sqn3(dir)) < 1e-15) continue;   // coincident poses
    double r[3] = {lx[i], ly[i], lt[i]};
    const double dl[3] = {dir[0] * (1.0 / 10.0), dir[1] * (1.0 / 10.0), dir[2] * (1.0 / 10.0)};
#pragma unroll 2
    for (int k = 0; k < 10; ++k) {
      double p1[3], p2[3], c12[3], d[3], c2[3], c1[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) { p1[q] = s1[q] - r[q]; p2[q] = s2[q] - r[q]; }
      cross3(p1, p2, c12);
      cross3(ds, c12, d);
#pragma unroll
      for (int q = 0; q < 3; ++q) d[q] = d[q] / ds_sq_norm;
      cross3(d, p2, c2);
      cross3(d, p1, c1);
      const double n2 = sqrt(sqn3(p2)), n1 = sqrt(sqn3(p1));
      const double f = 1.0 / sqn3(d);
      double phi[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) phi[q] = (c2[q] / n2 - c1[q] / n1) * f;
      H += ((0.0 + phi[0] * dl[0]) + phi[1] * dl[1]) + phi[2] * dl[2];
#pragma unroll
      for (int q = 0; q < 3; ++q) r[q] += dl[q];
    }
  }
  out[(size_t)b * M + l] = H / (4.0 * M_PI);
}

// The same signature for a batch that does not fill the chip with one lane per (band, obstacle) - a planning tick has a handful of
// bands and a few dozen obstacles: one workgroup per (band, tile of 16 obstacles), lane = (obstacle o of the tile, segment s of a
// chunk of 16 segments). The Biot-Savart terms of a chunk (10 integration steps per segment, r advanced step by step as in the
// reference) are computed by all 256 lanes at once and staged in LDS; then 16 lanes - one per obstacle - add the chunk's terms to
// their running sum in the reference's order. Every term and every addition is the one hsig3d_kernel performs, in the same order:
// the two kernels return identical bits; only the latency differs (a 5-band, 12-obstacle tick: 1.7 ms -> tens of microseconds).
// grid = (ceil(M / 16), B). Dynamic LDS: 3 * stride doubles.
constexpr int kHsTile = 16, kHsChunk = kThreads / kHsTile;
__global__ void __launch_bounds__(kThreads) hsig3d_small_kernel(const SceneDev sc, const BatchDev bt, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) double hs_lds[];
  __shared__ double terms[10 * kHsChunk * kHsTile];   // [step][segment of the chunk][obstacle of the tile]
  __shared__ int skip[kHsChunk];
  const int b = blockIdx.y, S = bt.stride, M = sc.M;
  const int n = bt.n[b];
  double* lx = hs_lds; double* ly = hs_lds + S; double* lt = hs_lds + 2 * S;
  const size_t so = (size_t)b * S;
  for (int i = threadIdx.x; i < n; i += kThreads) { lx[i] = bt.x[so + i]; ly[i] = bt.y[so + i]; lt[i] = (i < n - 1) ? bt.dt[so + i] : 0.0; }
  __syncthreads();
  if (threadIdx.x == 0) {   // next_transition_time += dt, left to right (:313-320); the time differences were staged in parallel
    double t = 0;
    for (int i = 0; i < n; ++i) { const double d = lt[i]; lt[i] = t; t += d; }
  }
  __syncthreads();
  const int o = threadIdx.x % kHsTile, s = threadIdx.x / kHsTile;
  const int l = blockIdx.x * kHsTile + o;
  const bool lane_ok = l < M;
  const int lc = lane_ok ? l : 0;
  const double s1[3] = {sc.cx[lc], sc.cy[lc], 0.0};
  const double tt = 120;
  const double s2[3] = {sc.cx[lc] + tt * sc.vx[lc], sc.cy[lc] + tt * sc.vy[lc], tt};
  const double ds[3] = {s2[0] - s1[0], s2[1] - s1[1], s2[2] - s1[2]};
  const double ds_sq_norm = sqn3(ds);
  double H = 0;
  for (int i0 = 0; i0 < n - 1; i0 += kHsChunk) {
    const int i = i0 + s;
    if (i < n - 1) {
      const double dir[3] = {lx[i + 1] - lx[i], ly[i + 1] - ly[i], lt[i + 1] - lt[i]};
      const bool coincident = sqrt(sqn3(dir)) < 1e-15;
      if (o == 0) skip[s] = coincident;
      if (!coincident && lane_ok) {
        double r[3] = {lx[i], ly[i], lt[i]};
        const double dl[3] = {dir[0] * (1.0 / 10.0), dir[1] * (1.0 / 10.0), dir[2] * (1.0 / 10.0)};
#pragma unroll 2
        for (int k = 0; k < 10; ++k) {
          double p1[3], p2[3], c12[3], d[3], c2[3], c1[3];
#pragma unroll
          for (int q = 0; q < 3; ++q) { p1[q] = s1[q] - r[q]; p2[q] = s2[q] - r[q]; }
          cross3(p1, p2, c12);
          cross3(ds, c12, d);
#pragma unroll
          for (int q = 0; q < 3; ++q) d[q] = d[q] / ds_sq_norm;
          cross3(d, p2, c2);
          cross3(d, p1, c1);
          const double n2 = sqrt(sqn3(p2)), n1 = sqrt(sqn3(p1));
          const double f = 1.0 / sqn3(d);
          double phi[3];
#pragma unroll
          for (int q = 0; q < 3; ++q) phi[q] = (c2[q] / n2 - c1[q] / n1) * f;
          terms[(k * kHsChunk + s) * kHsTile + o] = ((0.0 + phi[0] * dl[0]) + phi[1] * dl[1]) + phi[2] * dl[2];
#pragma unroll
          for (int q = 0; q < 3; ++q) r[q] += dl[q];
        }
      }
    }
    __syncthreads();
    if (s == 0 && lane_ok) {   // the reference's running sum: segments in order, steps in order
      const int cnt = min(kHsChunk, n - 1 - i0);
      for (int ss = 0; ss < cnt; ++ss) {
        if (skip[ss]) continue;
#pragma unroll
        for (int k = 0; k < 10; ++k) H += terms[(k * kHsChunk + ss) * kHsTile + o];
      }
    }
    __syncthreads();
  }
  if (s == 0 && lane_ok) out[(size_t)b * M + l] = H / (4.0 * M_PI);
}

// ---- 2-D ------------------------------------------------------------------------------------------------------------------
// The reference works in complex<long double> (x87: 64-bit mantissa, 15-bit exponent) because A_l = f0 * prod_j 1 / (o_l - o_j)
// leaves the fp64 range for a few hundred obstacles. fp64 with an explicit binary exponent keeps the range: value = (re, im) * 2^e.
struct CplxE { double re, im; int e; };
__device__ __forceinline__ void cnorm(CplxE& z) {
  const double m = fmax(fabs(z.re), fabs(z.im));
  if (m == 0 || !isfinite(m)) return;
  int k;
  (void)frexp(m, &k);
  z.re = ldexp(z.re, -k); z.im = ldexp(z.im, -k); z.e += k;
}
__device__ __forceinline__ void cdiv(CplxE& z, double br, double bi) {   // z /= (br + i bi)
  const double den = br * br + bi * bi;
  const double re = (z.re * br + z.im * bi) / den, im = (z.im * br - z.re * bi) / den;
  z.re = re; z.im = im;
  cnorm(z);
}

// prod_l = prod_{j != l, |o_l - o_j| >= 0.05} 1 / (o_l - o_j): the band-independent factor of A_l, once per obstacle table
__global__ void hsig2d_prod_kernel(const SceneDev sc, double* __restrict__ pre, double* __restrict__ pim, int* __restrict__ pex) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= sc.M) return;
  CplxE z{1.0, 0.0, 0};
  const double ox = sc.cx[l], oy = sc.cy[l];
  for (int j = 0; j < sc.M; ++j) {
    if (j == l) continue;
    const double dr = ox - sc.cx[j], di = oy - sc.cy[j];
    if (sqrt(dr * dr + di * di) < 0.05) continue;   // skip really close obstacles (:163-164)
    cdiv(z, dr, di);
  }
  pre[l] = z.re; pim[l] = z.im; pex[l] = z.e;
}

// one workgroup per band: lane l (strided) accumulates sum_i A_l * log_value(i, l) over the segments in order, with A_l kept
// as mantissa * 2^e; the lanes' sums are brought to the common exponent of the block and tree-reduced.
__global__ void __launch_bounds__(kThreads) hsig2d_kernel(const SceneDev sc, const BatchDev bt, double prescaler,
                                                           const double* __restrict__ pre, const double* __restrict__ pim,
                                                           const int* __restrict__ pex, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) double hs_lds[];
  __shared__ double red_re[kThreads], red_im[kThreads];
  __shared__ int red_e[kThreads];
  const int b = blockIdx.x, S = bt.stride, M = sc.M;
  const int n = bt.n[b];
  double* lx = hs_lds; double* ly = hs_lds + S;
  const size_t so = (size_t)b * S;
  for (int i = threadIdx.x; i < n; i += kThreads) { lx[i] = bt.x[so + i]; ly[i] = bt.y[so + i]; }
  __syncthreads();
  if (M == 0) { if (threadIdx.x == 0) { out[2 * b] = 0; out[2 * b + 1] = 0; } return; }
  int m = M - 1 > 5 ? M - 1 : 5;
  const int a = (int)ceil(double(m) / 2.0);
  const int bb = m - a;
  // coarse map guess from start and goal (:118-137)
  const double sx = lx[0], sy = ly[0], ex = lx[n - 1], ey = ly[n - 1];
  const double dx = ex - sx, dy = ey - sy;
  double blx, bly, trx, try_;
  if (sqrt(dx * dx + dy * dy) < 3.0) { blx = sx + 0; bly = sy - 3; trx = sx + 3; try_ = sy + 3; }
  else { blx = sx - (-dy); bly = sy - dx; trx = sx + dx + (-dy); try_ = sy + dy + dx; }
  CplxE acc{0.0, 0.0, 0};
  bool have = false;
  for (int l = threadIdx.x; l < M; l += kThreads) {
    const double ox = sc.cx[l], oy = sc.cy[l];
    // f0 = prescaler * a * (o - bl) * b * (o - tr)
    const double ur = ox - blx, ui = oy - bly, vr = ox - trx, vi = oy - try_;
    const double k = prescaler * (double)a * (double)bb;
    CplxE Al{k * (ur * vr - ui * vi), k * (ur * vi + ui * vr), 0};
    cnorm(Al);
    {   // times the band-independent product
      const double re = Al.re * pre[l] - Al.im * pim[l], im = Al.re * pim[l] + Al.im * pre[l];
      Al.re = re; Al.im = im; Al.e += pex[l];
      cnorm(Al);
    }
    double sr = 0, si = 0;   // sum_i log_value(i, l)  (A_l does not depend on the segment)
    for (int i = 0; i < n - 1; ++i) {
      const double a1r = lx[i] - ox, a1i = ly[i] - oy, a2r = lx[i + 1] - ox, a2i = ly[i + 1] - oy;
      const double diff2 = sqrt(a2r * a2r + a2i * a2i), diff1 = sqrt(a1r * a1r + a1i * a1i);
      if (diff2 == 0 || diff1 == 0) continue;
      const double log_real = log(diff2) - log(diff1);
      const double arg_diff = atan2(a2i, a2r) - atan2(a1i, a1r);
      double best = arg_diff;   // std::min_element with smaller_than_abs over {0, +2pi, -2pi, +4pi, -4pi}: first smallest |.|
      double c = arg_diff + 2 * M_PI; if (fabs(c) < fabs(best)) best = c;
      c = arg_diff - 2 * M_PI; if (fabs(c) < fabs(best)) best = c;
      c = arg_diff + 4 * M_PI; if (fabs(c) < fabs(best)) best = c;
      c = arg_diff - 4 * M_PI; if (fabs(c) < fabs(best)) best = c;
      sr += log_real; si += best;
    }
    CplxE term{Al.re * sr - Al.im * si, Al.re * si + Al.im * sr, Al.e};
    if (!have) { acc = term; have = true; }
    else {   // acc += term at the larger exponent
      const int e = acc.e > term.e ? acc.e : term.e;
      acc.re = ldexp(acc.re, acc.e - e) + ldexp(term.re, term.e - e);
      acc.im = ldexp(acc.im, acc.e - e) + ldexp(term.im, term.e - e);
      acc.e = e;
    }
  }
  red_re[threadIdx.x] = have ? acc.re : 0.0; red_im[threadIdx.x] = have ? acc.im : 0.0;
  red_e[threadIdx.x] = have ? acc.e : -100000;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const int e1 = red_e[threadIdx.x], e2 = red_e[threadIdx.x + s];
      const int e = e1 > e2 ? e1 : e2;
      if (e > -100000) {
        const double r1 = e1 > -100000 ? ldexp(red_re[threadIdx.x], e1 - e) : 0.0, i1 = e1 > -100000 ? ldexp(red_im[threadIdx.x], e1 - e) : 0.0;
        const double r2 = e2 > -100000 ? ldexp(red_re[threadIdx.x + s], e2 - e) : 0.0, i2 = e2 > -100000 ? ldexp(red_im[threadIdx.x + s], e2 - e) : 0.0;
        red_re[threadIdx.x] = r1 + r2; red_im[threadIdx.x] = i1 + i2; red_e[threadIdx.x] = e;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int e = red_e[0];
    out[2 * b] = e > -100000 ? ldexp(red_re[0], e) : 0.0;
    out[2 * b + 1] = e > -100000 ? ldexp(red_im[0], e) : 0.0;
  }
}

}  // namespace tebamd
