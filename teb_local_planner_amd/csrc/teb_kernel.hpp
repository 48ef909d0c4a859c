// teb_kernel.hpp — the fused optimizeTEB kernel: one workgroup (4 wave64) per candidate TEB, the whole
// outer loop (autoResize -> association -> LM iterations -> cost) in ONE launch, no host round trips.
//
// Replaces TebOptimalPlanner::optimizeTEB (src/optimal_planner.cpp:182-231) + the g2o back end it drives
//  for a batch of candidates (HomotopyClassPlanner::optimizeAllTEBs,
// src/homotopy_class_planner.cpp:466-493).
//
// LDS per workgroup (S = stride = max poses):
//   sx sy sth sdt tdyn : 5*S      state strip (+ time stamps of the dynamic-obstacle edges)
//   Hb                 : 44*S     banded normal matrix, canonical variable order var(i,c) = 4i+c,
//                                 row r holds H(r, r-d), d = 0..10 at Hb[r*11+d]; factored in place (LDL^T)
//   bv, dxv            : 4*S each right-hand side b = -J^T Omega e and the LM step
//   red                : 64       reduction scratch
#pragma once
#include "teb_edges.hpp"

namespace tebamd {

struct Lds {
  double *sx, *sy, *sth, *sdt, *tdyn, *Hb, *bv, *dxv, *red;
  int* ired;
};

__host__ __device__ inline size_t lds_bytes_for(int S) {
  return sizeof(double) * ((size_t)5 * S + (size_t)4 * S * kBand + (size_t)8 * S + 64) + 64 * sizeof(int);
}

__device__ __forceinline__ Lds carve(double* base, int S) {
  Lds l;
  l.sx = base; l.sy = l.sx + S; l.sth = l.sy + S; l.sdt = l.sth + S; l.tdyn = l.sdt + S;
  l.Hb = l.tdyn + S;
  l.bv = l.Hb + (size_t)4 * S * kBand;
  l.dxv = l.bv + 4 * S;
  l.red = l.dxv + 4 * S;
  l.ired = reinterpret_cast<int*>(l.red + 64);
  return l;
}

// ---- deterministic block reductions (fixed tree: lanes via shuffles, then the 4 waves in order) ---------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
template <int K>
__device__ __forceinline__ void block_sum(double* v, double* red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < K; ++q) {
    double s = wave_sum(v[q]);
    if (lane == 0) red[q * 4 + wv] = s;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < K; ++q) v[q] = ((red[q * 4 + 0] + red[q * 4 + 1]) + red[q * 4 + 2]) + red[q * 4 + 3];
  __syncthreads();
}
__device__ __forceinline__ double block_max(double v, double* red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  if (lane == 0) red[wv] = v;
  __syncthreads();
  double r = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  __syncthreads();
  return r;
}
// argmin with lowest-index tie break (sequential scan with strict '<' keeps the first minimum)
__device__ __forceinline__ int block_argmin(double v, int idx, double* red, int* ired) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double ov = __shfl_down(v, off, 64);
    int oi = __shfl_down(idx, off, 64);
    if (ov < v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
  if (lane == 0) { red[wv] = v; ired[wv] = idx; }
  __syncthreads();
  double bv = red[0];
  int bi = ired[0];
#pragma unroll
  for (int q = 1; q < 4; ++q)
    if (red[q] < bv || (red[q] == bv && ired[q] < bi)) { bv = red[q]; bi = ired[q]; }
  __syncthreads();
  return bi;
}

// ---- per-TEB constants visible to the edge evaluation -----------------------------------------------------
struct TebCtx {
  int b, n;
  int has_vs, has_vg, rotdir, via_en;
  double vs[3], vg[3];
  double w_obst;       // weight_obstacle * weight_multiplier
  bool inflated;
  const int* assoc_cnt;   // + b*stride
  const int* assoc;       // + b*cap*stride
  const int* via_pose;    // + b*via_cap
  int stride;
};

// All cost terms whose first vertex is pose i / timediff i (0 <= i <= n-2). JAC=true also accumulates
// J^T Omega J and J^T Omega e into the thread-local window accumulator.
template <bool JAC>
__device__ __forceinline__ void eval_index(const teb_amd_config_t& c, const SceneDev& sc, const TebCtx& t,
                                           const Lds& l, int i, Accum& A) {
  const int n = t.n;
  Win w;
  w.x0 = l.sx[i]; w.y0 = l.sy[i]; w.t0 = l.sth[i]; w.d0 = l.sdt[i];
  w.x1 = l.sx[i + 1]; w.y1 = l.sy[i + 1]; w.t1 = l.sth[i + 1];
  const bool has2 = (i + 2 <= n - 1);
  w.d1 = has2 ? l.sdt[i + 1] : 1.0;
  w.x2 = has2 ? l.sx[i + 2] : 0.0; w.y2 = has2 ? l.sy[i + 2] : 0.0; w.t2 = has2 ? l.sth[i + 2] : 0.0;
  const bool seg_active = (n > 2);   // g2o never activates an edge whose vertices are all fixed (n == 2)

  // ---- unary edges of pose i (AddEdgesObstacles :444-548, AddEdgesDynamicObstacles :646-673, AddEdgesViaPoints :675-718)
  const int cnt = (t.assoc_cnt != nullptr) ? t.assoc_cnt[i] : 0;
  if (i >= 1) {
    for (int k = 0; k < cnt; ++k) {
      int oi = t.assoc[(size_t)k * t.stride + i];
      edge_obstacle<JAC>(c, sc, oi, w, t.w_obst, t.inflated, A);
    }
    if (c.include_dynamic_obstacles && c.weight_obstacle != 0) {
      const double ti = l.tdyn[i];
      for (int k = 0; k < sc.n_dyn; ++k) edge_dynamic_obstacle<JAC>(c, sc, sc.dyn_idx[k], w, ti, A);
    }
    if (t.via_en && c.weight_viapoint != 0) {
      for (int v = 0; v < sc.nvia; ++v)
        if (t.via_pose[v] == i) edge_via_point<JAC>(c, sc.viax[v], sc.viay[v], w, A);
    }
  }
  // ---- AddEdgesVelocity :720-769
  if (c.max_vel_y == 0) {
    if (!(c.weight_max_vel_x == 0 && c.weight_max_vel_theta == 0)) edge_velocity<JAC>(c, w, A);
  } else {
    if (!(c.weight_max_vel_x == 0 && c.weight_max_vel_y == 0 && c.weight_max_vel_theta == 0))
      edge_velocity_holonomic<JAC>(c, w, A);
  }
  // ---- AddEdgesAcceleration :771-873
  if (!(c.weight_acc_lim_x == 0 && c.weight_acc_lim_theta == 0)) {
    const bool nonholo = (c.max_vel_y == 0 || c.acc_lim_y == 0);
    if (nonholo) {
      if (i == 0 && t.has_vs) edge_acceleration_se<JAC, true>(c, w, t.vs[0], t.vs[2], A);
      if (has2) edge_acceleration<JAC>(c, w, A);
      if (i == n - 2 && t.has_vg) edge_acceleration_se<JAC, false>(c, w, t.vg[0], t.vg[2], A);
    } else {
      if (i == 0 && t.has_vs) edge_acceleration_holonomic_se<JAC, true>(c, w, t.vs, A);
      if (has2) edge_acceleration_holonomic<JAC>(c, w, A);
      if (i == n - 2 && t.has_vg) edge_acceleration_holonomic_se<JAC, false>(c, w, t.vg, A);
    }
  }
  // ---- AddEdgesTimeOptimal :877-893, AddEdgesShortestPath :895-912
  if (c.weight_optimaltime != 0) edge_time_optimal<JAC>(c, w, A);
  if (c.weight_shortest_path != 0 && seg_active) edge_shortest_path<JAC>(c, w, A);
  // ---- kinematics :355-358, 916-958
  if (seg_active) {
    if (c.min_turning_radius == 0 || c.weight_kinematics_turning_radius == 0) {
      if (!(c.weight_kinematics_nh == 0 && c.weight_kinematics_forward_drive == 0)) edge_kinematics_diffdrive<JAC>(c, w, A);
    } else {
      if (!(c.weight_kinematics_nh == 0 && c.weight_kinematics_turning_radius == 0)) edge_kinematics_carlike<JAC>(c, w, A);
    }
  }
  // ---- AddEdgesPreferRotDir :961-997
  if (i < 3 && seg_active && c.weight_prefer_rotdir != 0 && (t.rotdir == TEB_AMD_ROT_LEFT || t.rotdir == TEB_AMD_ROT_RIGHT))
    edge_prefer_rotdir<JAC>(c, w, t.rotdir == TEB_AMD_ROT_LEFT ? 1.0 : -1.0, A);
  // ---- AddEdgesVelocityObstacleRatio :999-1021
  if (c.weight_velocity_obstacle_ratio > 0) {
    for (int k = 0; k < cnt; ++k) {
      int oi = t.assoc[(size_t)k * t.stride + i];
      edge_velocity_obstacle_ratio<JAC>(c, sc, oi, w, A);
    }
  }
}

// scatter the thread-local window into the LDS band; rows/cols of fixed variables are dropped
__device__ __forceinline__ void scatter(const Accum& A, const Lds& l, int i, int n) {
  const int base = 4 * i;
  const int last_pose = 4 * (n - 1);
#pragma unroll
  for (int a = 0; a < 11; ++a) {
    int ra = base + a;
    bool fa = (ra < 3) || (ra >= last_pose);
    if (fa) continue;
    l.bv[ra] -= A.g[a];
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      int rb = base + b;
      bool fb = (rb < 3) || (rb >= last_pose);
      if (fb) continue;
      l.Hb[ra * kBand + (a - b)] += A.H[a * (a + 1) / 2 + b];
    }
  }
}

// buildSystem: H = sum J^T Omega J, b = -sum J^T Omega e, and chi^2 per category at the current state.
__device__ inline void linearize(const teb_amd_config_t& c, const SceneDev& sc, const TebCtx& t, const Lds& l,
                                 double* cats /*4, out on all threads*/) {
  const int n = t.n, Nt = 4 * n, tid = threadIdx.x;
  for (int q = tid; q < Nt * kBand; q += kThreads) l.Hb[q] = 0;
  for (int q = tid; q < Nt; q += kThreads) l.bv[q] = 0;
  __syncthreads();
  Accum A;
  A.clear_chi();
  for (int k0 = 0; k0 < n - 1; k0 += kThreads) {
    const int i = k0 + tid;
    const bool active = i <= n - 2;
    A.clear();
    if (active) eval_index<true>(c, sc, t, l, i, A);
    for (int ph = 0; ph < 3; ++ph) {
      if (active && (i % 3) == ph) scatter(A, l, i, n);
      __syncthreads();
    }
  }
  // fixed variables (pose 0, pose n-1, the non-existing dt_{n-1}) become identity rows
  if (tid < 3) { l.Hb[tid * kBand] = 1.0; l.bv[tid] = 0; }
  if (tid >= 4 && tid < 8) { int r = 4 * (n - 1) + (tid - 4); l.Hb[r * kBand] = 1.0; l.bv[r] = 0; }
  cats[0] = A.chi[0]; cats[1] = A.chi[1]; cats[2] = A.chi[2]; cats[3] = A.chi[3];
  block_sum<4>(cats, l.red);
}

// computeActiveErrors + activeRobustChi2 at the current state
__device__ inline void evaluate(const teb_amd_config_t& c, const SceneDev& sc, const TebCtx& t, const Lds& l,
                                double* cats) {
  Accum A;   // only chi[] is live when JAC == false
  A.clear_chi();
  for (int i = threadIdx.x; i <= t.n - 2; i += kThreads) eval_index<false>(c, sc, t, l, i, A);
  cats[0] = A.chi[0]; cats[1] = A.chi[1]; cats[2] = A.chi[2]; cats[3] = A.chi[3];
  block_sum<4>(cats, l.red);
}

// ---- damped solve (K6 v1): in-LDS banded LDL^T by wave 0, right-looking, 65 work items per pivot ------------
// Solves (H + lambda I) dx = b. H is destroyed (L below the diagonal, D on it). Returns false iff a pivot
// d <= 0 (or NaN) is met, the condition under which CSparse's cs_chol gives up.
__device__ inline bool banded_ldlt_solve_wave0(const Lds& l, int Nt, double lambda) {
  const int lane = threadIdx.x;   // caller guarantees threadIdx.x < 64
  double* H = l.Hb;
  double* x = l.dxv;
  for (int r = lane; r < Nt; r += 64) x[r] = l.bv[r];
  // decode work item -> (i, j): items 0..54 trailing entry (i >= j >= 1), 55..64 rhs update of row i
  int wi, wj;
  {
    int w = lane, i = 1;
    while (w >= i && i <= 10) { w -= i; ++i; }
    if (i <= 10) { wi = i; wj = w + 1; }
    else { wi = lane - 54; wj = 0; }   // lanes 55..63 -> rhs rows 1..9 ; rhs row 10 handled by lane 0 below
  }
  __builtin_amdgcn_wave_barrier();
  bool ok = true;
  for (int k = 0; k < Nt; ++k) {
    const double d = H[k * kBand] + lambda;
    if (!(d > 0)) { ok = false; break; }
    const double inv = 1.0 / d;
    const double xk = x[k];
    if (k + wi < Nt) {
      const double ci = H[(k + wi) * kBand + wi];
      if (wj > 0) {
        const double cj = H[(k + wj) * kBand + wj];
        H[(k + wi) * kBand + (wi - wj)] -= (ci * inv) * cj;
      } else {
        x[k + wi] -= (ci * inv) * xk;
      }
    }
    if (lane == 0 && k + 10 < Nt) x[k + 10] -= (H[(k + 10) * kBand + 10] * inv) * xk;
    __builtin_amdgcn_wave_barrier();
    if (lane < 10 && k + lane + 1 < Nt) H[(k + lane + 1) * kBand + lane + 1] *= inv;   // L(k+i, k)
    if (lane == 0) H[k * kBand] = d;
    __builtin_amdgcn_wave_barrier();
  }
  if (!ok) return false;
  for (int r = lane; r < Nt; r += 64) x[r] = x[r] / H[r * kBand];
  __builtin_amdgcn_wave_barrier();
  for (int k = Nt - 1; k > 0; --k) {
    const double xk = x[k];
    if (lane < 10 && k - lane - 1 >= 0) x[k - lane - 1] -= H[k * kBand + lane + 1] * xk;
    __builtin_amdgcn_wave_barrier();
  }
  return true;
}

// ---- TimedElasticBand::autoResize (src/timed_elastic_band.cpp:227-286) -------------------------------------
// One sweep, executed by thread 0 as a streaming pass LDS strip -> global scratch (no O(n) inserts).
// Reproduces the sequential i-- re-check semantics: `cur` is the interval under test, `stack` holds the
// right halves produced by splits that still wait to be visited.
__device__ inline void autoresize_sweep_thread0(const teb_amd_config_t& c, const Lds& l, int n_in, double* ox,
                                                double* oy, double* oth, double* odt, double* stk, int stride,
                                                int* n_out, int* modified_out, int* overflow) {
  const double dt_ref = c.dt_ref, hyst = c.dt_hysteresis;
  const int Tin = n_in - 1;
  int T = Tin;           // sizeTimeDiffs()
  int j = 1;             // next unread input interval
  int sp = 0;            // stack size (entries: x, y, th, dt)
  int k = 0;             // emitted intervals
  bool modified = false;
  double cx = l.sx[0], cy = l.sy[0], cth = l.sth[0], cdt = l.sdt[0];
  const double gx = l.sx[n_in - 1], gy = l.sy[n_in - 1], gth = l.sth[n_in - 1];
  bool alive = Tin >= 1;
  while (alive) {
    const bool has_next = (sp > 0) || (j < Tin);
    if (cdt > dt_ref + hyst && T < c.max_samples) {
      if (cdt > 2 * dt_ref) {
        double newtime = 0.5 * cdt;
        double ex, ey, eth;   // Pose(i+1)
        if (sp > 0) { ex = stk[4 * (sp - 1)]; ey = stk[4 * (sp - 1) + 1]; eth = stk[4 * (sp - 1) + 2]; }
        else if (j < Tin) { ex = l.sx[j]; ey = l.sy[j]; eth = l.sth[j]; }
        else { ex = gx; ey = gy; eth = gth; }
        if (sp >= 64) { *overflow = 1; break; }
        // PoseSE2::average (pose_se2.h:266-269) with g2o::average_angle
        double sxn = cos(cth) + cos(eth), syn = sin(cth) + sin(eth);
        stk[4 * sp] = (cx + ex) / 2; stk[4 * sp + 1] = (cy + ey) / 2;
        stk[4 * sp + 2] = (sxn == 0 && syn == 0) ? 0.0 : atan2(syn, sxn);
        stk[4 * sp + 3] = newtime;
        ++sp;
        cdt = newtime;
        ++T;
        modified = true;
        continue;   // i-- : re-check the left half
      } else {
        if (has_next) {
          if (sp > 0) stk[4 * (sp - 1) + 3] += cdt - dt_ref;
          else l.sdt[j] += cdt - dt_ref;
        }
        cdt = dt_ref;
      }
    } else if (cdt < dt_ref - hyst && T > c.min_samples) {
      if (has_next) {
        // TimeDiff(i+1) += TimeDiff(i); deleteTimeDiff(i); deletePose(i+1); i--
        if (sp > 0) { cdt = stk[4 * (sp - 1) + 3] + cdt; --sp; }
        else { cdt = l.sdt[j] + cdt; ++j; }
        --T;
        modified = true;
        continue;
      } else if (k > 0) {
        // last interval: TimeDiff(i-1) += TimeDiff(i); deleteTimeDiff(i); deletePose(i)
        odt[k - 1] += cdt;
        --T;
        modified = true;
        alive = false;
        break;
      }
    }
    // emit cur, advance
    if (k >= stride - 1) { *overflow = 1; break; }
    ox[k] = cx; oy[k] = cy; oth[k] = cth; odt[k] = cdt;
    ++k;
    if (sp > 0) { --sp; cx = stk[4 * sp]; cy = stk[4 * sp + 1]; cth = stk[4 * sp + 2]; cdt = stk[4 * sp + 3]; }
    else if (j < Tin) { cx = l.sx[j]; cy = l.sy[j]; cth = l.sth[j]; cdt = l.sdt[j]; ++j; }
    else alive = false;
  }
  ox[k] = gx; oy[k] = gy; oth[k] = gth;
  *n_out = k + 1;
  *modified_out = modified ? 1 : 0;
}

__device__ inline int autoresize(const teb_amd_config_t& c, const Lds& l, int n, double* scratch, int stride,
                                 bool fast_mode, int* overflow_flag) {
  const int tid = threadIdx.x;
  double* ox = scratch; double* oy = ox + stride; double* oth = oy + stride; double* odt = oth + stride;
  double* stk = odt + stride;
  for (int rep = 0; rep < 100; ++rep) {
    // parallel pre-check: a sweep is a no-op iff no interval satisfies either trigger condition
    const int T = n - 1;
    int trig = 0;
    for (int i = tid; i < T; i += kThreads) {
      double d = l.sdt[i];
      if ((d > c.dt_ref + c.dt_hysteresis && T < c.max_samples) || (d < c.dt_ref - c.dt_hysteresis && T > c.min_samples)) trig = 1;
    }
    trig = __syncthreads_or(trig);
    if (!trig) break;
    if (tid == 0) {
      int n_out = n, mod = 0, ovf = 0;
      autoresize_sweep_thread0(c, l, n, ox, oy, oth, odt, stk, stride, &n_out, &mod, &ovf);
      l.ired[8] = n_out; l.ired[9] = mod; l.ired[10] = ovf;
      __threadfence_block();
    }
    __syncthreads();
    const int n_out = l.ired[8], mod = l.ired[9], ovf = l.ired[10];
    if (ovf) { *overflow_flag = 1; return n; }
    n = n_out;
    for (int i = tid; i < n; i += kThreads) {
      l.sx[i] = ox[i]; l.sy[i] = oy[i]; l.sth[i] = oth[i];
      if (i < n - 1) l.sdt[i] = odt[i];
    }
    __syncthreads();
    if (!mod || fast_mode) break;
  }
  return n;
}

// ---- obstacle association, AddEdgesObstacles (src/optimal_planner.cpp:444-548) -------------------------------
__device__ inline void associate(const teb_amd_config_t& c, const SceneDev& sc, const Lds& l, int n, int* assoc_cnt,
                                 int* assoc, int cap, int stride, int* overflow) {
  const int first_vertex = c.weight_velocity_obstacle_ratio == 0 ? 1 : 0;
  for (int i = threadIdx.x; i < n; i += kThreads) {
    int cnt = 0;
    if (i >= first_vertex && i < n - 1) {
      const double x = l.sx[i], y = l.sy[i], th = l.sth[i];
      const double ox_ = cos(th), oy_ = sin(th);
      double left_min = 1.7976931348623157e308, right_min = 1.7976931348623157e308;
      int left = -1, right = -1;
      const double force = c.min_obstacle_dist * c.obstacle_association_force_inclusion_factor;
      const double cutoff = c.min_obstacle_dist * c.obstacle_association_cutoff_factor;
      for (int k = 0; k < sc.n_static; ++k) {
        const int oi = sc.static_idx[k];
        double dist = footprint_distance(c, sc, oi, x, y, th, false, 0.0, nullptr);
        if (dist < force) {
          if (cnt < cap) assoc[(size_t)cnt * stride + i] = oi; else *overflow = 1;
          ++cnt;
          continue;
        }
        if (dist > cutoff) continue;
        // cross2d(pose_orient, centroid - position) > 0 -> left (misc.h:119-123)
        double vx_ = sc.cx[oi] - x, vy_ = sc.cy[oi] - y;
        if (ox_ * vy_ - vx_ * oy_ > 0) { if (dist < left_min) { left_min = dist; left = oi; } }
        else { if (dist < right_min) { right_min = dist; right = oi; } }
      }
      if (left >= 0) { if (cnt < cap) assoc[(size_t)cnt * stride + i] = left; else *overflow = 1; ++cnt; }
      if (right >= 0) { if (cnt < cap) assoc[(size_t)cnt * stride + i] = right; else *overflow = 1; ++cnt; }
      if (cnt > cap) cnt = cap;
    }
    assoc_cnt[i] = cnt;
  }
}

// =================================================================================================================
__global__ void __launch_bounds__(kThreads)
teb_optimize_kernel(const teb_amd_config_t c, const SceneDev sc, const BatchDev bt, const OptArgs args) {
  extern __shared__ __attribute__((aligned(16))) double lds_base[];
  const int b = blockIdx.x, tid = threadIdx.x, S = bt.stride;
  const Lds l = carve(lds_base, S);
  int n = bt.n[b];
  const size_t so = (size_t)b * S;

  // ---- K0: strip load, coalesced 8 B / lane
  for (int i = tid; i < n; i += kThreads) {
    l.sx[i] = bt.x[so + i]; l.sy[i] = bt.y[so + i]; l.sth[i] = bt.th[so + i];
    l.sdt[i] = (i < n - 1) ? bt.dt[so + i] : 0.0;
  }
  __syncthreads();

  TebCtx t;
  t.b = b; t.stride = S;
  t.has_vs = bt.has_vs[b]; t.has_vg = bt.has_vg[b]; t.rotdir = bt.rotdir[b]; t.via_en = bt.via_en[b];
#pragma unroll
  for (int q = 0; q < 3; ++q) { t.vs[q] = bt.vs[3 * b + q]; t.vg[q] = bt.vg[3 * b + q]; }
  t.inflated = c.inflation_dist > c.min_obstacle_dist;
  int* assoc_cnt = bt.assoc_cnt + so;
  int* assoc = bt.assoc + (size_t)b * bt.assoc_cap * S;
  int* via_pose = bt.via_pose + (size_t)b * bt.via_cap;
  t.assoc_cnt = assoc_cnt; t.assoc = assoc; t.via_pose = via_pose;
  double* Hbk = bt.Hbackup + (size_t)b * 4 * S * kBand;
  double* rs = bt.rs_scratch + (size_t)b * (4 * (size_t)S + 256);

  int status = TEB_AMD_TEB_OK, iters = 0, trials = 0;
  double chi2_final = 0, lambda = 0, cost = __longlong_as_double(0x7ff8000000000000LL);
  double last_cats[4] = {0, 0, 0, 0};
  double weight_multiplier = args.debug_linearize ? args.debug_weight_multiplier : 1.0;
  const bool fast_mode = !c.include_dynamic_obstacles;
  bool done = false;

  if (!c.optimization_activate) { status = TEB_AMD_TEB_FAILED; done = true; }

  for (int outer = 0; outer < args.outer && !done; ++outer) {
    // ---- K1: autoResize
    if (c.teb_autosize && !args.debug_linearize) {
      int ovf = 0;
      n = autoresize(c, l, n, rs, S, fast_mode, &ovf);
      if (ovf) { status = TEB_AMD_TEB_FAILED; if (tid == 0) bt.assoc_overflow[b] |= 2; break; }
    }
    t.n = n;
    // optimizeGraph guards (src/optimal_planner.cpp:370-382)
    if (c.max_vel_x < 0.01 || n < 2 || n < c.min_samples) { status = TEB_AMD_TEB_FAILED; break; }
    // ---- buildGraph side data: association (K2), dynamic-obstacle time stamps, via-point attachment
    t.w_obst = c.weight_obstacle * weight_multiplier;
    const bool obst_edges = !(c.weight_obstacle == 0 || weight_multiplier == 0) && !c.legacy_obstacle_association;
    if (obst_edges) {
      int ovf = 0;
      associate(c, sc, l, n, assoc_cnt, assoc, bt.assoc_cap, S, &ovf);
      if (ovf && tid < kThreads) bt.assoc_overflow[b] |= 1;
    } else {
      for (int i = tid; i < n; i += kThreads) assoc_cnt[i] = 0;
    }
    if (tid == 0) {   // :662-670, sequential left-to-right sum like the reference
      double time = l.sdt[0];
      for (int i = 1; i < n - 1; ++i) { l.tdyn[i] = time; time += l.sdt[i]; }
    }
    if (t.via_en && c.weight_viapoint != 0 && sc.nvia > 0 && n >= 3) {   // :675-718
      int start_pose_idx = 0;
      for (int v = 0; v < sc.nvia; ++v) {
        int index = -1;
        if (start_pose_idx >= 0 && start_pose_idx < n) {   // findClosestTrajectoryPose, timed_elastic_band.cpp:455-478
          double best = 1.7976931348623157e308; int bi = 0x7fffffff;
          const double vx = sc.viax[v], vy = sc.viay[v];
          for (int i = start_pose_idx + tid; i < n; i += kThreads) {
            double ddx = vx - l.sx[i], ddy = vy - l.sy[i];
            double d2 = ddx * ddx + ddy * ddy;
            if (d2 < best) { best = d2; bi = i; }
          }
          index = block_argmin(best, bi, l.red, l.ired);
          if (index == 0x7fffffff) index = -1;
        }
        if (c.via_points_ordered) start_pose_idx = index + 2;
        if (index > n - 2) index = n - 2;
        int attach = index;
        if (index < 1) attach = c.via_points_ordered ? 1 : -1;
        if (tid == 0) via_pose[v] = attach;
      }
    } else {
      for (int v = tid; v < sc.nvia; v += kThreads) via_pose[v] = -1;
    }
    __threadfence_block();
    __syncthreads();

    // ---- optimize(): Levenberg-Marquardt
    if (args.inner <= 0 && !args.debug_linearize) { status = TEB_AMD_TEB_FAILED; break; }   // optimize(0) returns 0
    double ni = 2;
    bool lm_ok = true;
    for (int it = 0; it < args.inner && lm_ok; ++it) {
      double cats[4];
      linearize(c, sc, t, l, cats);
      double currentChi = ((cats[0] + cats[1]) + cats[2]) + cats[3];
      if (args.debug_linearize) {
        if (b == 0) {
          const int Nt = 4 * n;
          for (int q = tid; q < Nt * kBand; q += kThreads) args.dbg_H[q] = l.Hb[q];
          for (int q = tid; q < Nt; q += kThreads) args.dbg_b[q] = l.bv[q];
          if (tid < 4) args.dbg_chi2[tid] = cats[tid];
        }
        done = true;
        break;
      }
      const int Nt = 4 * n;
      if (it == 0) {   // computeLambdaInit: tau * max |H_ii| over the free variables
        double m = 0;
        for (int r = tid; r < Nt; r += kThreads)
          if (r >= 3 && r < 4 * (n - 1)) m = fmax(m, fabs(l.Hb[r * kBand]));
        lambda = 1e-5 * block_max(m, l.red);
        ni = 2;
      }
      for (int q = tid; q < Nt * kBand; q += kThreads) Hbk[q] = l.Hb[q];   // saved for rejected trials
      double rho = 0;
      int qmax = 0;
      do {
        // --- damped solve
        if (tid < 64) {
          bool ok = banded_ldlt_solve_wave0(l, Nt, lambda);
          if (tid == 0) l.ired[0] = ok ? 1 : 0;
        }
        __syncthreads();
        const bool ok2 = l.ired[0] != 0;
        if (!ok2) { for (int r = tid; r < Nt; r += kThreads) l.dxv[r] = l.bv[r]; __syncthreads(); }
        // --- push + oplus (vertex_pose.h:195-198, vertex_timediff.h:113-116)
        double bx_[kMaxPoseIter], by_[kMaxPoseIter], bth_[kMaxPoseIter], bdt_[kMaxPoseIter];
        double sc_part = 0;
#pragma unroll
        for (int kk = 0; kk < kMaxPoseIter; ++kk) {
          const int i = tid + kk * kThreads;
          if (i < n) {
            bx_[kk] = l.sx[i]; by_[kk] = l.sy[i]; bth_[kk] = l.sth[i]; bdt_[kk] = l.sdt[i];
            if (i >= 1 && i <= n - 2) {
              l.sx[i] += l.dxv[4 * i]; l.sy[i] += l.dxv[4 * i + 1];
              l.sth[i] = normalize_theta(l.sth[i] + l.dxv[4 * i + 2]);
            }
            if (i <= n - 2) l.sdt[i] += l.dxv[4 * i + 3];
#pragma unroll
            for (int q = 0; q < 4; ++q) {   // computeScale: sum x_j (lambda x_j + b_j)
              double xj = l.dxv[4 * i + q];
              sc_part += xj * (lambda * xj + l.bv[4 * i + q]);
            }
          }
        }
        __syncthreads();
        double tc[5];
        evaluate(c, sc, t, l, tc);
        last_cats[0] = tc[0]; last_cats[1] = tc[1]; last_cats[2] = tc[2]; last_cats[3] = tc[3];
        double tempChi = ((tc[0] + tc[1]) + tc[2]) + tc[3];
        double scv[1] = {sc_part};
        block_sum<1>(scv, l.red);
        if (!ok2) tempChi = 1.7976931348623157e308;
        rho = (currentChi - tempChi);
        double scale = scv[0] + 1e-3;
        rho /= scale;
        ++trials;
        if (rho > 0 && isfinite(tempChi)) {
          double alpha = 1. - pow((2 * rho - 1), 3);
          alpha = fmin(alpha, 2. / 3.);
          double scaleFactor = fmax(1. / 3., alpha);
          lambda *= scaleFactor;
          ni = 2;
          currentChi = tempChi;
        } else {
          lambda *= ni;
          ni *= 2;
          // pop
#pragma unroll
          for (int kk = 0; kk < kMaxPoseIter; ++kk) {
            const int i = tid + kk * kThreads;
            if (i < n) { l.sx[i] = bx_[kk]; l.sy[i] = by_[kk]; l.sth[i] = bth_[kk]; l.sdt[i] = bdt_[kk]; }
          }
          if (!isfinite(lambda)) { ++qmax; __syncthreads(); break; }
          if (rho < 0 && qmax + 1 < 10)   // another trial follows: bring back the un-factored H
            for (int q = tid; q < Nt * kBand; q += kThreads) l.Hb[q] = Hbk[q];
        }
        __syncthreads();
        qmax++;
      } while (rho < 0 && qmax < 10);
      ++iters;
      chi2_final = currentChi;
      if (qmax == 10 || rho == 0 || !isfinite(lambda)) lm_ok = false;   // Terminate
      if (c.divergence_detection_enable) {   // setComputeBatchStatistics -> computeActiveErrors after each solve
        double fc[4];
        evaluate(c, sc, t, l, fc);
        last_cats[0] = fc[0]; last_cats[1] = fc[1]; last_cats[2] = fc[2]; last_cats[3] = fc[3];
        chi2_final = ((fc[0] + fc[1]) + fc[2]) + fc[3];
      }
    }
    if (done) break;
    // ---- computeCurrentCost (src/optimal_planner.cpp:1041-1094) on the stored errors of the last evaluation
    if (args.compute_cost && outer == args.outer - 1) {
      double cst = 0;
      if (args.alt_time) {
        double s[1] = {0};
        for (int i = tid; i < n - 1; i += kThreads) s[0] += l.sdt[i];
        block_sum<1>(s, l.red);
        cst += s[0];
      }
      cst += last_cats[CAT_OBST] * args.obst_scale;
      cst += last_cats[CAT_VIA] * args.via_scale;
      if (!args.alt_time) cst += last_cats[CAT_TIME];
      cst += last_cats[CAT_OTHER];
      cost = cst;
    }
    weight_multiplier *= c.weight_adapt_factor;
  }

  // ---- K0: strip store + results
  __syncthreads();
  int nonfinite = 0;
  for (int i = tid; i < n; i += kThreads) {
    double x = l.sx[i], y = l.sy[i], th = l.sth[i], d = (i < n - 1) ? l.sdt[i] : 0.0;
    bt.x[so + i] = x; bt.y[so + i] = y; bt.th[so + i] = th; bt.dt[so + i] = d;
    if (!(isfinite(x) && isfinite(y) && isfinite(th) && isfinite(d))) nonfinite = 1;
  }
  nonfinite = __syncthreads_or(nonfinite);
  if (tid == 0) {
    if (nonfinite) status = TEB_AMD_TEB_NONFINITE;
    bt.n[b] = n;
    bt.status[b] = status; bt.iters[b] = iters; bt.trials[b] = trials;
    bt.chi2[b] = chi2_final; bt.cost[b] = cost; bt.lambda[b] = lambda;
  }
}

}  // namespace tebamd
