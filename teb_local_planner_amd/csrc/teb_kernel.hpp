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
#include "teb_multicu.hpp"
#include "teb_autoresize_chain.hpp"

namespace tebamd {

#ifdef TEB_PROFILE
#define PROF_DECL long long prof_t0 = 0, prof_wg_t0 = clock64(), prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PROF_START() prof_t0 = clock64()
#define PROF_END(k) prof_acc[k] += clock64() - prof_t0
#else
// The phase split of the PRODUCT kernel: with OptArgs::phase_log set, lane 0 of every workgroup adds up the shader
// cycles (s_memtime) between the seven phase boundaries of the LM loop in ten words of the reduction scratch (LDS: no register lives
// across the loop for it) and writes them out with the results - ~ 14 clock reads per LM iteration, < 1 % of the launch; off (nullptr,
// the default): one scalar branch per boundary. The -DTEB_PROFILE build keeps the finer counters (rounds of the solve, edge groups).
#define PROF_DECL const bool plog_ = args.phase_log != nullptr; long long* const plog_acc_ = reinterpret_cast<long long*>(l.red + 40); \
  if (plog_ && threadIdx.x == 0) { for (int q_ = 0; q_ < 9; ++q_) plog_acc_[q_] = 0; plog_acc_[9] = clock64(); }
#define PROF_START() do { if (plog_ && threadIdx.x == 0) plog_acc_[8] = clock64(); } while (0)
#define PROF_END(k) do { if (plog_ && threadIdx.x == 0) plog_acc_[k] += clock64() - plog_acc_[8]; } while (0)
#endif

#ifdef TEB_PROFILE
__device__ long long g_ev_prof[8];   // thread 1 (pose 1) of workgroup 0: evaluate {static, dynamic, chain}, linearise {static, dynamic, chain}, trig, scatter
#define EVP_DECL long long evp_t0 = clock64(), evp_t1;
#define EVP(k) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); evp_t1 = clock64(); __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 0 && threadIdx.x == 1) g_ev_prof[k] += evp_t1 - evp_t0; evp_t0 = evp_t1; } while (0)
__device__ unsigned long long g_assoc_stats[4];   // generic association: candidates after the far-field cull, exact distances computed
__device__ unsigned long long g_ar_steps[4];   // autoResize machine, workgroup 0: -, steps, calls, cycles
__device__ unsigned long long g_near_recomputed, g_near_queries;   // lanes that recomputed their near mask / that asked for it
__device__ long long g_lin_prof[16];  // thread 0 of workgroup 0: sections of linearize()
#define LNP_DECL long long lnp_t0 = clock64(), lnp_t1;
#define LNP(k) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); lnp_t1 = clock64(); __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 0 && threadIdx.x == 0) g_lin_prof[k] += lnp_t1 - lnp_t0; lnp_t0 = lnp_t1; } while (0)
#elif defined(TEB_AMD_STAMP)
// (diagnostic build, -DTEB_AMD_STAMP=<id>: ONE section of the edge loops of the otherwise unchanged product kernel into the spare slot 7 of
// the phase log - no waits, no scheduling fences, lane 1 / lane 0 of wave 0. ids: evaluate {0 static, 1 dynamic, 2 between-pose terms},
// linearise {3, 4, 5 the same}, 10 + the sections of linearize() {0 zero H b, 1 trig, 2 near masks, 3 edges, 4 slice reduction, 5 scatter,
// 6 fixed rows + chi2 sum}, 18 .. 21 graph side data {trig, association, time stamps, via-points}; tools/stamp_sections.py)
#define EVP_DECL long long evp_t0 = clock64();
#define EVP(k) do { const long long t_ = clock64(); if ((k) == (TEB_AMD_STAMP) && threadIdx.x == 1) reinterpret_cast<long long*>(l.red + 40)[7] += t_ - evp_t0; evp_t0 = t_; } while (0)
#define LNP_DECL long long lnp_t0 = clock64();
#define LNP(k) do { const long long t_ = clock64(); if ((k) + 10 == (TEB_AMD_STAMP) && threadIdx.x == 0) reinterpret_cast<long long*>(l.red + 40)[7] += t_ - lnp_t0; lnp_t0 = t_; } while (0)
#else
#define EVP_DECL
#define EVP(k)
#define LNP_DECL
#define LNP(k)
#endif
// (experiment, -DTEB_AMD_POINTS_KEEP_GENERIC: the point-like instantiations branch on SceneDev::fast_points at run time and so keep the
// generic-shape code they never execute - the code round 2's single kernel carried)
#ifdef TEB_AMD_POINTS_KEEP_GENERIC
#define TEB_IF_FAST(F) if ((F) && sc.fast_points)
#define TEB_IF_NOT_FAST(F) if (!((F) && sc.fast_points))
#else
#define TEB_IF_FAST(F) if constexpr (F)
#define TEB_IF_NOT_FAST(F) if constexpr (!(F))
#endif
// Storage formats of the normal matrix (selected per handle by the pose capacity S and the obstacle cache):
//   SOLVER_BAND : Hb[4S][11] lower band in LDS (S <= 343); solved by the hybrid cyclic reduction (cr_solve_hybrid: level 0 from a
//                 band-form copy in HBM into a compact even-row system in LDS) or, teb_amd_options_t::band_ldlt, by the sequential
//                 in-LDS LDL^T of wave 0
//   SOLVER_BANDG: the same band in a per-band HBM buffer (S beyond the LDS band, or when the obstacle cache would not fit beside the LDS band)
//   SOLVER_CR   : block-tridiagonal in 8x8 blocks (two 4-scalar pose groups per block row): D_j (full, symmetric)
//                 and L_j (coupling to block row j-1), solved by block cyclic reduction with all 256 threads
//                 (log2(n/2) levels instead of 4n sequential pivots). Needs ~680*S bytes of LDS: S <= 238.
enum { SOLVER_BAND = 0, SOLVER_CR = 1, SOLVER_BANDG = 2 };   // BANDG: the band lives in HBM (bands too long for the LDS band: up to kThreads * kPoseIterBandHbm poses, LDS permitting)
typedef double teb_v2d __attribute__((ext_vector_type(2)));   // two doubles in one 16-byte access
constexpr int kBlk = 66;   // padded stride (doubles) of one 8x8 block: spreads concurrent eliminations over LDS banks

struct Lds {
  double *sx, *sy, *sth, *sdt, *tdyn, *cs, *sn, *Hb, *bv, *dxv, *red;
  double *Db, *Lb, *fb;   // SOLVER_CR only (Hb aliases Db)
  double *obx, *oby, *obvx, *obvy, *obr;   // obstacle cache (static list first, then the dynamic list)
  int* ired;
};

__host__ __device__ inline int nb_for(int S) { return (4 * S + 7) / 8; }
constexpr size_t kLdsHeadroomBytes = 1024;   // what the library leaves of a workgroup's LDS (teb_amd_create: lds_limit = sharedMemPerBlock - this)
__host__ __device__ inline size_t hmat_doubles(int S, int solver) {  // (LDS-resident part)
  if (solver == SOLVER_BANDG) return (size_t)6 * S + 256;   // the scratch of autoResize (edit script + new poses + split stack + runs); make_lds_plan may add to it
  if (solver == SOLVER_CR) return (size_t)nb_for(S) * (2 * kBlk + 8);
  // hybrid solve: even block rows in LDS (+ up to 14 doubles between its D and L regions, cr_solve_hybrid_impl). Rounded up to an even
  // count: the regions behind it (b, dx) are zeroed and copied in 16-byte accesses and must start on 16-byte boundaries (45 S is odd for odd S).
  const size_t band = (size_t)hbo(4 * S), compact = (size_t)((nb_for(S) + 1) / 2) * (2 * kBlk + 8) + 14;
  return ((band > compact ? band : compact) + 1) & ~(size_t)1;
}
// per-band HBM scratch of the solves: SOLVER_CR keeps a copy of H there; SOLVER_BAND / BANDG the 8x8 blocks (D, L, f) the reduction
// works on (the hybrid solve only reads them)
__host__ __device__ inline size_t hbm_scratch_doubles(int S, int solver) {
  const size_t nb = (size_t)nb_for(S);
  const size_t blocks = nb * (2 * kBlk + 8);   // (>= the band copy of the hybrid solve: nb * 88)
  const size_t own = hmat_doubles(S, solver);
  return own > blocks ? own : blocks;
}
// host: lay out the LDS; ob_entries = obstacles to cache (0 = no cache); lds_limit_bytes = what a workgroup may use on THIS device
// (teb_amd_handle::lds_limit = sharedMemPerBlock - kLdsHeadroomBytes: the band-in-HBM layout sizes its compact region from it, so a part or
// partition mode that reports less LDS gets a smaller region instead of a plan that no longer fits - ADVICE r05). Returns total bytes.
__host__ inline LdsPlan make_lds_plan(int S, int solver, int ob_entries, size_t lds_limit_bytes) {
  LdsPlan p;
  p.S = S; p.solver = solver;
  int o = 0;
  p.off_state = o; o += 7 * S;
  o = (o + 1) & ~1;
  int hm = (int)hmat_doubles(S, solver);
  if (solver == SOLVER_BANDG) {
    // Band in HBM: during a solve this region holds the compact system of the coarse levels of the reduction (cr_solve_t<true>: the fine
    // levels run on blocks in HBM until the surviving rows fit here; their right-hand side lives in the dx region). Round 5: whatever the
    // strips AND the obstacle cache leave of a workgroup's LDS, up to 64 block rows (2 x 66 doubles each) - a band of 338 .. 512 poses then
    // runs two levels through L2 instead of six. Long bands and big obstacle tables take the room first: the capacity limits are unchanged.
    const long long room = (long long)(lds_limit_bytes / 8) - (o + 8LL * S + 16 + 96 + 5LL * (ob_entries > 0 ? ob_entries : 0)), want = 64LL * 2 * kBlk;
    const long long extra = (room < want ? room : want) & ~1LL;
    if (extra > hm) hm = (int)extra;
  }
  p.off_H = o; o += hm;
  p.off_b = o; o += 4 * S + 8;
  p.off_dx = o; o += 4 * S + 8;
  p.off_red = o; o += 64 + 32;
  p.ob_cap = ob_entries;
  if (ob_entries > 0) { p.off_ob = o; o += 5 * ob_entries; } else p.off_ob = -1;
  p.total_bytes = o * (int)sizeof(double);
  return p;
}
__host__ inline size_t lds_bytes_for(int S, int solver, size_t lds_limit_bytes) { return (size_t)make_lds_plan(S, solver, 0, lds_limit_bytes).total_bytes; }

// gH: the band's slice of the HBM normal-matrix buffer (SOLVER_BANDG), else unused
__device__ __forceinline__ Lds carve(double* base, const LdsPlan& p, double* gH = nullptr, bool hb_global = false) {
  Lds l;
  const int S = p.S;
  l.sx = base + p.off_state; l.sy = l.sx + S; l.sth = l.sy + S; l.sdt = l.sth + S; l.tdyn = l.sdt + S;
  l.cs = l.tdyn + S; l.sn = l.cs + S;
  l.Hb = hb_global ? gH : base + p.off_H;
  l.Db = base + p.off_H; l.Lb = l.Db + (size_t)nb_for(S) * kBlk; l.fb = l.Lb + (size_t)nb_for(S) * kBlk;
  l.bv = base + p.off_b;
  l.dxv = base + p.off_dx;
  l.red = base + p.off_red;
  l.ired = reinterpret_cast<int*>(l.red + 64);
  l.obx = base + (p.off_ob >= 0 ? p.off_ob : 0); l.oby = l.obx + p.ob_cap; l.obvx = l.oby + p.ob_cap;
  l.obvy = l.obvx + p.ob_cap; l.obr = l.obvy + p.ob_cap;
  return l;
}

// ---- deterministic block reductions (fixed tree: lanes via shuffles, then the waves in order) ---------
// v[l] += v[l + off], off = 32, 16, .., 1: lane 0 ends with the sum over the wave (the tree of __shfl_down, without its six LDS-crossbar
// round trips: the two wide steps are lane swaps of gfx950, the four narrow ones DPP row shifts on the two halves of the double)
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  {
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
    v += __hiloint2double(hi[1], lo[1]);    // lanes 0..31: v[l + 32]
  }
  {
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
    v += __hiloint2double(hi[1], lo[1]);    // rows 0 and 2: v[l + 16]
  }
  v += dpp_move<0x108>(v);   // row_shl:8
  v += dpp_move<0x104>(v);   // row_shl:4
  v += dpp_move<0x102>(v);   // row_shl:2
  v += dpp_move<0x101>(v);   // row_shl:1
  return v;
}
template <int K>
__device__ __forceinline__ void block_sum(double* v, double* red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < K; ++q) {
    double s = wave_sum(v[q]);
    if (lane == 0) red[q * kWaves + wv] = s;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < K; ++q) {
    double a = red[q * kWaves];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) a += red[q * kWaves + w];
    v[q] = a;
  }
  __syncthreads();
}
__device__ __forceinline__ double block_max(double v, double* red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  if (lane == 0) red[wv] = v;
  __syncthreads();
  double r = red[0];
#pragma unroll
  for (int w = 1; w < kWaves; ++w) r = fmax(r, red[w]);
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
  for (int q = 1; q < kWaves; ++q)
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
  int assoc_cap;          // rows of the association list (= obstacle capacity)
  McuView mcu;            // multi-CU mode (generic scenes): delivered distance records / lists written by other workgroups
};

// radius of a circle about the pose that contains the robot, whatever its heading (bounding-circle culling of the generic shapes)
__device__ __forceinline__ double footprint_bound_radius(const teb_amd_config_t& c) {
  double frad = 0;
  if (c.footprint_type == TEB_AMD_FOOTPRINT_CIRCULAR) frad = c.footprint_radius;
  else if (c.footprint_type == TEB_AMD_FOOTPRINT_TWO_CIRCLES)
    frad = fmax(fabs(c.footprint_front_offset) + c.footprint_front_radius, fabs(c.footprint_rear_offset) + c.footprint_rear_radius);
  else if (c.footprint_type == TEB_AMD_FOOTPRINT_LINE || c.footprint_type == TEB_AMD_FOOTPRINT_POLYGON)
    for (int v = 0; v < c.footprint_n_vertices; ++v) frad = fmax(frad, sqrt(c.footprint_vx[v] * c.footprint_vx[v] + c.footprint_vy[v] * c.footprint_vy[v]));
  return frad;
}
// lower bound of calculateDistance(pose, obstacle) from the two bounding circles, shrunk by a guard band against its own rounding
// (NaN / infinite bounds compare false everywhere: such an obstacle is never culled)
__device__ __forceinline__ double distance_lower_bound(const SceneDev& sc, int oi, double x, double y, double frad) {
  const double bx = sc.cx[oi] - x, by = sc.cy[oi] - y;
  const double lbd = sqrt(bx * bx + by * by) - sc.brad[oi] - frad;
  return lbd - 1e-9 * (fabs(lbd) + sc.brad[oi] + frad + 1.0);
}

// All cost terms whose first vertex is pose i / timediff i (0 <= i <= n-2). JAC=true also accumulates
// J^T Omega J and J^T Omega e into the thread-local window accumulator.
// Edge dispatch of eval_index: MODE 0 = residuals only (computeActiveErrors), 1 = closed-form Jacobians,
// 2 = g2o central differences over the residual code (the two edges the reference linearises analytically -
// EdgeKinematicsDiffDrive edge_kinematics.h:112-149, EdgeTimeOptimal edge_time_optimal.h:93-99 - stay analytic).
// Inside CALL the window is W, the accumulator ACC_ and the Jacobian switch J_.
#define TEB_EDGE(VMASK, CAT, ...)                                                                                   \
  do {                                                                                                                \
    if constexpr (MODE == 2) {                                                                                        \
      numeric_edge<VMASK, CAT>(w, A, [&](const Win& W, RowRec& ACC_) { constexpr bool J_ = false; __VA_ARGS__; });          \
    } else {                                                                                                          \
      const Win& W = w; Accum& ACC_ = A; constexpr bool J_ = (MODE == 1); __VA_ARGS__;                                    \
    }                                                                                                                 \
  } while (0)

// Lanes per pose. A pass over the poses has kThreads lanes; when a LEFTOVER pass (the second pass of a band with more than kThreads
// poses, which would otherwise cost a full pass for a handful of poses) holds at most kThreads / 2 poses, G = 2, 4 or 8 adjacent lanes
// share a pose: slice sl of nsl takes a contiguous chunk of the dynamic-obstacle list (the bulk of the per-pose work), slice 0 also
// everything else of the pose. Partial sums are combined with lane shuffles (fixed tree: deterministic).
__device__ __forceinline__ int lanes_per_pose(int poses_left) {
  int G = 1;
  while (G < 8 && 2 * G * poses_left <= kThreads) G *= 2;
  return G;
}

// Far-field culling of the dynamic-obstacle edges, exact: beyond max(min_obstacle_dist + penalty_epsilon, dynamic_obstacle_inflation_dist)
// both residuals of EdgeDynamicObstacle and their Jacobians are exactly zero (penalties.h:75-87), so an obstacle farther than that from
// the pose (+ a relative guard band of 1e-12 against the rounding of the distance; + 1e-6 m in the numeric mode, whose residuals are
// evaluated 1e-9 away) adds nothing but zeros. dyn_near_mask is pass 1: the squared distance of the obstacles [kb, ke) (at most 64) of
// the dynamic list at the pose's time stamp, 6 operations each in independent chains, one bit per obstacle. It runs BEFORE the window
// accumulator of the pose is live, so that its unrolled chains do not compete with it for registers. Pass 2 (eval_index): each lane
// walks the set bits of its own mask in list order and evaluates only those edges (fp64 sqrt, divisions, penalties: > 100 operations
// each) - same operations, same order, same bits as the full loop, which spent > 95 % of its time adding zeros.
__device__ __forceinline__ double dyn_far_distance(const teb_amd_config_t& c) {
  return fmax(c.min_obstacle_dist + c.penalty_epsilon, c.dynamic_obstacle_inflation_dist) +
         (c.footprint_type == TEB_AMD_FOOTPRINT_CIRCULAR ? c.footprint_radius : 0.0);
}
template <int MODE>
__device__ __forceinline__ unsigned long long dyn_near_mask(const teb_amd_config_t& c, const SceneDev& sc, const Lds& l, int i, int kb, int ke,
                                                            double margin = 0.0) {
  const double far_d = dyn_far_distance(c) + margin;
  const double x = l.sx[i], y = l.sy[i], ti = l.tdyn[i];
  unsigned long long near = 0;
#pragma unroll 4   // independent chains: at one wave per SIMD only instruction-level parallelism hides the fp64 latency
  for (int k = kb; k < ke; ++k) {
    const int p = sc.n_static + k;
    // pos_ + t*centroid_velocity_ (obstacles.h:382-385)
    const double ddx = x - (l.obx[p] + ti * l.obvx[p]), ddy = y - (l.oby[p] + ti * l.obvy[p]);
    const double d2 = ddx * ddx + ddy * ddy;
    const double thr = (far_d + l.obr[p]) * (1.0 + 1e-12) + (MODE == 2 ? 1e-6 : 0.0);
    if (!(d2 >= thr * thr) || thr <= 0) near |= 1ull << (k - kb);   // non-finite distances count as near
  }
  return near;
}
// the chunk of the dynamic-obstacle list of slice sl of nsl (multiples of 4)
__device__ __forceinline__ void dyn_chunk(const SceneDev& sc, int sl, int nsl, int& d_lo, int& d_hi) {
  const int dchunk = nsl > 1 ? (((sc.n_dyn + nsl - 1) / nsl + 3) & ~3) : sc.n_dyn;
  d_lo = sl * dchunk < sc.n_dyn ? sl * dchunk : sc.n_dyn;
  d_hi = d_lo + dchunk < sc.n_dyn ? d_lo + dchunk : sc.n_dyn;
}
// The masks are reused across the linearisations and error evaluations of one outer iteration. The time stamps of the dynamic edges
// are frozen when the graph is built (src/optimal_planner.cpp:662-670), so the obstacle positions a pose sees do not change during
// optimize(); only the pose moves. A mask taken at the reference position r with the threshold widened by m holds every obstacle
// that is near at any position p with |p - r| <= m (triangle inequality), and a superset is all pass 2 needs: the edges it
// evaluates beyond the true threshold contribute exact zeros, like in the full loop. So each lane keeps (mask, r) per pose it serves
// (two of them, in registers: the passes beyond the second of a band-in-HBM kernel recompute their masks every time) and recomputes only
// when its pose has left the disc - or when the graph was rebuilt (r = NaN).
// m = kNearMarginFactor x the culling distance: the wider the disc the rarer the recomputation and the more zero edges in pass 2
// (headline kernel 4.21 ms without the cache; 4.04 / 3.99 / 3.96 / 3.99 ms at factor 0.5 / 1 / 2 / 3).
constexpr double kNearMarginFactor = 1.5;   // margin of the cached near masks in units of the culling distance (1, 2, 3 measured: HISTORY.md section 3)
struct NearCache {
  unsigned long long m0, m1;
  double rx0, ry0, rx1, ry1;
  bool off;   // teb_amd_options_t::no_near_cache: exact mask (no margin) at every pass
  __device__ __forceinline__ void invalidate() { rx0 = ry0 = rx1 = ry1 = __builtin_nan(""); m0 = m1 = 0; }
};
template <int MODE, bool FAST>
__device__ __forceinline__ unsigned long long dyn_near_cached(const teb_amd_config_t& c, const SceneDev& sc, const Lds& l, int i, int sl, int nsl,
                                                              NearCache& nc, int pass) {
  if (!(FAST && i >= 1 && c.include_dynamic_obstacles && c.weight_obstacle != 0)) return 0;
  const bool uncached = nc.off || (kMaxPoseIter > 2 && pass >= 2);   // two slots per lane: the third and fourth pass of a long band take the exact mask
  const double m = uncached ? 0.0 : kNearMarginFactor * dyn_far_distance(c);
  const double x = l.sx[i], y = l.sy[i];
  const double rx = pass == 0 ? nc.rx0 : nc.rx1, ry = pass == 0 ? nc.ry0 : nc.ry1;
  unsigned long long mask = pass == 0 ? nc.m0 : nc.m1;
  const double ddx = x - rx, ddy = y - ry;
  const double lim = fmax(m * (1.0 - 1e-6) - 2e-6, 0.0);   // (the numeric mode evaluates residuals 1e-9 away from the pose and culls 1e-6 wider)
  // left the disc, or no mask yet (NaN reference). When one lane of the wave has to recompute, the whole wave walks the loop anyway: every
  // lane then refreshes its mask at its current position (a fresh disc costs the others nothing and postpones their next recomputation)
  if (uncached || __any(!(ddx * ddx + ddy * ddy <= lim * lim))) {
    int d_lo, d_hi;
    dyn_chunk(sc, sl, nsl, d_lo, d_hi);
    mask = dyn_near_mask<MODE>(c, sc, l, i, d_lo, d_lo + 64 < d_hi ? d_lo + 64 : d_hi, m);
    if (pass == 0) { nc.m0 = mask; nc.rx0 = x; nc.ry0 = y; } else if (!(kMaxPoseIter > 2 && pass >= 2)) { nc.m1 = mask; nc.rx1 = x; nc.ry1 = y; }
#ifdef TEB_PROFILE
    atomicAdd(&g_near_recomputed, 1ull);
#endif
  }
#ifdef TEB_PROFILE
  atomicAdd(&g_near_queries, 1ull);
#endif
  return mask;
}

// PART (multi-CU mode only): 0 = every cost term of the pose, 1 = the static / dynamic obstacle edges alone, 2 = everything else. The four
// chi^2 categories are separate accumulators, so an error evaluation may run part 2 while the helpers still compute the distances part 1
// needs; a linearisation (one window accumulator, fixed edge order) always runs part 0.
template <int MODE, bool FAST, int PART = 0>
__device__ __forceinline__ void eval_index(const teb_amd_config_t& c, const SceneDev& sc, const TebCtx& t,
                                           const Lds& l, int i, Accum& A, unsigned long long near_first, int sl = 0, int nsl = 1) {
  constexpr bool JAC = (MODE == 1);
  const int n = t.n;
  const bool first = (sl == 0);
  int d_lo, d_hi;
  dyn_chunk(sc, sl, nsl, d_lo, d_hi);
  Win w;
  w.x0 = l.sx[i]; w.y0 = l.sy[i]; w.t0 = l.sth[i]; w.d0 = l.sdt[i];
  w.x1 = l.sx[i + 1]; w.y1 = l.sy[i + 1]; w.t1 = l.sth[i + 1];
  w.c0 = l.cs[i]; w.s0 = l.sn[i]; w.c1 = l.cs[i + 1]; w.s1 = l.sn[i + 1];
  const bool has2 = (i + 2 <= n - 1);
  w.d1 = has2 ? l.sdt[i + 1] : 1.0;
  w.x2 = has2 ? l.sx[i + 2] : 0.0; w.y2 = has2 ? l.sy[i + 2] : 0.0; w.t2 = has2 ? l.sth[i + 2] : 0.0;
  const bool seg_active = (n > 2);   // g2o never activates an edge whose vertices are all fixed (n == 2)

  // ---- unary edges of pose i (AddEdgesObstacles :444-548, AddEdgesDynamicObstacles :646-673, AddEdgesViaPoints :675-718)
  // association entries are POSITIONS in the static list (sc.static_idx / the LDS obstacle cache)
  // the static edges of the pose are dealt round-robin to its slices (k = sl, sl + nsl, ..)
  int pre[4] = {0, 0, 0, 0};   // first batch of list entries of the point-like fast path, fetched beside the count (rows sl, sl + nsl, ..: < capacity)
  if constexpr (FAST && MODE != 2) {
    if (i >= 1 && TEB_CFGI(NEW_ASSOCIATION)) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kq = sl + u * nsl;
        pre[u] = kq < t.assoc_cap ? t.assoc[(size_t)kq * t.stride + i] : 0;
      }
    }
  }
  const int cnt = FAST ? t.assoc_cnt[i] : ld_list(t.mcu.shared_lists, t.assoc_cnt + i);
  EVP_DECL
  if (i >= 1) {
   if constexpr (PART != 2) {
    TEB_IF_FAST(FAST) {
      if (TEB_CFGI(NEW_ASSOCIATION)) {
        // The association lists live in HBM (pose-major, coalesced over the lanes); a lane walks its list in order and at one wave per
        // SIMD every dependent load is a full round trip (L2 for the entry, LDS for the obstacle, then the sqrt chain: ~ 1.1 k cycles per
        // edge, measured). Four entries are fetched together and their distances computed side by side (independent chains); the
        // residual rows then follow in list order - same operations per edge, same order of accumulation.
        constexpr int kStaticBatch = 4;
        if constexpr (MODE == 2) {   // central differences: 13 residual evaluations per edge, the entry load is noise
          for (int k = sl; k < cnt; k += nsl) {
            const int p = t.assoc[(size_t)k * t.stride + i];
            const double ox = l.obx[p], oy = l.oby[p], orad = l.obr[p];
            TEB_EDGE(M_POSE0, CAT_OBST, edge_obstacle_fast<J_>(c, ox, oy, orad, W, t.w_obst, t.inflated, ACC_));
          }
        } else {
          // the first batch does not wait for the count: rows 0 .. of the list exist whatever it is (capacity = every obstacle), so
          // its loads are issued together with the load of cnt - one L2 round trip instead of two in front of every edge loop; every
          // further batch is fetched while the one before it is evaluated (round 5: each used to expose its own round trip)
          int cur[kStaticBatch];
#pragma unroll
          for (int u = 0; u < kStaticBatch; ++u) cur[u] = pre[u];
          for (int k0 = sl; k0 < cnt; k0 += kStaticBatch * nsl) {
            int pp[kStaticBatch];
            bool valid[kStaticBatch];
#pragma unroll
            for (int u = 0; u < kStaticBatch; ++u) {
              const int kq = k0 + u * nsl;
              valid[u] = kq < cnt;
              pp[u] = valid[u] ? cur[u] : 0;   // (rows beyond the count hold whatever the buffer held: never an index)
              const int kn = kq + kStaticBatch * nsl;
              cur[u] = kn < cnt ? t.assoc[(size_t)kn * t.stride + i] : 0;   // the next batch: in flight during this one
            }
            double dist[kStaticBatch], g0[kStaticBatch], g1[kStaticBatch];
#pragma unroll
            for (int u = 0; u < kStaticBatch; ++u) {
              double gr[2];
              dist[u] = pointlike_distance<JAC>(c, w.x0, w.y0, l.obx[pp[u]], l.oby[pp[u]], TEB_CFGI(RADIUS_FREE) ? 0.0 : l.obr[pp[u]], gr);   // (no radii in the static list: an exact zero, not read)
              g0[u] = JAC ? gr[0] : 0.0; g1[u] = JAC ? gr[1] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < kStaticBatch; ++u)
              if (valid[u]) {
                const double gr[2] = {g0[u], g1[u]};
                obstacle_rows<JAC>(c, dist[u], gr, t.w_obst, t.inflated, A);
              }
          }
        }
      } else {   // legacy lists carry the triple edge at the closest pose as one flagged entry
        for (int k = sl; k < cnt; k += nsl) {
          const int ent = t.assoc[(size_t)k * t.stride + i];
          const int p = ent & kAssocMask;
          const double ox = l.obx[p], oy = l.oby[p], orad = l.obr[p];
#pragma unroll 1
          for (int rep = (ent & kAssocTriple) ? 3 : 1; rep > 0; --rep)
            TEB_EDGE(M_POSE0, CAT_OBST, edge_obstacle_fast<J_>(c, ox, oy, orad, W, t.w_obst, t.inflated, ACC_));
        }
      }
      EVP(MODE == 0 ? 0 : 3);
      if (TEB_CFGI(DYNAMIC_EDGES)) {   // (profile: weight_obstacle != 0; without include_dynamic_obstacles the dynamic list is empty)
        const double ti = l.tdyn[i];
        const double far_d = dyn_far_distance(c);
        // far-field culling (dyn_near_mask above): the mask of the first 64 obstacles of the slice comes from the caller (dyn_near_cached,
        // before the accumulator went live; a superset of the near obstacles); further blocks (more than 64 dynamic obstacles per slice) are computed here
        for (int kb = d_lo; kb < d_hi; kb += 64) {
          const int ke = kb + 64 < d_hi ? kb + 64 : d_hi;
          unsigned long long near = (kb == d_lo) ? near_first : dyn_near_mask<MODE>(c, sc, l, i, kb, ke);
          while (near) {
            const int k = kb + __ffsll((long long)near) - 1;
            near &= near - 1;
            const int p = sc.n_static + k;
            const double ox = l.obx[p] + ti * l.obvx[p], oy = l.oby[p] + ti * l.obvy[p], orad = l.obr[p];
            // the cached mask is a superset (threshold widened by the margin of the cache): the exact test of dyn_near_mask, 8 operations,
            // drops the obstacles whose edge would only add zeros before the > 100 operations (x 13 with central differences) are spent
            const double fdx = w.x0 - ox, fdy = w.y0 - oy;
            const double fthr = (far_d + orad) * (1.0 + 1e-12) + (MODE == 2 ? 1e-6 : 0.0);
            if (fdx * fdx + fdy * fdy >= fthr * fthr && fthr > 0) continue;
            TEB_EDGE(M_POSE0, CAT_OBST, edge_dynamic_obstacle_fast<J_>(c, ox, oy, orad, W, ACC_));
          }
        }
      }
    } else {
      bool replayed = false;
      if constexpr (MODE != 2) {
        if (t.mcu.items != nullptr) {
          // multi-CU mode: record k of the pose = distance and gradient of list entry k, records cnt .. cnt + n_dyn - 1 = the dynamic
          // obstacles; the rows follow in the order of the loops below, from the numbers those loops would compute
          replayed = true;
          const double* it = t.mcu.items + i;
          for (int k = sl; k < cnt; k += nsl) {
            const int ent = ld_list(true, &t.assoc[(size_t)k * t.stride + i]);
            const double dist = ld_agent_f64(it + (size_t)(4 * k) * t.stride);
            double gr[3] = {0, 0, 0};
            if (JAC) { gr[0] = ld_agent_f64(it + (size_t)(4 * k + 1) * t.stride); gr[1] = ld_agent_f64(it + (size_t)(4 * k + 2) * t.stride);
                       gr[2] = ld_agent_f64(it + (size_t)(4 * k + 3) * t.stride); }
#pragma unroll 1
            for (int rep = (ent & kAssocTriple) ? 3 : 1; rep > 0; --rep) obstacle_rows_g<JAC>(c, dist, gr, t.w_obst, t.inflated, A);
          }
          if (c.include_dynamic_obstacles && c.weight_obstacle != 0) {
            for (int k = d_lo; k < d_hi; ++k) {
              const int q = cnt + k;
              const double dist = ld_agent_f64(it + (size_t)(4 * q) * t.stride);
              double gr[3] = {0, 0, 0};
              if (JAC) { gr[0] = ld_agent_f64(it + (size_t)(4 * q + 1) * t.stride); gr[1] = ld_agent_f64(it + (size_t)(4 * q + 2) * t.stride);
                         gr[2] = ld_agent_f64(it + (size_t)(4 * q + 3) * t.stride); }
              dynamic_obstacle_rows_g<JAC>(c, dist, gr, A);
            }
          }
        }
      }
      if (!replayed) {
        for (int k = sl; k < cnt; k += nsl) {
          const int ent = ld_list(t.mcu.shared_lists, &t.assoc[(size_t)k * t.stride + i]);
          const int oi = sc.static_idx[ent & kAssocMask];
#pragma unroll 1
          for (int rep = (ent & kAssocTriple) ? 3 : 1; rep > 0; --rep)
            TEB_EDGE(M_POSE0, CAT_OBST, edge_obstacle<J_>(c, sc, oi, W, t.w_obst, t.inflated, ACC_));
        }
        if (c.include_dynamic_obstacles && c.weight_obstacle != 0) {
          const double ti = l.tdyn[i];
          for (int k = d_lo; k < d_hi; ++k) {
            const int oi = sc.dyn_idx[k];
            TEB_EDGE(M_POSE0, CAT_OBST, edge_dynamic_obstacle<J_>(c, sc, oi, W, ti, ACC_));
          }
        }
      }
    }
   }   // PART != 2
   if constexpr (PART != 1)
    if (TEB_CFGI(VIA_POINTS) && first && t.via_en) {   // (profile: no via-points)
      for (int v = 0; v < sc.nvia; ++v)
        if (t.via_pose[v] == i) {
          const double vx = sc.viax[v], vy = sc.viay[v];
          TEB_EDGE(M_POSE0, CAT_VIA, edge_via_point<J_>(c, vx, vy, W, ACC_));
        }
    }
  }
  EVP(MODE == 0 ? 1 : 4);
  if constexpr (PART == 1) return;
  if (!first) return;   // the other slices only share the dynamic-obstacle edges
  // ---- AddEdgesVelocity :720-769
  if (TEB_CFGI(NONHOLONOMIC_VELOCITY)) {
    if (TEB_CFGI(VELOCITY_EDGES)) TEB_EDGE(M_SEG, CAT_OTHER, edge_velocity<J_>(c, W, ACC_));
  } else {
    if (!(c.weight_max_vel_x == 0 && c.weight_max_vel_y == 0 && c.weight_max_vel_theta == 0))
      TEB_EDGE(M_SEG, CAT_OTHER, edge_velocity_holonomic<J_>(c, W, ACC_));
  }
  // ---- AddEdgesAcceleration :771-873
  if (TEB_CFGI(ACCELERATION_EDGES)) {
    const bool nonholo = TEB_CFGI(NONHOLONOMIC_ACCELERATION);
    if (nonholo) {
      if (i == 0 && t.has_vs) TEB_EDGE(M_SEG, CAT_OTHER, (edge_acceleration_se<J_, true>(c, W, t.vs[0], t.vs[2], ACC_)));
      if (has2) TEB_EDGE(M_ALL, CAT_OTHER, edge_acceleration<J_>(c, W, ACC_));
      if (i == n - 2 && t.has_vg) TEB_EDGE(M_SEG, CAT_OTHER, (edge_acceleration_se<J_, false>(c, W, t.vg[0], t.vg[2], ACC_)));
    } else {
      if (i == 0 && t.has_vs) TEB_EDGE(M_SEG, CAT_OTHER, (edge_acceleration_holonomic_se<J_, true>(c, W, t.vs, ACC_)));
      if (has2) TEB_EDGE(M_ALL, CAT_OTHER, edge_acceleration_holonomic<J_>(c, W, ACC_));
      if (i == n - 2 && t.has_vg) TEB_EDGE(M_SEG, CAT_OTHER, (edge_acceleration_holonomic_se<J_, false>(c, W, t.vg, ACC_)));
    }
  }
  // ---- AddEdgesTimeOptimal :877-893 (analytic in the reference), AddEdgesShortestPath :895-912
  if (TEB_CFGI(TIME_OPTIMAL)) edge_time_optimal<MODE != 0>(c, w, A);
  if (TEB_CFGI(SHORTEST_PATH) && seg_active) TEB_EDGE(M_POSE0 | M_POSE1, CAT_OTHER, edge_shortest_path<J_>(c, W, ACC_));
  // ---- kinematics :355-358, 916-958 (diff-drive: analytic in the reference)
  if (seg_active) {
    if (TEB_CFGI(KIN_DIFF_DRIVE)) {
      if (TEB_CFGI(KIN_EDGES)) edge_kinematics_diffdrive<MODE != 0>(c, w, A);
    } else {
      if (!(c.weight_kinematics_nh == 0 && c.weight_kinematics_turning_radius == 0))
        TEB_EDGE(M_POSE0 | M_POSE1, CAT_OTHER, edge_kinematics_carlike<J_>(c, W, ACC_));
    }
  }
  // ---- AddEdgesPreferRotDir :961-997
  if (i < 3 && seg_active && c.weight_prefer_rotdir != 0 && (t.rotdir == TEB_AMD_ROT_LEFT || t.rotdir == TEB_AMD_ROT_RIGHT)) {
    const double dir = t.rotdir == TEB_AMD_ROT_LEFT ? 1.0 : -1.0;
    TEB_EDGE(M_POSE0 | M_POSE1, CAT_OTHER, edge_prefer_rotdir<J_>(c, W, dir, ACC_));
  }
  // ---- AddEdgesVelocityObstacleRatio :999-1021
  if (TEB_CFGI(VELOCITY_OBSTACLE_RATIO) && !c.legacy_obstacle_association) {   // obstacles_per_vertex_ stays empty in legacy mode
    for (int k = 0; k < cnt; ++k) {
      const int p = FAST ? t.assoc[(size_t)k * t.stride + i] : ld_list(t.mcu.shared_lists, &t.assoc[(size_t)k * t.stride + i]);
      bool replayed = false;
      if constexpr (!FAST && MODE != 2) {
        if (t.mcu.items != nullptr) {   // (multi-CU mode) calculateDistance(pose i, obstacle) is record k of the pose
          replayed = true;
          const double* it = t.mcu.items + i;
          const double dobs = ld_agent_f64(it + (size_t)(4 * k) * t.stride);
          double gr[3] = {0, 0, 0};
          if (JAC) { gr[0] = ld_agent_f64(it + (size_t)(4 * k + 1) * t.stride); gr[1] = ld_agent_f64(it + (size_t)(4 * k + 2) * t.stride);
                     gr[2] = ld_agent_f64(it + (size_t)(4 * k + 3) * t.stride); }
          edge_velocity_obstacle_ratio<JAC>(c, dobs, gr, w, A);
        }
      }
      if (!replayed)
        TEB_EDGE(M_SEG, CAT_OTHER, {
          double gr[3] = {0, 0, 0};
          double dobs;
          TEB_IF_FAST(FAST) dobs = pointlike_distance<J_>(c, W.x0, W.y0, l.obx[p], l.oby[p], l.obr[p], gr);
          else dobs = footprint_distance(c, sc, sc.static_idx[p & kAssocMask], W.x0, W.y0, W.c0, W.s0, false, 0.0, J_ ? gr : nullptr);
          edge_velocity_obstacle_ratio<J_>(c, dobs, gr, W, ACC_);
        });
    }
  }
  EVP(MODE == 0 ? 2 : 5);
}
#undef TEB_EDGE

// scatter the thread-local window into the LDS normal matrix; rows/cols of fixed variables are dropped. Step K (0, 1, 2) adds the rows of
// pose i + K (window rows 4K .. 4K+3): in one step every lane writes the rows of a different pose, so the three steps (a barrier between
// them) are free of conflicts with ALL lanes busy in each, and every entry receives its up to three contributions in the fixed order
// lane p, p-1, p-2 - deterministic, no atomics. In the block layout only the lower triangle of a diagonal block is kept (the
// factorisation reads nothing else).
// One contribution to an entry of the LDS normal matrix / right-hand side. The three scatter steps leave exactly one writer per entry
// and step, so this is no race either way; as an LDS atomic (ds_add_f64: one IEEE addition in the LDS unit, no return value) it is ONE
// instruction per entry instead of a read, an add and a write with the round trip in between: linearize() - 8 %, headline - 1.6 %,
// C2 - 3 % (round 5; bit-identical). Where the old value is already in a register the plain store wins: the same change in the rounds
// of the cyclic reduction, whose reads are prefetched under the elimination, cost 2 % (profiles/ab_edge_loops_r05.txt). (HBM band of
// long bands: plain update.)
template <int SOLVER>
__device__ __forceinline__ void hmat_add(double* p, double v) {
  if constexpr (SOLVER == SOLVER_BANDG) *p += v;   // (global_atomic_add_f64 + an acquire fence measured 30 % slower on 400 .. 944-pose bands)
  else (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int SOLVER, int K>
__device__ __forceinline__ void scatter(const Accum& A, const Lds& l, int i, int n) {
  const int base = 4 * i;
  const int last_pose = 4 * (n - 1);
#pragma unroll
  for (int a = 4 * K; a < (4 * K + 4 < 11 ? 4 * K + 4 : 11); ++a) {
    int ra = base + a;
    bool fa = (ra < 3) || (ra >= last_pose);
    if (fa) continue;
    hmat_add<SOLVER_BAND>(&l.bv[ra], -A.g[a]);   // (b lives in LDS in every layout; x - g == x + (-g))
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      int rb = base + b;
      bool fb = (rb < 3) || (rb >= last_pose);
      if (fb) continue;
      const double v = A.H[a * (a + 1) / 2 + b];
      if (SOLVER != SOLVER_CR) {
        hmat_add<SOLVER>(&l.Hb[hbo(ra) + (a - b)], v);
      } else {
        const int jr = ra >> 3, jc = rb >> 3;   // window spans at most two consecutive block rows
        if (jr == jc) hmat_add<SOLVER>(&l.Db[jr * kBlk + (ra & 7) * 8 + (rb & 7)], v);
        else hmat_add<SOLVER>(&l.Lb[jr * kBlk + (ra & 7) * 8 + (rb & 7)], v);
      }
    }
  }
}

// linear view of the live part of the normal matrix (band: Hb[0, Nt*11); blocks: Db[0, Nb*66) then Lb[0, Nb*66))
template <int SOLVER>
__device__ __forceinline__ double* hmat_ptr(const Lds& l, int q, int Nt) {
  if (SOLVER != SOLVER_CR) return l.Hb + q;
  const int half = ((Nt + 7) >> 3) * kBlk;
  return q < half ? l.Db + q : l.Lb + (q - half);
}

// the live part of the normal matrix <-> its backup in the band's HBM scratch, two doubles per access (both sides start on 16-byte
// boundaries, the block / band regions have even lengths: a pair never straddles the D | L boundary)
template <int SOLVER>
__device__ __forceinline__ void hmat_save(const Lds& l, int hsz, int Nt, double* __restrict__ g) {
  typedef double __attribute__((address_space(1))) gdouble_t;
  typedef teb_v2d __attribute__((address_space(1))) gv2d_t;
  for (int q = 2 * (int)threadIdx.x; q < hsz; q += 2 * kThreads)
    *reinterpret_cast<gv2d_t*>((gdouble_t*)g + q) = *reinterpret_cast<const teb_v2d*>(hmat_ptr<SOLVER>(l, q, Nt));
}
template <int SOLVER>
__device__ __forceinline__ void hmat_load(const Lds& l, int hsz, int Nt, const double* __restrict__ g) {
  typedef const double __attribute__((address_space(1))) gdouble_t;
  typedef const teb_v2d __attribute__((address_space(1))) gv2d_t;
  for (int q = 2 * (int)threadIdx.x; q < hsz; q += 2 * kThreads)
    *reinterpret_cast<teb_v2d*>(hmat_ptr<SOLVER>(l, q, Nt)) = *reinterpret_cast<gv2d_t*>((gdouble_t*)g + q);
}

// address of the diagonal entry of variable r
template <int SOLVER>
__device__ __forceinline__ double* diag_ptr(const Lds& l, int r) {
  return SOLVER != SOLVER_CR ? &l.Hb[hbo(r)] : &l.Db[(r >> 3) * kBlk + (r & 7) * 9];
}

// per-pose cos/sin cache: every cost term that needs the heading reads these instead of re-evaluating libm
__device__ __forceinline__ void refresh_trig(const Lds& l, int n) {
  for (int i = threadIdx.x; i < n; i += kThreads) { double sv, cv; sincos(l.sth[i], &sv, &cv); l.cs[i] = cv; l.sn[i] = sv; }   // one argument reduction for both
}

// buildSystem: H = sum J^T Omega J, b = -sum J^T Omega e, and chi^2 per category at the current state.
template <int SOLVER, int JMODE, bool FAST>
__device__ inline void linearize(const teb_amd_config_t& c, const SceneDev& sc, const TebCtx& t, const Lds& l, NearCache& nc,
                                 double* cats /*4, out on all threads*/, bool trig_is_current = false) {
  const int n = t.n, Nt = 4 * n, tid = threadIdx.x;
  const int hsz = (SOLVER != SOLVER_CR) ? hbo(Nt) : ((Nt + 7) >> 3) * 2 * kBlk;
  LNP_DECL
  // 16 bytes per store (the band, both block regions and b start on 16-byte boundaries; what a pair writes beyond an odd end is padding or
  // the next, not yet live row)
  if constexpr (SOLVER != SOLVER_CR) {
    for (int q = 2 * tid; q < hsz; q += 2 * kThreads) *reinterpret_cast<teb_v2d*>(l.Hb + q) = teb_v2d{0.0, 0.0};
  } else {
    const int half = ((Nt + 7) >> 3) * kBlk;   // (even)
    for (int q = 2 * tid; q < half; q += 2 * kThreads) {
      *reinterpret_cast<teb_v2d*>(l.Db + q) = teb_v2d{0.0, 0.0};
      *reinterpret_cast<teb_v2d*>(l.Lb + q) = teb_v2d{0.0, 0.0};
    }
  }
  for (int q = 2 * tid; q < Nt + 8; q += 2 * kThreads) *reinterpret_cast<teb_v2d*>(l.bv + q) = teb_v2d{0.0, 0.0};
  LNP(0);
  if (!trig_is_current) refresh_trig(l, n);   // (the LM loop keeps the cos / sin cache current itself: see the update step)
  __syncthreads();
  LNP(1);
  Accum A;
  A.clear_chi();
  for (int k0 = 0, pass = 0; k0 < n - 1; ++pass) {
    const int G = (k0 > 0 || kThreads > 256) ? lanes_per_pose(n - 1 - k0) : 1;   // slices only for a leftover pass: bands up to kThreads poses keep their summation order
    const int i = k0 + tid / G, sl = tid % G;
    const bool active = i <= n - 2;
    constexpr int EM = JMODE == TEB_AMD_JACOBIAN_G2O_NUMERIC ? 2 : 1;
    const unsigned long long near = active ? dyn_near_cached<EM, FAST>(c, sc, l, i, sl, G, nc, pass) : 0ull;   // before the accumulator is live
    LNP(2);
    A.clear();
    if (active) eval_index<EM, FAST>(c, sc, t, l, i, A, near, sl, G);
    LNP(3);
    if (G > 1) {   // the slices of a pose hold partial sums of its dynamic-obstacle rows: pose block (x, y, theta) of H and g
      for (int off = 1; off < G; off <<= 1) {
#pragma unroll
        for (int q = 0; q < 6; ++q) A.H[q] += __shfl_xor(A.H[q], off, 64);
#pragma unroll
        for (int q = 0; q < 3; ++q) A.g[q] += __shfl_xor(A.g[q], off, 64);
      }
    }
    LNP(4);
    if (active && sl == 0) scatter<SOLVER, 0>(A, l, i, n);
    __syncthreads();
    if (active && sl == 0) scatter<SOLVER, 1>(A, l, i, n);
    __syncthreads();
    if (active && sl == 0) scatter<SOLVER, 2>(A, l, i, n);
    __syncthreads();
    LNP(5);
    k0 += kThreads / G;
  }
  // fixed variables (pose 0, pose n-1, the non-existing dt_{n-1}) become identity rows
  if (tid < 3) { *diag_ptr<SOLVER>(l, tid) = 1.0; l.bv[tid] = 0; }
  if (tid >= 4 && tid < 8) { int r = 4 * (n - 1) + (tid - 4); *diag_ptr<SOLVER>(l, r) = 1.0; l.bv[r] = 0; }
  if (SOLVER == SOLVER_CR && tid >= 8 && tid < 12 && (n & 1)) {   // pad the last block row of an odd pose count
    int r = 4 * n + (tid - 8); *diag_ptr<SOLVER>(l, r) = 1.0;
  }
  cats[0] = A.chi[0]; cats[1] = A.chi[1]; cats[2] = A.chi[2]; cats[3] = A.chi[3];
  block_sum<4>(cats, l.red);
  LNP(6);
}

// computeActiveErrors + activeRobustChi2 at the current state. cats[4] rides along: a fifth per-lane value summed over the workgroup by the
// same reduction (the computeScale term of the LM step; one reduction and one pair of barriers less per trial)
template <bool FAST, int PART>
__device__ __forceinline__ void evaluate_pass(const teb_amd_config_t& c, const SceneDev& sc, const TebCtx& t, const Lds& l, NearCache& nc, Accum& A) {
  for (int k0 = 0, pass = 0; k0 < t.n - 1; ++pass) {
    const int G = (k0 > 0 || kThreads > 256) ? lanes_per_pose(t.n - 1 - k0) : 1;
    const int i = k0 + (int)threadIdx.x / G;
    if (i <= t.n - 2) {
      const unsigned long long near = dyn_near_cached<0, FAST>(c, sc, l, i, (int)threadIdx.x % G, G, nc, pass);
      eval_index<0, FAST, PART>(c, sc, t, l, i, A, near, (int)threadIdx.x % G, G);
    }
    k0 += kThreads / G;
  }
}
template <bool FAST>
__device__ inline void evaluate(const teb_amd_config_t& c, const SceneDev& sc, const TebCtx& t, const Lds& l, NearCache& nc,
                                double* cats /*5*/, bool trig_is_current = false) {
  if (!trig_is_current) {   // (callers that changed the headings and refreshed cos / sin themselves, behind a barrier, skip this)
    refresh_trig(l, t.n);
    __syncthreads();
  }
  Accum A;   // only chi[] is live when JAC == false
  A.clear_chi();
  evaluate_pass<FAST, 0>(c, sc, t, l, nc, A);
  cats[0] = A.chi[0]; cats[1] = A.chi[1]; cats[2] = A.chi[2]; cats[3] = A.chi[3];
  block_sum<5>(cats, l.red);
}
// The same in the multi-CU mode (generic scenes): the helpers compute the obstacle distances of the new state while this workgroup
// evaluates every other cost term; the obstacle rows follow from the delivered records. The chi^2 categories are separate sums, each in
// its single-CU order: same bits. Returns false when the helpers did not deliver (the band is flagged and given up).
__device__ inline bool evaluate_mcu(const teb_amd_config_t& c, const SceneDev& sc, const TebCtx& t, const Lds& l, NearCache& nc, McuMaster& m, int S,
                                    double* cats /*5*/) {
  refresh_trig(l, t.n);
  __syncthreads();
  mcu_publish(m, l.sx, l.sy, l.cs, l.sn, l.tdyn, t.n, S);
  mcu_issue(m, MCU_KIND_DIST, t.n);
  Accum A;
  A.clear_chi();
  const bool overlap = c.weight_velocity_obstacle_ratio == 0;   // (those edges read the records as well)
  if (overlap) evaluate_pass<false, 2>(c, sc, t, l, nc, A);
  const bool ok = mcu_wait(m, l.ired + 26);
  if (ok) {
    if (overlap) evaluate_pass<false, 1>(c, sc, t, l, nc, A);
    else evaluate_pass<false, 0>(c, sc, t, l, nc, A);
  }
  cats[0] = A.chi[0]; cats[1] = A.chi[1]; cats[2] = A.chi[2]; cats[3] = A.chi[3];
  block_sum<5>(cats, l.red);
  return ok;
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
    const double d = H[hbo(k)] + lambda;
    if (!(d > 0)) { ok = false; break; }
    const double inv = 1.0 / d;
    const double xk = x[k];
    if (k + wi < Nt) {
      const double ci = H[hbo(k + wi) + wi];
      if (wj > 0) {
        const double cj = H[hbo(k + wj) + wj];
        H[hbo(k + wi) + (wi - wj)] -= (ci * inv) * cj;
      } else {
        x[k + wi] -= (ci * inv) * xk;
      }
    }
    if (lane == 0 && k + 10 < Nt) x[k + 10] -= (H[hbo(k + 10) + 10] * inv) * xk;
    __builtin_amdgcn_wave_barrier();
    if (lane < 10 && k + lane + 1 < Nt) H[hbo(k + lane + 1) + lane + 1] *= inv;   // L(k+i, k)
    if (lane == 0) H[hbo(k)] = d;
    __builtin_amdgcn_wave_barrier();
  }
  if (!ok) return false;
  for (int r = lane; r < Nt; r += 64) x[r] = x[r] / H[hbo(r)];
  __builtin_amdgcn_wave_barrier();
  for (int k = Nt - 1; k > 0; --k) {
    const double xk = x[k];
    if (lane < 10 && k - lane - 1 >= 0) x[k - lane - 1] -= H[hbo(k) + lane + 1] * xk;
    __builtin_amdgcn_wave_barrier();
  }
  return true;
}


// ---- damped solve (K6 v2): block cyclic reduction on the 8x8 block-tridiagonal form -------------------------------
// (H + lambda I) x = b with H = blocktridiag(L_j, D_j, L_{j+1}^T), j = 0..Nb-1. Level l (stride s = 2^l) eliminates
// the block rows i = s, 3s, 5s, ... : x_i = P_i (f_i - L_i x_{i-s} - U_i x_{i+s}), P_i = D_i^{-1}, U_i = L_{i+s}^T,
// and folds the Schur complements into the surviving neighbours
//     D_{i-s} -= L_i^T P_i L_i     D_{i+s} -= U_i^T P_i U_i     L_{i+s} := -U_i^T P_i L_i     f_{i+-s} -= {L_i,U_i}^T P_i f_i
// Each elimination is served by 16 lanes (one per right-hand-side column of [L_i | U_i]); every lane factors D_i
// redundantly in registers (LDL^T, 8 pivots) so no intra-group communication is needed. 16 eliminations run
// concurrently per workgroup; depth = ceil(log2 Nb) levels instead of 4n sequential pivots.
// W_L = P L_i, W_U = P U_i and P f_i overwrite the slots of the eliminated row for the back substitution.
// The result is written to dxv; ired[0] = 0 iff some pivot was <= 0 (matrix not positive definite).
// Arguments of an out-of-line device function arrive in VGPRs and count as divergent: every loop bound, block count and address derived
// from them is then computed with vector integer instructions and every loop is an exec-mask loop - ~ 60 of the ~ 900 instructions of a
// reduction round, and each one is ~ 5 cycles at one wave per SIMD. The solves take wave-uniform copies (v_readfirstlane) at entry.
__device__ __forceinline__ int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni_d(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
template <class T> __device__ __forceinline__ T* uni_p(T* p) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ LdsPlan uni_plan(const LdsPlan& p) {
  LdsPlan u;
  u.S = uni_i(p.S); u.solver = uni_i(p.solver); u.off_state = uni_i(p.off_state); u.off_H = uni_i(p.off_H); u.off_b = uni_i(p.off_b);
  u.off_dx = uni_i(p.off_dx); u.off_red = uni_i(p.off_red); u.off_ob = uni_i(p.off_ob); u.ob_cap = uni_i(p.ob_cap); u.total_bytes = uni_i(p.total_bytes);
  return u;
}
__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}

// The damped solve is plain linear algebra without branches on residual values: here (and only here) a*b+c may fuse into
// v_fma_f64. Everything that decides which side of a penalty kink a residual falls on (edges, geometry, association, autoResize)
// is compiled with -ffp-contract=off. The reference has no bit-level contract for the solve either (CSparse eliminates in AMD
// order); measured effect: 5 - 8 % on the C4 step, parity tests unchanged (poses <= 1e-8, identical LM trial counts).
#define TEB_SOLVER_FMA _Pragma("clang fp contract(fast)")
// scheduling fences inside the Schur-product loops of the cyclic reduction (they bound the number of LDS operands in flight)
#define TEB_CR_FENCE_EVERY 4   // rows of a Schur product between two fences (1 fence per row, 2 and 8 rows, no fence: measured, HISTORY.md section 3)
#define TEB_CR_SCHED_BARRIER __builtin_amdgcn_sched_barrier(0);
// Pairs of consecutive doubles at 16-byte aligned addresses are fetched with one 16-byte access (LDS: ds_read_b128, 256 B/clk, where the
// 8-byte aligned pair the compiler forms by itself is a ds_read2_b64 at 128 B/clk). Every 8x8 block starts on a 16-byte boundary (kBlk is
// even, the regions start at even offsets of the 16-byte aligned LDS window / of the hipMalloc'ed scratch) and its rows are 64 bytes.
template <int N>
__device__ __forceinline__ void ld_row(const double* __restrict__ p, double* out) {   // out[0 .. N) = p[0 .. N), p 16-byte aligned
#pragma unroll
  for (int t = 0; t + 1 < N; t += 2) {
    const teb_v2d v = *reinterpret_cast<const teb_v2d*>(p + t);
    out[t] = v.x; out[t + 1] = v.y;
  }
  if (N & 1) out[N - 1] = p[N - 1];
}
struct Ldl8 {   // in-place LDL^T of one 8x8 SPD block: a[r(r+1)/2 + c] holds l_rc (r > c) and 1/d_r on the diagonal
  double a[36];
  __device__ __forceinline__ static constexpr int idx(int r, int c) { return r * (r + 1) / 2 + c; }   // r >= c
  __device__ __forceinline__ void load(const double* Di) {   // Di 16-byte aligned
    ld_row<1>(Di, a + idx(0, 0)); ld_row<2>(Di + 8, a + idx(1, 0)); ld_row<3>(Di + 16, a + idx(2, 0)); ld_row<4>(Di + 24, a + idx(3, 0));
    ld_row<5>(Di + 32, a + idx(4, 0)); ld_row<6>(Di + 40, a + idx(5, 0)); ld_row<7>(Di + 48, a + idx(6, 0)); ld_row<8>(Di + 56, a + idx(7, 0));
  }
  __device__ __forceinline__ bool factor() {
    TEB_SOLVER_FMA
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const double dk = a[idx(k, k)];
      ok = ok && (dk > 0);
      const double inv = fast_rcp(dk);
      a[idx(k, k)] = inv;
#pragma unroll
      for (int r = k + 1; r < 8; ++r) {
        const double ark = a[idx(r, k)];
        const double lrk = ark * inv;
#pragma unroll
        for (int cc = k + 1; cc <= r; ++cc) a[idx(r, cc)] -= lrk * a[idx(cc, k)];   // a[cc][k] still unscaled for cc >= r
        if (false) (void)ark;
      }
#pragma unroll
      for (int r = k + 1; r < 8; ++r) a[idx(r, k)] *= inv;   // scale the column after all its uses
    }
    return ok;
  }
  __device__ __forceinline__ void solve3(double* u, double* v, double* w) const {   // three right-hand sides at once
    TEB_SOLVER_FMA
#pragma unroll
    for (int k = 1; k < 8; ++k) {
#pragma unroll
      for (int m = 0; m < k; ++m) { const double lk = a[idx(k, m)]; u[k] -= lk * u[m]; v[k] -= lk * v[m]; w[k] -= lk * w[m]; }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { const double di = a[idx(k, k)]; u[k] *= di; v[k] *= di; w[k] *= di; }
#pragma unroll
    for (int k = 6; k >= 0; --k) {
#pragma unroll
      for (int m = k + 1; m < 8; ++m) { const double lk = a[idx(m, k)]; u[k] -= lk * u[m]; v[k] -= lk * v[m]; w[k] -= lk * w[m]; }
    }
  }
  __device__ __forceinline__ void solve(double* v) const {   // in place
    TEB_SOLVER_FMA
#pragma unroll
    for (int k = 1; k < 8; ++k) {
#pragma unroll
      for (int m = 0; m < k; ++m) v[k] -= a[idx(k, m)] * v[m];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] *= a[idx(k, k)];
#pragma unroll
    for (int k = 6; k >= 0; --k) {
#pragma unroll
      for (int m = k + 1; m < 8; ++m) v[k] -= a[idx(m, k)] * v[m];
    }
  }
};

// ---- operands by lane broadcast (the 16-lane rounds of the cyclic reduction below) ------------------------------------------------------
// gfx90a+ takes a DPP source on the 64-bit VALU only with row_newbcast:n - lane n of every 16-lane row feeds all lanes of the row - and
// v_fmac_f64 / v_mov_b64 have the encoding. An elimination served by one such row keeps L_i, U_i, the rows of D_i and the factor
// DISTRIBUTED over the registers of its lanes and reads every operand of a neighbour straight out of that lane's register: no LDS
// access between the loads of a round (24 doubles per lane) and its stores, where the 8-lane rounds streamed 196 doubles per lane through
// the LDS pipe the four waves share (round 4: the rounds were bound by exactly that).
// The compiler does not know these statements read another lane's register: the DPP hazard (2 wait states between the VALU write of a
// VGPR and its read through DPP) is covered by the s_nop at the head of every statement - whatever copy or reload the register allocator
// places in front of it has retired by then.
template <int I> struct IC { static constexpr int value = I; constexpr operator int() const { return I; } };
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(IC<I>{}); static_for<I + 1, N>(f); }
}
template <int I, int N, class F> __device__ __forceinline__ void static_for_down(F&& f) {   // I, I - 1, .., N
  if constexpr (I >= N) { f(IC<I>{}); static_for_down<I - 1, N>(f); }
}
template <int T, bool AFTER_BRANCH = false> __device__ __forceinline__ double bcast16(double x) {   // x of lane T of this lane's row
  double r;
  // AFTER_BRANCH: the first DPP statement behind an `if`: should the compiler ever close the branch with a VALU write of EXEC (v_cmpx:
  // today's LLVM forms it on gfx10.3+ only, gfx950 gets s_and_saveexec), the DPP read needs 5 wait states after it, not the 2 of a VGPR
  // write - and the hazard recogniser does not look inside an asm statement. 12 cycles per round.
  if constexpr (AFTER_BRANCH) asm("s_nop 4\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(T));
  else asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(T));
  return r;
}
// The statements below are whole passes - a pivot's update, the forward / backward substitution, two columns of the Schur products - so
// that ONE s_nop covers each (a wait state is an issue slot: s_nop 1 costs 8 cycles, measured 12.8 against 5.1 cycles per dependent
// v_fmac_f64_dpp, tools/micro/dpp_rate_bench.hip; one in front of every instruction was a fifth of a round). Inside a statement no DPP
// source is written: accumulators and right-hand sides are ordinary operands, which the hardware interlocks.
#define TEB_BC(t) " row_newbcast:" #t " row_mask:0xf bank_mask:0xf\n\t"
#define TEB_F1(t, a, x, y) "v_fmac_f64_dpp %" #a ", %" #x ", %" #y TEB_BC(t)            /* a += x(lane t) * y */
#define TEB_FN1(t, a, x, y) "v_fmac_f64_dpp %" #a ", -%" #x ", %" #y TEB_BC(t)          /* a -= x(lane t) * y */
#define TEB_F2(t, a, b, x, ya, yb) TEB_FN1(t, a, x, ya) TEB_FN1(t, b, x, yb)              /* two right-hand sides, one operand */
// operands: %0 .. %7 = Y[0 .. 7], %8 .. %15 = wf[0 .. 7] (in / out), %16 .. %22 = v[0 .. 6] (the factor: l_km in lane k's v[m])
#define TEB_CR16_FORWARD /* Y[k] -= l_km Y[m], m < k: l_km from lane k */ \
  TEB_F2(1, 1, 9, 16, 0, 8) TEB_F2(2, 2, 10, 16, 0, 8) TEB_F2(2, 2, 10, 17, 1, 9) \
  TEB_F2(3, 3, 11, 16, 0, 8) TEB_F2(3, 3, 11, 17, 1, 9) TEB_F2(3, 3, 11, 18, 2, 10) \
  TEB_F2(4, 4, 12, 16, 0, 8) TEB_F2(4, 4, 12, 17, 1, 9) TEB_F2(4, 4, 12, 18, 2, 10) \
  TEB_F2(4, 4, 12, 19, 3, 11) TEB_F2(5, 5, 13, 16, 0, 8) TEB_F2(5, 5, 13, 17, 1, 9) \
  TEB_F2(5, 5, 13, 18, 2, 10) TEB_F2(5, 5, 13, 19, 3, 11) TEB_F2(5, 5, 13, 20, 4, 12) \
  TEB_F2(6, 6, 14, 16, 0, 8) TEB_F2(6, 6, 14, 17, 1, 9) TEB_F2(6, 6, 14, 18, 2, 10) \
  TEB_F2(6, 6, 14, 19, 3, 11) TEB_F2(6, 6, 14, 20, 4, 12) TEB_F2(6, 6, 14, 21, 5, 13) \
  TEB_F2(7, 7, 15, 16, 0, 8) TEB_F2(7, 7, 15, 17, 1, 9) TEB_F2(7, 7, 15, 18, 2, 10) \
  TEB_F2(7, 7, 15, 19, 3, 11) TEB_F2(7, 7, 15, 20, 4, 12) TEB_F2(7, 7, 15, 21, 5, 13) \
  TEB_F2(7, 7, 15, 22, 6, 14)
// The backward pass starts every row with the product of its first term rounded on its own and the scaling by 1 / d_k fused into the
// subtraction - u_k = fma(u_k, 1 / d_k, - (l_{k+1,k} u_{k+1})) - because that is what the compiler made of Ldl8::solve3 in the 8-lane
// rounds of rounds 2 - 4 (`u[k] *= di; .. u[k] -= l * u[m]` under fp contract(fast): the multiply that feeds the subtraction from the
// left is the one that gets fused); with it the 16-lane rounds reproduce those rounds bit for bit (tools/micro/cr_round_bench.hip) and
// every fingerprint, golden vector and reference comparison of the earlier rounds stands. Two statements (operand limit): rows 6 .. 4,
// rows 3 .. 0. Operands: %0 .. %7 = Y, %8 .. %15 = wf, three scratch registers, then v[k] and 1 / d_k of the statement's rows.
#define TEB_CR16_BACKWARD_HI /* k = 6, 5, 4: scratch %16 .. %18 (outputs come first), v[k] = %19 .. %21, inv[k] = %22 .. %24 */ \
  "v_mov_b64_dpp %16, %19" TEB_BC(7) "v_mul_f64 %17, %16, %7\n\tv_mul_f64 %18, %16, %15\n\t" \
  "v_fma_f64 %6, %6, %22, -%17\n\tv_fma_f64 %14, %14, %22, -%18\n\t" "v_mov_b64_dpp %16, %20" TEB_BC(6) \
  "v_mul_f64 %17, %16, %6\n\tv_mul_f64 %18, %16, %14\n\t" "v_fma_f64 %5, %5, %23, -%17\n\tv_fma_f64 %13, %13, %23, -%18\n\t" \
  TEB_F2(7, 5, 13, 20, 7, 15) "v_mov_b64_dpp %16, %21" TEB_BC(5) \
  "v_mul_f64 %17, %16, %5\n\tv_mul_f64 %18, %16, %13\n\t" "v_fma_f64 %4, %4, %24, -%17\n\tv_fma_f64 %12, %12, %24, -%18\n\t" \
  TEB_F2(6, 4, 12, 21, 6, 14) TEB_F2(7, 4, 12, 21, 7, 15)
#define TEB_CR16_BACKWARD_LO /* k = 3 .. 0: scratch %16 .. %18, v[k] = %19 .. %22, inv[k] = %23 .. %26 */ \
  "v_mov_b64_dpp %16, %19" TEB_BC(4) "v_mul_f64 %17, %16, %4\n\tv_mul_f64 %18, %16, %12\n\t" \
  "v_fma_f64 %3, %3, %23, -%17\n\tv_fma_f64 %11, %11, %23, -%18\n\t" TEB_F2(5, 3, 11, 19, 5, 13) \
  TEB_F2(6, 3, 11, 19, 6, 14) TEB_F2(7, 3, 11, 19, 7, 15) \
  "v_mov_b64_dpp %16, %20" TEB_BC(3) "v_mul_f64 %17, %16, %3\n\tv_mul_f64 %18, %16, %11\n\t" \
  "v_fma_f64 %2, %2, %24, -%17\n\tv_fma_f64 %10, %10, %24, -%18\n\t" TEB_F2(4, 2, 10, 20, 4, 12) \
  TEB_F2(5, 2, 10, 20, 5, 13) TEB_F2(6, 2, 10, 20, 6, 14) \
  TEB_F2(7, 2, 10, 20, 7, 15) "v_mov_b64_dpp %16, %21" TEB_BC(2) \
  "v_mul_f64 %17, %16, %2\n\tv_mul_f64 %18, %16, %10\n\t" "v_fma_f64 %1, %1, %25, -%17\n\tv_fma_f64 %9, %9, %25, -%18\n\t" \
  TEB_F2(3, 1, 9, 21, 3, 11) TEB_F2(4, 1, 9, 21, 4, 12) \
  TEB_F2(5, 1, 9, 21, 5, 13) TEB_F2(6, 1, 9, 21, 6, 14) \
  TEB_F2(7, 1, 9, 21, 7, 15) "v_mov_b64_dpp %16, %22" TEB_BC(1) \
  "v_mul_f64 %17, %16, %1\n\tv_mul_f64 %18, %16, %9\n\t" "v_fma_f64 %0, %0, %26, -%17\n\tv_fma_f64 %8, %8, %26, -%18\n\t" \
  TEB_F2(2, 0, 8, 22, 2, 10) TEB_F2(3, 0, 8, 22, 3, 11) \
  TEB_F2(4, 0, 8, 22, 4, 12) TEB_F2(5, 0, 8, 22, 5, 13) \
  TEB_F2(6, 0, 8, 22, 6, 14) TEB_F2(7, 0, 8, 22, 7, 15)
// operands: %0 .. %15 = acc[0 .. 15], then (X[k], Y[k]) of two columns k: acc[t] += X[k](lane t) * Y[k], t < 8; acc[t] -= .., t >= 8.
// The upper eight are accumulated NEGATED (round 6): what is stored from them is the new coupling -U^T W_L, and the survivor above gets
// D - U^T W_U = D + acc; a sum of negated products is the negated sum bit for bit (round-to-nearest is symmetric), so the eight sign flips
// per lane and round in front of the stores are gone and nothing else changes.
#define TEB_CR16_SCHUR2 \
  TEB_F1(0, 0, 16, 17) TEB_F1(1, 1, 16, 17) TEB_F1(2, 2, 16, 17) TEB_F1(3, 3, 16, 17) \
  TEB_F1(4, 4, 16, 17) TEB_F1(5, 5, 16, 17) TEB_F1(6, 6, 16, 17) TEB_F1(7, 7, 16, 17) \
  TEB_FN1(8, 8, 16, 17) TEB_FN1(9, 9, 16, 17) TEB_FN1(10, 10, 16, 17) TEB_FN1(11, 11, 16, 17) \
  TEB_FN1(12, 12, 16, 17) TEB_FN1(13, 13, 16, 17) TEB_FN1(14, 14, 16, 17) TEB_FN1(15, 15, 16, 17) \
  TEB_F1(0, 0, 18, 19) TEB_F1(1, 1, 18, 19) TEB_F1(2, 2, 18, 19) TEB_F1(3, 3, 18, 19) \
  TEB_F1(4, 4, 18, 19) TEB_F1(5, 5, 18, 19) TEB_F1(6, 6, 18, 19) TEB_F1(7, 7, 18, 19) \
  TEB_FN1(8, 8, 18, 19) TEB_FN1(9, 9, 18, 19) TEB_FN1(10, 10, 18, 19) TEB_FN1(11, 11, 18, 19) \
  TEB_FN1(12, 12, 18, 19) TEB_FN1(13, 13, 18, 19) TEB_FN1(14, 14, 18, 19) TEB_FN1(15, 15, 18, 19)
template <int K> __device__ __forceinline__ void cr16_pivot_update(double (&v)[8], double lk);   // v[m] -= a_mk(lane m) * l_jk, m = K + 1 .. 7
template <> __device__ __forceinline__ void cr16_pivot_update<0>(double (&v)[8], double lk) {
  asm("s_nop 1\n\t"
  TEB_FN1(1, 0, 7, 8) TEB_FN1(2, 1, 7, 8) TEB_FN1(3, 2, 7, 8) TEB_FN1(4, 3, 7, 8)
  TEB_FN1(5, 4, 7, 8) TEB_FN1(6, 5, 7, 8) TEB_FN1(7, 6, 7, 8)
      : "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(v[0]), "v"(lk));
}
template <> __device__ __forceinline__ void cr16_pivot_update<1>(double (&v)[8], double lk) {
  asm("s_nop 1\n\t"
  TEB_FN1(2, 0, 6, 7) TEB_FN1(3, 1, 6, 7) TEB_FN1(4, 2, 6, 7) TEB_FN1(5, 3, 6, 7)
  TEB_FN1(6, 4, 6, 7) TEB_FN1(7, 5, 6, 7)
      : "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(v[1]), "v"(lk));
}
template <> __device__ __forceinline__ void cr16_pivot_update<2>(double (&v)[8], double lk) {
  asm("s_nop 1\n\t"
  TEB_FN1(3, 0, 5, 6) TEB_FN1(4, 1, 5, 6) TEB_FN1(5, 2, 5, 6) TEB_FN1(6, 3, 5, 6)
  TEB_FN1(7, 4, 5, 6)
      : "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(v[2]), "v"(lk));
}
template <> __device__ __forceinline__ void cr16_pivot_update<3>(double (&v)[8], double lk) {
  asm("s_nop 1\n\t"
  TEB_FN1(4, 0, 4, 5) TEB_FN1(5, 1, 4, 5) TEB_FN1(6, 2, 4, 5) TEB_FN1(7, 3, 4, 5)
      : "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(v[3]), "v"(lk));
}
template <> __device__ __forceinline__ void cr16_pivot_update<4>(double (&v)[8], double lk) {
  asm("s_nop 1\n\t"
  TEB_FN1(5, 0, 3, 4) TEB_FN1(6, 1, 3, 4) TEB_FN1(7, 2, 3, 4)
      : "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(v[4]), "v"(lk));
}
template <> __device__ __forceinline__ void cr16_pivot_update<5>(double (&v)[8], double lk) {
  asm("s_nop 1\n\t"
  TEB_FN1(6, 0, 2, 3) TEB_FN1(7, 1, 2, 3)
      : "+v"(v[6]), "+v"(v[7]) : "v"(v[5]), "v"(lk));
}
template <> __device__ __forceinline__ void cr16_pivot_update<6>(double (&v)[8], double lk) {
  asm("s_nop 1\n\t"
  TEB_FN1(7, 0, 1, 2)
      : "+v"(v[7]) : "v"(v[6]), "v"(lk));
}


// One elimination in the registers of a 16-lane row. Lane (q, c), q = lane bit 3, c = lane & 7, holds
//     v[m]   row c of D_i (its lower part, m <= c, is what counts; lanes q = 1 carry a copy nobody reads)
//     X[k]   q = 0: column c of L_i          q = 1: column c of U_i = row c of L_{i+s}
//     wf[k]  f_i (every lane)
// and leaves  Y = P_i X (column c of W_L / of W_U), wf = P_i f_i, acc / sx = its share of the Schur products:
//     q = 0:  acc[t] = (L_i^T W_L)[t][c]   acc[8 + t] = - (U_i^T W_L)[t][c]   sx = (L_i^T P_i f_i)[c]
//     q = 1:  acc[t] = (L_i^T W_U)[t][c]   acc[8 + t] = - (U_i^T W_U)[t][c]   sx = (U_i^T P_i f_i)[c]
// (acc[0 .. 8) of the lanes q = 1 is the transpose of U_i^T W_L, rounded differently; not used.) Everything is stored by COLUMN c: the 8
// lanes of a half row then touch 64 contiguous bytes per access, which the LDS serves without a bank conflict - a 64-byte row per lane in
// 16-byte accesses (lanes 64 bytes apart: 4-way conflicts in every ds_write_b128) measured 10 % slower per round.
// LDL^T by rows: at pivot k lane j > k scales its l_jk = a_jk / d_k and updates a_jm -= l_jk a_mk, m = k + 1 .. j, with a_mk read from
// lane m; the triangular solves read l_km from lane k / l_mk from lane m. Operation by operation this is Ldl8::factor / solve3 and the
// 8-lane round's products (same operands, same order of every sum; checked in IEEE arithmetic on the host), down to the compiler's
// choice in the back substitution of rounds 2 - 4 - it fused the FIRST term, u_k = fma(u_k, 1/d_k, -(l u_{k+1})), which the
// TEB_CR16_BACKWARD blocks spell out: results equal the 8-lane rounds bit for bit (tools/micro/cr_round_bench.hip compares them).
#ifndef TEB_CR16_STAMP
#define TEB_CR16_STAMP(k)   // (tools/micro/cr_round_bench.hip: cycle stamps between the passes)
#endif
__device__ __forceinline__ bool cr16_eliminate(double (&v)[8], const double (&X)[8], double (&Y)[8], double (&wf)[8], double (&acc)[16], double& sx) {
  TEB_SOLVER_FMA
  int ok = 1;
  TEB_CR16_STAMP(0)
  double inv[8];
  static_for<0, 8>([&](auto K) {
    constexpr int k = decltype(K)::value;
    const double dk = bcast16<k, k == 0>(v[k]);   // (k = 0: the first DPP read behind the `if (act)` of the round)
    ok &= (dk > 0);
    inv[k] = fast_rcp(dk);
    if constexpr (k < 7) {
      const double lk = v[k] * inv[k];
      cr16_pivot_update<k>(v, lk);   // (lanes j < m: their upper part, never read)
      v[k] = lk;                     // after all its uses as the unscaled a_jk
    }
  });
  TEB_CR16_STAMP(1)
#pragma unroll
  for (int k = 0; k < 8; ++k) Y[k] = X[k];
#define TEB_CR16_RHS "+v"(Y[0]), "+v"(Y[1]), "+v"(Y[2]), "+v"(Y[3]), "+v"(Y[4]), "+v"(Y[5]), "+v"(Y[6]), "+v"(Y[7]), \
                     "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(wf[4]), "+v"(wf[5]), "+v"(wf[6]), "+v"(wf[7])
#define TEB_CR16_FACTOR "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6])
  asm("s_nop 1\n\t" TEB_CR16_FORWARD : TEB_CR16_RHS : TEB_CR16_FACTOR);
  TEB_CR16_STAMP(2)
  Y[7] *= inv[7]; wf[7] *= inv[7];
  {
    double sl, s1, s2;   // scratch of the statements: l_{k+1,k} and the two first-term products
    asm("s_nop 1\n\t" TEB_CR16_BACKWARD_HI : TEB_CR16_RHS, "=&v"(sl), "=&v"(s1), "=&v"(s2) : "v"(v[6]), "v"(v[5]), "v"(v[4]), "v"(inv[6]), "v"(inv[5]), "v"(inv[4]));
    asm("s_nop 1\n\t" TEB_CR16_BACKWARD_LO : TEB_CR16_RHS, "=&v"(sl), "=&v"(s1), "=&v"(s2)
        : "v"(v[3]), "v"(v[2]), "v"(v[1]), "v"(v[0]), "v"(inv[3]), "v"(inv[2]), "v"(inv[1]), "v"(inv[0]));
  }
#undef TEB_CR16_RHS
#undef TEB_CR16_FACTOR
  TEB_CR16_STAMP(3)
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = 0;
  sx = 0;
#pragma unroll
  for (int k = 0; k < 8; k += 2) {
    asm("s_nop 1\n\t" TEB_CR16_SCHUR2
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]),
          "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15])
        : "v"(X[k]), "v"(Y[k]), "v"(X[k + 1]), "v"(Y[k + 1]));
    sx += X[k] * wf[k];
    sx += X[k + 1] * wf[k + 1];
  }
  TEB_CR16_STAMP(4)
  return ok != 0;
}

#ifdef TEB_PROFILE
__device__ long long g_cr_prof[8];
__device__ long long g_crw_prof[32];   // per group width (8, 16, 32, 64 lanes): 6 sections of a round + the number of rounds
#define CRP_DECL long long crp_t0 = clock64(), crp_t1;
#define CRP(k) do { crp_t1 = clock64(); if (blockIdx.x == 0 && threadIdx.x == 0) g_cr_prof[k] += crp_t1 - crp_t0; crp_t0 = crp_t1; } while (0)
#define CRR_DECL long long crr_t0 = clock64(), crr_t1;
#define CRR(k) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); crr_t1 = clock64(); __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 0 && threadIdx.x == 0) { g_crw_prof[kW * 8 + k] += crr_t1 - crr_t0; if (k == 5) g_crw_prof[kW * 8 + 6] += 1; } crr_t0 = crr_t1; } while (0)
#else
#define CRP_DECL
#define CRP(k)
#define CRR_DECL
#define CRR(k)
#endif
// ---- pieces of the block cyclic reduction, usable on blocks in either memory (all three layouts run their LDS-resident levels
//      through them; the HBM layout runs its finer levels on HBM-resident blocks and the coarser ones on a compact copy in LDS).
// bD / bF: distance in doubles between consecutive block rows of the system in D, L / in f (kBlk / 8 for a contiguous system).
// One round of a level with 16 lanes per elimination (cr16_eliminate): the eliminations e0 .. e0 + kThreads / 16 - 1 of the rows
// i = s (2 e + 1). Loads per lane: row c of D_i, its column of L_i (q = 0) or row of L_{i+s} (q = 1), f_i. Stores, all by column c
// (the 8 lanes of a half row touch 64 contiguous bytes per access: conflict-free): q = 0 folds L_i^T W_L into D_{i-s}, writes the new
// L_{i+s} = - U_i^T W_L and W_L into the slot of D_i; q = 1 writes W_U into the slot of L_i and, after the barrier (neighbouring
// eliminations share the surviving row between them), folds U_i^T W_U into D_{i+s}.
// A surviving row receives two updates: L^T W_L of the elimination above it (phase 1) and U^T W_U of the one below (phase 2), in that
// order inside a round. Rounds 2 - 4 ran 32 eliminations per round; with 16, the row between the eliminations 16 j + 15 and 16 j + 16,
// j even, would get the two in the other order - other bits in its 64 entries. A pair of rounds therefore behaves like one of the old
// ones: the first of a pair (PAIR = 1) leaves the phase-2 update of its last group pending (registers of those lanes), the second
// (PAIR = 2) applies it after its own barrier. PAIR = 0: a round on its own.
struct Cr16Pending { double acc[8]; double sx; int row; };   // U^T W_U column c and (U^T P f)[c] of the elimination below `row`; row < 0: nothing pending
template <int PAIR>
__device__ __forceinline__ bool cr_forward_round16(double* __restrict__ D, double* __restrict__ L, double* __restrict__ f, int Nb, int s,
                                                   int e0, int E, Cr16Pending& pend, int bD = kBlk, int bF = 8) {
  TEB_SOLVER_FMA
  constexpr int kW = 1;   // (profiling build) row of the per-width counters
  (void)kW;
  const int tid = threadIdx.x;
  const int grp = tid >> 4, c = tid & 7;
  const bool up = (tid & 8) != 0;
  const int e = e0 + grp;
  const bool act = e < E;
  const int i = s * (2 * e + 1);
  const bool hasU = act && (i + s < Nb);
  bool ok = true;
  double Y[8], wf[8], acc[16], dm[8];
  double sx = 0, fm = 0;
  CRR_DECL
  TEB_CR16_STAMP(5)
  if (act) {
    const double* Di = D + i * bD;
    const double* Li = L + i * bD;
    const double* Lp = L + (i + s) * bD;   // U_i^T, valid iff hasU
    double v[8], X[8];
    // Only the lower half row (q = 0) is ever read for the factor (every DPP source lane is < 8) and only it folds into row i - s in
    // phase 1: the upper half loads neither - its v stays whatever the registers held (it runs the pivot arithmetic on it, nobody
    // looks) - which takes 136 of the 264 bytes a lane of that half fetched per round off the LDS pipe the four waves share; the wait for
    // these loads is all a round's start consists of (round 6, bit-identical: - 2 % on the blocks layout, whose every level is such a
    // round; issuing the factor's row first or fencing the copies of X behind the factorisation lost to the compiler's own order).
    if (up) {
      ld_row<8>((hasU ? Lp : Li) + c * 8, X);   // row c of L_{i+s} (an address inside the blocks even without an upper neighbour)
      if (!hasU) {
#pragma unroll
        for (int k = 0; k < 8; ++k) X[k] = 0.0;
      }
    } else {
      ld_row<8>(Di + c * 8, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) X[k] = Li[k * 8 + c];
      // what phase 1 updates is fetched now, under the elimination: nobody writes row i - s before this group does (the lower
      // neighbour's turn at it comes after the barrier, or is pending)
      const double* Dm = D + (i - s) * bD;
#pragma unroll
      for (int t = 0; t < 8; ++t) dm[t] = Dm[t * 8 + c];
      fm = f[(i - s) * bF + c];
    }
    ld_row<8>(f + i * bF, wf);
    CRR(0);
    ok = cr16_eliminate(v, X, Y, wf, acc, sx);
    CRR(2);
    // No barrier before the writes: within a level the eliminated rows i = s (2 e + 1) and the blocks read for them (D_i, L_i,
    // L_{i+s}, f_i) belong to exactly one group, a group is a quarter of a wave, and the survivors' D / f are only updated by the group(s)
    // that eliminate their neighbours: the upper one in this phase, the lower one after the barrier. (ONE `if (act)` around the
    // elimination and its stores: closed and reopened, the 33 values that cross the gap became phi nodes the compiler zero-filled on the
    // inactive side - 18 moves per round.)
    double* slot = (up ? L : D) + i * bD;   // W_U -> L_i, W_L -> D_i: operands of the back substitution
#pragma unroll
    for (int k = 0; k < 8; ++k) slot[k * 8 + c] = Y[k];
    if (!up) {
      double* Dm = D + (i - s) * bD;
#pragma unroll
      for (int t = 0; t < 8; ++t) Dm[t * 8 + c] = dm[t] - acc[t];
      if (hasU) {
        double* Lp = L + (i + s) * bD;
#pragma unroll
        for (int t = 0; t < 8; ++t) Lp[t * 8 + c] = acc[8 + t];   // (= - U_i^T W_L: accumulated negated)
      }
      f[(i - s) * bF + c] = fm - sx;
      // P f_i: every lane of the row holds all 8 components; lane 0 stores them (four 16-byte accesses) instead of every lane selecting
      // its own out of its registers (15 compare / select instructions)
      if (c == 0) {
#pragma unroll
        for (int k = 0; k < 8; k += 2) *reinterpret_cast<teb_v2d*>(f + i * bF + k) = teb_v2d{wf[k], wf[k + 1]};
      }
    }
  }
  CRR(3);
  TEB_CR16_STAMP(6)
  __syncthreads();
  CRR(4);
  TEB_CR16_STAMP(7)
  // phase 2. The last group of the first round of a pair keeps its update when the second round has an elimination above that row.
  const bool keep = PAIR == 1 && grp == kThreads / 16 - 1 && e + 1 < E;
  if (PAIR == 2 && pend.row >= 0 && up) {   // (the lanes that kept it: same group, same half)
    double* Dp = D + pend.row * bD;
#pragma unroll
    for (int t = 0; t < 8; ++t) dm[t] = Dp[t * 8 + c];
    fm = f[pend.row * bF + c];
#pragma unroll
    for (int t = 0; t < 8; ++t) Dp[t * 8 + c] = dm[t] + pend.acc[t];   // (pend.acc = - U^T W_U)
    f[pend.row * bF + c] = fm - pend.sx;
  }
  if (PAIR == 1) {
    pend.row = (keep && hasU && up) ? i + s : -1;
#pragma unroll
    for (int t = 0; t < 8; ++t) pend.acc[t] = acc[8 + t];
    pend.sx = sx;
  }
  if (hasU && up && !keep) {
    double* Dp = D + (i + s) * bD;
#pragma unroll
    for (int t = 0; t < 8; ++t) dm[t] = Dp[t * 8 + c];
    fm = f[(i + s) * bF + c];
#pragma unroll
    for (int t = 0; t < 8; ++t) Dp[t * 8 + c] = dm[t] + acc[8 + t];   // (acc[8 ..] = - U_i^T W_U)
    f[(i + s) * bF + c] = fm - sx;
  }
  TEB_CR16_STAMP(8)
  // The first round of a pair needs no barrier behind its phase 2: the second round (eliminations e0 + 16 ..) reads and writes block rows
  // from s (2 e0 + 32) up only - its own eliminated rows, the survivor below its first group (whose update from this round is the pending
  // one, in registers) and the rows above - while this phase touched the survivors s (2 e + 2), e <= e0 + 14, and nothing of L.
  if (PAIR != 1) __syncthreads();
  CRR(5);
  TEB_CR16_STAMP(9)
  return ok;
}
// rows live at D + i * kBlk, L + i * kBlk (coupling of row i with the previous surviving row), f + i * 8; levels s_lo, 2 s_lo, .. < s_hi.
// (Rounds 2 - 4 served an elimination with 8 lanes, 32 per round, and widened the groups to 32 / 64 lanes at the coarse levels to shorten
// the Schur products read from LDS; with the operands read from registers one width serves all levels -
// tools/micro/cr_round_bench.hip keeps the 8-lane round for the comparison: same bits.)
__device__ __forceinline__ bool cr_forward(double* __restrict__ D, double* __restrict__ L, double* __restrict__ f, int Nb, int s_lo,
                                           int s_hi, int bD = kBlk, int bF = 8) {
  bool ok = true;
  Cr16Pending pend;
  pend.row = -1;
  for (int s = s_lo; s < s_hi; s <<= 1) {
    const int E = (Nb - 1 - s) / (2 * s) + 1;
    for (int e0 = 0; e0 < E; e0 += kThreads / 8) {
      if (E - e0 > kThreads / 16) {
        ok = cr_forward_round16<1>(D, L, f, Nb, s, e0, E, pend, bD, bF) && ok;
        ok = cr_forward_round16<2>(D, L, f, Nb, s, e0 + kThreads / 16, E, pend, bD, bF) && ok;
      } else
        ok = cr_forward_round16<0>(D, L, f, Nb, s, e0, E, pend, bD, bF) && ok;
    }
  }
  return ok;
}
#ifdef TEB_AMD_MFMA_SCHUR
// ---- the same forward step with the Schur update on the matrix cores (north_star: "MFMA only on the dense Schur block") -----------
// Per elimination the update of the neighbours is ONE 16 x 16 x 8 fp64 contraction
//     C = [L_i | U_i]^T  [W_L | W_U],   W = P_i [L_i | U_i]     (U_i = L_{i+s}^T)
//     C(0:8, 0:8) = L_i^T W_L  -> D_{i-s} -= .     C(8:16, 0:8) = U_i^T W_L -> L_{i+s} := -.     C(8:16, 8:16) = U_i^T W_U -> D_{i+s} -= .
// i.e. two v_mfma_f64_16x16x4_f64 per elimination (the quadrant C(0:8, 8:16) is the transpose of C(8:16, 0:8) and is dropped: 75 % of
// the MACs are useful). A wave serves its 8 eliminations one after the other; the factorisation of D_i and the three triangular solves
// stay per-lane VALU code (8 lanes per elimination, as in cr_forward). What the matrix instruction buys is not rate (on gfx950 the fp64
// matrix rate equals the fp64 vector rate) but operand traffic: a lane loads 2 + 2 operand values per elimination instead of streaming
// the whole of L_i and L_{i+s} (128 values) for its column of the product. Lane l holds A(m = l & 15, k = l >> 4),
// B(k = l >> 4, n = l & 15) and C(m = (l >> 4) + 4 r, n = l & 15), r = 0..3 (checked on the device by teb_amd_debug_mfma_selftest).
typedef double v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bool cr_forward_mfma(double* __restrict__ D, double* __restrict__ L, double* __restrict__ f, int Nb, int s_lo,
                                                int s_hi, int bD = kBlk, int bF = 8) {
  TEB_SOLVER_FMA
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int grp = tid >> 3, c = tid & 7;
  const int am = lane & 15, ak = lane >> 4, an = am & 7;
  const bool left = am < 8;
  bool ok = true;
  for (int s = s_lo; s < s_hi; s <<= 1) {
    const int E = (Nb - 1 - s) / (2 * s) + 1;
    for (int e0 = 0; e0 < E; e0 += kThreads / 8) {
      const int e = e0 + grp;
      const bool act = e < E;
      const int i = s * (2 * e + 1);
      const bool hasU = act && (i + s < Nb);
      // A operands of the 8 eliminations of this wave, fetched before anything overwrites L_i / L_{i+s}
      double A0[8], A1[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int eq = e0 + 8 * wv + q, iq = s * (2 * eq + 1);
        const bool v = left ? (eq < E) : (eq < E && iq + s < Nb);
        const double* src = left ? L + iq * bD + am : L + (iq + s) * bD + an * 8;   // L_i[k][m]  |  L_{i+s}[m - 8][k]
        const int st = left ? 8 : 1;
        A0[q] = v ? src[ak * st] : 0.0;
        A1[q] = v ? src[(ak + 4) * st] : 0.0;
      }
      double wL[8], wU[8], wf[8];
      double s1 = 0, s2 = 0;
      if (act) {
        double* Di = D + i * bD;
        double* Li = L + i * bD;
        const double* Lp = L + (i + s) * bD;   // U_i^T, valid iff hasU
        Ldl8 F;
        F.load(Di);
        ok = F.factor() && ok;
        double cl[8], cu[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          cl[k] = wL[k] = Li[k * 8 + c];
          cu[k] = wU[k] = hasU ? Lp[c * 8 + k] : 0.0;
          wf[k] = f[i * bF + k];
        }
        F.solve3(wL, wU, wf);
#pragma unroll
        for (int k = 0; k < 8; ++k) { s1 += cl[k] * wf[k]; s2 += cu[k] * wf[k]; }   // (L_i^T P f_i)[c], (L_{i+s} P f_i)[c]
        // W_L, W_U take the slots of the eliminated row: operands of the matrix instruction below, and of the back substitution later
#pragma unroll
        for (int k = 0; k < 8; ++k) { Di[k * 8 + c] = wL[k]; Li[k * 8 + c] = wU[k]; }
      }
      // same wave: LDS operations complete in order, so the operand loads below see the stores above; no other wave touches these slots
      v4d C[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int eq = e0 + 8 * wv + q, iq = s * (2 * eq + 1);
        C[q] = v4d{0.0, 0.0, 0.0, 0.0};
        if (eq < E) {
          const double* W = (left ? D : L) + iq * bD + an;   // W_L | W_U, entry (k, n)
          const double b0 = W[ak * 8], b1 = W[(ak + 4) * 8];
          C[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(A0[q], b0, C[q], 0, 0, 0);
          C[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(A1[q], b1, C[q], 0, 0, 0);
        }
      }
      // phase 1: D_{i-s} -= C(0:8, 0:8), L_{i+s} := -C(8:16, 0:8)   (lanes n < 8)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int eq = e0 + 8 * wv + q, iq = s * (2 * eq + 1);
        if (eq < E && left) {
          double* Dm = D + (iq - s) * bD;
          Dm[ak * 8 + am] -= C[q][0];
          Dm[(ak + 4) * 8 + am] -= C[q][1];
          if (iq + s < Nb) {
            double* Lq = L + (iq + s) * bD;
            Lq[ak * 8 + am] = -C[q][2];
            Lq[(ak + 4) * 8 + am] = -C[q][3];
          }
        }
      }
      if (act) { f[(i - s) * bF + c] -= s1; f[i * bF + c] = wf[c]; }
      __syncthreads();
      // phase 2: D_{i+s} -= C(8:16, 8:16)   (lanes n >= 8)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int eq = e0 + 8 * wv + q, iq = s * (2 * eq + 1);
        if (eq < E && !left && iq + s < Nb) {
          double* Dp = D + (iq + s) * bD;
          Dp[ak * 8 + an] -= C[q][2];
          Dp[(ak + 4) * 8 + an] -= C[q][3];
        }
      }
      if (hasU) f[(i + s) * bF + c] -= s2;
      __syncthreads();
    }
  }
  return ok;
}
// A(16 x 8) B(8 x 16) with the operand maps used above -> C (16 x 16); also times `reps` dependent-free issues per wave (clock64 ticks)
// (one definition: the host translation unit's)
#ifdef TEB_AMD_MAIN_TU
__global__ void mfma_selftest_kernel(const double* A, const double* B, double* Cout, int reps, long long* ticks) {
  const int lane = threadIdx.x & 63, am = lane & 15, ak = lane >> 4;
  const double a0 = A[am * 8 + ak], a1 = A[am * 8 + ak + 4], b0 = B[ak * 16 + am], b1 = B[(ak + 4) * 16 + am];
  v4d acc = {0.0, 0.0, 0.0, 0.0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) Cout[(ak + 4 * r) * 16 + am] = acc[r];
  v4d t0 = {0, 0, 0, 0}, t1 = t0, t2 = t0, t3 = t0;
  const long long c0 = clock64();
  for (int it = 0; it < reps; ++it) {   // four independent accumulators: issue-bound, not latency-bound
    t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, t0, 0, 0, 0);
    t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, t1, 0, 0, 0);
    t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, t2, 0, 0, 0);
    t3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, t3, 0, 0, 0);
  }
  const long long c1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = c1 - c0; ticks[1] = (long long)(t0[0] + t1[1] + t2[2] + t3[3]); }
}
#endif
#define TEB_CR_FORWARD cr_forward_mfma
#else
#define TEB_CR_FORWARD cr_forward
#endif

// x_0 = D_0^{-1} f_0 of the last surviving row
__device__ __forceinline__ bool cr_top(const double* __restrict__ D, double* __restrict__ f) {
  TEB_SOLVER_FMA
  bool ok = true;
  const int tid = threadIdx.x;
  if (tid < 8) {
    double v[8];
    ld_row<8>(f, v);
    Ldl8 F;
    F.load(D);
    ok = F.factor();
    F.solve(v);
    double mine = v[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) mine = (tid == k) ? v[k] : mine;
    f[tid] = mine;
  }
  return ok;
}
// back substitution for the levels s_from, s_from / 2, .., s_to
__device__ __forceinline__ void cr_backward(const double* __restrict__ D, const double* __restrict__ L, double* __restrict__ f, int Nb,
                                            int s_from, int s_to, int bD = kBlk, int bF = 8) {
  TEB_SOLVER_FMA
  const int tid = threadIdx.x;
  for (int s = s_from; s >= s_to; s >>= 1) {
    const int E = (Nb - 1 - s) / (2 * s) + 1;
    for (int u = tid; u < E * 8; u += kThreads) {
      const int e = u >> 3, r = u & 7;
      const int i = s * (2 * e + 1);
      double acc = f[i * bF + r], acc2 = 0;
      double WL[8], xm[8];
      ld_row<8>(D + i * bD + r * 8, WL);
      ld_row<8>(f + (i - s) * bF, xm);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc -= WL[k] * xm[k];
      if (i + s < Nb) {
        double WU[8], xp[8];
        ld_row<8>(L + i * bD + r * 8, WU);
        ld_row<8>(f + (i + s) * bF, xp);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc2 -= WU[k] * xp[k];
      }
      f[i * bF + r] = acc + acc2;
    }
    __syncthreads();
  }
}

// Out of line by default: measured on MI355X (headline step, round 2) 3.72 ms vs 4.39 ms inlined. The price of the call is the
// callee-saved register block it spills and reloads (~ 980 B per lane and call: 5 of the 7.8 GB of fabric traffic per launch);
// -DTEB_AMD_INLINE_SOLVE builds the inlined variant (4.1 GB per launch, 18 % slower: it spills inside the loops instead).
#ifdef TEB_AMD_INLINE_SOLVE
#define TEB_SOLVE_LINKAGE __forceinline__
#elif defined(TEB_AMD_SOLVE_CSR)
// (per-unit build flag, build.py: the plain calling convention - the callee saves and restores its callee-saved block - for the one
// instantiation in which the no-callee-saved call was found to corrupt the caller: see UNIT_FLAGS)
#define TEB_SOLVE_LINKAGE __noinline__
#else
#define TEB_SOLVE_LINKAGE __noinline__ __attribute__((not_tail_called))
#endif
// GLOBAL == false: the blocks are the LDS-resident normal matrix (SOLVER_CR), which the solve destroys.
// GLOBAL == true : the normal matrix is a band in HBM (SOLVER_BANDG, bands too long for any LDS layout); its 8x8 blocks (+ lambda)
//                  are expanded into a per-band HBM buffer (L2-resident: ~70 doubles per pose) and the same reduction runs there -
//                  fine levels as round trips to L2, coarse levels on a compact copy in LDS; no backup / restore of H since the
//                  band is never touched.
template <bool GLOBAL, bool HB_GLOBAL>
__device__ __forceinline__ void cr_solve_t_impl(const LdsPlan plan_, const SceneDev& sc, int n_, double lambda_, double* gbuf_, double* gH_) {
  TEB_SOLVER_FMA
  extern __shared__ __attribute__((aligned(16))) double lds_base[];
  CRP_DECL
#ifdef TEB_AMD_INLINE_SOLVE
  const LdsPlan plan = plan_; const int n = n_; const double lambda = lambda_; double* gbuf = gbuf_; double* gH = gH_;
#else
  const LdsPlan plan = uni_plan(plan_); const int n = uni_i(n_); const double lambda = uni_d(lambda_); double* gbuf = uni_p(gbuf_); double* gH = uni_p(gH_);
#endif
  const Lds l = carve(lds_base, plan, gH, HB_GLOBAL);
  const int tid = threadIdx.x;
  const int Nt = 4 * n, Nb = (Nt + 7) >> 3;
  double* __restrict__ D = GLOBAL ? gbuf : l.Db;
  double* __restrict__ L = GLOBAL ? gbuf + (size_t)Nb * kBlk : l.Lb;
  double* __restrict__ f = GLOBAL ? gbuf + (size_t)2 * Nb * kBlk : l.fb;
  if (GLOBAL) {
    const double* Hb = l.Hb;
    for (int q = tid; q < Nb * 128; q += kThreads) {
      const int j = q >> 7, w = q & 127, a = (w & 63) >> 3, bcol = w & 7;
      const int r = 8 * j + a;
      double v = 0;
      if (w < 64) {            // D_j[a][bcol]
        const int cc = 8 * j + bcol;
        if (r < Nt && cc < Nt) v = (cc <= r) ? Hb[hbo(r) + (r - cc)] : Hb[hbo(cc) + (cc - r)];
        else if (r == cc) v = 1.0;
        if (r == cc) v += lambda;
        D[j * kBlk + a * 8 + bcol] = v;
      } else {                 // L_j[a][bcol] = H[8j+a][8(j-1)+bcol]
        const int d = 8 + a - bcol;
        if (j >= 1 && r < Nt && d < kBand) v = Hb[hbo(r) + d];
        L[j * kBlk + a * 8 + bcol] = v;
      }
    }
    for (int q = tid; q < Nb * 8; q += kThreads) f[q] = (q < Nt) ? l.bv[q] : 0.0;
  } else {
    for (int q = tid; q < Nb * 8; q += kThreads) {
      f[q] = (q < Nt) ? l.bv[q] : 0.0;
      D[(q >> 3) * kBlk + (q & 7) * 9] += lambda;
    }
  }
  if (tid == 0) l.ired[0] = 1;
  __syncthreads();
  CRP(0);
  if constexpr (GLOBAL) {
    // Finer levels on the HBM blocks; as soon as the surviving rows (every s0-th) fit the LDS region this layout keeps for them (the
    // autoResize scratch, sized for up to 64 compact rows: hmat_doubles; until round 5 the obstacle cache was borrowed - a few rows) they
    // are copied there and the remaining levels, the top solve and their back substitution run in LDS. The compact system is an exact
    // re-indexing (row j' = j / s0, stride s' = s / s0), so the arithmetic is that of the all-HBM reduction.
    const int lds_doubles = plan.off_b - plan.off_H;   // (the region of the autoResize scratch: hmat_doubles; nobody else uses it during a solve)
    int s0 = 1;
    while (s0 < Nb && 2 * ((Nb + s0 - 1) / s0) * kBlk > lds_doubles) s0 <<= 1;
    const bool use_lds = lds_doubles > 0 && s0 < Nb && 8 * ((Nb + s0 - 1) / s0) <= 4 * plan.S + 8;
    if (!use_lds) s0 = Nb;   // everything on the HBM blocks
    bool ok = cr_forward(D, L, f, Nb, 1, s0 < Nb ? s0 : Nb);
    CRP(1);
    if (use_lds) {
      const int Nc = (Nb + s0 - 1) / s0;
      double* Dc = lds_base + plan.off_H;
      double* Lc = Dc + Nc * kBlk;
      double* fc = lds_base + plan.off_dx;
      for (int q = tid; q < Nc * 64; q += kThreads) {
        const int j = q >> 6, w = q & 63;
        Dc[j * kBlk + w] = D[(size_t)j * s0 * kBlk + w];
        Lc[j * kBlk + w] = L[(size_t)j * s0 * kBlk + w];
      }
      for (int q = tid; q < Nc * 8; q += kThreads) fc[q] = f[(size_t)(q >> 3) * s0 * 8 + (q & 7)];
      __syncthreads();
      CRP(2);
      ok = cr_forward(Dc, Lc, fc, Nc, 1, Nc) && ok;   // (the MFMA build keeps the vector code for this layout, SOLVER_BANDG)
      ok = cr_top(Dc, fc) && ok;
      __syncthreads();
      int stop = 1;
      while (stop * 2 < Nc) stop *= 2;
      if (Nc > 1) cr_backward(Dc, Lc, fc, Nc, stop, 1);
      for (int q = tid; q < Nc * 8; q += kThreads) f[(size_t)(q >> 3) * s0 * 8 + (q & 7)] = fc[q];
      __syncthreads();
      CRP(3);
      if (s0 > 1) cr_backward(D, L, f, Nb, s0 >> 1, 1);
    } else {
      ok = cr_top(D, f) && ok;
      __syncthreads();
      int stop = 1;
      while (stop * 2 < Nb) stop *= 2;
      if (Nb > 1) cr_backward(D, L, f, Nb, stop, 1);
    }
    if (!ok) l.ired[0] = 0;   // some lane met a pivot <= 0 (every lane of its 8-lane group did)
    __syncthreads();
    CRP(4);
    for (int q = tid; q < Nt; q += kThreads) l.dxv[q] = f[q];
    __syncthreads();
    CRP(5);
    return;
  }
  // forward reduction in place (8 lanes per elimination, 32 eliminations per round; TEB_CR_FORWARD = cr_forward, or the matrix-core variant)
  bool ok = TEB_CR_FORWARD(D, L, f, Nb, 1, Nb);
  // the last surviving block row: x_0 = D_0^{-1} f_0
  if (tid < 8) {
    double v[8];
    ld_row<8>(f, v);
    Ldl8 F;
    F.load(D);
    ok = F.factor() && ok;
    F.solve(v);
    double mine = v[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) mine = (tid == k) ? v[k] : mine;
    f[tid] = mine;
  }
  if (!ok) l.ired[0] = 0;
  __syncthreads();
  CRP(4);
  // back substitution, coarsest level first
  int stop = 1;
  while (stop * 2 < Nb) stop *= 2;
  for (int s = stop; s >= 1; s >>= 1) {
    const int E = (Nb - 1 - s) / (2 * s) + 1;
    for (int u = tid; u < E * 8; u += kThreads) {
      const int e = u >> 3, r = u & 7;
      const int i = s * (2 * e + 1);
      double acc = f[i * 8 + r], acc2 = 0;
      double WL[8], xm[8];
      ld_row<8>(D + i * kBlk + r * 8, WL);
      ld_row<8>(f + (i - s) * 8, xm);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc -= WL[k] * xm[k];
      if (i + s < Nb) {
        double WU[8], xp[8];
        ld_row<8>(L + i * kBlk + r * 8, WU);
        ld_row<8>(f + (i + s) * 8, xp);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc2 -= WU[k] * xp[k];
      }
      f[i * 8 + r] = acc + acc2;
    }
    __syncthreads();
  }
  for (int q = tid; q < Nt; q += kThreads) l.dxv[q] = f[q];
  __syncthreads();
  CRP(5);
}

template <bool GLOBAL, bool HB_GLOBAL>
__device__ TEB_SOLVE_LINKAGE void cr_solve_t(const LdsPlan plan, const SceneDev& sc, int n, double lambda, double* gbuf, double* gH) {
  cr_solve_t_impl<GLOBAL, HB_GLOBAL>(plan, sc, n, lambda, gbuf, gH);
}
// the copy the solver helpers of a small batch call (plain calling convention: see cr_solve_hybrid_helper below)
#ifdef TEB_AMD_INLINE_SOLVE
#define TEB_HELPER_SOLVE_LINKAGE_T __forceinline__
#else
#define TEB_HELPER_SOLVE_LINKAGE_T __noinline__
#endif
__device__ TEB_HELPER_SOLVE_LINKAGE_T void cr_solve_blocks_helper(const LdsPlan plan, const SceneDev& sc, int n, double lambda) {
  cr_solve_t_impl<false, false>(plan, sc, n, lambda, nullptr, nullptr);
}

// ---- damped solve (K6 v4, "hybrid"): for bands whose block layout does not fit the LDS (SOLVER_BAND) ------------------------------------
// The normal matrix is linearised into the LDS band. ONCE per LM iteration the band is copied as it is into the band's HBM scratch
// (cr_copy_band: coalesced, read-only from then on, lambda-free, L2-resident); the band region of the LDS is dead until the next
// linearisation. Every damped trial then runs
//   level 0: the odd block rows are eliminated straight from that copy (only the structurally non-zero entries are gathered, + lambda
//            on the fly; loads only, no read-modify-write on HBM) INTO a compact system of the even rows built in the former band
//            region of the LDS (35 S <= 44 S doubles), their (W_L, W_U, P f) records stay in registers;
//   levels >= 1, top solve, back substitution of the even rows: the in-LDS block cyclic reduction (cr_forward / cr_top / cr_backward);
//   back substitution of the odd rows from those records.
// Same arithmetic as cr_solve_t (an exact re-indexing: compact row j' = row 2 j'); no backup / restore of H, no obstacle-cache reload.
// gbuf: the band copy, [8 Nb][kBand]
// the band as it stands in LDS -> the band's HBM scratch, rows [0, 8 Nb): a coalesced copy (101 KB at 287 poses; 32 bands per XCD stay
// inside the 4 MB L2); the padding rows of an odd pose count become identity rows
// shared: solver helpers on other CUs read the copy as well (speculative trials, teb_multicu.hpp): written through, agent scope
__device__ __forceinline__ void cr_copy_band(const Lds& l, int n, double* __restrict__ gband, bool shared = false) {
  const int tid = threadIdx.x;
  const int Nt = 4 * n, Nb = (Nt + 7) >> 3;
  const double* Hb = l.Hb;
  const int live = hbo(Nt);   // rows [0, Nt) incl. their padding doubles: copied as they stand
  if (shared) {   // (8-byte write-through stores; 16-byte ones through inline asm measured 7 - 10 % slower end to end on C2 / C3 / C5)
    for (int q = tid; q < live; q += kThreads) st_agent_f64(gband + q, Hb[q]);
    for (int q = tid; q < (Nb * 8 - Nt) * kBand; q += kThreads) {   // the padding rows [Nt, 8 Nb) of an odd pose count become identity rows
      const int r = Nt + q / kBand, d = q - (r - Nt) * kBand;
      st_agent_f64(gband + hbo(r) + d, d == 0 ? 1.0 : 0.0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    // a linear copy, two doubles per access (the band starts on a 16-byte boundary in LDS and in the scratch)
    typedef double __attribute__((address_space(1))) gdouble_t;
    typedef teb_v2d __attribute__((address_space(1))) gv2d_t;
    for (int q = 2 * tid; q + 1 < live; q += 2 * kThreads) *reinterpret_cast<gv2d_t*>((gdouble_t*)gband + q) = *reinterpret_cast<const teb_v2d*>(Hb + q);
    if ((live & 1) && tid == 0) gband[live - 1] = Hb[live - 1];
    for (int q = tid; q < (Nb * 8 - Nt) * kBand; q += kThreads) {
      const int r = Nt + q / kBand, d = q - (r - Nt) * kBand;
      gband[hbo(r) + d] = d == 0 ? 1.0 : 0.0;
    }
  }
  __threadfence_block();
  __syncthreads();
}

constexpr int kHybridRounds = 2;   // level-0 rounds of 32 eliminations: at most 2 (kThreads / 8) = 64 odd rows are eliminated there (E0 below)
// Two out-of-line copies of the solve: the band's own (no callee-saved block, above) and the one the SOLVER HELPERS of a small batch
// call. The helper's copy keeps the plain calling convention. Reason: small-batch kernels with BOTH call sites on the no-callee-saved
// path failed on MI355X in two builds of this round (a profiling build and the band-layout small-batch kernel specialised on the
// TebConfig defaults: out-of-bounds global writes or a hang as soon as the helpers solved; the same sources with either call on the
// plain convention ran clean and bit-identical, as did every kernel with a single such call) while the register masks at both call
// sites check out in the ISA - an interaction inside the backend that this code does not depend on any more. The helper's solve is off
// the band's critical path, so its save / restore costs the band nothing.
#ifdef TEB_AMD_INLINE_SOLVE
#define TEB_HELPER_SOLVE_LINKAGE __forceinline__
#else
#define TEB_HELPER_SOLVE_LINKAGE __noinline__
#endif
// Level 0 of the hybrid solve, product L_i^T W_L by DPP: a 16-lane DPP row holds TWO 8-lane groups, so a term is two instructions - lane aa
// of the row for the lanes of the lower group (banks 0, 1), lane 8 + aa for the upper one (banks 2, 3); the other half keeps its
// accumulator. o1[aa] += cl[k](lane aa of the group) * wL[k], structural zeros (k > aa + 2) skipped, every o1[aa] receiving its terms in
// ascending k: the same products in the same order as the exchange through LDS rows they replace (49 terms, 98 instructions against 49 +
// 32 ds_read_b128 on a pipe that was the bottleneck of this phase: a diagnostic build without the reads ran the solve 3.8 k cycles
// shorter). ORDER MATTERS beyond the sums: on gfx950 a bank-masked v_fmac_f64_dpp that follows IMMEDIATELY on an instruction writing the
// same accumulator loses that write in its masked-off lanes and reads a stale accumulator in the others (measured,
// tools/micro/dpp64_mask_probe.hip: any instruction in between, even s_nop 0, and it is correct; full masks - the reduction rounds -
// chain correctly). So the statement runs k outermost: the 3 .. 8 accumulators of a k for the lower groups, then for the upper groups -
// no two consecutive instructions share an accumulator. tests/test_gpu_cr_rounds.py runs tools/micro/l0_dpp_probe (this statement
// against the same sums through __shfl, bit for bit).
// operands: %0 .. %7 = o1[0 .. 7], %8 .. %15 = cl[0 .. 7], %16 .. %23 = wL[0 .. 7]
#define TEB_L0LO(t, a, x, y) "v_fmac_f64_dpp %" #a ", %" #x ", %" #y " row_newbcast:" #t " row_mask:0xf bank_mask:0x3\n\t"
#define TEB_L0HI(t, a, x, y) "v_fmac_f64_dpp %" #a ", %" #x ", %" #y " row_newbcast:" TEB_L0_HI_##t " row_mask:0xf bank_mask:0xc\n\t"
#define TEB_L0_HI_0 "8"
#define TEB_L0_HI_1 "9"
#define TEB_L0_HI_2 "10"
#define TEB_L0_HI_3 "11"
#define TEB_L0_HI_4 "12"
#define TEB_L0_HI_5 "13"
#define TEB_L0_HI_6 "14"
#define TEB_L0_HI_7 "15"
#define TEB_L0_O1_DPP \
 TEB_L0LO(0, 0, 8, 16) TEB_L0LO(1, 1, 8, 16) TEB_L0LO(2, 2, 8, 16) TEB_L0LO(3, 3, 8, 16) TEB_L0LO(4, 4, 8, 16) TEB_L0LO(5, 5, 8, 16) TEB_L0LO(6, 6, 8, 16) TEB_L0LO(7, 7, 8, 16) \
 TEB_L0HI(0, 0, 8, 16) TEB_L0HI(1, 1, 8, 16) TEB_L0HI(2, 2, 8, 16) TEB_L0HI(3, 3, 8, 16) TEB_L0HI(4, 4, 8, 16) TEB_L0HI(5, 5, 8, 16) TEB_L0HI(6, 6, 8, 16) TEB_L0HI(7, 7, 8, 16) \
 TEB_L0LO(0, 0, 9, 17) TEB_L0LO(1, 1, 9, 17) TEB_L0LO(2, 2, 9, 17) TEB_L0LO(3, 3, 9, 17) TEB_L0LO(4, 4, 9, 17) TEB_L0LO(5, 5, 9, 17) TEB_L0LO(6, 6, 9, 17) TEB_L0LO(7, 7, 9, 17) \
 TEB_L0HI(0, 0, 9, 17) TEB_L0HI(1, 1, 9, 17) TEB_L0HI(2, 2, 9, 17) TEB_L0HI(3, 3, 9, 17) TEB_L0HI(4, 4, 9, 17) TEB_L0HI(5, 5, 9, 17) TEB_L0HI(6, 6, 9, 17) TEB_L0HI(7, 7, 9, 17) \
 TEB_L0LO(0, 0, 10, 18) TEB_L0LO(1, 1, 10, 18) TEB_L0LO(2, 2, 10, 18) TEB_L0LO(3, 3, 10, 18) TEB_L0LO(4, 4, 10, 18) TEB_L0LO(5, 5, 10, 18) TEB_L0LO(6, 6, 10, 18) TEB_L0LO(7, 7, 10, 18) \
 TEB_L0HI(0, 0, 10, 18) TEB_L0HI(1, 1, 10, 18) TEB_L0HI(2, 2, 10, 18) TEB_L0HI(3, 3, 10, 18) TEB_L0HI(4, 4, 10, 18) TEB_L0HI(5, 5, 10, 18) TEB_L0HI(6, 6, 10, 18) TEB_L0HI(7, 7, 10, 18) \
 TEB_L0LO(1, 1, 11, 19) TEB_L0LO(2, 2, 11, 19) TEB_L0LO(3, 3, 11, 19) TEB_L0LO(4, 4, 11, 19) TEB_L0LO(5, 5, 11, 19) TEB_L0LO(6, 6, 11, 19) TEB_L0LO(7, 7, 11, 19) \
 TEB_L0HI(1, 1, 11, 19) TEB_L0HI(2, 2, 11, 19) TEB_L0HI(3, 3, 11, 19) TEB_L0HI(4, 4, 11, 19) TEB_L0HI(5, 5, 11, 19) TEB_L0HI(6, 6, 11, 19) TEB_L0HI(7, 7, 11, 19) \
 TEB_L0LO(2, 2, 12, 20) TEB_L0LO(3, 3, 12, 20) TEB_L0LO(4, 4, 12, 20) TEB_L0LO(5, 5, 12, 20) TEB_L0LO(6, 6, 12, 20) TEB_L0LO(7, 7, 12, 20) \
 TEB_L0HI(2, 2, 12, 20) TEB_L0HI(3, 3, 12, 20) TEB_L0HI(4, 4, 12, 20) TEB_L0HI(5, 5, 12, 20) TEB_L0HI(6, 6, 12, 20) TEB_L0HI(7, 7, 12, 20) \
 TEB_L0LO(3, 3, 13, 21) TEB_L0LO(4, 4, 13, 21) TEB_L0LO(5, 5, 13, 21) TEB_L0LO(6, 6, 13, 21) TEB_L0LO(7, 7, 13, 21) \
 TEB_L0HI(3, 3, 13, 21) TEB_L0HI(4, 4, 13, 21) TEB_L0HI(5, 5, 13, 21) TEB_L0HI(6, 6, 13, 21) TEB_L0HI(7, 7, 13, 21) \
 TEB_L0LO(4, 4, 14, 22) TEB_L0LO(5, 5, 14, 22) TEB_L0LO(6, 6, 14, 22) TEB_L0LO(7, 7, 14, 22) \
 TEB_L0HI(4, 4, 14, 22) TEB_L0HI(5, 5, 14, 22) TEB_L0HI(6, 6, 14, 22) TEB_L0HI(7, 7, 14, 22) \
 TEB_L0LO(5, 5, 15, 23) TEB_L0LO(6, 6, 15, 23) TEB_L0LO(7, 7, 15, 23) \
 TEB_L0HI(5, 5, 15, 23) TEB_L0HI(6, 6, 15, 23) TEB_L0HI(7, 7, 15, 23)
// The products with the rows of L_{i+1} the same way (terms k >= aa - 2): o2[aa] -= cu[k](lane aa) * wL[k] (operands o2, cu, wL) and
// o3[aa] += cu[k](lane aa) * wU[k] (operands o3, cu, wU), two statements (operand limit).
#define TEB_L0LON(t, a, x, y) "v_fmac_f64_dpp %" #a ", -%" #x ", %" #y " row_newbcast:" #t " row_mask:0xf bank_mask:0x3\n\t"
#define TEB_L0HIN(t, a, x, y) "v_fmac_f64_dpp %" #a ", -%" #x ", %" #y " row_newbcast:" TEB_L0_HI_##t " row_mask:0xf bank_mask:0xc\n\t"
#define TEB_L0_O2_DPP \
 TEB_L0LON(0, 0, 8, 16) TEB_L0LON(1, 1, 8, 16) TEB_L0LON(2, 2, 8, 16) \
 TEB_L0HIN(0, 0, 8, 16) TEB_L0HIN(1, 1, 8, 16) TEB_L0HIN(2, 2, 8, 16) \
 TEB_L0LON(0, 0, 9, 17) TEB_L0LON(1, 1, 9, 17) TEB_L0LON(2, 2, 9, 17) TEB_L0LON(3, 3, 9, 17) \
 TEB_L0HIN(0, 0, 9, 17) TEB_L0HIN(1, 1, 9, 17) TEB_L0HIN(2, 2, 9, 17) TEB_L0HIN(3, 3, 9, 17) \
 TEB_L0LON(0, 0, 10, 18) TEB_L0LON(1, 1, 10, 18) TEB_L0LON(2, 2, 10, 18) TEB_L0LON(3, 3, 10, 18) TEB_L0LON(4, 4, 10, 18) \
 TEB_L0HIN(0, 0, 10, 18) TEB_L0HIN(1, 1, 10, 18) TEB_L0HIN(2, 2, 10, 18) TEB_L0HIN(3, 3, 10, 18) TEB_L0HIN(4, 4, 10, 18) \
 TEB_L0LON(0, 0, 11, 19) TEB_L0LON(1, 1, 11, 19) TEB_L0LON(2, 2, 11, 19) TEB_L0LON(3, 3, 11, 19) TEB_L0LON(4, 4, 11, 19) TEB_L0LON(5, 5, 11, 19) \
 TEB_L0HIN(0, 0, 11, 19) TEB_L0HIN(1, 1, 11, 19) TEB_L0HIN(2, 2, 11, 19) TEB_L0HIN(3, 3, 11, 19) TEB_L0HIN(4, 4, 11, 19) TEB_L0HIN(5, 5, 11, 19) \
 TEB_L0LON(0, 0, 12, 20) TEB_L0LON(1, 1, 12, 20) TEB_L0LON(2, 2, 12, 20) TEB_L0LON(3, 3, 12, 20) TEB_L0LON(4, 4, 12, 20) TEB_L0LON(5, 5, 12, 20) TEB_L0LON(6, 6, 12, 20) \
 TEB_L0HIN(0, 0, 12, 20) TEB_L0HIN(1, 1, 12, 20) TEB_L0HIN(2, 2, 12, 20) TEB_L0HIN(3, 3, 12, 20) TEB_L0HIN(4, 4, 12, 20) TEB_L0HIN(5, 5, 12, 20) TEB_L0HIN(6, 6, 12, 20) \
 TEB_L0LON(0, 0, 13, 21) TEB_L0LON(1, 1, 13, 21) TEB_L0LON(2, 2, 13, 21) TEB_L0LON(3, 3, 13, 21) TEB_L0LON(4, 4, 13, 21) TEB_L0LON(5, 5, 13, 21) TEB_L0LON(6, 6, 13, 21) TEB_L0LON(7, 7, 13, 21) \
 TEB_L0HIN(0, 0, 13, 21) TEB_L0HIN(1, 1, 13, 21) TEB_L0HIN(2, 2, 13, 21) TEB_L0HIN(3, 3, 13, 21) TEB_L0HIN(4, 4, 13, 21) TEB_L0HIN(5, 5, 13, 21) TEB_L0HIN(6, 6, 13, 21) TEB_L0HIN(7, 7, 13, 21) \
 TEB_L0LON(0, 0, 14, 22) TEB_L0LON(1, 1, 14, 22) TEB_L0LON(2, 2, 14, 22) TEB_L0LON(3, 3, 14, 22) TEB_L0LON(4, 4, 14, 22) TEB_L0LON(5, 5, 14, 22) TEB_L0LON(6, 6, 14, 22) TEB_L0LON(7, 7, 14, 22) \
 TEB_L0HIN(0, 0, 14, 22) TEB_L0HIN(1, 1, 14, 22) TEB_L0HIN(2, 2, 14, 22) TEB_L0HIN(3, 3, 14, 22) TEB_L0HIN(4, 4, 14, 22) TEB_L0HIN(5, 5, 14, 22) TEB_L0HIN(6, 6, 14, 22) TEB_L0HIN(7, 7, 14, 22) \
 TEB_L0LON(0, 0, 15, 23) TEB_L0LON(1, 1, 15, 23) TEB_L0LON(2, 2, 15, 23) TEB_L0LON(3, 3, 15, 23) TEB_L0LON(4, 4, 15, 23) TEB_L0LON(5, 5, 15, 23) TEB_L0LON(6, 6, 15, 23) TEB_L0LON(7, 7, 15, 23) \
 TEB_L0HIN(0, 0, 15, 23) TEB_L0HIN(1, 1, 15, 23) TEB_L0HIN(2, 2, 15, 23) TEB_L0HIN(3, 3, 15, 23) TEB_L0HIN(4, 4, 15, 23) TEB_L0HIN(5, 5, 15, 23) TEB_L0HIN(6, 6, 15, 23) TEB_L0HIN(7, 7, 15, 23)
#define TEB_L0_O3_DPP \
 TEB_L0LO(0, 0, 8, 16) TEB_L0LO(1, 1, 8, 16) TEB_L0LO(2, 2, 8, 16) \
 TEB_L0HI(0, 0, 8, 16) TEB_L0HI(1, 1, 8, 16) TEB_L0HI(2, 2, 8, 16) \
 TEB_L0LO(0, 0, 9, 17) TEB_L0LO(1, 1, 9, 17) TEB_L0LO(2, 2, 9, 17) TEB_L0LO(3, 3, 9, 17) \
 TEB_L0HI(0, 0, 9, 17) TEB_L0HI(1, 1, 9, 17) TEB_L0HI(2, 2, 9, 17) TEB_L0HI(3, 3, 9, 17) \
 TEB_L0LO(0, 0, 10, 18) TEB_L0LO(1, 1, 10, 18) TEB_L0LO(2, 2, 10, 18) TEB_L0LO(3, 3, 10, 18) TEB_L0LO(4, 4, 10, 18) \
 TEB_L0HI(0, 0, 10, 18) TEB_L0HI(1, 1, 10, 18) TEB_L0HI(2, 2, 10, 18) TEB_L0HI(3, 3, 10, 18) TEB_L0HI(4, 4, 10, 18) \
 TEB_L0LO(0, 0, 11, 19) TEB_L0LO(1, 1, 11, 19) TEB_L0LO(2, 2, 11, 19) TEB_L0LO(3, 3, 11, 19) TEB_L0LO(4, 4, 11, 19) TEB_L0LO(5, 5, 11, 19) \
 TEB_L0HI(0, 0, 11, 19) TEB_L0HI(1, 1, 11, 19) TEB_L0HI(2, 2, 11, 19) TEB_L0HI(3, 3, 11, 19) TEB_L0HI(4, 4, 11, 19) TEB_L0HI(5, 5, 11, 19) \
 TEB_L0LO(0, 0, 12, 20) TEB_L0LO(1, 1, 12, 20) TEB_L0LO(2, 2, 12, 20) TEB_L0LO(3, 3, 12, 20) TEB_L0LO(4, 4, 12, 20) TEB_L0LO(5, 5, 12, 20) TEB_L0LO(6, 6, 12, 20) \
 TEB_L0HI(0, 0, 12, 20) TEB_L0HI(1, 1, 12, 20) TEB_L0HI(2, 2, 12, 20) TEB_L0HI(3, 3, 12, 20) TEB_L0HI(4, 4, 12, 20) TEB_L0HI(5, 5, 12, 20) TEB_L0HI(6, 6, 12, 20) \
 TEB_L0LO(0, 0, 13, 21) TEB_L0LO(1, 1, 13, 21) TEB_L0LO(2, 2, 13, 21) TEB_L0LO(3, 3, 13, 21) TEB_L0LO(4, 4, 13, 21) TEB_L0LO(5, 5, 13, 21) TEB_L0LO(6, 6, 13, 21) TEB_L0LO(7, 7, 13, 21) \
 TEB_L0HI(0, 0, 13, 21) TEB_L0HI(1, 1, 13, 21) TEB_L0HI(2, 2, 13, 21) TEB_L0HI(3, 3, 13, 21) TEB_L0HI(4, 4, 13, 21) TEB_L0HI(5, 5, 13, 21) TEB_L0HI(6, 6, 13, 21) TEB_L0HI(7, 7, 13, 21) \
 TEB_L0LO(0, 0, 14, 22) TEB_L0LO(1, 1, 14, 22) TEB_L0LO(2, 2, 14, 22) TEB_L0LO(3, 3, 14, 22) TEB_L0LO(4, 4, 14, 22) TEB_L0LO(5, 5, 14, 22) TEB_L0LO(6, 6, 14, 22) TEB_L0LO(7, 7, 14, 22) \
 TEB_L0HI(0, 0, 14, 22) TEB_L0HI(1, 1, 14, 22) TEB_L0HI(2, 2, 14, 22) TEB_L0HI(3, 3, 14, 22) TEB_L0HI(4, 4, 14, 22) TEB_L0HI(5, 5, 14, 22) TEB_L0HI(6, 6, 14, 22) TEB_L0HI(7, 7, 14, 22) \
 TEB_L0LO(0, 0, 15, 23) TEB_L0LO(1, 1, 15, 23) TEB_L0LO(2, 2, 15, 23) TEB_L0LO(3, 3, 15, 23) TEB_L0LO(4, 4, 15, 23) TEB_L0LO(5, 5, 15, 23) TEB_L0LO(6, 6, 15, 23) TEB_L0LO(7, 7, 15, 23) \
 TEB_L0HI(0, 0, 15, 23) TEB_L0HI(1, 1, 15, 23) TEB_L0HI(2, 2, 15, 23) TEB_L0HI(3, 3, 15, 23) TEB_L0HI(4, 4, 15, 23) TEB_L0HI(5, 5, 15, 23) TEB_L0HI(6, 6, 15, 23) TEB_L0HI(7, 7, 15, 23)
template <int WHO> __device__ void cr_solve_hybrid_impl(const LdsPlan plan, int n, double lambda, double* gbuf);
__device__ TEB_SOLVE_LINKAGE void cr_solve_hybrid(const LdsPlan plan, int n, double lambda, double* gbuf) { cr_solve_hybrid_impl<0>(plan, n, lambda, gbuf); }
__device__ TEB_HELPER_SOLVE_LINKAGE void cr_solve_hybrid_helper(const LdsPlan plan, int n, double lambda, double* gbuf) { cr_solve_hybrid_impl<1>(plan, n, lambda, gbuf); }
template <int WHO> __device__ __forceinline__ void cr_solve_hybrid_impl(const LdsPlan plan_, int n_, double lambda_, double* gbuf_) {
  TEB_SOLVER_FMA
  extern __shared__ __attribute__((aligned(16))) double lds_base[];
  CRP_DECL
#ifdef TEB_AMD_INLINE_SOLVE
  const LdsPlan plan = plan_; const int n = n_; const double lambda = lambda_; double* gbuf = gbuf_;
#else
  const LdsPlan plan = uni_plan(plan_); const int n = uni_i(n_); const double lambda = uni_d(lambda_); double* gbuf = uni_p(gbuf_);
#endif
  const Lds l = carve(lds_base, plan);
  const int tid = threadIdx.x;
  const int Nt = 4 * n, Nb = (Nt + 7) >> 3, E = Nb >> 1;
  // Level 0 takes up to two rounds of 32 eliminations: 64 odd rows, a band of 256 poses. A longer band has up to 20 odd rows beyond them
  // (337 poses); a third round for those would keep most lane groups idle, and the bands that need it are the ones the launch waits
  // for. Those rows are not eliminated at level 0 instead: the block rows from 2 E0 on enter the compact system as they are (odd and
  // even, coupled by their original L blocks) and are reduced by its levels, which need no extra round for them. Compact row j is
  // block row 2 j for j <= E0 and block row j + E0 beyond. The larger compact system always fits the band region of a launch that holds
  // the band ((Nb - 64) (2 kBlk + 8) + 14 <= 45 (2 Nb - 1) <= hbo(4 plan.S) for every Nb <= 178; the layout ends at 337 poses, Nb = 169); a
  // launch for which it did not would be refused here, not overrun.
  constexpr int kLevel0Max = kHybridRounds * (kThreads / 8);
  if (E > kLevel0Max && (size_t)(Nb - kLevel0Max) * (2 * kBlk + 8) + 14 > (size_t)hbo(4 * plan.S)) {
    if (tid == 0) l.ired[0] = 0;   // reported like a failed factorisation
    __syncthreads();
    return;
  }
  const int E0 = E > kLevel0Max ? kLevel0Max : E;
  const int Nc = Nb - E0;   // (= the even rows alone, (Nb + 1) / 2, when every odd row is eliminated at level 0)
#define TEB_HYB_ROW(j) ((j) <= E0 ? 2 * (j) : (j) + E0)
  // Hg: the band copy, entry (r, c), c <= r <= c + 10, at Hg[r * 11 + (r - c)]. In terms of 8x8 blocks:
  //   D_j[a][b] (b <= a)        = Hg[(8 j + a) * 11 + (a - b)]
  //   L_j[a][b] = H(8j+a, 8(j-1)+b) = Hg[(8 j + a) * 11 + (8 + a - b)]   for b >= a - 2, structurally zero otherwise (49 of 64 entries)
  // (global address space spelled out: a generic pointer would be read with flat_load, which also counts on the LDS counter)
  typedef const double __attribute__((address_space(1))) gdouble_t;
  gdouble_t* __restrict__ Hg = (gdouble_t*)gbuf;
  // The L blocks start 64 (mod 128) bytes behind the D blocks: the slot writes of a round (lanes q = 0 into D_i, q = 1 into L_i, one
  // 16-lane group of a ds_write_b64, banks = 128-byte window) then fall into different halves of the window. kBlk * 8 = 16 (mod 128).
  const int padL = 2 * ((4 - (Nc & 7)) & 7);
  double* __restrict__ Dc = lds_base + plan.off_H;
  double* __restrict__ Lc = Dc + Nc * kBlk + padL;
  double* __restrict__ fc = Lc + Nc * kBlk;
  // compact system = the even block rows (+ lambda); their couplings are written by the eliminations (row 0 has none and is never
  // read). The loads of a batch are issued together: one L2 round trip per batch instead of one per element.
  // Lane (a8, b8) = entry of the 8x8 block, the same for every block this lane fills (kThreads = 4 x 64): what changes from block to
  // block is the row base alone, and hbo(8 R + hi) = 90 R + hbo(hi) is linear in the block row R - two integer instructions per element
  // where the general index arithmetic (q -> block, entry, band row, band offset) took nine.
  constexpr int kInitBatch = 8;
  static_assert(kThreads % 64 == 0 && hbo(8) == 90, "the lanes of a workgroup cover whole 8x8 blocks; hbo(8 R + h) = 90 R + hbo(h)");
  {
    const int w = tid & 63, a8 = w >> 3, b8 = w & 7;
    const int hi = a8 > b8 ? a8 : b8, lo = a8 > b8 ? b8 : a8;
    const int koff = hbo(hi) + (hi - lo);
    const bool on_diag = a8 == b8;
    for (int j0 = tid >> 6; j0 < Nc; j0 += (kThreads / 64) * kInitBatch) {
      double v[kInitBatch];
#pragma unroll
      for (int u = 0; u < kInitBatch; ++u) {
        const int j = j0 + u * (kThreads / 64);
        const int jj = j < Nc ? j : Nc - 1;   // (a valid block: the value is dropped below)
        v[u] = Hg[90 * TEB_HYB_ROW(jj) + koff];
      }
#pragma unroll
      for (int u = 0; u < kInitBatch; ++u) {
        const int j = j0 + u * (kThreads / 64);
        if (j < Nc) Dc[j * kBlk + w] = on_diag ? v[u] + lambda : v[u];
      }
    }
  }
  for (int q = tid; q < Nc * 8; q += kThreads) {
    const int src = 8 * TEB_HYB_ROW(q >> 3) + (q & 7);
    fc[q] = src < Nt ? l.bv[src] : 0.0;
  }
  // the couplings of the rows that skip level 0 (compact rows E0 + 1 ..): their original L blocks
  for (int q = (E0 + 1) * 64 + tid; q < Nc * 64; q += kThreads) {
    const int j = q >> 6, a8 = (q >> 3) & 7, b8 = q & 7;
    Lc[j * kBlk + (q & 63)] = b8 >= a8 - 2 ? Hg[hbo(8 * (j + E0) + a8) + (8 + a8 - b8)] : 0.0;
  }
  if (tid == 0) l.ired[0] = 1;
  __syncthreads();
  CRP(0);
  // level 0: 8 lanes per elimination of an odd row i = 2 e + 1 (lane c owns column c of L_i, of U_i = L_{i+1}^T and, redundantly, f_i).
  // The records W_L = P L_i, W_U = P U_i, P f_i of the eliminated rows stay in the registers of the lanes that computed them (column c
  // each) until the back substitution at the end: nothing but the read-only band copy crosses the LDS boundary during a solve.
  const int grp = tid >> 3, c = tid & 7;
  bool ok = true;
  double kL[kHybridRounds][8], kU[kHybridRounds][8];
#pragma unroll
  for (int rr = 0; rr < kHybridRounds; ++rr) {
    const int e = rr * (kThreads / 8) + grp;
    const bool act = e < E0;
    const int i = 2 * e + 1;
    const bool hasU = act && (i + 1 < Nb);
    double wL[8], wU[8], wf[8], o1[8], o2[8], o3[8];
    double s1 = 0, s2 = 0;
    if (rr * (kThreads / 8) < E0) {   // (uniform) this round has eliminations at all
      if (act) {
        gdouble_t* Hi = Hg + hbo(8 * i);         // band rows of block row i (8 i is a multiple of 4: row r of the block starts at Hi + hbo(r))
        gdouble_t* Hp = Hg + hbo(8 * (i + 1));   // ... of block row i + 1 (valid iff hasU)
        Ldl8 F;
        // Every lane of the group needs all 36 entries of D_i's lower triangle (it factors the block in its own registers). Fetched by
        // every lane that was 36 loads per lane of the same 36 values, at ~ 37 cycles per load instruction on the path the four waves
        // share (a diagnostic build with 8 ran the solve 2.1 k cycles shorter). Lane r of the group fetches row r instead (8 loads; what
        // lies left of its diagonal start is inside the band's scratch and dropped) and the rows go round by DPP: lane r of the DPP row for
        // the lower group, lane 8 + r for the upper one, bank-masked - all the lower halves first, then all the upper ones, so that no
        // instruction follows directly on one that wrote its destination (TEB_L0_O1_DPP's note on the masked back-to-back hazard).
        // Round 6: - 2.3 k cycles per solve, bit-identical (the same values reach the same registers).
        {
          double row[8];
#pragma unroll
          for (int cc = 0; cc < 8; ++cc) row[cc] = Hi[hbo(c) + (c - cc)];   // (16-byte loads of these descending runs measured the same: it is bytes per lane, not instructions)
#define TEB_L0BLO(t, d, x) "v_mov_b64_dpp %" #d ", %" #x " row_newbcast:" #t " row_mask:0xf bank_mask:0x3\n\t"
#define TEB_L0BHI(t, d, x) "v_mov_b64_dpp %" #d ", %" #x " row_newbcast:" TEB_L0_HI_##t " row_mask:0xf bank_mask:0xc\n\t"
        asm("s_nop 1\n\t" TEB_L0BLO(0, 0, 15) TEB_L0BLO(1, 1, 15) TEB_L0BLO(1, 2, 16) TEB_L0BLO(2, 3, 15) TEB_L0BLO(2, 4, 16) TEB_L0BLO(2, 5, 17) TEB_L0BLO(3, 6, 15) TEB_L0BLO(3, 7, 16) TEB_L0BLO(3, 8, 17) TEB_L0BLO(3, 9, 18) TEB_L0BLO(4, 10, 15) TEB_L0BLO(4, 11, 16) TEB_L0BLO(4, 12, 17) TEB_L0BLO(4, 13, 18) TEB_L0BLO(4, 14, 19)
            TEB_L0BHI(0, 0, 15) TEB_L0BHI(1, 1, 15) TEB_L0BHI(1, 2, 16) TEB_L0BHI(2, 3, 15) TEB_L0BHI(2, 4, 16) TEB_L0BHI(2, 5, 17) TEB_L0BHI(3, 6, 15) TEB_L0BHI(3, 7, 16) TEB_L0BHI(3, 8, 17) TEB_L0BHI(3, 9, 18) TEB_L0BHI(4, 10, 15) TEB_L0BHI(4, 11, 16) TEB_L0BHI(4, 12, 17) TEB_L0BHI(4, 13, 18) TEB_L0BHI(4, 14, 19)
            : "=&v"(F.a[Ldl8::idx(0, 0)]), "=&v"(F.a[Ldl8::idx(1, 0)]), "=&v"(F.a[Ldl8::idx(1, 1)]), "=&v"(F.a[Ldl8::idx(2, 0)]), "=&v"(F.a[Ldl8::idx(2, 1)]), "=&v"(F.a[Ldl8::idx(2, 2)]), "=&v"(F.a[Ldl8::idx(3, 0)]), "=&v"(F.a[Ldl8::idx(3, 1)]), "=&v"(F.a[Ldl8::idx(3, 2)]), "=&v"(F.a[Ldl8::idx(3, 3)]), "=&v"(F.a[Ldl8::idx(4, 0)]), "=&v"(F.a[Ldl8::idx(4, 1)]), "=&v"(F.a[Ldl8::idx(4, 2)]), "=&v"(F.a[Ldl8::idx(4, 3)]), "=&v"(F.a[Ldl8::idx(4, 4)])
            : "v"(row[0]), "v"(row[1]), "v"(row[2]), "v"(row[3]), "v"(row[4]));
        asm("s_nop 1\n\t" TEB_L0BLO(5, 0, 13) TEB_L0BLO(5, 1, 14) TEB_L0BLO(5, 2, 15) TEB_L0BLO(5, 3, 16) TEB_L0BLO(5, 4, 17) TEB_L0BLO(5, 5, 18) TEB_L0BLO(6, 6, 13) TEB_L0BLO(6, 7, 14) TEB_L0BLO(6, 8, 15) TEB_L0BLO(6, 9, 16) TEB_L0BLO(6, 10, 17) TEB_L0BLO(6, 11, 18) TEB_L0BLO(6, 12, 19)
            TEB_L0BHI(5, 0, 13) TEB_L0BHI(5, 1, 14) TEB_L0BHI(5, 2, 15) TEB_L0BHI(5, 3, 16) TEB_L0BHI(5, 4, 17) TEB_L0BHI(5, 5, 18) TEB_L0BHI(6, 6, 13) TEB_L0BHI(6, 7, 14) TEB_L0BHI(6, 8, 15) TEB_L0BHI(6, 9, 16) TEB_L0BHI(6, 10, 17) TEB_L0BHI(6, 11, 18) TEB_L0BHI(6, 12, 19)
            : "=&v"(F.a[Ldl8::idx(5, 0)]), "=&v"(F.a[Ldl8::idx(5, 1)]), "=&v"(F.a[Ldl8::idx(5, 2)]), "=&v"(F.a[Ldl8::idx(5, 3)]), "=&v"(F.a[Ldl8::idx(5, 4)]), "=&v"(F.a[Ldl8::idx(5, 5)]), "=&v"(F.a[Ldl8::idx(6, 0)]), "=&v"(F.a[Ldl8::idx(6, 1)]), "=&v"(F.a[Ldl8::idx(6, 2)]), "=&v"(F.a[Ldl8::idx(6, 3)]), "=&v"(F.a[Ldl8::idx(6, 4)]), "=&v"(F.a[Ldl8::idx(6, 5)]), "=&v"(F.a[Ldl8::idx(6, 6)])
            : "v"(row[0]), "v"(row[1]), "v"(row[2]), "v"(row[3]), "v"(row[4]), "v"(row[5]), "v"(row[6]));
        asm("s_nop 1\n\t" TEB_L0BLO(7, 0, 8) TEB_L0BLO(7, 1, 9) TEB_L0BLO(7, 2, 10) TEB_L0BLO(7, 3, 11) TEB_L0BLO(7, 4, 12) TEB_L0BLO(7, 5, 13) TEB_L0BLO(7, 6, 14) TEB_L0BLO(7, 7, 15)
            TEB_L0BHI(7, 0, 8) TEB_L0BHI(7, 1, 9) TEB_L0BHI(7, 2, 10) TEB_L0BHI(7, 3, 11) TEB_L0BHI(7, 4, 12) TEB_L0BHI(7, 5, 13) TEB_L0BHI(7, 6, 14) TEB_L0BHI(7, 7, 15)
            : "=&v"(F.a[Ldl8::idx(7, 0)]), "=&v"(F.a[Ldl8::idx(7, 1)]), "=&v"(F.a[Ldl8::idx(7, 2)]), "=&v"(F.a[Ldl8::idx(7, 3)]), "=&v"(F.a[Ldl8::idx(7, 4)]), "=&v"(F.a[Ldl8::idx(7, 5)]), "=&v"(F.a[Ldl8::idx(7, 6)]), "=&v"(F.a[Ldl8::idx(7, 7)])
            : "v"(row[0]), "v"(row[1]), "v"(row[2]), "v"(row[3]), "v"(row[4]), "v"(row[5]), "v"(row[6]), "v"(row[7]));
#undef TEB_L0BLO
#undef TEB_L0BHI
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) F.a[Ldl8::idx(k, k)] += lambda;
        double cl[8], cu[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          // (loaded unconditionally - every address lies inside the band's scratch, sized for 140 doubles per block row where the copy
          //  takes 90 - and the structural zeros selected afterwards: a branch around each of the 16 loads cost ten instructions apiece)
          const double lv = Hi[hbo(k) + (8 + k - c)], uv = Hp[hbo(c) + (8 + c - k)];
          cl[k] = wL[k] = (c >= k - 2) ? lv : 0.0;                 // L_i[k][c]
          cu[k] = wU[k] = (hasU && k >= c - 2) ? uv : 0.0;         // U_i[k][c] = L_{i+1}[c][k]
        }
        ld_row<8>(l.bv + 8 * i, wf);   // (b is zero beyond Nt up to 8 Nb: linearize() clears Nt + 8 entries, the helpers' copy pads the same way)
        ok = F.factor() && ok;
        F.solve3(wL, wU, wf);
        double dm[8];   // the entries of compact row e this lane updates, fetched before the products (nobody else writes them in this phase)
#pragma unroll
        for (int aa = 0; aa < 8; ++aa) dm[aa] = Dc[e * kBlk + aa * 8 + c];
        const double fm = fc[e * 8 + c];
#pragma unroll
        for (int aa = 0; aa < 8; ++aa) { o1[aa] = 0; o2[aa] = 0; o3[aa] = 0; }
        // The operands of the products are already in the group: lane aa holds column aa of L_i (cl) and row aa of L_{i+1} (cu), and the
        // other lanes read them there by DPP (TEB_L0_O1_DPP .. O3_DPP: same products, same order per sum). Up to round 5 they went
        // through LDS - every lane wrote its 8 values as a 64-byte row of a scratch block and read the 8 rows back, 64 ds_read_b128 per
        // lane and round on the pipe the four waves share: that exchange was 3.8 k cycles of a 62 k solve (diagnostic build without the
        // reads), the DPP form costs 147 more vector instructions per round and no LDS access. Headline - 6 %, bit-identical.
        asm("s_nop 1\n\t" TEB_L0_O1_DPP
            : "+v"(o1[0]), "+v"(o1[1]), "+v"(o1[2]), "+v"(o1[3]), "+v"(o1[4]), "+v"(o1[5]), "+v"(o1[6]), "+v"(o1[7])
            : "v"(cl[0]), "v"(cl[1]), "v"(cl[2]), "v"(cl[3]), "v"(cl[4]), "v"(cl[5]), "v"(cl[6]), "v"(cl[7]),
              "v"(wL[0]), "v"(wL[1]), "v"(wL[2]), "v"(wL[3]), "v"(wL[4]), "v"(wL[5]), "v"(wL[6]), "v"(wL[7]));   // (L_i^T W_L)[aa][c]
#pragma unroll
        for (int k = 0; k < 8; ++k) s1 += cl[k] * wf[k];                                    // (L_i^T P f_i)[c]
        if (hasU) {
          asm("s_nop 1\n\t" TEB_L0_O2_DPP
              : "+v"(o2[0]), "+v"(o2[1]), "+v"(o2[2]), "+v"(o2[3]), "+v"(o2[4]), "+v"(o2[5]), "+v"(o2[6]), "+v"(o2[7])
              : "v"(cu[0]), "v"(cu[1]), "v"(cu[2]), "v"(cu[3]), "v"(cu[4]), "v"(cu[5]), "v"(cu[6]), "v"(cu[7]),
                "v"(wL[0]), "v"(wL[1]), "v"(wL[2]), "v"(wL[3]), "v"(wL[4]), "v"(wL[5]), "v"(wL[6]), "v"(wL[7]));   // - (L_{i+1} W_L)[aa][c]
          asm("s_nop 1\n\t" TEB_L0_O3_DPP
              : "+v"(o3[0]), "+v"(o3[1]), "+v"(o3[2]), "+v"(o3[3]), "+v"(o3[4]), "+v"(o3[5]), "+v"(o3[6]), "+v"(o3[7])
              : "v"(cu[0]), "v"(cu[1]), "v"(cu[2]), "v"(cu[3]), "v"(cu[4]), "v"(cu[5]), "v"(cu[6]), "v"(cu[7]),
                "v"(wU[0]), "v"(wU[1]), "v"(wU[2]), "v"(wU[3]), "v"(wU[4]), "v"(wU[5]), "v"(wU[6]), "v"(wU[7]));   // (L_{i+1} W_U)[aa][c]
#pragma unroll
          for (int k = 0; k < 8; ++k) s2 += cu[k] * wf[k];                                  // (L_{i+1} P f_i)[c]
        }
        // fold into the compact rows e (= row i - 1) and e + 1 (= row i + 1); two phases: neighbouring eliminations share a row
#pragma unroll
        for (int aa = 0; aa < 8; ++aa) Dc[e * kBlk + aa * 8 + c] = dm[aa] - o1[aa];
        fc[e * 8 + c] = fm - s1;
        if (hasU) {
#pragma unroll
          for (int aa = 0; aa < 8; ++aa) Lc[(e + 1) * kBlk + aa * 8 + c] = o2[aa];
        }
        // the records of this elimination for the back substitution at the end: W_L, W_U stay in this lane's registers (set here, inside
        // the branch: lanes without an elimination never read theirs - the final store is guarded - and a select against zero per value
        // was 32 instructions per round); P f_i, which every lane of the group holds completely, goes to its place in the step vector
        // through lane 0 (the region is dead until this solve writes its result; a lane picking its own component out of eight registers
        // was another 15)
#pragma unroll
        for (int k = 0; k < 8; ++k) { kL[rr][k] = wL[k]; kU[rr][k] = wU[k]; }
        if (c == 0) {
#pragma unroll
          for (int k = 0; k < 8; k += 2) *reinterpret_cast<teb_v2d*>(l.dxv + 8 * i + k) = teb_v2d{wf[k], wf[k + 1]};
        }
      }
      __syncthreads();
      if (hasU) {
#pragma unroll
        for (int aa = 0; aa < 8; ++aa) Dc[(e + 1) * kBlk + aa * 8 + c] -= o3[aa];
        fc[(e + 1) * 8 + c] -= s2;
      }
      __syncthreads();
    }
  }
  CRP(1);
  // levels >= 1 on the compact system in LDS
  // (Round 6, measured and dropped - profiles/ab_pre_round_r06.txt: a band beyond 256 poses leaves 65 .. 80 compact rows, which the plain
  // level structure reduces in 8 - 10 rounds where 64 rows take 7; a pre-round on the last Nc - 64 odd rows + moving the survivors together
  // gets that to 8, but the rearrangement and its index arithmetic cost what the saved rounds give - every instruction of a wave is ~ 5
  // cycles at one wave per SIMD, a nearly empty round 4.7 k - and any second set of round instances slows the bands that never use it.)
  ok = TEB_CR_FORWARD(Dc, Lc, fc, Nc, 1, Nc) && ok;
  CRP(2);
  ok = cr_top(Dc, fc) && ok;
  if (!ok) l.ired[0] = 0;
  __syncthreads();
  int stop = 1;
  while (stop * 2 < Nc) stop *= 2;
  if (Nc > 1) cr_backward(Dc, Lc, fc, Nc, stop, 1);
  CRP(3);
  // x of the even rows, then the odd rows from the records in registers: x_i = P f_i - W_L x_{i-1} - W_U x_{i+1}; lane c holds column c of
  // W_L and W_U, so the 8 lanes of a group add up their column contributions (butterfly over c), then lane r keeps component r
  for (int q = tid; q < Nc * 8; q += kThreads) {
    const int dst = 8 * TEB_HYB_ROW(q >> 3) + (q & 7);
    if (dst < Nt) l.dxv[dst] = fc[q];
  }
#pragma unroll
  for (int rr = 0; rr < kHybridRounds; ++rr) {
    const int e = rr * (kThreads / 8) + grp;
    const bool act = e < E0;
    const int i = 2 * e + 1;
    const double xm = act ? fc[e * 8 + c] : 0.0;
    const double xp = (act && i + 1 < Nb) ? fc[(e + 1) * 8 + c] : 0.0;
    double mine = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      double t = kL[rr][r] * xm + kU[rr][r] * xp;
      // butterfly over the 8 lanes of the group with DPP moves (no LDS crossbar): lane ^ 1, lane ^ 2 are quad permutations; after them the
      // four lanes of a quad hold the same value, so the mirror within the half row (lane -> 7 - lane) delivers the other quad's sum
      t += dpp_move<0xB1>(t);    // quad_perm:[1,0,3,2]
      t += dpp_move<0x4E>(t);    // quad_perm:[2,3,0,1]
      t += dpp_move<0x141>(t);   // row_half_mirror
      mine = (c == r) ? t : mine;
    }
    if (act && 8 * i + c < Nt) l.dxv[8 * i + c] = l.dxv[8 * i + c] - mine;   // (P f_i was parked there by level 0)
  }
  __syncthreads();  CRP(4);
#undef TEB_HYB_ROW
}

// ---- TimedElasticBand::autoResize (src/timed_elastic_band.cpp:227-286) -------------------------------------
// The rules of a sweep read and write TIME DIFFERENCES only (and the interval count); poses are carried along, and a split inserts
// PoseSE2::average of the two poses either side (pose_se2.h:266-269). So a sweep is split in two:
//   1. one lane runs the reference's sequential rule machine on the dt array alone (exact `i--` re-check semantics: `cur` is the
//      interval under test, `stack` holds the right halves of splits that still wait to be visited) and emits an EDIT SCRIPT: the new
//      dt array, for every output pose a descriptor (input pose j, or new pose q) and for every new pose its two parents (input or
//      earlier new poses - repeated splits of one interval form a small tree) and its depth in that tree;
//   2. all lanes apply the script: new poses level by level (depth 1 = both parents are input poses), then a gather of the output strip.
// Pose descriptors: d < kNewPose = input pose d, else new pose d - kNewPose. New-pose record: parent A | parent B << 11 | depth << 22.
constexpr int kNewPose = 1024;
constexpr int kSplitStack = 64;
// scratch (doubles from off_scratch): odt[S] | nx[S] ny[S] nth[S] | ints: out_desc[S] rec[S] | stack dt[64] | ints: stack desc[64] |
//                                      u64: active masks[16] | ints: runs[S]
__host__ __device__ inline size_t autoresize_scratch_doubles(int S) { return (size_t)5 * S + kSplitStack + kSplitStack / 2 + 16 + (S + 1) / 2 + 8; }
constexpr int kActiveMasks = 16;   // 64 intervals each

// wave-uniform copy of a value: arguments of an out-of-line device function arrive in VGPRs and count as divergent, which would turn
// every `if` of the rule machine into exec-mask bookkeeping (measured: ~860 cycles per rule evaluation); from SGPRs the loop is
// compiled to scalar branches
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
// results: out[0] = n_out, out[1] = modified, out[2] = overflow, out[3] = #new poses, out[4] = deepest split tree, out[5] = #runs,
// out[6] = index of the emitted interval that receives the merged last interval (-1: none); its amount is odt_tail (ints at off_out,
// in units of ints from the LDS base).
// Stretches of intervals inside the dead band [dt_ref - hyst, dt_ref + hyst] that the machine reaches with nothing pending (no excess
// pushed onto them, no split halves waiting) pass through unchanged: the caller marks the intervals OUTSIDE the band in bit masks
// (one ballot per wave), the machine jumps from one marked interval to the next and leaves a run record (first output index, first
// input index, length) for the lanes to expand afterwards. The number of sequential steps is the number of marked intervals plus the
// lengths of the excess / split chains they start, typically a tenth of the band.
__device__ __noinline__ void autoresize_script_lane0(double dt_ref_, double hyst_, int max_samples_, int min_samples_, int off_state_,
                                                       int off_scratch_, int n_in_, int stride_, int off_out_) {
  // all operands are addressed from the dynamic LDS base inside this function, so that the sequential loop is compiled to
  // ds_read / ds_write (generic pointers handed in from the caller would make every access a flat_load / flat_store)
  extern __shared__ __attribute__((aligned(16))) double lds_base[];
  const double dt_ref = uni(dt_ref_), hyst = uni(hyst_);
  const int max_samples = uni(max_samples_), min_samples = uni(min_samples_), off_state = uni(off_state_), off_scratch = uni(off_scratch_),
            n_in = uni(n_in_), stride = uni(stride_);
  int* res = reinterpret_cast<int*>(lds_base) + uni(off_out_);
  int ovf = 0;
  const double* in_dt = lds_base + off_state + 3 * stride;   // Lds: sx sy sth sdt ...
  double* odt = lds_base + off_scratch;
  int* out_desc = reinterpret_cast<int*>(odt + 4 * stride);
  int* rec = out_desc + stride;
  double* stk_dt = odt + 5 * stride;
  int* stk_desc = reinterpret_cast<int*>(stk_dt + kSplitStack);
  const unsigned long long* masks = reinterpret_cast<const unsigned long long*>(stk_dt + kSplitStack + kSplitStack / 2);
  int* runs = reinterpret_cast<int*>(stk_dt + kSplitStack + kSplitStack / 2 + kActiveMasks);
#ifdef TEB_PROFILE
  if (blockIdx.x == 0) g_ar_steps[2] += 1;
  const long long ar_t0 = clock64();
#endif
  const int Tin = n_in - 1;
  int T = Tin;           // sizeTimeDiffs()
  int j = 1;             // next unread input interval
  int sp = 0;            // stack size
  int k = 0;             // emitted intervals
  int nn = 0, md = 0;    // new poses, deepest split tree
  int nruns = 0, tail_k = -1;
  int mchunk = -1; unsigned long long mcur = 0;   // register copy of the marks of the current chunk
  bool modified = false;
  int cdesc = 0, cdepth = 0;
  double cdt = in_dt[0];
  bool fresh = true;     // cur is the untouched input interval cdesc (nothing was added to it)
  // the next unread input interval is kept in a register one step ahead, so that its LDS latency overlaps the rule evaluation
  double pdt = (j < Tin) ? in_dt[j] : 0.0;
  bool ptouched = false; // an excess was pushed onto the prefetched interval
  bool alive = Tin >= 1;
  int top_desc = 0, top_depth = 0; double top_dt = 0;   // register copy of the stack top
  while (alive) {
#ifdef TEB_PROFILE
    if (blockIdx.x == 0) g_ar_steps[1] += 1;
#endif
    if (fresh && sp == 0) {
      // run of unmarked intervals starting at cur = input interval cdesc: up to the next marked one (or the end of the band). The mask of
      // the current 64-interval chunk is kept in registers: a marked interval (the common case where every step counts) costs no LDS read.
      const int j0 = cdesc;
      if ((j0 >> 6) != mchunk) { mchunk = j0 >> 6; mcur = mchunk < kActiveMasks ? masks[mchunk] : ~0ull; }
      int a = j0;
      if (!((mcur >> (j0 & 63)) & 1ull)) {
        a = Tin;
        unsigned long long m = mcur & (~0ull << (j0 & 63));
        for (int cidx = mchunk; ; ) {
          if (m) { a = (cidx << 6) + __ffsll((long long)m) - 1; break; }
          ++cidx;
          if (cidx >= kActiveMasks || (cidx << 6) >= Tin) break;
          m = masks[cidx];
        }
        if (a > Tin) a = Tin;
      }
      const int len = a - j0;
      if (len >= 2) {
        if (k + len > stride - 1) { ovf = 1; break; }
        runs[nruns++] = k | (j0 << 10) | (len << 20);
        k += len;
        j = a + 1;
        if (a < Tin) {
          cdesc = a; cdepth = 0; cdt = in_dt[a]; fresh = true;
          pdt = (j < Tin) ? in_dt[j] : 0.0; ptouched = false;
        } else { alive = false; break; }
      }
    }
    const bool has_next = (sp > 0) || (j < Tin);
    if (cdt > dt_ref + hyst && T < max_samples) {
      if (cdt > 2 * dt_ref) {
        const double newtime = 0.5 * cdt;
        // Pose(i+1): the stack top, else the next input pose, else the goal
        int edesc, edepth;
        if (sp > 0) { edesc = top_desc; edepth = top_depth; }
        else { edesc = (j < Tin) ? j : n_in - 1; edepth = 0; }
        if (sp >= kSplitStack || nn >= stride) { ovf = 1; break; }
        const int depth = 1 + (cdepth > edepth ? cdepth : edepth);
        rec[nn] = cdesc | (edesc << 11) | (depth << 22);
        md = depth > md ? depth : md;
        if (sp > 0) { stk_dt[sp - 1] = top_dt; stk_desc[sp - 1] = top_desc | (top_depth << 16); }   // spill the old top
        top_desc = kNewPose + nn; top_depth = depth; top_dt = newtime;
        ++nn; ++sp;
        cdt = newtime; fresh = false;
        ++T;
        modified = true;
        continue;   // i-- : re-check the left half
      } else {
        if (has_next) {
          if (sp > 0) top_dt += cdt - dt_ref;
          else { pdt += cdt - dt_ref; ptouched = true; }
        }
        cdt = dt_ref; fresh = false;
      }
    } else if (cdt < dt_ref - hyst && T > min_samples) {
      if (has_next) {
        // TimeDiff(i+1) += TimeDiff(i); deleteTimeDiff(i); deletePose(i+1); i--
        if (sp > 0) {
          cdt = top_dt + cdt; --sp;
          if (sp > 0) { top_dt = stk_dt[sp - 1]; const int e = stk_desc[sp - 1]; top_desc = e & 0xffff; top_depth = e >> 16; }
        } else {
          cdt = pdt + cdt; ++j;
          if (j < Tin) pdt = in_dt[j];
          ptouched = false;
        }
        fresh = false;
        --T;
        modified = true;
        continue;
      } else if (k > 0) {
        // last interval: TimeDiff(i-1) += TimeDiff(i); deleteTimeDiff(i); deletePose(i). The interval before may still sit in a run
        // record: the amount is applied after the runs have been expanded.
        tail_k = k - 1;
        odt[stride - 1] = cdt;   // (slot never used by an emitted interval: k <= stride - 1 intervals)
        --T;
        modified = true;
        alive = false;
        break;
      }
    }
    // emit cur, advance
    if (k >= stride - 1) { ovf = 1; break; }
    out_desc[k] = cdesc; odt[k] = cdt;
    ++k;
    if (sp > 0) {
      cdesc = top_desc; cdepth = top_depth; cdt = top_dt; --sp; fresh = false;
      if (sp > 0) { top_dt = stk_dt[sp - 1]; const int e = stk_desc[sp - 1]; top_desc = e & 0xffff; top_depth = e >> 16; }
    } else if (j < Tin) {
      cdesc = j; cdepth = 0; cdt = pdt; fresh = !ptouched; ++j;
      if (j < Tin) pdt = in_dt[j];
      ptouched = false;
    } else alive = false;
  }
#ifdef TEB_PROFILE
  if (blockIdx.x == 0) g_ar_steps[3] += clock64() - ar_t0;
#endif
  res[0] = k + 1; res[1] = modified ? 1 : 0; res[2] = ovf; res[3] = nn; res[4] = md; res[5] = nruns; res[6] = tail_k;
  if (!ovf) out_desc[k] = n_in - 1;   // the goal pose
}


// ---- one sweep as chains that run side by side (teb_autoresize_chain.hpp) ---------------------------------------------------------------
// All lanes: pass A (every input interval starts a chain), pointer doubling over next[] from interval 0, prefix sums over the members,
// pass B (the members write the edit script). Results in ired[16 .. 22] like autoresize_script_lane0 (no run records). Returns 0 when
// the sweep has to be run by the sequential machine instead: a member chain gave up, or one of the two sample-count guards may bind.
// Temporaries: 6 int arrays in the region of the new poses (nx ny nth), dead until the script is applied.
// Out of line, operands addressed from the dynamic LDS base (like autoresize_script_lane0): inlined into the kernel the loop of the chain
// machine inherited the kernel's register pressure - its constants were reloaded from scratch and from the kernel arguments in every
// rule evaluation (~ 750 cycles each); on its own it is allocated from an empty register file.
static_assert(kArNewPose == kNewPose, "descriptor convention");
__device__ __noinline__ __attribute__((not_tail_called)) int autoresize_chains(double dt_ref_, double hyst_, int max_samples_, int min_samples_,
                                                                               int n_, int off_state_, int off_scratch_, int stride_, int off_red_) {
  extern __shared__ __attribute__((aligned(16))) double lds_base[];
  // (arguments of an out-of-line function arrive in VGPRs: wave-uniform copies keep the addressing and the uniform branches scalar)
  const double dt_ref = uni(dt_ref_), hyst = uni(hyst_);
  const int max_samples = uni(max_samples_), min_samples = uni(min_samples_), n = uni(n_), off_state = uni(off_state_),
            off_scratch = uni(off_scratch_), stride = uni(stride_), off_red = uni(off_red_);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int Tin = n - 1;
  const double* in_dt = lds_base + off_state + 3 * stride;   // Lds: sx sy sth sdt ...
  double* odt = lds_base + off_scratch;
  int* out_desc = reinterpret_cast<int*>(odt + 4 * stride);
  int* rec = out_desc + stride;
  int* t_next = reinterpret_cast<int*>(odt + stride);
  int* t_cnt = t_next + stride; int* t_aux = t_cnt + stride; int* t_jump = t_aux + stride; int* t_reach = t_jump + stride; int* t_off = t_reach + stride;
  int* wsum = reinterpret_cast<int*>(lds_base + off_red);   // per-wave partial sums of the scans / reductions
  int* ired = reinterpret_cast<int*>(lds_base + off_red + 64);
  ArChainOut o;
  o.odt = odt; o.out_desc = out_desc; o.rec = rec; o.tail_dt = odt + stride - 1; o.k0 = 0; o.nn0 = 0;
  // pass A
#pragma unroll 1
  for (int kk = 0; kk < kMaxPoseIter; ++kk) {
    const int i = tid + kk * kThreads;
    if (i < Tin) {
      const ArChainResult r = ar_chain_run<false>(in_dt, Tin, i, dt_ref, hyst, o);
      t_next[i] = r.next; t_jump[i] = r.next;
      t_cnt[i] = r.emitted | (r.new_poses << 16);
      t_aux[i] = r.splits | (r.merges << 8) | (r.depth << 16) | (r.gave_up << 24) | (r.tail << 25);
      t_reach[i] = (i == 0) ? 1 : 0;
    }
  }
  __syncthreads();
  // the members of the sweep: 0, next[0], next[next[0]], .. (pointer doubling; a mark set early by a neighbour only adds members sooner)
  for (int round = 0; round < 12; ++round) {
    int marked = 0;
    int nj[kMaxPoseIter];
#pragma unroll
    for (int kk = 0; kk < kMaxPoseIter; ++kk) {
      const int i = tid + kk * kThreads;
      nj[kk] = Tin;
      if (i < Tin) {
        const int j = t_jump[i];
        if (j < Tin) {
          nj[kk] = t_jump[j];
          if (t_reach[i] && !t_reach[j]) { t_reach[j] = 1; marked = 1; }
        }
      }
    }
    if (!__syncthreads_or(marked)) break;
#pragma unroll
    for (int kk = 0; kk < kMaxPoseIter; ++kk) {
      const int i = tid + kk * kThreads;
      if (i < Tin) t_jump[i] = nj[kk];
    }
    __syncthreads();
  }
  // prefix sums over the members in band order: first emitted interval | first new pose << 16; totals of splits, merges, depth, flags
  int base = 0, tot_s = 0, tot_m = 0, md = 0, flags = 0;
#pragma unroll
  for (int kk = 0; kk < kMaxPoseIter; ++kk) {
    const int i = tid + kk * kThreads;
    const bool member = i < Tin && t_reach[i] != 0;
    const int v = member ? t_cnt[i] : 0;
    const int a = member ? t_aux[i] : 0;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(inc, off, 64); if (lane >= off) inc += u; }
    int sm = (a & 0xff) | (((a >> 8) & 0xff) << 16);   // splits | merges << 16
    int dp = (a >> 16) & 0xff, fl = a >> 24;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      sm += __shfl_down(sm, off, 64);
      const int d2 = __shfl_down(dp, off, 64); dp = d2 > dp ? d2 : dp;
      fl |= __shfl_down(fl, off, 64);
    }
    if (lane == 63) wsum[wv] = inc;
    if (lane == 0) { wsum[8 + wv] = sm; wsum[16 + wv] = dp; wsum[24 + wv] = fl; }
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const int ws = wsum[w];
      if (w < wv) before += ws;
      total += ws;
      const int s2 = wsum[8 + w];
      tot_s += s2 & 0xffff; tot_m += s2 >> 16;
      md = wsum[16 + w] > md ? wsum[16 + w] : md;
      flags |= wsum[24 + w];
    }
    if (i < Tin) t_off[i] = base + before + inc - v;
    base += total;
    __syncthreads();
  }
  const int K = base & 0xffff, NN = base >> 16;
  if (flags & 1) return 0;                                                      // a member gave up
  if (Tin + tot_s >= max_samples || Tin - tot_m <= min_samples) return 0;       // a guard may bind somewhere in the sweep
  const int ovf = (K > stride - 1 || NN > stride) ? 1 : 0;
  if (tid == 0) {
    ired[16] = K + 1; ired[17] = (tot_s + tot_m) > 0 ? 1 : 0; ired[18] = ovf; ired[19] = NN; ired[20] = md; ired[21] = 0;
    ired[22] = -1;
  }
  __syncthreads();
  if (ovf) return 1;
  // pass B
#pragma unroll 1
  for (int kk = 0; kk < kMaxPoseIter; ++kk) {
    const int i = tid + kk * kThreads;
    if (i < Tin && t_reach[i] != 0) {
      const int off = t_off[i];
      o.k0 = off & 0xffff; o.nn0 = off >> 16;
      const ArChainResult r = ar_chain_run<true>(in_dt, Tin, i, dt_ref, hyst, o);
      if (r.tail) ired[22] = o.k0 + r.emitted - 1;
    }
  }
  if (tid == 0) out_desc[K] = n - 1;   // the goal pose
  __threadfence_block();
  return 1;
}

__device__ inline int autoresize(const teb_amd_config_t& c, const Lds& l, int n, int off_state, int off_scratch, int stride,
                                 bool fast_mode, int* overflow_flag) {
  extern __shared__ __attribute__((aligned(16))) double lds_base[];
  const int tid = threadIdx.x;
  double* odt = lds_base + off_scratch; double* nx = odt + stride; double* ny = nx + stride; double* nth = ny + stride;
  const int* out_desc = reinterpret_cast<const int*>(odt + 4 * stride);
  const int* rec = out_desc + stride;
  for (int rep = 0; rep < 100; ++rep) {
    // parallel pre-check: a sweep is a no-op iff no interval satisfies either trigger condition; the same pass marks the intervals
    // outside the dead band for the rule machine (one ballot per wave and 64 intervals)
    const int T = n - 1;
    int trig = 0;
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(odt + 5 * stride + kSplitStack + kSplitStack / 2);
    for (int i0 = 0; i0 < kActiveMasks * 64; i0 += kThreads) {
      const int i = i0 + tid;
      bool mark = true;   // beyond the band: marked, so that no run crosses the end
      if (i < T) {
        const double d = l.sdt[i];
        const bool hi = d > c.dt_ref + c.dt_hysteresis, lo = d < c.dt_ref - c.dt_hysteresis;
        mark = hi || lo;
        if ((hi && T < c.max_samples) || (lo && T > c.min_samples)) trig = 1;
      }
      const unsigned long long bal = __ballot(mark);
      if ((tid & 63) == 0 && (i >> 6) < kActiveMasks) masks[i >> 6] = bal;
    }
    trig = __syncthreads_or(trig);
    if (!trig) break;
    // the sweep as parallel chains; the sequential machine below takes over when they decline (uniform decision)
    bool by_chains = false;
    if (T < kMaxPoseIter * kThreads) {
      // cos / sin of the poses as they are now (the cache may date from a rejected LM trial): the new poses average headings through them
      for (int i = tid; i < n; i += kThreads) { double sv, cv; sincos(l.sth[i], &sv, &cv); l.cs[i] = cv; l.sn[i] = sv; }
      by_chains = uni(autoresize_chains(c.dt_ref, c.dt_hysteresis, c.max_samples, c.min_samples, n, off_state, off_scratch, stride,
                                        (int)(l.red - lds_base))) != 0;
    }
    if (!by_chains) {
    if (tid == 0) {
#ifdef TEB_PROFILE
      const long long sw_t0 = clock64();
#endif
      autoresize_script_lane0(c.dt_ref, c.dt_hysteresis, c.max_samples, c.min_samples, off_state, off_scratch, n, stride,
                              (int)(reinterpret_cast<int*>(l.ired + 16) - reinterpret_cast<int*>(lds_base)));
#ifdef TEB_PROFILE
      l.ired[12] += (int)(clock64() - sw_t0); l.ired[13] += 1;
#endif
      __threadfence_block();
    } else if (tid >= 64) {
      // meanwhile the other waves refresh cos / sin of the poses as they are now (the cache may date from a rejected LM trial);
      // wave 0 is excluded so that lane 0 is not held up by its own wave
      for (int i = tid - 64; i < n; i += kThreads - 64) { double sv, cv; sincos(l.sth[i], &sv, &cv); l.cs[i] = cv; l.sn[i] = sv; }
    }
    }   // !by_chains
    __syncthreads();
    const int n_out = l.ired[16], mod = l.ired[17], ovf = l.ired[18], n_new = l.ired[19], md = l.ired[20], nruns = l.ired[21], tail_k = l.ired[22];
    if (ovf) { *overflow_flag = 1; return n; }
    // expand the run records: output intervals [k0, k0 + len) = input intervals [j0, j0 + len), unchanged
    {
      int* out_desc_w = reinterpret_cast<int*>(odt + 4 * stride);
      const int* runs = reinterpret_cast<const int*>(odt + 5 * stride + kSplitStack + kSplitStack / 2 + kActiveMasks);
      for (int q = tid; q < nruns; q += kThreads) {
        const int r = runs[q], k0 = r & 1023, j0 = (r >> 10) & 1023, len = r >> 20;
        for (int t2 = 0; t2 < len; ++t2) { out_desc_w[k0 + t2] = j0 + t2; odt[k0 + t2] = l.sdt[j0 + t2]; }
      }
      __syncthreads();
      if (tid == 0 && tail_k >= 0) odt[tail_k] += odt[stride - 1];   // the merged last interval (src/timed_elastic_band.cpp:274-279)
      __syncthreads();
    }
    // new poses, level by level: PoseSE2::average (pose_se2.h:266-269) with g2o::average_angle
    for (int d = 1; d <= md; ++d) {
      for (int q = tid; q < n_new; q += kThreads) {
        const int r = rec[q];
        if ((r >> 22) != d) continue;
        const int pa = r & 0x7ff, pb = (r >> 11) & 0x7ff;
        double ax, ay, ac, as, bx, by, bc, bs;
        if (pa < kNewPose) { ax = l.sx[pa]; ay = l.sy[pa]; ac = l.cs[pa]; as = l.sn[pa]; }
        else { const double t = nth[pa - kNewPose]; ax = nx[pa - kNewPose]; ay = ny[pa - kNewPose]; ac = cos(t); as = sin(t); }
        if (pb < kNewPose) { bx = l.sx[pb]; by = l.sy[pb]; bc = l.cs[pb]; bs = l.sn[pb]; }
        else { const double t = nth[pb - kNewPose]; bx = nx[pb - kNewPose]; by = ny[pb - kNewPose]; bc = cos(t); bs = sin(t); }
        const double sxn = ac + bc, syn = as + bs;
        nx[q] = (ax + bx) / 2; ny[q] = (ay + by) / 2;
        nth[q] = (sxn == 0 && syn == 0) ? 0.0 : atan2(syn, sxn);
      }
      __syncthreads();
    }
    // gather the output strip through registers (in place: every source is read before any destination is written)
    double gx[kMaxPoseIter], gy[kMaxPoseIter], gth[kMaxPoseIter], gdt[kMaxPoseIter];
#pragma unroll
    for (int kk = 0; kk < kMaxPoseIter; ++kk) {
      const int i = tid + kk * kThreads;
      if (i < n_out) {
        const int d = out_desc[i];
        if (d < kNewPose) { gx[kk] = l.sx[d]; gy[kk] = l.sy[d]; gth[kk] = l.sth[d]; }
        else { gx[kk] = nx[d - kNewPose]; gy[kk] = ny[d - kNewPose]; gth[kk] = nth[d - kNewPose]; }
        gdt[kk] = (i < n_out - 1) ? odt[i] : 0.0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < kMaxPoseIter; ++kk) {
      const int i = tid + kk * kThreads;
      if (i < n_out) {
        l.sx[i] = gx[kk]; l.sy[i] = gy[kk]; l.sth[i] = gth[kk];
        if (i < n_out - 1) l.sdt[i] = gdt[kk];
      }
    }
    n = n_out;
    __syncthreads();
    if (!mod || fast_mode) break;
  }
  return n;
}

// ---- obstacle association, AddEdgesObstacles (src/optimal_planner.cpp:444-548) -------------------------------
// One pose per lane scans the static list; a pass with at most kThreads / 2 poses (short bands, the second pass of bands longer than
// kThreads poses) gives G = 2, 4 or 8 adjacent lanes to a pose, each scanning a contiguous chunk of the list. The chunks are in list
// order, so concatenating their forced inclusions and taking the first minimum over the chunks reproduces the sequential result bit
// for bit: forced entries of a slice wait in registers (up to kSliceForced) for their offset (prefix sum over the slices); a pose
// that overflows them is redone by its first lane the sequential way.
constexpr int kSliceForced = 6;
struct AssocScan {
  double left_min, right_min;
  int left, right, cnt;
};
// scans static-list positions [k_lo, k_hi) for pose i; forced inclusions go to `emit(position)` in list order
// Pass 1 of the association of a point-like scene without radii, for a scan whose bounds are WAVE-UNIFORM (one lane per pose, the whole
// list): bit j of the result = obstacle ka + j of the static list is within thr of (x, y) or the comparison is unordered; kb - ka <= 32.
// The obstacle is the same for every lane, so its coordinates come through the scalar cache (constant address space: s_load_dwordx16 for
// 8 values, no VGPR, no LDS: with the LDS copy the four waves of the workgroup scanning at once are bound by the LDS pipe, 32 cycles per
// obstacle, and measured 112) and enter the subtraction as its scalar operand; the lane's bit is shifted in from VCC by ONE add-with-carry
// (the first obstacle ends up in the highest bit: reversed once at the end) instead of a conditional move and an OR with a computed
// constant. Six vector instructions per obstacle instead of ten and two LDS reads. The squared distance is contracted into one fma: a
// mask only has to be a SUPERSET of the obstacles the scan can care about (pass 2 takes their exact distances and makes the reference's
// decisions), the guard band of 1e-12 on the threshold is four orders of magnitude wider than the one rounding that differs - lists
// bit-identical (tests/test_gpu_parity.py: test_linearisation_and_association; the fingerprints).
__device__ __forceinline__ unsigned assoc_near_bits(const SceneDev& sc, int ka, int kb, double x, double y, double thr2) {
  typedef const __attribute__((address_space(4))) double* kptr;
  const kptr gx = (kptr)(unsigned long long)sc.lox, gy = (kptr)(unsigned long long)sc.loy;
  unsigned m = 0;
  const int nk = kb - ka;
  auto one = [&](double ox, double oy) {
    const double ddx = x - ox, ddy = y - oy;
    const double d2 = __builtin_fma(ddy, ddy, ddx * ddx);
    asm("v_cmp_ngt_f64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(d2), "v"(thr2) : "vcc");   // m = 2 m + !(d2 > thr2)
  };
  int k = ka;
  // batches of 8 (two s_load_dwordx16). Not double-buffered: the kernel has no 32 scalar registers to spare here - the compiler parks a
  // prefetched batch in VGPR lanes (64 v_writelane / v_readlane per batch) and waits for it at once
  for (; k + 8 <= kb; k += 8) {
    double ox[8], oy[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { ox[u] = gx[k + u]; oy[u] = gy[k + u]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) one(ox[u], oy[u]);
  }
  for (; k < kb; ++k) one(gx[k], gy[k]);
  return nk > 0 ? __builtin_bitreverse32(m) >> (32 - nk) : 0u;
}
template <bool FAST, bool UNIFORM = false, typename Emit>
__device__ __forceinline__ void assoc_scan(const teb_amd_config_t& c, const SceneDev& sc, const Lds& l, int i, int k_lo, int k_hi, AssocScan& r,
                                           Emit emit) {
  const double x = l.sx[i], y = l.sy[i];
  const double ox_ = l.cs[i], oy_ = l.sn[i];   // orientationUnitVec
  const double force = c.min_obstacle_dist * c.obstacle_association_force_inclusion_factor;
  const double cutoff = c.min_obstacle_dist * c.obstacle_association_cutoff_factor;
  auto visit = [&](int k, double dist, double ccx, double ccy) {
    if (dist < force) { emit(k); ++r.cnt; return; }
    if (dist > cutoff) return;
    // cross2d(pose_orient, centroid - position) > 0 -> left (misc.h:119-123)
    double vx_ = ccx - x, vy_ = ccy - y;
    if (ox_ * vy_ - vx_ * oy_ > 0) { if (dist < r.left_min) { r.left_min = dist; r.left = k; } }
    else { if (dist < r.right_min) { r.right_min = dist; r.right = k; } }
  };
  // the same decisions as selects, for the candidates of the point-like path (round 5: seven nested exec-mask branches per candidate cost
  // 12 % of pass 2; the rare forced inclusion keeps its branch)
  auto visit_sel = [&](int k, double dist, double ccx, double ccy) {
    const bool forced = dist < force;
    const bool cand = !forced && !(dist > cutoff);
    const double vx_ = ccx - x, vy_ = ccy - y;
    const bool on_left = ox_ * vy_ - vx_ * oy_ > 0;
    const bool ul = cand && on_left && dist < r.left_min, ur = cand && !on_left && dist < r.right_min;
    r.left_min = ul ? dist : r.left_min; r.left = ul ? k : r.left;
    r.right_min = ur ? dist : r.right_min; r.right = ur ? k : r.right;
    if (forced) { emit(k); ++r.cnt; }
  };
  (void)visit_sel; (void)visit;
  int k0 = k_lo;
  TEB_IF_FAST(FAST) {
    // Far-field culling, exact: an obstacle beyond max(cutoff, force) (+ a relative guard band of 1e-12 against the rounding of the
    // distance) from the pose is neither included nor a left / right candidate. Pass 1 (uniform over the wave): squared distances, one
    // bit per obstacle; pass 2: each lane visits the set bits of its own mask in list order - the decisions of the sequential scan,
    // without the square roots of the ~90 % of the obstacles that are out of reach.
    const double far_d = fmax(cutoff, force) + (c.footprint_type == TEB_AMD_FOOTPRINT_CIRCULAR ? c.footprint_radius : 0.0);
    for (; k0 < k_hi; k0 += 64) {
      const int ke = k0 + 64 < k_hi ? k0 + 64 : k_hi;
      unsigned long long near = 0;
      if (TEB_CFGI(RADIUS_FREE)) {
        // no radii in the static list (point obstacles): the threshold is one number, hoisted; the mask is built in its two 32-bit halves
        // (the bit of obstacle k is a scalar: a conditional move and an OR per obstacle). Same comparisons on the same values.
        const double thr = (far_d + 0.0) * (1.0 + 1e-12), thr2 = thr * thr;
        const bool all = thr <= 0;
        unsigned lo = 0, hi = 0;
        const int km = k0 + 32 < ke ? k0 + 32 : ke;
        if constexpr (UNIFORM) {
          lo = assoc_near_bits(sc, k0, km, x, y, thr2);
          hi = assoc_near_bits(sc, km, ke, x, y, thr2);
        } else {
#pragma unroll 4
        for (int k = k0; k < km; ++k) {
          const double ddx = x - l.obx[k], ddy = y - l.oby[k];
          const double d2 = ddx * ddx + ddy * ddy;
          lo |= !(d2 > thr2) ? 1u << (k - k0) : 0u;
        }
#pragma unroll 4
        for (int k = km; k < ke; ++k) {
          const double ddx = x - l.obx[k], ddy = y - l.oby[k];
          const double d2 = ddx * ddx + ddy * ddy;
          hi |= !(d2 > thr2) ? 1u << (k - km) : 0u;
        }
        }
        near = ((unsigned long long)hi << 32) | lo;
        if (all) near = ke - k0 >= 64 ? ~0ull : ((1ull << (ke - k0)) - 1ull);
      } else
      {
#pragma unroll 4
      for (int k = k0; k < ke; ++k) {
        const double ddx = x - l.obx[k], ddy = y - l.oby[k];
        const double d2 = ddx * ddx + ddy * ddy;
        const double thr = (far_d + l.obr[k]) * (1.0 + 1e-12);
        if (!(d2 > thr * thr) || thr <= 0) near |= 1ull << (k - k0);   // non-finite distances count as near
      }
      }
      while (near) {
        const int k = k0 + __ffsll((long long)near) - 1;
        near &= near - 1;
        const double ccx = l.obx[k], ccy = l.oby[k];
        // (no radii in the list: the subtraction of an exact zero is skipped with its LDS read)
        const double dist = pointlike_distance<false>(c, x, y, ccx, ccy, TEB_CFGI(RADIUS_FREE) ? 0.0 : l.obr[k], nullptr);
        visit_sel(k, dist, ccx, ccy);
      }
    }
    k0 = k_hi;
  }
  TEB_IF_NOT_FAST(FAST) {
    // Generic shapes: the same two passes with bounding circles. The robot lies within frad of the pose, the obstacle within brad of its
    // centroid, so their distance is at least |centroid - pose| - brad - frad; beyond max(cutoff, force) (+ guard band) the obstacle
    // cannot matter and its exact distance (segment / polygon loops, the expensive part: 75 % of BASELINE C5 before) is never computed.
    const double frad = footprint_bound_radius(c);
    const double far_d = fmax(cutoff, force) + frad;
    for (; k0 < k_hi; k0 += 64) {
      const int ke = k0 + 64 < k_hi ? k0 + 64 : k_hi;
      unsigned long long near = 0;
#pragma unroll 4
      for (int k = k0; k < ke; ++k) {
        const int oi = sc.static_idx[k];
        const double ddx = x - sc.cx[oi], ddy = y - sc.cy[oi];
        const double d2 = ddx * ddx + ddy * ddy;
        const double thr = (far_d + sc.brad[oi]) * (1.0 + 1e-9);
        if (!(d2 > thr * thr) || thr <= 0) near |= 1ull << (k - k0);   // NaN / infinite bounds count as near
      }
      while (near) {
        const int k = k0 + __ffsll((long long)near) - 1;
        near &= near - 1;
        const int oi = sc.static_idx[k];
        // The same bound decides most candidates without their exact distance: one that is certainly not force-included (bound >= force)
        // and certainly not closer than the nearest obstacle found so far on its side (bound >= that minimum; the scan keeps the FIRST
        // minimum, so an equal distance would not replace it either) leaves the scan's state untouched. Bit-identical, C5 14.1 -> 12.9 ms
        // (25 -> 10 exact distances per pose). Round 2 kept it out of the one kernel that served both scene kinds (its mere presence
        // cost the point-like configurations 1.5 % through code placement); the generic scenes now have their own instantiation.
        const double lb = distance_lower_bound(sc, oi, x, y, frad);
        const bool on_left = ox_ * (sc.cy[oi] - y) - (sc.cx[oi] - x) * oy_ > 0;
#ifdef TEB_PROFILE
        atomicAdd(&g_assoc_stats[0], 1ull);
#endif
        if (lb >= force && lb >= (on_left ? r.left_min : r.right_min)) continue;
#ifdef TEB_PROFILE
        atomicAdd(&g_assoc_stats[1], 1ull);
#endif
        const double dist = footprint_distance(c, sc, oi, x, y, ox_, oy_, false, 0.0, nullptr);
        visit(k, dist, sc.cx[oi], sc.cy[oi]);
      }
    }
  }
}

// Poses [p_begin, p_end) of a band of n poses. SHARED: the lists are read by other workgroups (multi-CU mode: a helper scans its pose
// tile, always sliced) and are written with agent-scope stores.
template <bool FAST, bool SHARED>
// stage / stage_ints: LDS scratch a helper lends its lanes for the forced inclusions of their slices (kThreads x chunk ints; a pose in a
// cluster of obstacles has more of them than the registers of kSliceForced hold, and the sequential fallback is the slow path).
__device__ inline void associate_range(const teb_amd_config_t& c, const SceneDev& sc, const Lds& l, int n, int p_begin, int p_end, int* assoc_cnt,
                                       int* assoc, int cap, int stride, int* overflow, int* stage = nullptr, int stage_ints = 0) {
  const int first_vertex = c.weight_velocity_obstacle_ratio == 0 ? 1 : 0;
  const double kMax = 1.7976931348623157e308;
  const int tid = threadIdx.x;
  for (int p0 = p_begin; p0 < p_end; ) {
    const int G = (SHARED || p0 > 0 || kThreads > 256) ? lanes_per_pose(p_end - p0) : 1;
    const int i = p0 + tid / G, sl = tid % G;
    const bool has_pose = i < p_end;
    const bool scans = has_pose && i >= first_vertex && i < n - 1;
    // the sequential scan of the whole list by one lane, writing straight into the list (also the fallback of the sliced scan)
    auto sequential = [&]() {
      AssocScan r = {kMax, kMax, -1, -1, 0};
      assoc_scan<FAST, true>(c, sc, l, i, 0, sc.n_static, r, [&](int k) { if (r.cnt < cap) st_list<SHARED>(&assoc[(size_t)r.cnt * stride + i], k); else *overflow = 1; });
      int cnt = r.cnt;
      if (r.left >= 0) { if (cnt < cap) st_list<SHARED>(&assoc[(size_t)cnt * stride + i], r.left); else *overflow = 1; ++cnt; }
      if (r.right >= 0) { if (cnt < cap) st_list<SHARED>(&assoc[(size_t)cnt * stride + i], r.right); else *overflow = 1; ++cnt; }
      st_list<SHARED>(&assoc_cnt[i], cnt > cap ? cap : cnt);
    };
    if (G == 1) {
      if (scans) sequential();
      else if (has_pose) st_list<SHARED>(&assoc_cnt[i], 0);
    } else {
      const int chunk = ((sc.n_static + G - 1) / G + 3) & ~3;
      const int k_lo = sl * chunk < sc.n_static ? sl * chunk : sc.n_static;
      const int k_hi = k_lo + chunk < sc.n_static ? k_lo + chunk : sc.n_static;
      AssocScan r = {kMax, kMax, -1, -1, 0};
      int forced[kSliceForced];
#pragma unroll
      for (int q = 0; q < kSliceForced; ++q) forced[q] = 0;
      const bool staged = SHARED && stage != nullptr && kThreads * chunk <= stage_ints;   // (a slice holds at most `chunk` entries)
      int* mine = stage + tid * chunk;
      if (scans) {
        if (staged) {
          assoc_scan<FAST>(c, sc, l, i, k_lo, k_hi, r, [&](int k) { mine[r.cnt] = k; });
        } else {
          assoc_scan<FAST>(c, sc, l, i, k_lo, k_hi, r, [&](int k) {
#pragma unroll
            for (int q = 0; q < kSliceForced; ++q) if (r.cnt == q) forced[q] = k;   // register file: no dynamic indexing
          });
        }
      }
      // over the G slices of the pose (adjacent lanes): offset of this slice's forced entries, total, first minima, "registers overflowed"
      int before = 0, total = r.cnt, spilled = !staged && r.cnt > kSliceForced;
      for (int off = 1; off < G; off <<= 1) {
        const int up = __shfl_up(total, off, 64);
        if (sl >= off) total += up;   // inclusive scan over sl
      }
      before = total - r.cnt;
      total = __shfl(total, (tid & 63) | (G - 1), 64);   // the last slice holds the sum
      for (int off = 1; off < G; off <<= 1) {
        spilled |= __shfl_xor(spilled, off, 64);
        const double lm = __shfl_xor(r.left_min, off, 64), rm = __shfl_xor(r.right_min, off, 64);
        const int li = __shfl_xor(r.left, off, 64), ri = __shfl_xor(r.right, off, 64);
        // strict '<' of the sequential scan keeps the FIRST minimum: on equal distances the lower list position wins
        if (li >= 0 && (r.left < 0 || lm < r.left_min || (lm == r.left_min && li < r.left))) { r.left_min = lm; r.left = li; }
        if (ri >= 0 && (r.right < 0 || rm < r.right_min || (rm == r.right_min && ri < r.right))) { r.right_min = rm; r.right = ri; }
      }
      if (scans) {
        const int full = total + (r.left >= 0) + (r.right >= 0);
        if (spilled || full > cap) {
          if (sl == 0) sequential();
        } else {
          if (staged) {
            for (int q = 0; q < r.cnt; ++q) st_list<SHARED>(&assoc[(size_t)(before + q) * stride + i], mine[q]);
          } else {
#pragma unroll
            for (int q = 0; q < kSliceForced; ++q) if (q < r.cnt) st_list<SHARED>(&assoc[(size_t)(before + q) * stride + i], forced[q]);
          }
          if (sl == 0) {
            int cnt = total;
            if (r.left >= 0) { st_list<SHARED>(&assoc[(size_t)cnt * stride + i], r.left); ++cnt; }
            if (r.right >= 0) { st_list<SHARED>(&assoc[(size_t)cnt * stride + i], r.right); ++cnt; }
            st_list<SHARED>(&assoc_cnt[i], cnt);
          }
        }
      } else if (has_pose && sl == 0) {
        st_list<SHARED>(&assoc_cnt[i], 0);
      }
    }
    p0 += kThreads / G;
  }
}
template <bool FAST>
__device__ inline void associate(const teb_amd_config_t& c, const SceneDev& sc, const Lds& l, int n, int* assoc_cnt,
                                 int* assoc, int cap, int stride, int* overflow) {
  associate_range<FAST, false>(c, sc, l, n, 0, n, assoc_cnt, assoc, cap, stride, overflow);
}

// ---- multi-CU mode: the helper workgroups of a band (teb_multicu.hpp) ---------------------------------------------------------------
// Distance helper j of D serves the pose tile [j P, (j + 1) P), P = ceil(n / D), of every phase the master issues, until MCU_KIND_EXIT, an abort
// or the timeout. The published poses are staged into the helper's own LDS strips (same layout as the master's), so the device
// functions of the single-CU path (association scan, footprint_distance) run here unchanged.
__device__ inline void mcu_helper(const teb_amd_config_t& c, const SceneDev& sc, const BatchDev& bt, const McuDev& mc, const LdsPlan& plan, int b, int j) {
  extern __shared__ __attribute__((aligned(16))) double lds_base[];
  const Lds l = carve(lds_base, plan);
  const int tid = threadIdx.x, S = bt.stride;
  unsigned* ctl = mc.ctl + (size_t)b * kMcuCtlWords;
  const double* pub = mc.pub + (size_t)b * kMcuPubArrays * S;
  double* items = mc.items + (size_t)b * mc.item_cap * 4 * S;
  int* assoc_cnt = bt.assoc_cnt + (size_t)b * S;
  int* assoc = bt.assoc + (size_t)b * bt.assoc_cap * S;
  int* prefix = reinterpret_cast<int*>(lds_base + plan.off_H);   // [P + 1] record offsets of the tile's poses (the normal-matrix region is free here)
  unsigned epoch = 0;
  for (;;) {
    mcu_trace(mc.trace, epoch, 0x10);
    // ONE lane polls ONE word (relaxed, agent scope); the others sleep at the barrier
    if (tid == 0) {
      unsigned cmd = 0;
      const long long t0 = realtime_ticks();
      for (;;) {
        cmd = ld_agent_u32(ctl + MCU_CMD);
        if ((cmd >> 8) > epoch) break;
        // (four times the master's patience: a master may be busy with a solve between two phases)
        if (ld_agent_u32(ctl + MCU_ABORT) != 0 || realtime_ticks() - t0 > 4 * mc.timeout_ticks + 100000) { cmd = ((epoch + 1) << 8) | MCU_KIND_EXIT; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      l.ired[24] = (int)cmd;
      l.ired[25] = (int)ld_agent_u32(ctl + MCU_N);
    }
    __syncthreads();
    const unsigned cmd = (unsigned)l.ired[24];
    const int n = l.ired[25];
    __syncthreads();
    epoch = cmd >> 8;
    const int kind = (int)(cmd & 0xffu);
    mcu_trace(mc.trace, epoch, 0x20 + (unsigned)kind);
    if (kind == MCU_KIND_EXIT || n < 2 || n > plan.S) return;
    const int P = (n + mc.D - 1) / mc.D;
    const int p_lo = j * P < n ? j * P : n, p_hi = p_lo + P < n ? p_lo + P : n;
    // stage the poses of the tile (agent-scope loads: the lines were written through by the master)
    for (int i = p_lo + tid; i < p_hi; i += kThreads) {
      l.sx[i] = ld_agent_f64(pub + i); l.sy[i] = ld_agent_f64(pub + S + i); l.cs[i] = ld_agent_f64(pub + 2 * S + i); l.sn[i] = ld_agent_f64(pub + 3 * S + i);
      l.tdyn[i] = ld_agent_f64(pub + 4 * S + i);
    }
    __syncthreads();
    mcu_trace(mc.trace, epoch, 0x30);
    if (kind == MCU_KIND_ASSOC) {
      int ovf = 0;
      if (p_lo < p_hi)
        associate_range<false, true>(c, sc, l, n, p_lo, p_hi, assoc_cnt, assoc, bt.assoc_cap, S, &ovf, reinterpret_cast<int*>(lds_base + plan.off_H),
                                     2 * (int)hmat_doubles(plan.S, plan.solver));
      if (ovf) or_agent_i32(bt.assoc_overflow + b, 1);
    } else {   // MCU_KIND_DIST: one (pose, record) pair per lane
      const bool dyn_on = c.include_dynamic_obstacles && c.weight_obstacle != 0;
      const int ndyn = dyn_on ? sc.n_dyn : 0;
      const int np = p_hi - p_lo;
      for (int q = tid; q < np; q += kThreads) {
        int cnt = ld_agent_i32(assoc_cnt + p_lo + q);
        if (p_lo + q >= n - 1) cnt = 0;                               // the goal pose carries no edge
        const bool unary = p_lo + q >= 1;                             // pose 0: only the velocity-obstacle-ratio edges read records
        prefix[q + 1] = cnt + (unary && p_lo + q < n - 1 ? ndyn : 0);
      }
      __syncthreads();
      if (tid == 0) { prefix[0] = 0; for (int q = 0; q < np; ++q) prefix[q + 1] += prefix[q]; }
      __syncthreads();
      const int total = np > 0 ? prefix[np] : 0;
      for (int w = tid; w < total; w += kThreads) {
        int lo = 0, hi = np - 1;                                      // largest q with prefix[q] <= w
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (prefix[mid] <= w) lo = mid; else hi = mid - 1; }
        const int i = p_lo + lo, k = w - prefix[lo];
        int cnt = ld_agent_i32(assoc_cnt + i);
        double gr[3], dist;
        if (k < cnt) {
          const int ent = ld_agent_i32(&assoc[(size_t)k * S + i]);
          dist = footprint_distance(c, sc, sc.static_idx[ent & kAssocMask], l.sx[i], l.sy[i], l.cs[i], l.sn[i], false, 0.0, gr);
        } else {
          dist = footprint_distance(c, sc, sc.dyn_idx[k - cnt], l.sx[i], l.sy[i], l.cs[i], l.sn[i], true, l.tdyn[i], gr);
        }
        double* it = items + i;
        st_agent_f64(it + (size_t)(4 * k) * S, dist); st_agent_f64(it + (size_t)(4 * k + 1) * S, gr[0]);
        st_agent_f64(it + (size_t)(4 * k + 2) * S, gr[1]); st_agent_f64(it + (size_t)(4 * k + 3) * S, gr[2]);
      }
    }
    // every storing wave drains, the workgroup meets, ONE lane arrives
    mcu_trace(mc.trace, epoch, 0x40);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) add_agent_u32(ctl + MCU_DONE, 1u);
    mcu_trace(mc.trace, epoch, 0x50);
  }
}

// ---- legacy association, AddEdgesObstaclesLegacy (src/optimal_planner.cpp:551-643) ---------------------------------------
// Per static obstacle: index = findClosestTrajectoryPose(obstacle) (src/timed_elastic_band.cpp:455-552; n/2 for all when
// obstacle_poses_affected >= n); skipped unless 1 < index <= n-2; one edge at index, then for nb = 0 .. floor(poses_affected/2)-1
// edges at index+nb and index-nb (so the closest pose carries the edge three times when poses_affected >= 2).
// Phase A: one thread per obstacle finds its pose; phase B: one thread per pose collects its obstacles in list order.
__device__ inline void associate_legacy(const teb_amd_config_t& c, const SceneDev& sc, const Lds& l, int n, int* assoc_cnt,
                                        int* assoc, int cap, int stride, int* overflow, int* idx) {
  const double kMax = 1.7976931348623157e308;
  for (int k = threadIdx.x; k < sc.n_static; k += kThreads) {
    int index;
    if (c.obstacle_poses_affected >= n) index = n / 2;
    else {
      const int oi = sc.static_idx[k];
      const int ty = sc.type[oi];
      const int v0 = sc.voff[oi], nv = sc.voff[oi + 1] - v0;
      double best = kMax;
      index = -1;
      if (ty == TEB_AMD_OBST_LINE || (ty == TEB_AMD_OBST_POLYGON && nv == 2)) {
        double ax, ay, bx, by;
        if (ty == TEB_AMD_OBST_LINE) { ax = sc.ax[oi]; ay = sc.ay[oi]; bx = sc.bx[oi]; by = sc.by[oi]; }
        else { ax = sc.pvx[v0]; ay = sc.pvy[v0]; bx = sc.pvx[v0 + 1]; by = sc.pvy[v0 + 1]; }
        for (int i = 0; i < n; ++i) {
          double cx, cy;
          const double d = point_segment(l.sx[i], l.sy[i], ax, ay, bx, by, cx, cy);
          if (d < best) { best = d; index = i; }
        }
      } else if (ty == TEB_AMD_OBST_POLYGON && nv > 2) {
        for (int i = 0; i < n; ++i) {
          const double px = l.sx[i], py = l.sy[i];
          double dp = kMax, cx, cy;
          for (int j = 0; j < nv - 1; ++j) {
            const double d = point_segment(px, py, sc.pvx[v0 + j], sc.pvy[v0 + j], sc.pvx[v0 + j + 1], sc.pvy[v0 + j + 1], cx, cy);
            if (d < dp) dp = d;   // std::min(dp, d)
          }
          const double dc = point_segment(px, py, sc.pvx[v0 + nv - 1], sc.pvy[v0 + nv - 1], sc.pvx[v0], sc.pvy[v0], cx, cy);
          if (dc < dp) dp = dc;
          if (dp < best) { best = dp; index = i; }
        }
      } else if (ty == TEB_AMD_OBST_POLYGON && nv == 0) {
        index = 0;
      } else {   // point-like reference: the point itself, a one-vertex polygon, else the centroid (circle, pill)
        double qx, qy;
        if (ty == TEB_AMD_OBST_POINT) { qx = sc.ax[oi]; qy = sc.ay[oi]; }
        else if (ty == TEB_AMD_OBST_POLYGON) { qx = sc.pvx[v0]; qy = sc.pvy[v0]; }
        else { qx = sc.cx[oi]; qy = sc.cy[oi]; }
        for (int i = 0; i < n; ++i) {   // squared distance, strict '<' (timed_elastic_band.cpp:455-478)
          const double ddx = qx - l.sx[i], ddy = qy - l.sy[i];
          const double d2 = ddx * ddx + ddy * ddy;
          if (d2 < best) { best = d2; index = i; }
        }
      }
    }
    idx[k] = index;
  }
  __threadfence_block();
  __syncthreads();
  const int half = c.obstacle_poses_affected / 2;   // floor(int / int), :583
  for (int i = threadIdx.x; i < n; i += kThreads) {
    int cnt = 0;
    if (i >= 1 && i <= n - 2) {
      for (int k = 0; k < sc.n_static; ++k) {
        const int index = idx[k];
        if (index <= 1 || index > n - 2) continue;
        int ent = -1;
        if (i == index) ent = half >= 1 ? (k | kAssocTriple) : k;
        else if ((i > index ? i - index : index - i) < half) ent = k;
        if (ent >= 0) {
          if (cnt < cap) assoc[(size_t)cnt * stride + i] = ent; else *overflow = 1;
          ++cnt;
        }
      }
      if (cnt > cap) cnt = cap;
    }
    assoc_cnt[i] = cnt;
  }
}

// =================================================================================================================
// ---- multi-CU mode: solver helper k of a band (speculative LM trials, teb_multicu.hpp) ------------------------------------------------
// Per LM iteration: wait for the command, take the right-hand side and lambda_k, (blocks-in-LDS layout: load the normal matrix from the
// band's HBM backup into this workgroup's own LDS), run the band's damped solve, write the step back through. The data were written
// through (sc1) by the band's workgroup: ONE agent-scope acquire after the poll drops this CU's stale lines, then plain loads.
template <int SOLVER>
__device__ inline void mcu_solver_helper(const SceneDev& sc, const BatchDev& bt, const McuDev& mc, const LdsPlan& plan, int b, int k) {
  extern __shared__ __attribute__((aligned(16))) double lds_base[];
  const Lds l = carve(lds_base, plan);
  const int tid = threadIdx.x, S = bt.stride;
  unsigned* ctl = mc.ctl + (size_t)b * kMcuCtlWords;
  const size_t slot = mcu_spec_slot(S);
  const double* in = mc.spec + (size_t)b * (mc.K + 1) * slot;
  double* out = mc.spec + ((size_t)b * (mc.K + 1) + k) * slot;
  double* Hbk = bt.Hbackup + (size_t)b * bt.hmat_stride;
  unsigned epoch = 0;
  for (;;) {
    mcu_trace(mc.trace, epoch, 0x60);
    if (tid == 0) {
      unsigned cmd = 0;
      const long long t0 = realtime_ticks();
      for (;;) {
        cmd = ld_agent_u32(ctl + MCU_SCMD);
        if (cmd > epoch) break;
        // patience: the band's own (multi_cu_timeout_us, 2 ms by default and at least that - the probe launches after a back-off shorten the
        // band's side only: an LM iteration of the largest bands is a tenth of it, the association before the first command a fifth). A band
        // that has given its helpers up says so (MCU_SABORT, spec_take / spec_wait_idle) and one that has finished sends the exit command;
        // the bound is for a band whose workgroup is not running at all - a helper that leaves early only costs its band the one
        // short wait of spec_take, never a bit of the result (was 4 x timeout + 1 ms)
        if (ld_agent_u32(ctl + MCU_SABORT) != 0 || realtime_ticks() - t0 > (mc.timeout_ticks > 200000 ? mc.timeout_ticks : 200000)) { cmd = kMcuSpecExit; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      l.ired[24] = (int)cmd;
      l.ired[25] = (int)ld_agent_u32(ctl + MCU_SN);
      if (cmd != kMcuSpecExit) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    const unsigned cmd = (unsigned)l.ired[24];
    const int n = l.ired[25];
    __syncthreads();
    if (cmd == kMcuSpecExit || n < 2 || n > plan.S) return;
    // every wave acquires before its plain loads of the band's buffers (H backup / band copy, right-hand side): the memory model asks each
    // reader for it, not only the lane that polled
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    epoch = cmd;
    mcu_trace(mc.trace, epoch, 0x61);
    const int Nt = 4 * n;
    const double lambda = in[4 * S + 8 + k];
    for (int r = tid; r < Nt + 8; r += kThreads) l.bv[r] = r < Nt ? in[r] : 0.0;
    if constexpr (SOLVER == SOLVER_CR) {
      const int hsz = ((Nt + 7) >> 3) * 2 * kBlk;
      hmat_load<SOLVER_CR>(l, hsz, Nt, Hbk);
    }
    __syncthreads();
    if constexpr (SOLVER == SOLVER_CR) cr_solve_blocks_helper(plan, sc, n, lambda);
    else cr_solve_hybrid_helper(plan, n, lambda, Hbk);
    for (int r = tid; r < Nt; r += kThreads) st_agent_f64(out + r, l.dxv[r]);
    if (tid == 0) st_agent_f64(out + 4 * S + 8, l.ired[0] != 0 ? 1.0 : 0.0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) st_agent_u32(ctl + MCU_SDONE + k, epoch);
    mcu_trace(mc.trace, epoch, 0x62);
  }
}

// SCENE: SCENE_POINTS = every obstacle and the footprint are point-like and the obstacle table is cached in LDS (SceneDev::fast_points),
// SCENE_GENERIC = any shapes (segment / polygon distance loops, bounding-circle culling). One instantiation each, so that neither pays
// for the other's code (placement, registers).
// The _SMALL kinds are the same two for SMALL BATCHES (teb_multicu.hpp): launched with helper workgroups on the CUs the batch leaves idle -
// K solver helpers per band (speculative LM trials) and, for generic scenes, D distance helpers. Closed-form Jacobians only.
// *_DEFAULTS: the two point-like kinds compiled with the configuration flags folded to the TebConfig defaults (teb_device.hpp: TEB_CFG).
// The generic-shape *_DEFAULTS kinds keep the kinematics flags at run time (-DTEB_AMD_PROFILE_ANY_KINEMATICS): polygon robots are car-like
// as often as not (BASELINE C5).
// *_WIDE (round 4): the two point-like kinds once more with every fold of the profile EXCEPT the via-points and the holonomic /
// non-holonomic choice of the velocity and acceleration edges (teb_device.hpp: TEB_PF_WIDE_*), for configurations that differ from the
// defaults in nothing else - a goal-directed planner with via-points, an omnidirectional base (-DTEB_AMD_PROFILE_WIDE).
enum { SCENE_POINTS = 0, SCENE_GENERIC = 1, SCENE_POINTS_SMALL = 2, SCENE_GENERIC_SMALL = 3, SCENE_POINTS_DEFAULTS = 4, SCENE_POINTS_SMALL_DEFAULTS = 5,
       SCENE_GENERIC_DEFAULTS = 6, SCENE_GENERIC_SMALL_DEFAULTS = 7, SCENE_POINTS_WIDE = 8, SCENE_POINTS_SMALL_WIDE = 9,
       SCENE_POINTS_LIGHT = 10, SCENE_POINTS_SMALL_LIGHT = 11,   // *_LIGHT: every cost-term flag at run time, only the never-reached bulk folded (TEB_PF_LIGHT_*)
       // *_CUSTOM: compiled at run time for the handle's configuration, every flag folded to its value there (teb_rtc.hpp, -DTEB_AMD_PROFILE_CUSTOM)
       SCENE_POINTS_CUSTOM = 12, SCENE_POINTS_SMALL_CUSTOM = 13, SCENE_GENERIC_CUSTOM = 14, SCENE_GENERIC_SMALL_CUSTOM = 15 };
template <int SOLVER, int JMODE, int SCENE>
__global__ void __launch_bounds__(kThreads)
teb_optimize_kernel(const teb_amd_config_t c, const SceneDev sc, const BatchDev bt, const OptArgs args,
                    const LdsPlan plan, const McuDev mc) {
  extern __shared__ __attribute__((aligned(16))) double lds_base[];
  constexpr bool FAST = SCENE == SCENE_POINTS || SCENE == SCENE_POINTS_SMALL || SCENE == SCENE_POINTS_DEFAULTS || SCENE == SCENE_POINTS_SMALL_DEFAULTS ||
                        SCENE == SCENE_POINTS_WIDE || SCENE == SCENE_POINTS_SMALL_WIDE || SCENE == SCENE_POINTS_LIGHT || SCENE == SCENE_POINTS_SMALL_LIGHT ||
                        SCENE == SCENE_POINTS_CUSTOM || SCENE == SCENE_POINTS_SMALL_CUSTOM;
  constexpr bool MCU = SCENE == SCENE_POINTS_SMALL || SCENE == SCENE_GENERIC_SMALL || SCENE == SCENE_POINTS_SMALL_DEFAULTS ||
                       SCENE == SCENE_GENERIC_SMALL_DEFAULTS || SCENE == SCENE_POINTS_SMALL_WIDE || SCENE == SCENE_POINTS_SMALL_LIGHT ||
                       SCENE == SCENE_POINTS_SMALL_CUSTOM || SCENE == SCENE_GENERIC_SMALL_CUSTOM;   // small-batch instantiation: helper workgroups possible
#ifdef TEB_AMD_DEFAULTS_PROFILE
  static_assert(SCENE >= SCENE_POINTS_DEFAULTS, "a unit compiled with the profile holds a *_DEFAULTS kind");
#else
  static_assert(SCENE < SCENE_POINTS_DEFAULTS, "*_DEFAULTS kinds need -DTEB_AMD_DEFAULTS_PROFILE");
#endif
#ifdef TEB_AMD_PROFILE_ANY_KINEMATICS
  static_assert(SCENE == SCENE_GENERIC_DEFAULTS || SCENE == SCENE_GENERIC_SMALL_DEFAULTS, "the generic-shape kinds of the profile keep the kinematics flags");
#endif
#ifdef TEB_AMD_PROFILE_CUSTOM
  static_assert(SCENE >= SCENE_POINTS_CUSTOM, "-DTEB_AMD_PROFILE_CUSTOM builds the *_CUSTOM kinds");
#else
  static_assert(SCENE < SCENE_POINTS_CUSTOM, "*_CUSTOM kinds are compiled at run time (teb_rtc.hpp)");
#endif
#ifdef TEB_AMD_PROFILE_WIDE
  static_assert(SCENE == SCENE_POINTS_WIDE || SCENE == SCENE_POINTS_SMALL_WIDE, "-DTEB_AMD_PROFILE_WIDE builds the *_WIDE kinds");
#else
  static_assert(SCENE != SCENE_POINTS_WIDE && SCENE != SCENE_POINTS_SMALL_WIDE, "*_WIDE kinds need -DTEB_AMD_PROFILE_WIDE");
#endif
#ifdef TEB_AMD_PROFILE_LIGHT
  static_assert(SCENE == SCENE_POINTS_LIGHT || SCENE == SCENE_POINTS_SMALL_LIGHT, "-DTEB_AMD_PROFILE_LIGHT builds the *_LIGHT kinds");
#else
  static_assert(SCENE != SCENE_POINTS_LIGHT && SCENE != SCENE_POINTS_SMALL_LIGHT, "*_LIGHT kinds need -DTEB_AMD_PROFILE_LIGHT");
#endif
  static_assert(!MCU || JMODE == TEB_AMD_JACOBIAN_ANALYTIC, "the small-batch kinds exist for closed-form Jacobians");
  if constexpr (MCU) {
    if (mc.K + mc.D > 0 && (int)blockIdx.x >= bt.B) {   // workgroups B .. B (1 + K + D) - 1: helpers of band (x - B) / (K + D)
      const int per = mc.K + mc.D;
      const int hb = ((int)blockIdx.x - bt.B) / per, j = ((int)blockIdx.x - bt.B) - hb * per;
      if (j < mc.K) {
        if constexpr (SOLVER != SOLVER_BANDG) mcu_solver_helper<SOLVER>(sc, bt, mc, plan, hb, j + 1);
      } else {
        if constexpr (!FAST) mcu_helper(c, sc, bt, mc, plan, hb, j - mc.K);
      }
      return;
    }
  }
  const int b = blockIdx.x, tid = threadIdx.x, S = bt.stride;
  if (b == 0 && tid == 0) { bt.clk[0] = clock64(); bt.clk[1] = wall_clock64(); }   // shader clock of this launch (a slow box is not a regression)
  const Lds l = carve(lds_base, plan, SOLVER == SOLVER_BANDG ? args.Hband + (size_t)b * args.hband_stride : nullptr, SOLVER == SOLVER_BANDG);
  McuMaster mm;
  mm.H = (MCU && !FAST) ? mc.D : 0; mm.K = (MCU && SOLVER != SOLVER_BANDG) ? mc.K : 0;
  mm.epoch = 0; mm.sepoch = 0; mm.failed = false; mm.spec_failed = false; mm.timeout = mc.timeout_ticks; mm.trace = mc.trace;
  mm.ctl = (mm.H > 0 || mm.K > 0) ? mc.ctl + (size_t)b * kMcuCtlWords : nullptr;
  mm.pub = mm.H > 0 ? mc.pub + (size_t)b * kMcuPubArrays * S : nullptr;
  mm.spec = mm.K > 0 ? mc.spec + (size_t)b * (mc.K + 1) * mcu_spec_slot(S) : nullptr;
  const bool mcu_on = MCU && mm.H > 0 && !TEB_CFGI(DEBUG_LINEARIZE) && !c.legacy_obstacle_association;
  const bool spec_on = MCU && mm.K > 0 && !TEB_CFGI(DEBUG_LINEARIZE) && !(SOLVER == SOLVER_BAND && TEB_CFGI(BAND_LDLT)) && !(mc.debug_flags & 4);
  if constexpr (FAST) {   // stage the point-like obstacle table once: static list first, then the dynamic list
    const int tot = sc.n_static + sc.n_dyn;
    for (int k = tid; k < tot; k += kThreads) {
      const int oi = (k < sc.n_static) ? sc.static_idx[k] : sc.dyn_idx[k - sc.n_static];
      l.obx[k] = sc.ax[oi]; l.oby[k] = sc.ay[oi]; l.obvx[k] = sc.vx[oi]; l.obvy[k] = sc.vy[oi];
      l.obr[k] = (sc.type[oi] == TEB_AMD_OBST_CIRCULAR) ? sc.rad[oi] : 0.0;
    }
  }
  int n = bt.n[b];
  const size_t so = (size_t)b * S;
  // Memory safety does not rest on the host's idea of the pose counts: a band longer than the LDS strips of THIS launch (an optimistic
  // layout chosen from a stale upper bound, counts written through teb_amd_device_state) is flagged like a band that outgrew the layout
  // during autoResize (the host repeats the launch in the handle's own layout) and leaves without touching the LDS.
  if (n > plan.S || n > S || n < 0) {
    if (tid == 0) {
      bt.assoc_overflow[b] = 2;
      bt.status[b] = TEB_AMD_TEB_FAILED; bt.iters[b] = 0; bt.last_iters[b] = 0; bt.trials[b] = 0;
      bt.chi2[b] = 0; bt.cost[b] = __longlong_as_double(0x7ff8000000000000LL); bt.lambda[b] = 0;
    }
    if (MCU && mm.H + mm.K > 0) mcu_exit(mm);   // the helpers of this band must not wait for a master that has left
    return;
  }

  // ---- K0: strip load, coalesced 8 B / lane
  for (int i = tid; i < n; i += kThreads) {
    l.sx[i] = bt.x[so + i]; l.sy[i] = bt.y[so + i]; l.sth[i] = bt.th[so + i];
    l.sdt[i] = (i < n - 1) ? bt.dt[so + i] : 0.0;
  }
  // this band's flags: no memset command per launch. Single-CU launches: only this workgroup touches them. Multi-CU mode: the helpers
  // may set bits too, so every access is an agent-scope atomic there (a plain store could be written back over a helper's bit later)
  if (tid == 0) { if (MCU && mm.H > 0) st_agent_i32(bt.assoc_overflow + b, 0); else bt.assoc_overflow[b] = 0; }
  __syncthreads();

  TebCtx t;
  t.b = b; t.stride = S;
  t.has_vs = bt.has_vs[b]; t.has_vg = bt.has_vg[b]; t.rotdir = bt.rotdir[b]; t.via_en = bt.via_en[b];
#pragma unroll
  for (int q = 0; q < 3; ++q) { t.vs[q] = bt.vs[3 * b + q]; t.vg[q] = bt.vg[3 * b + q]; }
  t.inflated = TEB_CFGI(INFLATED);
  int* assoc_cnt = bt.assoc_cnt + so;
  int* assoc = bt.assoc + (size_t)b * bt.assoc_cap * S;
  int* via_pose = bt.via_pose + (size_t)b * bt.via_cap;
  t.assoc_cnt = assoc_cnt; t.assoc = assoc; t.via_pose = via_pose; t.assoc_cap = bt.assoc_cap;
  t.mcu.items = nullptr; t.mcu.shared_lists = false;
  const double* mcu_items = mcu_on ? mc.items + (size_t)b * mc.item_cap * 4 * S : nullptr;
  double* Hbk = bt.Hbackup + (size_t)b * bt.hmat_stride;

  int status = TEB_AMD_TEB_OK, iters = 0, last_iters = 0, trials = 0;
  int optimized = c.optimization_activate ? 0 : bt.optimized[b];   // optimized_ = false (src/optimal_planner.cpp:189), after the early return
  double chi2_final = 0, lambda = 0, cost = __longlong_as_double(0x7ff8000000000000LL);
  double last_cats[4] = {0, 0, 0, 0};
  double weight_multiplier = TEB_CFGI(DEBUG_LINEARIZE) ? args.debug_weight_multiplier : 1.0;
  NearCache near_cache;   // near masks of the dynamic-obstacle edges, per lane (dyn_near_cached)
  near_cache.invalidate();
  near_cache.off = TEB_CFGI(NO_NEAR_CACHE);
  const bool fast_mode = !c.include_dynamic_obstacles;
  bool done = false;
  bool trig_stale = true;   // the cos / sin cache (l.cs, l.sn) does not match the headings (uniform)
  PROF_DECL
#ifdef TEB_PROFILE
  if (threadIdx.x == 0) { l.ired[12] = 0; l.ired[13] = 0; }
#endif

  if (!c.optimization_activate) { status = TEB_AMD_TEB_FAILED; done = true; }

  for (int outer = 0; outer < args.outer && !done; ++outer) {
    // ---- K1: autoResize
    if (c.teb_autosize && !TEB_CFGI(DEBUG_LINEARIZE)) {
      int ovf = 0;
      PROF_START();
      // edit script + new poses + split stack (autoresize_scratch_doubles: 5 S + 104 doubles) live in the LDS region of the normal matrix, which is rebuilt afterwards
      // (wave-uniform copy: the new pose count comes back out of LDS, i.e. in a VGPR, and every loop bound of the LM loop would be vector
      //  arithmetic and every loop an exec-mask loop from here on; all lanes hold the same value)
      n = uni_i(autoresize(c, l, n, plan.off_state, plan.off_H, plan.S, fast_mode, &ovf));   // plan.S: LDS strip spacing = pose capacity of this launch
      ovf = uni_i(ovf);
      PROF_END(0);
      if (ovf) { status = TEB_AMD_TEB_FAILED; if (tid == 0) { if (MCU && mm.H > 0) or_agent_i32(bt.assoc_overflow + b, 2); else bt.assoc_overflow[b] |= 2; } break; }
    }
    t.n = n;
    // optimizeGraph guards (src/optimal_planner.cpp:370-382)
    if (c.max_vel_x < 0.01 || n < 2 || n < c.min_samples) { status = TEB_AMD_TEB_FAILED; break; }
    // ---- buildGraph side data: association (K2), dynamic-obstacle time stamps, via-point attachment
    t.w_obst = c.weight_obstacle * weight_multiplier;
    PROF_START();
    LNP_DECL
    refresh_trig(l, n);
    trig_stale = false;
    __syncthreads();
    LNP(8);
    const bool obst_edges = !(c.weight_obstacle == 0 || weight_multiplier == 0);
    if (obst_edges) {
      int ovf = 0;
      if (!TEB_CFGI(NEW_ASSOCIATION))
        associate_legacy(c, sc, l, n, assoc_cnt, assoc, bt.assoc_cap, S, &ovf, bt.legacy_idx + (size_t)b * bt.assoc_cap);
      else if (mcu_on && (mc.debug_flags & 1)) {   // (diagnostic) this workgroup scans, the helpers of the DIST phases read the lists
        associate_range<FAST, true>(c, sc, l, n, 0, n, assoc_cnt, assoc, bt.assoc_cap, S, &ovf);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        t.mcu.shared_lists = true;
      } else if (mcu_on) {   // the helpers scan pose tiles; the lists come back through HBM (agent-scope accesses from here on)
        mcu_publish(mm, l.sx, l.sy, l.cs, l.sn, l.tdyn, n, S);
        mcu_issue(mm, MCU_KIND_ASSOC, n);
        t.mcu.shared_lists = true;
        if (!mcu_wait(mm, l.ired + 26)) { status = TEB_AMD_TEB_FAILED; if (tid == 0) or_agent_i32(bt.assoc_overflow + b, 4); break; }
      } else
        associate<FAST>(c, sc, l, n, assoc_cnt, assoc, bt.assoc_cap, S, &ovf);
      if (ovf && tid < kThreads) { if (mcu_on) { if (tid == 0) or_agent_i32(bt.assoc_overflow + b, 1); } else bt.assoc_overflow[b] |= 1; }
    } else {
      for (int i = tid; i < n; i += kThreads) assoc_cnt[i] = 0;
      t.mcu.shared_lists = false;
    }
    LNP(9);
    if (tid == 0) {   // :662-670, sequential left-to-right sum like the reference
      double time = l.sdt[0];
      int i = 1;
      for (; i + 8 <= n - 1; i += 8) {   // the loads of 8 intervals in flight at once; the additions keep their order
        double d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) d[u] = l.sdt[i + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) { l.tdyn[i + u] = time; time += d[u]; }
      }
      for (; i < n - 1; ++i) { l.tdyn[i] = time; time += l.sdt[i]; }
    }
    LNP(10);
    if (TEB_CFGI(VIA_POINTS) && t.via_en && n >= 3) {   // :675-718
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
    LNP(11);
    PROF_END(1);

    near_cache.invalidate();   // the graph was rebuilt: new pose numbering, new time stamps
    // ---- optimize(): Levenberg-Marquardt
    if (args.inner <= 0 && !TEB_CFGI(DEBUG_LINEARIZE)) { status = TEB_AMD_TEB_FAILED; break; }   // optimize(0) returns 0
    // multi-CU mode: the distance records of the freshly built graph for the first linearisation; the later ones find the records
    // of the error evaluation that accepted their state
    t.mcu.items = nullptr;
    if (mcu_on && obst_edges && !(mc.debug_flags & 2)) {
      mcu_publish(mm, l.sx, l.sy, l.cs, l.sn, l.tdyn, n, S);   // (refresh_trig above: cos / sin are those of the current headings)
      mcu_issue(mm, MCU_KIND_DIST, n);
      if (!mcu_wait(mm, l.ired + 26)) { status = TEB_AMD_TEB_FAILED; if (tid == 0) or_agent_i32(bt.assoc_overflow + b, 4); break; }
      t.mcu.items = mcu_items;
    }
    double ni = 2;
    bool lm_ok = true;
    last_iters = 0;   // SparseOptimizer::optimize clears the batch statistics
    for (int it = 0; it < args.inner && lm_ok; ++it) {
      double cats[4];
      PROF_START();
      // cos / sin of the headings are current here: refreshed after autoResize (above), by the update step of the accepted trial, or, after
      // a rejected last trial, never needed again in this optimize() (the loop ends)
      linearize<SOLVER, JMODE, FAST>(c, sc, t, l, near_cache, cats, !trig_stale);
      trig_stale = false;
      PROF_END(2);
      double currentChi = ((cats[0] + cats[1]) + cats[2]) + cats[3];
      if (TEB_CFGI(DEBUG_LINEARIZE)) {
        if (b == 0) {
          const int Nt = 4 * n;
          for (int q = tid; q < Nt * kBand; q += kThreads) {   // always exported in band form
            const int r = q / kBand, d = q % kBand, cc = r - d;
            double v = 0;
            if (cc >= 0) {
              if (SOLVER != SOLVER_CR) v = l.Hb[hbo(r) + d];
              else if ((r >> 3) == (cc >> 3)) v = l.Db[(r >> 3) * kBlk + (r & 7) * 8 + (cc & 7)];
              else if ((r >> 3) == (cc >> 3) + 1) v = l.Lb[(r >> 3) * kBlk + (r & 7) * 8 + (cc & 7)];
            }
            args.dbg_H[q] = v;
          }
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
          if (r >= 3 && r < 4 * (n - 1)) m = fmax(m, fabs(*diag_ptr<SOLVER>(l, r)));
        lambda = 1e-5 * block_max(m, l.red);
        ni = 2;
      }
      PROF_START();
      const int hsz = (SOLVER != SOLVER_CR) ? hbo(Nt) : ((Nt + 7) >> 3) * 2 * kBlk;
      const bool keep_copy = !(SOLVER != SOLVER_CR && !TEB_CFGI(BAND_LDLT));   // the HBM-block reductions never touch the band
      if constexpr (MCU) {
        if (spec_on) spec_wait_idle(mm, l.ired + 26);   // the solver helpers are done with the buffers of the previous iteration
      }
      const bool spec_now = MCU && spec_on && !mm.spec_failed;
#ifndef TEB_AMD_DIAG_EDGE_ONLY
      if (keep_copy) {
        if (spec_now) {   // (the solver helpers load the same backup: written through)
          for (int q = tid; q < hsz; q += kThreads) st_agent_f64(Hbk + q, *hmat_ptr<SOLVER>(l, q, Nt));
        } else {
          hmat_save<SOLVER>(l, hsz, Nt, Hbk);   // saved for rejected trials
        }
      }
      if (SOLVER == SOLVER_BAND && !TEB_CFGI(BAND_LDLT)) cr_copy_band(l, n, Hbk, spec_now);   // hybrid solve: the band to HBM once per iteration
#endif
      if constexpr (MCU) {
        if (spec_now) spec_issue(mm, l.bv, Nt, n, S, lambda, ni);   // retries 1 .. K start on their CUs now
      }
      PROF_END(3);
      double rho = 0;
      int qmax = 0;
      bool h_spent = false;   // blocks-in-LDS layout: the in-place solve has consumed H
      do {
        // --- damped solve
        PROF_START();
        bool taken = false;
        if constexpr (MCU) {
          if (spec_now && !mm.spec_failed && qmax >= 1 && qmax <= mm.K)
            taken = spec_take(mm, qmax, l.dxv, l.ired, Nt, S, l.ired + 26);   // the step of this retry was solved on a spare CU meanwhile
        }
#ifdef TEB_AMD_DIAG_EDGE_ONLY
        // Diagnostic build (tools/profile.sh, roofline.edge_evaluation.traffic of the bench line): the damped solve, its band copy and the
        // H backup are left out - a Jacobi step dx_r = b_r / (H_rr + lambda) stands in, so that the LM loop keeps linearising and evaluating -
        // and what the FETCH_SIZE / WRITE_SIZE counters then see is the HBM traffic of the edge phases alone. Never the product.
        taken = true;
        for (int r = tid; r < Nt; r += kThreads) l.dxv[r] = (r >= 3 && r < 4 * (n - 1)) ? l.bv[r] / (*diag_ptr<SOLVER>(l, r) + lambda) : 0.0;
        if (tid == 0) l.ired[0] = 1;
        __syncthreads();
#endif
        if (!taken) {
        if (keep_copy && h_spent) {   // bring back the un-factored H (lazily: a retry whose step came from a helper needs no H at all)
          hmat_load<SOLVER>(l, hsz, Nt, Hbk);
          __syncthreads();
        }
        h_spent = true;
        if constexpr (SOLVER == SOLVER_BAND) {
          if (TEB_CFGI(BAND_LDLT)) {
            if (tid < 64) {
              bool ok = banded_ldlt_solve_wave0(l, Nt, lambda);
              if (tid == 0) l.ired[0] = ok ? 1 : 0;
            }
            __syncthreads();
          } else {
            cr_solve_hybrid(plan, n, lambda, Hbk);
          }
        } else if constexpr (SOLVER == SOLVER_BANDG) {
          cr_solve_t<true, true>(plan, sc, n, lambda, Hbk, l.Hb);
        } else {
          cr_solve_t<false, false>(plan, sc, n, lambda, nullptr, nullptr);
        }
        }   // !taken
        PROF_END(4);
        PROF_START();
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
              const double th_new = normalize_theta(l.sth[i] + l.dxv[4 * i + 2]);
              l.sth[i] = th_new;
              // the cos / sin cache follows the heading right here (same sincos of the same value as refresh_trig would compute: same
              // bits), under the latency of the rest of the update, instead of in a pass of its own in front of every error evaluation
              // and every linearisation
              double sv, cv;
              sincos(th_new, &sv, &cv);
              l.cs[i] = cv; l.sn[i] = sv;
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
        trig_stale = false;   // (the update step above brought the cache along)
        double tc[5];
        tc[4] = sc_part;
        if (MCU && t.mcu.items != nullptr) {
          if (!evaluate_mcu(c, sc, t, l, near_cache, mm, S, tc)) { status = TEB_AMD_TEB_FAILED; if (tid == 0) or_agent_i32(bt.assoc_overflow + b, 4); done = true; lm_ok = false; break; }
        } else
          evaluate<FAST>(c, sc, t, l, near_cache, tc, true);
        last_cats[0] = tc[0]; last_cats[1] = tc[1]; last_cats[2] = tc[2]; last_cats[3] = tc[3];
        double tempChi = ((tc[0] + tc[1]) + tc[2]) + tc[3];
        double scv[1] = {tc[4]};
        PROF_END(5);
        PROF_START();
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
          trig_stale = true;   // the cache holds cos / sin of the rejected headings; the next update step (or refresh) overwrites them
          if (!isfinite(lambda)) { ++qmax; __syncthreads(); break; }
        }
        __syncthreads();
        PROF_END(6);
        qmax++;
      } while (rho < 0 && qmax < 10);
      if (args.iter_log && tid == 0 && iters < args.iter_log_cap) {   // "iteration= i chi2= .. lambda= .. levenbergIter= .." of g2o's verbose mode
        double* row = args.iter_log + ((size_t)b * args.iter_log_cap + iters) * 4;
        row[0] = currentChi; row[1] = lambda; row[2] = (double)qmax; row[3] = (double)n;
      }
      ++iters;
      ++last_iters;
      chi2_final = currentChi;
      if (qmax == 10 || rho == 0 || !isfinite(lambda)) lm_ok = false;   // Terminate
      if (TEB_CFGI(DIVERGENCE_DETECTION)) {   // setComputeBatchStatistics -> computeActiveErrors after each solve
        double fc[5];
        fc[4] = 0;
        if (MCU && t.mcu.items != nullptr) {
          if (!evaluate_mcu(c, sc, t, l, near_cache, mm, S, fc)) { status = TEB_AMD_TEB_FAILED; if (tid == 0) or_agent_i32(bt.assoc_overflow + b, 4); done = true; break; }
        } else
          evaluate<FAST>(c, sc, t, l, near_cache, fc, !trig_stale);
          trig_stale = false;
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
    optimized = 1;   // :220 (set before computeCurrentCost there; nothing in between reads it)
    weight_multiplier *= c.weight_adapt_factor;
  }

  // ---- K0: strip store + results
  if (MCU && mm.H + mm.K > 0) mcu_exit(mm);   // every way out of the loops above ends here: the helpers leave
  __syncthreads();
  int nonfinite = 0;
  for (int i = tid; i < n; i += kThreads) {
    double x = l.sx[i], y = l.sy[i], th = l.sth[i], d = (i < n - 1) ? l.sdt[i] : 0.0;
    bt.x[so + i] = x; bt.y[so + i] = y; bt.th[so + i] = th; bt.dt[so + i] = d;
    if (!(isfinite(x) && isfinite(y) && isfinite(th) && isfinite(d))) nonfinite = 1;
  }
  nonfinite = __syncthreads_or(nonfinite);
#ifdef TEB_PROFILE
  if (tid == 0 && args.dbg_H) args.dbg_H[16 + b] = (double)(clock64() - prof_wg_t0);   // whole-kernel cycles of this band's workgroup
  if (tid == 0 && b == 0 && args.dbg_H) {
    prof_acc[7] = (long long)l.ired[12] + 1000000000LL * l.ired[13];   // autoResize: cycles inside the sequential sweeps + 1e9 * #sweeps
    for (int q = 0; q < 8; ++q) args.dbg_H[q] = (double)prof_acc[q];
  }
#endif
#ifndef TEB_PROFILE
  if (plog_ && tid == 0) {
    double* out = args.phase_log + (size_t)b * kPhaseLogSlots;
    for (int q = 0; q < 8; ++q) out[q] = (double)plog_acc_[q];
    out[8] = (double)(clock64() - plog_acc_[9]);
  }
#endif
  if (tid == 0) {
    if (b == 0) { bt.clk[2] = clock64(); bt.clk[3] = wall_clock64(); }
    if (nonfinite) status = TEB_AMD_TEB_NONFINITE;
    bt.n[b] = n;
    bt.status[b] = status; bt.iters[b] = iters; bt.last_iters[b] = last_iters; bt.trials[b] = trials; bt.optimized[b] = optimized;
    bt.chi2[b] = chi2_final; bt.cost[b] = cost; bt.lambda[b] = lambda;
  }
}

}  // namespace tebamd
