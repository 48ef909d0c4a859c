// teb_device.hpp — device-side data model of the MI355X TEB optimiser (gfx950 only).
//
// HBM layout (all fp64 unless noted):
//   state strips   x,y,theta,dt : [B][stride]  SoA, one contiguous row per candidate TEB (coalesced 8 B/lane)
//   obstacle table              : SoA columns [M] (type,ax,ay,bx,by,radius,vx,vy,cx,cy,dynamic) + CSR polygon
//                                 vertices; read-only, shared by all workgroups (L2 resident)
//   association lists           : int32 [B][assoc_cap][stride]  (entry k of pose i at [(b*cap+k)*stride+i] so
//                                 that lane i reads entry k coalesced)
//   H backup                    : [B][4*stride*11] banded normal matrix saved once per LM iteration so a
//                                 rejected damping trial does not have to re-linearise
// LDS layout per workgroup (one workgroup = one TEB): see teb_kernel.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/teb_amd.h"

// ---- instantiations specialised on the configuration ------------------------------------------------------------------------------------
// Every cost term sits behind run-time flags of teb_amd_config_t (holonomic or not, which weights are zero, car-like or diff-drive, ..):
// uniform branches, but each one ends a basic block, so the sqrt / divide chains of neighbouring terms cannot be scheduled into each
// other's latency - and at one wave per SIMD that latency is all there is to fill. A translation unit compiled with
// -DTEB_AMD_DEFAULTS_PROFILE (teb_opt_inst.hip does it for the scene kinds *_DEFAULTS) folds those flags to the values they have in a
// default TebConfig: TEB_CFG(condition, value under the profile). The host launches such a kernel only when the handle's configuration
// satisfies every folded condition (config_matches_defaults_profile, teb_amd.hip); anything else runs the generic instantiation. Same
// operations in the same order on the taken paths: bit-identical bands (tests/test_gpu_config_profile.py).
#ifdef TEB_AMD_DEFAULTS_PROFILE
#define TEB_CFG(expr, dflt) (dflt)
#else
#define TEB_CFG(expr, dflt) (expr)
#endif

namespace tebamd {

constexpr int kThreads = 256;   // 4 wave64 per workgroup, one workgroup per candidate TEB (512 = two waves per SIMD: measured in round 2, 1100 - 1600 spill slots, 6.9 vs 4.8 ms)
constexpr int kWaves = kThreads / 64;
constexpr int kBand = 11;           // diagonal + scalar half-bandwidth 10
constexpr int kMaxPoseIter = 2;     // poses handled per thread: n <= kThreads * kMaxPoseIter

// LDS layout of one workgroup, computed on the host (offsets in doubles from the dynamic-LDS base).
struct LdsPlan {
  int S;            // pose capacity
  int solver;       // SOLVER_BAND / SOLVER_CR / SOLVER_BANDG
  int off_state;    // sx sy sth sdt tdyn cs sn : 7*S
  int off_H;        // normal matrix (band or blocks)
  int off_b;        // 4S+8
  int off_dx;       // 4S+8
  int off_red;      // 64 doubles + 64 ints
  int off_ob;       // obstacle cache: x y vx vy r, each ob_cap entries (point-like scenes only), or -1
  int ob_cap;
  int total_bytes;
};

struct SceneDev {
  int M;
  int fast_points;         // 1: every obstacle is Point/Circular, the footprint is Point/Circular and the
                           //    obstacle cache fits the LDS -> specialised distance path on LDS-resident data
  int static_radius_zero;  // 1: no obstacle of the static list has a radius (the far-field threshold of the association is one number)
  const int* type;
  const double *ax, *ay, *bx, *by, *rad, *vx, *vy, *cx, *cy;
  const double* brad;      // radius of a circle about the centroid (cx, cy) that contains the obstacle (far-field culling)
  const int* dyn;
  const int* voff;
  const double *pvx, *pvy;
  int n_static;            // obstacles visited by AddEdgesObstacles (all non-dynamic, or all if !include_dynamic)
  const int* static_idx;
  int n_dyn;               // obstacles visited by AddEdgesDynamicObstacles
  const int* dyn_idx;
  int nvia;
  const double *viax, *viay;
  // point-like scenes: x, y, radius (0 for points), velocity of the obstacles in the order of the LDS cache (static list, then dynamic
  // list). The far-field passes read them with wave-uniform indices, i.e. as scalar loads (no LDS traffic, no VGPR)
  const double *lox, *loy, *lor, *lovx, *lovy;
};

constexpr int kAssocTriple = 1 << 30;   // legacy association adds the edge at the closest pose three times (:583-641)
constexpr int kAssocMask = kAssocTriple - 1;

struct BatchDev {
  int B, stride;
  int* n;
  double *x, *y, *th, *dt;
  const int* has_vs;
  const double* vs;
  const int* has_vg;
  const double* vg;
  const int* rotdir;
  const int* via_en;
  // results
  int* status;
  int* optimized;   // TebOptimalPlanner::optimized_ (optimal_planner.h:691): an outer iteration of the last optimizeTEB call completed
  int* iters;
  int* trials;
  double* chi2;
  double* cost;
  double* lambda;
  // scratch
  int* assoc_cnt;   // [B][stride]
  int* assoc;       // [B][assoc_cap][stride]; entry = position in the static list, | kAssocTriple for 3 identical edges
  int assoc_cap;
  int* assoc_overflow;  // [B]
  int* legacy_idx;  // [B][assoc_cap] closest pose per static obstacle (legacy association only)
  int* via_pose;    // [B][via_cap]
  int via_cap;
  double* Hbackup;  // [B][hmat_stride]
  size_t hmat_stride;
  long long* clk;   // [4] workgroup 0: shader-clock counter and 100 MHz real-time counter at kernel entry, then at exit (teb_amd_last_shader_clock_mhz)
};

struct OptArgs {
  int inner, outer, compute_cost;
  double obst_scale, via_scale;
  int alt_time;
  double* Hband;         // SOLVER_BANDG: per-band normal matrix in band form in HBM, hband_stride doubles each
  size_t hband_stride;
  int band_ldlt;         // SOLVER_BAND only: 1 = sequential banded LDL^T in LDS (v1), 0 = hybrid cyclic reduction (cr_solve_hybrid)
  int no_near_cache;     // teb_amd_options_t::no_near_cache
  double* iter_log;      // opt-in (teb_amd_set_iteration_log): [B][iter_log_cap][4] = per LM iteration chi2, lambda, trials, pose count -
  int iter_log_cap;      //   what g2o prints per iteration with setVerbose(optimization_verbose) (src/optimal_planner.cpp:384); else nullptr
  int debug_linearize;   // test hook: stop after the first linearisation and dump H, b, chi2 categories
  double debug_weight_multiplier;
  double* dbg_H;         // [4*stride*kBand] of TEB 0
  double* dbg_b;         // [4*stride]
  double* dbg_chi2;      // [4]
};

}  // namespace tebamd
