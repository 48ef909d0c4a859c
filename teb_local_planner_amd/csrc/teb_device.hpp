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
// -DTEB_AMD_DEFAULTS_PROFILE (teb_opt_inst.hip does it for the scene kinds *_DEFAULTS, *_WIDE and *_LIGHT) folds those flags to the values they
// have in a default TebConfig. Same operations in the same order on the taken paths: bit-identical bands
// (tests/test_gpu_config_profile.py).
//
// ONE TABLE says what is folded (round 4; before, the host kept a hand-written mirror of the TEB_CFG sites). Per flag ID:
//   TEB_PF_EXPR_<ID>   the run-time condition, over `c` (teb_amd_config_t), `args` (OptArgs), `sc` (SceneDev) - whichever the site has
//   TEB_PF_DFLT_<ID>   its value under the profile (what the device code is folded to)
//   TEB_PF_HOST_<ID>   when the fold is VALID, evaluated on the host with the same three names in scope (normally EXPR == DFLT)
//   TEB_PF_WIDE_<ID>   1: the flag stays a run-time flag in the *_WIDE kinds (-DTEB_AMD_PROFILE_WIDE): via-points and holonomic robots
//                      run kernels that keep every other fold
//   TEB_PF_LIGHT_<ID>  1: the flag stays a run-time flag in the *_LIGHT kinds (-DTEB_AMD_PROFILE_LIGHT): every cost-term flag does; what
//                      those kinds fold is the bulk a planner configuration never reaches (legacy association, debug export, the
//                      sequential LDL^T, the uncached near masks, the second error evaluation of the divergence detection)
//   TEB_PF_KIN_<ID>    1: folded in the point-like kinds only (the generic-shape kinds keep diff-drive / car-like at run time,
//                      -DTEB_AMD_PROFILE_ANY_KINEMATICS)
// A device site writes TEB_CFGI(ID); the host's profile_matches() (teb_amd.hip) is generated from TEB_PF_ALL over the same entries.
#define TEB_PF_ALL(X)                                                                                                              \
  X(EXACT_ARC) X(COST_EXPONENT) X(NEW_ASSOCIATION) X(DYNAMIC_EDGES) X(VIA_POINTS) X(NONHOLONOMIC_VELOCITY) X(VELOCITY_EDGES)      \
  X(ACCELERATION_EDGES) X(NONHOLONOMIC_ACCELERATION) X(TIME_OPTIMAL) X(SHORTEST_PATH) X(VELOCITY_OBSTACLE_RATIO) X(RADIUS_FREE)   \
  X(DEBUG_LINEARIZE) X(BAND_LDLT) X(INFLATED) X(NO_NEAR_CACHE) X(DIVERGENCE_DETECTION) X(KIN_DIFF_DRIVE) X(KIN_EDGES)
#define TEB_PF_EXPR_EXACT_ARC (c.exact_arc_length)
#define TEB_PF_DFLT_EXACT_ARC false
#define TEB_PF_EXPR_COST_EXPONENT (c.obstacle_cost_exponent != 1.0 && c.min_obstacle_dist > 0.0)
#define TEB_PF_DFLT_COST_EXPONENT false
#define TEB_PF_EXPR_NEW_ASSOCIATION (!c.legacy_obstacle_association)
#define TEB_PF_DFLT_NEW_ASSOCIATION true
#define TEB_PF_EXPR_DYNAMIC_EDGES (c.include_dynamic_obstacles && c.weight_obstacle != 0)
#define TEB_PF_DFLT_DYNAMIC_EDGES true
#define TEB_PF_HOST_DYNAMIC_EDGES (c.weight_obstacle != 0)   /* without include_dynamic_obstacles the dynamic list is empty: the folded loop runs over nothing */
#define TEB_PF_EXPR_VIA_POINTS (sc.nvia > 0 && c.weight_viapoint != 0)
#define TEB_PF_DFLT_VIA_POINTS false
#define TEB_PF_WIDE_VIA_POINTS 1
#define TEB_PF_EXPR_NONHOLONOMIC_VELOCITY (c.max_vel_y == 0)
#define TEB_PF_DFLT_NONHOLONOMIC_VELOCITY true
#define TEB_PF_WIDE_NONHOLONOMIC_VELOCITY 1
#define TEB_PF_EXPR_VELOCITY_EDGES (!(c.weight_max_vel_x == 0 && c.weight_max_vel_theta == 0))
#define TEB_PF_DFLT_VELOCITY_EDGES true
#define TEB_PF_EXPR_ACCELERATION_EDGES (!(c.weight_acc_lim_x == 0 && c.weight_acc_lim_theta == 0))
#define TEB_PF_DFLT_ACCELERATION_EDGES true
#define TEB_PF_EXPR_NONHOLONOMIC_ACCELERATION (c.max_vel_y == 0 || c.acc_lim_y == 0)
#define TEB_PF_DFLT_NONHOLONOMIC_ACCELERATION true
#define TEB_PF_WIDE_NONHOLONOMIC_ACCELERATION 1
#define TEB_PF_EXPR_TIME_OPTIMAL (c.weight_optimaltime != 0)
#define TEB_PF_DFLT_TIME_OPTIMAL true
#define TEB_PF_EXPR_SHORTEST_PATH (c.weight_shortest_path != 0)
#define TEB_PF_DFLT_SHORTEST_PATH false
#define TEB_PF_EXPR_VELOCITY_OBSTACLE_RATIO (c.weight_velocity_obstacle_ratio > 0)
#define TEB_PF_DFLT_VELOCITY_OBSTACLE_RATIO false
#define TEB_PF_EXPR_RADIUS_FREE (sc.static_radius_zero != 0)
#define TEB_PF_DFLT_RADIUS_FREE true
#define TEB_PF_KIN_RADIUS_FREE 1   /* (a property of point-like scenes) */
#define TEB_PF_EXPR_DEBUG_LINEARIZE (args.debug_linearize != 0)
#define TEB_PF_DFLT_DEBUG_LINEARIZE false
#define TEB_PF_EXPR_BAND_LDLT (args.band_ldlt != 0)
#define TEB_PF_DFLT_BAND_LDLT false
#define TEB_PF_EXPR_INFLATED (c.inflation_dist > c.min_obstacle_dist)
#define TEB_PF_DFLT_INFLATED true
#define TEB_PF_EXPR_NO_NEAR_CACHE (args.no_near_cache != 0)
#define TEB_PF_DFLT_NO_NEAR_CACHE false
#define TEB_PF_EXPR_DIVERGENCE_DETECTION (c.divergence_detection_enable)
#define TEB_PF_DFLT_DIVERGENCE_DETECTION false
#define TEB_PF_EXPR_KIN_DIFF_DRIVE (c.min_turning_radius == 0 || c.weight_kinematics_turning_radius == 0)
#define TEB_PF_DFLT_KIN_DIFF_DRIVE true
#define TEB_PF_KIN_KIN_DIFF_DRIVE 1
#define TEB_PF_EXPR_KIN_EDGES (!(c.weight_kinematics_nh == 0 && c.weight_kinematics_forward_drive == 0))
#define TEB_PF_DFLT_KIN_EDGES true
#define TEB_PF_KIN_KIN_EDGES 1
// defaults of the optional columns
#ifndef TEB_PF_HOST_EXACT_ARC
#define TEB_PF_HOST_EXACT_ARC (TEB_PF_EXPR_EXACT_ARC == TEB_PF_DFLT_EXACT_ARC)
#endif
#define TEB_PF_HOST_COST_EXPONENT (TEB_PF_EXPR_COST_EXPONENT == TEB_PF_DFLT_COST_EXPONENT)
#define TEB_PF_HOST_NEW_ASSOCIATION (TEB_PF_EXPR_NEW_ASSOCIATION == TEB_PF_DFLT_NEW_ASSOCIATION)
#define TEB_PF_HOST_VIA_POINTS (TEB_PF_EXPR_VIA_POINTS == TEB_PF_DFLT_VIA_POINTS)
#define TEB_PF_HOST_NONHOLONOMIC_VELOCITY (TEB_PF_EXPR_NONHOLONOMIC_VELOCITY == TEB_PF_DFLT_NONHOLONOMIC_VELOCITY)
#define TEB_PF_HOST_VELOCITY_EDGES (TEB_PF_EXPR_VELOCITY_EDGES == TEB_PF_DFLT_VELOCITY_EDGES)
#define TEB_PF_HOST_ACCELERATION_EDGES (TEB_PF_EXPR_ACCELERATION_EDGES == TEB_PF_DFLT_ACCELERATION_EDGES)
#define TEB_PF_HOST_NONHOLONOMIC_ACCELERATION (TEB_PF_EXPR_NONHOLONOMIC_ACCELERATION == TEB_PF_DFLT_NONHOLONOMIC_ACCELERATION)
#define TEB_PF_HOST_TIME_OPTIMAL (TEB_PF_EXPR_TIME_OPTIMAL == TEB_PF_DFLT_TIME_OPTIMAL)
#define TEB_PF_HOST_SHORTEST_PATH (TEB_PF_EXPR_SHORTEST_PATH == TEB_PF_DFLT_SHORTEST_PATH)
#define TEB_PF_HOST_VELOCITY_OBSTACLE_RATIO (TEB_PF_EXPR_VELOCITY_OBSTACLE_RATIO == TEB_PF_DFLT_VELOCITY_OBSTACLE_RATIO)
#define TEB_PF_HOST_RADIUS_FREE (TEB_PF_EXPR_RADIUS_FREE == TEB_PF_DFLT_RADIUS_FREE)
#define TEB_PF_HOST_DEBUG_LINEARIZE (TEB_PF_EXPR_DEBUG_LINEARIZE == TEB_PF_DFLT_DEBUG_LINEARIZE)
#define TEB_PF_HOST_BAND_LDLT (TEB_PF_EXPR_BAND_LDLT == TEB_PF_DFLT_BAND_LDLT)
#define TEB_PF_HOST_INFLATED (TEB_PF_EXPR_INFLATED == TEB_PF_DFLT_INFLATED)
#define TEB_PF_HOST_NO_NEAR_CACHE (TEB_PF_EXPR_NO_NEAR_CACHE == TEB_PF_DFLT_NO_NEAR_CACHE)
#define TEB_PF_HOST_DIVERGENCE_DETECTION (TEB_PF_EXPR_DIVERGENCE_DETECTION == TEB_PF_DFLT_DIVERGENCE_DETECTION)
#define TEB_PF_HOST_KIN_DIFF_DRIVE (TEB_PF_EXPR_KIN_DIFF_DRIVE == TEB_PF_DFLT_KIN_DIFF_DRIVE)
#define TEB_PF_HOST_KIN_EDGES (TEB_PF_EXPR_KIN_EDGES == TEB_PF_DFLT_KIN_EDGES)
#define TEB_PF_WIDE_EXACT_ARC 0
#define TEB_PF_WIDE_COST_EXPONENT 0
#define TEB_PF_WIDE_NEW_ASSOCIATION 0
#define TEB_PF_WIDE_DYNAMIC_EDGES 0
#define TEB_PF_WIDE_VELOCITY_EDGES 0
#define TEB_PF_WIDE_ACCELERATION_EDGES 0
#define TEB_PF_WIDE_TIME_OPTIMAL 0
#define TEB_PF_WIDE_SHORTEST_PATH 0
#define TEB_PF_WIDE_VELOCITY_OBSTACLE_RATIO 0
#define TEB_PF_WIDE_RADIUS_FREE 0
#define TEB_PF_WIDE_DEBUG_LINEARIZE 0
#define TEB_PF_WIDE_BAND_LDLT 0
#define TEB_PF_WIDE_INFLATED 0
#define TEB_PF_WIDE_NO_NEAR_CACHE 0
#define TEB_PF_WIDE_DIVERGENCE_DETECTION 0
#define TEB_PF_WIDE_KIN_DIFF_DRIVE 0
#define TEB_PF_WIDE_KIN_EDGES 0
#define TEB_PF_KIN_EXACT_ARC 0
#define TEB_PF_KIN_COST_EXPONENT 0
#define TEB_PF_KIN_NEW_ASSOCIATION 0
#define TEB_PF_KIN_DYNAMIC_EDGES 0
#define TEB_PF_KIN_VIA_POINTS 0
#define TEB_PF_KIN_NONHOLONOMIC_VELOCITY 0
#define TEB_PF_KIN_VELOCITY_EDGES 0
#define TEB_PF_KIN_ACCELERATION_EDGES 0
#define TEB_PF_KIN_NONHOLONOMIC_ACCELERATION 0
#define TEB_PF_KIN_TIME_OPTIMAL 0
#define TEB_PF_KIN_SHORTEST_PATH 0
#define TEB_PF_KIN_VELOCITY_OBSTACLE_RATIO 0
#define TEB_PF_KIN_DEBUG_LINEARIZE 0
#define TEB_PF_KIN_BAND_LDLT 0
#define TEB_PF_KIN_INFLATED 0
#define TEB_PF_KIN_NO_NEAR_CACHE 0
#define TEB_PF_KIN_DIVERGENCE_DETECTION 0

#define TEB_PF_LIGHT_EXACT_ARC 1
#define TEB_PF_LIGHT_COST_EXPONENT 1
#define TEB_PF_LIGHT_NEW_ASSOCIATION 0
#define TEB_PF_LIGHT_DYNAMIC_EDGES 1
#define TEB_PF_LIGHT_VIA_POINTS 1
#define TEB_PF_LIGHT_NONHOLONOMIC_VELOCITY 1
#define TEB_PF_LIGHT_VELOCITY_EDGES 1
#define TEB_PF_LIGHT_ACCELERATION_EDGES 1
#define TEB_PF_LIGHT_NONHOLONOMIC_ACCELERATION 1
#define TEB_PF_LIGHT_TIME_OPTIMAL 1
#define TEB_PF_LIGHT_SHORTEST_PATH 1
#define TEB_PF_LIGHT_VELOCITY_OBSTACLE_RATIO 1
#define TEB_PF_LIGHT_RADIUS_FREE 1
#define TEB_PF_LIGHT_DEBUG_LINEARIZE 0
#define TEB_PF_LIGHT_BAND_LDLT 0
#define TEB_PF_LIGHT_INFLATED 1
#define TEB_PF_LIGHT_NO_NEAR_CACHE 0
#define TEB_PF_LIGHT_DIVERGENCE_DETECTION 0
#define TEB_PF_LIGHT_KIN_DIFF_DRIVE 1
#define TEB_PF_LIGHT_KIN_EDGES 1

// what a device site sees
#if defined(TEB_AMD_DEFAULTS_PROFILE)
#if defined(TEB_AMD_PROFILE_CUSTOM)
// compiled at run time for ONE configuration (teb_rtc.hpp): every flag folds to the value it has there, -DTEB_PF_VALUE_<ID>=true|false
#define TEB_CFGI(ID) (TEB_PF_VALUE_##ID)
#elif defined(TEB_AMD_PROFILE_LIGHT)
#define TEB_CFGI(ID) (TEB_PF_LIGHT_##ID ? (TEB_PF_EXPR_##ID) : (TEB_PF_DFLT_##ID))
#elif defined(TEB_AMD_PROFILE_WIDE) && defined(TEB_AMD_PROFILE_ANY_KINEMATICS)
#define TEB_CFGI(ID) ((TEB_PF_WIDE_##ID || TEB_PF_KIN_##ID) ? (TEB_PF_EXPR_##ID) : (TEB_PF_DFLT_##ID))
#elif defined(TEB_AMD_PROFILE_WIDE)
#define TEB_CFGI(ID) (TEB_PF_WIDE_##ID ? (TEB_PF_EXPR_##ID) : (TEB_PF_DFLT_##ID))
#elif defined(TEB_AMD_PROFILE_ANY_KINEMATICS)
#define TEB_CFGI(ID) (TEB_PF_KIN_##ID ? (TEB_PF_EXPR_##ID) : (TEB_PF_DFLT_##ID))
#else
#define TEB_CFGI(ID) (TEB_PF_DFLT_##ID)
#endif
#else
#define TEB_CFGI(ID) (TEB_PF_EXPR_##ID)
#endif

namespace tebamd {

#ifndef TEB_AMD_THREADS
#define TEB_AMD_THREADS 256
#endif
constexpr int kThreads = TEB_AMD_THREADS;   // 4 wave64 per workgroup, one workgroup per candidate TEB (512 = two waves per SIMD: measured in round 2, 1100 - 1600 spill slots, 6.9 vs 4.8 ms)
constexpr int kWaves = kThreads / 64;
constexpr int kBand = 11;           // diagonal + scalar half-bandwidth 10
// Where band row r (a scalar row: entries (r, r - d), d = 0 .. 10) starts in a band buffer: 11 doubles per row plus ONE padding double per
// pose (4 rows). Lane i of the linearisation owns the rows of pose i; with rows of 11 the lanes of a wave were 44 doubles = 88 banks apart,
// i.e. on 8 distinct bank positions (8-way conflicts in every read-modify-write of the scatter, a quarter of all LDS cycles of the kernel);
// 45 doubles apart they spread over 32. One layout for every band buffer (LDS band, its HBM copy, the HBM band of long bands).
__host__ __device__ constexpr int hbo(int r) { return r * kBand + (r >> 2); }
// Poses handled per thread: n <= kThreads * kMaxPoseIter. A property of the TRANSLATION UNIT: everything sized by it lives in registers
// (near-mask cache, pose backups of the LM loop, the gather of autoResize), so the layouts that hold at most 337 poses are compiled with 2
// and only the band-in-HBM instantiations (teb_opt_inst.hip, and teb_rtc.hpp for the kernels compiled at run time) with
// kPoseIterBandHbm = 4: bands beyond 512 poses (max_samples is a parameter of the reference, teb_config.h:78, 258; 500 is only its
// default) - as many as the LDS strips of that layout hold (21 doubles per pose: ~ 950 poses on MI355X, teb_amd_capacity).
#ifndef TEB_AMD_POSE_ITER
#define TEB_AMD_POSE_ITER 2
#endif
constexpr int kMaxPoseIter = TEB_AMD_POSE_ITER;
#ifdef TEB_AMD_SINGLE_TU
constexpr int kPoseIterBandHbm = kMaxPoseIter;   // (profiling builds: every instantiation inside the host's translation unit)
#else
constexpr int kPoseIterBandHbm = 4;
#endif
static_assert(kMaxPoseIter >= 2 && kMaxPoseIter <= 4, "pose descriptors of autoResize (kNewPose = 1024) and its 16 interval masks hold 1024 poses");

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
  int* last_iters;  // LM iterations of the LAST optimize() call of the band: g2o sizes batchStatistics() to the requested count and fills one
                    // entry per executed iteration, so .back() (src/optimal_planner.cpp:1036) holds a chi2 only when this equals `inner`
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
  double* phase_log;     // opt-in (teb_amd_set_phase_log): [B][kPhaseLogSlots] shader cycles per phase of the band's workgroup; else nullptr
};
constexpr int kPhaseLogSlots = 9;   // autoResize | association + via-points + time stamps | linearise | H backup | solve | update + evaluate | accept / restore | (spare) | whole workgroup

}  // namespace tebamd
