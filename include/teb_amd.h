/*
 * teb_amd.h — C-ABI of the MI355X-native TEB optimiser (libteb_amd.so).
 *
 * This is the drop-in boundary for ONE hot path of rst-tu-dortmund/teb_local_planner:
 *
 *   bool TebOptimalPlanner::optimizeTEB(int iterations_innerloop, int iterations_outerloop,
 *                                       bool compute_cost_afterwards, double obst_cost_scale,
 *                                       double viapoint_cost_scale, bool alternative_time_cost)
 *        reference: include/teb_local_planner/optimal_planner.h:231-232, src/optimal_planner.cpp:182-231
 *   void HomotopyClassPlanner::optimizeAllTEBs(int iter_innerloop, int iter_outerloop)
 *        reference: src/homotopy_class_planner.cpp:466-493
 *   TebOptimalPlannerPtr HomotopyClassPlanner::selectBestTeb()
 *        reference: src/homotopy_class_planner.cpp:564-667
 *
 * Everything is plain C: POD structs, caller-owned fp64 / int32 buffers, an opaque handle, int status
 * codes. No torch / HIP types appear in any signature (streams and device pointers are void*).
 * All arithmetic is fp64 like the reference (Eigen double).
 *
 * The same POD types are consumed by the CPU oracle (oracle/teb_oracle.h, teb_oracle_* symbols); the
 * oracle is test infrastructure and is never linked into libteb_amd.so.
 */
#ifndef TEB_AMD_H_
#define TEB_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TEB_AMD_ABI_VERSION 3
/* Largest pose capacity a handle can be created with on MI355X (band in HBM, four poses per lane; what 160 KB of LDS hold beside the
 * state strips: 951, rounded down to a multiple of 8). trajectory.max_samples (teb_config.h:78) is a parameter of the reference, 500 only
 * by default: bindings size a handle with min(max_samples + 1, TEB_AMD_MAX_POSES). teb_amd_capacity reports the device's own figure. */
#define TEB_AMD_MAX_POSES 944

/* ---- status codes (library calls) ------------------------------------------------------------- */
enum {
  TEB_AMD_OK = 0,
  TEB_AMD_ERR_INVALID_ARG = 1,   /* NULL pointer, negative size, n > max_poses, ...                */
  TEB_AMD_ERR_NO_DEVICE = 2,     /* no HIP device / wrong architecture (no CPU fallback exists)    */
  TEB_AMD_ERR_HIP = 3,           /* a HIP runtime call failed; see teb_amd_last_error()            */
  TEB_AMD_ERR_CAPACITY = 4,      /* max_poses does not fit the per-workgroup LDS budget etc.       */
  TEB_AMD_ERR_UNSUPPORTED = 5    /* feature combination not implemented by this build              */
};

/* ---- per-TEB result status (mirrors the bool returned by optimizeTEB) --------------------------- */
enum {
  TEB_AMD_TEB_OK = 0,            /* optimizeTEB would have returned true                            */
  TEB_AMD_TEB_FAILED = 1,        /* optimizeTEB would have returned false (guards of               */
                                 /*   src/optimal_planner.cpp:185-186, 370-382, 393-397)            */
  TEB_AMD_TEB_NONFINITE = 2      /* a non-finite residual / state was produced (the reference only  */
                                 /*   ROS_ASSERTs on this, e.g. g2o_types/edge_velocity.h:116)      */
};

/* ---- robot footprint models: include/teb_local_planner/robot_footprint_model.h:58-770 ----------- */
enum {
  TEB_AMD_FOOTPRINT_POINT = 0,       /* PointRobotFootprint      :104-176 */
  TEB_AMD_FOOTPRINT_CIRCULAR = 1,    /* CircularRobotFootprint   :229-300 */
  TEB_AMD_FOOTPRINT_TWO_CIRCLES = 2, /* TwoCirclesRobotFootprint :307-430 */
  TEB_AMD_FOOTPRINT_LINE = 3,        /* LineRobotFootprint       :439-635 */
  TEB_AMD_FOOTPRINT_POLYGON = 4      /* PolygonRobotFootprint    :644-770 */
};
#define TEB_AMD_MAX_FOOTPRINT_VERTICES 64   /* (16 until ABI 2; the reference's PolygonRobotFootprint has no limit, robot_footprint_model.h:664-683) */

/* ---- obstacle types: include/teb_local_planner/obstacles.h:67-1111 ------------------------------ */
enum {
  TEB_AMD_OBST_POINT = 0,    /* PointObstacle    (ax,ay)                          */
  TEB_AMD_OBST_CIRCULAR = 1, /* CircularObstacle (ax,ay), radius                  */
  TEB_AMD_OBST_LINE = 2,     /* LineObstacle     (ax,ay)-(bx,by)                  */
  TEB_AMD_OBST_PILL = 3,     /* PillObstacle     (ax,ay)-(bx,by), radius          */
  TEB_AMD_OBST_POLYGON = 4   /* PolygonObstacle  vertices [vert_offset[i], vert_offset[i+1]) */
};

/* ---- RotType: include/teb_local_planner/misc.h:53  (enum class RotType { left, none, right }) --- */
enum { TEB_AMD_ROT_LEFT = 0, TEB_AMD_ROT_NONE = 1, TEB_AMD_ROT_RIGHT = 2 };

/* ---- Jacobian modes (extension; the reference has only the g2o behaviour) ------------------------ */
enum {
  TEB_AMD_JACOBIAN_ANALYTIC = 0,    /* closed-form Jacobians, one-sided conventions of penalties.h:127-187 */
  TEB_AMD_JACOBIAN_G2O_NUMERIC = 1  /* g2o central differences, delta = 1e-9 (Base*Edge::linearizeOplus);  */
                                    /*   EdgeKinematicsDiffDrive / EdgeTimeOptimal stay analytic like the   */
                                    /*   reference (edge_kinematics.h:107-151, edge_time_optimal.h:98-107)  */
};

/*
 * Kernel-relevant subset of TebConfig (include/teb_local_planner/teb_config.h:62-230), same field
 * names, same defaults (teb_config.h:245-390) via teb_amd_config_default(). bools are int32.
 */
typedef struct teb_amd_config {
  /* trajectory */
  int32_t teb_autosize;            /* declared double in the reference, used as bool (teb_config.h:74) */
  double  dt_ref;
  double  dt_hysteresis;
  int32_t min_samples;
  int32_t max_samples;
  int32_t exact_arc_length;
  int32_t via_points_ordered;
  /* robot */
  double  max_vel_x;
  double  max_vel_x_backwards;
  double  max_vel_y;
  double  max_vel_trans;
  double  max_vel_theta;
  double  acc_lim_x;
  double  acc_lim_y;
  double  acc_lim_theta;
  double  min_turning_radius;
  /* obstacles */
  double  min_obstacle_dist;
  double  inflation_dist;
  double  dynamic_obstacle_inflation_dist;
  int32_t include_dynamic_obstacles;
  int32_t obstacle_poses_affected;
  int32_t legacy_obstacle_association;
  double  obstacle_association_force_inclusion_factor;
  double  obstacle_association_cutoff_factor;
  double  obstacle_proximity_ratio_max_vel;
  double  obstacle_proximity_lower_bound;
  double  obstacle_proximity_upper_bound;
  /* optim */
  int32_t no_inner_iterations;
  int32_t no_outer_iterations;
  int32_t optimization_activate;
  double  penalty_epsilon;
  double  weight_max_vel_x;
  double  weight_max_vel_y;
  double  weight_max_vel_theta;
  double  weight_acc_lim_x;
  double  weight_acc_lim_y;
  double  weight_acc_lim_theta;
  double  weight_kinematics_nh;
  double  weight_kinematics_forward_drive;
  double  weight_kinematics_turning_radius;
  double  weight_optimaltime;
  double  weight_shortest_path;
  double  weight_obstacle;
  double  weight_inflation;
  double  weight_dynamic_obstacle;
  double  weight_dynamic_obstacle_inflation;
  double  weight_velocity_obstacle_ratio;
  double  weight_viapoint;
  double  weight_prefer_rotdir;
  double  weight_adapt_factor;
  double  obstacle_cost_exponent;
  /* hcp (selection) */
  double  selection_cost_hysteresis;
  double  selection_prefer_initial_plan;
  double  selection_obst_cost_scale;
  double  selection_viapoint_cost_scale;
  int32_t selection_alternative_time_cost;
  /* recovery */
  int32_t divergence_detection_enable;           /* no constructor default in the reference; 0 here */
  double  divergence_detection_max_chi_squared;  /* cfg default 10 (cfg/TebLocalPlannerReconfigure.cfg:429-445) */
  /* robot_model (TebConfig::robot_model, teb_config.h:68) flattened */
  int32_t footprint_type;          /* TEB_AMD_FOOTPRINT_*                                              */
  double  footprint_radius;        /* circular: radius                                                 */
  double  footprint_front_offset;  /* two circles                                                      */
  double  footprint_front_radius;
  double  footprint_rear_offset;
  double  footprint_rear_radius;
  int32_t footprint_n_vertices;    /* line: 2 (start,end); polygon: >=1                                */
  double  footprint_vx[TEB_AMD_MAX_FOOTPRINT_VERTICES]; /* body-frame vertices                         */
  double  footprint_vy[TEB_AMD_MAX_FOOTPRINT_VERTICES];
  /* extension */
  int32_t jacobian_mode;           /* TEB_AMD_JACOBIAN_*                                               */
} teb_amd_config_t;

/*
 * Obstacle table (ObstContainer = std::vector<boost::shared_ptr<Obstacle>>, obstacles.h:262) as SoA.
 * All arrays have `count` entries except vert_offset (count+1, CSR) and vert_x/vert_y (vert_offset[count]).
 * vert_offset/vert_x/vert_y may be NULL when no TEB_AMD_OBST_POLYGON is present.
 * `dynamic` mirrors Obstacle::isDynamic() (set by setCentroidVelocity, obstacles.h:199-245).
 */
typedef struct teb_amd_obstacles {
  int32_t        count;
  const int32_t* type;
  const double*  ax;
  const double*  ay;
  const double*  bx;
  const double*  by;
  const double*  radius;
  const double*  vx;
  const double*  vy;
  const int32_t* dynamic;
  const int32_t* vert_offset;
  const double*  vert_x;
  const double*  vert_y;
} teb_amd_obstacles_t;

/*
 * A batch of B candidate TEBs (TebOptPlannerContainer, optimal_planner.h:702) as padded SoA:
 * pose i of TEB b lives at index b*stride + i. n[b] poses, n[b]-1 time differences.
 * First and last pose of every TEB are fixed (src/timed_elastic_band.cpp:330,377,398,442).
 * Per-TEB side inputs mirror TebOptimalPlanner members (optimal_planner.h:687-691):
 *   vel_start_ / vel_goal_ (pair<bool, Twist>: linear.x, linear.y, angular.z), prefer_rotdir_,
 *   and whether via_points_ is attached to this candidate (hcp.viapoints_all_candidates).
 */
typedef struct teb_amd_teb_batch {
  int32_t  count;      /* B */
  int32_t  stride;     /* >= max n[b]; <= max_poses of the handle */
  int32_t* n;          /* [B]   in: #poses; out (download): #poses after autoResize */
  double*  x;          /* [B*stride] */
  double*  y;
  double*  theta;
  double*  dt;         /* [B*stride]; dt[b*stride+i] connects pose i and i+1 */
  const int32_t* has_vel_start;   /* [B]  or NULL (= all 1: TebOptimalPlanner::initialize() fixes the start velocity,
                                   *      at zero until setVelocityStart, src/optimal_planner.cpp:94-97; 0 = free start
                                   *      velocity is an extension, the reference cannot express it) */
  const double*  vel_start;       /* [B*3] (vx, vy, omega) or NULL */
  const int32_t* has_vel_goal;    /* [B]  or NULL (= all 1, src/optimal_planner.cpp:99-102); 0 = setVelocityGoalFree() */
  const double*  vel_goal;        /* [B*3] or NULL */
  const int32_t* prefer_rotdir;   /* [B]  TEB_AMD_ROT_* or NULL (= none) */
  const int32_t* via_points_enabled; /* [B] or NULL (= all 1) */
} teb_amd_teb_batch_t;

/* Per-TEB outputs of one optimize_batch call. Any pointer may be NULL. */
typedef struct teb_amd_results {
  int32_t* status;         /* [B] TEB_AMD_TEB_*                                                       */
  int32_t* lm_iterations;  /* [B] LM iterations executed, summed over outer iterations (the "units")  */
  int32_t* lm_trials;      /* [B] damped-solve trials executed                                        */
  double*  chi2;           /* [B] chi^2 after the last LM iteration (g2o batchStatistics().back().chi2, */
                           /*     used by hasDiverged, src/optimal_planner.cpp:1023-1039)              */
  double*  cost;           /* [B] computeCurrentCost (src/optimal_planner.cpp:1041-1094); NaN if not asked */
  double*  lambda;         /* [B] final LM damping                                                     */
} teb_amd_results_t;

typedef struct teb_amd_handle teb_amd_handle_t;

/* -- library ------------------------------------------------------------------------------------ */
int  teb_amd_abi_version(void);
const char* teb_amd_last_error(void);           /* thread-local message of the last failing call */
void teb_amd_config_default(teb_amd_config_t* cfg);   /* TebConfig::TebConfig(), teb_config.h:245-390 */

/*
 * Behaviour switches of a handle (ABI 2; these were process-environment variables in ABI 1 - a library must not change its layout
 * because the host process happens to have a variable set). Zero-initialise, or teb_amd_options_default(), then override.
 */
enum { TEB_AMD_LAYOUT_AUTO = 0,        /* by capacity and obstacle count, and per launch by the current pose counts (see create) */
       TEB_AMD_LAYOUT_BLOCKS_LDS = 1,  /* normal matrix as 8x8 blocks in LDS, cyclic reduction in place (<= 238 poses)           */
       TEB_AMD_LAYOUT_BAND_LDS = 2,    /* band in LDS, cyclic reduction: level 0 from a band copy in HBM (<= 337 poses)         */
       TEB_AMD_LAYOUT_BAND_HBM = 3 };  /* band in HBM as well (<= TEB_AMD_MAX_POSES poses)                                      */
enum { TEB_AMD_HSIG3D_AUTO = 0, TEB_AMD_HSIG3D_WIDE = 1, TEB_AMD_HSIG3D_SMALL = 2 };
typedef struct teb_amd_options {
  int32_t struct_size;            /* sizeof(teb_amd_options_t) of the caller (forward compatibility); 0 = this header's         */
  int32_t layout;                 /* TEB_AMD_LAYOUT_*; anything but AUTO pins the layout (no per-launch choice)                 */
  int32_t fixed_layout;           /* 1: always launch in the capacity's own layout (no optimistic launch + repeat)              */
  int32_t band_ldlt;              /* 1: TEB_AMD_LAYOUT_BAND_LDS solves with the sequential banded LDL^T (cross-check, slow)     */
  int32_t generic_distance_path;  /* 1: never use the point-like LDS obstacle cache                                             */
  int32_t hsig3d_kernel;          /* TEB_AMD_HSIG3D_*: pin one of the two HSignature3d kernels                                  */
  int32_t no_near_cache;          /* 1: the near masks of the dynamic-obstacle edges are recomputed at every pass instead of    */
                                  /*    cached across one optimize() (cross-check: the results must not change by one bit)      */
  int32_t multi_cu;               /* small batches of generic-shape scenes on more than one CU (obstacle association and the       */
                                  /* robot <-> obstacle distances of every (pose, obstacle) pair are computed by helper workgroups */
                                  /* on the idle CUs, the bands are bit-identical to the single-CU result): 0 = automatic (ONE     */
                                  /* band, closed-form Jacobians, new association, enough obstacles x poses), -1 = never,         */
                                  /* n > 0 = at most n helper workgroups per band (and at most one per 6 poses of the capacity)    */
                                  /* on a batch of any size - with two or more bands and tens of helpers per band 0.05 - 2 % of    */
                                  /* the launches returned one band off the single-CU result (open defect). A launch with */
                                  /* distance helpers is synchronous and copies the strips first (it may have to be repeated); its */
                                  /* record buffer is bounded (1 GiB: beyond that, or when it cannot be allocated, the launch runs  */
                                  /* on one CU per band); misses are backed off, see teb_amd_multi_cu_backoff                       */
  int32_t multi_cu_timeout_us;    /* how long a band waits for its helper workgroups before it gives the launch up (it is then     */
                                  /* repeated on one CU per band); 0 = 2000 (2 ms: a phase takes some 10 us when the helpers have */
                                  /* CUs - they do unless other work occupies the device)                                          */
  int32_t speculative_trials;     /* small batches: the damped systems of the first retries of an LM iteration (lambda x 2, x 8,  */
                                  /* x 64) are solved on spare CUs while the band solves and evaluates its first trial, so that a  */
                                  /* rejected trial finds its step ready (bit-identical results): 0 = automatic (closed-form       */
                                  /* Jacobians, blocks-in-LDS or hybrid layout; as many of the three as the batch leaves CUs for:  */
                                  /* 3 solver workgroups per band up to 64 bands, 1 up to 128, none beyond), -1 = never,           */
                                  /* k = 1 .. 3: at most that many                                                                 */
  int32_t generic_config_path;    /* 1: never launch the kernel instantiations specialised on the TebConfig defaults. A configuration  */
                                  /* that takes the paths of a default TebConfig (non-holonomic, no via-points, new association, cost */
                                  /* exponent 1, no exact arc length, inflated obstacle edges, no batch statistics; point-like scenes: */
                                  /* diff-drive, point obstacles) runs kernels compiled with those flags folded: same operations,     */
                                  /* bit-identical bands, 10 - 25 % faster. Cross-check switch.                                        */
  int32_t compile_for_config;     /* a kernel compiled FOR the handle's configuration at run time (hipRTC, 2 - 5 s per instantiation, */
                                  /* cached per process and on disk): a configuration off the TebConfig defaults then runs a kernel  */
                                  /* with every flag of the profile table folded to ITS values - as fast as a default one,           */
                                  /* bit-identical bands. 0 = off, 1 = compile in the background at the first launch that needs it    */
                                  /* (the launches run the best pre-built kernel until the module is ready), 2 = that launch waits   */
                                  /* for the compiler. Needs libhiprtc and the ROCm headers ($ROCM_PATH, default /opt/rocm); the       */
                                  /* kernel sources are inside the library ($TEB_AMD_CSRC = a csrc directory to compile instead).     */
                                  /* Code objects are kept in $TEB_AMD_RTC_CACHE (default $XDG_CACHE_HOME/teb_amd or ~/.cache/teb_amd; */
                                  /* "off" = no disk cache), keyed by sources, flag values, instantiation and compiler version: the   */
                                  /* next process loads them in milliseconds. Where something is missing, or the compilation fails,   */
                                  /* the pre-built kernels run - it is never an error. The key holds scene-dependent flags too (via-  */
                                  /* points present, radii in the static list) and the layout of the launch: when one of them flips   */
                                  /* in a live planner, mode 1 runs the pre-built kernel until the new module is ready, mode 2 blocks  */
                                  /* that tick for the compiler - mode 2 is for benchmarks and tests, mode 1 for a robot. The disk    */
                                  /* cache is only used when its directory belongs to the calling user and nobody else can write it;  */
                                  /* its checksum detects corruption, it is no authentication.                                        */
  int32_t reserved[4];            /* must be 0                                                                                  */
} teb_amd_options_t;
void teb_amd_options_default(teb_amd_options_t* opt);

/*
 * create: one solver per GPU / host thread. device = HIP ordinal. stream = hipStream_t (as void*) to
 * launch on, or NULL for the handle's own stream. Fails (never falls back to CPU) when no gfx950 device.
 * max_poses = pose capacity of every band (trajectory.max_samples + 1 covers whatever autoResize can produce); up to
 * TEB_AMD_MAX_POSES (teb_amd_capacity). Layouts: normal matrix as 8x8 blocks in LDS up to 238 poses (208 beside a 500-obstacle cache) - the
 * fastest -, as a band in LDS up to 337, as a band in HBM beyond. Each launch uses the fastest layout that holds the current
 * bands with 10 % room to grow and is repeated in the capacity's own layout if autoResize outgrows that (teb_amd_options_t::fixed_layout:
 * always the capacity's layout); results do not depend on the layout.
 */
int  teb_amd_create(const teb_amd_config_t* cfg, int32_t max_tebs, int32_t max_poses,
                    int32_t max_obstacles, int32_t max_obstacle_vertices, int32_t max_via_points,
                    int32_t device, void* stream, teb_amd_handle_t** out);
/* the same with explicit options (NULL = defaults = teb_amd_create) */
int  teb_amd_create_ex(const teb_amd_config_t* cfg, int32_t max_tebs, int32_t max_poses,
                       int32_t max_obstacles, int32_t max_obstacle_vertices, int32_t max_via_points,
                       int32_t device, void* stream, const teb_amd_options_t* options, teb_amd_handle_t** out);
void teb_amd_destroy(teb_amd_handle_t* h);
/* dynamic_reconfigure. A change of include_dynamic_obstacles or of the footprint type re-derives the static / dynamic obstacle
 * lists and the distance path from the obstacle table the handle already holds (no re-upload needed). On an error (invalid
 * footprint / jacobian_mode) the handle keeps its previous configuration. */
int  teb_amd_set_config(teb_amd_handle_t* h, const teb_amd_config_t* cfg);

/* Scene, once per plan(): obstacles_ and via_points_ (optimal_planner.h:683-684). Host pointers. */
int  teb_amd_set_obstacles(teb_amd_handle_t* h, const teb_amd_obstacles_t* obst);
int  teb_amd_set_via_points(teb_amd_handle_t* h, int32_t count, const double* x, const double* y);

/* State strips host -> HBM and back. */
int  teb_amd_upload_tebs(teb_amd_handle_t* h, const teb_amd_teb_batch_t* batch);
int  teb_amd_download_tebs(teb_amd_handle_t* h, teb_amd_teb_batch_t* batch);

/*
 * The hot path: B x optimizeTEB (src/optimal_planner.cpp:182-231), i.e. optimizeAllTEBs
 * (src/homotopy_class_planner.cpp:466-493) when called with compute_cost=1 and the selection_* scales.
 * Asynchronous on the handle's stream; results are valid after teb_amd_synchronize / get_results. Two cases return only after the
 * launch has finished, because its per-band flags decide whether it has to be repeated: a handle launched in a faster layout than its
 * capacity's own (see teb_amd_create), and a launch with distance helpers (teb_amd_options_t::multi_cu).
 */
int  teb_amd_optimize_batch(teb_amd_handle_t* h, int32_t iterations_innerloop, int32_t iterations_outerloop,
                            int32_t compute_cost_afterwards, double obst_cost_scale,
                            double viapoint_cost_scale, int32_t alternative_time_cost);
int  teb_amd_synchronize(teb_amd_handle_t* h);
int  teb_amd_get_results(teb_amd_handle_t* h, teb_amd_results_t* out);   /* synchronises, copies D2H */

/*
 * Per-iteration log of the Levenberg-Marquardt loop, opt-in - the data of the line g2o prints per iteration when the reference sets
 * SparseOptimizer::setVerbose(cfg_->optim.optimization_verbose) (src/optimal_planner.cpp:384): "iteration= i chi2= .. lambda= ..
 * levenbergIter= ..". Row r (4 doubles) of band b describes LM iteration r of the last teb_amd_optimize_batch, counted over all outer
 * iterations: { chi^2 after the iteration (of the accepted state), lambda after it, damping trials the iteration took, pose count of
 * the band in that outer iteration }. At most TEB_AMD_ITERATION_LOG_ROWS rows per band are kept. Off by default (one extra store per
 * iteration and band when on). teb_amd_get_iteration_log synchronises; *n_rows = min(lm_iterations[b], capacity_rows, ROWS).
 */
#define TEB_AMD_ITERATION_LOG_ROWS 256
int  teb_amd_set_iteration_log(teb_amd_handle_t* h, int32_t enable);
int  teb_amd_get_iteration_log(teb_amd_handle_t* h, int32_t b, double* rows, int32_t capacity_rows, int32_t* n_rows);

/*
 * selectBestTeb (src/homotopy_class_planner.cpp:564-667) on the device-resident costs:
 * cost[last_best] *= selection_cost_hysteresis, cost[initial_plan] *= selection_prefer_initial_plan
 * (indices < 0 = none), strict '<' so the lowest index wins ties; TEBs whose status != OK still take
 * part exactly like in the reference (their stored cost_ is compared). Returns index in *best, its
 * (scaled) cost in *best_cost. The switching_blocking_period logic stays in the caller (needs ros::Time).
 */
int  teb_amd_select_best(teb_amd_handle_t* h, int32_t last_best, int32_t initial_plan,
                         int32_t* best, double* best_cost);

/* How the last teb_amd_optimize_batch ran: distance-helper and solver-helper workgroups per band of the multi-CU mode (0, 0 = one CU
 * per band), and whether the launch had to be repeated on one CU per band because the distance helpers did not arrive in time (a
 * device busy with other work). */
int  teb_amd_last_launch_info(teb_amd_handle_t* h, int32_t* distance_helpers_per_band, int32_t* solver_helpers_per_band,
                              int32_t* repeated_single_cu);
/* Back-off of the distance helpers on a device that is not empty: after a launch whose helpers came late (it was repeated on one CU per
 * band) the next *pause_length launches (4, doubling with every further miss up to 256) run on one CU per band without asking for
 * helpers - no wait, no repeat -, then one launch probes again; a probe that succeeds clears the pause. *launches_paused = how many of
 * them are still to come (0 = the next launch asks for helpers). A launch with distance helpers is synchronous (the per-band flags
 * decide about the repeat) and is preceded by a device-to-device copy of the strips; a paused one is neither. */
int  teb_amd_multi_cu_backoff(teb_amd_handle_t* h, int32_t* launches_paused, int32_t* pause_length);

/* -- zero-copy access for callers that already live on the GPU (benchmarks, torch interop) ---------- */
/*
 * f3 (candidate generation) — HomotopyClassPlanner::exploreEquivalenceClassesAndInitTebs after renewAndAnalyzeOldTebs
 * (src/homotopy_class_planner.cpp:318-340): GraphSearchInterface::createGraph (lrKeyPointGraph src/graph_search.cpp:95-223 when
 * simple_exploration, else ProbRoadmapGraph :227-340), the depth-first path enumeration (:45-91) and, for every start-goal path in
 * the reference's order, addAndInitNewTeb (homotopy_class_planner.hpp:66-93): band from the path (timed_elastic_band.hpp:46-183),
 * its equivalence class, addEquivalenceClassIfNew (src/homotopy_class_planner.cpp:189-212).
 */
typedef struct teb_amd_hcp_params {
  int32_t simple_exploration;                 /* hcp.simple_exploration: 1 = lrKeyPointGraph, 0 = ProbRoadmapGraph          */
  int32_t roadmap_graph_no_samples;           /* hcp.roadmap_graph_no_samples                                               */
  double  roadmap_graph_area_width;           /* hcp.roadmap_graph_area_width                                               */
  double  roadmap_graph_area_length_scale;    /* hcp.roadmap_graph_area_length_scale                                        */
  double  obstacle_heading_threshold;         /* hcp.obstacle_heading_threshold                                             */
  double  xy_goal_tolerance;                  /* goal_tolerance.xy_goal_tolerance                                           */
  int32_t max_number_classes;                 /* hcp.max_number_classes                                                     */
  int32_t max_number_plans_in_current_class;  /* hcp.max_number_plans_in_current_class                                      */
  double  h_signature_prescaler;              /* hcp.h_signature_prescaler                                                  */
  double  h_signature_threshold;              /* hcp.h_signature_threshold                                                  */
  int32_t allow_init_with_backwards_motion;   /* trajectory.allow_init_with_backwards_motion                                */
  int32_t delete_detours_backwards;           /* hcp.delete_detours_backwards                                               */
  double  detours_orientation_tolerance;      /* hcp.detours_orientation_tolerance                                          */
  double  length_start_orientation_vector;    /* hcp.length_start_orientation_vector                                        */
  double  max_ratio_detours_duration_best_duration; /* hcp.max_ratio_detours_duration_best_duration                        */
  int32_t global_plan_overwrite_orientation;  /* trajectory.global_plan_overwrite_orientation                               */
  int32_t viapoints_all_candidates;           /* hcp.viapoints_all_candidates                                               */
} teb_amd_hcp_params_t;
void teb_amd_hcp_params_default(teb_amd_hcp_params_t* p);   /* the defaults of teb_config.h:330-360 */
/*
 * The bands of the batch are the planner's tebs_ after renewAndAnalyzeOldTebs (teb_amd_compute_h_signatures,
 * teb_amd_filter_equivalence_classes, teb_amd_compact_bands; an empty batch is a planner without trajectories), best = index of
 * the last best band (its class may hold up to max_number_plans_in_current_class bands) or -1. New candidates are appended to the
 * batch while it holds fewer than max_number_classes (and than max_tebs) bands; *n_total = batch size afterwards.
 * start_vel = (linear.x, linear.y, angular.z) for setVelocityStart on the new bands or NULL; free_goal_vel = setVelocityGoalFree().
 * The collision tests of all vertex pairs against all obstacles, the band initialisation and the H-signatures of a chunk of
 * candidate paths run on the device; the depth-first enumeration (pointer chasing over the adjacency lists) runs on the host and
 * feeds the device in chunks of paths, which are accepted in the reference's order.
 * unit_samples: ProbRoadmapGraph only - [2*roadmap_graph_no_samples] uniform numbers in [0,1) replacing the planner's generator,
 * in draw order: per sample first the one scaled to the area width, then the one scaled to the area length (the order GCC gives
 * Eigen::Vector2d(distribution_x(g), distribution_y(g)), src/graph_search.cpp:274); NULL = the handle's own generator, the
 * default-seeded mt19937 + boost::random::uniform_real_distribution stream of the reference's member generator.
 * n_plan > 0: the initial plan of plan(initial_plan, ...) (positions and yaw of its poses; start / goal are its first / last pose):
 * it is tried as a candidate before the graph (addAndInitNewTeb(*initial_plan_, ...), src/homotopy_class_planner.cpp:326-329,
 * 412-440: band via initTrajectoryToGoal(plan, ...), its class is remembered as initial_plan_eq_class_). *initial_plan_teb =
 * getInitialPlanTEB() afterwards (:495-536): that band, else the first band of the remembered class, else -1 - the index to hand to
 * teb_amd_select_best. Via-points (updateReferenceTrajectoryViaPoints, :286-315): new candidates start without them; they are
 * enabled for all bands (viapoints_all_candidates) or, with an initial plan, for the bands of its class only.
 * *n_vertices / *n_paths (may be NULL): graph size and number of start-goal paths examined. max_paths > 0 bounds the enumeration
 * to that many start-goal paths and 10000 x max_paths vertex expansions (the reference has no bound: without new classes it
 * enumerates every simple path, and dead-end subtrees can be exponential in a large graph); TEB_AMD_OK is returned either way.
 */
int  teb_amd_explore_candidates(teb_amd_handle_t* h, const teb_amd_hcp_params_t* p, const double* start, const double* goal,
                                double dist_to_obst, const double* start_vel, int32_t free_goal_vel, int32_t best,
                                const double* unit_samples, int64_t max_paths, int32_t* n_total, int32_t* n_vertices, int32_t* n_paths,
                                int32_t n_plan, const double* plan_x, const double* plan_y, const double* plan_yaw,
                                int32_t* initial_plan_teb);
/* the graph of the last teb_amd_explore_candidates call: vertices (x, y) [n_vertices], adjacency bytes [n_vertices^2] (row = from) */
int  teb_amd_get_exploration_graph(teb_amd_handle_t* h, double* vx, double* vy, unsigned char* adjacency, int32_t capacity_vertices,
                                   int32_t* n_vertices);
/*
 * Compaction of the batch after teb_amd_filter_equivalence_classes: the bands with keep[b] != 0 move to the front, in the order of
 * renewAndAnalyzeOldTebs (the last best band, best >= 0, swapped with the first one, src/homotopy_class_planner.cpp:220-224); the
 * batch shrinks to them. *n_kept = batch size afterwards, *new_best = position of the best band afterwards (0) or -1.
 */
int  teb_amd_compact_bands(teb_amd_handle_t* h, const int32_t* keep, int32_t best, int32_t* n_kept, int32_t* new_best);
/*
 * HomotopyClassPlanner::deletePlansDetouringBackwards (src/homotopy_class_planner.cpp:766-838), the second half of
 * renewAndAnalyzeOldTebs when hcp.delete_detours_backwards: keep [B] in/out - bands with keep[b] = 0 (erased by
 * teb_amd_filter_equivalence_classes) are not looked at; a remaining band other than the last best one (best) is dropped when it
 * has fewer than 2 poses, no pose further than length_start_orientation_vector from its start, a start direction more than
 * detours_orientation_tolerance away from the best band's, was not optimised by the last teb_amd_optimize_batch call
 * (TebOptimalPlanner::isOptimized), or lasts more than max_ratio_detours_duration_best_duration times the best band's duration.
 * Start directions and durations of all bands are computed in one launch. Call before teb_amd_compact_bands.
 */
int  teb_amd_filter_detours(teb_amd_handle_t* h, const teb_amd_hcp_params_t* p, int32_t best, int32_t* keep);
/* TebOptimalPlanner::optimized_ (optimal_planner.h:386, 691) of the resident bands: set by teb_amd_optimize_batch exactly as
 * optimizeTEB sets it (src/optimal_planner.cpp:189, 220), cleared by teb_amd_upload_tebs and for new bands; flags [B] overrides it
 * for bands whose history lives on the host (get: flags receives the current values). */
int  teb_amd_set_optimized_flags(teb_amd_handle_t* h, const int32_t* flags);
int  teb_amd_get_optimized_flags(teb_amd_handle_t* h, int32_t* flags);
/* per-band attributes of the resident batch, each [B], any may be NULL: via-points attached (TebOptimalPlanner::via_points_ != NULL),
 * vel_start_.first, vel_goal_.first */
int  teb_amd_get_band_flags(teb_amd_handle_t* h, int32_t* via_points_enabled, int32_t* has_vel_start, int32_t* has_vel_goal);
/* Device pointers (hipDeviceptr as void*) of the resident SoA strips: x, y, theta, dt, each
 * [max_tebs*max_poses] doubles, and n [max_tebs] int32. Valid until destroy.                          */
int  teb_amd_device_state(teb_amd_handle_t* h, void** x, void** y, void** theta, void** dt, void** n,
                          int32_t* stride);
/* Snapshot / restore the resident strips device-to-device (used to re-run identical work per step). */
int  teb_amd_snapshot_state(teb_amd_handle_t* h);
int  teb_amd_restore_state(teb_amd_handle_t* h);
/*
 * ---- Producers and consumers of the DEVICE-RESIDENT strips --------------------------------
 * With these a planning tick is: update_and_prune (or init_trajectory_*) -> set_obstacles -> optimize_batch -> select_best ->
 * get_velocity_command, without uploading or downloading the bands.
 *
 * f1 — TimedElasticBand::initTrajectoryToGoal, the three overloads, writing slot b (0 <= b < max_tebs; slots >= the current
 * batch size extend it, with the TebOptimalPlanner::initialize() defaults: start / goal velocity fixed at zero, no preferred turning
 * direction, via-points enabled). Poses are (x, y, theta). TEB_AMD_ERR_CAPACITY when the band would exceed max_poses.
 */
/* initTrajectoryToGoal(start, goal, diststep, max_vel_x, min_samples, guess_backwards_motion), src/timed_elastic_band.cpp:325-377 */
int  teb_amd_init_trajectory_line(teb_amd_handle_t* h, int32_t b, const double* start, const double* goal, double diststep,
                                  double max_vel_x, int32_t min_samples, int32_t guess_backwards_motion);
/* initTrajectoryToGoal(plan, max_vel_x, max_vel_theta, estimate_orient, min_samples, guess_backwards_motion), :380-452; the plan
 * (std::vector<geometry_msgs::PoseStamped>) as n_plan positions + yaw (tf::getYaw of each orientation) */
int  teb_amd_init_trajectory_plan(teb_amd_handle_t* h, int32_t b, int32_t n_plan, const double* plan_x, const double* plan_y,
                                  const double* plan_yaw, double max_vel_x, double max_vel_theta, int32_t estimate_orient,
                                  int32_t min_samples, int32_t guess_backwards_motion);
/* template initTrajectoryToGoal(path_start, path_end, fun_position, max_vel_x, max_vel_theta, max_acc_x, max_acc_theta,
 * start_orientation, goal_orientation, min_samples, guess_backwards_motion), timed_elastic_band.hpp:46-183 (the overload
 * HomotopyClassPlanner feeds with graph vertices). Optional arguments: NULL = boost::none. */
int  teb_amd_init_trajectory_path(teb_amd_handle_t* h, int32_t b, int32_t n_path, const double* path_x, const double* path_y,
                                  double max_vel_x, double max_vel_theta, const double* max_acc_x, const double* start_orientation,
                                  const double* goal_orientation, int32_t min_samples, int32_t guess_backwards_motion);
/* TimedElasticBand::updateAndPruneTEB(new_start, new_goal, min_samples), src/timed_elastic_band.cpp:555-597, on band b or on every
 * band of the batch (b = -1: HomotopyClassPlanner::updateAllTEBs, src/homotopy_class_planner.cpp:539-562). NULL = boost::none. */
int  teb_amd_update_and_prune(teb_amd_handle_t* h, int32_t b, const double* new_start, const double* new_goal, int32_t min_samples);
/* setVelocityStart / setVelocityGoal / setVelocityGoalFree (optimal_planner.h:247-260) on band b or all (b = -1);
 * fixed = 0 frees the velocity (the stored twist is kept), v = (linear.x, linear.y, angular.z) or NULL to keep the stored twist. */
int  teb_amd_set_velocity_start(teb_amd_handle_t* h, int32_t b, int32_t fixed, const double* v);
int  teb_amd_set_velocity_goal(teb_amd_handle_t* h, int32_t b, int32_t fixed, const double* v);
/* current pose counts of the batch (after init / prune / optimise), n [count] */
int  teb_amd_get_pose_counts(teb_amd_handle_t* h, int32_t* n, int32_t* count);
/*
 * f2 — consumers of the optimised strip of band b, computed on the device for the whole batch in one launch and cached until the
 * bands change: TebOptimalPlanner::getVelocityCommand (src/optimal_planner.cpp:1135-1168; *ok = its return value; uses
 * trajectory.prevent_look_ahead_poses_near_goal, passed here because teb_amd_config_t does not carry it), getVelocityProfile
 * (:1170-1196, out [(n+1)*3] = linear.x, linear.y, angular.z), getFullTrajectory (:1198-1247, out [n*7] = x, y, theta,
 * linear.x, linear.y, angular.z, time_from_start) and hasDiverged (:1023-1039).
 */
int  teb_amd_get_velocity_command(teb_amd_handle_t* h, int32_t b, int32_t look_ahead_poses,
                                  int32_t prevent_look_ahead_poses_near_goal, double* vx, double* vy, double* omega, int32_t* ok);
int  teb_amd_get_velocity_profile(teb_amd_handle_t* h, int32_t b, double* out, int32_t capacity_rows, int32_t* rows);
int  teb_amd_get_full_trajectory(teb_amd_handle_t* h, int32_t b, double* out, int32_t capacity_rows, int32_t* rows);
int  teb_amd_has_diverged(teb_amd_handle_t* h, int32_t b, int32_t* diverged);
/*
 * What optimizer_->batchStatistics() of every resident band would hold after the last teb_amd_optimize_batch (hasDiverged reads
 * .back().chi2, src/optimal_planner.cpp:1029-1038): available [count] = the vector is not empty (divergence_detection_enable was set
 * for that call, :331, and optimize() ran), back_chi2 [count] = .back().chi2. g2o sizes the vector to the REQUESTED inner iteration
 * count and fills one entry per executed iteration, so a band whose last optimize() call stopped early has back_chi2 = 0 and does not
 * count as diverged; teb_amd_has_diverged applies the same rule. A binding keeps the pair per planner object.
 */
int  teb_amd_get_batch_statistics(teb_amd_handle_t* h, int32_t* available, double* back_chi2);

/*
 * f4 (arithmetic part) — TebOptimalPlanner::isTrajectoryFeasible (src/optimal_planner.cpp:1250-1308; declared optimal_planner.h:500,
 * called by the ROS adapter at src/teb_local_planner_ros.cpp:396) on the device-resident bands. The costmap is the uint8 grid of
 * costmap_2d::Costmap2D (cells[my * size_x + mx]; 254 lethal, 253 inscribed, 255 no information; world (wx, wy) -> cell
 * ((int)((wx - origin_x) / resolution), (int)((wy - origin_y) / resolution))), uploaded once per tick; the footprint test is
 * base_local_planner::CostmapModel::footprintCost (outline rasterised with Bresenham; < 3 vertices: the centre cell) and, as in the
 * reference, only its value -1 (lethal) makes a pose infeasible - unknown cells and poses off the map do not.
 * footprint = footprint_spec [n_footprint <= 64] in the robot frame; inscribed_radius > 0 = robot_inscribed_radius_ (the adapter gets it
 * from costmap_2d::calculateMinAndMaxDistances); min_resolution_collision_check_angular = cfg.trajectory.<same>; look_ahead_idx =
 * cfg.trajectory.feasibility_check_no_poses (< 0 or >= n: the whole band); feasibility_check_lookahead_distance <= 0: off.
 * b >= 0: that band, outputs of 1 entry; b = -1: every band of the batch, outputs [B] (HomotopyClassPlanner checks its best band only,
 * src/homotopy_class_planner.cpp:727-737 - pass its index). first_infeasible (may be NULL): number of footprint tests that pass, in the
 * reference's order, before the colliding one; -1 if feasible. TEB_AMD_ERR_CAPACITY when the radii ask for more than 2^22 samples.
 */
int  teb_amd_set_costmap(teb_amd_handle_t* h, const uint8_t* cells, int32_t size_x, int32_t size_y, double resolution,
                         double origin_x, double origin_y);
int  teb_amd_is_trajectory_feasible(teb_amd_handle_t* h, int32_t b, int32_t n_footprint, const double* footprint_x,
                                    const double* footprint_y, double inscribed_radius, double min_resolution_collision_check_angular,
                                    int32_t look_ahead_idx, double feasibility_check_lookahead_distance, int32_t* feasible,
                                    int32_t* first_infeasible);

/*
 * f3 (arithmetic core) — equivalence classes of the device-resident bands, as HomotopyClassPlanner::calculateEquivalenceClass
 * (homotopy_class_planner.hpp:46-62) computes them for every candidate in renewAndAnalyzeOldTebs: HSignature3d
 * (h_signature.h:281-347; one value per obstacle) when cfg.include_dynamic_obstacles, else HSignature (h_signature.h:96-188; one
 * complex value). One launch for the whole batch. prescaler = hcp.h_signature_prescaler. values (may be NULL): 3-D [B*M] in
 * obstacle-table order, 2-D [B*2] = (re, im). *width receives M or 2.
 */
int  teb_amd_compute_h_signatures(teb_amd_handle_t* h, double prescaler, double* values, int32_t* width);
/*
 * The class list of renewAndAnalyzeOldTebs / addEquivalenceClassIfNew / hasEquivalenceClass (src/homotopy_class_planner.cpp:178-254)
 * over the signatures of the last teb_amd_compute_h_signatures call: bands are visited in order, with the last best band first
 * (best >= 0; std::iter_swap with the first band); keep[b] = 1 iff band b opens a new class or is one of at most
 * max_number_plans_in_current_class bands in the best band's class; valid[b] = isValid(), reasonable[b] = isReasonable()
 * (h_signature.h:190-226, 349-409). threshold = hcp.h_signature_threshold. Output arrays [B], any may be NULL.
 */
int  teb_amd_filter_equivalence_classes(teb_amd_handle_t* h, double threshold, int32_t best, int32_t max_number_plans_in_current_class,
                                        int32_t* keep, int32_t* valid, int32_t* reasonable);

/*
 * ---- Multi-GPU: one process per GPU, the candidate batch sharded by contiguous blocks, scene replicated, no
 * data-path collective. The path's single exchange - best-trajectory selection - lives here, behind the C-ABI, so that the C++
 * drop-in (HomotopyClassPlannerAmd) can shard as well: an RCCL communicator over xGMI (librccl is dlopen'ed on first use).
 *   rank 0:    teb_amd_comm_unique_id(id)            -> ship the 128 bytes to the other ranks by any means (MPI, a ROS param, a file,
 *                                                        torch.distributed)
 *   all ranks: teb_amd_comm_create(id, rank, world, device, &comm)
 *   per plan:  teb_amd_optimize_batch(h, ...)         each rank on its own candidates
 *              teb_amd_select_best_distributed(...)    one ncclAllGather of 16 bytes per rank; same answer on every rank
 *              teb_amd_broadcast_band(...)             optional: the winner's strip (24 + 32 * capacity bytes) from its owner to all
 */
#define TEB_AMD_COMM_ID_BYTES 128
typedef struct teb_amd_comm teb_amd_comm_t;
int  teb_amd_comm_unique_id(char id[TEB_AMD_COMM_ID_BYTES]);
int  teb_amd_comm_create(const char id[TEB_AMD_COMM_ID_BYTES], int32_t rank, int32_t world, int32_t device, teb_amd_comm_t** out);
void teb_amd_comm_destroy(teb_amd_comm_t* comm);
/*
 * selectBestTeb (src/homotopy_class_planner.cpp:564-667) over the candidates of ALL ranks. This rank holds the global candidates
 * [global_offset, global_offset + B); last_best_global / initial_plan_global are global indices (< 0 = none) and receive the
 * hysteresis / prefer-initial-plan multipliers on the rank that owns them. Lowest cost wins, ties -> lowest global index (strict '<'
 * of :610). *best_global, *best_cost (scaled, bit-exact: costs travel as fp64), *owner_rank are identical on every rank.
 * Collective: every rank of the communicator must call it. Runs on the handle's stream and synchronises it.
 */
int  teb_amd_select_best_distributed(teb_amd_handle_t* h, teb_amd_comm_t* comm, int32_t global_offset, int32_t last_best_global,
                                     int32_t initial_plan_global, int32_t* best_global, double* best_cost, int32_t* owner_rank);
/*
 * The winner's strip from its owner to every rank (host buffers x, y, theta, dt of `capacity` doubles each, *n poses): what the rank
 * that talks to the robot needs for getVelocityCommand / visualisation. local_index is only read on owner_rank. capacity must be
 * the same on every rank and >= the winner's pose count (TEB_AMD_ERR_CAPACITY otherwise, on every rank). Collective.
 */
int  teb_amd_broadcast_band(teb_amd_handle_t* h, teb_amd_comm_t* comm, int32_t owner_rank, int32_t local_index, int32_t capacity,
                            int32_t* n, double* x, double* y, double* theta, double* dt);
/*
 * The winner's batch statistics travel with its strip (24 bytes more): what teb_amd_get_batch_statistics returns for the band on its
 * owner, as received by the last teb_amd_broadcast_band on this communicator. A rank that mirrors the winner answers hasDiverged
 * (src/optimal_planner.cpp:1023-1039; the plugin asks after every plan(), src/teb_local_planner_ros.cpp:374) like the owner does.
 */
int  teb_amd_comm_last_band_statistics(const teb_amd_comm_t* comm, int32_t* available, double* back_chi2);

/* Duration [ms] of the last optimize_batch call on the device, measured with HIP events on the launch stream: from before the
 * (first) kernel launch to after the last one, i.e. including a repeated launch when autoResize outgrew the optimistic layout. */
int  teb_amd_last_kernel_ms(teb_amd_handle_t* h, float* ms);
/* Shader clock [MHz] the last optimise kernel ran at: shader-cycle counter / 100 MHz real-time counter between entry and exit of its
 * first workgroup. The boxes of a pool sustain different clocks under the same load; a measurement quotes it so that a slow box is not
 * read as a regression. */
int  teb_amd_last_shader_clock_mhz(teb_amd_handle_t* h, double* mhz);
/* LDS bytes per workgroup and the largest pose count this build can optimise. */
int  teb_amd_capacity(teb_amd_handle_t* h, int32_t* lds_bytes, int32_t* max_poses_supported);

#ifdef __cplusplus
}
#endif
#endif /* TEB_AMD_H_ */
