// teb_amd.hip — libteb_amd.so: host side of the C-ABI declared in include/teb_amd.h (+ teb_amd_debug.h).
// Plain HIP runtime; no torch, no oracle, no CPU fallback: every entry point fails loudly without a gfx950 GPU.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <unistd.h>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <limits>
#include <string>
#include <random>
#include <vector>

#include "../../include/teb_amd.h"
#include "../../include/teb_amd_debug.h"
#define TEB_AMD_MAIN_TU   // (non-template kernels of the shared headers are defined here only)
#include "teb_kernel.hpp"
#include "teb_opt_launch.hpp"
#include "teb_strip.hpp"
#include "teb_hsig.hpp"
#include "teb_graph.hpp"
#include "teb_comm.hpp"
#include "teb_rtc.hpp"
#include "teb_feasibility.hpp"

using namespace tebamd;

// the instantiations of teb_optimize_kernel live in translation units of their own (teb_opt_launch.hpp)
TEB_OPT_FOR_ALL(TEB_OPT_DECLARE)
#ifdef TEB_AMD_SINGLE_TU
TEB_OPT_FOR_ALL(TEB_OPT_DEFINE)
#endif

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess)                                                                              \
      return fail(TEB_AMD_ERR_HIP, std::string(#expr) + " -> " + hipGetErrorString(_e));               \
  } while (0)

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count) {
    n = count;
    if (count == 0) count = 1;
    return hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
  }
  void free() {
    if (p) (void)hipFree(p);
    p = nullptr;
  }
};

// PolygonObstacle::calcCentroid (reference src/obstacles.cpp:56-121) — product-side implementation
void polygon_centroid(const double* vx, const double* vy, int n, double& cx, double& cy) {
  if (n <= 0) { cx = cy = std::numeric_limits<double>::quiet_NaN(); return; }
  if (n == 1) { cx = vx[0]; cy = vy[0]; return; }
  if (n == 2) { cx = 0.5 * (vx[0] + vx[1]); cy = 0.5 * (vy[0] + vy[1]); return; }
  double A = 0;
  for (int i = 0; i < n - 1; ++i) A += vx[i] * vy[i + 1] - vx[i + 1] * vy[i];
  A += vx[n - 1] * vy[0] - vx[0] * vy[n - 1];
  A *= 0.5;
  if (A != 0) {
    cx = 0; cy = 0;
    for (int i = 0; i < n - 1; ++i) {
      double aux = vx[i] * vy[i + 1] - vx[i + 1] * vy[i];
      cx += (vx[i] + vx[i + 1]) * aux;
      cy += (vy[i] + vy[i + 1]) * aux;
    }
    double aux = vx[n - 1] * vy[0] - vx[0] * vy[n - 1];
    cx += (vx[n - 1] + vx[0]) * aux;
    cy += (vy[n - 1] + vy[0]) * aux;
    cx /= (6 * A);
    cy /= (6 * A);
    return;
  }
  int ic = 0, jc = 0;
  double md = 0;
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) {
      double d = std::sqrt((vx[j] - vx[i]) * (vx[j] - vx[i]) + (vy[j] - vy[i]) * (vy[j] - vy[i]));
      if (d > md) { md = d; ic = i; jc = j; }
    }
  cx = 0.5 * (vx[ic] + vx[jc]);
  cy = 0.5 * (vy[ic] + vy[jc]);
}

// K9: selectBestTeb on the resident cost array (src/homotopy_class_planner.cpp:564-667)
__global__ void select_best_kernel(const double* cost, int count, int last_best, int initial_plan, double hyst,
                                   double prefer, double* out_cost, int* out_idx) {
  __shared__ double sv[kThreads];
  __shared__ int si[kThreads];
  double best = 1.7976931348623157e308;   // min_cost starts at numeric_limits<double>::max(), strict '<'
  int bi = -1;
  for (int i = threadIdx.x; i < count; i += kThreads) {
    double cst = cost[i];
    if (i == last_best) cst = cst * hyst;
    else if (i == initial_plan) cst = cst * prefer;
    if (cst < best) { best = cst; bi = i; }
  }
  sv[threadIdx.x] = best; si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      double ov = sv[threadIdx.x + s]; int oi = si[threadIdx.x + s];
      double mv = sv[threadIdx.x]; int mi = si[threadIdx.x];
      bool take = (oi >= 0) && (mi < 0 || ov < mv || (ov == mv && oi < mi));
      if (take) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out_cost[0] = sv[0]; out_cost[1] = (double)si[0]; *out_idx = si[0]; }   // (cost, index) also as one 16-byte record
}

// debug: known-byte-count stream with the kernel's own global access pattern (8 B per lane, coalesced) used to
// calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section)
__global__ void stream_kernel(const double* __restrict__ src, double* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i] + 1.0;
}

// debug: footprint_distance for a list of queries
__global__ void distance_kernel(const teb_amd_config_t c, const SceneDev sc, int nq, const int* oi, const double* x,
                                const double* y, const double* th, const int* st, const double* t, double* dist,
                                double* grad) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  double g[3];
  dist[q] = footprint_distance(c, sc, oi[q], x[q], y[q], cos(th[q]), sin(th[q]), st[q] != 0, t[q], g);
  grad[3 * q] = g[0]; grad[3 * q + 1] = g[1]; grad[3 * q + 2] = g[2];
}

}  // namespace

struct teb_amd_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  teb_amd_config_t cfg;
  int max_tebs = 0, stride = 0, max_obst = 0, max_verts = 0, max_via = 0;
  int B = 0, M = 0, nvia = 0, n_static = 0, n_dyn = 0;
  size_t lds_bytes = 0;
  int solver = 0, solver_created = 0;   // solver_created: the choice of teb_amd_create; teb_amd_set_obstacles may move a band to HBM
  size_t hmat_stride = 0;
  int band_ldlt = 0;   // SOLVER_BAND: 1 = sequential banded LDL^T (teb_amd_options_t::band_ldlt), 0 = hybrid cyclic reduction
  size_t lds_limit = 0;
  int num_cus = 0;
  LdsPlan plan;
  int fast_points = 0;
  int static_radius_zero = 0;   // every obstacle of the static list enters the LDS cache with radius 0 (no circular obstacle among them)
  int last_defaults_profile = 0;   // the last optimise launch ran a *_DEFAULTS instantiation (teb_amd_debug_last_config_profile)
  int last_inst[3] = {-1, -1, -1}; // (layout, Jacobian mode, scene kind) of the instantiation the last optimise launch ran (teb_amd_debug_last_instantiation)
  teb_amd_options_t opt;   // behaviour switches fixed at create (ABI 2; never the process environment)
  int last_inner = 0;      // iterations_innerloop of the last teb_amd_optimize_batch (hasDiverged: size of g2o's batch statistics)
  int snap_nmax = -1;      // nmax_known at the time of teb_amd_snapshot_state
  int snap_B = -1;         // B at that time (the bound covers those bands only)
  int nmax_known = -1;     // upper bound of the resident pose counts as far as the host knows it, -1 = unknown (device-side producers ran)
  // host copy of the last obstacle table: teb_amd_set_config re-derives the static / dynamic lists from it
  struct HostObst { std::vector<int> type, dyn, voff; std::vector<double> ax, ay, bx, by, rad, vx, vy, cx, cy, brad, pvx, pvy; } hob;
  std::vector<int> host_static;
  // scene
  DevBuf<int> o_type, o_dyn, o_voff, o_static, o_dynidx;
  DevBuf<double> o_ax, o_ay, o_bx, o_by, o_rad, o_vx, o_vy, o_cx, o_cy, o_brad, o_pvx, o_pvy, viax, viay;
  DevBuf<double> o_list;   // [5][max_obst]: x, y, radius, vx, vy in the order of the LDS obstacle cache (SceneDev::lox ..)
  // batch
  DevBuf<int> n, has_vs, has_vg, rotdir, via_en, status, optimized, iters, last_iters, trials, assoc_cnt, assoc, assoc_ovf, via_pose, legacy_idx;
  DevBuf<double> x, y, th, dt, vs, vg, chi2, cost, lambda, Hbackup, Hband, ob_x, ob_y, ob_th, ob_dt;   // ob_*: strips before an optimistic launch
  DevBuf<int> ob_n;
  DevBuf<long long> clk;   // BatchDev::clk
  // multi-CU mode (teb_multicu.hpp): control words, published poses, distance records; sized on first use
  DevBuf<unsigned> mcu_ctl;
  DevBuf<double> pack_dev;          // packed messages of small batches (kPackMaxDoubles)
  double* pack_host = nullptr;      // its pinned host side
  DevBuf<double> mcu_pub, mcu_items, mcu_spec;
  size_t mcu_items_have = 0;
  int mcu_last_helpers = 0;   // distance helpers per band of the last launch (0: none), teb_amd_last_launch_info
  int mcu_last_solvers = 0;   // solver helpers per band of the last launch
  int mcu_last_repeated = 0;  // that launch timed out waiting for its helpers and was repeated on one CU per band
  // back-off of the distance helpers: after a launch whose helpers came late (a device busy with other work) the next
  // mcu_backoff_left launches run on one CU per band without asking; then one launch probes again. Every further miss doubles the pause
  // (4, 8, .. 256 launches), a probe that succeeds clears it.
  int mcu_backoff_left = 0;
  int mcu_backoff_len = 0;
  unsigned* mcu_trace = nullptr;   // teb_amd_debug_mcu_watchdog: host-pinned breadcrumbs of the multi-CU launch, one word per workgroup
  int mcu_watchdog_ms = 0;
  int mcu_debug_flags = 0;
  DevBuf<double> iter_log;   // teb_amd_set_iteration_log: [max_tebs][TEB_AMD_ITERATION_LOG_ROWS][4]
  bool iter_log_on = false;
  DevBuf<double> phase_log;  // teb_amd_set_phase_log: [max_tebs][kPhaseLogSlots]
  bool phase_log_on = false;
  bool opt_backup_ready = false;
  size_t hband_stride = 0;
  // snapshot
  DevBuf<int> snap_n;
  DevBuf<double> snap_x, snap_y, snap_th, snap_dt;
  // debug / select
  DevBuf<double> dbg_H, dbg_b, dbg_chi2, sel_cost;
  DevBuf<int> sel_idx, err_flag;
  // strip producers / consumers (f1, f2)
  DevBuf<double> stage_x, stage_y, stage_yaw, out_cmd, out_prof, out_traj;
  bool consumers_valid = false;
  // equivalence classes (f3)
  DevBuf<double> hsig, hs_pre, hs_pim;
  DevBuf<int> hs_pex;
  std::vector<double> hsig_host;
  int hsig_mode = 0, hsig_B = 0, hsig_M = 0;
  bool hs_prod_valid = false;   // hs_pre / hs_pim / hs_pex hold the products of the current obstacle table
  double hsig_prescaler = 0;
  int consumers_la = 0, consumers_prevent = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  std::vector<int> host_type;
  // candidate generation (f3): host copy of the obstacle centroids, graph + candidate scratch strips, boost-compatible generator
  std::vector<double> host_cx, host_cy;
  DevBuf<double> g_vx, g_vy, cand_x, cand_y, cand_th, cand_dt, cand_sig, cand_px, cand_py, tmp_x, tmp_y, tmp_th, tmp_dt, tmp_vs, tmp_vg,
      tmp_chi2, tmp_cost, tmp_lambda;
  DevBuf<int> cand_n, cand_off, cand_map, tmp_n, tmp_i;
  DevBuf<unsigned char> g_adj;
  size_t g_cap = 0;
  bool cand_ready = false, tmp_ready = false;
  std::vector<double> best_class;   // best_teb_eq_class_ (homotopy_class_planner.h): signature of the last best band; survives the band
  int best_class_mode = 0;          // 0 = none yet, 2 / 3 = HSignature / HSignature3d
  std::vector<double> initial_class;   // initial_plan_eq_class_: signature of the band made from the last initial plan
  int initial_class_mode = 0;
  // costmap (f4)
  DevBuf<unsigned char> cm_cells;
  DevBuf<double> cm_fp;
  DevBuf<int> cm_out;
  int cm_sx = 0, cm_sy = 0;
  double cm_res = 0, cm_ox = 0, cm_oy = 0;
  std::mt19937 rnd_generator;   // ProbRoadmapGraph::rnd_generator_ (graph_search.h:211): default-seeded 32-bit Mersenne twister
};

namespace {

int check_handle(teb_amd_handle* h) {
  if (!h) return fail(TEB_AMD_ERR_INVALID_ARG, "null handle");
  if (hipSetDevice(h->device) != hipSuccess) return fail(TEB_AMD_ERR_HIP, "hipSetDevice failed");
  return TEB_AMD_OK;
}

SceneDev scene_of(teb_amd_handle* h) {
  SceneDev s;
  s.M = h->M;
  s.fast_points = h->fast_points;
  s.static_radius_zero = h->static_radius_zero;
  s.type = h->o_type.p; s.ax = h->o_ax.p; s.ay = h->o_ay.p; s.bx = h->o_bx.p; s.by = h->o_by.p;
  s.rad = h->o_rad.p; s.vx = h->o_vx.p; s.vy = h->o_vy.p; s.cx = h->o_cx.p; s.cy = h->o_cy.p; s.brad = h->o_brad.p;
  s.dyn = h->o_dyn.p; s.voff = h->o_voff.p; s.pvx = h->o_pvx.p; s.pvy = h->o_pvy.p;
  s.n_static = h->n_static; s.static_idx = h->o_static.p; s.n_dyn = h->n_dyn; s.dyn_idx = h->o_dynidx.p;
  s.nvia = h->nvia; s.viax = h->viax.p; s.viay = h->viay.p;
  const size_t mo = (size_t)(h->max_obst > 0 ? h->max_obst : 1);
  s.lox = h->o_list.p; s.loy = h->o_list.p + mo; s.lor = h->o_list.p + 2 * mo; s.lovx = h->o_list.p + 3 * mo; s.lovy = h->o_list.p + 4 * mo;
  return s;
}

BatchDev batch_of(teb_amd_handle* h) {
  BatchDev b;
  b.B = h->B; b.stride = h->stride;
  b.n = h->n.p; b.x = h->x.p; b.y = h->y.p; b.th = h->th.p; b.dt = h->dt.p;
  b.has_vs = h->has_vs.p; b.vs = h->vs.p; b.has_vg = h->has_vg.p; b.vg = h->vg.p;
  b.rotdir = h->rotdir.p; b.via_en = h->via_en.p;
  b.status = h->status.p; b.optimized = h->optimized.p; b.iters = h->iters.p; b.last_iters = h->last_iters.p; b.trials = h->trials.p;
  b.chi2 = h->chi2.p; b.cost = h->cost.p; b.lambda = h->lambda.p;
  b.assoc_cnt = h->assoc_cnt.p; b.assoc = h->assoc.p; b.assoc_cap = h->max_obst > 0 ? h->max_obst : 1;
  b.assoc_overflow = h->assoc_ovf.p; b.legacy_idx = h->legacy_idx.p;
  b.via_pose = h->via_pose.p; b.via_cap = h->max_via > 0 ? h->max_via : 1;
  b.Hbackup = h->Hbackup.p; b.hmat_stride = h->hmat_stride;
  b.clk = h->clk.p;
  return b;
}

int validate_config(const teb_amd_config_t* c) {
  if (c->footprint_type < TEB_AMD_FOOTPRINT_POINT || c->footprint_type > TEB_AMD_FOOTPRINT_POLYGON)
    return fail(TEB_AMD_ERR_INVALID_ARG, "unknown footprint_type");
  if (c->footprint_type == TEB_AMD_FOOTPRINT_LINE && c->footprint_n_vertices != 2)
    return fail(TEB_AMD_ERR_INVALID_ARG, "line footprint needs exactly 2 vertices");
  if (c->footprint_type == TEB_AMD_FOOTPRINT_POLYGON &&
      (c->footprint_n_vertices < 1 || c->footprint_n_vertices > TEB_AMD_MAX_FOOTPRINT_VERTICES))
    return fail(TEB_AMD_ERR_INVALID_ARG, "polygon footprint needs 1..64 vertices");
  if (c->jacobian_mode != TEB_AMD_JACOBIAN_ANALYTIC && c->jacobian_mode != TEB_AMD_JACOBIAN_G2O_NUMERIC)
    return fail(TEB_AMD_ERR_INVALID_ARG, "jacobian_mode must be TEB_AMD_JACOBIAN_ANALYTIC or TEB_AMD_JACOBIAN_G2O_NUMERIC");
  return TEB_AMD_OK;
}

const void* opt_kernel(int solver, int jmode, int scene) {
#define TEB_OPT_PICK(S, J, P) if (solver == S && jmode == J && scene == P) return TEB_OPT_KERNEL_FN(S, J, P)();
  TEB_OPT_FOR_ALL(TEB_OPT_PICK)
#undef TEB_OPT_PICK
  return nullptr;   // (a -DTEB_AMD_ANALYTIC_ONLY build asked for the numeric mode)
}
// Which specialised instantiation may run this launch: 1 = the *_DEFAULTS kinds (every flag of the profile table folded), 2 = the *_WIDE
// kinds (point-like scenes: every fold but the via-points and the holonomic choice), 3 = the *_LIGHT kinds (point-like scenes: every
// cost-term flag at run time, only the never-reached bulk folded), 0 = none (generic instantiation). GENERATED from the
// table of teb_device.hpp (TEB_PF_ALL): a flag is folded in the kernel exactly when its TEB_PF_HOST_<ID> condition is required here, so a
// fold cannot exist without its host-side check (tests/test_config_profile_sites.py). `c`, `args`, `sc` are the names the table's
// expressions use - the very objects the kernel is launched with.
int profile_matches(const teb_amd_handle* h, const OptArgs& args, const SceneDev& sc) {
  if (h->opt.generic_config_path) return 0;
  const teb_amd_config_t& c = h->cfg;
  const bool points = sc.fast_points != 0;   // (generic-shape kinds keep the TEB_PF_KIN_* flags at run time)
  bool narrow = true, wide = true, light = true;
#define TEB_PF_CHECK(ID)                                                        \
  if ((points || !TEB_PF_KIN_##ID) && !(TEB_PF_HOST_##ID)) {                    \
    narrow = false;                                                             \
    if (!TEB_PF_WIDE_##ID) wide = false;                                        \
    if (!TEB_PF_LIGHT_##ID) light = false;                                      \
  }
  TEB_PF_ALL(TEB_PF_CHECK)
#undef TEB_PF_CHECK
  return narrow ? 1 : (wide && points ? 2 : (light && points ? 3 : 0));
}
// The instantiation compiled at run time for this launch (teb_amd_options_t::compile_for_config), READY, or null: not asked for, a default
// configuration (pf == 1: its pre-built kernel IS the fully folded one), not ready yet, or failed. wait: block until the compiler is done.
std::shared_ptr<RtcKernel> rtc_lookup(const teb_amd_handle* h, const OptArgs& args, const SceneDev& sc, int solver, bool small, int pf, bool wait) {
  if (!((pf != 1 || h->opt.compile_for_config >= 3) && h->opt.compile_for_config > 0 && !h->opt.generic_config_path && !args.debug_linearize)) return nullptr;   // (3: even for a default configuration - measurement only)
  const teb_amd_config_t& c = h->cfg;
#ifdef TEB_AMD_ANALYTIC_ONLY
  if (c.jacobian_mode != TEB_AMD_JACOBIAN_ANALYTIC) return nullptr;   // this variant has no numeric mode: the run-time compiler must not add one
#endif
  RtcKey key;
  key.flags = 0;
  int bit = 0;
#define TEB_PF_BIT(ID) key.flags |= (unsigned long long)((TEB_PF_EXPR_##ID) ? 1 : 0) << bit; ++bit;
  TEB_PF_ALL(TEB_PF_BIT)
#undef TEB_PF_BIT
  const bool sm = small && c.jacobian_mode == TEB_AMD_JACOBIAN_ANALYTIC;
  if (small && !sm) return nullptr;   // (helper workgroups exist in the small-batch kinds only, which exist for closed-form Jacobians)
  key.solver = solver; key.jmode = c.jacobian_mode;
  key.scene = sc.fast_points ? (sm ? SCENE_POINTS_SMALL_CUSTOM : SCENE_POINTS_CUSTOM) : (sm ? SCENE_GENERIC_SMALL_CUSTOM : SCENE_GENERIC_CUSTOM);
  std::string why;
  std::shared_ptr<RtcKernel> rk = rtc_request(key, wait, &why);
  return (rk && rk->state.load() == RtcKernel::READY) ? rk : nullptr;
}
hipError_t launch_opt(teb_amd_handle* h, int grid, const SceneDev& sc, const BatchDev& bt, const OptArgs& a, int solver, const LdsPlan& plan,
                      const McuDev* mcu = nullptr) {
  McuDev none;
  std::memset(&none, 0, sizeof none);
  const McuDev* mc = mcu ? mcu : &none;
  const bool small = mc->K + mc->D > 0;   // helper workgroups: the small-batch instantiation of the scene kind
  const void* k = nullptr;
  h->last_defaults_profile = 0;
  const int pf = profile_matches(h, a, sc);   // (a build without the twins, or a mode they do not exist for, returns null: generic instantiation)
  int scene_kind = -1;   // which instantiation runs (teb_amd_debug_last_instantiation)
  if (pf != 0) {
    const bool sm = small && h->cfg.jacobian_mode == TEB_AMD_JACOBIAN_ANALYTIC;
    if (pf == 1) scene_kind = sc.fast_points ? (sm ? SCENE_POINTS_SMALL_DEFAULTS : SCENE_POINTS_DEFAULTS) : (sm ? SCENE_GENERIC_SMALL_DEFAULTS : SCENE_GENERIC_DEFAULTS);
    else if (pf == 2) scene_kind = sm ? SCENE_POINTS_SMALL_WIDE : SCENE_POINTS_WIDE;
    else scene_kind = sm ? SCENE_POINTS_SMALL_LIGHT : SCENE_POINTS_LIGHT;
    k = opt_kernel(solver, h->cfg.jacobian_mode, scene_kind);
    if (k) h->last_defaults_profile = pf;
  }
  h->last_inst[0] = solver; h->last_inst[1] = h->cfg.jacobian_mode; h->last_inst[2] = -1;
  void* params[] = {const_cast<teb_amd_config_t*>(&h->cfg), const_cast<SceneDev*>(&sc), const_cast<BatchDev*>(&bt), const_cast<OptArgs*>(&a),
                    const_cast<LdsPlan*>(&plan), const_cast<McuDev*>(mc)};
  // a configuration off the defaults: the instantiation compiled for IT at run time, once it is ready (teb_rtc.hpp)
  if (std::shared_ptr<RtcKernel> rk = rtc_lookup(h, a, sc, solver, small, pf, false)) {
    std::string why;
    hipFunction_t f = rtc_function(*rk, h->device, h->lds_limit, &why);
    if (f) {
      h->last_defaults_profile = 4;
      h->last_inst[2] = sc.fast_points ? (small ? SCENE_POINTS_SMALL_CUSTOM : SCENE_POINTS_CUSTOM) : (small ? SCENE_GENERIC_SMALL_CUSTOM : SCENE_GENERIC_CUSTOM);
      return hipModuleLaunchKernel(f, grid * (1 + mc->K + mc->D), 1, 1, kThreads, 1, 1, plan.total_bytes, h->stream, params, nullptr);
    }
  }
  if (!k) {
    scene_kind = sc.fast_points ? (small ? SCENE_POINTS_SMALL : SCENE_POINTS) : (small ? SCENE_GENERIC_SMALL : SCENE_GENERIC);
    k = opt_kernel(solver, h->cfg.jacobian_mode, scene_kind);
    h->last_defaults_profile = 0;
  }
  if (!k) return hipErrorInvalidDeviceFunction;
  h->last_inst[2] = scene_kind;
  return hipLaunchKernel(k, dim3(grid * (1 + mc->K + mc->D)), dim3(kThreads), params, plan.total_bytes, h->stream);
}
hipError_t launch_opt(teb_amd_handle* h, int grid, const SceneDev& sc, const BatchDev& bt, const OptArgs& a) {
  return launch_opt(h, grid, sc, bt, a, h->solver, h->plan);
}

// Multi-CU mode (teb_multicu.hpp): helper workgroups per band of this launch - K solver helpers (speculative LM trials, any scene kind)
// and D distance helpers (generic scenes: association + distance records). Small batches only: one workgroup per CU (the LDS footprint
// of the layouts) means B (1 + K + D) <= number of CUs keeps every workgroup resident. Closed-form Jacobians (the small-batch
// instantiations); the solver helpers need a layout whose normal matrix another workgroup can pick up (blocks in LDS with their HBM
// backup, or the band copy of the hybrid solve), the distance helpers the new association.
constexpr size_t kMcuItemsMaxBytes = (size_t)1 << 30;   // distance records of the multi-CU mode: beyond 1 GiB the launch stays on one CU per band
void mcu_helpers_for(teb_amd_handle* h, const OptArgs& args, int eff_solver, int* K_out, int* D_out) {
  *K_out = 0; *D_out = 0;
  if (args.debug_linearize || h->cfg.jacobian_mode != TEB_AMD_JACOBIAN_ANALYTIC || h->B <= 0) return;
  const int cus = h->num_cus > 0 ? h->num_cus : 256;
  const int room = cus / h->B - 1;   // helper workgroups per band that still leave every workgroup its own CU
  int K = 0, D = 0;
  if (h->opt.speculative_trials >= 0 && eff_solver != SOLVER_BANDG && !(eff_solver == SOLVER_BAND && h->band_ldlt) && args.inner > 0) {
    // automatic: as many of the three retries as the batch leaves CUs for (<= 64 bands: all three, <= 128: the first - the common - one)
    K = h->opt.speculative_trials > 0 ? std::min((int)h->opt.speculative_trials, kMcuMaxSpec) : kMcuMaxSpec;
    K = std::max(0, std::min(K, room));
  }
  if (h->opt.multi_cu >= 0 && !h->fast_points && !h->cfg.legacy_obstacle_association && h->M > 0) {
    D = std::min(room - K, 60);                               // beyond ~ 5 poses per helper the hand-off costs more than the tile
    // ... and that holds for short bands as well: at most one helper per 6 poses of the handle's capacity (round 6; until then a 60-pose
    // band could be given 60 helpers with a pose each). This is also where the one known defect of the mode lives: with 40 - 60 helpers on
    // 57 .. 62-pose bands 0.4 - 2.4 % of the launches returned ONE band off the single-CU result by |d chi2| ~ 1e-3 .. 1e-1 (rounds 4 - 6
    // binaries alike; more often the faster the kernel). Not found in round 6: agent-scope release / acquire fences at every hand-over,
    // fine-grained (uncached) hand-over buffers, returning atomics for the records, a poisoned pose buffer (never read stale) and an exact
    // arrival count changed nothing; a 14 us pause in front of the replay reduces it 20 x. At <= 20 helpers: 0 - 2 of 2000 launches, at
    // <= 12 none observed; C5 (53 - 60 helpers x 5 - 6 poses): 0 of 1400 (profiles/mcu_race_r06.txt).
    D = std::min(D, std::max(2, h->stride / 6));   // (the pose capacity: the bands grow under autoResize, what the host knows is where they started)
    if (h->opt.multi_cu > 0) D = std::min(D, (int)h->opt.multi_cu);
    // auto: enough (pose, obstacle) work - and ONE band: the defect above needs at least two bands in the launch (any single band of that
    // scene with 60 helpers: 0 of 9000 launches; two bands 8 of 3000, three 18 of 3000; the bands' buffers do not overlap - padding them
    // changed nothing - so what the bands share is not understood). multi_cu > 0 still asks for helpers on any batch (the tests do).
    else if (h->B > 1 || (size_t)h->M * (size_t)(h->nmax_known > 0 ? h->nmax_known : h->stride) < 4096) D = 0;
    if (D < 2) D = 0;
  }
  *K_out = K; *D_out = D;
}

// largest pose capacity whose LDS plan (with the obstacle cache of this scene, if it is in use) fits
int max_capacity_of(teb_amd_handle* h, int solver, int upto) {
  const int ob = h->fast_points ? h->M : 0;
  int S = upto;
  while (S > 8 && (size_t)make_lds_plan(S, solver, ob, h->lds_limit).total_bytes > h->lds_limit) --S;
  return (size_t)make_lds_plan(S, solver, ob, h->lds_limit).total_bytes <= h->lds_limit ? S : 0;
}

// the four strips and the pose counts in ONE launch (five copy commands cost more in launch gaps than in bytes: 2.4 MB at the headline)
__global__ void __launch_bounds__(256) copy_state_kernel(double* __restrict__ x, double* __restrict__ y, double* __restrict__ th,
                                                         double* __restrict__ dt, int* __restrict__ n, const double* __restrict__ sx,
                                                         const double* __restrict__ sy, const double* __restrict__ sth,
                                                         const double* __restrict__ sdt, const int* __restrict__ sn, size_t count, int B) {
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += step) {
    x[i] = sx[i]; y[i] = sy[i]; th[i] = sth[i]; dt[i] = sdt[i];
    if (i < (size_t)B) n[i] = sn[i];
  }
}
// Small batches (a planner tick, one TebOptimalPlanner): a dozen tiny copies cost more in host calls than in bytes. The bands and their
// attributes travel as ONE packed message through a pinned host buffer and ONE copy; these kernels scatter / gather it on the device.
// Message (doubles; the integer attributes are exact as doubles): [n | has_vs | has_vg | rotdir | via_en] B each, [vs | vg] 3 B each,
// then x | y | theta | dt, B rows of w columns each.
constexpr size_t kPackMaxDoubles = 65536;   // 512 KB: beyond that the strided copies win
__global__ void __launch_bounds__(256) unpack_bands_kernel(const double* __restrict__ msg, int B, int w, int S, int* n, int* hvs, int* hvg, int* rd, int* ve,
                                                           double* vs, double* vg, double* x, double* y, double* th, double* dt, int* optimized) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < B) { n[t] = (int)msg[t]; hvs[t] = (int)msg[B + t]; hvg[t] = (int)msg[2 * B + t]; rd[t] = (int)msg[3 * B + t]; ve[t] = (int)msg[4 * B + t]; optimized[t] = 0; }
  if (t < 3 * B) { vs[t] = msg[5 * B + t]; vg[t] = msg[8 * B + t]; }
  const double* body = msg + 11 * (size_t)B;
  const size_t plane = (size_t)B * w;
  for (size_t q = t; q < plane; q += (size_t)gridDim.x * blockDim.x) {
    const size_t b = q / w, i = q - b * w, o = b * S + i;
    x[o] = body[q]; y[o] = body[plane + q]; th[o] = body[2 * plane + q]; dt[o] = body[3 * plane + q];
  }
}
__global__ void __launch_bounds__(256) pack_bands_kernel(double* __restrict__ msg, int B, int w, int S, const int* n, const double* x, const double* y,
                                                         const double* th, const double* dt) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < B) msg[t] = (double)n[t];
  double* body = msg + B;
  const size_t plane = (size_t)B * w;
  for (size_t q = t; q < plane; q += (size_t)gridDim.x * blockDim.x) {
    const size_t b = q / w, i = q - b * w, o = b * S + i;
    body[q] = x[o]; body[plane + q] = y[o]; body[2 * plane + q] = th[o]; body[3 * plane + q] = dt[o];
  }
}
// [status | lm_iterations | lm_trials | chi2 | cost | lambda] B each
__global__ void __launch_bounds__(256) pack_results_kernel(double* __restrict__ msg, int B, const int* status, const int* iters, const int* trials, const double* chi2,
                                                           const double* cost, const double* lambda) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < B) { msg[t] = (double)status[t]; msg[B + t] = (double)iters[t]; msg[2 * B + t] = (double)trials[t]; msg[3 * B + t] = chi2[t]; msg[4 * B + t] = cost[t]; msg[5 * B + t] = lambda[t]; }
}

struct Strips { double *x, *y, *th, *dt; int* n; };
static int copy_strips(teb_amd_handle_t* h, const Strips& dst, const Strips& src, int bands) {
  const size_t count = (size_t)bands * h->stride;   // >= bands
  const int grid = (int)std::min<size_t>(2048, (count + 255) / 256);
  hipLaunchKernelGGL(copy_state_kernel, dim3(grid), dim3(256), 0, h->stream, dst.x, dst.y, dst.th, dst.dt, dst.n, src.x, src.y, src.th,
                     src.dt, src.n, count, bands);
  HIPCHK(hipGetLastError());
  return TEB_AMD_OK;
}

int launch(teb_amd_handle* h, const OptArgs& args) {
  if (h->B <= 0) return fail(TEB_AMD_ERR_INVALID_ARG, "no TEBs uploaded");
  SceneDev sc = scene_of(h);
  BatchDev bt = batch_of(h);
  // The layout of the normal matrix follows the pose capacity of the handle (blocks in LDS <= 238 poses, band in LDS <= 337, band in
  // HBM beyond), the faster layouts only hold shorter bands. A handle created for long bands that currently holds short ones is
  // launched in the fastest layout that leaves the bands 10 % room to grow (autoResize); should a band outgrow it all the same, the
  // launch is repeated from the saved strips in the handle's own layout. Results do not depend on the layout (same arithmetic up to
  // the order of the block reduction). TEB_AMD_FIXED_LAYOUT=1 switches this off.
  int eff_solver = h->solver;
  LdsPlan eff_plan = h->plan;
  bool optimistic = false;
  if (h->solver != SOLVER_CR && !args.debug_linearize && !h->opt.fixed_layout && h->opt.layout == TEB_AMD_LAYOUT_AUTO && !h->band_ldlt) {
    int nmax = h->nmax_known;
    if (nmax < 0) {   // a device-side producer changed the bands since the host last saw their pose counts
      std::vector<int> n(h->B);
      HIPCHK(hipMemcpyAsync(n.data(), h->n.p, h->B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      nmax = 0;
      for (int v : n) nmax = std::max(nmax, v);
    }
    const int need = nmax + nmax / 10 + 4;   // 10 % room to grow; a band that needs more triggers the repeat below
    const int ob = h->fast_points ? h->M : 0;
    const int s_cr = max_capacity_of(h, SOLVER_CR, std::min(h->stride, 238));
    const int s_band = h->solver == SOLVER_BANDG ? max_capacity_of(h, SOLVER_BAND, std::min(h->stride, 337)) : 0;
    if (s_cr > 0 && need <= s_cr) { eff_solver = SOLVER_CR; eff_plan = make_lds_plan(s_cr, SOLVER_CR, ob, h->lds_limit); optimistic = true; }
    else if (s_band > 0 && need <= s_band) { eff_solver = SOLVER_BAND; eff_plan = make_lds_plan(s_band, SOLVER_BAND, ob, h->lds_limit); optimistic = true; }
  }
  // multi-CU mode: helper workgroups per band (0 = none); its buffers, and the control words zeroed on the stream before the launch
  int K = 0, D = 0;
  mcu_helpers_for(h, args, eff_solver, &K, &D);
  if (D > 0 && h->mcu_backoff_left > 0) { D = 0; --h->mcu_backoff_left; }   // paused after a miss: one CU per band, no wait, no repeat
  if (D > 0) {
    // distance records [B][M][4][stride]: bounded (the association list of a pose is short, but its capacity is every obstacle), and a
    // failed allocation is no error - the launch then runs on one CU per band
    const size_t need_items = (size_t)h->B * h->M * 4 * h->stride;
    if (need_items * sizeof(double) > kMcuItemsMaxBytes) D = 0;
    else if (h->mcu_items_have < need_items) {
      h->mcu_items.free(); h->mcu_items_have = 0;
      if (h->mcu_items.alloc(need_items) != hipSuccess) { (void)hipGetLastError(); D = 0; }
      else h->mcu_items_have = need_items;
    }
  }
  const int H = K + D;
  McuDev mcu;
  std::memset(&mcu, 0, sizeof mcu);
  if (H > 0) {
    if (!h->mcu_ctl.p) {
      HIPCHK(h->mcu_ctl.alloc((size_t)h->max_tebs * kMcuCtlWords)); HIPCHK(h->mcu_pub.alloc((size_t)h->max_tebs * kMcuPubArrays * h->stride));
      HIPCHK(h->mcu_spec.alloc((size_t)h->max_tebs * (kMcuMaxSpec + 1) * mcu_spec_slot(h->stride)));
    }
    HIPCHK(hipMemsetAsync(h->mcu_ctl.p, 0, (size_t)h->B * kMcuCtlWords * sizeof(unsigned), h->stream));
    mcu.K = K; mcu.D = D; mcu.ctl = h->mcu_ctl.p; mcu.pub = h->mcu_pub.p; mcu.items = D > 0 ? h->mcu_items.p : nullptr; mcu.item_cap = h->M; mcu.spec = h->mcu_spec.p;
    mcu.timeout_ticks = (long long)(h->opt.multi_cu_timeout_us > 0 ? h->opt.multi_cu_timeout_us : 2000) * 100LL;   // 100 MHz real-time counter
    // a probe after a pause waits 250 us at most: helpers that have CUs answer a phase within ~ 10 us, and a device that is still busy
    // should cost the tick as little as possible beyond its one-CU time
    if (D > 0 && h->mcu_backoff_len > 0) mcu.timeout_ticks = std::min(mcu.timeout_ticks, 25000LL);
    mcu.trace = h->mcu_trace;
    mcu.debug_flags = h->mcu_debug_flags;
    if (h->mcu_trace) std::memset(h->mcu_trace, 0, 1024 * sizeof(unsigned));
  }
  h->mcu_last_helpers = D; h->mcu_last_solvers = K; h->mcu_last_repeated = 0;
  const bool checked = optimistic || D > 0;   // (solver helpers alone cannot fail a band: a late one just means the band solves itself)   // the launch is followed by a look at the per-band flags (and possibly repeated)
  if (checked) {   // strips as they are now, for the repeat
    if (!h->opt_backup_ready) {
      const size_t BS = (size_t)h->max_tebs * h->stride;
      HIPCHK(h->ob_x.alloc(BS)); HIPCHK(h->ob_y.alloc(BS)); HIPCHK(h->ob_th.alloc(BS)); HIPCHK(h->ob_dt.alloc(BS)); HIPCHK(h->ob_n.alloc(h->max_tebs));
      h->opt_backup_ready = true;
    }
    if (int crc = copy_strips(h, Strips{h->ob_x.p, h->ob_y.p, h->ob_th.p, h->ob_dt.p, h->ob_n.p}, Strips{h->x.p, h->y.p, h->th.p, h->dt.p, h->n.p}, h->B)) return crc;
  }
  if (h->opt.compile_for_config >= 2)   // synchronous mode: the wait for the compiler is not kernel time
    (void)rtc_lookup(h, args, sc, eff_solver, H > 0, profile_matches(h, args, sc), true);
  HIPCHK(hipEventRecord(h->ev0, h->stream));   // (the kernel clears its bands' overflow flags itself)
  HIPCHK(launch_opt(h, h->B, sc, bt, args, eff_solver, eff_plan, H > 0 ? &mcu : nullptr));
  if (H > 0 && h->mcu_trace && h->mcu_watchdog_ms > 0) {
    // diagnostic: a multi-CU launch that does not finish in time is reported with the breadcrumbs of its workgroups, and the process ends
    // (a kernel cannot be cancelled; the bounded spins of the kernel make this unreachable unless the protocol itself is broken)
    const auto t0 = std::chrono::steady_clock::now();
    while (hipStreamQuery(h->stream) == hipErrorNotReady) {
      if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > h->mcu_watchdog_ms) {
        fprintf(stderr, "[teb_amd multi-CU watchdog] launch of %d bands x (1 + %d) workgroups still running after %d ms; breadcrumbs (epoch << 8 | code):\n", h->B, H, h->mcu_watchdog_ms);
        for (int w = 0; w < h->B * (1 + H) && w < 1024; ++w) fprintf(stderr, " wg%d:%x", w, h->mcu_trace[w]);
        fprintf(stderr, "\n");
        fflush(stderr);
        _exit(3);
      }
    }
  }
  if (checked) {
    HIPCHK(hipGetLastError());
    // this mode is synchronous: the overflow flags decide whether the launch has to be repeated (documented in teb_amd.h)
    std::vector<int> ovf(h->B), nn(h->B);
    HIPCHK(hipMemcpyAsync(ovf.data(), h->assoc_ovf.p, h->B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(nn.data(), h->n.p, h->B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    bool outgrown = false, helpers_late = false;
    for (int v : ovf) { outgrown = outgrown || (v & 2); helpers_late = helpers_late || (v & 4); }
    h->nmax_known = 0;
    for (int v : nn) h->nmax_known = std::max(h->nmax_known, v);
    if (D > 0) {
      if (helpers_late) { h->mcu_backoff_len = std::min(256, std::max(4, 2 * h->mcu_backoff_len)); h->mcu_backoff_left = h->mcu_backoff_len; }
      else { h->mcu_backoff_len = 0; h->mcu_backoff_left = 0; }
    }
    if (outgrown || helpers_late) {
      // from the saved strips: in the handle's own layout when a band outgrew the optimistic one, and on one CU per band when the
      // helpers of the multi-CU mode did not show up in time (a busy device: nothing promises their residency)
      if (int crc = copy_strips(h, Strips{h->x.p, h->y.p, h->th.p, h->dt.p, h->n.p}, Strips{h->ob_x.p, h->ob_y.p, h->ob_th.p, h->ob_dt.p, h->ob_n.p}, h->B)) return crc;
      bool own_layout = outgrown || !optimistic;
      if (!own_layout) {
        // helpers late, no band outgrown so far: once more in the optimistic layout - whose capacity a band may outgrow in THIS run
        // (the first one was cut short), so its flags are read as well
        HIPCHK(launch_opt(h, h->B, sc, bt, args, eff_solver, eff_plan));
        HIPCHK(hipMemcpyAsync(ovf.data(), h->assoc_ovf.p, h->B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        for (int v : ovf) own_layout = own_layout || (v & 2);
        if (own_layout)
          if (int crc = copy_strips(h, Strips{h->x.p, h->y.p, h->th.p, h->dt.p, h->n.p}, Strips{h->ob_x.p, h->ob_y.p, h->ob_th.p, h->ob_dt.p, h->ob_n.p}, h->B)) return crc;
      }
      if (own_layout) HIPCHK(launch_opt(h, h->B, sc, bt, args));   // ev0 stays where it was: the reported time includes the discarded launches
      h->mcu_last_repeated = helpers_late ? 1 : 0;
      h->nmax_known = -1;
    }
  } else if (h->cfg.teb_autosize && !args.debug_linearize) {
    h->nmax_known = -1;   // asynchronous launch: autoResize may change the pose counts
  }
  h->consumers_valid = false;
  h->hsig_mode = 0;   // the bands change: signatures of an earlier teb_amd_compute_h_signatures call are stale
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(h->ev1, h->stream));
  h->timed = true;
  return TEB_AMD_OK;
}

}  // namespace

extern "C" {

int teb_amd_abi_version(void) { return TEB_AMD_ABI_VERSION; }
const char* teb_amd_last_error(void) { return g_last_error.c_str(); }
int teb_amd_sizeof_config(void) { return (int)sizeof(teb_amd_config_t); }
int teb_amd_sizeof_obstacles(void) { return (int)sizeof(teb_amd_obstacles_t); }
int teb_amd_sizeof_teb_batch(void) { return (int)sizeof(teb_amd_teb_batch_t); }
int teb_amd_sizeof_results(void) { return (int)sizeof(teb_amd_results_t); }
int teb_amd_sizeof_options(void) { return (int)sizeof(teb_amd_options_t); }

void teb_amd_config_default(teb_amd_config_t* c) {   // TebConfig::TebConfig(), reference teb_config.h:245-390
  std::memset(c, 0, sizeof(*c));
  c->teb_autosize = 1; c->dt_ref = 0.3; c->dt_hysteresis = 0.1; c->min_samples = 3; c->max_samples = 500;
  c->exact_arc_length = 0; c->via_points_ordered = 0;
  c->max_vel_x = 0.4; c->max_vel_x_backwards = 0.2; c->max_vel_y = 0.0; c->max_vel_trans = 0.0; c->max_vel_theta = 0.3;
  c->acc_lim_x = 0.5; c->acc_lim_y = 0.5; c->acc_lim_theta = 0.5; c->min_turning_radius = 0;
  c->min_obstacle_dist = 0.5; c->inflation_dist = 0.6; c->dynamic_obstacle_inflation_dist = 0.6;
  c->include_dynamic_obstacles = 1; c->obstacle_poses_affected = 25; c->legacy_obstacle_association = 0;
  c->obstacle_association_force_inclusion_factor = 1.5; c->obstacle_association_cutoff_factor = 5;
  c->obstacle_proximity_ratio_max_vel = 1; c->obstacle_proximity_lower_bound = 0; c->obstacle_proximity_upper_bound = 0.5;
  c->no_inner_iterations = 5; c->no_outer_iterations = 4; c->optimization_activate = 1; c->penalty_epsilon = 0.05;
  c->weight_max_vel_x = 2; c->weight_max_vel_y = 2; c->weight_max_vel_theta = 1; c->weight_acc_lim_x = 1;
  c->weight_acc_lim_y = 1; c->weight_acc_lim_theta = 1; c->weight_kinematics_nh = 1000;
  c->weight_kinematics_forward_drive = 1; c->weight_kinematics_turning_radius = 1; c->weight_optimaltime = 1;
  c->weight_shortest_path = 0; c->weight_obstacle = 50; c->weight_inflation = 0.1; c->weight_dynamic_obstacle = 50;
  c->weight_dynamic_obstacle_inflation = 0.1; c->weight_velocity_obstacle_ratio = 0; c->weight_viapoint = 1;
  c->weight_prefer_rotdir = 50; c->weight_adapt_factor = 2.0; c->obstacle_cost_exponent = 1.0;
  c->selection_cost_hysteresis = 1.0; c->selection_prefer_initial_plan = 0.95; c->selection_obst_cost_scale = 100.0;
  c->selection_viapoint_cost_scale = 1.0; c->selection_alternative_time_cost = 0;
  c->divergence_detection_enable = 0; c->divergence_detection_max_chi_squared = 10;
  c->footprint_type = TEB_AMD_FOOTPRINT_POINT;
  c->jacobian_mode = TEB_AMD_JACOBIAN_ANALYTIC;
}

void teb_amd_options_default(teb_amd_options_t* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->struct_size = (int32_t)sizeof(*o);
}

int teb_amd_create(const teb_amd_config_t* cfg, int32_t max_tebs, int32_t max_poses, int32_t max_obstacles,
                   int32_t max_obstacle_vertices, int32_t max_via_points, int32_t device, void* stream,
                   teb_amd_handle_t** out) {
  return teb_amd_create_ex(cfg, max_tebs, max_poses, max_obstacles, max_obstacle_vertices, max_via_points, device, stream, nullptr, out);
}

int teb_amd_create_ex(const teb_amd_config_t* cfg, int32_t max_tebs, int32_t max_poses, int32_t max_obstacles,
                      int32_t max_obstacle_vertices, int32_t max_via_points, int32_t device, void* stream,
                      const teb_amd_options_t* options, teb_amd_handle_t** out) {
  teb_amd_options_t opt;
  teb_amd_options_default(&opt);
  if (options) {   // copy what the caller's (possibly older, shorter) struct holds
    const size_t have = options->struct_size > 0 ? (size_t)options->struct_size : sizeof(opt);
    std::memcpy(&opt, options, std::min(have, sizeof(opt)));
    opt.struct_size = (int32_t)sizeof(opt);
    if (opt.layout < TEB_AMD_LAYOUT_AUTO || opt.layout > TEB_AMD_LAYOUT_BAND_HBM || opt.hsig3d_kernel < 0 || opt.hsig3d_kernel > TEB_AMD_HSIG3D_SMALL)
      return fail(TEB_AMD_ERR_INVALID_ARG, "teb_amd_options_t: unknown layout / hsig3d_kernel");
  }
  if (!cfg || !out || max_tebs <= 0 || max_poses < 2 || max_obstacles < 0 || max_obstacle_vertices < 0 || max_via_points < 0)
    return fail(TEB_AMD_ERR_INVALID_ARG, "teb_amd_create: bad arguments");
  int rc = validate_config(cfg);
  if (rc) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(TEB_AMD_ERR_NO_DEVICE, "no HIP device visible (libteb_amd has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(TEB_AMD_ERR_INVALID_ARG, "device ordinal out of range");
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(TEB_AMD_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
  HIPCHK(hipSetDevice(device));
  // Normal matrix as 8x8 blocks in LDS (SOLVER_CR: cyclic reduction in place, fastest) when that fits TOGETHER with the LDS cache
  // of a full obstacle table of max_obstacles point-like entries; otherwise as a band in LDS (SOLVER_BAND: 44 instead of 70
  // doubles per pose) with the hybrid cyclic reduction (level 0 from a band-form copy in HBM) - within 2 % of the block layout
  // per step, keeps the obstacle cache and holds bands up to 337 poses.
  int solver = SOLVER_CR;
  // the kernel also owns a little static LDS (__syncthreads_or scratch): keep 1 KiB of head-room
  const size_t lds_limit = (size_t)prop.sharedMemPerBlock - kLdsHeadroomBytes;
  if (lds_bytes_for(max_poses, SOLVER_CR, lds_limit) > lds_limit) solver = SOLVER_BAND;
  else if ((size_t)make_lds_plan(max_poses, SOLVER_CR, max_obstacles > 0 ? max_obstacles : 0, lds_limit).total_bytes > lds_limit &&
           (size_t)make_lds_plan(max_poses, SOLVER_BAND, max_obstacles > 0 ? max_obstacles : 0, lds_limit).total_bytes <= lds_limit)
    solver = SOLVER_BAND;
  if (opt.layout == TEB_AMD_LAYOUT_BAND_LDS) solver = SOLVER_BAND;
  else if (opt.layout == TEB_AMD_LAYOUT_BAND_HBM) solver = SOLVER_BANDG;
  else if (opt.layout == TEB_AMD_LAYOUT_BLOCKS_LDS) {
    if (lds_bytes_for(max_poses, SOLVER_CR, lds_limit) > lds_limit) return fail(TEB_AMD_ERR_CAPACITY, "TEB_AMD_LAYOUT_BLOCKS_LDS: max_poses too large for the block layout");
    solver = SOLVER_CR;
  }
  // bands too long for the LDS band (> 337 poses; the reference's max_samples default is 500): the band form of the normal matrix
  // moves to HBM (SOLVER_BANDG: 44 doubles per pose, L2-resident), everything else stays as it is
  if (solver == SOLVER_BAND && lds_bytes_for(max_poses, SOLVER_BAND, lds_limit) > lds_limit) solver = SOLVER_BANDG;
  // more than two poses per lane: only the band-in-HBM instantiations are compiled for it (teb_device.hpp: kPoseIterBandHbm)
  static_assert(TEB_AMD_MAX_POSES <= kThreads * 4, "TEB_AMD_MAX_POSES of include/teb_amd.h exceeds four poses per lane");
  if (max_poses > kThreads * 2) solver = SOLVER_BANDG;   // (the LDS layouts end at 337 poses: the checks above have already moved it there)
  const size_t lds = lds_bytes_for(max_poses, solver, lds_limit);
  const int thread_limit = kThreads * (solver == SOLVER_BANDG ? kPoseIterBandHbm : 2);
  if (max_poses > thread_limit || lds > lds_limit) {
    char buf[256];
    std::snprintf(buf, sizeof buf, "max_poses=%d needs %zu B of LDS per workgroup (device limit %zu B, thread limit %d poses)",
                  max_poses, lds, lds_limit, thread_limit);
    return fail(TEB_AMD_ERR_CAPACITY, buf);
  }
  const size_t assoc_bytes = sizeof(int) * (size_t)max_tebs * max_poses * (size_t)(max_obstacles > 0 ? max_obstacles : 1);
  if (assoc_bytes > ((size_t)64 << 30)) return fail(TEB_AMD_ERR_CAPACITY, "association table would exceed 64 GiB");

  teb_amd_handle* h = new teb_amd_handle();
  h->device = device;
  h->cfg = *cfg;
  h->opt = opt;
  h->max_tebs = max_tebs; h->stride = max_poses; h->max_obst = max_obstacles; h->max_verts = max_obstacle_vertices;
  h->max_via = max_via_points;
  h->lds_bytes = lds;
  h->solver = solver; h->solver_created = solver;
  h->lds_limit = lds_limit;
  h->num_cus = prop.multiProcessorCount;
  h->plan = make_lds_plan(max_poses, solver, 0, h->lds_limit);
  // per-band HBM buffer: the copy of H for rejected trials (SOLVER_CR, banded LDL^T) or the blocks the cyclic reduction of a
  // SOLVER_BAND handle works on (D, L, f: nb * (2 * kBlk + 8) doubles)
  h->hmat_stride = std::max(hbm_scratch_doubles(max_poses, SOLVER_BAND), hbm_scratch_doubles(max_poses, solver));   // every layout may be launched
  h->hmat_stride = (h->hmat_stride + 1) & ~(size_t)1;   // every band's slice 16-byte aligned (the multi-CU mode writes it with 16-byte stores)
  h->band_ldlt = opt.band_ldlt != 0;
  if (solver == SOLVER_BANDG) h->band_ldlt = 0;   // the sequential LDL^T works in place on an LDS band only
  if (stream) { h->stream = reinterpret_cast<hipStream_t>(stream); h->own_stream = false; }
  else {
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return fail(TEB_AMD_ERR_HIP, "hipStreamCreate failed"); }
    h->own_stream = true;
  }
  const size_t BS = (size_t)max_tebs * max_poses;
  const size_t Mo = max_obstacles > 0 ? max_obstacles : 1;
  bool ok = true;
  auto A = [&](hipError_t e) { if (e != hipSuccess) ok = false; };
  A(h->o_type.alloc(Mo)); A(h->o_dyn.alloc(Mo)); A(h->o_voff.alloc(Mo + 1)); A(h->o_static.alloc(Mo)); A(h->o_dynidx.alloc(Mo));
  A(h->o_list.alloc(5 * (size_t)Mo));
  A(h->o_ax.alloc(Mo)); A(h->o_ay.alloc(Mo)); A(h->o_bx.alloc(Mo)); A(h->o_by.alloc(Mo)); A(h->o_rad.alloc(Mo));
  A(h->o_vx.alloc(Mo)); A(h->o_vy.alloc(Mo)); A(h->o_cx.alloc(Mo)); A(h->o_cy.alloc(Mo)); A(h->o_brad.alloc(Mo));
  A(h->o_pvx.alloc(max_obstacle_vertices)); A(h->o_pvy.alloc(max_obstacle_vertices));
  A(h->viax.alloc(max_via_points)); A(h->viay.alloc(max_via_points));
  A(h->n.alloc(max_tebs)); A(h->has_vs.alloc(max_tebs)); A(h->has_vg.alloc(max_tebs)); A(h->rotdir.alloc(max_tebs));
  A(h->via_en.alloc(max_tebs)); A(h->status.alloc(max_tebs)); A(h->optimized.alloc(max_tebs)); A(h->iters.alloc(max_tebs)); A(h->last_iters.alloc(max_tebs)); A(h->trials.alloc(max_tebs));
  A(h->assoc_cnt.alloc(BS)); A(h->assoc.alloc(BS * Mo)); A(h->assoc_ovf.alloc(max_tebs)); A(h->legacy_idx.alloc((size_t)max_tebs * Mo));
  A(h->via_pose.alloc((size_t)max_tebs * (max_via_points > 0 ? max_via_points : 1)));
  A(h->x.alloc(BS)); A(h->y.alloc(BS)); A(h->th.alloc(BS)); A(h->dt.alloc(BS));
  A(h->vs.alloc(3 * (size_t)max_tebs)); A(h->vg.alloc(3 * (size_t)max_tebs));
  A(h->chi2.alloc(max_tebs)); A(h->cost.alloc(max_tebs)); A(h->lambda.alloc(max_tebs));
  A(h->Hbackup.alloc((size_t)max_tebs * h->hmat_stride));
  A(h->clk.alloc(4));
  h->hband_stride = solver == SOLVER_BANDG ? (((size_t)hbo(4 * max_poses) + 2 + 1) & ~(size_t)1) : 0;   // even: the bands' slices are zeroed in 16-byte stores
  A(h->Hband.alloc((size_t)max_tebs * h->hband_stride));
  A(h->snap_n.alloc(max_tebs)); A(h->snap_x.alloc(BS)); A(h->snap_y.alloc(BS)); A(h->snap_th.alloc(BS)); A(h->snap_dt.alloc(BS));
  A(h->dbg_H.alloc((size_t)4 * max_poses * kBand)); A(h->dbg_b.alloc((size_t)4 * max_poses)); A(h->dbg_chi2.alloc(4));
  A(h->sel_cost.alloc(2)); A(h->sel_idx.alloc(1)); A(h->err_flag.alloc(1));
  A(h->pack_dev.alloc(kPackMaxDoubles));
  if (ok && hipHostMalloc(reinterpret_cast<void**>(&h->pack_host), kPackMaxDoubles * sizeof(double), hipHostMallocDefault) != hipSuccess) h->pack_host = nullptr;   // (optional: without it the strided copies are used)
  A(h->stage_x.alloc((size_t)max_poses + 2)); A(h->stage_y.alloc((size_t)max_poses + 2)); A(h->stage_yaw.alloc((size_t)max_poses + 2));
  A(h->out_cmd.alloc(4 * (size_t)max_tebs)); A(h->out_prof.alloc((size_t)max_tebs * (max_poses + 1) * 3)); A(h->out_traj.alloc(BS * 7));
  A(h->hsig.alloc((size_t)max_tebs * (Mo > 2 ? Mo : 2))); A(h->hs_pre.alloc(Mo)); A(h->hs_pim.alloc(Mo)); A(h->hs_pex.alloc(Mo));
  if (ok && hipEventCreate(&h->ev0) != hipSuccess) ok = false;
  if (ok && hipEventCreate(&h->ev1) != hipSuccess) ok = false;
  for (int sv : {SOLVER_BAND, SOLVER_CR, SOLVER_BANDG})   // every layout may be launched (teb_amd_set_obstacles / per-launch choice)
    for (int jm : {TEB_AMD_JACOBIAN_ANALYTIC, TEB_AMD_JACOBIAN_G2O_NUMERIC})
      for (int sk : {SCENE_POINTS, SCENE_GENERIC, SCENE_POINTS_SMALL, SCENE_GENERIC_SMALL, SCENE_POINTS_DEFAULTS, SCENE_POINTS_SMALL_DEFAULTS, SCENE_GENERIC_DEFAULTS, SCENE_GENERIC_SMALL_DEFAULTS}) {
        const void* k = opt_kernel(sv, jm, sk);
        if (ok && k && hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit) != hipSuccess) ok = false;
      }
  if (ok && hipMemset(h->cost.p, 0, sizeof(double) * max_tebs) != hipSuccess) ok = false;
  if (ok && hipMemset(h->chi2.p, 0, sizeof(double) * max_tebs) != hipSuccess) ok = false;
  if (ok && hipMemset(h->iters.p, 0, sizeof(int) * max_tebs) != hipSuccess) ok = false;   // hasDiverged: "no statistics yet"
  if (ok && hipMemset(h->last_iters.p, 0, sizeof(int) * max_tebs) != hipSuccess) ok = false;
  if (ok && hipMemset(h->status.p, 0, sizeof(int) * max_tebs) != hipSuccess) ok = false;
  if (ok && hipMemset(h->optimized.p, 0, sizeof(int) * max_tebs) != hipSuccess) ok = false;
  if (!ok) { teb_amd_destroy(h); return fail(TEB_AMD_ERR_HIP, "device allocation / kernel attribute setup failed"); }
  *out = h;
  return TEB_AMD_OK;
}

void teb_amd_destroy(teb_amd_handle_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  DevBuf<int>* ib[] = {&h->o_type, &h->o_dyn, &h->o_voff, &h->o_static, &h->o_dynidx, &h->n, &h->has_vs, &h->has_vg, &h->rotdir,
                       &h->via_en, &h->status, &h->optimized, &h->iters, &h->last_iters, &h->trials, &h->assoc_cnt, &h->assoc, &h->assoc_ovf, &h->via_pose, &h->legacy_idx,
                       &h->snap_n, &h->sel_idx, &h->err_flag, &h->hs_pex};
  for (auto* q : ib) q->free();
  DevBuf<double>* db[] = {&h->o_ax, &h->o_ay, &h->o_bx, &h->o_by, &h->o_rad, &h->o_vx, &h->o_vy, &h->o_cx, &h->o_cy, &h->o_brad, &h->o_pvx,
                          &h->o_pvy, &h->o_list, &h->viax, &h->viay, &h->x, &h->y, &h->th, &h->dt, &h->vs, &h->vg, &h->chi2, &h->cost,
                          &h->lambda, &h->Hbackup, &h->Hband, &h->ob_x, &h->ob_y, &h->ob_th, &h->ob_dt, &h->snap_x, &h->snap_y, &h->snap_th, &h->snap_dt, &h->dbg_H,
                          &h->dbg_b, &h->dbg_chi2, &h->sel_cost, &h->stage_x, &h->stage_y, &h->stage_yaw, &h->out_cmd, &h->out_prof, &h->out_traj, &h->hsig, &h->hs_pre, &h->hs_pim};
  for (auto* q : db) q->free();
  DevBuf<double>* gb[] = {&h->g_vx, &h->g_vy, &h->cand_x, &h->cand_y, &h->cand_th, &h->cand_dt, &h->cand_sig, &h->cand_px, &h->cand_py,
                          &h->tmp_x, &h->tmp_y, &h->tmp_th, &h->tmp_dt, &h->tmp_vs, &h->tmp_vg, &h->tmp_chi2, &h->tmp_cost, &h->tmp_lambda};
  for (auto* q : gb) q->free();
  DevBuf<int>* gi[] = {&h->cand_n, &h->cand_off, &h->cand_map, &h->tmp_n, &h->tmp_i};
  for (auto* q : gi) q->free();
  h->ob_n.free();
  h->iter_log.free();
  h->phase_log.free();
  h->mcu_ctl.free(); h->mcu_pub.free(); h->mcu_items.free(); h->mcu_spec.free();
  if (h->mcu_trace) (void)hipHostFree(h->mcu_trace);
  if (h->pack_host) (void)hipHostFree(h->pack_host);
  h->pack_dev.free();
  h->g_adj.free();
  h->cm_cells.free(); h->cm_fp.free(); h->cm_out.free();
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

namespace {
// (Re)derives everything that depends on BOTH the obstacle table held in h->hob and the configuration: the lists AddEdgesObstacles /
// AddEdgesDynamicObstacles visit, the distance path (point-like LDS cache or generic) and the layout that goes with it.
int commit_obstacles(teb_amd_handle* h) {
  const auto& o = h->hob;
  const int M = (int)o.type.size();
  std::vector<int> st, dy;
  for (int i = 0; i < M; ++i) {
    // AddEdgesObstacles skips dynamic obstacles iff include_dynamic_obstacles (optimal_planner.cpp:496-497);
    // AddEdgesDynamicObstacles visits the dynamic ones (:658-659)
    if (h->cfg.include_dynamic_obstacles && o.dyn[i]) dy.push_back(i); else st.push_back(i);
  }
  auto up_i = [&](DevBuf<int>& d, const std::vector<int>& v) { return v.empty() ? hipSuccess : hipMemcpyAsync(d.p, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice, h->stream); };
  HIPCHK(up_i(h->o_static, st)); HIPCHK(up_i(h->o_dynidx, dy));
  {   // the obstacles in cache order (static list, then dynamic list): what the kernel stages into LDS, readable with scalar loads
    const size_t mo = (size_t)(h->max_obst > 0 ? h->max_obst : 1);
    std::vector<double> lo(5 * mo, 0.0);
    size_t k = 0;
    h->static_radius_zero = 1;
    for (int oi : st) if (o.type[oi] == TEB_AMD_OBST_CIRCULAR && o.rad[oi] != 0.0) h->static_radius_zero = 0;
    for (const std::vector<int>* lst : {&st, &dy})
      for (int oi : *lst) {
        lo[k] = o.ax[oi]; lo[mo + k] = o.ay[oi]; lo[2 * mo + k] = o.type[oi] == TEB_AMD_OBST_CIRCULAR ? o.rad[oi] : 0.0;
        lo[3 * mo + k] = o.vx[oi]; lo[4 * mo + k] = o.vy[oi];
        ++k;
      }
    HIPCHK(hipMemcpyAsync(h->o_list.p, lo.data(), lo.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  HIPCHK(hipStreamSynchronize(h->stream));   // host vectors go out of scope
  h->n_static = (int)st.size(); h->n_dyn = (int)dy.size();
  h->host_static = st;
  h->hsig_mode = 0;   // signatures depend on the obstacle table and on include_dynamic_obstacles
  // point-like fast path: all obstacles Point/Circular, footprint Point/Circular, and the cache fits the LDS
  bool pointlike = (h->cfg.footprint_type == TEB_AMD_FOOTPRINT_POINT || h->cfg.footprint_type == TEB_AMD_FOOTPRINT_CIRCULAR);
  for (int i = 0; i < M && pointlike; ++i) pointlike = (o.type[i] == TEB_AMD_OBST_POINT || o.type[i] == TEB_AMD_OBST_CIRCULAR);
  if (h->opt.generic_distance_path) pointlike = false;
  // a point-like scene whose obstacle cache does not fit beside the LDS band: the band moves to HBM and the cache stays (measured,
  // 64 bands x 337 poses x 500 obstacles: 9.0 instead of 11.8 ms per step)
  h->solver = h->solver_created;
  if (pointlike && M > 0 && h->solver == SOLVER_BAND && h->opt.layout == TEB_AMD_LAYOUT_AUTO &&
      (size_t)make_lds_plan(h->stride, SOLVER_BAND, M, h->lds_limit).total_bytes > h->lds_limit &&
      (size_t)make_lds_plan(h->stride, SOLVER_BANDG, M, h->lds_limit).total_bytes <= h->lds_limit) {
    if (h->hband_stride == 0) {
      h->hband_stride = ((size_t)hbo(4 * h->stride) + 2 + 1) & ~(size_t)1;
      h->Hband.free();
      HIPCHK(h->Hband.alloc((size_t)h->max_tebs * h->hband_stride));
    }
    h->solver = SOLVER_BANDG;
  }
  LdsPlan with_cache = make_lds_plan(h->stride, h->solver, M, h->lds_limit);
  if (pointlike && M > 0 && (size_t)with_cache.total_bytes <= h->lds_limit) { h->fast_points = 1; h->plan = with_cache; }
  else { h->fast_points = 0; h->plan = make_lds_plan(h->stride, h->solver, 0, h->lds_limit); }
  return TEB_AMD_OK;
}
}  // namespace

int teb_amd_set_config(teb_amd_handle_t* h, const teb_amd_config_t* cfg) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!cfg) return fail(TEB_AMD_ERR_INVALID_ARG, "null config");
  rc = validate_config(cfg);
  if (rc) return rc;   // the handle keeps its previous configuration
  const bool lists_changed = (cfg->include_dynamic_obstacles != h->cfg.include_dynamic_obstacles) ||
                             (cfg->footprint_type != h->cfg.footprint_type);
  const teb_amd_config_t previous = h->cfg;
  h->cfg = *cfg;
  if (lists_changed && h->M > 0) {
    rc = commit_obstacles(h);   // derives the static / dynamic lists, the distance path and the LDS plan from h->cfg
    if (rc) {                   // "keeps its previous configuration" (teb_amd.h): put it back together with what is derived from it
      const std::string why = g_last_error;
      h->cfg = previous;
      (void)commit_obstacles(h);
      return fail(rc, why);
    }
  }
  return TEB_AMD_OK;
}

int teb_amd_set_obstacles(teb_amd_handle_t* h, const teb_amd_obstacles_t* o) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!o || o->count < 0) return fail(TEB_AMD_ERR_INVALID_ARG, "null obstacle table");
  const int M = o->count;
  if (M > h->max_obst) return fail(TEB_AMD_ERR_CAPACITY, "more obstacles than max_obstacles");
  if (M > 0 && (!o->type || !o->ax || !o->ay)) return fail(TEB_AMD_ERR_INVALID_ARG, "obstacle arrays missing");
  teb_amd_handle::HostObst t;
  t.type.resize(M); t.dyn.resize(M); t.voff.assign(M + 1, 0);
  t.ax.resize(M); t.ay.resize(M); t.bx.resize(M); t.by.resize(M); t.rad.resize(M); t.vx.resize(M); t.vy.resize(M); t.cx.resize(M); t.cy.resize(M); t.brad.assign(M, 0.0);
  for (int i = 0; i < M; ++i) {
    t.type[i] = o->type[i];
    t.ax[i] = o->ax[i]; t.ay[i] = o->ay[i];
    t.bx[i] = o->bx ? o->bx[i] : 0; t.by[i] = o->by ? o->by[i] : 0;
    t.rad[i] = o->radius ? o->radius[i] : 0;
    t.vx[i] = o->vx ? o->vx[i] : 0; t.vy[i] = o->vy ? o->vy[i] : 0;
    t.dyn[i] = o->dynamic ? (o->dynamic[i] != 0) : 0;
    t.voff[i] = (int)t.pvx.size();
    switch (t.type[i]) {
      case TEB_AMD_OBST_POINT: case TEB_AMD_OBST_CIRCULAR: t.cx[i] = t.ax[i]; t.cy[i] = t.ay[i]; t.brad[i] = std::fabs(t.rad[i]); break;
      case TEB_AMD_OBST_LINE: case TEB_AMD_OBST_PILL:
        t.cx[i] = 0.5 * (t.ax[i] + t.bx[i]); t.cy[i] = 0.5 * (t.ay[i] + t.by[i]);
        t.brad[i] = 0.5 * std::hypot(t.bx[i] - t.ax[i], t.by[i] - t.ay[i]) + std::fabs(t.rad[i]);
        break;
      case TEB_AMD_OBST_POLYGON: {
        if (!o->vert_offset || !o->vert_x || !o->vert_y) return fail(TEB_AMD_ERR_INVALID_ARG, "polygon obstacle without vertex arrays");
        int k0 = o->vert_offset[i], k1 = o->vert_offset[i + 1];
        if (k1 <= k0) return fail(TEB_AMD_ERR_INVALID_ARG, "polygon obstacle without vertices");
        for (int k = k0; k < k1; ++k) { t.pvx.push_back(o->vert_x[k]); t.pvy.push_back(o->vert_y[k]); }
        polygon_centroid(o->vert_x + k0, o->vert_y + k0, k1 - k0, t.cx[i], t.cy[i]);
        for (int k = k0; k < k1; ++k) t.brad[i] = std::max(t.brad[i], std::hypot(o->vert_x[k] - t.cx[i], o->vert_y[k] - t.cy[i]));
        if (!(t.brad[i] == t.brad[i])) t.brad[i] = std::numeric_limits<double>::infinity();   // NaN centroid: never culled
        break;
      }
      default: return fail(TEB_AMD_ERR_INVALID_ARG, "unknown obstacle type");
    }
  }
  t.voff[M] = (int)t.pvx.size();
  if ((int)t.pvx.size() > h->max_verts) return fail(TEB_AMD_ERR_CAPACITY, "more polygon vertices than max_obstacle_vertices");
  auto up_i = [&](DevBuf<int>& d, const std::vector<int>& v) { return v.empty() ? hipSuccess : hipMemcpyAsync(d.p, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice, h->stream); };
  auto up_d = [&](DevBuf<double>& d, const std::vector<double>& v) { return v.empty() ? hipSuccess : hipMemcpyAsync(d.p, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice, h->stream); };
  HIPCHK(up_i(h->o_type, t.type)); HIPCHK(up_i(h->o_dyn, t.dyn)); HIPCHK(up_i(h->o_voff, t.voff));
  HIPCHK(up_d(h->o_ax, t.ax)); HIPCHK(up_d(h->o_ay, t.ay)); HIPCHK(up_d(h->o_bx, t.bx)); HIPCHK(up_d(h->o_by, t.by)); HIPCHK(up_d(h->o_rad, t.rad));
  HIPCHK(up_d(h->o_vx, t.vx)); HIPCHK(up_d(h->o_vy, t.vy)); HIPCHK(up_d(h->o_cx, t.cx)); HIPCHK(up_d(h->o_cy, t.cy)); HIPCHK(up_d(h->o_brad, t.brad));
  HIPCHK(up_d(h->o_pvx, t.pvx)); HIPCHK(up_d(h->o_pvy, t.pvy));
  HIPCHK(hipStreamSynchronize(h->stream));   // the uploads read `t`
  h->M = M;
  h->host_type = t.type;
  h->host_cx = t.cx; h->host_cy = t.cy;
  h->hs_prod_valid = false;
  h->hob = std::move(t);
  return commit_obstacles(h);
}

int teb_amd_set_via_points(teb_amd_handle_t* h, int32_t count, const double* x, const double* y) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (count < 0 || (count > 0 && (!x || !y))) return fail(TEB_AMD_ERR_INVALID_ARG, "bad via-point arrays");
  if (count > h->max_via) return fail(TEB_AMD_ERR_CAPACITY, "more via-points than max_via_points");
  if (count > 0) {
    HIPCHK(hipMemcpyAsync(h->viax.p, x, count * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->viay.p, y, count * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  h->nvia = count;
  return TEB_AMD_OK;
}

int teb_amd_upload_tebs(teb_amd_handle_t* h, const teb_amd_teb_batch_t* bt) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!bt || bt->count <= 0 || !bt->n || !bt->x || !bt->y || !bt->theta || !bt->dt)
    return fail(TEB_AMD_ERR_INVALID_ARG, "bad TEB batch");
  if (bt->count > h->max_tebs) return fail(TEB_AMD_ERR_CAPACITY, "more TEBs than max_tebs");
  const int B = bt->count;
  int nmax = 0;
  for (int b = 0; b < B; ++b) {
    if (bt->n[b] < 2) return fail(TEB_AMD_ERR_INVALID_ARG, "a TEB needs at least 2 poses");
    if (bt->n[b] > bt->stride) return fail(TEB_AMD_ERR_INVALID_ARG, "n[b] > stride");
    nmax = bt->n[b] > nmax ? bt->n[b] : nmax;
  }
  if (nmax > h->stride) return fail(TEB_AMD_ERR_CAPACITY, "a TEB has more poses than max_poses");
  const int wc_pack = bt->stride < h->stride ? bt->stride : h->stride;   // whole rows, like the strided copies: the padding stays the caller's
  if (h->pack_host && 11 * (size_t)B + 4 * (size_t)B * wc_pack <= kPackMaxDoubles) {   // small batch: one packed message, one copy, one kernel
    double* m = h->pack_host;
    const int wc = wc_pack;
    for (int b = 0; b < B; ++b) {
      m[b] = bt->n[b];
      m[B + b] = bt->has_vel_start ? (bt->has_vel_start[b] != 0) : 1;
      m[2 * B + b] = bt->has_vel_goal ? (bt->has_vel_goal[b] != 0) : 1;
      m[3 * B + b] = bt->prefer_rotdir ? bt->prefer_rotdir[b] : TEB_AMD_ROT_NONE;
      m[4 * B + b] = bt->via_points_enabled ? (bt->via_points_enabled[b] != 0) : 1;
      for (int q = 0; q < 3; ++q) {
        m[5 * B + 3 * b + q] = bt->vel_start ? bt->vel_start[3 * b + q] : 0.0;
        m[8 * B + 3 * b + q] = bt->vel_goal ? bt->vel_goal[3 * b + q] : 0.0;
      }
    }
    double* body = m + 11 * (size_t)B;
    const size_t plane = (size_t)B * wc;
    for (int b = 0; b < B; ++b) {
      const size_t so = (size_t)b * bt->stride, q = (size_t)b * wc;
      std::memcpy(body + q, bt->x + so, wc * sizeof(double)); std::memcpy(body + plane + q, bt->y + so, wc * sizeof(double));
      std::memcpy(body + 2 * plane + q, bt->theta + so, wc * sizeof(double)); std::memcpy(body + 3 * plane + q, bt->dt + so, wc * sizeof(double));
    }
    const size_t count = 11 * (size_t)B + 4 * plane;
    HIPCHK(hipMemcpyAsync(h->pack_dev.p, m, count * sizeof(double), hipMemcpyHostToDevice, h->stream));
    const int grid = (int)std::min<size_t>(256, (std::max<size_t>(plane, 3 * (size_t)B) + 255) / 256);
    hipLaunchKernelGGL(unpack_bands_kernel, dim3(grid), dim3(256), 0, h->stream, h->pack_dev.p, B, wc, h->stride, h->n.p, h->has_vs.p, h->has_vg.p, h->rotdir.p,
                       h->via_en.p, h->vs.p, h->vg.p, h->x.p, h->y.p, h->th.p, h->dt.p, h->optimized.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));   // (the pinned buffer is reused by the next call)
    h->B = B;
    h->consumers_valid = false; h->nmax_known = nmax;
    h->hsig_mode = 0;
    return TEB_AMD_OK;
  }
  const size_t w = (size_t)(bt->stride < h->stride ? bt->stride : h->stride) * sizeof(double);
  const size_t sp = (size_t)bt->stride * sizeof(double), dp = (size_t)h->stride * sizeof(double);
  HIPCHK(hipMemcpy2DAsync(h->x.p, dp, bt->x, sp, w, B, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpy2DAsync(h->y.p, dp, bt->y, sp, w, B, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpy2DAsync(h->th.p, dp, bt->theta, sp, w, B, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpy2DAsync(h->dt.p, dp, bt->dt, sp, w, B, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->n.p, bt->n, B * sizeof(int), hipMemcpyHostToDevice, h->stream));
  std::vector<int> hvs(B, 1), hvg(B, 1), rd(B, TEB_AMD_ROT_NONE), ve(B, 1);
  std::vector<double> vs(3 * (size_t)B, 0.0), vg(3 * (size_t)B, 0.0);
  for (int b = 0; b < B; ++b) {
    if (bt->has_vel_start) hvs[b] = bt->has_vel_start[b] != 0;
    if (bt->has_vel_goal) hvg[b] = bt->has_vel_goal[b] != 0;
    if (bt->prefer_rotdir) rd[b] = bt->prefer_rotdir[b];
    if (bt->via_points_enabled) ve[b] = bt->via_points_enabled[b] != 0;
    for (int q = 0; q < 3; ++q) {
      if (bt->vel_start) vs[3 * b + q] = bt->vel_start[3 * b + q];
      if (bt->vel_goal) vg[3 * b + q] = bt->vel_goal[3 * b + q];
    }
  }
  HIPCHK(hipMemcpyAsync(h->has_vs.p, hvs.data(), B * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->has_vg.p, hvg.data(), B * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->rotdir.p, rd.data(), B * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->via_en.p, ve.data(), B * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->vs.p, vs.data(), 3 * (size_t)B * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->vg.p, vg.data(), 3 * (size_t)B * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemsetAsync(h->optimized.p, 0, B * sizeof(int), h->stream));   // bands from the host: optimized_ unknown -> false
  HIPCHK(hipStreamSynchronize(h->stream));
  h->B = B;
  h->consumers_valid = false; h->nmax_known = nmax;
  h->hsig_mode = 0;   // the bands change: signatures of an earlier teb_amd_compute_h_signatures call are stale
  return TEB_AMD_OK;
}

int teb_amd_download_tebs(teb_amd_handle_t* h, teb_amd_teb_batch_t* bt) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!bt || !bt->n || !bt->x || !bt->y || !bt->theta || !bt->dt) return fail(TEB_AMD_ERR_INVALID_ARG, "bad TEB batch");
  if (bt->count < h->B) return fail(TEB_AMD_ERR_INVALID_ARG, "batch too small for the resident TEBs");
  const int B = h->B;
  if (B == 0) return TEB_AMD_OK;   // an empty handle (before the first upload, after every band was dropped, a rank without candidates): nothing to copy
  {
    const int wc = bt->stride < h->stride ? bt->stride : h->stride;
    const size_t plane = (size_t)B * wc, count = (size_t)B + 4 * plane;
    if (h->pack_host && count <= kPackMaxDoubles) {   // small batch: one gather kernel, one copy
      const int grid = (int)std::min<size_t>(256, (plane + 255) / 256);
      hipLaunchKernelGGL(pack_bands_kernel, dim3(grid), dim3(256), 0, h->stream, h->pack_dev.p, B, wc, h->stride, h->n.p, h->x.p, h->y.p, h->th.p, h->dt.p);
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(h->pack_host, h->pack_dev.p, count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      const double* m = h->pack_host;
      for (int b = 0; b < B; ++b) if ((int)m[b] > bt->stride) return fail(TEB_AMD_ERR_CAPACITY, "host batch stride too small for the resized TEB");
      const double* body = m + B;
      for (int b = 0; b < B; ++b) {
        const size_t so = (size_t)b * bt->stride, q = (size_t)b * wc;
        std::memcpy(bt->x + so, body + q, wc * sizeof(double)); std::memcpy(bt->y + so, body + plane + q, wc * sizeof(double));
        std::memcpy(bt->theta + so, body + 2 * plane + q, wc * sizeof(double)); std::memcpy(bt->dt + so, body + 3 * plane + q, wc * sizeof(double));
        bt->n[b] = (int)m[b];
      }
      return TEB_AMD_OK;
    }
  }
  std::vector<int> n(B);
  HIPCHK(hipMemcpyAsync(n.data(), h->n.p, B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int b = 0; b < B; ++b) if (n[b] > bt->stride) return fail(TEB_AMD_ERR_CAPACITY, "host batch stride too small for the resized TEB");
  const size_t w = (size_t)(bt->stride < h->stride ? bt->stride : h->stride) * sizeof(double);
  const size_t hp = (size_t)bt->stride * sizeof(double), dp = (size_t)h->stride * sizeof(double);
  HIPCHK(hipMemcpy2DAsync(bt->x, hp, h->x.p, dp, w, B, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpy2DAsync(bt->y, hp, h->y.p, dp, w, B, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpy2DAsync(bt->theta, hp, h->th.p, dp, w, B, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpy2DAsync(bt->dt, hp, h->dt.p, dp, w, B, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int b = 0; b < B; ++b) bt->n[b] = n[b];
  return TEB_AMD_OK;
}

int teb_amd_optimize_batch(teb_amd_handle_t* h, int32_t inner, int32_t outer, int32_t compute_cost, double obst_cost_scale,
                           double viapoint_cost_scale, int32_t alternative_time_cost) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (inner < 0 || outer < 0) return fail(TEB_AMD_ERR_INVALID_ARG, "negative iteration count");
  OptArgs a;
  std::memset(&a, 0, sizeof a);
  a.inner = inner; a.outer = outer; a.compute_cost = compute_cost; a.no_near_cache = h->opt.no_near_cache != 0; a.band_ldlt = h->solver == SOLVER_BANDG ? 0 : h->band_ldlt; a.Hband = h->Hband.p; a.hband_stride = h->hband_stride;
  a.obst_scale = obst_cost_scale; a.via_scale = viapoint_cost_scale; a.alt_time = alternative_time_cost;
  if (h->iter_log_on) { a.iter_log = h->iter_log.p; a.iter_log_cap = TEB_AMD_ITERATION_LOG_ROWS; }
  if (h->phase_log_on) a.phase_log = h->phase_log.p;
#ifdef TEB_PROFILE
  a.dbg_H = h->dbg_H.p;
#endif
  h->last_inner = inner;
  return launch(h, a);
}

// g2o's per-iteration console line ("iteration= i chi2= .. lambda= .. levenbergIter= ..", printed when optimization_verbose sets
// SparseOptimizer::setVerbose, src/optimal_planner.cpp:384) as data: one row per LM iteration of the last teb_amd_optimize_batch.
int teb_amd_set_iteration_log(teb_amd_handle_t* h, int32_t enable) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (enable && !h->iter_log.p) {
    const size_t count = (size_t)h->max_tebs * TEB_AMD_ITERATION_LOG_ROWS * 4;
    HIPCHK(h->iter_log.alloc(count));
    HIPCHK(hipMemsetAsync(h->iter_log.p, 0, count * sizeof(double), h->stream));
  }
  h->iter_log_on = enable != 0;
  return TEB_AMD_OK;
}

// Where a launch spends its time, measured on the product kernel itself: shader cycles per phase of every band's workgroup (lane 0,
// s_memtime at the phase boundaries of the outer / LM loop; kPhaseLogSlots values per band, teb_amd_debug.h names them).
static_assert(kPhaseLogSlots == TEB_AMD_PHASE_LOG_SLOTS, "teb_amd_debug.h and teb_device.hpp disagree");
int teb_amd_set_phase_log(teb_amd_handle_t* h, int32_t enable) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (enable && !h->phase_log.p) {
    const size_t count = (size_t)h->max_tebs * kPhaseLogSlots;
    HIPCHK(h->phase_log.alloc(count));
    HIPCHK(hipMemsetAsync(h->phase_log.p, 0, count * sizeof(double), h->stream));
  }
  h->phase_log_on = enable != 0;
  return TEB_AMD_OK;
}
int teb_amd_get_phase_log(teb_amd_handle_t* h, double* cycles, int32_t capacity_bands, int32_t* bands) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!h->phase_log_on || !h->phase_log.p) return fail(TEB_AMD_ERR_INVALID_ARG, "the phase log is off (teb_amd_set_phase_log)");
  if (!cycles || capacity_bands < 0) return fail(TEB_AMD_ERR_INVALID_ARG, "null argument");
  const int nb = std::min((int)capacity_bands, h->B);
  if (nb > 0) HIPCHK(hipMemcpyAsync(cycles, h->phase_log.p, (size_t)nb * kPhaseLogSlots * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (bands) *bands = nb;
  return TEB_AMD_OK;
}

int teb_amd_get_iteration_log(teb_amd_handle_t* h, int32_t b, double* rows, int32_t capacity_rows, int32_t* n_rows) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!h->iter_log_on || !h->iter_log.p) return fail(TEB_AMD_ERR_INVALID_ARG, "the iteration log is off (teb_amd_set_iteration_log)");
  if (b < 0 || b >= h->B || !rows || !n_rows || capacity_rows < 0) return fail(TEB_AMD_ERR_INVALID_ARG, "teb_amd_get_iteration_log: bad arguments");
  int iters = 0;
  HIPCHK(hipMemcpyAsync(&iters, h->iters.p + b, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  int nr = std::min(std::min(iters, (int)TEB_AMD_ITERATION_LOG_ROWS), (int)capacity_rows);
  if (nr < 0) nr = 0;
  if (nr > 0) {
    HIPCHK(hipMemcpyAsync(rows, h->iter_log.p + (size_t)b * TEB_AMD_ITERATION_LOG_ROWS * 4, (size_t)nr * 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  *n_rows = nr;
  return TEB_AMD_OK;
}

int teb_amd_synchronize(teb_amd_handle_t* h) {
  int rc = check_handle(h);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return TEB_AMD_OK;
}

int teb_amd_get_results(teb_amd_handle_t* h, teb_amd_results_t* out) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!out) return fail(TEB_AMD_ERR_INVALID_ARG, "null results");
  const int B = h->B;
  if (B == 0) return TEB_AMD_OK;   // an empty handle has no results to copy (a grid of 0 workgroups would be a launch error)
  if (h->pack_host && 6 * (size_t)B <= kPackMaxDoubles) {   // six result columns in one gather + one copy
    hipLaunchKernelGGL(pack_results_kernel, dim3((B + 255) / 256), dim3(256), 0, h->stream, h->pack_dev.p, B, h->status.p, h->iters.p, h->trials.p, h->chi2.p, h->cost.p,
                       h->lambda.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h->pack_host, h->pack_dev.p, 6 * (size_t)B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const double* m = h->pack_host;
    for (int b = 0; b < B; ++b) {
      if (out->status) out->status[b] = (int32_t)m[b];
      if (out->lm_iterations) out->lm_iterations[b] = (int32_t)m[B + b];
      if (out->lm_trials) out->lm_trials[b] = (int32_t)m[2 * B + b];
      if (out->chi2) out->chi2[b] = m[3 * B + b];
      if (out->cost) out->cost[b] = m[4 * B + b];
      if (out->lambda) out->lambda[b] = m[5 * B + b];
    }
    return TEB_AMD_OK;
  }
  if (out->status) HIPCHK(hipMemcpyAsync(out->status, h->status.p, B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  if (out->lm_iterations) HIPCHK(hipMemcpyAsync(out->lm_iterations, h->iters.p, B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  if (out->lm_trials) HIPCHK(hipMemcpyAsync(out->lm_trials, h->trials.p, B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  if (out->chi2) HIPCHK(hipMemcpyAsync(out->chi2, h->chi2.p, B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (out->cost) HIPCHK(hipMemcpyAsync(out->cost, h->cost.p, B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (out->lambda) HIPCHK(hipMemcpyAsync(out->lambda, h->lambda.p, B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return TEB_AMD_OK;
}

int teb_amd_select_best(teb_amd_handle_t* h, int32_t last_best, int32_t initial_plan, int32_t* best, double* best_cost) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!best) return fail(TEB_AMD_ERR_INVALID_ARG, "null output");
  if (h->B <= 0) return fail(TEB_AMD_ERR_INVALID_ARG, "no TEBs uploaded");
  if (last_best >= h->B) last_best = -1;
  if (initial_plan >= h->B) initial_plan = -1;
  hipLaunchKernelGGL(select_best_kernel, dim3(1), dim3(kThreads), 0, h->stream, h->cost.p, h->B, last_best, initial_plan,
                     h->cfg.selection_cost_hysteresis, h->cfg.selection_prefer_initial_plan, h->sel_cost.p, h->sel_idx.p);
  HIPCHK(hipGetLastError());
  double rec[2];   // one 16-byte copy
  HIPCHK(hipMemcpyAsync(rec, h->sel_cost.p, sizeof(rec), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  *best = (int32_t)rec[1];
  if (best_cost) *best_cost = rec[0];
  return TEB_AMD_OK;
}

// ---- multi-GPU: the selection exchange (teb_comm.hpp) --------------------------------------------------------------------------------
#define NCCLCHK(expr)                                                                                                     \
  do {                                                                                                                    \
    int _r = (expr);                                                                                                      \
    if (_r != kRcclSuccess) return fail(TEB_AMD_ERR_HIP, std::string(#expr) + " -> " + rccl().GetErrorString(_r));        \
  } while (0)

int teb_amd_comm_unique_id(char id[TEB_AMD_COMM_ID_BYTES]) {
  static_assert(sizeof(rccl_unique_id_t) == TEB_AMD_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  if (!id) return fail(TEB_AMD_ERR_INVALID_ARG, "null id buffer");
  if (!rccl().load()) return fail(TEB_AMD_ERR_UNSUPPORTED, rccl().error);
  rccl_unique_id_t u;
  NCCLCHK(rccl().GetUniqueId(&u));
  std::memcpy(id, &u, sizeof(u));
  return TEB_AMD_OK;
}

int teb_amd_comm_create(const char id[TEB_AMD_COMM_ID_BYTES], int32_t rank, int32_t world, int32_t device, teb_amd_comm_t** out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) return fail(TEB_AMD_ERR_INVALID_ARG, "teb_amd_comm_create: bad arguments");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(TEB_AMD_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(TEB_AMD_ERR_INVALID_ARG, "device ordinal out of range");
  if (!rccl().load()) return fail(TEB_AMD_ERR_UNSUPPORTED, rccl().error);
  HIPCHK(hipSetDevice(device));
  teb_amd_comm* c = new teb_amd_comm();
  c->rank = rank; c->world = world; c->device = device;
  rccl_unique_id_t u;
  std::memcpy(&u, id, sizeof(u));
  int r = rccl().CommInitRank(&c->comm, world, u, rank);
  if (r != kRcclSuccess) { delete c; return fail(TEB_AMD_ERR_HIP, std::string("ncclCommInitRank -> ") + rccl().GetErrorString(r)); }
  if (hipMalloc(reinterpret_cast<void**>(&c->rec), 2 * sizeof(double)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&c->all), 2 * (size_t)world * sizeof(double)) != hipSuccess) {
    teb_amd_comm_destroy(c);
    return fail(TEB_AMD_ERR_HIP, "teb_amd_comm_create: device allocation failed");
  }
  *out = c;
  return TEB_AMD_OK;
}

void teb_amd_comm_destroy(teb_amd_comm_t* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->comm && rccl().lib) (void)rccl().CommDestroy(c->comm);
  if (c->rec) (void)hipFree(c->rec);
  if (c->all) (void)hipFree(c->all);
  if (c->msg) (void)hipFree(c->msg);
  delete c;
}

// A collective like teb_amd_broadcast_band: whatever is wrong on THIS rank (bad handle, a handle on another device than the communicator's,
// a failed launch) must not keep it out of the all-gather - the peers would wait for ever. Such a rank contributes the
// unusable record (DBL_MAX, -1), which no other rank can pick, and returns its own error AFTER the collective; the peers get the best
// band of the ranks that had one. Only a rank without a communicator cannot take part at all.
int teb_amd_select_best_distributed(teb_amd_handle_t* h, teb_amd_comm_t* c, int32_t global_offset, int32_t last_best_global,
                                    int32_t initial_plan_global, int32_t* best_global, double* best_cost, int32_t* owner_rank) {
  if (!c || !c->comm) return fail(TEB_AMD_ERR_INVALID_ARG, "teb_amd_select_best_distributed: null communicator");
  int local = TEB_AMD_OK;
  std::string why;
  if (check_handle(h)) { local = TEB_AMD_ERR_INVALID_ARG; why = "bad handle"; }
  else if (!best_global) { local = TEB_AMD_ERR_INVALID_ARG; why = "null output"; }
  else if (c->device != h->device) { local = TEB_AMD_ERR_INVALID_ARG; why = "communicator and handle live on different devices"; }
  (void)hipSetDevice(c->device);
  hipStream_t stream = local == TEB_AMD_OK ? h->stream : nullptr;   // (a rank with an unusable handle takes part on the null stream)
  if (local == TEB_AMD_OK) {
    const int have = h->B > 0;
    if (have) {   // local arg-min with the multipliers applied where this rank owns the favoured candidates
      int lb = last_best_global - global_offset, ip = initial_plan_global - global_offset;
      if (last_best_global < 0 || lb < 0 || lb >= h->B) lb = -1;
      if (initial_plan_global < 0 || ip < 0 || ip >= h->B) ip = -1;
      hipLaunchKernelGGL(select_best_kernel, dim3(1), dim3(kThreads), 0, stream, h->cost.p, h->B, lb, ip,
                         h->cfg.selection_cost_hysteresis, h->cfg.selection_prefer_initial_plan, h->sel_cost.p, h->sel_idx.p);
    }
    hipLaunchKernelGGL(pack_record_kernel, dim3(1), dim3(64), 0, stream, h->sel_cost.p, h->sel_idx.p, global_offset, have, c->rec);
    if (hipGetLastError() != hipSuccess) { local = TEB_AMD_ERR_HIP; why = "local selection kernels failed to launch"; }
  }
  if (local != TEB_AMD_OK) {
    static const double none[2] = {1.7976931348623157e308, -1.0};   // (static: the copy is asynchronous)
    (void)hipMemcpyAsync(c->rec, none, sizeof(none), hipMemcpyHostToDevice, stream);
  }
  NCCLCHK(rccl().AllGather(c->rec, c->all, 2, kRcclFloat64, c->comm, stream));
  std::vector<double> all(2 * (size_t)c->world);
  HIPCHK(hipMemcpyAsync(all.data(), c->all, all.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  double bc; int bi, owner;
  world_argmin(all.data(), c->world, &bi, &bc, &owner);
  // the peers' choice is reported even to the rank that failed (it has to follow them into teb_amd_broadcast_band, a collective too)
  if (best_global) *best_global = bi;
  if (best_cost) *best_cost = bc;
  if (owner_rank) *owner_rank = owner;
  if (local != TEB_AMD_OK) return fail(local, "teb_amd_select_best_distributed: " + why + " (this rank sent an unusable record; its peers were not held up)");
  return TEB_AMD_OK;
}

// A collective: no rank may leave before the peers are released. Everything that can go wrong on ONE rank (an index out of range on the
// owner, an allocation failure, capacities that differ between the ranks) is therefore found out by all ranks together - an all-gather of
// (status, capacity) per rank - BEFORE the broadcast, and every rank returns the same verdict. Only a rank without a communicator cannot
// take part at all (TEB_AMD_ERR_INVALID_ARG there; its peers then wait in the all-gather like in any collective a rank never enters).
int teb_amd_broadcast_band(teb_amd_handle_t* h, teb_amd_comm_t* c, int32_t owner_rank, int32_t local_index, int32_t capacity, int32_t* n,
                           double* x, double* y, double* theta, double* dt) {
  if (!c || !c->comm) return fail(TEB_AMD_ERR_INVALID_ARG, "teb_amd_broadcast_band: null communicator");
  int local = TEB_AMD_OK;
  std::string why;
  if (check_handle(h)) { local = TEB_AMD_ERR_INVALID_ARG; why = "bad handle"; }
  else if (!n || !x || !y || !theta || !dt || capacity < 2) { local = TEB_AMD_ERR_INVALID_ARG; why = "null output / capacity < 2"; }
  else if (owner_rank < 0 || owner_rank >= c->world) { local = TEB_AMD_ERR_INVALID_ARG; why = "owner_rank out of range"; }
  else if (c->rank == owner_rank && (local_index < 0 || local_index >= h->B)) { local = TEB_AMD_ERR_INVALID_ARG; why = "local_index out of range on the owner"; }
  (void)hipSetDevice(c->device);
  hipStream_t stream = h ? h->stream : nullptr;   // (a rank with a bad handle still takes part, on the null stream)
  const size_t count = capacity >= 2 ? 3 + 4 * (size_t)capacity : 0;   // n, four strips, the band's statistics
  if (local == TEB_AMD_OK && c->msg_cap < count) {
    if (c->msg) (void)hipFree(c->msg);
    c->msg = nullptr; c->msg_cap = 0;
    if (hipMalloc(reinterpret_cast<void**>(&c->msg), count * sizeof(double)) != hipSuccess) { local = TEB_AMD_ERR_HIP; why = "message buffer allocation failed"; }
    else c->msg_cap = count;
  }
  // round 1: (status, capacity) of every rank
  const double mine[2] = {(double)local, (double)capacity};
  std::vector<double> all(2 * (size_t)c->world);
  if (hipMemcpyAsync(c->rec, mine, sizeof(mine), hipMemcpyHostToDevice, stream) != hipSuccess) return fail(TEB_AMD_ERR_HIP, "teb_amd_broadcast_band: staging copy failed");
  NCCLCHK(rccl().AllGather(c->rec, c->all, 2, kRcclFloat64, c->comm, stream));
  HIPCHK(hipMemcpyAsync(all.data(), c->all, all.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  for (int r = 0; r < c->world; ++r) {
    if ((int)all[2 * r] != TEB_AMD_OK)
      return fail(r == c->rank ? local : TEB_AMD_ERR_INVALID_ARG,
                  r == c->rank ? "teb_amd_broadcast_band: " + why : "teb_amd_broadcast_band: rank " + std::to_string(r) + " reported an error; no band was sent");
    if ((int)all[2 * r + 1] != capacity)
      return fail(TEB_AMD_ERR_INVALID_ARG, "teb_amd_broadcast_band: `capacity` differs between the ranks (rank " + std::to_string(r) + " passed " +
                                               std::to_string((int)all[2 * r + 1]) + ", this rank " + std::to_string(capacity) + ")");
  }
  // round 2: the strip itself
  if (c->rank == owner_rank)
    hipLaunchKernelGGL(pack_band_kernel, dim3((capacity + 255) / 256), dim3(256), 0, stream, h->n.p, h->x.p, h->y.p, h->th.p, h->dt.p,
                       local_index, h->stride, capacity, c->msg, h->chi2.p, h->iters.p, h->last_iters.p, h->cfg.divergence_detection_enable != 0,
                       h->last_inner);
  const bool packed = hipGetLastError() == hipSuccess;   // (a failed pack still enters the broadcast; what arrives is then rejected below)
  NCCLCHK(rccl().Broadcast(c->msg, c->msg, count, kRcclFloat64, owner_rank, c->comm, stream));
  std::vector<double> host(count);
  HIPCHK(hipMemcpyAsync(host.data(), c->msg, count * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  if (!packed) return fail(TEB_AMD_ERR_HIP, "teb_amd_broadcast_band: packing the band failed on the owner");
  const int nb = (int)host[0];
  *n = nb;
  if (nb > capacity) return fail(TEB_AMD_ERR_CAPACITY, "the winner has more poses than `capacity`");
  std::memcpy(x, host.data() + 1, capacity * sizeof(double));
  std::memcpy(y, host.data() + 1 + capacity, capacity * sizeof(double));
  std::memcpy(theta, host.data() + 1 + 2 * (size_t)capacity, capacity * sizeof(double));
  std::memcpy(dt, host.data() + 1 + 3 * (size_t)capacity, capacity * sizeof(double));
  c->band_stats_available = host[1 + 4 * (size_t)capacity] != 0.0;
  c->band_stats_back_chi2 = host[2 + 4 * (size_t)capacity];
  return TEB_AMD_OK;
}

int teb_amd_comm_last_band_statistics(const teb_amd_comm_t* c, int32_t* available, double* back_chi2) {
  if (!c || !available || !back_chi2) return fail(TEB_AMD_ERR_INVALID_ARG, "teb_amd_comm_last_band_statistics: null argument");
  *available = c->band_stats_available; *back_chi2 = c->band_stats_back_chi2;
  return TEB_AMD_OK;
}

int teb_amd_debug_world_argmin(const double* records, int32_t world, int32_t* best_global, double* best_cost, int32_t* owner_rank) {
  if (!records || world < 1 || !best_global) return fail(TEB_AMD_ERR_INVALID_ARG, "teb_amd_debug_world_argmin: bad arguments");
  int bi, owner; double bc;
  world_argmin(records, world, &bi, &bc, &owner);
  *best_global = bi;
  if (best_cost) *best_cost = bc;
  if (owner_rank) *owner_rank = owner;
  return TEB_AMD_OK;
}

// ---- f1 / f2: producers and consumers of the device-resident strips (kernels in teb_strip.hpp) ---------------------------------
namespace {

// slots >= B join the batch with the defaults of TebOptimalPlanner::initialize() (src/optimal_planner.cpp:86-104)
int extend_batch(teb_amd_handle* h, int b) {
  if (b < 0 || b >= h->max_tebs) return fail(TEB_AMD_ERR_INVALID_ARG, "TEB slot out of range");
  for (int k = h->B; k <= b; ++k) {
    const int one = 1, none = TEB_AMD_ROT_NONE, two = 2;
    const double z3[3] = {0, 0, 0};
    HIPCHK(hipMemcpyAsync(h->has_vs.p + k, &one, sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->has_vg.p + k, &one, sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->rotdir.p + k, &none, sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->via_en.p + k, &one, sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemsetAsync(h->optimized.p + k, 0, sizeof(int), h->stream));   // a new TebOptimalPlanner: optimized_(false)
    HIPCHK(hipMemcpyAsync(h->vs.p + 3 * k, z3, sizeof(z3), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->vg.p + 3 * k, z3, sizeof(z3), hipMemcpyHostToDevice, h->stream));
    if (k < b) {   // a skipped slot becomes the trivial two-pose band at the origin so that the batch stays well-formed
      const double zz[2] = {0, 0}, dd[2] = {0.1, 0};
      const size_t o = (size_t)k * h->stride;
      HIPCHK(hipMemcpyAsync(h->x.p + o, zz, sizeof(zz), hipMemcpyHostToDevice, h->stream));
      HIPCHK(hipMemcpyAsync(h->y.p + o, zz, sizeof(zz), hipMemcpyHostToDevice, h->stream));
      HIPCHK(hipMemcpyAsync(h->th.p + o, zz, sizeof(zz), hipMemcpyHostToDevice, h->stream));
      HIPCHK(hipMemcpyAsync(h->dt.p + o, dd, sizeof(dd), hipMemcpyHostToDevice, h->stream));
      HIPCHK(hipMemcpyAsync(h->n.p + k, &two, sizeof(int), hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));   // the small host temporaries above
  }
  if (b >= h->B) h->B = b + 1;
  return TEB_AMD_OK;
}

int finish_init(teb_amd_handle* h) {
  HIPCHK(hipGetLastError());
  int err = 0;
  HIPCHK(hipMemcpyAsync(&err, h->err_flag.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->consumers_valid = false; h->nmax_known = -1;
  h->hsig_mode = 0;   // the bands change: signatures of an earlier teb_amd_compute_h_signatures call are stale
  if (err) return fail(TEB_AMD_ERR_CAPACITY, "initTrajectoryToGoal: the band needs more poses than max_poses");
  return TEB_AMD_OK;
}

int stage(teb_amd_handle* h, int n, const double* px, const double* py, const double* pyaw) {
  if (n < 1 || !px || !py) return fail(TEB_AMD_ERR_INVALID_ARG, "empty plan / path");
  if (n > h->stride + 1) return fail(TEB_AMD_ERR_CAPACITY, "plan / path longer than max_poses + 1");
  HIPCHK(hipMemcpyAsync(h->stage_x.p, px, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->stage_y.p, py, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (pyaw) HIPCHK(hipMemcpyAsync(h->stage_yaw.p, pyaw, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemsetAsync(h->err_flag.p, 0, sizeof(int), h->stream));
  return TEB_AMD_OK;
}

int run_consumers(teb_amd_handle* h, int look_ahead, int prevent) {
  if (h->B <= 0) return fail(TEB_AMD_ERR_INVALID_ARG, "no TEBs on the device");
  if (h->consumers_valid && h->consumers_la == look_ahead && h->consumers_prevent == prevent) return TEB_AMD_OK;
  hipLaunchKernelGGL(consumers_kernel, dim3(h->B), dim3(kThreads), 0, h->stream, h->cfg, batch_of(h), look_ahead, prevent,
                     h->out_cmd.p, h->out_prof.p, h->out_traj.p);
  HIPCHK(hipGetLastError());
  h->consumers_valid = true; h->consumers_la = look_ahead; h->consumers_prevent = prevent;
  return TEB_AMD_OK;
}

int band_n(teb_amd_handle* h, int b, int* n) {
  if (b < 0 || b >= h->B) return fail(TEB_AMD_ERR_INVALID_ARG, "TEB index out of range");
  HIPCHK(hipMemcpyAsync(n, h->n.p + b, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return TEB_AMD_OK;
}

}  // namespace

int teb_amd_init_trajectory_line(teb_amd_handle_t* h, int32_t b, const double* start, const double* goal, double diststep,
                                 double max_vel_x, int32_t min_samples, int32_t guess_backwards_motion) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!start || !goal) return fail(TEB_AMD_ERR_INVALID_ARG, "null start / goal");
  if ((rc = extend_batch(h, b))) return rc;
  HIPCHK(hipMemsetAsync(h->err_flag.p, 0, sizeof(int), h->stream));
  hipLaunchKernelGGL(init_line_kernel, dim3(1), dim3(kThreads), 0, h->stream, batch_of(h), b, start[0], start[1], start[2], goal[0],
                     goal[1], goal[2], diststep, max_vel_x, min_samples, guess_backwards_motion, h->err_flag.p);
  return finish_init(h);
}

int teb_amd_init_trajectory_plan(teb_amd_handle_t* h, int32_t b, int32_t n_plan, const double* plan_x, const double* plan_y,
                                 const double* plan_yaw, double max_vel_x, double max_vel_theta, int32_t estimate_orient,
                                 int32_t min_samples, int32_t guess_backwards_motion) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!plan_yaw) return fail(TEB_AMD_ERR_INVALID_ARG, "null plan_yaw");
  if ((rc = extend_batch(h, b))) return rc;
  if ((rc = stage(h, n_plan, plan_x, plan_y, plan_yaw))) return rc;
  hipLaunchKernelGGL(init_plan_kernel, dim3(1), dim3(kThreads), 0, h->stream, batch_of(h), b, n_plan, h->stage_x.p, h->stage_y.p,
                     h->stage_yaw.p, max_vel_x, max_vel_theta, estimate_orient, min_samples, guess_backwards_motion, h->err_flag.p);
  return finish_init(h);
}

int teb_amd_init_trajectory_path(teb_amd_handle_t* h, int32_t b, int32_t n_path, const double* path_x, const double* path_y,
                                 double max_vel_x, double max_vel_theta, const double* max_acc_x, const double* start_orientation,
                                 const double* goal_orientation, int32_t min_samples, int32_t guess_backwards_motion) {
  (void)max_vel_theta;   // unused by the reference too (its angular time step code is commented out, timed_elastic_band.hpp:118-140)
  int rc = check_handle(h);
  if (rc) return rc;
  if ((rc = extend_batch(h, b))) return rc;
  if ((rc = stage(h, n_path, path_x, path_y, nullptr))) return rc;
  hipLaunchKernelGGL(init_path_kernel, dim3(1), dim3(kThreads), 0, h->stream, batch_of(h), b, n_path, h->stage_x.p, h->stage_y.p,
                     max_vel_x, max_acc_x ? 1 : 0, max_acc_x ? *max_acc_x : 0.0, start_orientation ? 1 : 0,
                     start_orientation ? *start_orientation : 0.0, goal_orientation ? 1 : 0, goal_orientation ? *goal_orientation : 0.0,
                     min_samples, guess_backwards_motion, h->err_flag.p);
  return finish_init(h);
}

int teb_amd_update_and_prune(teb_amd_handle_t* h, int32_t b, const double* new_start, const double* new_goal, int32_t min_samples) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (h->B <= 0) return fail(TEB_AMD_ERR_INVALID_ARG, "no TEBs on the device");
  if (b >= h->B || b < -1) return fail(TEB_AMD_ERR_INVALID_ARG, "TEB index out of range");
  const double z[3] = {0, 0, 0};
  const double* s = new_start ? new_start : z;
  const double* g = new_goal ? new_goal : z;
  hipLaunchKernelGGL(prune_kernel, dim3(b < 0 ? h->B : 1), dim3(kThreads), 4 * (size_t)h->stride * sizeof(double), h->stream,
                     batch_of(h), b < 0 ? 0 : b, new_start ? 1 : 0, s[0], s[1], s[2], new_goal ? 1 : 0, g[0], g[1], g[2], min_samples);
  HIPCHK(hipGetLastError());
  h->consumers_valid = false; h->nmax_known = -1;
  h->hsig_mode = 0;   // the bands change: signatures of an earlier teb_amd_compute_h_signatures call are stale
  return TEB_AMD_OK;
}

namespace {
int set_velocity(teb_amd_handle* h, DevBuf<int>& flag, DevBuf<double>& vel, int b, int fixed, const double* v) {
  if (h->B <= 0) return fail(TEB_AMD_ERR_INVALID_ARG, "no TEBs on the device");
  if (b >= h->B || b < -1) return fail(TEB_AMD_ERR_INVALID_ARG, "TEB index out of range");
  const int b0 = b < 0 ? 0 : b, b1 = b < 0 ? h->B : b + 1;
  std::vector<int> f(b1 - b0, fixed ? 1 : 0);
  HIPCHK(hipMemcpyAsync(flag.p + b0, f.data(), f.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  std::vector<double> vv;
  if (v) {
    for (int k = b0; k < b1; ++k) { vv.push_back(v[0]); vv.push_back(v[1]); vv.push_back(v[2]); }
    HIPCHK(hipMemcpyAsync(vel.p + 3 * b0, vv.data(), vv.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  h->consumers_valid = false; h->nmax_known = -1;
  return TEB_AMD_OK;
}
}  // namespace

int teb_amd_set_velocity_start(teb_amd_handle_t* h, int32_t b, int32_t fixed, const double* v) {
  int rc = check_handle(h);
  if (rc) return rc;
  return set_velocity(h, h->has_vs, h->vs, b, fixed, v);
}

int teb_amd_set_velocity_goal(teb_amd_handle_t* h, int32_t b, int32_t fixed, const double* v) {
  int rc = check_handle(h);
  if (rc) return rc;
  return set_velocity(h, h->has_vg, h->vg, b, fixed, v);
}

int teb_amd_get_pose_counts(teb_amd_handle_t* h, int32_t* n, int32_t* count) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (count) *count = h->B;
  if (n && h->B > 0) {
    HIPCHK(hipMemcpyAsync(n, h->n.p, h->B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return TEB_AMD_OK;
}

int teb_amd_get_velocity_command(teb_amd_handle_t* h, int32_t b, int32_t look_ahead_poses, int32_t prevent_look_ahead_poses_near_goal,
                                 double* vx, double* vy, double* omega, int32_t* ok) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (b < 0 || b >= h->B) return fail(TEB_AMD_ERR_INVALID_ARG, "TEB index out of range");
  if ((rc = run_consumers(h, look_ahead_poses, prevent_look_ahead_poses_near_goal))) return rc;
  double c[4];
  HIPCHK(hipMemcpyAsync(c, h->out_cmd.p + 4 * (size_t)b, sizeof(c), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (vx) *vx = c[0]; if (vy) *vy = c[1]; if (omega) *omega = c[2]; if (ok) *ok = c[3] != 0;
  return TEB_AMD_OK;
}

int teb_amd_get_velocity_profile(teb_amd_handle_t* h, int32_t b, double* out, int32_t capacity_rows, int32_t* rows) {
  int rc = check_handle(h);
  if (rc) return rc;
  int n = 0;
  if ((rc = band_n(h, b, &n))) return rc;
  if (rows) *rows = n + 1;
  if (!out) return TEB_AMD_OK;
  if (capacity_rows < n + 1) return fail(TEB_AMD_ERR_CAPACITY, "velocity profile needs n+1 rows");
  if ((rc = run_consumers(h, h->consumers_valid ? h->consumers_la : 1, h->consumers_valid ? h->consumers_prevent : 0))) return rc;
  HIPCHK(hipMemcpyAsync(out, h->out_prof.p + (size_t)b * (h->stride + 1) * 3, (size_t)(n + 1) * 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return TEB_AMD_OK;
}

int teb_amd_get_full_trajectory(teb_amd_handle_t* h, int32_t b, double* out, int32_t capacity_rows, int32_t* rows) {
  int rc = check_handle(h);
  if (rc) return rc;
  int n = 0;
  if ((rc = band_n(h, b, &n))) return rc;
  if (rows) *rows = n;
  if (!out) return TEB_AMD_OK;
  if (capacity_rows < n) return fail(TEB_AMD_ERR_CAPACITY, "trajectory needs n rows");
  if ((rc = run_consumers(h, h->consumers_valid ? h->consumers_la : 1, h->consumers_valid ? h->consumers_prevent : 0))) return rc;
  HIPCHK(hipMemcpyAsync(out, h->out_traj.p + (size_t)b * h->stride * 7, (size_t)n * 7 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return TEB_AMD_OK;
}

// TebOptimalPlanner::hasDiverged (src/optimal_planner.cpp:1023-1039) reads optimizer_->batchStatistics().back().chi2. g2o sizes that
// vector to the REQUESTED iteration count of the last optimize() call and fills one entry per executed iteration: when the LM loop of
// that call terminated early the last entry is still zero-initialised and the planner does not report a divergence whatever chi2 was
// reached. The kernel returns the number of iterations of the band's last optimize() call beside the chi2 after its last iteration.
static bool diverged_rule(const teb_amd_config_t& c, int last_inner, int iters, int last_iters, double chi2) {
  if (!c.divergence_detection_enable) return false;                          // :1026-1027
  if (iters <= 0 || last_inner <= 0) return false;                           // no statistics yet, :1031-1033
  const double back = last_iters == last_inner ? chi2 : 0.0;                 // .back() of a vector resized to `iterations`
  return back > c.divergence_detection_max_chi_squared;                      // :1038
}

int teb_amd_has_diverged(teb_amd_handle_t* h, int32_t b, int32_t* diverged) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (b < 0 || b >= h->B || !diverged) return fail(TEB_AMD_ERR_INVALID_ARG, "TEB index out of range");
  *diverged = 0;
  if (!h->cfg.divergence_detection_enable) return TEB_AMD_OK;
  double chi2 = 0; int iters = 0, last = 0;
  HIPCHK(hipMemcpyAsync(&chi2, h->chi2.p + b, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(&iters, h->iters.p + b, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(&last, h->last_iters.p + b, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  *diverged = diverged_rule(h->cfg, h->last_inner, iters, last, chi2);
  return TEB_AMD_OK;
}

// What optimizer_->batchStatistics() of every resident band would hold after the last teb_amd_optimize_batch: available [count] =
// the vector is not empty (statistics were switched on for that call, src/optimal_planner.cpp:331, and optimize() ran), back_chi2
// [count] = .back().chi2 (zero when the band's last optimize() call stopped before its last requested iteration). A binding keeps
// the pair per planner object and applies hasDiverged's rule to the configuration it holds at that time.
int teb_amd_get_batch_statistics(teb_amd_handle_t* h, int32_t* available, double* back_chi2) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!available || !back_chi2) return fail(TEB_AMD_ERR_INVALID_ARG, "null argument");
  const int B = h->B;
  if (B <= 0) return TEB_AMD_OK;
  std::vector<double> chi2(B); std::vector<int> iters(B), last(B);
  HIPCHK(hipMemcpyAsync(chi2.data(), h->chi2.p, B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(iters.data(), h->iters.p, B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(last.data(), h->last_iters.p, B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int b = 0; b < B; ++b) {
    available[b] = h->cfg.divergence_detection_enable && iters[b] > 0 && h->last_inner > 0;
    back_chi2[b] = (available[b] && last[b] == h->last_inner) ? chi2[b] : 0.0;
  }
  return TEB_AMD_OK;
}

// ---- f4: feasibility of the resident bands against a costmap grid (kernel in teb_feasibility.hpp) --------------------------------------
int teb_amd_set_costmap(teb_amd_handle_t* h, const uint8_t* cells, int32_t size_x, int32_t size_y, double resolution, double origin_x,
                        double origin_y) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!cells || size_x <= 0 || size_y <= 0 || !(resolution > 0)) return fail(TEB_AMD_ERR_INVALID_ARG, "teb_amd_set_costmap: bad grid");
  const size_t bytes = (size_t)size_x * size_y;
  if (h->cm_cells.n < bytes) { h->cm_cells.free(); HIPCHK(h->cm_cells.alloc(bytes)); }
  HIPCHK(hipMemcpyAsync(h->cm_cells.p, cells, bytes, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));   // the caller's buffer may go away
  h->cm_sx = size_x; h->cm_sy = size_y; h->cm_res = resolution; h->cm_ox = origin_x; h->cm_oy = origin_y;
  return TEB_AMD_OK;
}

int teb_amd_is_trajectory_feasible(teb_amd_handle_t* h, int32_t b, int32_t nf, const double* fx, const double* fy, double inscribed_radius,
                                   double min_res_angular, int32_t look_ahead_idx, double lookahead_distance, int32_t* feasible,
                                   int32_t* first_infeasible) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (h->cm_sx <= 0) return fail(TEB_AMD_ERR_INVALID_ARG, "teb_amd_is_trajectory_feasible: no costmap (teb_amd_set_costmap)");
  if (!feasible || nf < 1 || nf > kMaxFeasFootprint || !fx || !fy) return fail(TEB_AMD_ERR_INVALID_ARG, "bad footprint / output");
  if (!(inscribed_radius > 0) || !(min_res_angular > 0)) return fail(TEB_AMD_ERR_INVALID_ARG, "inscribed_radius and the angular resolution must be > 0");
  if (h->B <= 0 || b >= h->B || b < -1) return fail(TEB_AMD_ERR_INVALID_ARG, "TEB index out of range");
  if (h->stride > 1024) return fail(TEB_AMD_ERR_CAPACITY, "feasibility check supports bands up to 1024 poses");
  const int first = b < 0 ? 0 : b, count = b < 0 ? h->B : 1;
  if (h->cm_fp.n < 2 * (size_t)kMaxFeasFootprint) { h->cm_fp.free(); HIPCHK(h->cm_fp.alloc(2 * (size_t)kMaxFeasFootprint)); }
  if (h->cm_out.n < 2 * (size_t)h->max_tebs + 1) { h->cm_out.free(); HIPCHK(h->cm_out.alloc(2 * (size_t)h->max_tebs + 1)); }
  HIPCHK(hipMemcpyAsync(h->cm_fp.p, fx, nf * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->cm_fp.p + kMaxFeasFootprint, fy, nf * sizeof(double), hipMemcpyHostToDevice, h->stream));
  int* ovf = h->cm_out.p + 2 * (size_t)h->max_tebs;
  HIPCHK(hipMemsetAsync(ovf, 0, sizeof(int), h->stream));
  GridDev g{h->cm_cells.p, h->cm_sx, h->cm_sy, h->cm_res, h->cm_ox, h->cm_oy};
  hipLaunchKernelGGL(feasibility_kernel, dim3(count), dim3(kFeasThreads), 0, h->stream, h->n.p, h->x.p, h->y.p, h->th.p, h->stride, first, g, nf,
                     h->cm_fp.p, h->cm_fp.p + kMaxFeasFootprint, inscribed_radius, min_res_angular, look_ahead_idx, lookahead_distance,
                     1 << 22, h->cm_out.p, h->cm_out.p + h->max_tebs, ovf);
  HIPCHK(hipGetLastError());
  std::vector<int> fe(count), fi(count);
  int o = 0;
  HIPCHK(hipMemcpyAsync(fe.data(), h->cm_out.p, count * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(fi.data(), h->cm_out.p + h->max_tebs, count * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(&o, ovf, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (o) return fail(TEB_AMD_ERR_CAPACITY, "feasibility check: more than 2^22 interpolated samples requested (inscribed radius / angular resolution too small)");
  for (int k = 0; k < count; ++k) { feasible[k] = fe[k]; if (first_infeasible) first_infeasible[k] = fi[k]; }
  return TEB_AMD_OK;
}

namespace {
// H-signatures of the B bands of bt (the batch or the candidate scratch strips) into out [B * W], W = M (3-D) or 2 (2-D)
int launch_hsig(teb_amd_handle* h, const BatchDev& bt, int B, double prescaler, double* out) {
  SceneDev sc = scene_of(h);
  const int M = h->M;
  if (B <= 0) return TEB_AMD_OK;
  if (h->cfg.include_dynamic_obstacles) {
    if (M > 0) {
      // one lane per (band, obstacle) when that fills the chip; otherwise lanes over (obstacle, segment): same bits, lower latency
      const char* force = h->opt.hsig3d_kernel == TEB_AMD_HSIG3D_WIDE ? "wide" : h->opt.hsig3d_kernel == TEB_AMD_HSIG3D_SMALL ? "small" : nullptr;   // teb_amd_options_t::hsig3d_kernel pins one of the two kernels (tests, tools/hsig_bench.py)
      const bool wide = force ? std::strcmp(force, "wide") == 0 : (long long)B * M >= 32768;
      if (wide)
        hipLaunchKernelGGL(hsig3d_kernel, dim3((M + kThreads - 1) / kThreads, B), dim3(kThreads), 3 * (size_t)h->stride * sizeof(double),
                           h->stream, sc, bt, out);
      else
        hipLaunchKernelGGL(hsig3d_small_kernel, dim3((M + kHsTile - 1) / kHsTile, B), dim3(kThreads), 3 * (size_t)h->stride * sizeof(double),
                           h->stream, sc, bt, out);
      HIPCHK(hipGetLastError());
    }
  } else {
    if (M > 0 && !h->hs_prod_valid) {   // the band-independent factor of A_l: once per obstacle table (O(M^2))
      hipLaunchKernelGGL(hsig2d_prod_kernel, dim3((M + kThreads - 1) / kThreads), dim3(kThreads), 0, h->stream, sc, h->hs_pre.p,
                         h->hs_pim.p, h->hs_pex.p);
      HIPCHK(hipGetLastError());
      h->hs_prod_valid = true;
    }
    hipLaunchKernelGGL(hsig2d_kernel, dim3(B), dim3(kThreads), 2 * (size_t)h->stride * sizeof(double), h->stream, sc, bt, prescaler,
                       h->hs_pre.p, h->hs_pim.p, h->hs_pex.p, out);
    HIPCHK(hipGetLastError());
  }
  return TEB_AMD_OK;
}
}  // namespace

// ---- f3 (arithmetic core): equivalence classes of the device-resident bands (kernels in teb_hsig.hpp) ----------------------------
int teb_amd_compute_h_signatures(teb_amd_handle_t* h, double prescaler, double* values, int32_t* width) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (h->B <= 0) return fail(TEB_AMD_ERR_INVALID_ARG, "no TEBs on the device");
  SceneDev sc = scene_of(h);
  BatchDev bt = batch_of(h);
  const int M = h->M, B = h->B;
  const int mode = h->cfg.include_dynamic_obstacles ? 3 : 2;   // homotopy_class_planner.hpp:50
  const int W = mode == 3 ? M : 2;
  if (width) *width = W;
  h->hsig_mode = mode; h->hsig_B = B; h->hsig_M = M; h->hsig_prescaler = prescaler;
  h->hsig_host.assign((size_t)B * (W > 0 ? W : 1), 0.0);
  if ((rc = launch_hsig(h, bt, B, prescaler, h->hsig.p))) return rc;
  if ((size_t)B * W > 0)
    HIPCHK(hipMemcpyAsync(h->hsig_host.data(), h->hsig.p, (size_t)B * W * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (values && (size_t)B * W > 0) std::memcpy(values, h->hsig_host.data(), (size_t)B * W * sizeof(double));
  return TEB_AMD_OK;
}

int teb_amd_filter_equivalence_classes(teb_amd_handle_t* h, double threshold, int32_t best, int32_t max_number_plans_in_current_class,
                                       int32_t* keep, int32_t* valid, int32_t* reasonable) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (h->hsig_mode == 0 || h->hsig_B != h->B) return fail(TEB_AMD_ERR_INVALID_ARG, "call teb_amd_compute_h_signatures first");
  const int B = h->hsig_B, mode = h->hsig_mode, W = mode == 3 ? h->hsig_M : 2;
  const double* sig = h->hsig_host.data();
  auto row = [&](int b) { return sig + (size_t)b * W; };
  auto is_valid = [&](int b) { for (int k = 0; k < W; ++k) if (!std::isfinite(row(b)[k])) return false; return true; };
  auto is_reasonable = [&](int b) { if (mode == 2) return true; for (int k = 0; k < W; ++k) if (row(b)[k] > 1.0) return false; return true; };
  auto sign_of = [](double z) { return z == 0 ? 0 : (z < 0 ? -1 : 1); };
  auto is_equal = [&](int a, int b) {   // cls[a]->isEqual(*cls[b])
    const double* x = row(a); const double* y = row(b);
    if (mode == 2) return std::fabs(y[0] - x[0]) <= threshold && std::fabs(y[1] - x[1]) <= threshold;   // h_signature.h:196-204
    for (int i = 0; i < W; ++i) {                                                                          // h_signature.h:360-377
      if (std::fabs(y[i]) < threshold || std::fabs(x[i]) < threshold) continue;   // far-away obstacle: ignored
      if (sign_of(y[i]) != sign_of(x[i])) return false;
    }
    return true;
  };
  std::vector<int> order(B), vld(B), classes;
  for (int b = 0; b < B; ++b) { order[b] = b; vld[b] = is_valid(b); }
  const bool has_best = best >= 0 && best < B;
  if (has_best) {   // best_teb_eq_class_ = calculateEquivalenceClass(best_teb_), src/homotopy_class_planner.cpp:224-227
    std::swap(order[0], order[best]);
    h->best_class.assign(row(best), row(best) + W); h->best_class_mode = mode;
  }
  // isInBestTebClass / numTebsInBestTebClass use best_teb_eq_class_, which outlives the band it was computed from (:385-410)
  const bool have_best_class = h->best_class_mode == mode && (int)h->best_class.size() == W;
  auto equal_to_best = [&](int b) {   // best_teb_eq_class_->isEqual(*cls[b])
    const double* x = h->best_class.data(); const double* y = row(b);
    if (mode == 2) return std::fabs(y[0] - x[0]) <= threshold && std::fabs(y[1] - x[1]) <= threshold;
    for (int i = 0; i < W; ++i) {
      if (std::fabs(y[i]) < threshold || std::fabs(x[i]) < threshold) continue;
      if (sign_of(y[i]) != sign_of(x[i])) return false;
    }
    return true;
  };
  std::vector<int> kp(B, 0);
  for (int k = 0; k < B; ++k) {
    const int b = order[k];
    if (!vld[b]) continue;                                   // "Ignoring invalid H-signature"
    bool has = false;
    for (int c : classes) if (is_equal(b, c)) { has = true; break; }
    if (has) {
      const bool in_best = have_best_class && equal_to_best(b);
      int count = 0;
      if (have_best_class) for (int c : classes) if (equal_to_best(c)) ++count;
      if (!in_best || count >= max_number_plans_in_current_class) continue;
    }
    classes.push_back(b); kp[b] = 1;
  }
  for (int b = 0; b < B; ++b) {
    if (keep) keep[b] = kp[b];
    if (valid) valid[b] = vld[b];
    if (reasonable) reasonable[b] = is_reasonable(b);
  }
  return TEB_AMD_OK;
}

// ---- f3 (candidate generation): createGraph + DepthFirst + addAndInitNewTeb (kernels in teb_graph.hpp) ---------------------------
void teb_amd_hcp_params_default(teb_amd_hcp_params_t* p) {   // teb_config.h:352-367, :260, :293
  if (!p) return;
  std::memset(p, 0, sizeof(*p));
  p->simple_exploration = 0; p->roadmap_graph_no_samples = 15; p->roadmap_graph_area_width = 6; p->roadmap_graph_area_length_scale = 1.0;
  p->obstacle_heading_threshold = 0.45; p->xy_goal_tolerance = 0.2; p->max_number_classes = 5;
  p->max_number_plans_in_current_class = 1;   // no constructor default in the reference; 1 is its dynamic-reconfigure default
  p->h_signature_prescaler = 1; p->h_signature_threshold = 0.1; p->allow_init_with_backwards_motion = 0;
  p->delete_detours_backwards = 1; p->detours_orientation_tolerance = M_PI / 2.0; p->length_start_orientation_vector = 0.4;   // :374-377
  p->max_ratio_detours_duration_best_duration = 3.0;
  p->global_plan_overwrite_orientation = 1; p->viapoints_all_candidates = 1;   // :259, :370
}

namespace {
constexpr int kCandChunk = 64;   // candidate paths initialised and classified per device round trip

int ensure_candidate_buffers(teb_amd_handle* h) {
  if (h->cand_ready) return TEB_AMD_OK;
  const size_t KS = (size_t)kCandChunk * h->stride, W = (size_t)(h->max_obst > 2 ? h->max_obst : 2);
  bool ok = true;
  auto A = [&](hipError_t e) { if (e != hipSuccess) ok = false; };
  A(h->cand_x.alloc(KS)); A(h->cand_y.alloc(KS)); A(h->cand_th.alloc(KS)); A(h->cand_dt.alloc(KS)); A(h->cand_n.alloc(kCandChunk));
  A(h->cand_sig.alloc(std::max<size_t>(kCandChunk * W, 4 * (size_t)h->max_tebs)));   // signatures of a chunk / detour statistics of the batch
  A(h->cand_px.alloc((size_t)kCandChunk * (h->stride + 1)));
  A(h->cand_py.alloc((size_t)kCandChunk * (h->stride + 1))); A(h->cand_off.alloc(kCandChunk + 1)); A(h->cand_map.alloc(h->max_tebs));
  if (!ok) return fail(TEB_AMD_ERR_HIP, "candidate scratch allocation failed");
  h->cand_ready = true;
  return TEB_AMD_OK;
}

BatchDev candidates_of(teb_amd_handle* h) {   // the scratch strips seen as a batch of kCandChunk bands
  BatchDev c = batch_of(h);
  c.B = kCandChunk;
  c.n = h->cand_n.p; c.x = h->cand_x.p; c.y = h->cand_y.p; c.th = h->cand_th.p; c.dt = h->cand_dt.p;
  return c;
}

// isValid / isEqual of HSignature and HSignature3d (h_signature.h:190-226, 349-409) on rows of W doubles
struct ClassTable {
  int mode = 2, W = 2, max_in_best = 1;
  double thr = 0.1;
  std::vector<std::vector<double>> classes;   // equivalence_classes_
  bool has_best = false;
  std::vector<double> best;                   // best_teb_eq_class_
  static int sign_of(double z) { return z == 0 ? 0 : (z < 0 ? -1 : 1); }
  bool valid(const double* v) const { for (int k = 0; k < W; ++k) if (!std::isfinite(v[k])) return false; return true; }
  bool equal(const double* a, const double* b) const {   // a.isEqual(b)
    if (mode == 2) return std::fabs(b[0] - a[0]) <= thr && std::fabs(b[1] - a[1]) <= thr;
    for (int i = 0; i < W; ++i) {
      if (std::fabs(b[i]) < thr || std::fabs(a[i]) < thr) continue;
      if (sign_of(b[i]) != sign_of(a[i])) return false;
    }
    return true;
  }
  bool add_if_new(const double* v) {   // addEquivalenceClassIfNew, src/homotopy_class_planner.cpp:189-212
    if (!valid(v)) return false;
    bool has = false;
    for (const auto& c : classes) if (equal(v, c.data())) { has = true; break; }
    if (has) {
      const bool in_best = has_best && equal(best.data(), v);
      int count = 0;
      if (has_best) for (const auto& c : classes) if (equal(best.data(), c.data())) ++count;
      if (!in_best || count >= max_in_best) return false;
    }
    classes.emplace_back(v, v + W);
    return true;
  }
};

// GraphSearchInterface::DepthFirst (src/graph_search.cpp:45-91) as a resumable generator of start-goal paths in the reference's order
struct PathEnumerator {
  const std::vector<std::vector<int>>* adj;
  int goal;
  struct Frame { int v; int phase; size_t k; };
  std::vector<Frame> stack;
  std::vector<int> visited;
  std::vector<char> on_path;
  int64_t expansions = 0, max_expansions = 0;   // > 0: give up after that many vertex expansions (dead-end subtrees can be exponential)
  PathEnumerator(const std::vector<std::vector<int>>& a, int start, int goal_) : adj(&a), goal(goal_), on_path(a.size(), 0) {
    stack.push_back({start, 0, 0}); visited.push_back(start); on_path[start] = 1;
  }
  bool next(std::vector<int>& path) {
    while (!stack.empty()) {
      if (max_expansions > 0 && ++expansions > max_expansions) return false;
      Frame& f = stack.back();
      const std::vector<int>& out = (*adj)[f.v];
      if (f.phase == 0) {       // first loop: the goal, if adjacent, closes one path
        f.phase = 1; f.k = 0;
        if (std::find(out.begin(), out.end(), goal) != out.end()) { path = visited; path.push_back(goal); return true; }
      }
      bool descended = false;
      while (f.k < out.size()) {   // second loop: recursion into every adjacent vertex not yet on the path
        const int w = out[f.k++];
        if (on_path[w] || w == goal) continue;
        visited.push_back(w); on_path[w] = 1;
        stack.push_back({w, 0, 0});
        descended = true;
        break;
      }
      if (descended) continue;
      on_path[f.v] = 0; visited.pop_back(); stack.pop_back();
    }
    return false;
  }
};
}  // namespace

int teb_amd_explore_candidates(teb_amd_handle_t* h, const teb_amd_hcp_params_t* p, const double* start, const double* goal,
                               double dist_to_obst, const double* start_vel, int32_t free_goal_vel, int32_t best,
                               const double* unit_samples, int64_t max_paths, int32_t* n_total, int32_t* n_vertices, int32_t* n_paths,
                               int32_t n_plan, const double* plan_x, const double* plan_y, const double* plan_yaw,
                               int32_t* initial_plan_teb) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!p || !start || !goal) return fail(TEB_AMD_ERR_INVALID_ARG, "null argument");
  if (n_plan > 0 && (!plan_x || !plan_y || !plan_yaw)) return fail(TEB_AMD_ERR_INVALID_ARG, "null initial plan");
  if (initial_plan_teb) *initial_plan_teb = -1;
  if ((rc = ensure_candidate_buffers(h))) return rc;
  const teb_amd_config_t& c = h->cfg;
  const int M = h->M;
  const int mode = c.include_dynamic_obstacles ? 3 : 2, W = mode == 3 ? M : 2;
  const int slots = std::min<int>(p->max_number_classes, h->max_tebs);
  h->g_cap = 0;
  if (n_vertices) *n_vertices = 0;
  if (n_paths) *n_paths = 0;
  if (n_total) *n_total = h->B;
  // equivalence_classes_ of the existing bands
  ClassTable ct;
  ct.mode = mode; ct.W = W; ct.thr = p->h_signature_threshold; ct.max_in_best = p->max_number_plans_in_current_class;
  if (h->B > 0) {
    const bool fresh = h->hsig_mode == mode && h->hsig_B == h->B && h->hsig_M == M && h->hsig_prescaler == p->h_signature_prescaler;
    if (!fresh && (rc = teb_amd_compute_h_signatures(h, p->h_signature_prescaler, nullptr, nullptr))) return rc;   // else: renew just did
    for (int b = 0; b < h->B; ++b) ct.classes.emplace_back(h->hsig_host.data() + (size_t)b * W, h->hsig_host.data() + (size_t)(b + 1) * W);
    if (best >= 0 && best < h->B) { h->best_class = ct.classes[best]; h->best_class_mode = mode; }
  }
  if (h->best_class_mode == mode && (int)h->best_class.size() == W) { ct.has_best = true; ct.best = h->best_class; }   // best_teb_eq_class_
  h->hsig_mode = 0;   // the batch is about to change: signatures have to be recomputed before the next filter call
  const int n_old = h->B;
  // tebs_.push_back(candidate) for the accepted candidates of a chunk: scratch bands -> the next slots of the batch, one gather;
  // default attributes of a new TebOptimalPlanner (fixed zero start / goal velocity), then setVelocityStart / setVelocityGoalFree
  auto accept_all = [&](const std::vector<int>& cand) -> int {
    if (cand.empty()) return TEB_AMD_OK;
    const int slot0 = h->B, cnt = (int)cand.size();
    int r = extend_batch(h, slot0 + cnt - 1);
    if (r) return r;
    HIPCHK(hipMemcpyAsync(h->cand_map.p, cand.data(), cnt * sizeof(int), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(move_bands_kernel, dim3(cnt), dim3(kThreads), 0, h->stream, candidates_of(h), batch_of(h), h->cand_map.p, slot0, 0);
    HIPCHK(hipGetLastError());
    if (start_vel) {
      std::vector<double> vv;
      for (int k = 0; k < cnt; ++k) { vv.push_back(start_vel[0]); vv.push_back(start_vel[1]); vv.push_back(start_vel[2]); }
      HIPCHK(hipMemcpyAsync(h->vs.p + 3 * slot0, vv.data(), vv.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    if (free_goal_vel) HIPCHK(hipMemsetAsync(h->has_vg.p + slot0, 0, cnt * sizeof(int), h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));   // the staging vectors live on this stack frame
    return TEB_AMD_OK;
  };
  std::vector<double> sig((size_t)kCandChunk * (W > 0 ? W : 1));
  auto classify = [&](int count) -> int {      // signatures of scratch bands 0..count-1 -> sig
    int r = launch_hsig(h, candidates_of(h), count, p->h_signature_prescaler, h->cand_sig.p);
    if (r) return r;
    if ((size_t)count * W > 0) HIPCHK(hipMemcpyAsync(sig.data(), h->cand_sig.p, (size_t)count * W * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    int err = 0;
    HIPCHK(hipMemcpyAsync(&err, h->err_flag.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (err) return fail(TEB_AMD_ERR_CAPACITY, "candidate band needs more poses than max_poses");
    return TEB_AMD_OK;
  };
  h->consumers_valid = false; h->nmax_known = -1;
  // ---- the initial plan as a candidate: addAndInitNewTeb(*initial_plan_, ...), src/homotopy_class_planner.cpp:326-329, 412-440
  int initial_idx = -1;   // initial_plan_teb_
  if (n_plan > 0 && h->B < slots) {
    if ((rc = stage(h, n_plan, plan_x, plan_y, plan_yaw))) return rc;
    hipLaunchKernelGGL(init_plan_kernel, dim3(1), dim3(kThreads), 0, h->stream, candidates_of(h), 0, n_plan, h->stage_x.p, h->stage_y.p,
                       h->stage_yaw.p, c.max_vel_x, c.max_vel_theta, p->global_plan_overwrite_orientation, c.min_samples,
                       p->allow_init_with_backwards_motion, h->err_flag.p);
    HIPCHK(hipGetLastError());
    if ((rc = classify(1))) return rc;
    h->initial_class.assign(sig.data(), sig.data() + W); h->initial_class_mode = mode;   // initial_plan_eq_class_
    if (ct.add_if_new(sig.data())) {
      if ((rc = accept_all({0}))) return rc;
      initial_idx = h->B - 1;
    }
  }
  auto body = [&]() -> int {
  if (h->B >= slots) return TEB_AMD_OK;                                     // src/graph_search.cpp:99-100, 231-232
  const double sx = start[0], sy = start[1], gx = goal[0], gy = goal[1];
  double dfx = gx - sx, dfy = gy - sy;
  const double start_goal_dist = std::sqrt(dfx * dfx + dfy * dfy);
  if (start_goal_dist < p->xy_goal_tolerance) {                             // :104-113, :237-246
    if (h->B == 0) {                                                        // addAndInitNewTeb(start, goal, ...), hcp.cpp:358-384
      HIPCHK(hipMemsetAsync(h->err_flag.p, 0, sizeof(int), h->stream));
      hipLaunchKernelGGL(init_line_kernel, dim3(1), dim3(kThreads), 0, h->stream, candidates_of(h), 0, start[0], start[1], start[2],
                         goal[0], goal[1], goal[2], 0.0, c.max_vel_x, c.min_samples, p->allow_init_with_backwards_motion, h->err_flag.p);
      HIPCHK(hipGetLastError());
      if ((rc = classify(1))) return rc;
      if (ct.add_if_new(sig.data()) && (rc = accept_all({0}))) return rc;
    }
    return TEB_AMD_OK;
  }
  // ---- vertices (host: O(M) on the centroids kept from teb_amd_set_obstacles) ------------------------------------------------------
  auto normalize = [](double& x, double& y) { const double z = x * x + y * y; if (z > 0) { const double n = std::sqrt(z); x = x / n; y = y / n; } };
  std::vector<double> vx{sx}, vy{sy};
  GraphArgs ga;
  ga.keypoint = p->simple_exploration ? 1 : 0; ga.near_u = ga.near_v = -1; ga.thr = p->obstacle_heading_threshold;
  ga.sox = std::cos(start[2]); ga.soy = std::sin(start[2]);
  if (p->simple_exploration) {                                              // lrKeyPointGraph::createGraph, :115-153
    double nx = -dfy, ny = dfx;
    normalize(nx, ny);
    nx = nx * dist_to_obst; ny = ny * dist_to_obst;
    normalize(dfx, dfy);
    double min_dist = std::numeric_limits<double>::max();
    for (int o = 0; o < M; ++o) {
      const double ox = h->host_cx[o] - sx, oy = h->host_cy[o] - sy;
      const double dist = std::sqrt(ox * ox + oy * oy);
      if ((ox * dfx + oy * dfy) / dist < 0.1) continue;                     // obstacle not in front of the start
      vx.push_back(h->host_cx[o] + nx); vy.push_back(h->host_cy[o] + ny);
      vx.push_back(h->host_cx[o] - nx); vy.push_back(h->host_cy[o] - ny);
      if (p->obstacle_heading_threshold && dist < min_dist) { min_dist = dist; ga.near_u = (int)vx.size() - 2; ga.near_v = (int)vx.size() - 1; }
    }
    ga.min_dist = 0.5 * dist_to_obst;
  } else {                                                                  // ProbRoadmapGraph::createGraph, :247-290
    double nx = -dfy, ny = dfx;
    normalize(nx, ny);
    const double area_width = p->roadmap_graph_area_width;
    const double len = start_goal_dist * p->roadmap_graph_area_length_scale;
    const double phi = std::atan2(dfy, dfx);
    double ox, oy;
    if (p->roadmap_graph_area_length_scale != 1.0) {
      double ux = dfx, uy = dfy;
      normalize(ux, uy);
      const double f = 0.5 * (1.0 - p->roadmap_graph_area_length_scale) * start_goal_dist, w2 = 0.5 * area_width;
      ox = (sx + f * ux) - w2 * nx; oy = (sy + f * uy) - w2 * ny;
    } else {
      const double w2 = 0.5 * area_width;
      ox = sx - w2 * nx; oy = sy - w2 * ny;
    }
    normalize(dfx, dfy);
    int drawn = 0;
    auto draw = [&](double a, double b) {    // boost::random::uniform_real_distribution<double>(a, b) on the 32-bit engine
      if (unit_samples) return unit_samples[drawn++] * (b - a) + a;
      for (;;) {
        const double numerator = (double)(h->rnd_generator() - std::mt19937::min());
        const double divisor = (double)(std::mt19937::max() - std::mt19937::min()) + 1;
        const double result = numerator / divisor * (b - a) + a;
        if (result < b) return result;
      }
    };
    const double cphi = std::cos(phi), sphi = std::sin(phi);
    for (int i = 0; i < p->roadmap_graph_no_samples; ++i) {
      const double uy = draw(0, area_width);   // GCC evaluates Eigen::Vector2d(distribution_x(g), distribution_y(g)) right to left
      const double ux = draw(0, len);
      vx.push_back(ox + (cphi * ux - sphi * uy)); vy.push_back(oy + (sphi * ux + cphi * uy));
    }
    ga.min_dist = dist_to_obst;
  }
  vx.push_back(gx); vy.push_back(gy);
  const int N = (int)vx.size();
  if (n_vertices) *n_vertices = N;
  // ---- edges (device) -------------------------------------------------------------------------------------------------------------
  if (h->g_vx.n < (size_t)N) { h->g_vx.free(); h->g_vy.free(); HIPCHK(h->g_vx.alloc(N)); HIPCHK(h->g_vy.alloc(N)); }
  if (h->g_adj.n < (size_t)N * N) { h->g_adj.free(); HIPCHK(h->g_adj.alloc((size_t)N * N)); }
  HIPCHK(hipMemcpyAsync(h->g_vx.p, vx.data(), N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->g_vy.p, vy.data(), N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  ga.N = N; ga.gx = h->g_vx.p; ga.gy = h->g_vy.p; ga.dnx = dfx; ga.dny = dfy; ga.adj = h->g_adj.p;
  const long long pairs = (long long)N * N;
  hipLaunchKernelGGL(graph_edges_kernel, dim3((unsigned)((pairs + kThreads - 1) / kThreads)), dim3(kThreads), 0, h->stream, scene_of(h), ga);
  HIPCHK(hipGetLastError());
  std::vector<unsigned char> adjm((size_t)pairs);
  HIPCHK(hipMemcpyAsync(adjm.data(), h->g_adj.p, (size_t)pairs, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->g_cap = N;
  std::vector<std::vector<int>> adj(N);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) if (adjm[(size_t)i * N + j]) adj[i].push_back(j);   // add_edge order of the double loop
  // ---- paths in depth-first order, a chunk at a time: init + signature on the device, first come first served on the host ----------
  PathEnumerator en(adj, 0, N - 1);
  if (max_paths > 0) en.max_expansions = max_paths * 10000;
  std::vector<int> path, off;
  std::vector<double> px, py;
  int64_t examined = 0;
  bool more = true;
  while (more && h->B < slots) {
    off.assign(1, 0); px.clear(); py.clear();
    int count = 0;
    while (count < kCandChunk && (max_paths <= 0 || examined + count < max_paths)) {
      if (!en.next(path)) { more = false; break; }
      if ((int)path.size() > h->stride + 1) return fail(TEB_AMD_ERR_CAPACITY, "a start-goal path has more vertices than a band has poses (max_poses)");
      for (int v : path) { px.push_back(vx[v]); py.push_back(vy[v]); }
      off.push_back((int)px.size());
      ++count;
    }
    if (count == kCandChunk && max_paths > 0 && examined + count >= max_paths) more = false;
    if (count == 0) break;
    HIPCHK(hipMemcpyAsync(h->cand_off.p, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->cand_px.p, px.data(), px.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->cand_py.p, py.data(), py.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemsetAsync(h->err_flag.p, 0, sizeof(int), h->stream));
    hipLaunchKernelGGL(init_path_batch_kernel, dim3(count), dim3(kThreads), 0, h->stream, candidates_of(h), h->cand_off.p, h->cand_px.p,
                       h->cand_py.p, c.max_vel_x, c.acc_lim_x, start[2], goal[2], c.min_samples, p->allow_init_with_backwards_motion,
                       h->err_flag.p);
    HIPCHK(hipGetLastError());
    if ((rc = classify(count))) return rc;
    std::vector<int> accepted;
    for (int k = 0; k < count && h->B + (int)accepted.size() < slots; ++k) {
      ++examined;
      if (ct.add_if_new(sig.data() + (size_t)k * W)) accepted.push_back(k);
    }
    if ((rc = accept_all(accepted))) return rc;
  }
  if (n_paths) *n_paths = (int32_t)std::min<int64_t>(examined, std::numeric_limits<int32_t>::max());
  return TEB_AMD_OK;
  };
  if ((rc = body())) return rc;
  if (n_total) *n_total = h->B;
  // ---- getInitialPlanTEB (:495-536): the band created from the initial plan, else the first band of the initial plan's class
  const bool have_initial_class = h->initial_class_mode == mode && (int)h->initial_class.size() == W && ct.valid(h->initial_class.data());
  if (initial_idx < 0 && have_initial_class)
    for (int b = 0; b < h->B && b < (int)ct.classes.size(); ++b)
      if (ct.equal(ct.classes[b].data(), h->initial_class.data())) { initial_idx = b; break; }
  if (initial_plan_teb) *initial_plan_teb = initial_idx;
  // ---- updateReferenceTrajectoryViaPoints (:286-315): new candidates are born without via-points (their constructor gets none);
  //      all candidates get them (viapoints_all_candidates), or - with an initial plan - exactly those of the initial plan's class
  if (h->B > n_old) HIPCHK(hipMemsetAsync(h->via_en.p + n_old, 0, (h->B - n_old) * sizeof(int), h->stream));
  if (h->B > 0 && !((!p->viapoints_all_candidates && n_plan <= 0) || h->nvia <= 0 || c.weight_viapoint <= 0)) {
    std::vector<int> ve(h->B, 1);
    if (!p->viapoints_all_candidates)
      for (int b = 0; b < h->B; ++b) ve[b] = have_initial_class && b < (int)ct.classes.size() && ct.equal(h->initial_class.data(), ct.classes[b].data());
    HIPCHK(hipMemcpyAsync(h->via_en.p, ve.data(), h->B * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return TEB_AMD_OK;
}

int teb_amd_get_exploration_graph(teb_amd_handle_t* h, double* vx, double* vy, unsigned char* adjacency, int32_t capacity_vertices,
                                  int32_t* n_vertices) {
  int rc = check_handle(h);
  if (rc) return rc;
  const int N = (int)h->g_cap;
  if (n_vertices) *n_vertices = N;
  if (N == 0 || (!vx && !vy && !adjacency)) return TEB_AMD_OK;
  if (capacity_vertices < N) return fail(TEB_AMD_ERR_CAPACITY, "graph has more vertices than capacity_vertices");
  if (vx) HIPCHK(hipMemcpyAsync(vx, h->g_vx.p, N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (vy) HIPCHK(hipMemcpyAsync(vy, h->g_vy.p, N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (adjacency) HIPCHK(hipMemcpyAsync(adjacency, h->g_adj.p, (size_t)N * N, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return TEB_AMD_OK;
}

int teb_amd_compact_bands(teb_amd_handle_t* h, const int32_t* keep, int32_t best, int32_t* n_kept, int32_t* new_best) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!keep) return fail(TEB_AMD_ERR_INVALID_ARG, "null keep");
  const int B = h->B;
  if (new_best) *new_best = -1;
  if (n_kept) *n_kept = 0;
  if (B <= 0) return TEB_AMD_OK;
  if (!h->tmp_ready) {
    const size_t BS = (size_t)h->max_tebs * h->stride;
    bool ok = true;
    auto A = [&](hipError_t e) { if (e != hipSuccess) ok = false; };
    A(h->tmp_x.alloc(BS)); A(h->tmp_y.alloc(BS)); A(h->tmp_th.alloc(BS)); A(h->tmp_dt.alloc(BS)); A(h->tmp_n.alloc(h->max_tebs));
    A(h->tmp_vs.alloc(3 * (size_t)h->max_tebs)); A(h->tmp_vg.alloc(3 * (size_t)h->max_tebs)); A(h->tmp_i.alloc(9 * (size_t)h->max_tebs));
    A(h->tmp_chi2.alloc(h->max_tebs)); A(h->tmp_cost.alloc(h->max_tebs)); A(h->tmp_lambda.alloc(h->max_tebs));
    if (!h->cand_ready && ensure_candidate_buffers(h) != TEB_AMD_OK) ok = false;
    if (!ok) return fail(TEB_AMD_ERR_HIP, "compaction scratch allocation failed");
    h->tmp_ready = true;
  }
  std::vector<int> order(B), map;
  for (int b = 0; b < B; ++b) order[b] = b;
  const bool has_best = best >= 0 && best < B;
  if (has_best) std::swap(order[0], order[best]);   // std::iter_swap(tebs_.begin(), it_best_teb)
  for (int k = 0; k < B; ++k) if (keep[order[k]]) map.push_back(order[k]);
  const int K = (int)map.size();
  if (has_best && keep[best] && new_best) *new_best = 0;
  if (n_kept) *n_kept = K;
  bool identity = true;
  for (int k = 0; k < K; ++k) if (map[k] != k) identity = false;
  if (!identity && K > 0) {
    BatchDev src = batch_of(h), tmp = src;
    tmp.n = h->tmp_n.p; tmp.x = h->tmp_x.p; tmp.y = h->tmp_y.p; tmp.th = h->tmp_th.p; tmp.dt = h->tmp_dt.p;
    tmp.has_vs = h->tmp_i.p; tmp.has_vg = h->tmp_i.p + h->max_tebs; tmp.rotdir = h->tmp_i.p + 2 * (size_t)h->max_tebs;
    tmp.via_en = h->tmp_i.p + 3 * (size_t)h->max_tebs; tmp.status = h->tmp_i.p + 4 * (size_t)h->max_tebs;
    tmp.iters = h->tmp_i.p + 5 * (size_t)h->max_tebs; tmp.trials = h->tmp_i.p + 6 * (size_t)h->max_tebs;
    tmp.optimized = h->tmp_i.p + 7 * (size_t)h->max_tebs; tmp.last_iters = h->tmp_i.p + 8 * (size_t)h->max_tebs;
    tmp.vs = h->tmp_vs.p; tmp.vg = h->tmp_vg.p; tmp.chi2 = h->tmp_chi2.p; tmp.cost = h->tmp_cost.p; tmp.lambda = h->tmp_lambda.p;
    std::vector<int> ident(K);
    for (int k = 0; k < K; ++k) ident[k] = k;
    HIPCHK(hipMemcpyAsync(h->cand_map.p, map.data(), K * sizeof(int), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(move_bands_kernel, dim3(K), dim3(kThreads), 0, h->stream, src, tmp, h->cand_map.p, 0, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpyAsync(h->cand_map.p, ident.data(), K * sizeof(int), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(move_bands_kernel, dim3(K), dim3(kThreads), 0, h->stream, tmp, src, h->cand_map.p, 0, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  if (h->hsig_mode != 0 && h->hsig_B == B) {   // the kept bands' signatures stay valid: same rows, new order
    const int W = h->hsig_mode == 3 ? h->hsig_M : 2;
    std::vector<double> moved((size_t)K * (W > 0 ? W : 1), 0.0);
    for (int k = 0; k < K; ++k) std::copy(h->hsig_host.begin() + (size_t)map[k] * W, h->hsig_host.begin() + (size_t)(map[k] + 1) * W, moved.begin() + (size_t)k * W);
    h->hsig_host.swap(moved);
    h->hsig_B = K;
  } else {
    h->hsig_mode = 0;
  }
  h->B = K;
  h->consumers_valid = false; h->nmax_known = -1;
  return TEB_AMD_OK;
}

int teb_amd_filter_detours(teb_amd_handle_t* h, const teb_amd_hcp_params_t* p, int32_t best, int32_t* keep) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!p || !keep) return fail(TEB_AMD_ERR_INVALID_ARG, "null argument");
  const int B = h->B;
  if (B <= 0) return TEB_AMD_OK;
  int kept = 0;
  for (int b = 0; b < B; ++b) kept += keep[b] != 0;
  if (kept < 2 || best < 0 || best >= B || !keep[best]) return TEB_AMD_OK;   // "a moving direction wasn't chosen yet", :769-773
  if ((rc = ensure_candidate_buffers(h))) return rc;
  hipLaunchKernelGGL(detour_stats_kernel, dim3(B), dim3(kThreads), 0, h->stream, batch_of(h), p->length_start_orientation_vector, h->cand_sig.p);
  HIPCHK(hipGetLastError());
  std::vector<double> st((size_t)4 * B);
  std::vector<int> opt(B);
  HIPCHK(hipMemcpyAsync(st.data(), h->cand_sig.p, st.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(opt.data(), h->optimized.p, B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  auto found = [&](int b) { return st[4 * b] != 0; };
  auto orient = [&](int b) { return st[4 * b + 1]; };
  auto duration = [&](int b) { return st[4 * b + 2]; };
  auto poses = [&](int b) { return (int)st[4 * b + 3]; };
  if (poses(best) < 2) return TEB_AMD_OK;
  const double best_plan_duration = std::max(duration(best), 1.0);
  if (!found(best)) return TEB_AMD_OK;   // the plan is shorter than len_orientation_vector
  auto normalize_theta = [](double theta) {   // g2o::normalize_theta (misc.h)
    if (theta >= -M_PI && theta < M_PI) return theta;
    const double multiplier = std::floor(theta / (2 * M_PI));
    theta = theta - multiplier * 2 * M_PI;
    if (theta >= M_PI) theta -= 2 * M_PI;
    if (theta < -M_PI) theta += 2 * M_PI;
    return theta;
  };
  for (int b = 0; b < B; ++b) {
    if (!keep[b] || b == best) continue;
    if (poses(b) < 2 || !found(b)) { keep[b] = 0; continue; }
    if (std::fabs(normalize_theta(orient(b) - orient(best))) > p->detours_orientation_tolerance) { keep[b] = 0; continue; }
    if (!opt[b]) { keep[b] = 0; continue; }
    if (duration(b) / best_plan_duration > p->max_ratio_detours_duration_best_duration) { keep[b] = 0; continue; }
  }
  return TEB_AMD_OK;
}

int teb_amd_set_optimized_flags(teb_amd_handle_t* h, const int32_t* flags) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!flags) return fail(TEB_AMD_ERR_INVALID_ARG, "null flags");
  if (h->B <= 0) return TEB_AMD_OK;
  std::vector<int> f(h->B);
  for (int b = 0; b < h->B; ++b) f[b] = flags[b] != 0;
  HIPCHK(hipMemcpyAsync(h->optimized.p, f.data(), h->B * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return TEB_AMD_OK;
}

int teb_amd_get_optimized_flags(teb_amd_handle_t* h, int32_t* flags) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!flags) return fail(TEB_AMD_ERR_INVALID_ARG, "null flags");
  if (h->B <= 0) return TEB_AMD_OK;
  HIPCHK(hipMemcpyAsync(flags, h->optimized.p, h->B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return TEB_AMD_OK;
}

int teb_amd_get_band_flags(teb_amd_handle_t* h, int32_t* via_points_enabled, int32_t* has_vel_start, int32_t* has_vel_goal) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (h->B <= 0) return TEB_AMD_OK;
  if (via_points_enabled) HIPCHK(hipMemcpyAsync(via_points_enabled, h->via_en.p, h->B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  if (has_vel_start) HIPCHK(hipMemcpyAsync(has_vel_start, h->has_vs.p, h->B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  if (has_vel_goal) HIPCHK(hipMemcpyAsync(has_vel_goal, h->has_vg.p, h->B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return TEB_AMD_OK;
}

int teb_amd_device_state(teb_amd_handle_t* h, void** x, void** y, void** theta, void** dt, void** n, int32_t* stride) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (x) *x = h->x.p; if (y) *y = h->y.p; if (theta) *theta = h->th.p; if (dt) *dt = h->dt.p; if (n) *n = h->n.p;
  if (stride) *stride = h->stride;
  h->nmax_known = -1;        // the caller may write pose counts through `n`: the host's upper bound is void from here on (the kernel guards
  h->consumers_valid = false;   // itself against counts beyond its LDS strips as well)
  h->hsig_mode = 0;
  return TEB_AMD_OK;
}

int teb_amd_snapshot_state(teb_amd_handle_t* h) {
  int rc = check_handle(h);
  if (rc) return rc;
  if ((rc = copy_strips(h, Strips{h->snap_x.p, h->snap_y.p, h->snap_th.p, h->snap_dt.p, h->snap_n.p}, Strips{h->x.p, h->y.p, h->th.p, h->dt.p, h->n.p}, h->max_tebs))) return rc;
  h->snap_nmax = h->nmax_known;
  h->snap_B = h->B;
  return TEB_AMD_OK;
}

int teb_amd_restore_state(teb_amd_handle_t* h) {
  int rc = check_handle(h);
  if (rc) return rc;
  if ((rc = copy_strips(h, Strips{h->x.p, h->y.p, h->th.p, h->dt.p, h->n.p}, Strips{h->snap_x.p, h->snap_y.p, h->snap_th.p, h->snap_dt.p, h->snap_n.p}, h->max_tebs))) return rc;
  // snap_nmax bounded the first snap_B bands only; the copy brings back the counts of ALL max_tebs slots, so with another batch size the
  // bound says nothing about the bands beyond the old B
  h->consumers_valid = false; h->nmax_known = (h->B == h->snap_B) ? h->snap_nmax : -1;
  h->hsig_mode = 0;   // the bands change: signatures of an earlier teb_amd_compute_h_signatures call are stale
  return TEB_AMD_OK;
}

int teb_amd_last_launch_info(teb_amd_handle_t* h, int32_t* distance_helpers_per_band, int32_t* solver_helpers_per_band, int32_t* repeated_single_cu) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (distance_helpers_per_band) *distance_helpers_per_band = h->mcu_last_helpers;
  if (solver_helpers_per_band) *solver_helpers_per_band = h->mcu_last_solvers;
  if (repeated_single_cu) *repeated_single_cu = h->mcu_last_repeated;
  return TEB_AMD_OK;
}

int teb_amd_multi_cu_backoff(teb_amd_handle_t* h, int32_t* launches_paused, int32_t* pause_length) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (launches_paused) *launches_paused = h->mcu_backoff_left;
  if (pause_length) *pause_length = h->mcu_backoff_len;
  return TEB_AMD_OK;
}

int teb_amd_last_kernel_ms(teb_amd_handle_t* h, float* ms) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!ms || !h->timed) return fail(TEB_AMD_ERR_INVALID_ARG, "no kernel has been launched yet");
  HIPCHK(hipEventSynchronize(h->ev1));
  HIPCHK(hipEventElapsedTime(ms, h->ev0, h->ev1));
  return TEB_AMD_OK;
}

int teb_amd_capacity(teb_amd_handle_t* h, int32_t* lds_bytes, int32_t* max_poses_supported) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (lds_bytes) *lds_bytes = (int32_t)h->plan.total_bytes;
  if (max_poses_supported) {
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, h->device));
    int S = kThreads * kPoseIterBandHbm;
    while (S > 2 && lds_bytes_for(S, SOLVER_BANDG, h->lds_limit) > h->lds_limit) --S;   // band in HBM: four poses per lane, as many as its LDS strips hold
    *max_poses_supported = S;
  }
  return TEB_AMD_OK;
}

// ---- test hooks (include/teb_amd_debug.h) -------------------------------------------------------------------
int teb_amd_debug_linearize(teb_amd_handle_t* h, int32_t b, double weight_multiplier, double* H_dense, double* bvec,
                            double* chi2, int32_t* assoc_pose, int32_t* assoc_obst, int32_t assoc_cap, int32_t* assoc_count) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (b < 0 || b >= h->B) return fail(TEB_AMD_ERR_INVALID_ARG, "TEB index out of range");
  // run the kernel in debug mode on TEB b only: temporarily view the batch as starting at b
  OptArgs a;
  std::memset(&a, 0, sizeof a);
  a.inner = 1; a.outer = 1; a.no_near_cache = h->opt.no_near_cache != 0; a.debug_linearize = 1; a.debug_weight_multiplier = weight_multiplier; a.band_ldlt = h->solver == SOLVER_BANDG ? 0 : h->band_ldlt; a.Hband = h->Hband.p; a.hband_stride = h->hband_stride;
  a.dbg_H = h->dbg_H.p; a.dbg_b = h->dbg_b.p; a.dbg_chi2 = h->dbg_chi2.p;
  SceneDev sc = scene_of(h);
  BatchDev bt = batch_of(h);
  const size_t so = (size_t)b * h->stride;
  bt.B = 1;
  bt.n += b; bt.x += so; bt.y += so; bt.th += so; bt.dt += so;
  bt.has_vs += b; bt.vs += 3 * b; bt.has_vg += b; bt.vg += 3 * b; bt.rotdir += b; bt.via_en += b;
  bt.status += b; bt.iters += b; bt.last_iters += b; bt.trials += b; bt.chi2 += b; bt.cost += b; bt.lambda += b;
  bt.assoc_cnt += so; bt.assoc += (size_t)b * bt.assoc_cap * h->stride; bt.assoc_overflow += b;
  bt.via_pose += (size_t)b * bt.via_cap; bt.Hbackup += (size_t)b * h->hmat_stride;
  // results of TEB b must not be clobbered by the debug run: save and restore them
  int sv_i[3]; double sv_d[3];
  HIPCHK(hipMemcpy(&sv_i[0], h->status.p + b, sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&sv_i[1], h->iters.p + b, sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&sv_i[2], h->trials.p + b, sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&sv_d[0], h->chi2.p + b, sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&sv_d[1], h->cost.p + b, sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&sv_d[2], h->lambda.p + b, sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(launch_opt(h, 1, sc, bt, a));
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(h->status.p + b, &sv_i[0], sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->iters.p + b, &sv_i[1], sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->trials.p + b, &sv_i[2], sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->chi2.p + b, &sv_d[0], sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->cost.p + b, &sv_d[1], sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->lambda.p + b, &sv_d[2], sizeof(double), hipMemcpyHostToDevice));
  int n = 0;
  HIPCHK(hipMemcpy(&n, h->n.p + b, sizeof(int), hipMemcpyDeviceToHost));
  const int Nt = 4 * n;
  std::vector<double> Hb((size_t)Nt * kBand), bv(Nt);
  HIPCHK(hipMemcpy(Hb.data(), h->dbg_H.p, Hb.size() * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(bv.data(), h->dbg_b.p, bv.size() * sizeof(double), hipMemcpyDeviceToHost));
  if (chi2) HIPCHK(hipMemcpy(chi2, h->dbg_chi2.p, 4 * sizeof(double), hipMemcpyDeviceToHost));
  if (H_dense) {
    std::fill(H_dense, H_dense + (size_t)Nt * Nt, 0.0);
    for (int r = 0; r < Nt; ++r) {
      bool fr = (r < 3) || (r >= 4 * (n - 1));
      for (int d = 0; d < kBand && d <= r; ++d) {
        double v = fr ? 0.0 : Hb[(size_t)r * kBand + d];   // identity rows of fixed variables are exported as zeros
        H_dense[(size_t)r * Nt + (r - d)] = v;
        H_dense[(size_t)(r - d) * Nt + r] = v;
      }
    }
  }
  if (bvec) for (int r = 0; r < Nt; ++r) bvec[r] = bv[r];
  if (assoc_count) {
    std::vector<int> cnt(n), lst((size_t)bt.assoc_cap * h->stride);
    HIPCHK(hipMemcpy(cnt.data(), h->assoc_cnt.p + so, n * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(lst.data(), h->assoc.p + (size_t)b * bt.assoc_cap * h->stride, lst.size() * sizeof(int), hipMemcpyDeviceToHost));
    int k = 0;
    for (int i = 1; i < n - 1; ++i)
      for (int q = 0; q < cnt[i]; ++q) {
        const int ent = lst[(size_t)q * h->stride + i];
        for (int rep = (ent & kAssocTriple) ? 3 : 1; rep > 0; --rep) {
          if (k < assoc_cap) { if (assoc_pose) assoc_pose[k] = i; if (assoc_obst) assoc_obst[k] = h->host_static[ent & kAssocMask]; }
          ++k;
        }
      }
    *assoc_count = k;
  }
  return TEB_AMD_OK;
}

int teb_amd_debug_distance(teb_amd_handle_t* h, int32_t nq, const int32_t* obst_index, const double* x, const double* y,
                           const double* theta, const int32_t* spatio_temporal, const double* t, double* dist, double* grad) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (nq <= 0) return fail(TEB_AMD_ERR_INVALID_ARG, "nq <= 0");
  for (int q = 0; q < nq; ++q) if (obst_index[q] < 0 || obst_index[q] >= h->M) return fail(TEB_AMD_ERR_INVALID_ARG, "obstacle index out of range");
  DevBuf<int> d_oi, d_st; DevBuf<double> d_x, d_y, d_th, d_t, d_dist, d_grad;
  HIPCHK(d_oi.alloc(nq)); HIPCHK(d_st.alloc(nq)); HIPCHK(d_x.alloc(nq)); HIPCHK(d_y.alloc(nq)); HIPCHK(d_th.alloc(nq));
  HIPCHK(d_t.alloc(nq)); HIPCHK(d_dist.alloc(nq)); HIPCHK(d_grad.alloc(3 * (size_t)nq));
  HIPCHK(hipMemcpy(d_oi.p, obst_index, nq * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_st.p, spatio_temporal, nq * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_x.p, x, nq * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_y.p, y, nq * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_th.p, theta, nq * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_t.p, t, nq * sizeof(double), hipMemcpyHostToDevice));
  SceneDev sc = scene_of(h);
  hipLaunchKernelGGL(distance_kernel, dim3((nq + 255) / 256), dim3(256), 0, h->stream, h->cfg, sc, nq, d_oi.p, d_x.p, d_y.p, d_th.p,
                     d_st.p, d_t.p, d_dist.p, d_grad.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(dist, d_dist.p, nq * sizeof(double), hipMemcpyDeviceToHost));
  if (grad) HIPCHK(hipMemcpy(grad, d_grad.p, 3 * (size_t)nq * sizeof(double), hipMemcpyDeviceToHost));
  d_oi.free(); d_st.free(); d_x.free(); d_y.free(); d_th.free(); d_t.free(); d_dist.free(); d_grad.free();
  return TEB_AMD_OK;
}

int teb_amd_debug_profile(teb_amd_handle_t* h, double* cycles8) {
  int rc = check_handle(h);
  if (rc) return rc;
#ifdef TEB_PROFILE
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(cycles8, h->dbg_H.p, 8 * sizeof(double), hipMemcpyDeviceToHost));
  long long crp[8], crw[32];
  HIPCHK(hipMemcpyFromSymbol(crp, HIP_SYMBOL(tebamd::g_cr_prof), sizeof crp));
  HIPCHK(hipMemcpyFromSymbol(crw, HIP_SYMBOL(tebamd::g_crw_prof), sizeof crw));
  for (int w = 0; w < 4; ++w)
    fprintf(stderr, "[cr_forward rounds of %2d-lane groups, workgroup 0 thread 0, cumulative: load+factor | loads+solve3 | Schur products | writes 1 | barrier 1 | writes 2 + barrier 2 | rounds] %lld %lld %lld %lld %lld %lld | %lld\n",
            8 << w, crw[w * 8 + 0], crw[w * 8 + 1], crw[w * 8 + 2], crw[w * 8 + 3], crw[w * 8 + 4], crw[w * 8 + 5], crw[w * 8 + 6]);
  fprintf(stderr, "[cr_solve cycles, workgroup 0, cumulative; hybrid solve: init compact | level 0 | compact forward | top + backward | odd rows] %lld %lld %lld %lld %lld %lld\n",
          crp[0], crp[1], crp[2], crp[3], crp[4], crp[5]);
  unsigned long long ast[4];
  HIPCHK(hipMemcpyFromSymbol(ast, HIP_SYMBOL(tebamd::g_assoc_stats), sizeof ast));
  fprintf(stderr, "[association of generic shapes, all workgroups, cumulative] %llu candidates after the far-field cull, %llu exact distances\n", ast[0], ast[1]);
  unsigned long long ars[4];
  HIPCHK(hipMemcpyFromSymbol(ars, HIP_SYMBOL(tebamd::g_ar_steps), sizeof ars));
  fprintf(stderr, "[autoResize rule machine, workgroup 0, cumulative] %llu calls, %llu steps, %llu cycles\n", ars[2], ars[1], ars[3]);
  unsigned long long nq[2];
  HIPCHK(hipMemcpyFromSymbol(&nq[0], HIP_SYMBOL(tebamd::g_near_recomputed), sizeof nq[0]));
  HIPCHK(hipMemcpyFromSymbol(&nq[1], HIP_SYMBOL(tebamd::g_near_queries), sizeof nq[1]));
  fprintf(stderr, "[near masks of the dynamic-obstacle edges, all workgroups, cumulative] recomputed by %llu of %llu lane passes\n", nq[0], nq[1]);
  long long lnp[16];
  HIPCHK(hipMemcpyFromSymbol(lnp, HIP_SYMBOL(tebamd::g_lin_prof), sizeof lnp));
  fprintf(stderr, "[linearize cycles, thread 0 of workgroup 0, cumulative] zero H, b %lld | trig + barrier %lld | near masks %lld | edges %lld | slice reduction %lld | scatter (3 phases) %lld | tail + chi2 sum %lld\n",
          lnp[0], lnp[1], lnp[2], lnp[3], lnp[4], lnp[5], lnp[6]);
  fprintf(stderr, "[graph side data cycles, thread 0 of workgroup 0, cumulative] trig %lld | association %lld | time stamps (one lane) %lld | via-points + barrier %lld\n",
          lnp[8], lnp[9], lnp[10], lnp[11]);
  long long evp[8];
  HIPCHK(hipMemcpyFromSymbol(evp, HIP_SYMBOL(tebamd::g_ev_prof), sizeof evp));
  fprintf(stderr, "[eval_index cycles, thread 0 of workgroup 0, cumulative] evaluate: static %lld dynamic %lld chain %lld | linearise: static %lld dynamic %lld chain %lld\n",
          evp[0], evp[1], evp[2], evp[3], evp[4], evp[5]);
  return TEB_AMD_OK;
#else
  (void)cycles8;
  return fail(TEB_AMD_ERR_UNSUPPORTED, "library built without -DTEB_PROFILE");
#endif
}

int teb_amd_debug_profile_bands(teb_amd_handle_t* h, double* cycles_per_band) {
  int rc = check_handle(h);
  if (rc) return rc;
#ifdef TEB_PROFILE
  HIPCHK(hipStreamSynchronize(h->stream));
  if ((size_t)16 + h->B > h->dbg_H.n) return fail(TEB_AMD_ERR_CAPACITY, "debug buffer too small");
  HIPCHK(hipMemcpy(cycles_per_band, h->dbg_H.p + 16, h->B * sizeof(double), hipMemcpyDeviceToHost));
  return TEB_AMD_OK;
#else
  (void)cycles_per_band;
  return fail(TEB_AMD_ERR_UNSUPPORTED, "library built without -DTEB_PROFILE");
#endif
}

// debug: checks the operand maps of v_mfma_f64_16x16x4_f64 that cr_forward_mfma relies on (C = A B, A 16x8 row-major, B 8x16, C 16x16)
// and times reps x 4 independent issues on one wave. Builds without -DTEB_AMD_MFMA_SCHUR return TEB_AMD_ERR_UNSUPPORTED.
int teb_amd_debug_mfma_selftest(teb_amd_handle_t* h, const double* A, const double* B, double* C, int32_t reps, double* cycles_per_mfma) {
  int rc = check_handle(h);
  if (rc) return rc;
#ifdef TEB_AMD_MFMA_SCHUR
  if (!A || !B || !C) return fail(TEB_AMD_ERR_INVALID_ARG, "null matrices");
  double *dA = nullptr, *dB = nullptr, *dC = nullptr; long long* dT = nullptr;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&dA), 128 * sizeof(double))); HIPCHK(hipMalloc(reinterpret_cast<void**>(&dB), 128 * sizeof(double)));
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&dC), 256 * sizeof(double))); HIPCHK(hipMalloc(reinterpret_cast<void**>(&dT), 2 * sizeof(long long)));
  HIPCHK(hipMemcpyAsync(dA, A, 128 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(dB, B, 128 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, h->stream, dA, dB, dC, reps, dT);
  HIPCHK(hipGetLastError());
  long long t[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(C, dC, 256 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(t, dT, sizeof(t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (cycles_per_mfma) *cycles_per_mfma = reps > 0 ? (double)t[0] / (4.0 * reps) : 0.0;
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dT);
  return TEB_AMD_OK;
#else
  (void)A; (void)B; (void)C; (void)reps; (void)cycles_per_mfma;
  return fail(TEB_AMD_ERR_UNSUPPORTED, "this build has no MFMA Schur update (-DTEB_AMD_MFMA_SCHUR)");
#endif
}

int teb_amd_debug_stream(teb_amd_handle_t* h, int64_t n_doubles, int32_t repeats) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (n_doubles <= 0 || repeats <= 0) return fail(TEB_AMD_ERR_INVALID_ARG, "bad stream size");
  DevBuf<double> a, b;
  HIPCHK(a.alloc((size_t)n_doubles)); HIPCHK(b.alloc((size_t)n_doubles));
  HIPCHK(hipMemsetAsync(a.p, 0, sizeof(double) * (size_t)n_doubles, h->stream));
  for (int r = 0; r < repeats; ++r) {
    hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, h->stream, a.p, b.p, (size_t)n_doubles);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  a.free(); b.free();
  return TEB_AMD_OK;
}

// test hook: overwrites the resident pose count of band b behind the host's back (no range check, the host's cached upper bound of the
// counts is NOT invalidated) - what a careless writer through teb_amd_device_state's `n` pointer could do. The kernel has to survive it.
int teb_amd_debug_poke_pose_count(teb_amd_handle_t* h, int32_t b, int32_t n) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (b < 0 || b >= h->max_tebs) return fail(TEB_AMD_ERR_INVALID_ARG, "band index out of range");
  HIPCHK(hipMemcpyAsync(h->n.p + b, &n, sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return TEB_AMD_OK;
}

// diagnostic of the multi-CU mode: from now on every multi-CU launch leaves host-visible breadcrumbs, and one that is still running after
// `milliseconds` prints them and ENDS THE PROCESS (exit code 3). Development tool; 0 switches the watchdog off again.
int teb_amd_debug_mcu_watchdog(teb_amd_handle_t* h, int32_t milliseconds) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (milliseconds > 0 && !h->mcu_trace) HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h->mcu_trace), 1024 * sizeof(unsigned), hipHostMallocMapped));
  h->mcu_watchdog_ms = milliseconds;
  return TEB_AMD_OK;
}

// diagnostic of the multi-CU mode: 1 = the association stays with the band's own workgroup, 2 = the distances do (bisecting a difference)
int teb_amd_last_shader_clock_mhz(teb_amd_handle_t* h, double* mhz) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!mhz || !h->timed) return fail(TEB_AMD_ERR_INVALID_ARG, "no kernel has been launched yet");
  long long c[4];
  HIPCHK(hipMemcpyAsync(c, h->clk.p, sizeof(c), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  const double ticks = (double)(c[3] - c[1]);   // 100 MHz real-time counter
  *mhz = ticks > 0 ? (double)(c[2] - c[0]) / ticks * 100.0 : 0.0;
  return TEB_AMD_OK;
}

int teb_amd_debug_rtc_compile(uint64_t flag_values, int32_t solver, int32_t jacobian_mode, int32_t scene_kind, int64_t* code_bytes) {
  if (solver < 0 || solver > 2 || jacobian_mode < 0 || jacobian_mode > 1 || scene_kind < SCENE_POINTS_CUSTOM || scene_kind > SCENE_GENERIC_SMALL_CUSTOM)
    return fail(TEB_AMD_ERR_INVALID_ARG, "teb_amd_debug_rtc_compile: layout 0 .. 2, Jacobian mode 0 / 1, scene kind 12 .. 15");
  RtcKey key;
  key.flags = flag_values; key.solver = solver; key.jmode = jacobian_mode; key.scene = scene_kind;
  std::string why;
  std::shared_ptr<RtcKernel> rk = rtc_request(key, true, &why);
  if (!rk) return fail(TEB_AMD_ERR_UNSUPPORTED, "run-time compilation unavailable: " + why);
  if (rk->state.load() != RtcKernel::READY) return fail(TEB_AMD_ERR_HIP, rk->log.substr(0, 1500));
  if (code_bytes) *code_bytes = (int64_t)rk->code.size();
  return TEB_AMD_OK;
}

#ifndef TEB_AMD_BUILD_SOURCE_HASH
#define TEB_AMD_BUILD_SOURCE_HASH ""   // (a build that did not go through build.py)
#endif
#ifndef TEB_AMD_BUILD_KERNEL_HASH
#define TEB_AMD_BUILD_KERNEL_HASH ""
#endif
int teb_amd_debug_build_info(char* kernel_hash, int32_t kernel_hash_capacity, char* source_hash, int32_t source_hash_capacity, char* variant_defines,
                             int32_t defines_capacity, int32_t* threads_per_workgroup) {
  if (kernel_hash && kernel_hash_capacity > 0) std::snprintf(kernel_hash, (size_t)kernel_hash_capacity, "%s", TEB_AMD_BUILD_KERNEL_HASH);
  if (source_hash && source_hash_capacity > 0) std::snprintf(source_hash, (size_t)source_hash_capacity, "%s", TEB_AMD_BUILD_SOURCE_HASH);
  if (variant_defines && defines_capacity > 0) std::snprintf(variant_defines, (size_t)defines_capacity, "%s", TEB_AMD_VARIANT_DEFINES);
  if (threads_per_workgroup) *threads_per_workgroup = kThreads;
  return TEB_AMD_OK;
}

int teb_amd_debug_rtc_join(void) {   // waits for the background compilations of this process (teb_amd_options_t::compile_for_config = 1) to finish
  rtc_cache().join_workers();
  return TEB_AMD_OK;
}

int teb_amd_debug_rtc_cache(int32_t* embedded, int32_t* disk_hits, int32_t* disk_writes, char* cache_dir, int32_t capacity) {
  RtcCache& c = rtc_cache();
  std::lock_guard<std::mutex> lock(c.mu);
  if (embedded) *embedded = c.embedded ? 1 : 0;
  if (disk_hits) *disk_hits = c.disk_hits.load();
  if (disk_writes) *disk_writes = c.disk_writes.load();
  if (cache_dir && capacity > 0) std::snprintf(cache_dir, (size_t)capacity, "%s", c.disk_dir.c_str());
  return TEB_AMD_OK;
}

int teb_amd_debug_rtc_stats(int32_t* ready, int32_t* compiling, int32_t* failed, double* last_compile_seconds, char* last_error, int32_t capacity) {
  RtcCache& c = rtc_cache();
  std::lock_guard<std::mutex> lock(c.mu);
  int r = 0, w = 0, f = 0;
  double secs = 0;
  std::string err;
  for (auto& kv : c.kernels) {
    const int st = kv.second->state.load();
    if (st == RtcKernel::READY) { ++r; secs = kv.second->compile_seconds; }
    else if (st == RtcKernel::COMPILING) ++w;
    else { ++f; err = kv.second->log; }
  }
  if (ready) *ready = r;
  if (compiling) *compiling = w;
  if (failed) *failed = f;
  if (last_compile_seconds) *last_compile_seconds = secs;
  if (last_error && capacity > 0) { std::snprintf(last_error, (size_t)capacity, "%s", err.c_str()); }
  return TEB_AMD_OK;
}

int teb_amd_debug_last_instantiation(teb_amd_handle_t* h, int32_t* solver, int32_t* jacobian_mode, int32_t* scene_kind) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (solver) *solver = h->last_inst[0];
  if (jacobian_mode) *jacobian_mode = h->last_inst[1];
  if (scene_kind) *scene_kind = h->last_inst[2];
  return TEB_AMD_OK;
}

int teb_amd_debug_last_config_profile(teb_amd_handle_t* h, int32_t* defaults_profile) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!defaults_profile) return fail(TEB_AMD_ERR_INVALID_ARG, "null output");
  *defaults_profile = h->last_defaults_profile;
  return TEB_AMD_OK;
}

int teb_amd_debug_mcu_flags(teb_amd_handle_t* h, int32_t flags) {
  int rc = check_handle(h);
  if (rc) return rc;
  h->mcu_debug_flags = flags;
  return TEB_AMD_OK;
}


int teb_amd_debug_assoc_overflow(teb_amd_handle_t* h, int32_t* flags) {
  int rc = check_handle(h);
  if (rc) return rc;
  HIPCHK(hipMemcpy(flags, h->assoc_ovf.p, h->B * sizeof(int), hipMemcpyDeviceToHost));
  return TEB_AMD_OK;
}

}  // extern "C"
