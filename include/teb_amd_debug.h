/*
 * teb_amd_debug.h — test hooks of libteb_amd.so. NOT part of the drop-in boundary; used by tests/ to
 * compare intermediate quantities (normal equations, association lists, distances) with the oracle.
 */
#ifndef TEB_AMD_DEBUG_H_
#define TEB_AMD_DEBUG_H_
#include "teb_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

/* sizeof() of the POD structs as compiled into the library (ABI drift check for the ctypes mirror). */
int teb_amd_sizeof_config(void);
int teb_amd_sizeof_obstacles(void);
int teb_amd_sizeof_teb_batch(void);
int teb_amd_sizeof_results(void);
int teb_amd_sizeof_options(void);

/*
 * Build the cost graph of resident TEB b with the given weight_multiplier and linearise once (no
 * autoResize, state untouched). Outputs in the canonical index space var(i,c) = 4*i + c:
 *   H_dense [4n*4n] row-major, bvec [4n], chi2[4] = {obstacle-type, via-point, time-optimal, other};
 *   the (pose, obstacle) association pairs in edge order (up to assoc_cap) and their number.
 * Any output pointer may be NULL.
 */
int teb_amd_debug_linearize(teb_amd_handle_t* h, int32_t b, double weight_multiplier, double* H_dense,
                            double* bvec, double* chi2, int32_t* assoc_pose, int32_t* assoc_obst,
                            int32_t assoc_cap, int32_t* assoc_count);

/* footprint <-> obstacle distance and gradient (d/dx, d/dy, d/dtheta) for nq queries, on the GPU. */
int teb_amd_debug_distance(teb_amd_handle_t* h, int32_t nq, const int32_t* obst_index, const double* x,
                           const double* y, const double* theta, const int32_t* spatio_temporal,
                           const double* t, double* dist, double* grad);

/* phase cycle counters of workgroup 0 of the last optimize_batch (only in a -DTEB_PROFILE build):
 * [0] autoResize [1] association+via+time stamps [2] linearise [3] H backup [4] damped solve
 * [5] update+chi2 evaluation [6] accept/reject (+H restore) */
int teb_amd_debug_profile(teb_amd_handle_t* h, double* cycles8);
/* -DTEB_PROFILE builds: clock64() cycles of every band's workgroup in the last optimize_batch launch, [B] */
int teb_amd_debug_profile_bands(teb_amd_handle_t* h, double* cycles_per_band);

/* Streams n_doubles fp64 values global->global (8 B per lane, coalesced; reads and writes n_doubles*8 bytes each)
 * `repeats` times: a known byte count to calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE against. */
int teb_amd_debug_stream(teb_amd_handle_t* h, int64_t n_doubles, int32_t repeats);
/* Operand maps of v_mfma_f64_16x16x4_f64 as used by the MFMA Schur update (builds with -DTEB_AMD_MFMA_SCHUR; others return
 * TEB_AMD_ERR_UNSUPPORTED): C [16x16] = A [16x8, row-major] * B [8x16]; *cycles_per_mfma = clock ticks per issue on one wave. */
int teb_amd_debug_mfma_selftest(teb_amd_handle_t* h, const double* A, const double* B, double* C, int32_t reps, double* cycles_per_mfma);

/* The host-side reduction of teb_amd_select_best_distributed on its own (no GPU, no communicator): `records` = world x (cost, global
 * index as double; index < 0 = that rank holds no candidate), as the all-gather delivers them. Lowest cost wins, ties go to the lowest
 * global index (strict '<' of selectBestTeb, src/homotopy_class_planner.cpp:610). best_global = -1 when no rank has a candidate. */
int teb_amd_debug_world_argmin(const double* records, int32_t world, int32_t* best_global, double* best_cost, int32_t* owner_rank);

/* Overwrites the resident pose count of band b WITHOUT any check and without telling the host side (its cached upper bound of the
 * counts stays as it was): reproduces a careless write through teb_amd_device_state's `n` pointer. The optimise kernel must refuse such
 * a band (status FAILED, flag bit 1) instead of running off its LDS strips. */
int teb_amd_debug_poke_pose_count(teb_amd_handle_t* h, int32_t b, int32_t n);

/* Development aid of the multi-CU mode: every multi-CU launch leaves host-visible breadcrumbs (one word per workgroup), and a launch that
 * is still running after `milliseconds` prints them to stderr and ENDS THE PROCESS with exit code 3 (a kernel cannot be cancelled).
 * 0 = off. */
int teb_amd_debug_mcu_watchdog(teb_amd_handle_t* h, int32_t milliseconds);

/* Development aid of the multi-CU mode: bit 0 = the association scan stays with the band's own workgroup, bit 1 = the distance records do
 * (to bisect a difference between the two modes). */
int teb_amd_debug_mcu_flags(teb_amd_handle_t* h, int32_t flags);

/* Which kernel instantiation the last optimise launch ran: 1 = specialised on the TebConfig defaults (every flag of the profile table of
 * csrc/teb_device.hpp folded at compile time), 2 = the same folds except the via-points and the holonomic choice of the velocity /
 * acceleration edges (point-like scenes), 3 = every cost-term flag at run time, only the never-reached bulk folded (point-like scenes),
 * 4 = compiled at run time for this configuration (teb_amd_options_t::compile_for_config), 0 = the generic one (teb_amd_options_t::generic_config_path forces it). */
int teb_amd_debug_last_config_profile(teb_amd_handle_t* h, int32_t* defaults_profile);
/* The instantiation teb_optimize_kernel<layout, Jacobian mode, scene kind> the last optimise launch ran: layout 0 band in LDS, 1 blocks in
 * LDS, 2 band in HBM; scene kind as numbered in csrc/teb_opt_launch.hpp (0 .. 11 pre-built, 12 .. 15 compiled at run time). -1 before the
 * first launch. tests/test_gpu_every_instantiation.py launches every pre-built one and checks with this that it did. */
int teb_amd_debug_last_instantiation(teb_amd_handle_t* h, int32_t* solver, int32_t* jacobian_mode, int32_t* scene_kind);
/* Run-time compiled instantiations of this process (teb_amd_options_t::compile_for_config): how many are ready / still compiling /
 * failed, the compile time of the last one that finished [s], and the reason of the last failure (empty string if none). */
int teb_amd_debug_rtc_stats(int32_t* ready, int32_t* compiling, int32_t* failed, double* last_compile_seconds, char* last_error, int32_t capacity);
/* Blocks until every background compilation of this process (compile_for_config = 1) has finished. The library does the same when it is
 * unloaded / at exit - a compiler thread must not outlive the library - which can hold a process for the rest of a hiprtcCompileProgram
 * call (seconds); a host that cares calls this at a time of its choosing. */
int teb_amd_debug_rtc_join(void);
/* Compiles (and waits for) the run-time instantiation for the given flag values - bit i = value of flag i of TEB_PF_ALL
 * (csrc/teb_device.hpp) -, layout (0 band in LDS, 1 blocks in LDS, 2 band in HBM), Jacobian mode and scene kind (12 .. 15: the *_CUSTOM
 * kinds). Needs no GPU: hipRTC cross-compiles for gfx950, so the CPU test stage covers the run-time compilation path. Returns
 * TEB_AMD_OK iff the code object was produced; *code_bytes = its size. */
int teb_amd_debug_rtc_compile(uint64_t flag_values, int32_t solver, int32_t jacobian_mode, int32_t scene_kind, int64_t* code_bytes);
/* Where the run-time compiler of this process takes its sources and keeps its code objects: *embedded = 1 iff it compiles the copy of the
 * kernel sources inside the library (0: a csrc directory), *disk_hits / *disk_writes = code objects loaded from / written to the disk
 * cache so far, cache_dir = its directory (empty string: disabled). Valid after the first request (teb_amd_debug_rtc_compile or a launch
 * with compile_for_config). */
int teb_amd_debug_rtc_cache(int32_t* embedded, int32_t* disk_hits, int32_t* disk_writes, char* cache_dir, int32_t capacity);

/* What THIS BINARY was built from, as build.py recorded it when it compiled the host translation unit (empty strings for a build that
 * did not go through build.py): kernel_hash = sha256 prefix over the optimise kernel's translation unit with comments and white space
 * stripped + the per-unit compiler flags (teb_local_planner_amd/build.py: kernel_hash(); what bench.py calls source_hash and what a
 * committed profiles/ summary is tied to), source_hash = the same over every device + host source as it stands (build.py: source_hash()),
 * variant_defines = the -D flags of the build variant ("" for the product), *threads_per_workgroup = the optimise kernel's workgroup
 * size. bench.py reports kernel_hash as config.binary_hash and attaches a profiles/ summary only when summary, source tree and binary
 * carry the same hash. Needs no handle and no GPU. */
int teb_amd_debug_build_info(char* kernel_hash, int32_t kernel_hash_capacity, char* source_hash, int32_t source_hash_capacity, char* variant_defines,
                             int32_t defines_capacity, int32_t* threads_per_workgroup);

/*
 * Phase split of the PRODUCT kernel, opt-in (one scalar branch per phase boundary when off; ~ 14 clock reads per LM iteration and band
 * when on, < 1 % of the launch): shader cycles of every band's workgroup per phase of the last teb_amd_optimize_batch -
 * TEB_AMD_PHASE_LOG_SLOTS doubles per band: [0] autoResize (src/timed_elastic_band.cpp:227-286), [1] buildGraph side data (obstacle
 * association, via-point attachment, time stamps of the dynamic edges; src/optimal_planner.cpp:444-718), [2] linearisation (buildSystem),
 * [3] backup of H for rejected trials (blocks layout), [4] the damped solves, [5] update + computeActiveErrors of the trials, [6] accept /
 * restore, [7] spare, [8] the workgroup from entry to exit. teb_amd_get_phase_log synchronises; *bands = min(count, capacity_bands).
 */
#define TEB_AMD_PHASE_LOG_SLOTS 9
int teb_amd_set_phase_log(teb_amd_handle_t* h, int32_t enable);
int teb_amd_get_phase_log(teb_amd_handle_t* h, double* cycles, int32_t capacity_bands, int32_t* bands);

/* per-TEB flags of the last launch: bit0 association list overflow, bit1 autoResize capacity overflow */
int teb_amd_debug_assoc_overflow(teb_amd_handle_t* h, int32_t* flags);

#ifdef __cplusplus
}
#endif
#endif
