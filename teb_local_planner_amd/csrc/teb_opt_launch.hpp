// teb_opt_launch.hpp — how the host side of libteb_amd.so reaches the instantiations of teb_optimize_kernel.
//
// Every (solver layout, Jacobian mode, scene kind) instantiation is ~ 0.5 MB of code and 20 - 50 s of compile time; each one is built
// in a translation unit of its own (teb_opt_inst.hip with -DTEB_INST_SOLVER / _JMODE / _SCENE), in parallel, and hands the host side
// the address of its kernel through one hidden function. -DTEB_AMD_SINGLE_TU (the tools/ builds with profiling counters, which read
// __device__ symbols of the kernel's translation unit) defines them all inside teb_amd.hip instead.
#pragma once

#define TEB_OPT_CAT_(a, b, c, d) a##_##b##_##c##_##d
#define TEB_OPT_CAT(a, b, c, d) TEB_OPT_CAT_(a, b, c, d)
#define TEB_OPT_KERNEL_FN(S, J, P) TEB_OPT_CAT(teb_opt_kernel, S, J, P)
#define TEB_OPT_DECLARE(S, J, P) __attribute__((visibility("hidden"))) const void* TEB_OPT_KERNEL_FN(S, J, P)();
#define TEB_OPT_DEFINE(S, J, P)                                                                                      \
  __attribute__((visibility("hidden"))) const void* TEB_OPT_KERNEL_FN(S, J, P)() {                                   \
    return reinterpret_cast<const void*>(&tebamd::teb_optimize_kernel<S, J, P>);                                     \
  }
// solver: 0 SOLVER_BAND, 1 SOLVER_CR, 2 SOLVER_BANDG; jmode: 0 analytic, 1 g2o numeric; scene: 0 SCENE_POINTS, 1 SCENE_GENERIC,
// 2 SCENE_POINTS_SMALL, 3 SCENE_GENERIC_SMALL (small batches with helper workgroups, closed-form Jacobians only)
// 4 SCENE_POINTS_DEFAULTS, 5 SCENE_POINTS_SMALL_DEFAULTS, 6 SCENE_GENERIC_DEFAULTS, 7 SCENE_GENERIC_SMALL_DEFAULTS: the kinds with the configuration flags folded to the TebConfig defaults
// 8 SCENE_POINTS_WIDE, 9 SCENE_POINTS_SMALL_WIDE: the point-like kinds with every fold but the via-points and the holonomic choice
// 10 SCENE_POINTS_LIGHT, 11 SCENE_POINTS_SMALL_LIGHT: the point-like kinds with every cost-term flag at run time (only the never-reached bulk folded)
// (own translation units only: -DTEB_AMD_SINGLE_TU and the MFMA library, -DTEB_AMD_NO_DEFAULTS_TWINS, do without them; the host then
// launches the generic instantiation)
// (not built: the point-like small-batch kinds 2, 5, 9, 11 of the band-in-HBM layout - that layout runs without solver helpers, and a
// point-like scene has no distance helpers, so no launch can select them; round 6, tests/test_gpu_every_instantiation.py)
#define TEB_OPT_FOR_GENERIC_ANALYTIC(X) X(0, 0, 0) X(1, 0, 0) X(2, 0, 0) X(0, 0, 1) X(1, 0, 1) X(2, 0, 1) X(0, 0, 2) X(1, 0, 2) X(0, 0, 3) X(1, 0, 3) X(2, 0, 3)
#if defined(TEB_AMD_SINGLE_TU) || defined(TEB_AMD_NO_DEFAULTS_TWINS)
#define TEB_OPT_FOR_ANALYTIC(X) TEB_OPT_FOR_GENERIC_ANALYTIC(X)
#else
#define TEB_OPT_FOR_ANALYTIC(X) TEB_OPT_FOR_GENERIC_ANALYTIC(X) X(0, 0, 4) X(1, 0, 4) X(2, 0, 4) X(0, 0, 5) X(1, 0, 5) \
                                X(0, 0, 6) X(1, 0, 6) X(2, 0, 6) X(0, 0, 7) X(1, 0, 7) X(2, 0, 7) \
                                X(0, 0, 8) X(1, 0, 8) X(2, 0, 8) X(0, 0, 9) X(1, 0, 9) \
                                X(0, 0, 10) X(1, 0, 10) X(2, 0, 10) X(0, 0, 11) X(1, 0, 11)
#endif
#ifdef TEB_AMD_ANALYTIC_ONLY
#define TEB_OPT_FOR_ALL(X) TEB_OPT_FOR_ANALYTIC(X)
#else
#if defined(TEB_AMD_SINGLE_TU) || defined(TEB_AMD_NO_DEFAULTS_TWINS)
#define TEB_OPT_FOR_ALL(X) TEB_OPT_FOR_ANALYTIC(X) X(0, 1, 0) X(1, 1, 0) X(2, 1, 0) X(0, 1, 1) X(1, 1, 1) X(2, 1, 1)
#else
#define TEB_OPT_FOR_ALL(X) TEB_OPT_FOR_ANALYTIC(X) X(0, 1, 0) X(1, 1, 0) X(2, 1, 0) X(0, 1, 1) X(1, 1, 1) X(2, 1, 1) X(0, 1, 4) X(1, 1, 4) X(2, 1, 4)
#endif
#endif
