// teb_opt_inst.hip — ONE instantiation of teb_optimize_kernel per translation unit (see teb_opt_launch.hpp):
//   hipcc -c -DTEB_INST_SOLVER=<0|1|2> -DTEB_INST_JMODE=<0|1> -DTEB_INST_SCENE=<0|1|2|3> teb_opt_inst.hip
#include <hip/hip_runtime.h>

#ifdef TEB_INST_STUB
// (tools/ experiments, build.py variant "exp": an instantiation the experiment does not launch - the host gets a null kernel address and
// refuses the launch)
#include "teb_opt_launch.hpp"
__attribute__((visibility("hidden"))) const void* TEB_OPT_KERNEL_FN(TEB_INST_SOLVER, TEB_INST_JMODE, TEB_INST_SCENE)() { return nullptr; }
#else

#if defined(TEB_INST_SCENE) && TEB_INST_SCENE >= 4
#define TEB_AMD_DEFAULTS_PROFILE 1   // scene kinds 4 .. 11: configuration flags folded to the TebConfig defaults (teb_device.hpp: TEB_CFG)
#endif
#if defined(TEB_INST_SCENE) && (TEB_INST_SCENE == 6 || TEB_INST_SCENE == 7)
#define TEB_AMD_PROFILE_ANY_KINEMATICS 1   // generic-shape kinds 6, 7: diff-drive / car-like stays a run-time choice
#endif
#if defined(TEB_INST_SCENE) && (TEB_INST_SCENE == 8 || TEB_INST_SCENE == 9)
#define TEB_AMD_PROFILE_WIDE 1   // point-like kinds 8, 9: via-points and holonomic / non-holonomic edges stay run-time choices
#endif
#if defined(TEB_INST_SCENE) && TEB_INST_SCENE >= 10
#define TEB_AMD_PROFILE_LIGHT 1   // point-like kinds 10, 11: every cost-term flag at run time, the never-reached bulk folded
#endif
#if defined(TEB_INST_SOLVER) && TEB_INST_SOLVER == 2 && !defined(TEB_AMD_POSE_ITER)
#define TEB_AMD_POSE_ITER 4   // band in HBM: up to four poses per lane (teb_device.hpp: kPoseIterBandHbm)
#endif
#include "teb_kernel.hpp"
#include "teb_opt_launch.hpp"

#if !defined(TEB_INST_SOLVER) || !defined(TEB_INST_JMODE) || !defined(TEB_INST_SCENE)
#error "teb_opt_inst.hip needs -DTEB_INST_SOLVER, -DTEB_INST_JMODE and -DTEB_INST_SCENE"
#endif
static_assert(tebamd::SOLVER_BAND == 0 && tebamd::SOLVER_CR == 1 && tebamd::SOLVER_BANDG == 2, "teb_opt_launch.hpp numbers the layouts");
static_assert(TEB_AMD_JACOBIAN_ANALYTIC == 0 && TEB_AMD_JACOBIAN_G2O_NUMERIC == 1, "teb_opt_launch.hpp numbers the Jacobian modes");
static_assert(tebamd::SCENE_POINTS == 0 && tebamd::SCENE_GENERIC == 1 && tebamd::SCENE_POINTS_SMALL == 2 && tebamd::SCENE_GENERIC_SMALL == 3 &&
              tebamd::SCENE_POINTS_DEFAULTS == 4 && tebamd::SCENE_POINTS_SMALL_DEFAULTS == 5 && tebamd::SCENE_GENERIC_DEFAULTS == 6 &&
              tebamd::SCENE_GENERIC_SMALL_DEFAULTS == 7 && tebamd::SCENE_POINTS_WIDE == 8 && tebamd::SCENE_POINTS_SMALL_WIDE == 9 && tebamd::SCENE_POINTS_LIGHT == 10 && tebamd::SCENE_POINTS_SMALL_LIGHT == 11,
              "teb_opt_launch.hpp numbers the scene kinds");

static_assert(TEB_INST_SOLVER != 2 || tebamd::kMaxPoseIter == tebamd::kPoseIterBandHbm, "the host sizes band-in-HBM handles for kPoseIterBandHbm poses per lane");

TEB_OPT_DEFINE(TEB_INST_SOLVER, TEB_INST_JMODE, TEB_INST_SCENE)
#endif   // TEB_INST_STUB
