"""The kernel instantiations specialised on the TebConfig defaults (csrc/teb_device.hpp: TEB_CFG; teb_amd_options_t::generic_config_path):
a configuration that takes the default paths runs them and gets the bands of the generic instantiation bit for bit; any configuration that
leaves one of the folded paths must run the generic instantiation."""
import copy

import numpy as np
import pytest

from teb_local_planner_amd import _abi, planner, scenes
from teb_local_planner_amd.config import RobotFootprintModel

pytestmark = pytest.mark.gpu


def run(cfg, obst, via, batch, **opt):
    s = planner.make_solver(cfg, obst, via, batch, options=_abi.Options(**opt) if opt else None)
    s.optimize(cfg.optim.no_inner_iterations, cfg.optim.no_outer_iterations, True, 100.0, 1.0, False)
    res = s.results()
    out = s.download(batch.copy())
    prof = s.last_config_profile()
    info = s.last_launch_info()
    s.close()
    return out, res, prof, info


def same_bits(a, ra, b, rb):
    np.testing.assert_array_equal(a.n, b.n)
    for f in ("x", "y", "theta", "dt"):
        np.testing.assert_array_equal(getattr(a, f), getattr(b, f))
    for f in ("status", "lm_iterations", "lm_trials", "chi2", "cost"):
        np.testing.assert_array_equal(getattr(ra, f), getattr(rb, f))


@pytest.mark.parametrize("scene", ["c2", "c3", "c4_blocks", "c4_hybrid", "c4_band_hbm"])
def test_default_configuration_runs_the_specialised_kernel_with_identical_bits(scene):
    if scene == "c2":
        cfg, obst, via, batch = scenes.scene_c2(stride=208)                       # one band, solver helpers (small-batch kind)
    elif scene == "c3":
        cfg, obst, via, batch = scenes.scene_c3(B=16, n=150, M=200, stride=208)
    elif scene == "c4_blocks":
        cfg, obst, via, batch = scenes.scene_c4(B=24, stride=208); cfg.trajectory.teb_autosize = False
    elif scene == "c4_hybrid":
        cfg, obst, via, batch = scenes.scene_c4(B=24, stride=288)
    else:
        cfg, obst, via, batch = scenes.scene_c4(B=8, n=300, stride=420)
    a, ra, pa, ia = run(cfg, obst, via, batch)
    b, rb, pb, ib = run(cfg, obst, via, batch, generic_config_path=True)
    assert pa and not pb, (pa, pb)
    assert ia == ib
    same_bits(a, ra, b, rb)


@pytest.mark.parametrize("layout", ["band", "blocks"])
@pytest.mark.parametrize("B", [2, 24, 64])
def test_small_batches_with_solver_helpers_in_both_kernels(layout, B):
    """Small batches run with solver helper workgroups (speculative LM trials): specialised and generic kernel, with the helpers and
    confined to one CU per band - four launches, one result. (The band-layout small-batch kernel of the specialised kind is the one that
    exposed the call-convention interaction described at cr_solve_hybrid_helper.)"""
    if layout == "band":
        cfg, obst, via, batch = scenes.scene_c4(B=B, stride=288)
    else:
        cfg, obst, via, batch = scenes.scene_c4(B=B, stride=208); cfg.trajectory.teb_autosize = False
    t = run(cfg, obst, via, batch)
    g = run(cfg, obst, via, batch, generic_config_path=True)
    t0 = run(cfg, obst, via, batch, speculative_trials=-1)
    g0 = run(cfg, obst, via, batch, generic_config_path=True, speculative_trials=-1)
    assert t[2] and t0[2] and not g[2] and not g0[2]
    assert t[3][1] > 0 and g[3][1] > 0 and t0[3][1] == 0 and g0[3][1] == 0, (t[3], g[3], t0[3], g0[3])
    for other in (g, t0, g0):
        same_bits(t[0], t[1], other[0], other[1])


@pytest.mark.parametrize("stride", [208, 288])
def test_numeric_jacobian_mode_has_its_specialised_kernel_too(stride):
    cfg, obst, via, batch = scenes.scene_c4(B=12, stride=stride)
    cfg.jacobian_mode = 1   # TEB_AMD_JACOBIAN_G2O_NUMERIC
    if stride == 208:
        cfg.trajectory.teb_autosize = False
    t = run(cfg, obst, via, batch)
    g = run(cfg, obst, via, batch, generic_config_path=True)
    assert t[2] and not g[2]
    same_bits(t[0], t[1], g[0], g[1])


@pytest.mark.parametrize("scene", ["c5_carlike_polygons", "mixed_polygon_diffdrive", "mixed_two_circles"])
def test_generic_shape_scenes_have_specialised_kernels_for_either_kinematics(scene):
    """polygon / line / two-circle robots and obstacles: the generic-shape kinds of the profile keep diff-drive / car-like at run time"""
    if scene == "c5_carlike_polygons":
        cfg, obst, via, batch = scenes.scene_c5(stride=320)           # 60 distance + 3 solver helpers
    else:
        cfg, obst, via, batch = scenes.scene_small_mixed(footprint="polygon" if "polygon" in scene else "two_circles", with_via=False)
    t = run(cfg, obst, via, batch)
    g = run(cfg, obst, via, batch, generic_config_path=True)
    assert t[2] and not g[2]
    assert t[3] == g[3]
    same_bits(t[0], t[1], g[0], g[1])
    t0 = run(cfg, obst, via, batch, multi_cu=-1, speculative_trials=-1)
    assert t0[2] and t0[3][:2] == (0, 0)
    same_bits(t[0], t[1], t0[0], t0[1])


def test_circular_footprint_is_inside_the_profile():
    cfg, obst, via, batch = scenes.scene_c3(B=8, n=100, M=120, stride=208)
    cfg.robot_model = RobotFootprintModel.circular(0.2)
    t = run(cfg, obst, via, batch)
    g = run(cfg, obst, via, batch, generic_config_path=True)
    assert t[2] and not g[2]
    same_bits(t[0], t[1], g[0], g[1])


def _with(cfg, **kw):
    c = copy.deepcopy(cfg)
    for k, v in kw.items():
        grp, name = k.split("__")
        setattr(getattr(c, grp), name, v)
    return c


def test_configurations_off_the_default_cost_terms_run_the_light_kinds_with_identical_bits():
    """Round 4: a configuration that leaves the defaults in a cost-term flag runs the *_LIGHT kinds (every cost-term flag at run time, only
    the never-reached bulk folded: legacy association, debug export, sequential LDL^T, uncached near masks, divergence detection) and gets
    the generic kernel's bands bit for bit; what those kinds fold still sends a configuration to the generic kernel."""
    cfg0, obst, via, batch = scenes.scene_c3(B=4, n=60, M=40, stride=96)
    light = {
        "no velocity edges": _with(cfg0, optim__weight_max_vel_x=0.0, optim__weight_max_vel_theta=0.0),
        "no acceleration edges": _with(cfg0, optim__weight_acc_lim_x=0.0, optim__weight_acc_lim_theta=0.0),
        "no time-optimal edges": _with(cfg0, optim__weight_optimaltime=0.0),
        "shortest path": _with(cfg0, optim__weight_shortest_path=1.0),
        "car-like": _with(cfg0, robot__min_turning_radius=0.8, optim__weight_kinematics_turning_radius=1.0),
        "no kinematics edges": _with(cfg0, optim__weight_kinematics_nh=0.0, optim__weight_kinematics_forward_drive=0.0),
        "velocity-obstacle ratio": _with(cfg0, optim__weight_velocity_obstacle_ratio=1.0),
        "no obstacle edges": _with(cfg0, optim__weight_obstacle=0.0),
        "exact arc length": _with(cfg0, trajectory__exact_arc_length=True),
        "cost exponent": _with(cfg0, optim__obstacle_cost_exponent=1.5),
        "holonomic with shortest path": _with(cfg0, robot__max_vel_y=0.2, robot__acc_lim_y=0.3, robot__max_vel_trans=0.5, optim__weight_shortest_path=1.0),
    }
    for name, cfg in light.items():
        t = run(cfg, obst, via, batch)
        g = run(cfg, obst, via, batch, generic_config_path=True)
        assert t[2] == 3 and g[2] == 0, (name, t[2], g[2])
        assert (np.asarray(t[1].status) != _abi.TEB_NONFINITE).all(), name
        same_bits(t[0], t[1], g[0], g[1])
    generic = {
        "legacy association": _with(cfg0, obstacles__legacy_obstacle_association=True),
        "divergence detection": _with(cfg0, recovery__divergence_detection_enable=True),
    }
    for name, cfg in generic.items():
        _, res, prof, _ = run(cfg, obst, via, batch)
        assert prof == 0, name
        assert (np.asarray(res.status) != _abi.TEB_NONFINITE).all(), name
    assert run(cfg0, obst, via, batch, no_near_cache=True)[2] == 0
    assert run(cfg0, obst, via, batch)[2] == 1, "the unchanged configuration is on the folded paths"


@pytest.mark.parametrize("layout", ["blocks", "hybrid"])
@pytest.mark.parametrize("what", ["via_points", "holonomic", "holonomic_with_via_points", "holonomic_velocity_only"])
def test_via_points_and_holonomic_robots_run_the_wide_kinds_with_identical_bits(what, layout):
    """Round 4: configurations that differ from the defaults only in via-points and / or a holonomic base run the
    *_WIDE kinds (every other fold of the profile kept, teb_device.hpp: TEB_PF_WIDE_*), full batch and small batch with solver helpers,
    and get the generic kernel's bands bit for bit."""
    for B in (3, 24):
        if layout == "hybrid":
            cfg, obst, via, batch = scenes.scene_c4(B=B, stride=288)
        else:
            cfg, obst, via, batch = scenes.scene_c4(B=B, stride=208); cfg.trajectory.teb_autosize = False
        if "holonomic" in what:
            cfg.robot.max_vel_y = 0.2; cfg.robot.max_vel_trans = 0.5
            cfg.robot.acc_lim_y = 0.0 if what == "holonomic_velocity_only" else 0.3   # acc_lim_y == 0: holonomic velocity, non-holonomic acceleration edges
            cfg.optim.weight_max_vel_y = 2.0; cfg.optim.weight_acc_lim_y = 1.0
        if "via" in what:
            cfg.optim.weight_viapoint = 1.0
            via = [(5.0, 0.3), (10.0, -0.2), (15.0, 0.25)]
            batch.via_points_enabled[:] = 1
            batch.via_points_enabled[0] = 0      # per-band switch stays a run-time flag
        t = run(cfg, obst, via, batch)
        g = run(cfg, obst, via, batch, generic_config_path=True)
        assert t[2] == 2 and g[2] == 0, (what, layout, B, t[2], g[2])
        assert t[3] == g[3]
        same_bits(t[0], t[1], g[0], g[1])


def test_wide_kinds_do_not_take_what_they_fold():
    cfg0, obst, via, batch = scenes.scene_c3(B=4, n=60, M=40, stride=96)
    cfg = _with(cfg0, robot__max_vel_y=0.2, robot__acc_lim_y=0.3, optim__weight_shortest_path=1.0)   # holonomic AND a flag the wide kinds fold
    assert run(cfg, obst, via, batch)[2] == 3                                                      # (the light kinds take it)
    cfg = _with(cfg0, robot__max_vel_y=0.2, robot__acc_lim_y=0.3)
    cfg.jacobian_mode = 1                                                                          # the wide kinds exist for closed forms only
    assert run(cfg, obst, via, batch)[2] == 0


def test_kernel_compiled_for_the_configuration_at_run_time():
    """teb_amd_options_t::compile_for_config (csrc/teb_rtc.hpp): a configuration off the defaults gets an instantiation with EVERY flag of
    the profile table folded to its own values, compiled by hipRTC from the library's sources - the bands of the generic kernel bit for
    bit, at the speed of a default configuration. Synchronous mode first (the launch waits for the compiler), then a second handle finds
    the module in the process-wide cache; a generic-shape scene off the defaults; the numeric Jacobian mode."""
    import time
    cfg, obst, via, batch = scenes.scene_c4(B=24, stride=288)
    cfg.optim.weight_shortest_path = 1.0
    cfg.optim.obstacle_cost_exponent = 1.5
    cfg.robot.max_vel_y = 0.2; cfg.robot.acc_lim_y = 0.3; cfg.robot.max_vel_trans = 0.5
    g = run(cfg, obst, via, batch, generic_config_path=True)
    t0 = time.perf_counter()
    t = run(cfg, obst, via, batch, compile_for_config=2)
    first = time.perf_counter() - t0
    ready, compiling, failed, secs, err = planner.TebBatchSolver.rtc_stats()
    assert t[2] == 4 and g[2] == 0, (t[2], g[2], err)
    assert failed == 0 and ready >= 1, (ready, compiling, failed, err)
    same_bits(t[0], t[1], g[0], g[1])
    t0 = time.perf_counter()
    t2 = run(cfg, obst, via, batch, compile_for_config=2)
    again = time.perf_counter() - t0
    assert t2[2] == 4
    same_bits(t2[0], t2[1], g[0], g[1])
    print("compiled for the configuration: %.1f s in the compiler, first call %.1f s, second handle %.3f s (cached module)" % (secs, first, again))
    assert again < 0.5 * first
    # blocks-in-LDS layout, small batch with solver helpers
    cfg2, obst2, via2, batch2 = scenes.scene_c3(B=4, n=60, M=40, stride=96)
    cfg2.optim.weight_velocity_obstacle_ratio = 1.0
    a = run(cfg2, obst2, via2, batch2, compile_for_config=2)
    b = run(cfg2, obst2, via2, batch2, generic_config_path=True)
    assert a[2] == 4 and a[3] == b[3] and a[3][1] > 0, (a[2], a[3], b[3])
    same_bits(a[0], a[1], b[0], b[1])
    # generic shapes off the defaults (polygon footprint, via-points, every obstacle type)
    cfg3, obst3, via3, batch3 = scenes.scene_small_mixed(footprint="polygon")
    a = run(cfg3, obst3, via3, batch3, compile_for_config=2, multi_cu=-1, speculative_trials=-1)
    b = run(cfg3, obst3, via3, batch3, generic_config_path=True, multi_cu=-1, speculative_trials=-1)
    assert a[2] == 4 and b[2] == 0
    same_bits(a[0], a[1], b[0], b[1])
    # the reference's own linearisation scheme off the defaults
    cfg.jacobian_mode = 1
    a = run(cfg, obst, via, batch, compile_for_config=2)
    b = run(cfg, obst, via, batch, generic_config_path=True)
    assert a[2] == 4 and b[2] == 0
    same_bits(a[0], a[1], b[0], b[1])


def test_kernel_compiled_for_the_configuration_on_a_band_beyond_512_poses():
    """The run-time compiler instantiates the band-in-HBM layout with four poses per lane like the pre-built units (csrc/teb_rtc.hpp passes
    TEB_AMD_POSE_ITER): a 600-pose band off the defaults - the compiled kernel (profile 4) returns the generic kernel's bands bit for bit."""
    cfg, obst, via, _ = scenes.scene_c4(B=2, stride=288)
    cfg.trajectory.teb_autosize = False
    cfg.trajectory.max_samples = 1000
    cfg.optim.weight_shortest_path = 0.7
    batch = _abi.TebBatchHost(2, 640)
    for b, n in enumerate((600, 587)):
        batch.set_teb(b, *scenes.sine_band(n, 0.25 * n, 0.2 - 0.3 * b, 1.0, cfg.robot.max_vel_x))
    a = run(cfg, obst, via, batch, compile_for_config=2)
    g = run(cfg, obst, via, batch, generic_config_path=True)
    ready, compiling, failed, secs, err = planner.TebBatchSolver.rtc_stats()
    assert a[2] == 4 and g[2] == 0 and failed == 0, (a[2], g[2], err)
    assert (a[1].status == _abi.TEB_OK).all()
    same_bits(a[0], a[1], g[0], g[1])


def test_background_compilation_runs_the_prebuilt_kernel_until_the_module_is_ready():
    import time
    cfg, obst, via, batch = scenes.scene_c3(B=6, n=80, M=60, stride=128)
    cfg.optim.weight_shortest_path = 0.5        # (a flag combination no other test compiles)
    cfg.trajectory.exact_arc_length = True
    g = run(cfg, obst, via, batch, generic_config_path=True)
    s = planner.make_solver(cfg, obst, via, batch, options=_abi.Options(compile_for_config=1))
    s.snapshot()
    seen = []
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 120.0:
        s.restore()
        s.optimize(cfg.optim.no_inner_iterations, cfg.optim.no_outer_iterations, True, 100.0, 1.0, False)
        prof = s.last_config_profile()
        seen.append(prof)
        same_bits(s.download(batch.copy()), s.results(), g[0], g[1])     # whichever kernel ran: the same bands
        if prof == 4:
            break
        time.sleep(0.05)
    s.close()
    assert seen[0] == 3, seen[:3]                 # the first launch did not wait: the light kinds ran
    assert seen[-1] == 4, (seen[-5:], planner.TebBatchSolver.rtc_stats())
    print("background compilation: %d launches on the pre-built kernel, then the compiled one after %.1f s" % (len(seen) - 1, time.perf_counter() - t0))


_DISK_CHILD = r"""
import ctypes as C, hashlib, json, sys, time
sys.path.insert(0, %r)
from teb_local_planner_amd import scenes, planner, _abi
cfg, obst, via, batch = scenes.scene_c3(B=6, n=80, M=60, stride=128)
cfg.optim.weight_shortest_path = 0.25
cfg.optim.obstacle_cost_exponent = 1.25
t0 = time.perf_counter()
s = planner.make_solver(cfg, obst, via, batch, options=_abi.Options(compile_for_config=2))
s.optimize(cfg.optim.no_inner_iterations, cfg.optim.no_outer_iterations, True, 100.0, 1.0, False)
out = s.download(batch.copy()); res = s.results()
dt = time.perf_counter() - t0
h = hashlib.sha256()
for a in (out.n, out.x, out.y, out.theta, out.dt, res.chi2, res.cost, res.lm_trials): h.update(a.tobytes())
L = planner.lib()
L.teb_amd_debug_rtc_cache.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_char_p, C.c_int32]
e = C.c_int32(-1); hits = C.c_int32(-1); w = C.c_int32(-1); buf = C.create_string_buffer(1024)
L.teb_amd_debug_rtc_cache(C.byref(e), C.byref(hits), C.byref(w), buf, 1024)
print(json.dumps(dict(profile=s.last_config_profile(), bits=h.hexdigest(), seconds=dt, embedded=e.value, hits=hits.value, writes=w.value)))
"""


def test_a_second_process_runs_the_code_object_the_first_one_left_on_disk(tmp_path):
    """The disk cache of the run-time compiler (csrc/teb_rtc.hpp) on the GPU: the first process compiles (from the sources embedded in the
    library) and stores, the second loads the stored code object as a module and runs it - same kernel kind, same bits, no compiler."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(planner.__file__)))
    env = dict(os.environ, TEB_AMD_RTC_CACHE=str(tmp_path / "cache"), AMD_COMGR_CACHE="0")
    runs = []
    for _ in range(2):
        p = subprocess.run([sys.executable, "-c", _DISK_CHILD % root], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        runs.append(json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]))
    a, b = runs
    assert a["profile"] == 4 and b["profile"] == 4, runs
    assert a["embedded"] == 1 and (a["hits"], a["writes"]) == (0, 1) and (b["hits"], b["writes"]) == (1, 0), runs
    assert a["bits"] == b["bits"], runs
    print("compiled in the first process %.1f s (whole call), loaded from disk in the second %.1f s" % (a["seconds"], b["seconds"]))


@pytest.mark.parametrize("layout", ["band", "blocks"])
@pytest.mark.parametrize("kind", ["prebuilt", "generic", "compiled"])
def test_every_kernel_kind_on_the_rarely_taken_cost_terms_with_and_without_solver_helpers(layout, kind):
    """Round 4 found three instantiations (generic small-batch band, light full-batch band, light small-batch blocks) faulting - memory
    aperture violation - as soon as obstacle_cost_exponent != 1 sent the edge loops through pow(): a path no test took. The cause is the
    no-callee-saved call of the solve in the big instantiations (the class of backend interaction round 3 met with two such call sites);
    every instantiation that keeps the cost terms at run time calls the solve on the plain convention now (build.py: UNIT_FLAGS). This
    matrix (tools/kind_matrix.py; one process per (layout, kind): a GPU fault aborts only that one, its last line says where) runs each kind a configuration can
    reach - the host's pre-built pick, the generic kernel forced, the kernel compiled for the configuration - in both layouts through
    nine rarely taken configurations (four for the compiled kind: every case costs two compilations) WITH solver helpers and without, and holds the two launches bit-identical."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "kind_matrix.py"), layout, kind, "all"], cwd=root, capture_output=True, text=True, timeout=900)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
