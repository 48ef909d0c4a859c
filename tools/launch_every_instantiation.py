"""Launch EVERY pre-built instantiation of teb_optimize_kernel (layout x Jacobian mode x scene kind, csrc/teb_opt_launch.hpp) once on the GPU
and check that it ran, that it was the instantiation meant (teb_amd_debug_last_instantiation) and that its bands are finite and ok.
Why: a miscompiled unit (profiles/fault_bisect_r05.txt: the no-callee-saved call of the out-of-line solve under interprocedural register
allocation; which unit is hit moves with its register allocation) aborts the process with a memory aperture violation at the first solve -
this turns a moved fault into a failed test instead of a failed robot. The configurations take the rarely used paths where
a kind has them (cost exponent != 1 -> pow(), shortest path, via-points).
Usage (GPU box):  python tools/launch_every_instantiation.py            every layout, each in a process of its own
                  python tools/launch_every_instantiation.py <layout>   the cases of one layout (blocks | band | bandg) in this process"""
import subprocess
import sys

sys.path.insert(0, ".")
LAYOUTS = {"band": 0, "blocks": 1, "bandg": 2}
# scene kinds (csrc/teb_kernel.hpp): 0 points, 1 generic, 2 / 3 their small-batch twins, 4 .. 7 the same four on the TebConfig defaults,
# 8 / 9 wide, 10 / 11 light
# (band in HBM, layout 2, runs without solver helpers - mcu_helpers_for, csrc/teb_amd.hip - so its point-like small-batch kinds 2, 5, 9, 11
#  cannot be launched and are not built: build.py, csrc/teb_opt_launch.hpp)
EXPECTED = ({(lay, 0, k) for lay in range(3) for k in range(12)} | {(lay, 1, k) for lay in range(3) for k in (0, 1, 4)}) - {(2, 0, k) for k in (2, 5, 9, 11)}


def cases(layout):
    """(label, cfg, obst, via, batch, options, expected (jmode, kind))"""
    from teb_local_planner_amd import scenes, _abi
    lay = {"band": "band", "blocks": "cr", "bandg": "bandg"}[layout]
    stride = {"band": 288, "blocks": 208, "bandg": 400}[layout]

    def pts(B):
        c, o, v, b = scenes.scene_c4(B=B, n=120, stride=stride)
        if layout == "blocks":
            c.trajectory.teb_autosize = False
        return c, o, v, b

    def poly(B, with_via=True):
        return scenes.scene_small_mixed(B=B, n=40, stride=stride, footprint="polygon", with_via=with_via)

    out = []
    for helpers in (True, False):   # small-batch kinds run when the launch has helper workgroups (B small, closed-form Jacobians)
        spec = {} if helpers else {"speculative_trials": -1, "multi_cu": -1}
        sm = 1 if helpers else 0
        # point-like scenes: defaults, wide (via-points), light (shortest path + cost exponent), generic (forced, cost exponent)
        c, o, v, b = pts(8)
        out.append(("points defaults", c, o, v, b, dict(spec), (0, 4 + sm)))
        c, o, v, b = pts(8)
        c.optim.weight_viapoint = 1.0; v = [(5.0, 0.3), (10.0, -0.2)]; b.via_points_enabled[:] = 1
        out.append(("points wide (via-points)", c, o, v, b, dict(spec), (0, 8 + sm)))
        c, o, v, b = pts(8)
        c.optim.weight_shortest_path = 1.0; c.optim.obstacle_cost_exponent = 1.5
        out.append(("points light (shortest path, exponent)", c, o, v, b, dict(spec), (0, 10 + sm)))
        c, o, v, b = pts(8)
        c.optim.obstacle_cost_exponent = 1.5
        out.append(("points generic (forced, exponent)", c, o, v, b, dict(spec, generic_config_path=True), (0, 0 + 2 * sm)))
        # generic shapes: defaults profile, generic forced. Their small-batch kinds run with DISTANCE helpers (multi_cu) or solver helpers.
        # (multi_cu = 2: two DISTANCE helpers per band even on this small scene - the band-in-HBM layout has no solver helpers, its
        #  generic-shape small-batch kinds run with distance helpers only)
        pspec = dict(spec) if not helpers else dict(spec, multi_cu=2)
        c, o, v, b = poly(3, with_via=False)   # (via-points are not folded by the profile of the defaults)
        out.append(("polygon defaults", c, o, v, b, dict(pspec), (0, 6 + sm)))
        c, o, v, b = poly(3)
        c.optim.obstacle_cost_exponent = 1.5
        out.append(("polygon generic (forced, exponent)", c, o, v, b, dict(pspec, generic_config_path=True), (0, 1 + 2 * sm)))
    # the reference's own linearisation scheme (g2o central differences): full-batch kinds 0, 1, 4
    c, o, v, b = pts(8); c.jacobian_mode = _abi.JACOBIAN_G2O_NUMERIC
    out.append(("numeric points defaults", c, o, v, b, {}, (1, 4)))
    c, o, v, b = pts(8); c.jacobian_mode = _abi.JACOBIAN_G2O_NUMERIC; c.optim.obstacle_cost_exponent = 1.5
    out.append(("numeric points generic", c, o, v, b, {"generic_config_path": True}, (1, 0)))
    c, o, v, b = poly(3); c.jacobian_mode = _abi.JACOBIAN_G2O_NUMERIC
    out.append(("numeric polygon", c, o, v, b, {}, (1, 1)))
    return [(lbl, c, o, v, b, _abi.Options(layout=lay, fixed_layout=True, **opt), exp) for lbl, c, o, v, b, opt, exp in out]


def run_layout(layout):
    import numpy as np
    from teb_local_planner_amd import planner, _abi
    seen = set()
    for lbl, cfg, obst, via, batch, opt, (jm, kind) in cases(layout):
        print("%s | %s .." % (layout, lbl), flush=True)    # (a GPU fault ends the process: the last line says where)
        s = planner.make_solver(cfg, obst, via, batch, options=opt)
        s.optimize(2, 2, True, 100.0, 1.0, False)
        s.synchronize()
        inst = s.last_instantiation()
        res = s.results()
        out = s.download(batch.copy())
        hl = s.last_launch_info()
        s.close()
        ok = bool((res.status == _abi.TEB_OK).all()) and all(np.isfinite(a).all() for a in (out.x, out.y, out.theta, out.dt)) and bool(np.isfinite(res.cost).all())
        print("%s | %s -> instantiation %s helpers %s %s" % (layout, lbl, inst, hl[:2], "ok" if ok else "BAD RESULT"), flush=True)
        if not ok:
            return 1
        if inst != (LAYOUTS[layout], jm, kind):
            # (band in HBM has no solver helpers: its point-like small-batch kinds are unreachable by design and reported, not failed)
            print("%s | %s: expected instantiation %s" % (layout, lbl, (LAYOUTS[layout], jm, kind)), flush=True)
        seen.add(inst)
    missing = sorted(k for k in EXPECTED if k[0] == LAYOUTS[layout] and k not in seen)
    print("%s: launched %d instantiations; not reached: %s" % (layout, len(seen), missing), flush=True)
    return 0


if __name__ == "__main__":
    if len(sys.argv) > 1:
        sys.exit(run_layout(sys.argv[1]))
    rc = 0
    for layout in LAYOUTS:
        r = subprocess.run([sys.executable, __file__, layout], capture_output=True, text=True, timeout=900)
        print(r.stdout.strip(), flush=True)
        if r.returncode != 0:
            rc = 1
            print("rc %d FAULT or failure in layout %s: %s" % (r.returncode, layout, ([l for l in r.stderr.splitlines() if "HSA_STATUS" in l or "Error" in l] or [""])[-1][-160:]), flush=True)
    sys.exit(rc)
