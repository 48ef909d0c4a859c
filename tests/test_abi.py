"""The C-ABI library loads on a CPU-only box, exports every symbol include/*.h declares, and its POD
layouts match the ctypes mirror. No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest

from teb_local_planner_amd import _abi, planner
from teb_local_planner_amd.config import TebConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(teb_amd_[a-z0-9_]+)\s*\(", src)))


@pytest.mark.parametrize("header", ["teb_amd.h", "teb_amd_debug.h"])
def test_exports_every_declared_symbol(header):
    L = planner.lib()
    names = _declared(header)
    assert len(names) >= 4
    for n in names:
        assert hasattr(L, n), "libteb_amd.so does not export %s" % n


def test_struct_sizes_match_ctypes_mirror():
    L = planner.lib()
    assert L.teb_amd_abi_version() == 3
    o = _abi.Options(layout="band")
    assert o.struct_size == C.sizeof(_abi.Options) == L.teb_amd_sizeof_options()
    d = _abi.Options(layout=3, fixed_layout=True)
    L.teb_amd_options_default(C.byref(d))
    assert d.layout == 0 and d.fixed_layout == 0 and d.struct_size == C.sizeof(_abi.Options)
    assert L.teb_amd_sizeof_config() == C.sizeof(_abi.Config)
    assert L.teb_amd_sizeof_obstacles() == C.sizeof(_abi.Obstacles)
    assert L.teb_amd_sizeof_teb_batch() == C.sizeof(_abi.TebBatch)
    assert L.teb_amd_sizeof_results() == C.sizeof(_abi.Results)


def test_max_poses_constant_matches_the_header():
    src = open(os.path.join(ROOT, "include", "teb_amd.h")).read()
    m = re.search(r"#define\s+TEB_AMD_MAX_POSES\s+(\d+)", src)
    assert m and int(m.group(1)) == _abi.MAX_POSES
    # what the band-in-HBM layout needs in LDS at that capacity (21 doubles per pose + 368, csrc/teb_kernel.hpp: make_lds_plan) fits MI355X's
    # 160 KB minus the 1 KiB the library keeps for the kernel's static LDS
    assert (21 * _abi.MAX_POSES + 368) * 8 <= 160 * 1024 - 1024 < (21 * (_abi.MAX_POSES + 8) + 368) * 8


def test_default_config_matches_reference_defaults():
    """teb_amd_config_default == TebConfig() of the Python mirror == teb_config.h:245-390."""
    L = planner.lib()
    c = _abi.Config()
    L.teb_amd_config_default(C.byref(c))
    p = TebConfig().to_c()
    for name, _ in _abi.Config._fields_:
        a, b = getattr(c, name), getattr(p, name)
        if hasattr(a, "__len__"):
            assert list(a) == list(b), name
        else:
            assert a == b, (name, a, b)
    # spot-check against the reference constructor (teb_config.h:321-345)
    assert c.no_inner_iterations == 5 and c.no_outer_iterations == 4
    assert c.weight_kinematics_nh == 1000 and c.weight_obstacle == 50 and c.weight_adapt_factor == 2.0
    assert c.dt_ref == 0.3 and c.dt_hysteresis == 0.1 and c.max_samples == 500
    assert c.min_obstacle_dist == 0.5 and c.inflation_dist == 0.6 and c.include_dynamic_obstacles == 1


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from teb_local_planner_amd import scenes
    with pytest.raises(planner.TebAmdError) as e:
        planner.make_solver(*scenes.scene_c1())
    assert e.value.code == _abi.ERR_NO_DEVICE


def test_world_argmin_of_the_distributed_selection():
    """Host-side reduction of teb_amd_select_best_distributed over the gathered (cost, global index) records (no GPU, no RCCL):
    lowest cost wins, ties go to the lowest GLOBAL index whatever rank holds it (strict '<' of selectBestTeb,
    src/homotopy_class_planner.cpp:610), empty ranks (index -1) and ranks whose best cost is DBL_MAX are handled."""
    import numpy as np
    L = planner.lib()
    DMAX = 1.7976931348623157e308

    def run(recs):
        a = np.ascontiguousarray(np.array(recs, np.float64).reshape(-1))
        best = C.c_int32(-7); cost = C.c_double(0); owner = C.c_int32(-7)
        assert L.teb_amd_debug_world_argmin(_abi._ptr(a, C.c_double), len(recs), C.byref(best), C.byref(cost), C.byref(owner)) == 0
        return best.value, cost.value, owner.value

    assert run([(3.0, 5), (2.0, 40), (4.0, 70)]) == (40, 2.0, 1)
    assert run([(2.0, 64), (2.0, 3), (2.0, 130)]) == (3, 2.0, 1)              # tie: lowest global index, owner = the rank that holds it
    assert run([(DMAX, -1), (5.0, 9)]) == (9, 5.0, 1)                           # rank 0 holds no candidate
    assert run([(DMAX, -1), (DMAX, -1)])[0] == -1                               # nobody does
    assert run([(DMAX, 2), (DMAX, -1), (DMAX, 1)]) == (1, DMAX, 2)              # all costs at DBL_MAX: still the lowest index
    assert run([(1.0, 0)]) == (0, 1.0, 0)
    assert L.teb_amd_debug_world_argmin(None, 2, None, None, None) != 0
