"""SURVEY section 8 row g (north_star: "MFMA only on the dense Schur block"): the build whose cyclic reduction folds the Schur
complements with v_mfma_f64_16x16x4_f64 (libteb_amd_mfma.so = -DTEB_AMD_MFMA_SCHUR, built by teb_local_planner_amd.build next to the
product). It replaces the block solver the reference selects at include/teb_local_planner/optimal_planner.h:75-78 just like the vector
build; it is not the product because it is slower end to end (HISTORY.md section 3) - but it is built by build() and exercised here:
  * the operand maps of the matrix instruction (teb_amd_debug_mfma_selftest): C = A B exactly (max |C - A B| = 0 on integer-valued
    operands, where every product and sum is exact in fp64);
  * the measured configurations (tests/test_gpu_measured_configs.py, all 256 headline bands, C2, C3, C4 fixed) against the oracle
    with the library swapped in (a child interpreter: the library is process-global)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
MFMA_LIB = os.path.join(ROOT, "teb_local_planner_amd", "libteb_amd_mfma.so")

_SELFTEST = r'''
import ctypes as C, numpy as np, sys
from teb_local_planner_amd import planner, scenes, _abi
L = planner.lib()
cfg, obst, via, batch = scenes.scene_c1()
s = planner.make_solver(cfg, obst, via, batch)
rng = np.random.default_rng(1)
A = np.ascontiguousarray(rng.integers(-9, 10, (16, 8)).astype(np.float64)); B = np.ascontiguousarray(rng.integers(-9, 10, (8, 16)).astype(np.float64))
Cm = np.zeros((16, 16)); cyc = C.c_double(0)
L.teb_amd_debug_mfma_selftest.argtypes = [C.c_void_p, _abi.p_f64, _abi.p_f64, _abi.p_f64, C.c_int32, C.POINTER(C.c_double)]
rc = L.teb_amd_debug_mfma_selftest(s._h, _abi._ptr(A, C.c_double), _abi._ptr(B, C.c_double), _abi._ptr(Cm, C.c_double), 2048, C.byref(cyc))
print("rc", rc, "maxerr", float(np.abs(Cm - A @ B).max()), "cycles_per_mfma", cyc.value)
sys.exit(0 if rc == 0 and np.array_equal(Cm, A @ B) and cyc.value > 0 else 1)
'''


def _env():
    if not os.path.exists(MFMA_LIB):
        pytest.fail("teb_local_planner_amd/libteb_amd_mfma.so is missing: __graft_entry__.build() builds it")
    e = dict(os.environ)
    e["TEB_AMD_LIB"] = MFMA_LIB
    e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
    return e


def test_matrix_instruction_operand_maps_are_exact():
    r = subprocess.run([sys.executable, "-c", _SELFTEST], env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr[-2000:])
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
    assert "maxerr 0.0" in r.stdout


def test_product_library_has_no_matrix_variant():
    from teb_local_planner_amd import planner, scenes, _abi
    L = planner.lib()
    s = planner.make_solver(*scenes.scene_c1())
    z = np.zeros(256)
    L.teb_amd_debug_mfma_selftest.argtypes = [C.c_void_p, _abi.p_f64, _abi.p_f64, _abi.p_f64, C.c_int32, C.POINTER(C.c_double)]
    assert L.teb_amd_debug_mfma_selftest(s._h, _abi._ptr(z, C.c_double), _abi._ptr(z, C.c_double), _abi._ptr(z, C.c_double), 1, None) == _abi.ERR_UNSUPPORTED
    s.close()


def test_measured_configurations_pass_with_the_matrix_build():
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_measured_configs.py")],
                       env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=900)
    print(r.stdout[-3000:], r.stderr[-1000:])
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


_RTC_CHILD = r'''
import hashlib, json, sys, tempfile, os
os.environ["TEB_AMD_RTC_CACHE"] = tempfile.mkdtemp()
import numpy as np
from teb_local_planner_amd import planner, scenes, _abi
def run(**opt):
    cfg, obst, via, batch = scenes.scene_c3(B=8, n=120, M=80, stride=208)
    cfg.optim.weight_shortest_path = 1.0          # off the defaults: the matrix build launches its generic kernel, or the one compiled for it
    s = planner.make_solver(cfg, obst, via, batch, options=_abi.Options(**opt))
    s.optimize(5, 4, True, 100.0, 1.0, False); s.synchronize()
    out = s.download(batch.copy()); r = s.results()
    h = hashlib.sha256()
    for a in (out.n, out.x, out.y, out.theta, out.dt, r.cost, r.chi2, r.lm_trials): h.update(np.ascontiguousarray(a).tobytes())
    p = s.last_config_profile(); s.close()
    return h.hexdigest(), p
a, pa = run()
b, pb = run(compile_for_config=2)
cfg, obst, via, batch = scenes.scene_c1(); cfg.jacobian_mode = _abi.JACOBIAN_G2O_NUMERIC
try:
    s = planner.make_solver(cfg, obst, via, batch, options=_abi.Options(compile_for_config=2)); s.optimize(5, 4, True, 100.0, 1.0, False); numeric = "ran"
except planner.TebAmdError as e:
    numeric = "refused"
print(json.dumps(dict(same=a == b, prebuilt=pa, compiled=pb, numeric=numeric)))
'''


def test_run_time_compiled_kernel_is_the_same_variant():
    """a kernel compiled at run time (teb_amd_options_t::compile_for_config) in the MATRIX build carries the build's defines
    (-DTEB_AMD_MFMA_SCHUR ..: csrc/teb_rtc.hpp appends TEB_AMD_VARIANT_DEFINES to the compiler's options and to the disk-cache key), so it
    is the same arithmetic as the pre-built kernels it replaces mid-run - bit-identical bands - and it does not add the numeric Jacobian
    mode to a variant that was built without it."""
    import json
    r = subprocess.run([sys.executable, "-c", _RTC_CHILD], env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["compiled"] == 4 and d["prebuilt"] != 4          # the second run launched the instantiation compiled for the configuration
    assert d["same"], d
    assert d["numeric"] == "refused"
