"""The run-time compilation path (csrc/teb_rtc.hpp, teb_amd_options_t::compile_for_config) without a GPU: hipRTC cross-compiles for
gfx950, so the library can be asked to produce the instantiation for a set of flag values here. What this holds: the kernel sources
compile under hipRTC (its built-in headers differ from hipcc's: M_PI / HUGE_VAL are passed as options), for every *_CUSTOM scene kind,
with flags folded to the defaults and to their opposites, in both Jacobian modes; and a second request is served from the cache."""
import ctypes as C
import os
import re
import time

import pytest

from teb_local_planner_amd import planner

CSRC = os.path.join(os.path.dirname(os.path.abspath(planner.__file__)), "csrc")


def _flag_table():
    dev = open(os.path.join(CSRC, "teb_device.hpp")).read()
    ids = re.findall(r"X\((\w+)\)", dev[dev.index("#define TEB_PF_ALL(X)"):dev.index("#define TEB_PF_EXPR_EXACT_ARC")])
    dflt = {i: re.search(r"#define TEB_PF_DFLT_%s (\w+)" % i, dev).group(1) == "true" for i in ids}
    return ids, dflt


def _compile(mask, solver, jmode, kind):
    L = planner.lib()
    L.teb_amd_debug_rtc_compile.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]
    n = C.c_int64(0)
    rc = L.teb_amd_debug_rtc_compile(mask, solver, jmode, kind, C.byref(n))
    return rc, n.value, L.teb_amd_last_error().decode(errors="replace")


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/libhiprtc.so"), reason="no libhiprtc in this image")
def test_the_kernel_sources_compile_under_hiprtc_for_any_flag_values():
    ids, dflt = _flag_table()
    assert len(ids) == 20
    defaults = sum((1 << k) for k, i in enumerate(ids) if dflt[i])
    opposite = defaults ^ ((1 << len(ids)) - 1)
    # debug export and the sequential LDL^T stay off in the "opposite" set (they are launch options, not planner configurations)
    for name in ("DEBUG_LINEARIZE", "BAND_LDLT"):
        opposite &= ~(1 << ids.index(name))
    t0 = time.perf_counter()
    rc, size, err = _compile(defaults, 0, 0, 12)                  # point-like scene, band layout, closed forms, defaults
    first = time.perf_counter() - t0
    assert rc == 0 and size > 50000, (rc, size, err[:800])
    t0 = time.perf_counter()
    rc, size2, err = _compile(defaults, 0, 0, 12)
    assert rc == 0 and size2 == size and time.perf_counter() - t0 < 0.25 * first   # served from the cache
    for mask, solver, jmode, kind in ((opposite, 1, 0, 13),        # everything the other way round, blocks layout, small batch with helpers
                                      (opposite, 2, 1, 12),        # band in HBM, the reference's numeric linearisation
                                      (defaults, 0, 0, 15)):       # generic shapes, small batch with distance + solver helpers
        rc, size, err = _compile(mask, solver, jmode, kind)
        assert rc == 0 and size > 50000, (mask, solver, jmode, kind, rc, err[:800])
    assert _compile(defaults, 0, 0, 4)[0] != 0                     # only the *_CUSTOM kinds are compiled at run time
