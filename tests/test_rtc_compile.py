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


_CHILD = r"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, %r)
from teb_local_planner_amd import planner
L = planner.lib()
L.teb_amd_debug_rtc_compile.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]
L.teb_amd_debug_rtc_cache.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_char_p, C.c_int32]
n = C.c_int64(0)
t0 = time.perf_counter()
rc = L.teb_amd_debug_rtc_compile(int(sys.argv[1]), 0, 0, 12, C.byref(n))
dt = time.perf_counter() - t0
e = C.c_int32(-1); h = C.c_int32(-1); w = C.c_int32(-1); buf = C.create_string_buffer(1024)
L.teb_amd_debug_rtc_cache(C.byref(e), C.byref(h), C.byref(w), buf, 1024)
print(json.dumps(dict(rc=rc, size=n.value, seconds=dt, embedded=e.value, hits=h.value, writes=w.value, dir=buf.value.decode(),
                      err=L.teb_amd_last_error().decode(errors="replace")[:600])))
"""


def _child(mask, env_extra):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(planner.__file__)))
    env = dict(os.environ)
    env.pop("TEB_AMD_CSRC", None)
    env["AMD_COMGR_CACHE"] = "0"          # (comgr keeps a cache of its own in recent ROCm releases: these compilations are to be real ones)
    env.update(env_extra)
    p = subprocess.run([sys.executable, "-c", _CHILD % root, str(mask)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/libhiprtc.so"), reason="no libhiprtc in this image")
def test_a_deployed_library_compiles_from_its_embedded_sources_and_the_next_process_loads_from_disk(tmp_path):
    """The library alone in a directory - no csrc/ beside it, no include/ two levels up - compiles the instantiation from the copy of the
    kernel sources it carries; the code object lands in the disk cache, a second process gets it from there in milliseconds, a damaged
    cache file is recompiled (and replaced), and a csrc directory named by $TEB_AMD_CSRC takes precedence with a key of its own."""
    import shutil
    ids, dflt = _flag_table()
    defaults = sum((1 << k) for k, i in enumerate(ids) if dflt[i])
    lib = tmp_path / "deploy" / "libteb_amd.so"
    lib.parent.mkdir()
    shutil.copy(planner.lib_path() if hasattr(planner, "lib_path") else os.path.join(os.path.dirname(os.path.abspath(planner.__file__)), "libteb_amd.so"), lib)
    cache = tmp_path / "cache"
    env = {"TEB_AMD_LIB": str(lib), "TEB_AMD_RTC_CACHE": str(cache)}
    first = _child(defaults, env)
    assert first["rc"] == 0 and first["size"] > 50000, first
    assert first["embedded"] == 1 and first["hits"] == 0 and first["writes"] == 1 and first["dir"] == str(cache), first
    files = sorted(os.listdir(cache))
    assert len(files) == 1 and files[0].endswith(".co") and os.path.getsize(cache / files[0]) > first["size"], files
    second = _child(defaults, env)
    assert second["rc"] == 0 and second["size"] == first["size"] and second["hits"] == 1 and second["writes"] == 0, second
    assert second["seconds"] < 0.25 * first["seconds"], (first["seconds"], second["seconds"])
    # a damaged file (truncated by a full disk, say) fails its check, is recompiled and replaced
    blob = open(cache / files[0], "rb").read()
    open(cache / files[0], "wb").write(blob[:len(blob) // 2])
    third = _child(defaults, env)
    assert third["rc"] == 0 and third["hits"] == 0 and third["writes"] == 1, third
    assert open(cache / files[0], "rb").read() == blob
    # other flag values: another key
    other = _child(defaults ^ (1 << ids.index("SHORTEST_PATH")), env)
    assert other["rc"] == 0 and other["hits"] == 0 and len(os.listdir(cache)) == 2, other
    # the disk cache can be switched off
    off = _child(defaults, dict(env, TEB_AMD_RTC_CACHE="off"))
    assert off["rc"] == 0 and off["hits"] == 0 and off["writes"] == 0 and off["dir"] == "", off
    # a csrc directory on request: the files are compiled, not the embedded copy
    tree = _child(defaults, dict(env, TEB_AMD_CSRC=CSRC))
    assert tree["rc"] == 0 and tree["embedded"] == 0 and tree["size"] == first["size"], tree
