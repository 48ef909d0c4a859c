"""The reference-side C++ binding (teb_local_planner_amd/host) has its own CMake target. Where the reference
checkout and cmake exist it is configured and built stand-alone against the reference's headers (third-party APIs: the stand-in
headers of oracle/ref_shim/include) - no GPU needed: it only has to compile and link against libteb_amd.so."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference") or shutil.which("cmake") is None, reason="needs the reference checkout and cmake")
def test_host_binding_builds_with_its_own_cmake_target(tmp_path):
    from teb_local_planner_amd import build
    build.build()
    host = os.path.join(ROOT, "teb_local_planner_amd", "host")
    cfg = ["cmake", "-S", host, "-B", str(tmp_path), "-DTEB_LOCAL_PLANNER_SRC=/root/reference",
           "-DTEB_AMD_EXTRA_INCLUDE_DIRS=" + os.path.join(ROOT, "oracle", "ref_shim", "include"), "-DCMAKE_CXX_FLAGS=-w"]
    subprocess.check_call(cfg, stdout=subprocess.DEVNULL)
    subprocess.check_call(["cmake", "--build", str(tmp_path)], stdout=subprocess.DEVNULL)
    assert os.path.exists(os.path.join(str(tmp_path), "libteb_amd_backend.so"))
    # the accessor patch names every member the default build reaches through `#define private public`
    patch = open(os.path.join(host, "patches", "footprint_and_signature_accessors.patch")).read()
    for member in ("radius_", "front_offset_", "front_radius_", "rear_offset_", "rear_radius_", "line_start_", "line_end_", "vertices_",
                   "hsignature_", "hsignature3d_"):
        assert member in patch
