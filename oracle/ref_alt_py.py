"""The SAME ctypes wrapper as oracle/ref_py.py over a second build of the reference's sources: oracle/_ref/libteb_ref_alt.so
(-O3, FMA contraction on, sin / cos left to the compiler: oracle/ref_shim/Makefile) - TEST INFRASTRUCTURE.

Reference-vs-reference on the same bands is the noise floor of the reference's own code under a change of nothing but the compiler's
rounding choices; oracle/refcode_compare.py (ref_vs_ref) measures it, tests/test_gpu_reference_code.py and bench.py's parity_check
hold the device's distance to the reference against it."""
import importlib.util
import os

from oracle import ref_py as _strict

_spec = importlib.util.spec_from_file_location("oracle._ref_alt_impl", _strict.__file__)
_m = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_m)   # a second instance of the wrapper module: its own library handle
_m.SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libteb_ref_alt.so")
SO = _m.SO


def available():
    return os.path.exists(SO) or os.path.isdir("/root/reference")


optimize_batch = _m.optimize_batch
optimize_teb = _m.optimize_teb
lib = _m.lib


# a third wrapper instance: the strict flags, another compiler (oracle/_ref/libteb_ref_clang.so; present where the image has clang)
_spec_c = importlib.util.spec_from_file_location("oracle._ref_clang_impl", _strict.__file__)
clang = importlib.util.module_from_spec(_spec_c)
_spec_c.loader.exec_module(clang)
clang.SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libteb_ref_clang.so")
