"""bitwise fingerprint of the optimised bands of a build (TEB_AMD_LIB=...); the same list as tests/test_gpu_bit_fingerprint.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import fingerprint_cases as F
for name in F.CASES:
    h, ms = F.fingerprint(name)
    print(name, h, "kernel %.3f ms" % ms)
