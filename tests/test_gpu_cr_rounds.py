"""The inline-asm DPP rounds of the cyclic reduction (csrc/teb_kernel.hpp: cr_forward_round16, cr16_eliminate, TEB_CR16_*) against the
retained 8-lane LDS-operand round of rounds 2 - 4 (tools/micro/cr_round_bench.hip): every entry a later level or the back substitution
reads must carry the same bits, at level sizes either side of every boundary of the round structure (E % 16 != 0, the PAIR 1 / 2 boundary).
ADVICE r05: those statements sit outside the compiler's hazard recogniser and reproduce one compiler's FMA contraction by hand; this is
the assertion on those assumptions. The binary is built by __graft_entry__.build() (build.py: build_micro)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "micro", "cr_round_bench")


def test_dpp_rounds_equal_the_8_lane_rounds_bit_for_bit():
    assert os.path.exists(BIN), "tools/micro/cr_round_bench is missing: __graft_entry__.build() builds it (teb_local_planner_amd/build.py: build_micro)"
    r = subprocess.run([BIN, "--compare"], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0 and "rounds identical" in r.stdout, r.stdout[-2000:] + r.stderr[-500:]
    assert r.stdout.count("0 of ") == 16, r.stdout
