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


def _run(name, timeout=120):
    b = os.path.join(ROOT, "tools", "micro", name)
    assert os.path.exists(b), "tools/micro/%s is missing: __graft_entry__.build() builds it (teb_local_planner_amd/build.py: build_micro)" % name
    r = subprocess.run([b], capture_output=True, text=True, timeout=timeout)
    print(r.stdout)
    return r


def test_level0_dpp_products_equal_the_shuffled_sums_bit_for_bit():
    """Level 0 of the hybrid solve exchanges the operands of its Schur products by bank-masked 64-bit DPP (csrc/teb_kernel.hpp:
    TEB_L0_O1_DPP .. TEB_L0_O3_DPP): 294 instructions of inline asm whose ORDER avoids a hardware hazard. Same sums through __shfl."""
    r = _run("l0_dpp_probe")
    assert r.returncode == 0 and "0 of 6144 entries differ" in r.stdout, r.stdout[-2000:] + r.stderr[-500:]


def test_bank_masked_dpp64_patterns_the_kernel_relies_on():
    """tools/micro/dpp64_mask_probe.hip: masked-off lanes keep their accumulator; one instruction between two DPP fmacs on the same
    accumulator is enough; masked moves back to back, a plain read after / a plain write before a masked fmac are correct."""
    r = _run("dpp64_mask_probe")
    assert r.returncode == 0 and "relied-upon patterns: 0 wrong values" in r.stdout, r.stdout[-2000:] + r.stderr[-500:]
