"""Every pre-built instantiation of the optimise kernel is launched once (tools/launch_every_instantiation.py, one process per layout:
a GPU memory fault aborts only that process and its last line says which case). ADVICE r05: the miscompile of the out-of-line solve call
(profiles/fault_bisect_r05.txt) moves between translation units with their register allocation; a unit nobody launches in the tests
would fault on a robot first. The fault is deterministic at the first solve of a launch, so launching every unit once - on the
configuration paths that faulted before (cost exponent != 1, shortest path, via-points) - turns it into a failed test of that build.
The safe calling conventions were measured and cost + 0.9 % time and 5 x the fabric traffic (IPRA off) or + 3.2 % (plain convention) on
the headline (teb_local_planner_amd/build.py above HIPCC_FLAGS): the default kinds keep the cheap call, this test is the net."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNREACHABLE = {}   # (the band-in-HBM layout's point-like small-batch kinds cannot be launched and are no longer built: tools/launch_every_instantiation.py)


@pytest.mark.parametrize("layout", ["band", "blocks", "bandg"])
def test_every_prebuilt_instantiation_of_the_layout_launches(layout):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "launch_every_instantiation.py"), layout], cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0, "layout %s: rc %d after: %s | %s" % (layout, r.returncode, (r.stdout.strip().splitlines() or ["-"])[-1], r.stderr[-400:])
    assert "BAD RESULT" not in r.stdout
    m = re.search(r"launched (\d+) instantiations; not reached: (.*)$", r.stdout.strip().splitlines()[-1])
    assert m, r.stdout[-500:]
    missing = set(eval(m.group(2)))   # noqa: S307 - a list of int tuples printed by the tool above
    assert missing <= UNREACHABLE.get(layout, set()), "instantiations no case launched: %s" % sorted(missing - UNREACHABLE.get(layout, set()))
