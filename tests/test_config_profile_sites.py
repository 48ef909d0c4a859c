"""Host logic of the kernels specialised on the TebConfig defaults: every configuration field a device source folds with TEB_CFG /
TEB_KIN_CFG (csrc/teb_device.hpp) must be checked by config_matches_defaults_profile / launch_opt in csrc/teb_amd.hip before such a kernel
is launched - a fold without its host-side condition would silently compute a different cost function."""
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "teb_local_planner_amd", "csrc")


def _macro_args(text, name):
    """first arguments of every NAME( .. , .. ) invocation (balanced parentheses)"""
    out = []
    for m in re.finditer(r"\b%s\(" % name, text):
        depth, i, start = 1, m.end(), m.end()
        first_end = None
        while depth:
            ch = text[i]
            if ch == "(":
                depth += 1
            elif ch == ")":
                depth -= 1
            elif ch == "," and depth == 1 and first_end is None:
                first_end = i
            i += 1
        out.append(text[start:first_end])
    return out


def test_every_folded_flag_has_its_host_side_condition():
    folded = set()
    sites = 0
    for f in ("teb_kernel.hpp", "teb_edges.hpp"):
        text = open(os.path.join(CSRC, f)).read()
        text = re.sub(r"#define TEB_(KIN_)?CFG\(.*", "", text)
        for name in ("TEB_CFG", "TEB_KIN_CFG"):
            for arg in _macro_args(text, name):
                sites += 1
                folded.update(re.findall(r"\b(?:c|args|sc)\.(\w+)", arg))
    assert sites >= 20, sites
    host = open(os.path.join(CSRC, "teb_amd.hip")).read()
    body = host[host.index("bool config_matches_defaults_profile"):host.index("if (!k) k = opt_kernel")]
    checked = set(re.findall(r"\bc\.(\w+)", body)) | set(re.findall(r"\bh->(?:opt\.)?(\w+)", body)) | set(re.findall(r"\ba\.(\w+)", body))
    # the kinematics flags are folded in the point-like kinds only, and checked for them only (scene_part); fields that appear under
    # another name on the host: the via-points (nvia), the near-mask switch (opt.no_near_cache), the footprint (fast_points)
    alias = {"include_dynamic_obstacles": "weight_obstacle",   # without include_dynamic_obstacles the dynamic list is empty: only weight_obstacle decides
             "acc_lim_y": "max_vel_y"}                       # max_vel_y == 0 alone makes the acceleration edges non-holonomic
    missing = sorted(f for f in folded if alias.get(f, f) not in checked)
    assert not missing, "folded on the device but not checked on the host: %s" % missing
    for must in ("nvia", "generic_config_path", "fast_points", "static_radius_zero", "band_ldlt", "debug_linearize", "no_near_cache"):
        assert must in checked, must
