"""Host logic of the kernels specialised on the configuration (csrc/teb_device.hpp: the profile table TEB_PF_*). A device source folds a
flag by writing TEB_CFGI(ID); the host's profile_matches() (csrc/teb_amd.hip) is generated from TEB_PF_ALL over the same table. What can
still go wrong is checked here, without a GPU: an id used at a device site but missing from TEB_PF_ALL (its host-side condition would
never be evaluated - a fold without its check silently computes another cost function), a table entry without its columns, and a
hand-written fold that bypasses the table."""
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "teb_local_planner_amd", "csrc")


def _read(f):
    return open(os.path.join(CSRC, f)).read()


def test_every_folded_flag_is_in_the_table_the_host_check_is_generated_from():
    dev = _read("teb_device.hpp")
    listed = re.findall(r"X\((\w+)\)", dev[dev.index("#define TEB_PF_ALL(X)"):dev.index("#define TEB_PF_EXPR_EXACT_ARC")])
    assert len(listed) >= 20 and len(set(listed)) == len(listed), listed
    used = set()
    sites = 0
    for f in ("teb_kernel.hpp", "teb_edges.hpp", "teb_geometry.hpp", "teb_multicu.hpp", "teb_autoresize_chain.hpp"):
        ids = re.findall(r"\bTEB_CFGI\((\w+)\)", _read(f))
        sites += len(ids)
        used.update(ids)
    assert sites >= 25, sites
    assert used <= set(listed), "folded at a device site but absent from TEB_PF_ALL: %s" % sorted(used - set(listed))
    assert set(listed) <= used, "in the table but folded nowhere: %s" % sorted(set(listed) - used)
    for ident in listed:   # every column of every entry
        for col in ("EXPR", "DFLT", "HOST", "WIDE", "LIGHT", "KIN"):
            assert re.search(r"#define TEB_PF_%s_%s\b" % (col, ident), dev), (col, ident)


def test_the_host_check_is_the_generated_one_and_nothing_folds_by_hand():
    host = _read("teb_amd.hip")
    body = host[host.index("int profile_matches("):host.index("hipError_t launch_opt(")]
    assert "TEB_PF_ALL(TEB_PF_CHECK)" in body and "TEB_PF_HOST_##ID" in body and "TEB_PF_WIDE_##ID" in body and "TEB_PF_LIGHT_##ID" in body and "TEB_PF_KIN_##ID" in body
    assert "generic_config_path" in body
    assert "config_matches_defaults_profile" not in host          # the hand-kept mirror of rounds 3 is gone
    for f in ("teb_kernel.hpp", "teb_edges.hpp"):                  # no fold outside the table
        text = _read(f)
        assert not re.search(r"\bTEB_(KIN_)?CFG\(", text), f
    # the only other use of the build flag: the branch-free footprint radius (no configuration flag is folded there)
    edges = _read("teb_edges.hpp")
    assert edges.count("TEB_AMD_DEFAULTS_PROFILE") == 1 and "footprint_radius : 0.0" in edges
