"""Instruction statistics of one kernel instantiation, section by section (no GPU needed): the method behind round 6's speed-up. At one
wave per SIMD every wave instruction costs ~ 5 cycles whatever it is (tools/micro/dpp_rate_bench.hip, icache_bench.hip), so what is not
arithmetic is overhead one can count: spill traffic (v_readlane / v_writelane / v_accvgpr_*), selects (v_cndmask + v_cmp), exec-mask
branches (s_and_saveexec), vector integer address arithmetic, zero-fill moves, s_nop.
usage: python tools/isa_stats.py [solver jmode scene]      (default 0 0 4: the headline's kernel)
Compiles csrc/teb_opt_inst.hip to gfx950 assembly (hipcc --offload-device-only -S, ~ 40 s) and prints, for the optimise kernel and every
out-of-line function it calls, the instruction mix and the mix of every barrier-to-barrier section."""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from teb_local_planner_amd import build as B   # noqa: E402


def classify(op):
    if op.startswith(("v_readlane", "v_writelane", "v_accvgpr", "scratch_")): return "spill"
    if op.startswith(("v_cndmask", "v_cmp")): return "select/compare"
    if op.startswith(("s_and_saveexec", "s_or_saveexec", "s_andn2_saveexec", "s_cbranch", "s_branch")): return "branch"
    if op in ("s_nop", "s_waitcnt", "s_barrier"): return op
    if op.startswith("s_"): return "scalar"
    if op.startswith(("ds_", "global_", "flat_", "buffer_")): return "memory"
    if op.startswith("v_mov"): return "move"
    if "f64" in op or op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_div", "v_ldexp", "v_frexp", "v_trig", "v_floor", "v_fract")): return "fp64"
    return "vector int/other"


def main():
    sv, jm, sk = (sys.argv[1:4] + ["0", "0", "4"])[:3] if len(sys.argv) >= 4 else ("0", "0", "4")
    obj = "opt_%s_%s_%s.o" % (sv, jm, sk)
    out = os.path.join(tempfile.mkdtemp(), "k.s")
    cmd = ([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + B.HIPCC_FLAGS + list(B.UNIT_FLAGS.get(obj, [])) +
           ["-DTEB_INST_SOLVER=" + sv, "-DTEB_INST_JMODE=" + jm, "-DTEB_INST_SCENE=" + sk, "--offload-device-only", "-S",
            os.path.join(B.CSRC, "teb_opt_inst.hip"), "-o", out, "-w"])
    subprocess.check_call(cmd, cwd=B.CSRC)
    lines = open(out).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for (i0, name) in starts:
        end = next(i for i in range(i0, len(lines)) if lines[i].strip().startswith(".size"))
        ins = [l.strip() for l in lines[i0:end] if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
        ins = [x.split(";")[0].strip() for x in ins]
        short = re.sub(r"^_ZN6tebamd(\d+)", "", name)[:60]
        tot = collections.Counter(classify(x.split()[0]) for x in ins)
        print("== %s: %d instructions | %s" % (short, len(ins), "  ".join("%s %d" % kv for kv in tot.most_common())))
        bars = [k for k, x in enumerate(ins) if x.startswith("s_barrier")]
        for a, z in zip([0] + bars, bars + [len(ins)]):
            if z - a < 200:
                continue
            c = collections.Counter(classify(x.split()[0]) for x in ins[a:z])
            print("   section %6d .. %6d (%5d): %s" % (a, z, z - a, "  ".join("%s %d" % kv for kv in c.most_common())))


if __name__ == "__main__":
    main()
