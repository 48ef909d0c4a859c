"""Builds libteb_amd.so (hipcc, gfx950 only) in-tree so it travels with the repo snapshot to the GPU box.

The host side (csrc/teb_amd.hip: C-ABI, small kernels) and every instantiation of the optimise kernel (csrc/teb_opt_inst.hip with
-DTEB_INST_SOLVER / _JMODE / _SCENE: 3 layouts x 2 Jacobian modes x 2 scene kinds) are separate translation units compiled in
parallel into build/<variant>/*.o and linked into one shared library. Variants:
    product : libteb_amd.so
    mfma    : libteb_amd_mfma.so  (-DTEB_AMD_MFMA_SCHUR -DTEB_AMD_ANALYTIC_ONLY: the Schur update of the cyclic reduction on
              v_mfma_f64_16x16x4_f64 - SURVEY section 8 row g; exercised by tests/test_gpu_mfma_build.py)"""
import concurrent.futures
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libteb_amd.so")
LIB_MFMA = os.path.join(HERE, "libteb_amd_mfma.so")
HEADERS = ["teb_device.hpp", "teb_comm.hpp", "teb_feasibility.hpp", "teb_geometry.hpp", "teb_edges.hpp", "teb_kernel.hpp", "teb_strip.hpp",
           "teb_hsig.hpp", "teb_graph.hpp", "teb_opt_launch.hpp", "teb_multicu.hpp", "teb_autoresize_chain.hpp", "teb_rtc.hpp",
           os.path.join("..", "..", "include", "teb_amd.h"), os.path.join("..", "..", "include", "teb_amd_debug.h")]

# -ffp-contract=off: the parity contract is against a plain IEEE mul/add restatement of the reference;
# letting the compiler fuse a*b+c would change which side of a penalty kink borderline residuals fall on.
# The calling convention of the out-of-line damped solve (round 6, ADVICE r05 / VERDICT r05 item 8). The solve is called on a
# no-callee-saved convention (`not_tail_called` internal function: LLVM skips its callee-saved block and hands the callers its clobber
# mask through interprocedural register allocation), and that combination miscompiled units whose register allocation happened to
# move - a memory aperture violation at the FIRST solve of a launch (profiles/fault_bisect_r05.txt). The two safe conventions were
# measured on every unit (MI355X, same box, alternating; profiles/ab_threads_ipra_r06.txt and the round-6 profile taken with IPRA off):
#   -mllvm -enable-ipra=0 everywhere : headline + 0.9 .. 1.3 % time - and the callee's save / restore of ~ 170 registers per call shows up
#                                      as 4.63 GB of fabric traffic per launch instead of 0.91 GB (scratch 336 -> 656 B per lane);
#   -DTEB_AMD_SOLVE_CSR everywhere   : headline + 3.2 %.
# Neither is within the 0.5 % the verdict set for shipping it, so the kinds every default user launches (defaults / wide / generic-shape
# defaults) keep the cheap call, the kinds off the defaults keep -DTEB_AMD_SOLVE_CSR (UNIT_FLAGS below), and the net under all of them
# is tests/test_gpu_every_instantiation.py: EVERY pre-built (layout, Jacobian mode, kind) is launched once, on the configuration paths
# that faulted before - the fault is deterministic at the first solve, so a unit whose register allocation moved into it fails the
# GPU test suite of its build, not a robot. Kernels compiled at run time take the plain convention (csrc/teb_rtc.hpp).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]

# Per-unit flags of the product. -DTEB_AMD_POINTS_KEEP_GENERIC on the point-like band-layout instantiation (the headline's kernel): it
# branches on SceneDev::fast_points at run time and so keeps the generic-shape code it never executes. Measured on MI355X, same box,
# alternating libraries, 11 launches each, +-0.3 %: headline launch 3.71 ms without the flag, 3.62 ms with it, 3.59 ms for round 2's
# single kernel (which carried that code for every scene). What changes is the register allocation / placement of the hot loops, not
# the work; no compiler flag tried (scheduler strategies, -O2, -Os) moves it. Kept because the headline is what is measured.
# -DTEB_AMD_INLINE_SOLVE on the small-batch, blocks-in-LDS instantiation specialised on the TebConfig defaults (C2 / C3 / the planner
# tick run it): the specialised kernels are small enough (~ 90 KB) for the solve to be inlined there without the spills that made it
# 21 % slower in the generic kernel - C2 1.36 -> 1.31 ms, C3 1.57 -> 1.48 (same box, alternating); in the band layout and in the
# full-batch kinds it still loses (headline 2.59 -> 3.08 ms), so those keep the call.
# -DTEB_AMD_SOLVE_CSR (the out-of-line solve on the plain calling convention: the callee saves its callee-saved block) on every kind that
# keeps the cost terms at run time - generic kinds 0 .. 3, light kinds 10, 11. Without it some of these big kernels fault (memory aperture
# violation at the first LM iteration): round 4 met three of them, in round 5's tree it is the full-batch light kind of the band layout
# (opt_0_0_10). Bisected on MI355X (profiles/fault_bisect_r05.txt, tools/fault_probe.py): not pow() (inlined), not the data (one band, one
# iteration), not the stack size; `-mllvm -enable-ipra=0` on the ONE unit cures it like this flag does - the no-callee-saved call relies on
# LLVM's interprocedural register allocation handing the caller the callee's clobber set, and that combination is what miscompiles units of
# this size (which units: moves with their register allocation). The defaults / wide kinds keep the cheaper call (cost of the safe
# conventions and the test that guards them: above HIPCC_FLAGS). Kernels compiled at run time take the plain convention too (csrc/teb_rtc.hpp).
UNIT_FLAGS = {"opt_0_0_0.o": ["-DTEB_AMD_POINTS_KEEP_GENERIC"], "opt_1_0_5.o": ["-DTEB_AMD_INLINE_SOLVE"]}
# The instantiations that keep EVERY cost term at run time (generic kinds 0 .. 3, light kinds 10, 11) call the solve on the plain
# convention (-DTEB_AMD_SOLVE_CSR): see the note above UNIT_FLAGS.
for _sv in (0, 1, 2):
    for _jm in (0, 1):
        for _sk in (0, 1, 2, 3, 10, 11):
            UNIT_FLAGS.setdefault("opt_%d_%d_%d.o" % (_sv, _jm, _sk), []).append("-DTEB_AMD_SOLVE_CSR")

VARIANTS = {
    "product": dict(lib=LIB, defines=[], jmodes=(0, 1)),
    "mfma": dict(lib=LIB_MFMA, defines=["-DTEB_AMD_MFMA_SCHUR", "-DTEB_AMD_ANALYTIC_ONLY", "-DTEB_AMD_NO_DEFAULTS_TWINS"], jmodes=(0,)),
    # closed-form Jacobians only: the quick build the tools/ A/B experiments use (build(variant="analytic", extra_defines=[..], out=..))
    "analytic": dict(lib=os.path.join(HERE, "..", "tools", "libteb_amd_ar.so"), defines=["-DTEB_AMD_ANALYTIC_ONLY"], jmodes=(0,)),
    # kernel experiments (tools/ab.sh): closed-form Jacobians, and only the instantiations the measured configurations launch are real -
    # the others are stubs that return a null kernel address (teb_opt_inst.hip, -DTEB_INST_STUB), so the launch fails loudly if reached
    "exp": dict(lib=os.path.join(HERE, "..", "tools", "libteb_amd_exp.so"), defines=["-DTEB_AMD_ANALYTIC_ONLY"], jmodes=(0,),
                only=("opt_0_0_4.o", "opt_1_0_4.o", "opt_0_0_5.o", "opt_1_0_5.o", "opt_0_0_7.o", "opt_1_0_7.o")),
}
PRODUCT_VARIANTS = ("product", "mfma")


def _units(variant, only=None):
    """(object name, source, extra defines) of every translation unit of the variant."""
    v = dict(VARIANTS[variant])
    if only is not None:
        v["only"] = tuple(only)
    units = [("teb_amd.o", "teb_amd.hip", [])]
    for jm in v["jmodes"]:
        for sv in (0, 1, 2):
            # closed-form Jacobians: + the small-batch scene kinds (helper workgroups) and, unless the variant opts out, the point-like
            # kinds specialised on the TebConfig defaults (4, 5: teb_device.hpp, TEB_CFG)
            twins = "-DTEB_AMD_NO_DEFAULTS_TWINS" not in v["defines"]
            # (8, 9: the point-like kinds with every fold but the via-points and the holonomic choice, TEB_PF_WIDE_* in teb_device.hpp;
            #  10, 11: every cost-term flag at run time, TEB_PF_LIGHT_*)
            for sk in (((0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11) if twins else (0, 1, 2, 3)) if jm == 0 else ((0, 1, 4) if twins else (0, 1))):
                if sv == 2 and sk in (2, 5, 9, 11):
                    continue   # band in HBM has no solver helpers: its point-like small-batch kinds cannot be launched (teb_opt_launch.hpp)
                name = "opt_%d_%d_%d.o" % (sv, jm, sk)
                stub = ["-DTEB_INST_STUB"] if ("only" in v and name not in v["only"]) else []
                units.append((name, "teb_opt_inst.hip",
                              ["-DTEB_INST_SOLVER=%d" % sv, "-DTEB_INST_JMODE=%d" % jm, "-DTEB_INST_SCENE=%d" % sk] + stub))
    return units


# what a kernel instantiation is made of (the host translation unit depends on everything)
KERNEL_DEPS = ["teb_opt_inst.hip", "teb_device.hpp", "teb_geometry.hpp", "teb_edges.hpp", "teb_kernel.hpp", "teb_opt_launch.hpp", "teb_multicu.hpp", "teb_autoresize_chain.hpp",
               os.path.join("..", "..", "include", "teb_amd.h")]


def _newest_source(files=None):
    t = 0.0
    for f in (files or ["teb_amd.hip", "teb_opt_inst.hip"] + HEADERS):
        p = os.path.join(CSRC, f)
        if os.path.exists(p):
            t = max(t, os.path.getmtime(p))
    return max(t, os.path.getmtime(os.path.abspath(__file__)))


def _stale(path):
    return not os.path.exists(path) or os.path.getmtime(path) < _newest_source()


def source_hash():
    """sha256 over the device + host sources of the library (what a committed profile is tied to)."""
    h = hashlib.sha256()
    for f in sorted(["teb_amd.hip", "teb_opt_inst.hip"] + HEADERS):
        p = os.path.join(CSRC, f)
        if os.path.exists(p):
            h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def kernel_hash(variant_defines=(), unit_flags=None):
    """sha256 over the translation unit of the optimise kernel (teb_opt_inst.hip and what it includes) with comments and white space
    stripped - a comment edit does not make a new binary - + the compiler flags of the units and of the variant. The host side
    (teb_amd.hip) and the kernels of the rows either side of the path do not change the profiled kernel. What a committed rocprof summary
    is tied to: bench.py computes it on the tree (config.source_hash), build() embeds it in the binary (config.binary_hash,
    teb_amd_debug_build_info), tools/export_profile.py records both."""
    import re
    h = hashlib.sha256()
    kernel_files = ("teb_autoresize_chain.hpp", "teb_device.hpp", "teb_edges.hpp", "teb_geometry.hpp", "teb_kernel.hpp", "teb_multicu.hpp", "teb_opt_inst.hip", "teb_opt_launch.hpp")
    h.update(repr(sorted((UNIT_FLAGS if unit_flags is None else unit_flags).items())).encode() + repr(HIPCC_FLAGS).encode())
    if variant_defines:
        h.update(repr(list(variant_defines)).encode())
    for f in sorted(os.listdir(CSRC)):
        if f in kernel_files:
            src = open(os.path.join(CSRC, f), "r", errors="replace").read()
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)      # block comments
            src = re.sub(r"//[^\n]*", "", src)                   # line comments (no string literal of the sources holds "//")
            h.update(f.encode())
            h.update("".join(src.split()).encode())
    return h.hexdigest()[:16]


# what the optimise kernel's translation unit includes: embedded into the library for teb_amd_options_t::compile_for_config, so that a
# deployed libteb_amd.so compiles the instantiation of a configuration without the source tree beside it
RTC_SOURCES = ["teb_kernel.hpp", "teb_edges.hpp", "teb_geometry.hpp", "teb_device.hpp", "teb_multicu.hpp", "teb_autoresize_chain.hpp",
               os.path.join("..", "..", "include", "teb_amd.h")]


def write_rtc_sources(bdir):
    """<bdir>/teb_rtc_embedded.inc: the kernel sources as string literals (name, text) + their hash, included by csrc/teb_rtc.hpp under
    -DTEB_AMD_RTC_EMBEDDED. teb_device.hpp's include of the C-ABI header is rewritten to the flat name the in-memory header gets."""
    parts, names = [], []
    h = hashlib.sha256()
    for f in RTC_SOURCES:
        text = open(os.path.join(CSRC, f), "r").read()
        text = text.replace('#include "../../include/teb_amd.h"', '#include "teb_amd.h"')
        assert ')TEBSRC"' not in text
        h.update(text.encode())
        names.append(os.path.basename(f))
        # (string literals are split: one literal per 8 KB keeps every compiler's limit far away; adjacent literals concatenate)
        chunks = [text[k:k + 8192] for k in range(0, len(text), 8192)]
        parts.append("\n".join('R"TEBSRC(%s)TEBSRC"' % c for c in chunks))
    out = os.path.join(bdir, "teb_rtc_embedded.inc")
    body = ("// generated by build.py (write_rtc_sources): do not edit\n"
            "static const int kRtcEmbeddedCount = %d;\n" % len(names)
            + "static const char* const kRtcEmbeddedNames[] = {%s};\n" % ", ".join('"%s"' % n for n in names)
            + "static const char* const kRtcEmbeddedSources[] = {\n%s\n};\n" % ",\n".join(parts)
            + 'static const char kRtcEmbeddedHash[] = "%s";\n' % h.hexdigest()[:16])
    if not os.path.exists(out) or open(out).read() != body:
        with open(out, "w") as fh:
            fh.write(body)
    return out


def build(force=False, verbose=False, variant="product", jobs=None, extra_defines=(), out=None, unit_flags=None, only=None):
    """Compile the variant if its sources are newer than the library. Returns the library path.
    unit_flags: {object name: [extra compiler flags]} for single translation units (compiler-flag experiments of tools/).
    only: object names of the instantiations to build for real, the others become stubs (default: the variant's own list, if it has one)."""
    v = VARIANTS[variant]
    lib = os.path.abspath(out or v["lib"])
    if not (force or _stale(lib)):
        return lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    unit_flags = dict(UNIT_FLAGS, **(unit_flags or {})) if variant in PRODUCT_VARIANTS else (unit_flags or {})
    salt = " ".join(extra_defines) + "".join("|%s:%s" % (k, " ".join(v)) for k, v in sorted(unit_flags.items())) + ("|only:" + ",".join(only) if only is not None else "")
    tag = variant if not salt else variant + "_" + hashlib.sha256(salt.encode()).hexdigest()[:8]
    bdir = os.path.join(HERE, "build", tag)
    os.makedirs(bdir, exist_ok=True)
    jobs = jobs or int(os.environ.get("TEB_AMD_BUILD_JOBS", "0")) or min(os.cpu_count() or 1, 8)
    newest_all, newest_kernel = _newest_source(), _newest_source(KERNEL_DEPS)

    embedded = write_rtc_sources(bdir)
    src_hash, k_hash = source_hash(), kernel_hash(list(v["defines"]) + list(extra_defines), unit_flags)

    def compile_unit(u):
        obj, src, defs = u
        o = os.path.join(bdir, obj)
        newest = newest_kernel if src == "teb_opt_inst.hip" else newest_all
        if not force and os.path.exists(o) and os.path.getmtime(o) >= newest:
            return o
        if src == "teb_amd.hip":   # the host side carries the kernel sources for the run-time compiler (teb_rtc.hpp) and the variant's defines
            defs = defs + ["-DTEB_AMD_RTC_EMBEDDED", "-I" + os.path.dirname(embedded),
                           '-DTEB_AMD_VARIANT_DEFINES="%s"' % " ".join(list(v["defines"]) + list(extra_defines)),
                           '-DTEB_AMD_BUILD_SOURCE_HASH="%s"' % src_hash, '-DTEB_AMD_BUILD_KERNEL_HASH="%s"' % k_hash]   # read back by teb_amd_debug_build_info: ties a profile to the BINARY
        cmd = [hipcc] + HIPCC_FLAGS + v["defines"] + list(extra_defines) + defs + list(unit_flags.get(obj, [])) + ["-c", os.path.join(CSRC, src), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)
        return o

    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(compile_unit, _units(variant, only)))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return lib


MICRO = os.path.join(HERE, "..", "tools", "micro")


def build_micro(force=False):
    """The micro-benchmarks the GPU tests run (no compiler run belongs on the GPU box): tools/micro/cr_round_bench - the reduction rounds of
    the product against the retained 8-lane round, bit for bit; l0_dpp_probe - the DPP statements of level 0 of the hybrid solve against
    the same sums through __shfl; dpp64_mask_probe - the bank-masked 64-bit DPP patterns those statements rely on
    (tests/test_gpu_cr_rounds.py). Returns the first binary."""
    outs = []
    for name, with_kernel in (("cr_round_bench", True), ("l0_dpp_probe", True), ("dpp64_mask_probe", False)):
        src, out = os.path.join(MICRO, name + ".hip"), os.path.abspath(os.path.join(MICRO, name))
        newest = max(os.path.getmtime(src), _newest_source(KERNEL_DEPS)) if with_kernel else os.path.getmtime(src)
        if force or not os.path.exists(out) or os.path.getmtime(out) < newest:
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            subprocess.check_call([hipcc] + [f for f in HIPCC_FLAGS if f != "-fPIC"] + ["-w", "-I", CSRC, src, "-o", out])
        outs.append(out)
    return outs[0]


def build_all(force=False, verbose=False):
    return [build(force=force, verbose=verbose, variant=k) for k in PRODUCT_VARIANTS]


if __name__ == "__main__":
    import sys
    names = [a for a in sys.argv[1:] if a in VARIANTS] or list(PRODUCT_VARIANTS)
    for k in names:
        print(build(force="--force" in sys.argv, verbose=True, variant=k))
