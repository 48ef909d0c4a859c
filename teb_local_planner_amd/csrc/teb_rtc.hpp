// teb_rtc.hpp — the optimise kernel compiled FOR the handle's configuration at run time (round 4).
//
// The kernels that ship in libteb_amd.so fold the configuration flags of the profile table (teb_device.hpp: TEB_PF_*) to the values of a
// default TebConfig (*_DEFAULTS), or leave some of them (*_WIDE, *_LIGHT) or all of them (generic) at run time. A configuration off the
// defaults therefore runs a slower kernel than a default one although its flags are just as constant for the life of the planner.
// With teb_amd_options_t::compile_for_config the library compiles the instantiation it needs itself: the table says which -D flags
// (TEB_PF_VALUE_<ID> = the flag's value in THIS configuration: every TEB_CFGI site folds to it), hipRTC compiles csrc/teb_kernel.hpp with
// the flags of the build (gfx950, -O3, -ffp-contract=off: same operations in the same order, bit-identical bands), the code object is
// loaded as a module and cached per (flag values, layout, Jacobian mode, scene kind) for the life of the process. One instantiation
// takes ~ 5 s to compile (the offline build spends most of its minute per unit elsewhere); asynchronously the launches run the best
// pre-built kernel until the module is ready.
//
// libhiprtc is loaded with dlopen on first use (users who leave the option off neither link nor load it). The kernel sources travel INSIDE
// the library (build.py embeds them as string literals, -DTEB_AMD_RTC_EMBEDDED: a deployed libteb_amd.so needs no source tree beside it;
// $TEB_AMD_CSRC, or a build without the embedded copy, reads them from a csrc directory instead), hip/hip_runtime.h comes from the ROCm
// installation. Compiled code objects are kept on disk across processes ($TEB_AMD_RTC_CACHE, default ~/.cache/teb_amd; "off" disables):
// keyed by the sources' hash, the flag values, the instantiation and the compiler's version, written atomically, checked on load.
#pragma once
#include <dlfcn.h>
#include <limits.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace tebamd {

struct RtcApi {
  typedef struct RtcProgramOpaque* program_t;
  int (*CreateProgram)(program_t*, const char*, const char*, int, const char**, const char**) = nullptr;
  int (*DestroyProgram)(program_t*) = nullptr;
  int (*AddNameExpression)(program_t, const char*) = nullptr;
  int (*CompileProgram)(program_t, int, const char**) = nullptr;
  int (*GetProgramLogSize)(program_t, size_t*) = nullptr;
  int (*GetProgramLog)(program_t, char*) = nullptr;
  int (*GetLoweredName)(program_t, const char*, const char**) = nullptr;
  int (*GetCodeSize)(program_t, size_t*) = nullptr;
  int (*GetCode)(program_t, char*) = nullptr;
  int (*Version)(int*, int*) = nullptr;
  std::string error;
  std::mutex mu;
  bool ready = false, failed = false;
  bool load() {
    std::lock_guard<std::mutex> lock(mu);
    if (ready) return true;
    if (failed) return false;
    const char* names[] = {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
    void* l = nullptr;
    for (const char* n : names) if ((l = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!l) { const char* e = dlerror(); error = std::string("libhiprtc not found: ") + (e ? e : "?"); failed = true; return false; }
    bool ok = true;
    auto sym = [&](const char* s) { void* p = dlsym(l, s); if (!p) { error = std::string("libhiprtc lacks ") + s; ok = false; } return p; };
    CreateProgram = reinterpret_cast<decltype(CreateProgram)>(sym("hiprtcCreateProgram"));
    DestroyProgram = reinterpret_cast<decltype(DestroyProgram)>(sym("hiprtcDestroyProgram"));
    AddNameExpression = reinterpret_cast<decltype(AddNameExpression)>(sym("hiprtcAddNameExpression"));
    CompileProgram = reinterpret_cast<decltype(CompileProgram)>(sym("hiprtcCompileProgram"));
    GetProgramLogSize = reinterpret_cast<decltype(GetProgramLogSize)>(sym("hiprtcGetProgramLogSize"));
    GetProgramLog = reinterpret_cast<decltype(GetProgramLog)>(sym("hiprtcGetProgramLog"));
    GetLoweredName = reinterpret_cast<decltype(GetLoweredName)>(sym("hiprtcGetLoweredName"));
    GetCodeSize = reinterpret_cast<decltype(GetCodeSize)>(sym("hiprtcGetCodeSize"));
    GetCode = reinterpret_cast<decltype(GetCode)>(sym("hiprtcGetCode"));
    Version = reinterpret_cast<decltype(Version)>(dlsym(l, "hiprtcVersion"));   // (optional: part of the disk-cache key)
    if (!ok) { failed = true; return false; }
    ready = true;
    return true;
  }
};
inline RtcApi& rtc_api() { static RtcApi api; return api; }

// one compiled instantiation
struct RtcKernel {
  enum State { COMPILING = 0, READY = 1, FAILED = 2 };
  std::atomic<int> state{COMPILING};
  std::vector<char> code;        // the code object (kept: a module is loaded per device context on first use)
  std::string lowered, log;
  double compile_seconds = 0;
  bool from_disk = false;        // the code object came from the disk cache of an earlier process
  std::mutex mu;                 // guards the per-device modules
  std::map<int, hipFunction_t> fn;   // device ordinal -> function
};

struct RtcKey {
  unsigned long long flags;   // bit i = value of flag i of TEB_PF_ALL in this configuration
  int solver, jmode, scene;
  bool operator<(const RtcKey& o) const {
    if (flags != o.flags) return flags < o.flags;
    if (solver != o.solver) return solver < o.solver;
    if (jmode != o.jmode) return jmode < o.jmode;
    return scene < o.scene;
  }
};

#ifdef TEB_AMD_RTC_EMBEDDED
#include "teb_rtc_embedded.inc"
#endif

inline unsigned long long rtc_fnv(const void* data, size_t n, unsigned long long h = 1469598103934665603ull) {
  const unsigned char* p = static_cast<const unsigned char*>(data);
  for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
  return h;
}

// The defines of the build variant this library is (build.py passes them to the host translation unit, e.g. "-DTEB_AMD_MFMA_SCHUR
// -DTEB_AMD_ANALYTIC_ONLY" for libteb_amd_mfma.so): a kernel compiled at run time has to be the SAME variant as the pre-built ones it
// replaces mid-run - other Schur arithmetic would change the bands when the module becomes ready - and they are part of
// the disk-cache key (two variants of the library must not share code objects).
#ifndef TEB_AMD_VARIANT_DEFINES
#define TEB_AMD_VARIANT_DEFINES ""
#endif
inline std::vector<std::string> rtc_variant_defines() {
  std::vector<std::string> out;
  const std::string all = TEB_AMD_VARIANT_DEFINES;
  size_t i = 0;
  while (i < all.size()) {
    while (i < all.size() && all[i] == ' ') ++i;
    size_t j = i;
    while (j < all.size() && all[j] != ' ') ++j;
    if (j > i) out.push_back(all.substr(i, j - i));
    i = j;
  }
  return out;
}

struct RtcCache {
  std::mutex mu;
  std::map<RtcKey, std::shared_ptr<RtcKernel>> kernels;
  // the compiler threads, joined when the library goes (exit, dlclose): a thread still inside hiprtcCompileProgram must not outlive the
  // function-local statics it uses (this object, rtc_api()). rtc_api() is constructed first, i.e. destroyed after this object.
  // A process that exits while a background compilation is in flight waits here for the rest of that hiprtcCompileProgram call;
  // teb_amd_debug_rtc_join() (include/teb_amd_debug.h) is the explicit form for a host that wants to do that wait at a time of its choosing.
  // Finished workers are reaped whenever a new one is started (reap_finished), so the list holds the compilations in flight, not a history.
  std::vector<std::pair<std::thread, std::shared_ptr<RtcKernel>>> workers;
  RtcCache() { (void)rtc_api(); }
  ~RtcCache() { join_workers(); }
  void join_workers() {
    std::vector<std::pair<std::thread, std::shared_ptr<RtcKernel>>> w;
    { std::lock_guard<std::mutex> lock(mu); w.swap(workers); }
    for (auto& t : w) if (t.first.joinable()) t.first.join();
  }
  void reap_finished() {   // caller holds mu; a worker whose kernel has left COMPILING is past its last use of the compiler: the join is short
    for (size_t i = 0; i < workers.size();) {
      if (workers[i].second->state.load() != RtcKernel::COMPILING) {
        if (workers[i].first.joinable()) workers[i].first.join();
        workers.erase(workers.begin() + (long)i);
      } else ++i;
    }
  }
  std::string csrc_dir, rocm_include, disk_dir;
  unsigned long long source_hash = 0;
  bool embedded = false;        // the sources compiled are the copy inside the library
  bool located = false;
  std::atomic<int> disk_hits{0}, disk_writes{0};
  // the sources of the kernel: the embedded copy, or <$TEB_AMD_CSRC | directory of libteb_amd.so + /csrc>; hip headers: $ROCM_PATH/include
  // or /opt/rocm/include; the disk cache: $TEB_AMD_RTC_CACHE | $XDG_CACHE_HOME/teb_amd | $HOME/.cache/teb_amd ("off" / "0" / no home: none)
  bool locate(std::string* why) {
    if (located) return true;
    const char* env = getenv("TEB_AMD_CSRC");
#ifdef TEB_AMD_RTC_EMBEDDED
    embedded = !(env && *env);
#endif
    if (embedded) {
#ifdef TEB_AMD_RTC_EMBEDDED
      source_hash = rtc_fnv(kRtcEmbeddedHash, sizeof kRtcEmbeddedHash);
#endif
    } else {
      if (env && *env) csrc_dir = env;
      else {
        Dl_info info;
        if (!dladdr(reinterpret_cast<const void*>(&rtc_api), &info) || !info.dli_fname) { *why = "cannot locate libteb_amd.so"; return false; }
        std::string p = info.dli_fname;
        const size_t k = p.find_last_of('/');
        csrc_dir = (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/csrc";
      }
      // (every file the translation unit includes goes into the key of the disk cache)
      const char* files[] = {"teb_kernel.hpp", "teb_edges.hpp", "teb_geometry.hpp", "teb_device.hpp", "teb_multicu.hpp", "teb_autoresize_chain.hpp",
                             "../../include/teb_amd.h"};
      unsigned long long h = 1469598103934665603ull;
      for (const char* name : files) {
        FILE* f = fopen((csrc_dir + "/" + name).c_str(), "rb");
        if (!f) { *why = "kernel sources not found at " + csrc_dir + " (" + name + "; set TEB_AMD_CSRC)"; return false; }
        char buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) h = rtc_fnv(buf, n, h);
        fclose(f);
      }
      source_hash = h;
    }
    const char* rp = getenv("ROCM_PATH");
    rocm_include = std::string(rp && *rp ? rp : "/opt/rocm") + "/include";
    const char* dc = getenv("TEB_AMD_RTC_CACHE");
    if (dc && *dc) { if (std::string(dc) != "off" && std::string(dc) != "0") disk_dir = dc; }
    else {
      const char* xdg = getenv("XDG_CACHE_HOME");
      const char* home = getenv("HOME");
      if (xdg && *xdg) disk_dir = std::string(xdg) + "/teb_amd";
      else if (home && *home) disk_dir = std::string(home) + "/.cache/teb_amd";
    }
    located = true;
    return true;
  }
};
inline RtcCache& rtc_cache() { static RtcCache c; return c; }
inline std::atomic<int>& rtc_cache_disk_hits() { return rtc_cache().disk_hits; }
inline std::atomic<int>& rtc_cache_disk_writes() { return rtc_cache().disk_writes; }

// names of the table's flags in TEB_PF_ALL order (-DTEB_PF_VALUE_<name>=0|1)
inline const std::vector<std::string>& rtc_flag_names() {
  static const std::vector<std::string> names = {
#define TEB_RTC_NAME(ID) #ID,
      TEB_PF_ALL(TEB_RTC_NAME)
#undef TEB_RTC_NAME
  };
  return names;
}

// what a compilation needs from the cache object (copied: the compiler thread outlives no lock)
struct RtcEnv {
  std::string csrc_dir, rocm_include, disk_dir;
  unsigned long long source_hash = 0;
  bool embedded = false;
};

// ---- the disk cache: <dir>/<key>.co = "TEBRTC01" | key | lowered-name length | code length | lowered name | code | FNV-1a of the code
inline std::string rtc_disk_path(const RtcEnv& env, unsigned long long key) {
  char name[64];
  snprintf(name, sizeof name, "/%016llx.co", key);
  return env.disk_dir + name;
}
// A code object read from disk runs in this process's GPU context: the directory has to be the user's own (owner = effective uid, no
// write permission for group / others) and the file is opened without following a symbolic link. The checksum inside the file detects
// corruption (a torn write, a bad disk); it is no protection against someone who can write the directory - the ownership test is.
inline bool rtc_disk_dir_trusted(const std::string& dir) {
  // (a cache directory reached through a symbolic link - ~/.cache on another volume - is judged by what the link resolves to)
  char real[PATH_MAX];
  if (!realpath(dir.c_str(), real)) return false;
  struct stat st;
  if (lstat(real, &st) != 0 || !S_ISDIR(st.st_mode)) return false;
  return st.st_uid == geteuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}
inline bool rtc_disk_load(const RtcEnv& env, unsigned long long key, RtcKernel& k) {
  if (env.disk_dir.empty() || !rtc_disk_dir_trusted(env.disk_dir)) return false;
  const int fd = open(rtc_disk_path(env, key).c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
  if (fd < 0) return false;
  struct stat fst;
  if (fstat(fd, &fst) != 0 || !S_ISREG(fst.st_mode) || fst.st_uid != geteuid()) { close(fd); return false; }
  FILE* f = fdopen(fd, "rb");
  if (!f) { close(fd); return false; }
  bool ok = false;
  char magic[8];
  unsigned long long fkey = 0, clen = 0, sum = 0;
  unsigned int llen = 0;
  if (fread(magic, 1, 8, f) == 8 && memcmp(magic, "TEBRTC01", 8) == 0 && fread(&fkey, 8, 1, f) == 1 && fkey == key && fread(&llen, 4, 1, f) == 1 &&
      fread(&clen, 8, 1, f) == 1 && llen > 0 && llen < 4096 && clen > 0 && clen < (1ull << 30)) {
    std::string lowered(llen, '\0');
    std::vector<char> code((size_t)clen);
    if (fread(&lowered[0], 1, llen, f) == llen && fread(code.data(), 1, (size_t)clen, f) == (size_t)clen && fread(&sum, 8, 1, f) == 1 &&
        sum == rtc_fnv(code.data(), code.size())) {
      k.lowered = lowered;
      k.code.swap(code);
      ok = true;
    }
  }
  fclose(f);
  return ok;
}
inline bool rtc_disk_store(const RtcEnv& env, unsigned long long key, const RtcKernel& k) {
  if (env.disk_dir.empty()) return false;
  // (parents one level up are expected to exist: ~/.cache or the directory the user named; a failure just means no cache)
  const size_t slash = env.disk_dir.find_last_of('/');
  if (slash != std::string::npos && slash > 0) (void)mkdir(env.disk_dir.substr(0, slash).c_str(), 0700);
  (void)mkdir(env.disk_dir.c_str(), 0700);
  if (!rtc_disk_dir_trusted(env.disk_dir)) return false;   // (somebody else's directory, or one others can write: no cache)
  const std::string path = rtc_disk_path(env, key);
  char tmp[64];
  snprintf(tmp, sizeof tmp, ".tmp.%ld", (long)getpid());
  const std::string tpath = path + tmp;
  const int fd = open(tpath.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
  if (fd < 0) return false;
  FILE* f = fdopen(fd, "wb");
  if (!f) { close(fd); (void)unlink(tpath.c_str()); return false; }
  const unsigned int llen = (unsigned int)k.lowered.size();
  const unsigned long long clen = k.code.size(), sum = rtc_fnv(k.code.data(), k.code.size());
  bool ok = fwrite("TEBRTC01", 1, 8, f) == 8 && fwrite(&key, 8, 1, f) == 1 && fwrite(&llen, 4, 1, f) == 1 && fwrite(&clen, 8, 1, f) == 1 &&
            fwrite(k.lowered.data(), 1, llen, f) == llen && fwrite(k.code.data(), 1, (size_t)clen, f) == (size_t)clen && fwrite(&sum, 8, 1, f) == 1;
  ok = (fclose(f) == 0) && ok;
  if (ok) ok = rename(tpath.c_str(), path.c_str()) == 0;   // atomic: a reader sees the old file, no file, or the whole new one
  if (!ok) (void)unlink(tpath.c_str());
  return ok;
}

inline void rtc_compile(const RtcKey key, std::shared_ptr<RtcKernel> k, const RtcEnv env) {
  RtcApi& api = rtc_api();
  const auto t0 = std::chrono::steady_clock::now();
  auto fail = [&](const std::string& why) { k->log = why; k->state.store(RtcKernel::FAILED); };
  const std::string src = "#include \"teb_kernel.hpp\"\n";
  char name[128];
  snprintf(name, sizeof name, "tebamd::teb_optimize_kernel<%d, %d, %d>", key.solver, key.jmode, key.scene);
  std::vector<std::string> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                                   "-DM_PI=3.14159265358979323846", "-DHUGE_VAL=__builtin_huge_val()",   // (hipRTC's built-in headers lack the two math.h macros)
                                   "-DTEB_AMD_DEFAULTS_PROFILE=1", "-DTEB_AMD_PROFILE_CUSTOM=1",
                                   // The out-of-line solve on the PLAIN calling convention (the callee saves its callee-saved block): the
                                   // convention without it relies on LLVM's interprocedural register allocation handing the caller the callee's
                                   // exact clobber set, and that combination miscompiles some instantiations (round 5: the full-batch light
                                   // kind in the band layout faults at its first LM iteration; -mllvm -enable-ipra=0 on that one unit cures it,
                                   // so does this flag; tools/fault_probe.py, HISTORY.md section 3). The pre-built kinds that keep the cheaper
                                   // call are the ones every test and bench of the repository runs; a kernel compiled here is one of 2^20
                                   // nobody has run before, so it takes the convention that has never failed (cost: ~ 3 % of its launch).
                                   "-DTEB_AMD_SOLVE_CSR=1"};
  if (key.solver == 2) opts.push_back("-DTEB_AMD_POSE_ITER=" + std::to_string(kPoseIterBandHbm));   // band in HBM: as the pre-built units (teb_opt_inst.hip)
  for (const std::string& d : rtc_variant_defines()) opts.push_back(d);   // (part of the disk key below: the options are hashed)
  const std::vector<std::string>& names = rtc_flag_names();
  for (size_t i = 0; i < names.size(); ++i) opts.push_back("-DTEB_PF_VALUE_" + names[i] + "=" + (((key.flags >> i) & 1ull) ? "true" : "false"));
  // the key of the disk cache: sources, options (flag values, target), instantiation, compiler version
  unsigned long long dkey = rtc_fnv(&env.source_hash, sizeof env.source_hash);
  for (const std::string& o : opts) dkey = rtc_fnv(o.data(), o.size() + 1, dkey);
  dkey = rtc_fnv(name, strlen(name) + 1, dkey);
  int vmaj = 0, vmin = 0;
  if (api.Version) (void)api.Version(&vmaj, &vmin);
  dkey = rtc_fnv(&vmaj, sizeof vmaj, dkey);
  dkey = rtc_fnv(&vmin, sizeof vmin, dkey);
  if (rtc_disk_load(env, dkey, *k)) {
    k->from_disk = true;
    ++rtc_cache_disk_hits();
    k->compile_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    k->state.store(RtcKernel::READY);
    return;
  }
  if (!env.embedded) opts.push_back("-I" + env.csrc_dir);
  opts.push_back("-I" + env.rocm_include);
  RtcApi::program_t prog = nullptr;
  int nh = 0;
  const char* const* hsrc = nullptr;
  const char* const* hname = nullptr;
#ifdef TEB_AMD_RTC_EMBEDDED
  if (env.embedded) { nh = kRtcEmbeddedCount; hsrc = kRtcEmbeddedSources; hname = kRtcEmbeddedNames; }
#endif
  if (api.CreateProgram(&prog, src.c_str(), "teb_amd_rtc.hip", nh, const_cast<const char**>(hsrc), const_cast<const char**>(hname)) != 0)
    return fail("hiprtcCreateProgram failed");
  api.AddNameExpression(prog, name);
  std::vector<const char*> copts;
  for (const std::string& o : opts) copts.push_back(o.c_str());
  const int rc = api.CompileProgram(prog, (int)copts.size(), copts.data());
  size_t ls = 0;
  api.GetProgramLogSize(prog, &ls);
  std::string log(ls, '\0');
  if (ls) api.GetProgramLog(prog, &log[0]);
  if (rc != 0) { api.DestroyProgram(&prog); return fail("hiprtcCompileProgram failed: " + log.substr(0, 2000)); }
  const char* lowered = nullptr;
  if (api.GetLoweredName(prog, name, &lowered) != 0 || !lowered) { api.DestroyProgram(&prog); return fail("hiprtcGetLoweredName failed"); }
  k->lowered = lowered;
  size_t cs = 0;
  api.GetCodeSize(prog, &cs);
  k->code.resize(cs);
  if (cs == 0 || api.GetCode(prog, k->code.data()) != 0) { api.DestroyProgram(&prog); return fail("hiprtcGetCode failed"); }
  api.DestroyProgram(&prog);
  if (rtc_disk_store(env, dkey, *k)) ++rtc_cache_disk_writes();
  k->compile_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  k->log = log;
  k->state.store(RtcKernel::READY);
}

// The instantiation for `key`: starts its compilation on first request; wait = true blocks until it is ready (or failed).
inline std::shared_ptr<RtcKernel> rtc_request(const RtcKey& key, bool wait, std::string* why) {
  RtcCache& c = rtc_cache();
  std::shared_ptr<RtcKernel> k;
  {
    std::lock_guard<std::mutex> lock(c.mu);
    if (!c.locate(why)) return nullptr;
    if (!rtc_api().load()) { *why = rtc_api().error; return nullptr; }
    auto it = c.kernels.find(key);
    if (it != c.kernels.end()) k = it->second;
    else {
      k = std::make_shared<RtcKernel>();
      c.kernels[key] = k;
      RtcEnv env;
      env.csrc_dir = c.csrc_dir; env.rocm_include = c.rocm_include; env.disk_dir = c.disk_dir; env.source_hash = c.source_hash; env.embedded = c.embedded;
      c.reap_finished();
      c.workers.emplace_back(std::thread(rtc_compile, key, k, env), k);   // joined by reap_finished / ~RtcCache / teb_amd_debug_rtc_join
    }
  }
  if (wait)
    while (k->state.load() == RtcKernel::COMPILING) std::this_thread::sleep_for(std::chrono::milliseconds(5));
  return k;
}

// the function of a READY instantiation on `device` (module loaded on first use); nullptr on failure
inline hipFunction_t rtc_function(RtcKernel& k, int device, size_t lds_limit, std::string* why) {
  std::lock_guard<std::mutex> lock(k.mu);
  auto it = k.fn.find(device);
  if (it != k.fn.end()) return it->second;
  hipModule_t mod = nullptr;
  hipFunction_t f = nullptr;
  if (hipModuleLoadData(&mod, k.code.data()) != hipSuccess) { (void)hipGetLastError(); *why = "hipModuleLoadData failed"; k.fn[device] = nullptr; return nullptr; }
  if (hipModuleGetFunction(&f, mod, k.lowered.c_str()) != hipSuccess) { (void)hipGetLastError(); *why = "hipModuleGetFunction failed"; k.fn[device] = nullptr; return nullptr; }
  // dynamic LDS beyond 64 KB: the attribute of the function (as for the built-in instantiations, teb_amd_create)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit) != hipSuccess) (void)hipGetLastError();
  k.fn[device] = f;
  return f;
}

}  // namespace tebamd
