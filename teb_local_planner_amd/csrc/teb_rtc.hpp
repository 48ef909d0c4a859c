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
// libhiprtc is loaded with dlopen on first use (users who leave the option off neither link nor load it); the kernel sources are read
// from the csrc directory next to libteb_amd.so (the in-tree layout), hip/hip_runtime.h from the ROCm installation.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
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

struct RtcCache {
  std::mutex mu;
  std::map<RtcKey, std::shared_ptr<RtcKernel>> kernels;
  std::string csrc_dir, rocm_include;
  bool located = false;
  // the sources of the kernel: <directory of libteb_amd.so>/csrc (or $TEB_AMD_CSRC); hip headers: $ROCM_PATH/include or /opt/rocm/include
  bool locate(std::string* why) {
    if (located) return true;
    const char* env = getenv("TEB_AMD_CSRC");
    if (env && *env) csrc_dir = env;
    else {
      Dl_info info;
      if (!dladdr(reinterpret_cast<const void*>(&rtc_api), &info) || !info.dli_fname) { *why = "cannot locate libteb_amd.so"; return false; }
      std::string p = info.dli_fname;
      const size_t k = p.find_last_of('/');
      csrc_dir = (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/csrc";
    }
    FILE* f = fopen((csrc_dir + "/teb_kernel.hpp").c_str(), "r");
    if (!f) { *why = "kernel sources not found at " + csrc_dir + " (set TEB_AMD_CSRC)"; return false; }
    fclose(f);
    const char* rp = getenv("ROCM_PATH");
    rocm_include = std::string(rp && *rp ? rp : "/opt/rocm") + "/include";
    located = true;
    return true;
  }
};
inline RtcCache& rtc_cache() { static RtcCache c; return c; }

// names of the table's flags in TEB_PF_ALL order (-DTEB_PF_VALUE_<name>=0|1)
inline const std::vector<std::string>& rtc_flag_names() {
  static const std::vector<std::string> names = {
#define TEB_RTC_NAME(ID) #ID,
      TEB_PF_ALL(TEB_RTC_NAME)
#undef TEB_RTC_NAME
  };
  return names;
}

inline void rtc_compile(const RtcKey key, std::shared_ptr<RtcKernel> k, const std::string csrc_dir, const std::string rocm_include) {
  RtcApi& api = rtc_api();
  const auto t0 = std::chrono::steady_clock::now();
  auto fail = [&](const std::string& why) { k->log = why; k->state.store(RtcKernel::FAILED); };
  const std::string src = "#include \"teb_kernel.hpp\"\n";
  RtcApi::program_t prog = nullptr;
  if (api.CreateProgram(&prog, src.c_str(), "teb_amd_rtc.hip", 0, nullptr, nullptr) != 0) return fail("hiprtcCreateProgram failed");
  char name[128];
  snprintf(name, sizeof name, "tebamd::teb_optimize_kernel<%d, %d, %d>", key.solver, key.jmode, key.scene);
  api.AddNameExpression(prog, name);
  std::vector<std::string> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + csrc_dir, "-I" + rocm_include,
                                   "-DM_PI=3.14159265358979323846", "-DHUGE_VAL=__builtin_huge_val()",   // (hipRTC's built-in headers lack the two math.h macros)
                                   "-DTEB_AMD_DEFAULTS_PROFILE=1", "-DTEB_AMD_PROFILE_CUSTOM=1"};
  const std::vector<std::string>& names = rtc_flag_names();
  for (size_t i = 0; i < names.size(); ++i) opts.push_back("-DTEB_PF_VALUE_" + names[i] + "=" + (((key.flags >> i) & 1ull) ? "true" : "false"));
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
      std::thread(rtc_compile, key, k, c.csrc_dir, c.rocm_include).detach();
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
