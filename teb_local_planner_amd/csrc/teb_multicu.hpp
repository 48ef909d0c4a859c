// teb_multicu.hpp — small batches of generic-shape scenes on more than one CU (SURVEY.md section 7, K2 "association as its own tiled
// kernel"). A candidate TEB is one workgroup on one CU: with B <= 16 candidates 240 of the 256 CUs idle while
// the band's lanes walk segment / polygon distance loops one obstacle after the other (BASELINE C5: 84 % of the cycles are work
// that is independent per (pose, obstacle): src/optimal_planner.cpp:483-547, distance_calculations.h:236-262).
//
// In this mode a launch has B x (1 + H) workgroups: workgroup b < B is the MASTER of band b and runs the unchanged optimizeTEB loop;
// workgroups B + b * H + j, j < H, are the HELPERS of band b (persistent for the launch). The master hands out two kinds of phases:
//   ASSOC : AddEdgesObstacles' association scan, pose tiles of 8 lanes per pose  -> association lists (HBM)
//   DIST  : robot <-> obstacle distance + gradient of every (pose, list entry) and (pose, dynamic obstacle) pair, one pair per lane
//           -> item records (dist, d/dx, d/dy, d/dtheta) in HBM
// and then replays the residual rows of those edges from the delivered records in list order - the same numbers the single-CU path
// computes (same device functions on the same inputs), accumulated in the same order: the bands are bit-identical (tested).
// The records of the error evaluation of an accepted LM trial are the records the next linearisation needs (same state), so a
// linearisation asks for a DIST phase only right after buildGraph.
//
// Inter-workgroup visibility follows the guide's rule for gfx950 (8 XCDs with private L2s, per-CU L1s that other CUs never
// refresh): EVERY shared word - control words, published poses, association lists, item records - is written AND read with
// agent-scope relaxed atomics on global-address-space pointers (sc1 write-through stores / sc1 loads, "8-byte agent atomics on both
// sides"); every storing wave drains its stores (s_waitcnt vmcnt(0)) and the workgroup meets at a barrier before ONE lane stores the
// flag / bumps the arrival counter. No fences, nothing placement-dependent. Every spin is bounded by the 100 MHz real-time counter:
// a master that does not get its arrivals flags the band (assoc_overflow bit 2) and the host repeats the launch on one CU per band;
// a helper that hears nothing for the timeout leaves.
#pragma once
#include "teb_device.hpp"

namespace tebamd {

typedef __attribute__((address_space(1))) unsigned int gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;

__device__ __forceinline__ unsigned ld_agent_u32(const unsigned* p) {
  return __hip_atomic_load(reinterpret_cast<const gu32*>(reinterpret_cast<unsigned long long>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent_u32(unsigned* p, unsigned v) {
  __hip_atomic_store(reinterpret_cast<gu32*>(reinterpret_cast<unsigned long long>(p)), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_agent_i32(const int* p) { return (int)ld_agent_u32(reinterpret_cast<const unsigned*>(p)); }
__device__ __forceinline__ void st_agent_i32(int* p, int v) { st_agent_u32(reinterpret_cast<unsigned*>(p), (unsigned)v); }
__device__ __forceinline__ double ld_agent_f64(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const gu64*>(reinterpret_cast<unsigned long long>(p)), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent_f64(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<gu64*>(reinterpret_cast<unsigned long long>(p)), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned add_agent_u32(unsigned* p, unsigned v) {
  return __hip_atomic_fetch_add(reinterpret_cast<gu32*>(reinterpret_cast<unsigned long long>(p)), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void or_agent_i32(int* p, int v) {
  __hip_atomic_fetch_or(reinterpret_cast<gu32*>(reinterpret_cast<unsigned long long>(p)), (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// association lists are shared between workgroups only in the multi-CU mode: plain accesses otherwise
template <bool SHARED> __device__ __forceinline__ void st_list(int* p, int v) { if constexpr (SHARED) st_agent_i32(p, v); else *p = v; }
__device__ __forceinline__ int ld_list(bool shared, const int* p) { return shared ? ld_agent_i32(p) : *p; }

__device__ __forceinline__ long long realtime_ticks() { return (long long)__builtin_amdgcn_s_memrealtime(); }   // 100 MHz, independent of the shader clock

// control block of one band: 64 words, the contended ones on cache lines of their own
constexpr int kMcuCtlWords = 64;
enum { MCU_CMD = 0, MCU_DONE = 16, MCU_N = 32, MCU_ABORT = 33, MCU_SABORT = 34, MCU_SCMD = 48, MCU_SN = 49, MCU_SDONE = 52 /* + k, k = 1 .. kMcuMaxSpec */ };
enum { MCU_KIND_ASSOC = 1, MCU_KIND_DIST = 2, MCU_KIND_EXIT = 3 };
constexpr unsigned kMcuSpecExit = 0xffffffffu;
constexpr int kMcuPubArrays = 5;     // x, y, cos, sin, time stamp of the dynamic edges
constexpr int kMcuMaxSpec = 3;       // speculative damped solves per LM iteration: lambda * 2, * 8, * 64 (the first three rejections)
__host__ __device__ inline size_t mcu_spec_slot(int S) { return (size_t)4 * S + 16; }   // doubles per slot: 4 S + 8 values, then 8 meta

// A launch of the small-batch instantiations has B x (1 + K + D) workgroups: per band its own workgroup, K solver helpers (speculative LM
// trials, below) and D distance helpers (generic scenes: association + distance records, above).
struct McuDev {            // kernel argument; K == D == 0: single-CU launch, the pointers are not touched
  int K;                   // solver helpers per band
  int D;                   // distance helpers per band (generic scenes only)
  unsigned* ctl;           // [B][kMcuCtlWords], zeroed by the host before every launch
  double* pub;             // [B][kMcuPubArrays][S]
  double* items;           // [B][item_cap][4][S]: record q of item k of pose i at ((k * 4 + q) * S + i)
  int item_cap;            // >= association entries + dynamic obstacles of a pose
  double* spec;            // [B][1 + K][mcu_spec_slot(S)]: slot 0 = right-hand side b (+ the lambdas at 4 S + 8 + k), slot k = step of solver helper k (+ "factorisation ok" at 4 S + 8)
  long long timeout_ticks; // of the real-time counter (100 MHz)
  unsigned* trace;         // diagnostic (teb_amd_debug_mcu_watchdog): host-visible breadcrumbs, one word per workgroup, or nullptr
  int debug_flags;         // diagnostic (teb_amd_debug_mcu_flags): 1 = the master associates itself, 2 = it computes the distances itself, 4 = it solves every trial itself
};
// breadcrumb of this workgroup: (epoch << 8) | code, system scope so that the host can read it while the kernel runs
__device__ __forceinline__ void mcu_trace(unsigned* trace, unsigned epoch, unsigned code) {
  if (trace && threadIdx.x == 0) __hip_atomic_store(trace + blockIdx.x, (epoch << 8) | code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// what the master's edge evaluation needs to replay delivered records (lives in TebCtx)
struct McuView {
  const double* items;     // of this band, or nullptr: compute the distances locally
  bool shared_lists;       // association lists were written by other workgroups: read them with agent-scope loads
};

// ---- master side ----------------------------------------------------------------------------------------------------------------
struct McuMaster {
  int H;                   // distance helpers of the band
  int K;                   // solver helpers of the band
  unsigned sepoch;         // speculative solves issued so far (one per LM iteration)
  bool spec_failed;        // a solver helper did not deliver in time: this workgroup solves every further trial itself
  double* spec;
  unsigned* ctl;
  double* pub;
  unsigned epoch;          // phases issued so far
  bool failed;             // an arrival timed out: the band is flagged, no further phases are issued
  long long timeout;
  unsigned* trace;
};

// publish the poses (and time stamps) of the phase: every lane its poses, write-through; then the storing waves drain and meet
__device__ __forceinline__ void mcu_publish(const McuMaster& m, const double* sx, const double* sy, const double* cs, const double* sn, const double* tdyn, int n,
                                            int S) {
  for (int i = threadIdx.x; i < n; i += kThreads) {
    st_agent_f64(m.pub + i, sx[i]); st_agent_f64(m.pub + S + i, sy[i]); st_agent_f64(m.pub + 2 * S + i, cs[i]); st_agent_f64(m.pub + 3 * S + i, sn[i]);
    st_agent_f64(m.pub + 4 * S + i, tdyn[i]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  mcu_trace(m.trace, m.epoch, 1);
}
// one lane: the command word of the next phase (the poses are already out)
__device__ __forceinline__ void mcu_issue(McuMaster& m, int kind, int n) {
  ++m.epoch;
  if (threadIdx.x == 0 && !m.failed) {
    st_agent_u32(m.ctl + MCU_N, (unsigned)n);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st_agent_u32(m.ctl + MCU_CMD, (m.epoch << 8) | (unsigned)kind);
  }
  mcu_trace(m.trace, m.epoch, 2);
}
// all lanes: until the H helpers have delivered the phase (or the timeout); returns false once the band has failed
__device__ __forceinline__ bool mcu_wait(McuMaster& m, int* lds_flag) {
  if (threadIdx.x == 0) {
    int ok = m.failed ? 0 : 1;
    if (ok) {
      const unsigned want = m.epoch * (unsigned)m.H;
      const long long t0 = realtime_ticks();
      while (ld_agent_u32(m.ctl + MCU_DONE) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (realtime_ticks() - t0 > m.timeout) { ok = 0; st_agent_u32(m.ctl + MCU_ABORT, 1u); break; }
      }
    }
    *lds_flag = ok;
  }
  __syncthreads();
  const bool ok = *lds_flag != 0;
  __syncthreads();
  if (!ok) m.failed = true;
  mcu_trace(m.trace, m.epoch, ok ? 3 : 4);
  return ok;
}
__device__ __forceinline__ void mcu_exit(McuMaster& m) {
  if (threadIdx.x == 0) {
    ++m.epoch;
    if (m.H > 0) st_agent_u32(m.ctl + MCU_CMD, (m.epoch << 8) | (unsigned)MCU_KIND_EXIT);
    if (m.K > 0) st_agent_u32(m.ctl + MCU_SCMD, kMcuSpecExit);
  }
  mcu_trace(m.trace, m.epoch, 9);
}

// ---- speculative LM trials ------------------------------------------------------------------------------------------------------
// OptimizationAlgorithmLevenberg::solve retries a rejected step with lambda * 2, * 4, * 8 ..: the damped systems
// of the first K retries are known as soon as the iteration is linearised. The band's workgroup publishes the right-hand side (the
// normal matrix is in its HBM backup / band copy anyway, written through in this mode) and the K lambdas; solver helper k solves
// (H + lambda_k I) dx = b with the SAME routine on a spare CU while this workgroup solves and evaluates trial 0. A rejected trial then
// finds its step ready instead of spending another solve - the accepted step, and every bit of it, is what the sequential loop computes.
__device__ __forceinline__ void spec_wait_idle(McuMaster& m, int* lds_flag) {   // the helpers have left the buffers of the previous iteration
  if (m.sepoch == 0 || m.spec_failed) return;
  if (threadIdx.x == 0) {
    int ok = 1;
    const long long t0 = realtime_ticks();
    for (int k = 1; k <= m.K && ok; ++k)
      while (ld_agent_u32(m.ctl + MCU_SDONE + k) < m.sepoch) {
        __builtin_amdgcn_s_sleep(1);
        if (realtime_ticks() - t0 > m.timeout) { ok = 0; st_agent_u32(m.ctl + MCU_SABORT, 1u); break; }
      }
    *lds_flag = ok;
  }
  __syncthreads();
  if (*lds_flag == 0) m.spec_failed = true;
  __syncthreads();
}
// right-hand side b (Nt values) and the lambdas of the retries, then the command; lambda0 / ni0: the damping of trial 0 and its multiplier
__device__ __forceinline__ void spec_issue(McuMaster& m, const double* bv, int Nt, int n, int S, double lambda0, double ni0) {
  for (int r = threadIdx.x; r < Nt; r += kThreads) st_agent_f64(m.spec + r, bv[r]);
  if (threadIdx.x == 0) {
    double lam = lambda0, ni = ni0;
    for (int k = 1; k <= m.K; ++k) { lam *= ni; ni *= 2; st_agent_f64(m.spec + 4 * S + 8 + k, lam); }   // exactly the products of the retry loop
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ++m.sepoch;
  if (threadIdx.x == 0) {
    st_agent_u32(m.ctl + MCU_SN, (unsigned)n);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st_agent_u32(m.ctl + MCU_SCMD, m.sepoch);
  }
}
// the step of retry k into dxv; returns false when the helper did not deliver in time (the caller solves itself from then on). The
// patience here is short - a few solve times, kMcuSpecPatienceTicks: a helper that has not finished by the time its step is wanted is
// a helper without a CU (a busy device, or a batch that fills the chip exactly), and waiting for it would cost more than solving.
constexpr long long kMcuSpecPatienceTicks = 20000;   // 200 us of the 100 MHz counter
__device__ __forceinline__ bool spec_take(McuMaster& m, int k, double* dxv, int* ok_flag, int Nt, int S, int* lds_flag) {
  if (threadIdx.x == 0) {
    int ok = 1;
    const long long t0 = realtime_ticks();
    const long long patience = m.timeout < kMcuSpecPatienceTicks ? m.timeout : kMcuSpecPatienceTicks;
    while (ld_agent_u32(m.ctl + MCU_SDONE + k) < m.sepoch) {
      __builtin_amdgcn_s_sleep(1);
      if (realtime_ticks() - t0 > patience) { ok = 0; st_agent_u32(m.ctl + MCU_SABORT, 1u); break; }
    }
    *lds_flag = ok;
  }
  __syncthreads();
  const bool ok = *lds_flag != 0;
  __syncthreads();
  if (!ok) { m.spec_failed = true; return false; }
  const double* out = m.spec + (size_t)k * mcu_spec_slot(S);
  for (int r = threadIdx.x; r < Nt; r += kThreads) dxv[r] = ld_agent_f64(out + r);
  if (threadIdx.x == 0) *ok_flag = ld_agent_f64(out + 4 * S + 8) != 0.0 ? 1 : 0;
  __syncthreads();
  return true;
}

}  // namespace tebamd
