// teb_comm.hpp — the path's only exchange step behind the C-ABI: best-trajectory selection across the ranks
// that share one candidate batch, and the broadcast of the winner's strip. One process per GPU; RCCL over xGMI.
//
// Reference: the reduction is HomotopyClassPlanner::selectBestTeb (src/homotopy_class_planner.cpp:564-667) - an arg-min with strict '<'
// (lowest index wins ties, :610) over costs that already carry the hysteresis / prefer-initial-plan multipliers. Each rank reduces its
// own candidates on the device (select_best_kernel), contributes ONE 16-byte record (cost f64, global index as f64) to an
// ncclAllGather, and every rank takes the lexicographic minimum of the world records - bit-exact costs, no second collective.
//
// librccl is loaded with dlopen on first use and its handful of entry points are declared HERE (names private to this library, values
// those of rccl.h 2.x: ncclSuccess = 0, ncclFloat64 = 8, a 128-byte unique id): single-GPU users of libteb_amd.so neither link nor load
// librccl, and building the library does not need the RCCL headers.
#pragma once
#include <dlfcn.h>

#include <mutex>
#include <string>

namespace tebamd {

typedef struct RcclCommOpaque* rccl_comm_t;
struct rccl_unique_id_t { char internal[128]; };
enum { kRcclSuccess = 0, kRcclFloat64 = 8 };

struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(rccl_unique_id_t*) = nullptr;
  int (*CommInitRank)(rccl_comm_t*, int, rccl_unique_id_t, int) = nullptr;
  int (*CommDestroy)(rccl_comm_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, rccl_comm_t, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string error;
  std::mutex mu;      // teb_amd_comm_* may be called from several host threads (one handle per thread)
  bool ready = false;
  bool load() {
    std::lock_guard<std::mutex> lock(mu);
    if (ready) return true;
    // a copy already in the process (e.g. the one PyTorch ships) is reused: two RCCL instances in one process would each claim the GPUs
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* l = nullptr;
    for (const char* n : names) if ((l = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD))) break;
    if (!l) for (const char* n : names) if ((l = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!l) { const char* e = dlerror(); error = std::string("librccl not found: ") + (e ? e : "?"); return false; }
    bool ok = true;
    auto sym = [&](const char* s) { void* p = dlsym(l, s); if (!p) { error = std::string("librccl lacks ") + s; ok = false; } return p; };
    GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
    CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
    Broadcast = reinterpret_cast<decltype(Broadcast)>(sym("ncclBroadcast"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) { dlclose(l); return false; }
    lib = l;
    ready = true;
    return true;
  }
};
inline RcclApi& rccl() { static RcclApi api; return api; }

// Lexicographic minimum over the gathered records (cost, global index as f64; index < 0 = the rank has no candidate): the strict '<' of
// selectBestTeb (:610) keeps the first minimum, i.e. on equal costs the lowest global index wins, whatever rank holds it. Plain host
// code, the same on every rank (so all ranks agree without a second collective); unit-tested through teb_amd_debug_world_argmin.
inline void world_argmin(const double* records, int world, int* best_index, double* best_cost, int* owner_rank) {
  double bc = 1.7976931348623157e308; int bi = -1, owner = -1;
  for (int r = 0; r < world; ++r) {
    const double cst = records[2 * r]; const int idx = (int)records[2 * r + 1];
    if (idx < 0) continue;
    if (bi < 0 || cst < bc || (cst == bc && idx < bi)) { bc = cst; bi = idx; owner = r; }
  }
  *best_index = bi; *best_cost = bc; *owner_rank = owner;
}

// record of this rank for the all-gather: (scaled cost, global index) from the result of select_best_kernel; an empty rank sends (max, -1)
__global__ void pack_record_kernel(const double* sel_cost, const int* sel_idx, int offset, int have, double* rec) {
  if (threadIdx.x == 0) {
    const int i = have ? *sel_idx : -1;
    rec[0] = (have && i >= 0) ? *sel_cost : 1.7976931348623157e308;
    rec[1] = i >= 0 ? (double)(offset + i) : -1.0;
  }
}
// winner's strip -> one contiguous message [n | x | y | theta | dt | statistics], each strip `cap` doubles; statistics = (available,
// back_chi2) of teb_amd_get_batch_statistics: what hasDiverged reads travels with the band, so a mirror of the winner answers like its owner
__global__ void pack_band_kernel(const int* n, const double* x, const double* y, const double* th, const double* dt, int b, int stride, int cap,
                                 double* msg, const double* chi2, const int* iters, const int* last_iters, int stats_on, int last_inner) {
  const int nb = n[b];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    msg[0] = (double)nb;
    const bool avail = stats_on && iters[b] > 0 && last_inner > 0;
    msg[1 + 4 * (size_t)cap] = avail ? 1.0 : 0.0;
    msg[2 + 4 * (size_t)cap] = (avail && last_iters[b] == last_inner) ? chi2[b] : 0.0;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += gridDim.x * blockDim.x) {
    const bool in = i < nb && i < stride;
    const size_t o = (size_t)b * stride + i;
    msg[1 + i] = in ? x[o] : 0.0; msg[1 + cap + i] = in ? y[o] : 0.0; msg[1 + 2 * cap + i] = in ? th[o] : 0.0; msg[1 + 3 * cap + i] = in ? dt[o] : 0.0;
  }
}

}  // namespace tebamd

struct teb_amd_comm {
  tebamd::rccl_comm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  double* rec = nullptr;       // [2] this rank's record
  double* all = nullptr;       // [2 * world]
  double* msg = nullptr;       // broadcast message, msg_cap doubles
  size_t msg_cap = 0;
  int band_stats_available = 0;   // statistics that came with the last broadcast band (teb_amd_comm_last_band_statistics)
  double band_stats_back_chi2 = 0;
};
