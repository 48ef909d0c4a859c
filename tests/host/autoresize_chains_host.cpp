// Host emulation of the chain-parallel autoResize sweep (teb_local_planner_amd/csrc/teb_autoresize_chain.hpp): the lanes of the device run
// one after the other here, the reach / scan phases are plain loops. Compiled by tests/test_autoresize_chains.py with g++; the chain
// machine itself is the device's source, unchanged. TEST INFRASTRUCTURE ONLY.
#include <cstring>
#include <vector>
#define TEB_AR_HD inline
#include "../../teb_local_planner_amd/csrc/teb_autoresize_chain.hpp"

using namespace tebamd;

// One sweep over dt[0 .. Tin). Outputs: odt[K], out_desc[K + 1], rec[NN]; info[0..9] = K, NN, modified, deepest tree, tail_k, declined
// (1: a member gave up, 2: a guard may bind), members, longest member chain (rule evaluations), longest chain of any lane, sum of member steps.
// Returns 0, or 1 if the output capacity `cap` would be exceeded.
extern "C" int teb_host_autoresize_chains(const double* dt, int Tin, double dt_ref, double hyst, int min_samples, int max_samples, int cap,
                                          double* odt, int* out_desc, int* rec, double* tail_dt, int* info) {
  std::vector<ArChainResult> res(Tin);
  ArChainOut o;
  o.odt = nullptr; o.out_desc = nullptr; o.rec = nullptr; o.tail_dt = nullptr; o.k0 = 0; o.nn0 = 0;
  for (int i = 0; i < Tin; ++i) res[i] = ar_chain_run<false>(dt, Tin, i, dt_ref, hyst, o);
  std::vector<char> reach(Tin, 0);
  for (int i = 0; i < Tin; i = res[i].next) reach[i] = 1;
  int K = 0, NN = 0, S = 0, M = 0, md = 0, gave = 0, members = 0, longest_member = 0, longest_any = 0, member_steps = 0;
  std::vector<int> k0(Tin, 0), nn0(Tin, 0);
  for (int i = 0; i < Tin; ++i) {
    if (!reach[i]) continue;
    k0[i] = K; nn0[i] = NN; ++members;
    K += res[i].emitted; NN += res[i].new_poses; S += res[i].splits; M += res[i].merges;
    md = res[i].depth > md ? res[i].depth : md;
    gave |= res[i].gave_up;
    longest_member = res[i].steps > longest_member ? res[i].steps : longest_member;
    member_steps += res[i].steps;
  }
  for (int i = 0; i < Tin; ++i) longest_any = res[i].steps > longest_any ? res[i].steps : longest_any;
  std::memset(info, 0, 10 * sizeof(int));
  info[4] = -1; info[6] = members; info[7] = longest_member; info[8] = longest_any; info[9] = member_steps;
  if (gave) { info[5] = 1; return 0; }
  if (Tin + S >= max_samples || Tin - M <= min_samples) { info[5] = 2; return 0; }
  if (K > cap - 1 || NN > cap) return 1;
  o.odt = odt; o.out_desc = out_desc; o.rec = rec; o.tail_dt = tail_dt;
  for (int i = 0; i < Tin; ++i) {
    if (!reach[i]) continue;
    o.k0 = k0[i]; o.nn0 = nn0[i];
    const ArChainResult r = ar_chain_run<true>(dt, Tin, i, dt_ref, hyst, o);
    if (r.tail) info[4] = o.k0 + r.emitted - 1;
  }
  out_desc[K] = Tin;   // the goal pose (n_in - 1)
  info[0] = K; info[1] = NN; info[2] = (S + M) > 0; info[3] = md;
  return 0;
}
