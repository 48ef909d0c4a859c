// One sweep of TimedElasticBand::autoResize (/root/reference/src/timed_elastic_band.cpp:227-286) cut into CHAINS that run side by side.
//
// The reference walks the time differences once, left to right, and every rule it applies hands something on to the right: a split
// leaves the right half waiting, "too long" pushes the excess onto the next interval, "too short" merges into the next interval. The
// sequential rule machine (autoresize_script_lane0, teb_kernel.hpp) therefore carries a state - the interval under test, the stack of
// waiting halves, the (possibly touched) next input interval. But whenever that state is CLEAN - the stack is empty and the interval
// under test is an input interval nothing was added to - everything the machine does from there on is a function of the input intervals
// to the right alone (given that the two sample-count guards of the rules do not bind, see below). So:
//
//   pass A  every lane i starts a machine in the clean state at input interval i and runs it to the next clean state: the CHAIN of i.
//           It records where the chain ends (next[i]) and what it produces (emitted intervals, new poses, splits, merges).
//   reach   the sweep of the reference is the sequence of chains 0 -> next[0] -> next[next[0]] -> .. ; pointer doubling marks its members.
//   scan    prefix sums over the members give every chain its place in the output (first emitted interval, first new pose).
//   pass B  the members run their chain again and write the edit script (the same records the sequential machine writes).
//
// Lanes that are not members computed a chain nobody needs; the price is their lock-step company, the gain is that a sweep costs the
// LONGEST chain (10 - 25 rule evaluations on the BASELINE bands) instead of the sum of all chains (200 - 330).
// The guards: a rule fires only while sizeTimeDiffs() < max_samples (split / push) resp. > min_samples (merge). The chains assume both
// hold; the sweep checks afterwards that they did at every point of the sequence - T_in + all splits < max_samples and T_in - all merges
// > min_samples bound the count from both sides - and otherwise (or when a member gave up: chain longer than kArChainSteps evaluations,
// more than kArChainStack halves waiting) the sequential machine runs the sweep. Same rules, same order, same edit script: the bands
// are bit-identical (tests/test_autoresize_chains.py on the host, the autoResize parity tests and the bit fingerprints on the device).
//
// This header has no device-only construct: the host test compiles it with TEB_AR_HD empty and emulates the lanes one after the other.
#pragma once
#ifndef TEB_AR_HD
#define TEB_AR_HD __host__ __device__ __forceinline__
#endif

namespace tebamd {

constexpr int kArChainSteps = 64;   // rule evaluations a chain may take before it gives up
constexpr int kArChainStack = 4;    // split halves waiting in a chain (deeper: the sequential machine takes the sweep)
constexpr int kArNewPose = 1024;    // = kNewPose of teb_kernel.hpp: descriptors >= this denote new poses

struct ArChainResult {
  int next;       // input interval at which the machine is clean again (T_in: the chain ran to the end of the band)
  int emitted;    // intervals it emits
  int new_poses;  // poses it inserts
  int splits, merges;
  int depth;      // deepest split tree
  int gave_up;    // 1: step or stack limit hit, the counts are meaningless
  int tail;       // 1: it ended with the rule of the last interval (merged into the interval emitted before it)
  int steps;      // rule evaluations
};

// where a member chain writes (pass B); null pointers in pass A
struct ArChainOut {
  double* odt;      // [k0 + ..] emitted time differences
  int* out_desc;    // [k0 + ..] descriptor of the pose that starts the emitted interval
  int* rec;         // [nn0 + ..] new-pose records: parent A | parent B << 11 | depth << 22
  double* tail_dt;  // amount merged into emitted interval k0 + emitted - 1 by the last-interval rule
  int k0, nn0;
};

// The machine of autoresize_script_lane0 started clean at input interval i (no run records: an unmarked interval is a chain of one
// evaluation). in_dt[0 .. Tin): the time differences; n_in = Tin + 1 poses. first_chain: nothing was emitted before this chain (i == 0).
template <bool EMIT>
TEB_AR_HD ArChainResult ar_chain_run(const double* in_dt, int Tin, int i, double dt_ref, double hyst, const ArChainOut& o) {
  ArChainResult r;
  r.next = Tin; r.emitted = 0; r.new_poses = 0; r.splits = 0; r.merges = 0; r.depth = 0; r.gave_up = 0; r.tail = 0; r.steps = 0;
  const int n_in = Tin + 1;
  int j = i + 1;                       // next unread input interval
  int cdesc = i, cdepth = 0;
  double cdt = in_dt[i];
  double pdt = (j < Tin) ? in_dt[j] : 0.0;
  bool ptouched = false;
  // waiting halves, top first (s0); shifted on push / pop: no dynamic indexing, everything stays in registers
  int sp = 0;
  double s0t = 0, s1t = 0, s2t = 0, s3t = 0;
  int s0d = 0, s1d = 0, s2d = 0, s3d = 0;   // descriptor | depth << 16
  int k = 0, nn = 0;
  const double hi_lim = dt_ref + hyst, lo_lim = dt_ref - hyst, big_lim = 2 * dt_ref;
  for (int step = 0; ; ++step) {
    if (step >= kArChainSteps) { r.gave_up = 1; break; }
    r.steps = step + 1;
    const bool has_next = (sp > 0) || (j < Tin);
    bool emit = true;
    if (cdt > hi_lim) {                                   // (.. && sizeTimeDiffs() < max_samples: checked by the sweep)
      if (cdt > big_lim) {
        // split: the left half is re-checked, the right half waits; Pose(i+1) = the waiting half on top, else the next input pose, else the goal
        const double newtime = 0.5 * cdt;
        int edesc, edepth;
        if (sp > 0) { edesc = s0d & 0xffff; edepth = s0d >> 16; }
        else { edesc = (j < Tin) ? j : n_in - 1; edepth = 0; }
        if (sp >= kArChainStack) { r.gave_up = 1; break; }
        const int depth = 1 + (cdepth > edepth ? cdepth : edepth);
        if (EMIT) o.rec[o.nn0 + nn] = cdesc | (edesc << 11) | (depth << 22);
        r.depth = depth > r.depth ? depth : r.depth;
        s3t = s2t; s3d = s2d; s2t = s1t; s2d = s1d; s1t = s0t; s1d = s0d;
        s0t = newtime; s0d = (kArNewPose + (EMIT ? o.nn0 : 0) + nn) | (depth << 16);
        ++nn; ++sp; ++r.splits;
        cdt = newtime;
        continue;
      }
      if (has_next) {
        if (sp > 0) s0t += cdt - dt_ref;
        else { pdt += cdt - dt_ref; ptouched = true; }
      }
      cdt = dt_ref;
    } else if (cdt < lo_lim) {                            // (.. && sizeTimeDiffs() > min_samples: checked by the sweep)
      if (has_next) {
        if (sp > 0) {
          cdt = s0t + cdt; --sp;
          s0t = s1t; s0d = s1d; s1t = s2t; s1d = s2d; s2t = s3t; s2d = s3d;
        } else {
          cdt = pdt + cdt; ++j;
          if (j < Tin) pdt = in_dt[j];
          ptouched = false;
        }
        ++r.merges;
        continue;
      }
      if (i > 0 || k > 0) {   // last interval and something was emitted before it: merged backwards
        if (EMIT) *o.tail_dt = cdt;
        r.tail = 1; ++r.merges;
        emit = false;
      }
    }
    if (!emit) break;          // (r.next stays Tin)
    if (EMIT) { o.out_desc[o.k0 + k] = cdesc; o.odt[o.k0 + k] = cdt; }
    ++k;
    if (sp > 0) {
      cdesc = s0d & 0xffff; cdepth = s0d >> 16; cdt = s0t; --sp;
      s0t = s1t; s0d = s1d; s1t = s2t; s1d = s2d; s2t = s3t; s2d = s3d;
    } else if (j < Tin) {
      const bool clean = !ptouched;
      cdesc = j; cdepth = 0; cdt = pdt; ++j;
      if (j < Tin) pdt = in_dt[j];
      ptouched = false;
      if (clean) { r.next = cdesc; break; }   // the machine is clean at input interval cdesc: the next chain starts there
    } else break;
  }
  r.emitted = k; r.new_poses = nn;
  return r;
}

}  // namespace tebamd
