// 64-bit DPP under a bank mask on gfx950 (v_fmac_f64_dpp / v_mov_b64_dpp, row_newbcast:N): which lanes are written, and what happens to the
// accumulator when the previous instruction has just written it. Measured on MI355X (round 6):
//   * bank_mask works for 64-bit DPP: masked-off lanes keep their destination / accumulator;
//   * HAZARD (not in the ISA guide's table, not covered by the compiler for inline asm): a DPP v_fmac_f64 that follows IMMEDIATELY on a
//     DPP v_fmac_f64 with the same accumulator, either of them bank-masked, goes wrong in the lanes either mask excludes - the second
//     one's masked-off lanes lose the first one's write, the lanes the first one masked off accumulate onto a stale value (rows 1, B, C);
//   * any instruction in between (s_nop 0, an unrelated VALU op) and it is correct; full masks chain correctly (the reduction rounds);
//     a plain VALU read after a masked write, a plain write before a masked fmac and back-to-back masked MOVES are correct (A, D, E).
// csrc/teb_kernel.hpp (TEB_L0_O1_DPP .., the D_i broadcast of level 0) orders its statements so that no DPP instruction follows directly on
// one that wrote its destination. Exit code 0 iff every pattern the kernel RELIES on behaves (tests/test_gpu_cr_rounds.py).
// Build: hipcc --offload-arch=gfx950 -O2 tools/micro/dpp64_mask_probe.hip -o tools/micro/dpp64_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int GAP> __device__ double two(double x) {   // lower half from lane 0, then upper half from lane 8, GAP wait states between
  double acc = 7.0, o = 1.0;
  if constexpr (GAP < 0)
    asm volatile("s_nop 4\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0x3\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xc" : "+v"(acc) : "v"(x), "v"(o));
  else
    asm volatile("s_nop 4\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0x3\n\ts_nop %3\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xc" : "+v"(acc) : "v"(x), "v"(o), "n"(GAP));
  return acc;
}
__device__ double two_other_between(double x) {   // an independent VALU instruction between the two
  double acc = 7.0, o = 1.0, t = 3.0;
  asm volatile("s_nop 4\n\tv_fmac_f64_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0x3\n\tv_add_f64 %1, %1, %1\n\t"
               "v_fmac_f64_dpp %0, %2, %3 row_newbcast:8 row_mask:0xf bank_mask:0xc" : "+v"(acc), "+v"(t) : "v"(x), "v"(o));
  return acc + 0 * t;
}
__device__ double same_mask_chain(double x) {   // dependent chain, full mask (what the reduction rounds do): 7 + x0 + x8
  double acc = 7.0, o = 1.0;
  asm volatile("s_nop 4\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f64_dpp %0, %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(o));
  return acc;
}
__device__ double masked_then_plain_read(double x) {   // (A) masked fmac, then a plain VALU instruction reads the accumulator: want 2 * (7 + x0) lower, 14 upper
  double acc = 7.0, o = 1.0, r = 0.0;
  asm volatile("s_nop 4\n\tv_fmac_f64_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0x3\n\tv_add_f64 %1, %0, %0" : "+v"(acc), "=&v"(r) : "v"(x), "v"(o));
  return r;
}
__device__ double masked_then_full(double x) {   // (B) masked fmac, then a full-mask DPP fmac on the same accumulator: want 7 + x0 + x8 lower, 7 + x8 upper
  double acc = 7.0, o = 1.0;
  asm volatile("s_nop 4\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0x3\n\t"
               "v_fmac_f64_dpp %0, %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(o));
  return acc;
}
__device__ double full_then_masked(double x) {   // (C) full-mask fmac, then a masked one: want 7 + x0 lower, 7 + x0 + x8 upper
  double acc = 7.0, o = 1.0;
  asm volatile("s_nop 4\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f64_dpp %0, %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xc" : "+v"(acc) : "v"(x), "v"(o));
  return acc;
}
__device__ double masked_movs(double x) {   // (D) two masked moves into the same destination back to back: want x0 lower, x8 upper
  double d = -1.0;
  asm volatile("s_nop 4\n\tv_mov_b64_dpp %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0x3\n\t"
               "v_mov_b64_dpp %0, %1 row_newbcast:8 row_mask:0xf bank_mask:0xc" : "+v"(d) : "v"(x));
  return d;
}
__device__ double plain_then_masked(double x) {   // (E) a plain VALU write of the accumulator, then a masked fmac: want 14 + x0 lower, 14 upper
  double acc = 7.0, o = 1.0;
  asm volatile("s_nop 4\n\tv_add_f64 %0, %0, %0\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0x3" : "+v"(acc) : "v"(x), "v"(o));
  return acc;
}
__global__ void probe(double* out) {
  const int lane = threadIdx.x;
  double x = 100.0 + lane;
  double r[12] = {two<-1>(x), two<0>(x), two<1>(x), two<3>(x), two<7>(x), two_other_between(x), same_mask_chain(x), masked_then_plain_read(x), masked_then_full(x), full_then_masked(x), masked_movs(x), plain_then_masked(x)};
  for (int k = 0; k < 12; ++k) out[k * 64 + lane] = r[k];
}
int main() {
  double* d; (void)hipMalloc(&d, 12 * 64 * sizeof(double));
  probe<<<1, 64>>>(d);
  static double h[12 * 64]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[12] = {"lower then upper, back to back   (want 107 x 8, 115 x 8)", "s_nop 0 between                  (want 107 x 8, 115 x 8)", "s_nop 1 between                  (want 107 x 8, 115 x 8)",
                        "s_nop 3 between                  (want 107 x 8, 115 x 8)", "s_nop 7 between                  (want 107 x 8, 115 x 8)", "v_add_f64 between                (want 107 x 8, 115 x 8)",
                        "full masks, dependent chain      (want 215 x 16)        ", "(A) masked, then plain read      (want 214 x 8, 14 x 8) ", "(B) masked, then full mask       (want 215 x 8, 115 x 8)",
                        "(C) full mask, then masked       (want 107 x 8, 215 x 8)", "(D) masked moves, same dest      (want 100 x 8, 108 x 8)", "(E) plain write, then masked     (want 114 x 8, 14 x 8) "};
  for (int k = 0; k < 12; ++k) { printf("%s:", nm[k]); for (int l = 0; l < 16; ++l) printf(" %g", h[64 * k + l]); printf("\n"); }
  // the patterns the kernel relies on (all 64 lanes; rows of 16): one instruction between, the full-mask chain, A, D, E
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int row0 = l & ~15; const bool lo = (l & 8) == 0;
    const double x0 = 100.0 + row0, x8 = 108.0 + row0;
    for (int k : {1, 2, 3, 4, 5}) bad += h[64 * k + l] != (lo ? 7 + x0 : 7 + x8);
    bad += h[64 * 6 + l] != 7 + x0 + x8;
    bad += h[64 * 7 + l] != (lo ? 2 * (7 + x0) : 14.0);
    bad += h[64 * 10 + l] != (lo ? x0 : x8);
    bad += h[64 * 11 + l] != (lo ? 14 + x0 : 14.0);
  }
  bool hazard_seen = false;
  for (int l = 0; l < 64; ++l) hazard_seen |= h[l] != (((l & 8) == 0) ? 107.0 + (l & ~15) : 115.0 + (l & ~15));
  printf("relied-upon patterns: %d wrong values; back-to-back masked fmac hazard %s on this device\n", bad, hazard_seen ? "PRESENT" : "absent");
  return bad ? 1 : 0;
}
