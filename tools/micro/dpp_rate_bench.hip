// Issue rates of the lane-broadcast forms a 16-lane row of the cyclic reduction could take its operands through, one wave per SIMD
// (4 waves per workgroup, one workgroup): cycles per instruction over a long unrolled run of INDEPENDENT instructions.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/dpp_rate_bench.hip -o tools/micro/dpp_rate_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include "teb_kernel.hpp"

#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void __launch_bounds__(256) k(double* p, long long* out, int reps) {
  double a0 = p[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double x = p[threadIdx.x + 256], y = p[threadIdx.x + 512];
  int xi = (int)x, b0 = 0, b1 = 1, b2 = 2, b3 = 3, b4 = 4, b5 = 5, b6 = 6, b7 = 7;
  __syncthreads();
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0) {   // plain v_fma_f64 (v_fmac_f64), 8 independent accumulators
      REP8(asm volatile("v_fmac_f64 %0, %8, %9\n\tv_fmac_f64 %1, %8, %9\n\tv_fmac_f64 %2, %8, %9\n\tv_fmac_f64 %3, %8, %9\n\t"
                        "v_fmac_f64 %4, %8, %9\n\tv_fmac_f64 %5, %8, %9\n\tv_fmac_f64 %6, %8, %9\n\tv_fmac_f64 %7, %8, %9"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
    } else if (MODE == 1) {   // v_fmac_f64_dpp row_newbcast
      REP8(asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f64_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f64_dpp %4, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f64_dpp %6, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
    } else if (MODE == 2) {   // v_mov_b64_dpp row_newbcast
      REP8(asm volatile("v_mov_b64_dpp %0, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %1, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                        "v_mov_b64_dpp %2, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %3, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "v_mov_b64_dpp %4, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %5, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                        "v_mov_b64_dpp %6, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %7, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));)
    } else if (MODE == 3) {   // v_mov_b32_dpp row_newbcast
      REP8(asm volatile("v_mov_b32_dpp %0, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                        "v_mov_b32_dpp %2, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "v_mov_b32_dpp %4, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                        "v_mov_b32_dpp %6, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf"
                        : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(xi));)
    } else if (MODE == 4) {   // v_fmac_f32_dpp row_newbcast (what a 32-bit ALU op with a DPP source costs)
      REP8(asm volatile("v_fmac_f32_dpp %0, %8, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %8, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %2, %8, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %3, %8, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %4, %8, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %5, %8, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %6, %8, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %7, %8, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf"
                        : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(xi));)
    } else if (MODE == 5) {   // ds_read_b128 of one address per 8 lanes (the operand reads of the 8-lane round), 8 in flight
      extern __shared__ __attribute__((aligned(16))) double lds[];
      const double* q = lds + (threadIdx.x >> 3) * 16;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        double2 v0 = *reinterpret_cast<const double2*>(q); double2 v1 = *reinterpret_cast<const double2*>(q + 2); double2 v2 = *reinterpret_cast<const double2*>(q + 4);
        double2 v3 = *reinterpret_cast<const double2*>(q + 6); double2 v4 = *reinterpret_cast<const double2*>(q + 8); double2 v5 = *reinterpret_cast<const double2*>(q + 10);
        double2 v6 = *reinterpret_cast<const double2*>(q + 12); double2 v7 = *reinterpret_cast<const double2*>(q + 14);
        asm volatile("" :: "v"(v0.x), "v"(v1.x), "v"(v2.x), "v"(v3.x), "v"(v4.x), "v"(v5.x), "v"(v6.x), "v"(v7.x), "v"(v0.y), "v"(v1.y), "v"(v2.y), "v"(v3.y), "v"(v4.y), "v"(v5.y), "v"(v6.y), "v"(v7.y) : "memory");
      }
    } else if (MODE == 7) {   // dependent chain, plain
      REP8(asm volatile("v_fmac_f64 %0, %1, %2\n\tv_fmac_f64 %0, %1, %2\n\tv_fmac_f64 %0, %1, %2\n\tv_fmac_f64 %0, %1, %2\n\t"
                        "v_fmac_f64 %0, %1, %2\n\tv_fmac_f64 %0, %1, %2\n\tv_fmac_f64 %0, %1, %2\n\tv_fmac_f64 %0, %1, %2" : "+v"(a0) : "v"(x), "v"(y));)
    } else if (MODE == 8) {   // dependent chain on the accumulator, DPP source fixed
      REP8(asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(x), "v"(y));)
    } else if (MODE == 9) {   // the same with s_nop 1 in front of each
      REP8(asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(x), "v"(y));)
    } else if (MODE == 10) {   // the DPP source is the previous result (v_mov_b64_dpp chain, s_nop 1 in front of each)
      REP8(asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a0));)
    } else if (MODE == 11) {   // fmac whose DPP source is the previous fmac's result, s_nop 1 between (the pivot chain's shape)
      REP8(asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_fmac_f64_dpp %1, %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_fmac_f64_dpp %1, %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_fmac_f64_dpp %1, %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_fmac_f64_dpp %1, %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1) : "v"(y));)
    } else if (MODE == 12 || MODE == 14) {   // the passes of cr16_eliminate as they stand in the product header
      double Y[8] = {a0, a1, a2, a3, a4, a5, a6, a7}, wf[8] = {a7, a6, a5, a4, a3, a2, a1, a0}, v[8] = {x, y, x + 1, y + 1, x + 2, y + 2, x + 3, y + 3};
      double acc[16];
      for (int t = 0; t < 16; ++t) acc[t] = a0 + t;
      for (int u = 0; u < 8; ++u) {
        if (MODE == 12)
          asm volatile("s_nop 1\n\t" TEB_CR16_FORWARD : "+v"(Y[0]), "+v"(Y[1]), "+v"(Y[2]), "+v"(Y[3]), "+v"(Y[4]), "+v"(Y[5]), "+v"(Y[6]), "+v"(Y[7]),
                       "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(wf[4]), "+v"(wf[5]), "+v"(wf[6]), "+v"(wf[7])
                       : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]));
        else
          asm volatile("s_nop 1\n\t" TEB_CR16_SCHUR2
                       : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]),
                         "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15])
                       : "v"(v[0]), "v"(Y[0]), "v"(v[1]), "v"(Y[1]));
      }
      a0 = Y[0] + Y[7] + wf[3] + acc[0] + acc[15] + acc[7]; a1 = Y[1] + wf[1] + acc[1]; a2 = Y[2] + acc[2] + wf[2]; a3 = Y[3] + acc[3]; a4 = Y[4] + acc[4]; a5 = Y[5] + acc[5]; a6 = Y[6] + acc[6] + acc[8] + acc[9] + acc[10] + acc[11] + acc[12] + acc[13] + acc[14];
    } else if (MODE == 15) {   // fmac with the negated DPP source, 8 independent accumulators
      REP8(asm volatile("v_fmac_f64_dpp %0, -%8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, -%8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f64_dpp %2, -%8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, -%8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f64_dpp %4, -%8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, -%8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f64_dpp %6, -%8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, -%8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
    } else if (MODE == 16) {   // two accumulators alternating (the shape of a substitution row), source operand produced long ago
      REP8(asm volatile("v_fmac_f64_dpp %0, -%2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, -%2, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f64_dpp %0, -%2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, -%2, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f64_dpp %0, -%2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, -%2, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f64_dpp %0, -%2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, -%2, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                        : "+v"(a0), "+v"(a1) : "v"(x), "v"(y), "v"(a7));)
    } else if (MODE == 17) {   // src1 is the result of the previous instruction (through an ordinary source, not the accumulator)
      REP8(asm volatile("v_fmac_f64_dpp %0, -%2, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, -%2, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f64_dpp %0, -%2, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, -%2, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f64_dpp %0, -%2, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, -%2, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f64_dpp %0, -%2, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, -%2, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                        : "+v"(a0), "+v"(a1) : "v"(x));)
    } else if (MODE == 6) {   // v_readlane_b32 x2 + v_fma with an SGPR-pair operand: the wave-uniform broadcast
      REP8(asm volatile("v_readlane_b32 s20, %8, 3\n\tv_readlane_b32 s21, %9, 3\n\ts_nop 3\n\t"
                        "v_fmac_f64 %0, s[20:21], %10\n\tv_fmac_f64 %1, s[20:21], %10\n\tv_fmac_f64 %2, s[20:21], %10\n\tv_fmac_f64 %3, s[20:21], %10\n\t"
                        "v_fmac_f64 %4, s[20:21], %10\n\tv_fmac_f64 %5, s[20:21], %10\n\tv_fmac_f64 %6, s[20:21], %10\n\tv_fmac_f64 %7, s[20:21], %10"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "v"(y) : "s20", "s21");)
    }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  p[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
}

template <int MODE> static void run(const char* what, int per_rep) {
  double* p; long long* o;
  hipMalloc(&p, 1024 * sizeof(double)); hipMalloc(&o, 8);
  hipMemset(p, 0, 1024 * sizeof(double));
  const int reps = 200;
  for (int it = 0; it < 2; ++it) { hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(256), 4096, 0, p, o, reps); hipDeviceSynchronize(); }
  long long c; hipMemcpy(&c, o, 8, hipMemcpyDeviceToHost);
  printf("%-64s %6.2f cycles per instruction (4 waves, one per SIMD)\n", what, (double)c / reps / per_rep);
  hipFree(p); hipFree(o);
}
int main() {
  run<0>("v_fmac_f64 (plain)", 64);
  run<1>("v_fmac_f64_dpp row_newbcast", 64);
  run<2>("v_mov_b64_dpp row_newbcast", 64);
  run<3>("v_mov_b32_dpp row_newbcast", 64);
  run<4>("v_fmac_f32_dpp row_newbcast", 64);
  run<5>("ds_read_b128, one address per 8 lanes", 64);
  run<6>("2 v_readlane_b32 + 8 v_fmac_f64 with the SGPR pair (per 10)", 8);
  run<7>("dependent v_fmac_f64 chain", 64);
  run<8>("dependent v_fmac_f64_dpp chain (accumulator)", 64);
  run<9>("dependent v_fmac_f64_dpp chain, s_nop 1 in front of each", 64);
  run<10>("v_mov_b64_dpp chain through the DPP source, s_nop 1 each", 64);
  run<11>("v_fmac_f64_dpp chain through the DPP source, s_nop 1 each", 64);
  run<12>("TEB_CR16_FORWARD (56 fmac + nop), cycles per fmac", 8 * 56);
  run<14>("TEB_CR16_SCHUR2 (32 fmac + nop), cycles per fmac", 8 * 32);
  run<15>("v_fmac_f64_dpp with negated source, independent", 64);
  run<16>("v_fmac_f64_dpp two alternating accumulators", 64);
  run<17>("v_fmac_f64_dpp, src1 = previous result", 64);
  return 0;
}
