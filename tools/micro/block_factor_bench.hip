// "one wave per partition running a sequential 8x8-block LDL^T with ONE MATRIX ELEMENT PER LANE". The building
// block of that design is the factorisation of an 8x8 pivot block by the 64 lanes of a wave together; the building block of what the
// kernel does today (block cyclic reduction, teb_kernel.hpp) is the same factorisation done by every lane on its own copy in registers
// (Ldl8::factor, redundant over the 8 lanes of a group). In a sequential partition sweep the block factorisation sits on the critical
// path of EVERY block row, so its latency bounds the design from below: rows per partition x (factor + the two triangular products).
// This benchmark measures both, from LDS-resident blocks, at one wave per SIMD (the optimise kernel's occupancy), alone and with all four
// waves of the workgroup busy:
//   A  per lane, in registers: load the lower triangle (36 values), 8 pivots, everything unrolled       (today)
//   B  one element per lane: lane (r, c) holds D[r][c]; per pivot k the lanes need 1 / d_k, l_rk and l_ck - the value of lane (k, k),
//      of lane (r, k) (same 8-lane group: ds_swizzle broadcast) and of lane (c, k) (another group: ds_bpermute)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I teb_local_planner_amd/csrc tools/micro/block_factor_bench.hip -o tools/micro/block_factor_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "teb_kernel.hpp"
using namespace tebamd;

__device__ __forceinline__ double bperm_f64(double v, int src_lane) {
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <int K>
__device__ __forceinline__ double pivot_step(double a, int r, int c, int lane) {
  // after step K - 1 lane (r, c), r, c >= K, holds the Schur complement entry; lanes of row / column K hold the pivot row / column
  const double dk = rl_f64(a, 9 * K);                       // D[K][K]: wave-uniform -> v_readlane
  const double inv = fast_rcp(dk);
  const double ark = bcast8<K>(a);                          // D[r][K]: lane (r, K) of my own 8-lane group
  const double ack = bperm_f64(a, 8 * c + K);               // D[c][K]: lane (c, K)
  double out = a;
  if (r > K && c > K) out = a - (ark * inv) * ack;          // trailing update
  else if (r > K && c == K) out = a * inv;                  // l_rK
  else if (r == K && c == K) out = inv;                     // 1 / d_K on the diagonal, like Ldl8
  (void)lane;
  return out;
}

// which = 0: variant A, 1: variant B. Each active wave factors `reps` blocks one after the other (dependent through a checksum that is
// added to the next block's diagonal, as the Schur complement of one block row feeds the next in a sweep).
__global__ void __launch_bounds__(kThreads) factor_kernel(int which, int waves, int reps, long long* cycles, double* sink) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  double* D = lds + wv * 32 * kBlk;     // 32 blocks per wave
  for (int q = lane; q < 32 * kBlk; q += 64) { const int w = q % kBlk, r = w >> 3, c = w & 7; D[q] = (w < 64 && r == c) ? 20.0 + 0.01 * (q % 7) : 0.01 * ((q * 7) % 13); }
  __syncthreads();
  double carry = 0;
  const long long t0 = clock64();
  if (wv < waves) {
    if (which == 0) {
      for (int it = 0; it < reps; ++it) {
        Ldl8 F;
        F.load(D + (it & 31) * kBlk);
        F.a[0] += carry;
        const bool ok = F.factor();
        carry = ok ? 1e-9 * F.a[Ldl8::idx(7, 7)] : 1.0;
      }
    } else {
      const int r = lane >> 3, c = lane & 7;
      for (int it = 0; it < reps; ++it) {
        const double* Di = D + (it & 31) * kBlk;
        double a = Di[(r >= c ? r : c) * 8 + (r >= c ? c : r)];   // symmetric: both triangles from the stored lower one
        if (lane == 0) a += carry;
        a = pivot_step<0>(a, r, c, lane); a = pivot_step<1>(a, r, c, lane); a = pivot_step<2>(a, r, c, lane); a = pivot_step<3>(a, r, c, lane);
        a = pivot_step<4>(a, r, c, lane); a = pivot_step<5>(a, r, c, lane); a = pivot_step<6>(a, r, c, lane); a = pivot_step<7>(a, r, c, lane);
        carry = 1e-9 * rl_f64(a, 63);
      }
    }
  }
  const long long t1 = clock64();
  if (lane == 0 && wv == 0) cycles[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * kThreads + tid] = carry;
}

int main() {
  long long* d_c; double* d_s;
  hipMalloc(&d_c, 256 * sizeof(long long)); hipMalloc(&d_s, 256 * kThreads * sizeof(double));
  hipFuncSetAttribute((const void*)factor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int reps = 256;
  for (int grid : {1, 256})
    for (int which : {0, 1})
      for (int waves : {1, 4}) {
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
          hipLaunchKernelGGL(factor_kernel, dim3(grid), dim3(kThreads), 150 * 1024, 0, which, waves, reps, d_c, d_s);
          hipDeviceSynchronize();
          std::vector<long long> c(grid);
          hipMemcpy(c.data(), d_c, grid * sizeof(long long), hipMemcpyDeviceToHost);
          double sum = 0; for (auto v : c) sum += v;
          best = std::min(best, sum / grid / reps);
        }
        printf("%s, %d wave(s) busy, %3d workgroups: %7.0f cycles per dependent 8x8 LDL^T\n",
               which == 0 ? "A per lane in registers (today)   " : "B one matrix element per lane     ", waves, grid, best);
      }
  return 0;
}
