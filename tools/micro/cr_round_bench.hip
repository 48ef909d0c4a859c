// Micro-benchmark of one round of the block cyclic reduction on LDS-resident blocks: the product's round of 16-lane rows (operands by
// DPP, cr_forward_round16) and the LDS-operand round of rounds 2 - 4 (cr_forward_round<M>, below) per group width, with 1 .. 4 waves of the
// workgroup active, alone on a CU (1 workgroup) or with every CU busy (256 workgroups); and how far their results are apart.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I teb_local_planner_amd/csrc tools/micro/cr_round_bench.hip -o tools/micro/cr_round_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
// cycle stamps of ONE 16-lane round (lane 0, the last repetition): s_memtime into LDS, no memory traffic between the stamps
__shared__ long long s_stamp[16];
#ifdef CR_BENCH_STAMPS
#define TEB_CR16_STAMP(k) { __builtin_amdgcn_sched_barrier(0); const long long now_ = clock64(); if (threadIdx.x == 0) s_stamp[k] = now_; __builtin_amdgcn_sched_barrier(0); }
#endif
#include "teb_kernel.hpp"
using namespace tebamd;

// The round of rounds 2 - 4 (kept here for the comparison): 8 M lanes per elimination, lane (q, c) owns rows q R .. q R + R - 1 of column
// c of the three Schur products, every lane factors D_i in its own registers (Ldl8) and streams the product operands - the whole of L_i
// and L_{i+s} - from LDS: 196 doubles per lane and round at M = 1, the pipe the four waves share.
namespace tebamd {
template <int M>
__device__ __forceinline__ bool cr_forward_round(double* __restrict__ D, double* __restrict__ L, double* __restrict__ f, int Nb, int s,
                                                 int e0, int E, int bD = kBlk, int bF = 8) {
  TEB_SOLVER_FMA
  constexpr int R = 8 / M;
  constexpr int kW = M == 1 ? 0 : M == 2 ? 1 : M == 4 ? 2 : 3;   // (profiling build) row of the per-width counters
  (void)kW;
  const int tid = threadIdx.x;
  const int grp = tid / (8 * M), c = tid & 7, q = (tid >> 3) & (M - 1), a0 = q * R;
  const int e = e0 + grp;
  const bool act = e < E;
  const int i = s * (2 * e + 1);
  const bool hasU = act && (i + s < Nb);
  bool ok = true;
  double wL[8], wU[8], wf[8], o1[R], o2[R], o3[R];
  double s1 = 0, s2 = 0;
  CRR_DECL
  if (act) {
    const double* Di = D + i * bD;
    const double* Li = L + i * bD;
    const double* Lp = L + (i + s) * bD;   // U_i^T, valid iff hasU
    Ldl8 F;
    F.load(Di);
    ok = F.factor();
    CRR(0);
    double cu[8];
    ld_row<8>((hasU ? Lp : Li) + c * 8, cu);   // row c of L_{i+s} (an address inside the blocks even without an upper neighbour)
    ld_row<8>(f + i * bF, wf);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      wL[k] = Li[k * 8 + c];
      wU[k] = hasU ? cu[k] : 0.0;
    }
    F.solve3(wL, wU, wf);
    CRR(1);
#pragma unroll
    for (int t = 0; t < R; ++t) { o1[t] = 0; o2[t] = 0; o3[t] = 0; }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      double li[R];
      ld_row<R>(Li + k * 8 + a0, li);
#pragma unroll
      for (int t = 0; t < R; ++t) o1[t] += li[t] * wL[k];
      s1 += Li[k * 8 + c] * wf[k];
      if ((k % TEB_CR_FENCE_EVERY) == TEB_CR_FENCE_EVERY - 1) { TEB_CR_SCHED_BARRIER }
    }
    if (hasU) {
#pragma unroll
      for (int t = 0; t < R; ++t) {
        double lp[8];
        ld_row<8>(Lp + (a0 + t) * 8, lp);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          o2[t] -= lp[k] * wL[k];
          o3[t] += lp[k] * wU[k];
        }
        if ((t % TEB_CR_FENCE_EVERY) == TEB_CR_FENCE_EVERY - 1) { TEB_CR_SCHED_BARRIER }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) s2 += cu[k] * wf[k];
    }
  }
  CRR(2);
  // No barrier before the writes: within a level the eliminated rows i = s (2 e + 1) and the blocks read for them (D_i, L_i,
  // L_{i+s}, f_i) belong to exactly one group, a group never straddles two waves (8 M <= 64), and the survivors' D / f are only written
  // (never read) in this level.
  if (act) {
    double* Dm = D + (i - s) * bD;
    double* Di = D + i * bD;
    double* Li = L + i * bD;
#pragma unroll
    for (int t = 0; t < R; ++t) Dm[(a0 + t) * 8 + c] -= o1[t];
    if (hasU) {
      double* Lp = L + (i + s) * bD;
#pragma unroll
      for (int t = 0; t < R; ++t) Lp[(a0 + t) * 8 + c] = o2[t];
    }
    if (q == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { Di[k * 8 + c] = wL[k]; Li[k * 8 + c] = wU[k]; }
      f[(i - s) * bF + c] -= s1;
      f[i * bF + c] = wf[c];
    }
  }
  CRR(3);
  __syncthreads();
  CRR(4);
  if (hasU) {
    double* Dp = D + (i + s) * bD;
#pragma unroll
    for (int t = 0; t < R; ++t) Dp[(a0 + t) * 8 + c] -= o3[t];
    if (q == 0) f[(i + s) * bF + c] -= s2;
  }
  __syncthreads();
  CRR(5);
  return ok;
}
}  // namespace tebamd

template <int M>
__global__ void __launch_bounds__(kThreads) round_kernel(int Nb, int s, int E, int reps, long long* cycles, double* sink) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* D = lds;
  double* L = D + Nb * kBlk;
  double* f = L + Nb * kBlk;
  for (int q = threadIdx.x; q < Nb * kBlk; q += kThreads) {
    const int w = q % kBlk, r = w >> 3, c = w & 7;
    D[q] = (w < 64 && r == c) ? 20.0 + 0.01 * (q % 7) : 0.01 * ((q * 7) % 13);
    L[q] = 0.02 * ((q * 5) % 11) - 0.1;
  }
  for (int q = threadIdx.x; q < Nb * 8; q += kThreads) f[q] = 0.5 + 0.001 * q;
  __syncthreads();
  bool ok = true;
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) ok = cr_forward_round<M>(D, L, f, Nb, s, 0, E) && ok;
  const long long t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  if (!ok && threadIdx.x == 0) sink[0] = D[0];
  if (threadIdx.x == 0 && blockIdx.x == 0) sink[1] = D[1] + f[3];
}

// the 16-lane round (operands by row_newbcast from the neighbours' registers): E eliminations take ceil(E / 16) rounds
__global__ void __launch_bounds__(kThreads) round16_kernel(int Nb, int s, int E, int reps, long long* cycles, double* sink) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* D = lds;
  double* L = D + Nb * kBlk;
  double* f = L + Nb * kBlk;
  for (int q = threadIdx.x; q < Nb * kBlk; q += kThreads) {
    const int w = q % kBlk, r = w >> 3, c = w & 7;
    D[q] = (w < 64 && r == c) ? 20.0 + 0.01 * (q % 7) : 0.01 * ((q * 7) % 13);
    L[q] = 0.02 * ((q * 5) % 11) - 0.1;
  }
  for (int q = threadIdx.x; q < Nb * 8; q += kThreads) f[q] = 0.5 + 0.001 * q;
  __syncthreads();
  bool ok = true;
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r)
    ok = cr_forward(D, L, f, 2 * E + 1, s, 2 * s) && ok;   // one level: ceil(E / 16) rounds
  const long long t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  if (!ok && threadIdx.x == 0) sink[0] = D[0];
  if (threadIdx.x == 0 && blockIdx.x == 0) sink[1] = D[1] + f[3];
#ifdef CR_BENCH_STAMPS
  if (threadIdx.x < 16 && blockIdx.x == 0) sink[2 + threadIdx.x] = (double)(s_stamp[threadIdx.x] - s_stamp[5]);
#endif
}
static void run16(int E, int grid) {
  const int s = 1, Nb = 2 * E + 1, reps = 50;
  long long* d_c; double* d_s;
  hipMalloc(&d_c, grid * sizeof(long long)); hipMalloc(&d_s, 32 * sizeof(double));
  hipFuncSetAttribute((const void*)round16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int it = 0; it < 2; ++it) {
    hipLaunchKernelGGL(round16_kernel, dim3(grid), dim3(kThreads), 150 * 1024, 0, Nb, s, E, reps, d_c, d_s);
    hipDeviceSynchronize();
  }
  std::vector<long long> c(grid);
  hipMemcpy(c.data(), d_c, grid * sizeof(long long), hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : c) sum += v;
  const int rounds = (E + 15) / 16;
  printf("  16-lane rows  , %3d eliminations (%d round%s), %3d workgroups: %7.0f cycles per level, %7.0f per round\n", E, rounds, rounds > 1 ? "s" : " ",
         grid, sum / grid / reps, sum / grid / reps / rounds);
#ifdef CR_BENCH_STAMPS
  if (grid == 1) {
    double st[18];
    hipMemcpy(st, d_s, sizeof st, hipMemcpyDeviceToHost);
    const double* t = st + 2;
    printf("      last round, lane 0, cycles since entry: loads done %.0f | factor %.0f | forward %.0f | scale + backward %.0f | products %.0f | stores 1 %.0f | barrier %.0f | stores 2 %.0f | barrier %.0f\n",
           t[0], t[1], t[2], t[3], t[4], t[6], t[7], t[8], t[9]);
  }
#endif
  hipFree(d_c); hipFree(d_s);
}

// same system through one level of 8-lane rounds and of 16-lane rounds: every entry a later level or the back substitution reads must
// come out with the same bits (D of the surviving rows: lower triangle; the slots of the eliminated rows, the new couplings, f: all)
template <int WHICH>
__global__ void __launch_bounds__(kThreads) level_kernel(int Nb, int E, double* out) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* D = lds;
  double* L = D + Nb * kBlk;
  double* f = L + Nb * kBlk;
  for (int q = threadIdx.x; q < Nb * kBlk; q += kThreads) {
    const int w = q % kBlk, r = w >> 3, c = w & 7, j = q / kBlk;
    const int lo = r > c ? c : r, hi = r > c ? r : c;
    D[q] = w >= 64 ? 0.0 : (r == c) ? 20.0 + 0.37 * ((j * 5 + r) % 7) : 0.6 * (((j * 31 + hi * 8 + lo) * 7) % 13) / 13.0 - 0.3;   // symmetric, diagonally dominant
    L[q] = w >= 64 ? 0.0 : 0.9 * (((q * 5) % 11) / 11.0) - 0.45;
  }
  for (int q = threadIdx.x; q < Nb * 8; q += kThreads) f[q] = 0.5 + 0.013 * ((q * 3) % 17);
  __syncthreads();
  bool ok = true;
  if (WHICH == 0) { for (int e0 = 0; e0 < E; e0 += kThreads / 8) ok = cr_forward_round<1>(D, L, f, Nb, 1, e0, E) && ok; }
  else ok = cr_forward(D, L, f, Nb, 1, 2);   // one level
  __syncthreads();
  for (int q = threadIdx.x; q < (2 * Nb * kBlk + Nb * 8); q += kThreads) out[q] = lds[q];
  if (threadIdx.x == 0) out[2 * Nb * kBlk + Nb * 8] = ok ? 1.0 : 0.0;
}
static long compare_levels(int E) {
  const int Nb = 2 * E + 1;
  const size_t count = (size_t)2 * Nb * kBlk + Nb * 8 + 1;
  double* d[2];
  std::vector<double> h[2];
  for (int w = 0; w < 2; ++w) {
    hipMalloc(&d[w], count * sizeof(double));
    if (w == 0) { hipFuncSetAttribute((const void*)level_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                  hipLaunchKernelGGL(level_kernel<0>, dim3(1), dim3(kThreads), 150 * 1024, 0, Nb, E, d[w]); }
    else        { hipFuncSetAttribute((const void*)level_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                  hipLaunchKernelGGL(level_kernel<1>, dim3(1), dim3(kThreads), 150 * 1024, 0, Nb, E, d[w]); }
    hipDeviceSynchronize();
    h[w].resize(count);
    hipMemcpy(h[w].data(), d[w], count * sizeof(double), hipMemcpyDeviceToHost);
    hipFree(d[w]);
  }
  long diff = 0, checked = 0; double worst = 0;
  for (int j = 0; j < Nb; ++j)
    for (int w = 0; w < 64; ++w) {
      const int r = w >> 3, c = w & 7;
      for (int which = 0; which < 2; ++which) {   // D, L
        if (which == 0 && (j % 2 == 0) && c > r) continue;   // upper triangle of a surviving diagonal block: never read
        if (which == 1 && j == 0) continue;                  // row 0 has no coupling
        const size_t q = (size_t)which * Nb * kBlk + (size_t)j * kBlk + w;
        ++checked;
        if (memcmp(&h[0][q], &h[1][q], 8)) { ++diff; worst = fmax(worst, fabs(h[0][q] - h[1][q])); }
      }
    }
  for (int q = 0; q < Nb * 8; ++q) { ++checked; const size_t a = (size_t)2 * Nb * kBlk + q; if (memcmp(&h[0][a], &h[1][a], 8)) { ++diff; worst = fmax(worst, fabs(h[0][a] - h[1][a])); } }
  printf("  level of %3d eliminations, 8-lane vs 16-lane rounds: %ld of %ld entries differ (worst %.3e), ok %g / %g\n", E, diff, checked, worst,
         h[0][count - 1], h[1][count - 1]);
  return diff + (h[0][count - 1] != h[1][count - 1] ? 1 : 0);
}

template <int M>
static void run(int E, int grid) {
  const int s = 1, Nb = 2 * E + 1, reps = 50;
  long long* d_c; double* d_s;
  hipMalloc(&d_c, grid * sizeof(long long)); hipMalloc(&d_s, 16);
  const size_t lds = (size_t)(2 * Nb * kBlk + Nb * 8) * sizeof(double);
  hipFuncSetAttribute((const void*)round_kernel<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int it = 0; it < 2; ++it) {
    hipLaunchKernelGGL(round_kernel<M>, dim3(grid), dim3(kThreads), 150 * 1024, 0, Nb, s, E, reps, d_c, d_s);
    hipDeviceSynchronize();
  }
  std::vector<long long> c(grid);
  hipMemcpy(c.data(), d_c, grid * sizeof(long long), hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : c) sum += v;
  printf("  %2d-lane groups, %3d eliminations (%d waves busy), %3d workgroups: %7.0f cycles per round (lds %zu B)\n", 8 * M, E, (E * 8 * M + 63) / 64,
         grid, sum / grid / reps, lds);
  hipFree(d_c); hipFree(d_s);
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "--compare")) {
    // tests/test_gpu_cr_rounds.py: the inline-asm DPP rounds of the product against the retained 8-lane LDS-operand round, bit for bit, at
    // level sizes either side of every boundary of the round structure (one group, a partial round, exactly one round, the PAIR 1 / 2
    // boundary at 16 | 17 and 32 | 33, a partial second pair)
    long bad = 0;
    for (int E : {1, 2, 5, 15, 16, 17, 24, 31, 32, 33, 40, 47, 48, 49, 58, 64}) bad += compare_levels(E);
    printf("%s\n", bad ? "ROUNDS DIFFER" : "rounds identical");
    return bad ? 1 : 0;
  }
  for (int E : {1, 5, 16, 32, 58}) compare_levels(E);
  for (int grid : {1, 256}) {
    run16(1, grid); run16(8, grid); run16(16, grid); run16(32, grid); run16(58, grid);
    run<1>(1, grid); run<1>(8, grid); run<1>(16, grid); run<1>(32, grid);
    run<2>(1, grid); run<2>(16, grid);
    run<4>(1, grid); run<4>(8, grid);
    run<8>(1, grid); run<8>(4, grid);
  }
  return 0;
}
