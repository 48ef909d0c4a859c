// Micro-benchmark of one round of the block cyclic reduction (cr_forward_round<M>) on LDS-resident blocks: cycles per round for each
// group width with 1 .. 4 waves of the workgroup active, alone on a CU (1 workgroup) or with every CU busy (256 workgroups).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I teb_local_planner_amd/csrc tools/micro/cr_round_bench.hip -o tools/micro/cr_round_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "teb_kernel.hpp"
using namespace tebamd;

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

int main() {
  for (int grid : {1, 256}) {
    run<1>(1, grid); run<1>(8, grid); run<1>(16, grid); run<1>(32, grid);
    run<2>(1, grid); run<2>(16, grid);
    run<4>(1, grid); run<4>(8, grid);
    run<8>(1, grid); run<8>(4, grid);
  }
  return 0;
}
