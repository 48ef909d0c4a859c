// Dependent-instruction latencies of the primitives the optimise kernel's critical path is made of, measured the way the kernel runs
// them: one wave per SIMD (a 256-thread workgroup with a 150 KB LDS request, i.e. one workgroup per CU), clock64() around long
// dependent chains, alone on the chip and with every CU busy. bench.py's roofline.latency_model multiplies these by the chain lengths
// of one LM iteration (HISTORY.md section 4).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/micro/latency_probe.hip -o tools/micro/latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

constexpr int kThreads = 256;
constexpr int kChain = 512;

__device__ __forceinline__ double fast_rcp(double d) {   // the pivot reciprocal of the cyclic reduction (teb_kernel.hpp)
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}

// what: 0 fma, 1 fast_rcp, 2 sqrt, 3 div, 4 sincos, 5 LDS pointer chase, 6 global pointer chase over 16 KB (vector-L1 hits), 7 __syncthreads,
//       8 DPP move pair, 9 global pointer chase over 1 MB (L2 hits: beyond the 32 KB L1, inside the 4 MB L2 of the XCD)
__global__ void __launch_bounds__(kThreads) probe(int what, long long* cycles, double* sink, const int* chase, double seed) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  int* li = reinterpret_cast<int*>(lds);
  for (int q = threadIdx.x; q < 4096; q += kThreads) li[q] = (q * 97 + 13) & 4095;   // a permutation-free walk is fine: what matters is the dependence
  __syncthreads();
  double x = seed + 1e-9 * threadIdx.x, y = 1.0000001;
  int p = threadIdx.x & 4095;
  const long long t0 = clock64();
  if (what == 0) {
#pragma unroll 16
    for (int k = 0; k < kChain; ++k) x = fma(x, y, 1e-9);
  } else if (what == 1) {
#pragma unroll 8
    for (int k = 0; k < kChain; ++k) x = fast_rcp(x) + 1.0;   // (+ 1 keeps the value in range; one add is part of the measured step)
  } else if (what == 2) {
#pragma unroll 8
    for (int k = 0; k < kChain; ++k) x = sqrt(x) + 1.0;
  } else if (what == 3) {
#pragma unroll 8
    for (int k = 0; k < kChain; ++k) x = y / x + 1.0;
  } else if (what == 4) {
#pragma unroll 2
    for (int k = 0; k < kChain; ++k) { double s, c; sincos(x, &s, &c); x = s + c; }
  } else if (what == 5) {
#pragma unroll 8
    for (int k = 0; k < kChain; ++k) p = li[p];
  } else if (what == 6) {
#pragma unroll 8
    for (int k = 0; k < kChain; ++k) p = chase[p];
  } else if (what == 7) {
    for (int k = 0; k < kChain; ++k) __syncthreads();
  } else if (what == 9) {
    p = (threadIdx.x * 4099 + blockIdx.x * 131) & 262143;
#pragma unroll 8
    for (int k = 0; k < kChain; ++k) p = chase[4096 + p];
  } else {
#pragma unroll 8
    for (int k = 0; k < kChain; ++k) {
      const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0xB1, 0xf, 0xf, true);
      const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0xB1, 0xf, 0xf, true);
      x += __hiloint2double(hi, lo);
    }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * kThreads + threadIdx.x] = x + p;
}

int main() {
  const char* names[] = {"fma_f64", "fast_rcp_f64_plus_add", "sqrt_f64_plus_add", "div_f64_plus_add", "sincos_f64_plus_add", "lds_load", "l1_load", "syncthreads_4_waves", "dpp_move_f64_plus_add", "l2_load"};
  std::vector<int> h(4096 + 262144);
  for (int q = 0; q < 4096; ++q) h[q] = (q * 97 + 13) & 4095;
  for (unsigned q = 0; q < 262144; ++q) h[4096 + q] = (int)((q * 1664525u + 1013904223u) & 262143u);   // a bijection: the walk of a lane never short-cycles
  int* d_chase; long long* d_c; double* d_s;
  hipMalloc(&d_chase, h.size() * sizeof(int)); hipMemcpy(d_chase, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice);
  hipMalloc(&d_c, 256 * sizeof(long long)); hipMalloc(&d_s, 256 * kThreads * sizeof(double));
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  printf("{");
  for (int grid : {1, 256}) {
    printf("%s\"workgroups_%d\": {", grid == 1 ? "" : ", ", grid);
    for (int w = 0; w < 10; ++w) {
      double best = 1e30;
      for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(grid), dim3(kThreads), 150 * 1024, 0, w, d_c, d_s, d_chase, 1.5);
        hipDeviceSynchronize();
        std::vector<long long> c(grid);
        hipMemcpy(c.data(), d_c, grid * sizeof(long long), hipMemcpyDeviceToHost);
        double sum = 0; for (auto v : c) sum += v;
        const double per = sum / grid / kChain;
        if (per < best) best = per;
      }
      printf("%s\"%s\": %.2f", w ? ", " : "", names[w], best);
    }
    printf("}");
  }
  printf(", \"unit\": \"clock64 ticks per dependent step, one wave per SIMD\", \"chain\": %d}\n", kChain);
  return 0;
}
