#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// straight-line body of BODY_KB kilobytes (8-byte SALU instructions), looped: cycles per pass when the body fits / does not fit the I-cache
template <int KB> __global__ void __launch_bounds__(256) body_kernel(int reps, long long* out) {
  long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    if constexpr (KB == 8)   asm volatile(".rept 1024\n s_mov_b32 s20, 0x12345678\n .endr" ::: "s20");
    if constexpr (KB == 16)  asm volatile(".rept 2048\n s_mov_b32 s20, 0x12345678\n .endr" ::: "s20");
    if constexpr (KB == 32)  asm volatile(".rept 4096\n s_mov_b32 s20, 0x12345678\n .endr" ::: "s20");
    if constexpr (KB == 48)  asm volatile(".rept 6144\n s_mov_b32 s20, 0x12345678\n .endr" ::: "s20");
    if constexpr (KB == 64)  asm volatile(".rept 8192\n s_mov_b32 s20, 0x12345678\n .endr" ::: "s20");
    if constexpr (KB == 96)  asm volatile(".rept 12288\n s_mov_b32 s20, 0x12345678\n .endr" ::: "s20");

  }
  long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
template <int KB> void run(int grid) {
  long long* d; hipMalloc(&d, grid * 8);
  const int reps = 20;
  for (int it = 0; it < 2; ++it) { hipLaunchKernelGGL(body_kernel<KB>, dim3(grid), dim3(256), 0, 0, reps, d); hipDeviceSynchronize(); }
  std::vector<long long> h(grid); hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += v;
  printf("body %3d KB, %3d workgroups of 4 waves: %8.0f cycles per pass = %.1f per 64-byte line (%.2f per instruction)\n", KB, grid, s / grid / reps, s / grid / reps / (KB * 16.0), s / grid / reps / (KB * 128.0));
  hipFree(d);
}
int main() {
  for (int grid : {1, 256}) { run<8>(grid); run<16>(grid); run<32>(grid); run<48>(grid); run<64>(grid); run<96>(grid); }
  return 0;
}
