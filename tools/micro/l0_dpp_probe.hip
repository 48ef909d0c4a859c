// The DPP statements of level 0 of the hybrid solve (csrc/teb_kernel.hpp: TEB_L0_O1_DPP, TEB_L0_O2_DPP, TEB_L0_O3_DPP) against the same sums
// through __shfl, bit for bit: 256 lanes = 32 groups of 8, structural zeros as in the kernel. They sit outside the compiler's hazard
// recogniser and depend on an instruction ORDER (tools/micro/dpp64_mask_probe.hip): this is the assertion on that.
// Build: teb_local_planner_amd/build.py: build_micro (hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I teb_local_planner_amd/csrc ..)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "teb_kernel.hpp"
__global__ void probe(const double* in, double* out) {
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 7, g0 = lane & ~7;
  double cl[8], cu[8], wL[8], wU[8], o1[8], o2[8], o3[8], r1[8], r2[8], r3[8];
  for (int k = 0; k < 8; ++k) {
    cl[k] = (c >= k - 2) ? in[tid * 32 + k] : 0.0;
    cu[k] = (k >= c - 2) ? in[tid * 32 + 8 + k] : 0.0;
    wL[k] = in[tid * 32 + 16 + k]; wU[k] = in[tid * 32 + 24 + k];
    o1[k] = o2[k] = o3[k] = r1[k] = r2[k] = r3[k] = 0;
  }
  for (int aa = 0; aa < 8; ++aa)
    for (int k = 0; k < 8; ++k) {
      if (aa >= k - 2) r1[aa] = __builtin_fma(__shfl(cl[k], g0 + aa), wL[k], r1[aa]);
      if (k >= aa - 2) { r2[aa] = __builtin_fma(-__shfl(cu[k], g0 + aa), wL[k], r2[aa]); r3[aa] = __builtin_fma(__shfl(cu[k], g0 + aa), wU[k], r3[aa]); }
    }
  asm("s_nop 1\n\t" TEB_L0_O1_DPP
      : "+v"(o1[0]), "+v"(o1[1]), "+v"(o1[2]), "+v"(o1[3]), "+v"(o1[4]), "+v"(o1[5]), "+v"(o1[6]), "+v"(o1[7])
      : "v"(cl[0]), "v"(cl[1]), "v"(cl[2]), "v"(cl[3]), "v"(cl[4]), "v"(cl[5]), "v"(cl[6]), "v"(cl[7]),
        "v"(wL[0]), "v"(wL[1]), "v"(wL[2]), "v"(wL[3]), "v"(wL[4]), "v"(wL[5]), "v"(wL[6]), "v"(wL[7]));
  asm("s_nop 1\n\t" TEB_L0_O2_DPP
      : "+v"(o2[0]), "+v"(o2[1]), "+v"(o2[2]), "+v"(o2[3]), "+v"(o2[4]), "+v"(o2[5]), "+v"(o2[6]), "+v"(o2[7])
      : "v"(cu[0]), "v"(cu[1]), "v"(cu[2]), "v"(cu[3]), "v"(cu[4]), "v"(cu[5]), "v"(cu[6]), "v"(cu[7]),
        "v"(wL[0]), "v"(wL[1]), "v"(wL[2]), "v"(wL[3]), "v"(wL[4]), "v"(wL[5]), "v"(wL[6]), "v"(wL[7]));
  asm("s_nop 1\n\t" TEB_L0_O3_DPP
      : "+v"(o3[0]), "+v"(o3[1]), "+v"(o3[2]), "+v"(o3[3]), "+v"(o3[4]), "+v"(o3[5]), "+v"(o3[6]), "+v"(o3[7])
      : "v"(cu[0]), "v"(cu[1]), "v"(cu[2]), "v"(cu[3]), "v"(cu[4]), "v"(cu[5]), "v"(cu[6]), "v"(cu[7]),
        "v"(wU[0]), "v"(wU[1]), "v"(wU[2]), "v"(wU[3]), "v"(wU[4]), "v"(wU[5]), "v"(wU[6]), "v"(wU[7]));
  for (int k = 0; k < 8; ++k) {
    out[tid * 48 + k] = o1[k]; out[tid * 48 + 8 + k] = o2[k]; out[tid * 48 + 16 + k] = o3[k];
    out[tid * 48 + 24 + k] = r1[k]; out[tid * 48 + 32 + k] = r2[k]; out[tid * 48 + 40 + k] = r3[k];
  }
}
int main() {
  static double h[256 * 32], r[256 * 48];
  unsigned s = 12345; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (double)(s >> 8) / (1 << 24) - 0.5; }
  double *di, *dout; (void)hipMalloc(&di, sizeof(h)); (void)hipMalloc(&dout, sizeof(r));
  (void)hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  probe<<<1, 256>>>(di, dout);
  (void)hipMemcpy(r, dout, sizeof(r), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 256; ++l) for (int k = 0; k < 24; ++k) if (r[l * 48 + k] != r[l * 48 + 24 + k]) { if (bad < 12) printf("lane %d value %d dpp %.17g ref %.17g\n", l, k, r[l * 48 + k], r[l * 48 + 24 + k]); ++bad; }
  printf("level-0 DPP products: %d of %d entries differ\n", bad, 256 * 24);
  return bad ? 1 : 0;
}
