// Micro-benchmark of the damped solve on LDS-resident 8x8 blocks: the plain block cyclic reduction (cr_forward + cr_top + cr_backward,
// rounds 1 - 3) against the partitioned solve of round 4 (part_solve_blocks: interior rows by sweeps in registers, cyclic reduction on
// the interface rows only). Same SPD block-tridiagonal system, solutions compared, cycles per solve with one workgroup alone and with
// one per CU. Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I teb_local_planner_amd/csrc tools/micro/part_solve_bench.hip -o tools/micro/part_solve_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "teb_kernel.hpp"
namespace tebamd {
// ---- partitioned solve: interior rows by sweeps in registers, cyclic reduction on the interface rows only (round 4) ----------------------
// The block-tridiagonal system is cut into P <= 32 partitions of q = ceil(Nb / 32) consecutive block rows, aligned to the END of the
// band (partition 0 holds the remainder, r0 = Nb - (P - 1) q rows). The LAST row of a partition is its interface row; the q - 1 rows
// before it are interior. One 8-lane group per partition eliminates its interior rows left to right IN REGISTERS - no barrier and no
// read-modify-write of shared memory between the steps: the running diagonal block D', right-hand side f' and the fill-in G that couples
// the row under elimination to the interface row a of the partition to the left (lane c holds column c) travel in registers; per row
//     W2 = P G,  W1 = P L_{k+1}^T,  wf = P f'         (P = D'^-1, LDL^T repeated by the 8 lanes as in the cyclic reduction)
//     D_a -= G^T W2,  f_a -= G^T wf                   (accumulated per lane: column c)
//     D'_{k+1} = D_{k+1} - L_{k+1} W1,  f'_{k+1} = f_{k+1} - L_{k+1} wf,  G_{k+1} = - L_{k+1} W2
// and what is left after the last interior row is folded into the partition's own interface row b (D_b, f_b, its coupling L_b to a) and,
// one barrier later, into a (D_a, f_a). The interface rows - every q-th block row, <= 32 of them - form a block-tridiagonal system of
// their own, reduced by the cyclic reduction where they lie (row stride q: 5 levels instead of 8 for 200 poses), then the interior rows
// are back-substituted, again per partition without barriers: x_k = wf_k - W2_k x_a - W1_k x_{k+1}. The records (W2, W1, wf) of an
// eliminated row take its own slots (D_k, L_k, f_k), as in the cyclic reduction. Sequential depth: q - 1 sweep steps (3 at 200 poses, 4 at
// 288) + log2(P) rounds, every sweep step with all partitions busy - against 8 - 9 rounds before, the fine ones through two barriers
// and shared-memory read-modify-writes each. Cross-lane operands (columns of G held by the neighbours) travel by ds_swizzle.
constexpr int kPartGroups = kThreads / 8;
struct PartGeom { int q, P, r0; };
__host__ __device__ inline PartGeom part_geom(int Nb) {
  PartGeom g;
  g.q = (Nb + kPartGroups - 1) / kPartGroups;
  if (g.q < 1) g.q = 1;
  g.P = (Nb + g.q - 1) / g.q;
  g.r0 = Nb - (g.P - 1) * g.q;
  return g;
}
// One interior row. In: F.a = lower triangle of D' (unfactored), G = column c of the fill-in, X = row c of L_{k+1}, fcur = f'.
// Cross-lane operands travel through LDS in 16-byte accesses (a ds_swizzle moves 4 bytes per lane and instruction and holds the LDS pipe
// as long; measured: 13 k cycles per row with swizzles): every lane writes its column of G as a ROW of the scratch block GT (the slot of
// the row under elimination, free since its D / L were loaded), so that "column r of G" is a contiguous 64-byte read for all.
// Ln = the 8 x 8 block L_{k+1} (rows contiguous).
// Out: the records wG (column c of P G), wL (column c of P L_{k+1}^T), wf (P f', all 8); S = column c of L_{k+1} P L_{k+1}^T,
// Gn = column c of - L_{k+1} P G, t = element c of L_{k+1} P f'; Da += column c of G^T P G, fa += element c of G^T P f'.
__device__ __forceinline__ bool part_step(Ldl8& F, const double* G, const double* X, const double* fcur, const double* __restrict__ Ln,
                                          double* __restrict__ GT, int c, bool hasA,
                                          double* wG, double* wL, double* wf, double* S, double* Gn, double& t, double* Da, double& fa) {
  TEB_SOLVER_FMA
  if (hasA) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) *reinterpret_cast<teb_v2d*>(GT + c * 8 + j) = teb_v2d{G[j], G[j + 1]};
  }
  const bool ok = F.factor();
#pragma unroll
  for (int k = 0; k < 8; ++k) { wG[k] = G[k]; wL[k] = X[k]; wf[k] = fcur[k]; }
  F.solve3(wG, wL, wf);
  double t_ = 0, fa_ = fa;
#pragma unroll
  for (int k = 0; k < 8; ++k) { t_ += X[k] * wf[k]; fa_ += G[k] * wf[k]; }
  t = t_; fa = fa_;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    double lr[8];
    ld_row<8>(Ln + r * 8, lr);
    double s_ = 0, g_ = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { s_ += lr[j] * wL[j]; g_ -= lr[j] * wG[j]; }
    S[r] = s_; Gn[r] = g_;
    if ((r % TEB_CR_FENCE_EVERY) == TEB_CR_FENCE_EVERY - 1) { TEB_CR_SCHED_BARRIER }
  }
  if (hasA) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      double gr[8];
      ld_row<8>(GT + r * 8, gr);   // column r of G
      double d_ = Da[r];
#pragma unroll
      for (int j = 0; j < 8; ++j) d_ += gr[j] * wG[j];
      Da[r] = d_;
      if ((r % TEB_CR_FENCE_EVERY) == TEB_CR_FENCE_EVERY - 1) { TEB_CR_SCHED_BARRIER }
    }
  }
  return ok;
}

// Blocks in LDS (lambda already on the diagonals, f = right-hand side): forward sweeps + interface reduction + top + back substitution.
// The solution replaces f. Returns false iff some pivot was <= 0 (seen by every lane of the group that met it).
__device__ __forceinline__ bool part_solve_blocks(double* D, double* L, double* f, int Nb) {
  TEB_SOLVER_FMA
  const int tid = threadIdx.x, p = tid >> 3, c = tid & 7;
  const PartGeom g = part_geom(Nb);
  const bool actg = p < g.P;
  const int b = g.r0 - 1 + p * g.q;              // interface row of this group's partition
  const int kfirst = (b - (g.q - 1) > 0) ? b - (g.q - 1) : 0;   // its first interior row (partition 0 may have fewer than q - 1)
  const bool sweeps = actg && kfirst < b;
  const bool hasA = p > 0;
  bool ok = true;
  double Da[8] = {0, 0, 0, 0, 0, 0, 0, 0}, fa = 0, S[8], Gn[8], t = 0;
  if (sweeps) {
    Ldl8 F;
    double G[8], fcur[8];
    F.load(D + kfirst * kBlk);
    ld_row<8>(f + kfirst * 8, fcur);
#pragma unroll
    for (int j = 0; j < 8; ++j) G[j] = hasA ? L[kfirst * kBlk + j * 8 + c] : 0.0;
    for (int k = kfirst; k < b; ++k) {
      double X[8], wG[8], wL[8], wf[8];
      const double* Ln = L + (k + 1) * kBlk;
      double* Dk = D + k * kBlk;   // free: D_k is in F, L_k was consumed by the previous step (or as the initial G)
      double* Lk = L + k * kBlk;
      ld_row<8>(Ln + c * 8, X);
      ok = part_step(F, G, X, fcur, Ln, Dk, c, hasA, wG, wL, wf, S, Gn, t, Da, fa) && ok;
      if (k + 1 < b) {
        // D'_{k+1}, f'_{k+1} for every lane of the group: the columns of S (and t) pass through the free slots as rows
#pragma unroll
        for (int j = 0; j < 8; j += 2) *reinterpret_cast<teb_v2d*>(Lk + c * 8 + j) = teb_v2d{S[j], S[j + 1]};
        f[k * 8 + c] = t;
        F.load(D + (k + 1) * kBlk);
        ld_row<8>(f + (k + 1) * 8, fcur);
        {
          double tt[8];
          ld_row<8>(f + k * 8, tt);
#pragma unroll
          for (int r = 0; r < 8; ++r) fcur[r] -= tt[r];
        }
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {   // entry (r, cc), r >= cc, of L P L^T = element r of lane cc's S = Lk[cc * 8 + r]
          double sc_[8];
          if (cc < 7) {
            ld_row<8>(Lk + cc * 8, sc_);
#pragma unroll
            for (int r = cc; r < 8; ++r) F.a[Ldl8::idx(r, cc)] -= sc_[r];
          } else F.a[Ldl8::idx(7, 7)] -= Lk[7 * 8 + 7];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) G[j] = Gn[j];
      }
      // records of row k in its own slots: D_k <- P G (coupling to a), L_k <- P L_{k+1}^T (coupling to k + 1), f_k <- P f'
#pragma unroll
      for (int j = 0; j < 8; ++j) { Dk[j * 8 + c] = wG[j]; Lk[j * 8 + c] = wL[j]; }
      f[k * 8 + c] = wf[c];
    }
    // what is left goes into the partition's own interface row b ...
    double* Db = D + b * kBlk;
#pragma unroll
    for (int r = 0; r < 8; ++r) Db[r * 8 + c] -= S[r];
    if (hasA) {
      double* Lb = L + b * kBlk;
#pragma unroll
      for (int r = 0; r < 8; ++r) Lb[r * 8 + c] = Gn[r];
    }
    f[b * 8 + c] -= t;
  }
  __syncthreads();
  if (sweeps && hasA) {   // ... and into the interface row a of the partition to the left (which has received its own part above)
    const int a = b - g.q;
    double* Dm = D + a * kBlk;
#pragma unroll
    for (int r = 0; r < 8; ++r) Dm[r * 8 + c] -= Da[r];
    f[a * 8 + c] -= fa;
  }
  __syncthreads();
  // the interface rows r0 - 1 + j q, j < P, where they lie
  double* Dq = D + (g.r0 - 1) * kBlk;
  double* Lq = L + (g.r0 - 1) * kBlk;
  double* fq = f + (g.r0 - 1) * 8;
  const int bD = g.q * kBlk, bF = g.q * 8;
  ok = cr_forward(Dq, Lq, fq, g.P, 1, g.P, bD, bF) && ok;
  ok = cr_top(Dq, fq) && ok;
  __syncthreads();
  int stop = 1;
  while (stop * 2 < g.P) stop *= 2;
  if (g.P > 1) cr_backward(Dq, Lq, fq, g.P, stop, 1, bD, bF);
  // interior rows, last to first: lane r of the group computes component r
  if (sweeps) {
    const int a = b - g.q;   // (p > 0)
    double xa[8], xn[8];
    if (hasA) ld_row<8>(f + a * 8, xa);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) xa[j] = 0.0;
    }
    for (int k = b - 1; k >= kfirst; --k) {
      double W2[8], W1[8];
      ld_row<8>(f + (k + 1) * 8, xn);
      ld_row<8>(D + k * kBlk + c * 8, W2);
      ld_row<8>(L + k * kBlk + c * 8, W1);
      double acc = f[k * 8 + c], acc2 = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) { acc -= W2[j] * xa[j]; acc2 -= W1[j] * xn[j]; }
      f[k * 8 + c] = acc + acc2;
    }
  }
  __syncthreads();
  return ok;
}

}  // namespace tebamd
using namespace tebamd;

template <int MODE>
__global__ void __launch_bounds__(kThreads) solve_kernel(int Nb, const double* gD, const double* gL, const double* gf, double* xout, long long* cycles, int reps) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* D = lds;
  double* L = D + Nb * kBlk;
  double* f = L + Nb * kBlk;
  long long acc = 0;
  bool ok = true;
  for (int r = 0; r < reps; ++r) {
    for (int q = threadIdx.x; q < Nb * kBlk; q += kThreads) { D[q] = gD[q]; L[q] = gL[q]; }
    for (int q = threadIdx.x; q < Nb * 8; q += kThreads) f[q] = gf[q];
    __syncthreads();
    const long long t0 = clock64();
    if (MODE == 0) {
      ok = cr_forward(D, L, f, Nb, 1, Nb) && ok;
      ok = cr_top(D, f) && ok;
      __syncthreads();
      int stop = 1;
      while (stop * 2 < Nb) stop *= 2;
      if (Nb > 1) cr_backward(D, L, f, Nb, stop, 1);
    } else {
      ok = part_solve_blocks(D, L, f, Nb) && ok;
    }
    __syncthreads();
    acc += clock64() - t0;
  }
  if (threadIdx.x == 0) cycles[blockIdx.x] = acc;
  if (blockIdx.x == 0) for (int q = threadIdx.x; q < Nb * 8; q += kThreads) xout[q] = ok ? f[q] : NAN;
}

int main() {
  hipFuncSetAttribute((const void*)solve_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)solve_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int n : {40, 64, 100, 130, 150, 200, 238}) {
    const int Nb = (4 * n + 7) / 8;
    std::mt19937 gen(1234 + n);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    std::vector<double> hD(Nb * kBlk, 0.0), hL(Nb * kBlk, 0.0), hf(Nb * 8);
    for (int j = 0; j < Nb; ++j) {
      for (int a = 0; a < 8; ++a)
        for (int b = 0; b <= a; ++b) {
          const double v = (a == b) ? 30.0 + 5.0 * u(gen) : u(gen);
          hD[j * kBlk + a * 8 + b] = v; hD[j * kBlk + b * 8 + a] = v;
        }
      for (int a = 0; a < 8; ++a)
        for (int b = 0; b < 8; ++b) hL[j * kBlk + a * 8 + b] = (j > 0 && b >= a - 2) ? 2.0 * u(gen) : 0.0;   // the band structure of the real blocks
      for (int a = 0; a < 8; ++a) hf[j * 8 + a] = u(gen);
    }
    double *dD, *dL, *df, *dx; long long* dc;
    hipMalloc(&dD, hD.size() * 8); hipMalloc(&dL, hL.size() * 8); hipMalloc(&df, hf.size() * 8); hipMalloc(&dx, hf.size() * 8); hipMalloc(&dc, 256 * 8);
    hipMemcpy(dD, hD.data(), hD.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dL, hL.data(), hL.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(df, hf.data(), hf.size() * 8, hipMemcpyHostToDevice);
    const size_t lds = (size_t)(2 * Nb * kBlk + Nb * 8) * sizeof(double);
    const int reps = 20;
    std::vector<double> x0(Nb * 8), x1(Nb * 8);
    double cyc[2][2];
    for (int gi = 0; gi < 2; ++gi) {
      const int grid = gi == 0 ? 1 : 256;
      for (int mode = 0; mode < 2; ++mode) {
        for (int it = 0; it < 2; ++it) {
          if (mode == 0) hipLaunchKernelGGL(solve_kernel<0>, dim3(grid), dim3(kThreads), lds, 0, Nb, dD, dL, df, dx, dc, reps);
          else hipLaunchKernelGGL(solve_kernel<1>, dim3(grid), dim3(kThreads), lds, 0, Nb, dD, dL, df, dx, dc, reps);
          hipDeviceSynchronize();
        }
        std::vector<long long> c(grid);
        hipMemcpy(c.data(), dc, grid * 8, hipMemcpyDeviceToHost);
        double sum = 0; for (auto v : c) sum += v;
        cyc[gi][mode] = sum / grid / reps;
        hipMemcpy((mode == 0 ? x0 : x1).data(), dx, Nb * 8 * 8, hipMemcpyDeviceToHost);
      }
    }
    // residual of both against the original system (host, fp64) and their mutual difference
    auto resid = [&](const std::vector<double>& x) {
      double worst = 0;
      for (int j = 0; j < Nb; ++j)
        for (int a = 0; a < 8; ++a) {
          double s = -hf[j * 8 + a];
          for (int b = 0; b < 8; ++b) {
            s += hD[j * kBlk + a * 8 + b] * x[j * 8 + b];
            if (j > 0) s += hL[j * kBlk + a * 8 + b] * x[(j - 1) * 8 + b];
            if (j + 1 < Nb) s += hL[(j + 1) * kBlk + b * 8 + a] * x[(j + 1) * 8 + b];
          }
          worst = std::fmax(worst, std::fabs(s));
        }
      return worst;
    };
    double diff = 0; for (int q = 0; q < Nb * 8; ++q) diff = std::fmax(diff, std::fabs(x0[q] - x1[q]));
    const PartGeom g = part_geom(Nb);
    printf("n = %3d poses, Nb = %3d block rows (q = %d, P = %d, r0 = %d): plain CR %7.0f / %7.0f cycles (alone / 256 workgroups), partitioned %7.0f / %7.0f  -> %.2fx / %.2fx;"
           " residual %.1e / %.1e, |x_cr - x_part| %.1e\n", n, Nb, g.q, g.P, g.r0, cyc[0][0], cyc[1][0], cyc[0][1], cyc[1][1], cyc[0][0] / cyc[0][1], cyc[1][0] / cyc[1][1],
           resid(x0), resid(x1), diff);
    hipFree(dD); hipFree(dL); hipFree(df); hipFree(dx); hipFree(dc);
  }
  return 0;
}
