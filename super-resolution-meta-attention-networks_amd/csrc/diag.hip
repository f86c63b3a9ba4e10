// Diagnostics: the practical fp32-MFMA ceiling of this chip under sustained load, and the in-kernel clock.
// Not on the product path; used by tools/kbench.py to put roofline fractions in context (DVFS: the chip
// lowers its clock under a dense MFMA stream, MI355X_MICROARCH.md "DVFS give-back").
#include "sisr_common.h"

// Each wave issues `iters` x 8 v_mfma_f32_32x32x2_f32 on two accumulators (the conv kernel's inner pattern)
// with operands in registers.  clk[0..1] of block 0: s_memtime / s_memrealtime deltas around the loop.
__global__ __launch_bounds__(256) void mfma_peak_kernel(int iters, float* __restrict__ out,
                                                        unsigned long long* __restrict__ clk) {
  f32x16 acc0 = {0}, acc1 = {0};
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
    }
    a += 1e-7f;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  out[(long)blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = t1 - t0;
    clk[1] = r1 - r0;
  }
}

// blocks of 256 threads (one wave per SIMD each); `blocks` = 256 * waves-per-SIMD fills the chip.
extern "C" int sisr_diag_mfma_peak(int blocks, int iters, float* out, unsigned long long* clk, void* stream) {
  if (blocks <= 0 || iters <= 0 || !out || !clk) return SISR_ERR_ARG;
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, out, clk);
  return sisr_check_launch();
}
