// Diagnostics: the practical fp32-MFMA ceiling of this chip under sustained load, and the in-kernel clock.
// Not on the product path; used by tools/kbench.py to put roofline fractions in context (DVFS: the chip
// lowers its clock under a dense MFMA stream, MI355X_MICROARCH.md "DVFS give-back").
#include "sisr_common.h"

// Each wave issues `iters` x 8 v_mfma_f32_32x32x2_f32 on two accumulators (the conv kernel's inner pattern)
// with operands in registers.  clk[0..1] of block 0: s_memtime / s_memrealtime deltas around the loop.
// RAND: operands are per-lane pseudo-random values in [-1, 1), four different pairs per loop body (a chip that lowers its
// clock under load holds a lower clock on random operand bits than on near-constant ones: MI355X guide, DVFS give-back).
template <bool RAND>
__global__ __launch_bounds__(256) void mfma_peak_kernel(int iters, float* __restrict__ out,
                                                        unsigned long long* __restrict__ clk) {
  f32x16 acc0 = {0}, acc1 = {0};
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  float ra[4], rb[4];
  if (RAND) {
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h = h * 1664525u + 1013904223u;
      ra[e] = (float)(int)h * (1.0f / 2147483648.0f);
      h = h * 1664525u + 1013904223u;
      rb[e] = (float)(int)h * (1.0f / 2147483648.0f);
    }
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(RAND ? ra[e] : a, RAND ? rb[e] : b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(RAND ? rb[e] : b, RAND ? ra[3 - e] : a, acc1, 0, 0, 0);
    }
    if (!RAND) a += 1e-7f;  // RAND: the eight operand registers stay as they are -- no vector instruction but the MFMAs
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

// blocks of 256 threads (one wave per SIMD each); `blocks` = 256 * waves-per-SIMD fills the chip.  iters < 0: |iters|
// iterations on random operands.
extern "C" int sisr_diag_mfma_peak(int blocks, int iters, float* out, unsigned long long* clk, void* stream) {
  if (blocks <= 0 || iters == 0 || !out || !clk) return SISR_ERR_ARG;
  if (iters > 0)
    hipLaunchKernelGGL(mfma_peak_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, out, clk);
  else
    hipLaunchKernelGGL(mfma_peak_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, -iters, out, clk);
  return sisr_check_launch();
}

// What does one instruction of a given kind cost next to the fp32 MFMA stream?  Every wave issues iters x 8
// v_mfma_f32_32x32x2_f32 (two accumulators, as the conv kernel's K loop) with COUNT filler instructions of kind KIND
// spread over the eight MFMA gaps.  Time per iteration minus the COUNT = 0 time, divided by COUNT = what a filler adds.
//   1 v_add_f32   2 v_and_b32   3 s_add_u32   4 ds_read_b128   5 global_load_dwordx4 (L2-resident)   6 ds_write_b128
//   7 global_store_dword   8 v_mov_b32   9 s_nop 0   10 v_pk_add_f32   11 v_lshl_add_u64   12 buffer_load_dwordx4 (saddr form)
template <int KIND, int COUNT>
__global__ __launch_bounds__(256) void mfma_fill_kernel(int iters, float* __restrict__ out, const float* __restrict__ src,
                                                        unsigned long long* __restrict__ clk) {
  __shared__ __attribute__((aligned(16))) float lds[256 * 4 + 64];
  f32x16 acc0 = {0}, acc1 = {0};
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  float f0 = a, f1 = b;
  unsigned u0 = threadIdx.x, u1 = blockIdx.x;
  unsigned s0 = blockIdx.x;
  f32x4 q0 = {0.f, 0.f, 0.f, 0.f};
  unsigned long long w0 = threadIdx.x;
  const unsigned lofs = threadIdx.x * 16;
  const float* gp = src + threadIdx.x * 4;
  float* op = out + (long)blockIdx.x * 256 + threadIdx.x;
  lds[threadIdx.x] = a;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (e & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < (COUNT + 7 - e) / 8; ++k) {
        if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f0) : "v"(f1));
        if (KIND == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u0) : "v"(u1));
        if (KIND == 3) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) : : "scc");
        if (KIND == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(q0) : "v"(lofs));
        if (KIND == 5) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q0) : "v"(gp));
        if (KIND == 6) asm volatile("ds_write_b128 %0, %1" : : "v"(lofs), "v"(q0));
        if (KIND == 7) asm volatile("global_store_dword %0, %1, off" : : "v"(op), "v"(f0));
        if (KIND == 8) asm volatile("v_mov_b32 %0, %1" : "=v"(u0) : "v"(u1));
        if (KIND == 9) asm volatile("s_nop 0");
        if (KIND == 10) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(w0));
        if (KIND == 11) asm volatile("v_lshl_add_u64 %0, %0, 0, %0" : "+v"(w0));
        if (KIND == 12) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(q0) : "v"(lofs), "s"(src));
      }
    }
    // memory fillers: let a few iterations' worth stay in flight (issue cost, not latency, is what is measured)
    if ((KIND == 4 || KIND == 6) && (i & 3) == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((KIND == 5 || KIND == 7 || KIND == 12) && (i & 3) == 3 && COUNT <= 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((KIND == 5 || KIND == 7 || KIND == 12) && (i & 1) == 1 && COUNT == 16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((KIND == 5 || KIND == 7 || KIND == 12) && COUNT == 32) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = f0 + (float)u0 + (float)s0 + q0[0] + q0[3] + (float)(unsigned)w0;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  *op = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = t1 - t0;
    clk[1] = r1 - r0;
  }
}

extern "C" int sisr_diag_mfma_fill(int blocks, int iters, int kind, int count, float* out, const float* src,
                                   unsigned long long* clk, void* stream) {
  if (blocks <= 0 || iters <= 0 || !out || !src || !clk) return SISR_ERR_ARG;
#define FILL(K, C) \
  if (kind == K && count == C) { hipLaunchKernelGGL((mfma_fill_kernel<K, C>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, out, src, clk); return sisr_check_launch(); }
#define FILLK(K) FILL(K, 8) FILL(K, 16) FILL(K, 32)
  FILL(1, 0)
  FILLK(1) FILLK(2) FILLK(3) FILLK(4) FILLK(5) FILLK(6) FILLK(7) FILLK(8) FILLK(9) FILLK(10) FILLK(11) FILLK(12)
#undef FILLK
#undef FILL
  return SISR_ERR_UNSUPPORTED;
}
