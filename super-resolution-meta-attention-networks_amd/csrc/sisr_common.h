// Shared device/host definitions for the gfx950 SISR kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A 64-channel-chunked view of an activation in HBM.  Element (b, h, w, chunk q, c) lives at
//   p + b*sB + h*sH + w*sW + (q / cdiv)*chi + (q % cdiv)*clo + c          (all strides in floats)
// Plain NHWC with C = 64*k channels: sW = C, cdiv = huge, clo = 64.
// PixelShuffle(r) output seen as the conv's (B,H,W,64*r*r) result: physical tensor is
// [B][H*r][W*r][64]; sH = r*(W*r*64), sW = r*64, cdiv = r, chi = W*r*64, clo = 64, and chunk
// q = i*r + j carries original channels {c*r*r + q}.  The shuffle is therefore free: it is only an
// address map (ref: advanced/common.py:28-31 conv -> nn.PixelShuffle).
struct View {
  long sB, sH, sW;
  long chi, clo;
  int cdiv;
  __host__ __device__ long chunk(int q) const { return (long)(q / cdiv) * chi + (long)(q % cdiv) * clo; }
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// Zero a float4 with a lane mask by bitwise AND: exact zeros even if the (clamped-address) load
// returned Inf/NaN, and -- unlike `ok ? t : 0` -- hipcc cannot turn it back into a branch around the
// load (which would serialise the staging loads behind s_waitcnt vmcnt(0)).
__device__ __forceinline__ f32x4 sisr_keep_if(f32x4 t, bool ok) {
  const unsigned m = ok ? 0xffffffffu : 0u;
  u32x4 b = __builtin_bit_cast(u32x4, t);
  b &= (u32x4){m, m, m, m};
  return __builtin_bit_cast(f32x4, b);
}

// The gated skip t * g + x, with the product rounded BEFORE the sum (no fma contraction) -- the arithmetic of the
// reference's `x * y` followed by `res += x` on fp32 tensors.  It is formed by the stand-alone gate kernel and by the
// GATE prologues of the conv kernels, and all of them must agree to the bit, with each other and with the reference:
// the next block's ReLU mask is taken from this map, an element of it that differs by one ulp flips a mask bit now
// and then, and one flipped bit moves that block's weight gradient by ~3e-3 (found with a float64 run of the oracle).
__device__ __forceinline__ f32x4 sisr_mul_add4(f32x4 a, f32x4 b, f32x4 c) {
#pragma clang fp contract(off)
  const f32x4 p = a * b;
  return p + c;
}

// Buffer addressing.  Beside the fp32 MFMA stream a vector-memory instruction with a 64-bit per-lane address
// (`global_load_dwordx4 v, v[a:a+1], off`) costs its SIMD 30-40 cycles, and every VALU instruction that builds such an
// address ~4.5 more; the same access as SGPR resource + 32-bit lane offset + SGPR offset (`buffer_load_dwordx4 v, voff,
// s[rsrc], soff offen`) is free at the K loop's density, like SALU work (measured: tools/mfma_fill.py).  So the hot
// kernels address every operand as {wave-uniform base -> resource, lane-constant byte offset, scalar byte offset}.
// Raw resource (stride 0), bounds check off (num_records = 2^32 - 1): the offsets are in range by construction.
typedef __amdgpu_buffer_rsrc_t sisr_rsrc_t;
__device__ __forceinline__ sisr_rsrc_t sisr_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0xffffffffu, 0x00020000);
}
__device__ __forceinline__ f32x4 sisr_buf_load4(sisr_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff_bytes, (int)soff_bytes, 0));
}
__device__ __forceinline__ float sisr_buf_load1(sisr_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff_bytes, (int)soff_bytes, 0));
}
__device__ __forceinline__ void sisr_buf_store4(f32x4 v, sisr_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff_bytes, (int)soff_bytes, 0);
}
__device__ __forceinline__ void sisr_buf_store1(float v, sisr_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff_bytes, (int)soff_bytes, 0);
}
// write-through forms (cache policy sc1): the bytes leave the XCD's L2 as they are written instead of in the write-back burst
// at the end of the kernel (MI355X guide, "boundary": + B / 6 TB/s behind B dirty bytes)
__device__ __forceinline__ void sisr_buf_store1_wt(float v, sisr_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff_bytes, (int)soff_bytes, 16);
}
__device__ __forceinline__ void sisr_buf_store4_wt(f32x4 v, sisr_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff_bytes, (int)soff_bytes, 16);
}

#define SISR_OK 0
#define SISR_ERR_ARG (-1)
#define SISR_ERR_ALIGN (-2)
#define SISR_ERR_LAUNCH (-3)
#define SISR_ERR_UNSUPPORTED (-4)

static inline int sisr_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SISR_OK : SISR_ERR_LAUNCH - (int)e * 16;
}
// Dynamic LDS above 64 KB needs an explicit opt-in on the kernel (gfx950 has 160 KB per CU).  Idempotent,
// so the unsynchronised once-flag is safe.
#define SISR_ALLOW_LDS(kernel, bytes)                                                                       \
  do {                                                                                                      \
    static bool done_ = false;                                                                              \
    if (!done_) {                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),                                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes));                  \
      done_ = true;                                                                                         \
    }                                                                                                       \
  } while (0)

static inline bool sisr_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
