// SFTMD pieces that are not 3x3 convs over 64-channel chunks.
// ref: Code/SISR/models/SFTMD_variants/architectures.py:25-56 (StandardSft: x * sigmoid(mul) + add),
//      :110-176 (SFTMD: LeakyReLU(0.2) head / upscale, 9x9 64 -> 3 output conv, clamp to [0, 1]).
// The 3x3 convs of the network run on the MFMA kernels of conv3x3_mfma.hip (LeakyReLU as epilogue / mask slope); here:
//   sft_compose     the four conv weights / biases of an SFT layer placed into the merged / block-diagonal tensors of its
//                   two MFMA convs, and the reverse split of their gradients (one launch each)
//   sft_combine     out = [relu](x * sigmoid(y2[:64]) + y2[64:]) on pixel-strided maps, + copy of the metadata chunk;
//                   backward -> dx and d y2 (ReLU mask recomputed)
//   copy_chunk / add2 / leaky  strided 64-channel helpers (metadata chunk fill, fea_mid + fea_bef, LeakyReLU after the
//                   3-channel head conv and its backward)
//   conv9_*         the 9x9 64 -> 3 output conv: forward, input gradient (+ LeakyReLU mask of the map it feeds back into),
//                   weight / bias gradient (ordered two-stage sum), and the clamp's forward / backward
// All HBM- / VALU-bound and small next to the network's 3x3 convs; written for clarity, coalesced 256-B rows.
#include "sisr_common.h"
#include "conv_rgb_out.h"

__device__ __forceinline__ float sft_sigmoid(float z) { return 1.f / (1.f + expf(-z)); }

static unsigned sft_blocks(long n) {
  long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 65535 ? 65535 : b));
}

// ------------------------------------------------------------------ weight composition
// One launch per SFT layer and direction.  split == 0: the four conv weights / biases of the layer -> the merged
//   WA [64][128][9]  rows 0..31 = mul_conv1 ([32][64 + M][9]), rows 32..63 = add_conv1, input channels >= 64 + M zero
//   bA [64]          (mul_conv1.bias | add_conv1.bias)
//   WB [128][64][9]  rows 0..63 = mul_conv2 ([64][32][9]) on inputs 0..31, rows 64..127 = add_conv2 on inputs 32..63
//   bB [128]         (mul_conv2.bias | add_conv2.bias)
// split == 1: the reverse copy (gradients of the merged tensors -> the eight parameter gradients).
struct SftCompose {
  float *mw1, *mb1, *aw1, *ab1, *mw2, *mb2, *aw2, *ab2, *WA, *bA, *WB, *bB;
  int M, split;
};
__global__ __launch_bounds__(256) void sft_compose_kernel(SftCompose c) {
  constexpr int NA = 64 * 128 * 9, NB = 128 * 64 * 9;
  const int cin = 64 + c.M;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < NA + NB + 64 + 128; i += gridDim.x * 256) {
    float *merged, *part = nullptr;
    if (i < NA) {
      const int t = i % 9, r = i / 9, o = r >> 7, ci = r & 127;
      merged = c.WA + i;
      if (ci < cin) part = (o < 32 ? c.mw1 : c.aw1) + ((long)(o & 31) * cin + ci) * 9 + t;
    } else if (i < NA + NB) {
      const int k = i - NA, t = k % 9, r = k / 9, o = r >> 6, ci = r & 63;
      merged = c.WB + k;
      if ((o < 64) == (ci < 32)) part = (o < 64 ? c.mw2 : c.aw2) + ((long)(o & 63) * 32 + (ci & 31)) * 9 + t;
    } else if (i < NA + NB + 64) {
      const int o = i - NA - NB;
      merged = c.bA + o;
      part = (o < 32 ? c.mb1 : c.ab1) + (o & 31);
    } else {
      const int o = i - NA - NB - 64;
      merged = c.bB + o;
      part = (o < 64 ? c.mb2 : c.ab2) + (o & 63);
    }
    if (!c.split) *merged = part ? *part : 0.f;
    else if (part) *part = *merged;
  }
}

// every SFT layer of a network in one launch (once per training step, before the one-launch weight packing): table =
// n SftCompose records in device memory, blockIdx.y = layer
__global__ __launch_bounds__(256) void sft_compose_many_kernel(const SftCompose* __restrict__ table) {
  const SftCompose c = table[blockIdx.y];
  constexpr int NA = 64 * 128 * 9, NB = 128 * 64 * 9;
  const int cin = 64 + c.M;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < NA + NB + 64 + 128; i += gridDim.x * 256) {
    float *merged, *part = nullptr;
    if (i < NA) {
      const int t = i % 9, r = i / 9, o = r >> 7, ci = r & 127;
      merged = c.WA + i;
      if (ci < cin) part = (o < 32 ? c.mw1 : c.aw1) + ((long)(o & 31) * cin + ci) * 9 + t;
    } else if (i < NA + NB) {
      const int k = i - NA, t = k % 9, r = k / 9, o = r >> 6, ci = r & 63;
      merged = c.WB + k;
      if ((o < 64) == (ci < 32)) part = (o < 64 ? c.mw2 : c.aw2) + ((long)(o & 63) * 32 + (ci & 31)) * 9 + t;
    } else if (i < NA + NB + 64) {
      const int o = i - NA - NB;
      merged = c.bA + o;
      part = (o < 32 ? c.mb1 : c.ab1) + (o & 31);
    } else {
      const int o = i - NA - NB - 64;
      merged = c.bB + o;
      part = (o < 64 ? c.mb2 : c.ab2) + (o & 63);
    }
    *merged = part ? *part : 0.f;
  }
}

extern "C" size_t sisr_sft_compose_record_bytes(void) { return sizeof(SftCompose); }

// table: n records {mul_w1, mul_b1, add_w1, add_b1, mul_w2, mul_b2, add_w2, add_b2, WA, bA, WB, bB (device pointers), int M,
// int split (ignored: always composes)} in device memory
extern "C" int sisr_sft_compose_many(const void* table, int n, void* stream) {
  if (!table || n <= 0 || n > 65535) return SISR_ERR_ARG;
  hipLaunchKernelGGL(sft_compose_many_kernel, dim3(72, n), dim3(256), 0, (hipStream_t)stream, static_cast<const SftCompose*>(table));
  return sisr_check_launch();
}

extern "C" int sisr_sft_compose(float* mw1, float* mb1, float* aw1, float* ab1, float* mw2, float* mb2, float* aw2, float* ab2,
                                float* WA, float* bA, float* WB, float* bB, int M, int split, void* stream) {
  if (!mw1 || !mb1 || !aw1 || !ab1 || !mw2 || !mb2 || !aw2 || !ab2 || !WA || !bA || !WB || !bB || M < 0 || M > 64)
    return SISR_ERR_ARG;
  const SftCompose c = {mw1, mb1, aw1, ab1, mw2, mb2, aw2, ab2, WA, bA, WB, bB, M, split};
  hipLaunchKernelGGL(sft_compose_kernel, dim3(288), dim3(256), 0, (hipStream_t)stream, c);
  return sisr_check_launch();
}

// ------------------------------------------------------------------ SFT combine
// x: 64 features per pixel at pixel stride xs (floats); y2: [npix][128] (mul pre-activation | add); out at stride os.
// md (nullable): [npix][64] copied into out's second chunk (os must be 128 then).  One float4 per thread.
__global__ __launch_bounds__(256) void sft_combine_fwd_kernel(const float* __restrict__ x, long xs, const float* __restrict__ y2,
                                                              const float* __restrict__ md, float* __restrict__ out, long os,
                                                              long npix, int relu) {
  const long total = npix * 16;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long p = i >> 4;
    const int c4 = (int)(i & 15) * 4;
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + p * xs + c4);
    const f32x4 mv = *reinterpret_cast<const f32x4*>(y2 + p * 128 + c4);
    const f32x4 av = *reinterpret_cast<const f32x4*>(y2 + p * 128 + 64 + c4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = xv[e] * sft_sigmoid(mv[e]) + av[e];
      o[e] = relu ? fmaxf(v, 0.f) : v;
    }
    *reinterpret_cast<f32x4*>(out + p * os + c4) = o;
    if (md) *reinterpret_cast<f32x4*>(out + p * os + 64 + c4) = *reinterpret_cast<const f32x4*>(md + p * 64 + c4);
  }
}

// dout at stride ds -> dx [npix][64] (plain) and dy2 [npix][128]
__global__ __launch_bounds__(256) void sft_combine_bwd_kernel(const float* __restrict__ dout, long ds,
                                                              const float* __restrict__ x, long xs,
                                                              const float* __restrict__ y2, float* __restrict__ dx,
                                                              float* __restrict__ dy2, long npix, int relu) {
  const long total = npix * 16;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long p = i >> 4;
    const int c4 = (int)(i & 15) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(dout + p * ds + c4);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + p * xs + c4);
    const f32x4 mv = *reinterpret_cast<const f32x4*>(y2 + p * 128 + c4);
    const f32x4 av = *reinterpret_cast<const f32x4*>(y2 + p * 128 + 64 + c4);
    f32x4 gx, gm, ga;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float s = sft_sigmoid(mv[e]);
      const float pre = xv[e] * s + av[e];
      const float ge = (relu && !(pre > 0.f)) ? 0.f : g[e];
      gx[e] = ge * s;
      gm[e] = ge * xv[e] * s * (1.f - s);
      ga[e] = ge;
    }
    *reinterpret_cast<f32x4*>(dx + p * 64 + c4) = gx;
    *reinterpret_cast<f32x4*>(dy2 + p * 128 + c4) = gm;
    *reinterpret_cast<f32x4*>(dy2 + p * 128 + 64 + c4) = ga;
  }
}

extern "C" int sisr_sft_combine_fwd(const float* x, long x_stride, const float* y2, const float* md, float* out,
                                    long out_stride, long npix, int relu, void* stream) {
  if (!x || !y2 || !out || npix <= 0 || x_stride < 64 || out_stride < 64 || (x_stride & 3) || (out_stride & 3)) return SISR_ERR_ARG;
  if (md && out_stride < 128) return SISR_ERR_ARG;
  if (!sisr_aligned16(x) || !sisr_aligned16(y2) || !sisr_aligned16(out) || !sisr_aligned16(md)) return SISR_ERR_ALIGN;
  hipLaunchKernelGGL(sft_combine_fwd_kernel, dim3(sft_blocks(npix * 16)), dim3(256), 0, (hipStream_t)stream, x, x_stride, y2,
                     md, out, out_stride, npix, relu);
  return sisr_check_launch();
}

extern "C" int sisr_sft_combine_bwd(const float* dout, long dout_stride, const float* x, long x_stride, const float* y2,
                                    float* dx, float* dy2, long npix, int relu, void* stream) {
  if (!dout || !x || !y2 || !dx || !dy2 || npix <= 0 || dout_stride < 64 || x_stride < 64 || (dout_stride & 3) || (x_stride & 3))
    return SISR_ERR_ARG;
  if (!sisr_aligned16(dout) || !sisr_aligned16(x) || !sisr_aligned16(y2) || !sisr_aligned16(dx) || !sisr_aligned16(dy2))
    return SISR_ERR_ALIGN;
  hipLaunchKernelGGL(sft_combine_bwd_kernel, dim3(sft_blocks(npix * 16)), dim3(256), 0, (hipStream_t)stream, dout,
                     dout_stride, x, x_stride, y2, dx, dy2, npix, relu);
  return sisr_check_launch();
}

// ------------------------------------------------------------------ strided 64-channel helpers
// op 0: out = a                      (copy: metadata chunk fill)
// op 1: out = a + b                  (fea_mid + fea_bef; gradient sums)
// op 2: out = leaky(a)               (a > 0 ? a : 0.2 a)
// op 3: out = b * (a > 0 ? 1 : 0.2)  (LeakyReLU backward: a = the activation's output, b = incoming gradient)
// op 4: out = a * b                  (WeakSft with as many maps as features, and its input gradient)
// op 5: out = relu(a)
// op 6: out = a > 0 ? b : 0          (ReLU backward: a = the activation's output)
// op 7: out = a * b[channel 0]       (WeakSft with one map: b's first channel broadcast over the 64 features)
__global__ __launch_bounds__(256) void map64_kernel(const float* a, long as, const float* b, long bs,
                                                    float* out, long os, long npix, int op) {
  const long total = npix * 16;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long p = i >> 4;
    const int c4 = (int)(i & 15) * 4;
    const f32x4 av = *reinterpret_cast<const f32x4*>(a + p * as + c4);
    f32x4 o = av;
    if (op == 1) {
      o = av + *reinterpret_cast<const f32x4*>(b + p * bs + c4);
    } else if (op == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = av[e] > 0.f ? av[e] : 0.2f * av[e];
    } else if (op == 3) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(b + p * bs + c4);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = av[e] > 0.f ? bv[e] : 0.2f * bv[e];
    } else if (op == 4) {
      o = av * *reinterpret_cast<const f32x4*>(b + p * bs + c4);
    } else if (op == 5) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf(av[e], 0.f);
    } else if (op == 6) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(b + p * bs + c4);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = av[e] > 0.f ? bv[e] : 0.f;
    } else if (op == 7) {
      o = av * b[p * bs];
    }
    *reinterpret_cast<f32x4*>(out + p * os + c4) = o;
  }
}

extern "C" int sisr_map64(const float* a, long a_stride, const float* b, long b_stride, float* out, long out_stride, long npix,
                          int op, void* stream) {
  if (!a || !out || npix <= 0 || op < 0 || op > 7 || ((op == 1 || op == 3 || op == 4 || op == 6 || op == 7) && !b)) return SISR_ERR_ARG;
  if ((a_stride & 3) || (out_stride & 3) || (b && (b_stride & 3)) || a_stride < 64 || out_stride < 64) return SISR_ERR_ARG;
  if (!sisr_aligned16(a) || !sisr_aligned16(b) || !sisr_aligned16(out)) return SISR_ERR_ALIGN;
  hipLaunchKernelGGL(map64_kernel, dim3(sft_blocks(npix * 16)), dim3(256), 0, (hipStream_t)stream, a, a_stride, b, b_stride, out,
                     out_stride, npix, op);
  return sisr_check_launch();
}

// ------------------------------------------------------------------ 9x9 output conv, 64 -> 3 (OIHW weight [3][64][9][9])
// All three directions run on v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate) with the 3-channel side staged in LDS:
//   forward  conv_rgb_out.h (shared with the 3x3 64 -> 3 tail conv): Z[q][(co, kw)] = sum_{kh, ci} x[q + (kh - 4) rows][ci]
//            w[co][ci][kh][kw] (M = 32 pixels of a row, N = 27 -> 32, K = 9 x 64), then y[co][p] = b[co] + sum_kw Z[p + kw - 4][(co, kw)]
//   dgrad    dx[p][ci] = sum_{n = (co, kh, kw)} dy[co][p + 4 - (kh, kw)] w[n][ci]        (M = pixels, N = 64, K = 243 -> 270:
//            kw padded to 10 so that the two K-halves of a lane pair are neighbouring columns), LeakyReLU' mask epilogue
//   wgrad    dw[ci][n] = sum_p x[p][ci] dy[co][p + 4 - (kh, kw)]                         (M = 64, N = 243 -> 256, K = pixels),
//            per-workgroup partial sums, ordered second stage
// Operand maps of the instruction (as in conv3x3_mfma.hip): A lane = (row i = lane & 31, k = lane >> 5), B lane = (column
// j = lane & 31, k = lane >> 5), D register r = row (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of column lane & 31.
#define T9 81
#define C9_GRID 512  // persistent grid (2 workgroups per CU); fixed, so the summation order does not depend on the device

// ---- input gradient
#define D9_TR 8
#define D9_HC 40           // 32 + 8 halo columns
#define D9_HS (16 * D9_HC)  // one channel's halo: (8 + 8) rows
#define D9_W2 (270 * 64)   // floats: [(co, kh, kp, k)][ci], kw = 2 kp + k, kw == 9 -> 0
__global__ __launch_bounds__(256) void conv9_dgrad_mfma_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                               const float* __restrict__ mask, float* __restrict__ dx, int B,
                                                               int H, int W, int tiles_w, int tiles_h, long ntiles) {
  extern __shared__ __attribute__((aligned(16))) float lds9[];
  float* w2 = lds9;
  float* dyh = lds9 + D9_W2 + 4;  // 4 floats of zero padding in front: kw == 9 reads column -1 (times a zero weight)
  for (int i = threadIdx.x; i < D9_W2; i += 256) {
    const int ci = i & 63, n = i >> 6, k = n & 1, kp = (n >> 1) % 5, ckh = n / 10;
    const int kw = 2 * kp + k;
    w2[i] = kw < 9 ? w[((long)(ckh / 9) * 64 + ci) * T9 + (ckh % 9) * 9 + kw] : 0.f;
  }
  if (threadIdx.x < 4) lds9[D9_W2 + threadIdx.x] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, kk = lane >> 5;
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int tw = (int)(t % tiles_w);
    const long rr = t / tiles_w;
    const int th = (int)(rr % tiles_h), b = (int)(rr / tiles_h);
    const int ty0 = th * D9_TR, tx0 = tw * 32;
    __syncthreads();  // the previous tile's gathers are done (and, first time, w2 is staged)
    for (int i = threadIdx.x; i < 3 * D9_HS; i += 256) {
      const int co = i / D9_HS, rem = i - co * D9_HS, hr = rem / D9_HC, hc = rem - hr * D9_HC;
      const int gy = ty0 - 4 + hr, gx = tx0 - 4 + hc;
      dyh[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? dy[(((long)b * 3 + co) * H + gy) * W + gx] : 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      const int py = wave + 4 * half, gy = ty0 + py;
      if (gy >= H) break;  // uniform per wave
      f32x16 acc0 = {0}, acc1 = {0};
      const float* ap = dyh + py * D9_HC + li - kk + 8;
      const float* bp = w2 + kk * 64 + li;
#pragma unroll 1
      for (int ckh = 0; ckh < 27; ++ckh) {
        const int co = ckh / 9, kh = ckh - co * 9;
        const float* a_row = ap + co * D9_HS + (8 - kh) * D9_HC;
        const float* b_row = bp + ckh * 640;
#pragma unroll
        for (int kp = 0; kp < 5; ++kp) {
          const float a = a_row[-2 * kp];
          const float b0 = b_row[kp * 128], b1 = b_row[kp * 128 + 32];
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
        }
      }
      const long row = (((long)b * H + gy) * W + tx0) * 64;
      float m0[16], m1[16];
      if (mask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = (r & 3) + 8 * (r >> 2) + 4 * kk;
          const long off = row + (long)min(px, W - 1 - tx0) * 64 + li;
          m0[r] = mask[off];
          m1[r] = mask[off + 32];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int px = (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (tx0 + px < W) {
          float v0 = acc0[r], v1 = acc1[r];
          if (mask) {
            v0 = m0[r] > 0.f ? v0 : 0.2f * v0;
            v1 = m1[r] > 0.f ? v1 : 0.2f * v1;
          }
          dx[row + (long)px * 64 + li] = v0;
          dx[row + (long)px * 64 + 32 + li] = v1;
        }
      }
    }
  }
}

// ---- weight gradient
#define W9_TR 8
#define W9_TC 64
#define W9_HC (W9_TC + 8)
#define W9_HS ((W9_TR + 8) * W9_HC)
#define W9_PART (64 * 256 + 4)  // floats per workgroup: [ci][n = co * 81 + tap, 256 wide] + 3 bias sums
__global__ __launch_bounds__(256) void conv9_wgrad_mfma_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ part, int B, int H, int W, int tiles_w,
                                                               int tiles_h, long ntiles) {
  __shared__ float dyh[3 * W9_HS];
  __shared__ float bred[3 * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kk = lane >> 5;
  int nbase[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int n = min((2 * wave + u) * 32 + li, 242);  // columns >= 243 are computed on a valid address and never stored
    const int co = n / T9, tp = n - co * T9, kh = tp / 9, kw = tp - kh * 9;
    nbase[u] = co * W9_HS + (8 - kh) * W9_HC + (8 - kw) + kk;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) acc[u][mt] = (f32x16){0};
  float bs0 = 0.f, bs1 = 0.f, bs2 = 0.f;
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int tw = (int)(t % tiles_w);
    const long rr = t / tiles_w;
    const int th = (int)(rr % tiles_h), b = (int)(rr / tiles_h);
    const int ty0 = th * W9_TR, tx0 = tw * W9_TC;
    __syncthreads();
    for (int i = tid; i < 3 * W9_HS; i += 256) {
      const int co = i / W9_HS, rem = i - co * W9_HS, hr = rem / W9_HC, hc = rem - hr * W9_HC;
      const int gy = ty0 - 4 + hr, gx = tx0 - 4 + hc;
      const float v = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? dy[(((long)b * 3 + co) * H + gy) * W + gx] : 0.f;
      dyh[i] = v;
      const bool own = hr >= 4 && hr < 4 + W9_TR && hc >= 4 && hc < 4 + W9_TC;  // the tile's own pixels: bias gradient
      const float vo = own ? v : 0.f;
      bs0 += co == 0 ? vo : 0.f;
      bs1 += co == 1 ? vo : 0.f;
      bs2 += co == 2 ? vo : 0.f;
    }
    __syncthreads();
    const int rows = min(W9_TR, H - ty0);
#pragma unroll 1
    for (int py = 0; py < rows; ++py) {
      const float* xrow = x + (((long)b * H + ty0 + py) * W) * 64 + li;
#pragma unroll 1
      for (int pxb = 0; pxb < W9_TC; pxb += 16) {
        if (tx0 + pxb >= W) break;
        float a[2][8], bq[2][8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int gx = tx0 + pxb + 2 * q + kk;
          const float* xa = xrow + (long)min(gx, W - 1) * 64;
          const float a0 = xa[0], a1 = xa[32];
          a[0][q] = gx < W ? a0 : 0.f;
          a[1][q] = gx < W ? a1 : 0.f;
          bq[0][q] = dyh[nbase[0] + py * W9_HC + pxb + 2 * q];
          bq[1][q] = dyh[nbase[1] + py * W9_HC + pxb + 2 * q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
              acc[u][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][q], bq[u][q], acc[u][mt], 0, 0, 0);
      }
    }
  }
  float* out = part + (long)blockIdx.x * W9_PART;
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        out[ci * 256 + (2 * wave + u) * 32 + li] = acc[u][mt][r];
      }
  __syncthreads();
  bred[tid] = bs0;
  bred[256 + tid] = bs1;
  bred[512 + tid] = bs2;
  __syncthreads();
  if (tid < 3) {
    float sacc = 0.f;
    for (int k = 0; k < 256; ++k) sacc += bred[tid * 256 + k];
    out[64 * 256 + tid] = sacc;
  }
}

__global__ __launch_bounds__(256) void conv9_wgrad_reduce_kernel(const float* __restrict__ part, int nparts,
                                                                 float* __restrict__ dw, float* __restrict__ db) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 64 * 243) {
    const int ci = i / 243, n = i - ci * 243;
    float s = 0.f;
    for (int k = 0; k < nparts; ++k) s += part[(long)k * W9_PART + ci * 256 + n];
    const int co = n / T9, tp = n - co * T9;
    dw[((long)co * 64 + ci) * T9 + tp] = s;
  } else if (i < 64 * 243 + 3 && db) {
    const int co = i - 64 * 243;
    float s = 0.f;
    for (int k = 0; k < nparts; ++k) s += part[(long)k * W9_PART + 64 * 256 + co];
    db[co] = s;
  }
}

// clamp to [0, 1] (ref: SFTMD.forward `torch.clamp(out, min, max)`): op 0 out = clamp(a); op 1 out = b * [0 <= a <= 1]
__global__ __launch_bounds__(256) void clamp01_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ out, long n, int op) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = a[i];
    out[i] = op == 0 ? fminf(fmaxf(v, 0.f), 1.f) : ((v >= 0.f && v <= 1.f) ? b[i] : 0.f);
  }
}


extern "C" int sisr_conv9_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, void* stream) {
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0) return SISR_ERR_ARG;
  if (!sisr_aligned16(x)) return SISR_ERR_ALIGN;
  RgbOutParams p = {};
  p.x = x;
  p.sB = (long)H * W * 64;
  p.sH = (long)W * 64;
  p.sW = 64;
  p.w = w;
  p.so = 64 * T9;
  p.si = T9;
  p.bias = bias;
  p.y = y;
  p.B = B;
  p.H = H;
  p.W = W;
  return rgb_out_launch<9, 1>(p, stream);
}

extern "C" int sisr_conv9_dgrad(const float* dy, const float* w, const float* leaky_mask, float* dx, int B, int H, int W,
                                void* stream) {
  if (!dy || !w || !dx || B <= 0 || H <= 0 || W <= 0) return SISR_ERR_ARG;
  const int tiles_w = (W + 31) / 32, tiles_h = (H + D9_TR - 1) / D9_TR;
  const long ntiles = (long)B * tiles_h * tiles_w;
  const size_t lds = (D9_W2 + 4 + 3 * D9_HS) * sizeof(float);
  SISR_ALLOW_LDS(conv9_dgrad_mfma_kernel, lds);
  hipLaunchKernelGGL(conv9_dgrad_mfma_kernel, dim3((unsigned)(ntiles < C9_GRID ? ntiles : C9_GRID)), dim3(256), lds,
                     (hipStream_t)stream, dy, w, leaky_mask, dx, B, H, W, tiles_w, tiles_h, ntiles);
  return sisr_check_launch();
}

static long conv9_wgrad_tiles(int B, int H, int W, int* tiles_w, int* tiles_h) {
  *tiles_w = (W + W9_TC - 1) / W9_TC;
  *tiles_h = (H + W9_TR - 1) / W9_TR;
  return (long)B * *tiles_h * *tiles_w;
}
extern "C" size_t sisr_conv9_wgrad_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  int tw, th;
  const long nt = conv9_wgrad_tiles(B, H, W, &tw, &th);
  return (size_t)(nt < C9_GRID ? nt : C9_GRID) * W9_PART * sizeof(float);
}
extern "C" int sisr_conv9_wgrad(const float* x, const float* dy, float* dw, float* db, float* workspace, size_t workspace_bytes,
                                int B, int H, int W, void* stream) {
  if (!x || !dy || !dw || !workspace || B <= 0 || H <= 0 || W <= 0) return SISR_ERR_ARG;
  if (workspace_bytes < sisr_conv9_wgrad_workspace_bytes(B, H, W)) return SISR_ERR_ARG;
  int tiles_w, tiles_h;
  const long ntiles = conv9_wgrad_tiles(B, H, W, &tiles_w, &tiles_h);
  const int parts = (int)(ntiles < C9_GRID ? ntiles : C9_GRID);
  hipLaunchKernelGGL(conv9_wgrad_mfma_kernel, dim3(parts), dim3(256), 0, (hipStream_t)stream, x, dy, workspace, B, H, W, tiles_w,
                     tiles_h, ntiles);
  int rc = sisr_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(conv9_wgrad_reduce_kernel, dim3((64 * 243 + 3 + 255) / 256), dim3(256), 0, (hipStream_t)stream, workspace,
                     parts, dw, db);
  return sisr_check_launch();
}

extern "C" int sisr_clamp01(const float* a, const float* grad, float* out, long n, int backward, void* stream) {
  if (!a || !out || n <= 0 || (backward && !grad)) return SISR_ERR_ARG;
  hipLaunchKernelGGL(clamp01_kernel, dim3(sft_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, grad, out, n, backward);
  return sisr_check_launch();
}
