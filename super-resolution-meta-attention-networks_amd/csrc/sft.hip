// SFTMD pieces that are not 3x3 convs over 64-channel chunks.
// ref: Code/SISR/models/SFTMD_variants/architectures.py:25-56 (StandardSft: x * sigmoid(mul) + add),
//      :110-176 (SFTMD: LeakyReLU(0.2) head / upscale, 9x9 64 -> 3 output conv, clamp to [0, 1]).
// The 3x3 convs of the network run on the MFMA kernels of conv3x3_mfma.hip (LeakyReLU as epilogue / mask slope); here:
//   compose_oihw2   two weight blocks placed into one zero-padded OIHW tensor (the merged / block-diagonal SFT convs) and
//                   the reverse split of its gradient
//   sft_combine     out = [relu](x * sigmoid(y2[:64]) + y2[64:]) on pixel-strided maps, + copy of the metadata chunk;
//                   backward -> dx and d y2 (ReLU mask recomputed)
//   copy_chunk / add2 / leaky  strided 64-channel helpers (metadata chunk fill, fea_mid + fea_bef, LeakyReLU after the
//                   3-channel head conv and its backward)
//   conv9_*         the 9x9 64 -> 3 output conv: forward, input gradient (+ LeakyReLU mask of the map it feeds back into),
//                   weight / bias gradient (ordered two-stage sum), and the clamp's forward / backward
// All HBM- / VALU-bound and small next to the network's 3x3 convs; written for clarity, coalesced 256-B rows.
#include "sisr_common.h"

__device__ __forceinline__ float sft_sigmoid(float z) { return 1.f / (1.f + expf(-z)); }

static unsigned sft_blocks(long n) {
  long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 65535 ? 65535 : b));
}

// ------------------------------------------------------------------ weight composition
struct Compose2 {
  int cop, cip, taps;
  int oa0, ia0, coa, cia, ob0, ib0, cob, cib;
};
// split == 0: dst[cop][cip][taps] = 0 except block A (a[coa][cia][taps]) at (oa0, ia0) and block B at (ob0, ib0)
// split == 1: a / b <- the two blocks of dst (gradient of the composition)
__global__ __launch_bounds__(256) void compose_oihw2_kernel(float* a, float* b, float* dst, Compose2 c, int split) {
  const long total = (long)c.cop * c.cip * c.taps;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int t = (int)(i % c.taps);
    const long r = i / c.taps;
    const int o = (int)(r / c.cip), ci = (int)(r - (long)o * c.cip);
    const bool inA = o >= c.oa0 && o < c.oa0 + c.coa && ci >= c.ia0 && ci < c.ia0 + c.cia;
    const bool inB = o >= c.ob0 && o < c.ob0 + c.cob && ci >= c.ib0 && ci < c.ib0 + c.cib;
    const long ia = ((long)(o - c.oa0) * c.cia + (ci - c.ia0)) * c.taps + t;
    const long ib = ((long)(o - c.ob0) * c.cib + (ci - c.ib0)) * c.taps + t;
    if (!split) {
      dst[i] = inA ? a[ia] : (inB ? b[ib] : 0.f);
    } else {
      if (inA) a[ia] = dst[i];
      if (inB) b[ib] = dst[i];
    }
  }
}

extern "C" int sisr_compose_oihw2(float* a, float* b, float* dst, int cop, int cip, int taps, int oa0, int ia0, int coa,
                                  int cia, int ob0, int ib0, int cob, int cib, int split, void* stream) {
  if (!a || !b || !dst || cop <= 0 || cip <= 0 || taps <= 0 || coa <= 0 || cia <= 0 || cob <= 0 || cib <= 0)
    return SISR_ERR_ARG;
  if (oa0 < 0 || ia0 < 0 || ob0 < 0 || ib0 < 0 || oa0 + coa > cop || ob0 + cob > cop || ia0 + cia > cip || ib0 + cib > cip)
    return SISR_ERR_ARG;
  const Compose2 c = {cop, cip, taps, oa0, ia0, coa, cia, ob0, ib0, cob, cib};
  hipLaunchKernelGGL(compose_oihw2_kernel, dim3(sft_blocks((long)cop * cip * taps)), dim3(256), 0, (hipStream_t)stream, a, b,
                     dst, c, split);
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
    }
    *reinterpret_cast<f32x4*>(out + p * os + c4) = o;
  }
}

extern "C" int sisr_map64(const float* a, long a_stride, const float* b, long b_stride, float* out, long out_stride, long npix,
                          int op, void* stream) {
  if (!a || !out || npix <= 0 || op < 0 || op > 3 || ((op == 1 || op == 3) && !b)) return SISR_ERR_ARG;
  if ((a_stride & 3) || (out_stride & 3) || (b && (b_stride & 3)) || a_stride < 64 || out_stride < 64) return SISR_ERR_ARG;
  if (!sisr_aligned16(a) || !sisr_aligned16(b) || !sisr_aligned16(out)) return SISR_ERR_ALIGN;
  hipLaunchKernelGGL(map64_kernel, dim3(sft_blocks(npix * 16)), dim3(256), 0, (hipStream_t)stream, a, a_stride, b, b_stride, out,
                     out_stride, npix, op);
  return sisr_check_launch();
}

// ------------------------------------------------------------------ 9x9 output conv, 64 -> 3 (OIHW weight [3][64][9][9])
#define K9 9
#define T9 81
// Weights re-ordered once per launch into LDS as [tap][c4 = 16][co = 3][4 ch]: 48 B per (tap, lane), conflict-free.
__device__ __forceinline__ void conv9_stage_w(const float* __restrict__ w, float* wl) {
  for (int i = threadIdx.x; i < T9 * 16 * 12; i += 256) {
    const int e = i & 3, co = (i >> 2) % 3, c4 = (i / 12) & 15, t = i / 192;
    wl[i] = w[((long)co * 64 + c4 * 4 + e) * T9 + t];
  }
}

// forward: 16 lanes per output pixel, a lane owns 4 input channels; x NHWC [B][H][W][64], y NCHW [B][3][H][W] (pre-clamp)
__global__ __launch_bounds__(256) void conv9_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int B, int H,
                                                        int W) {
  extern __shared__ __attribute__((aligned(16))) float wl[];
  conv9_stage_w(w, wl);
  __syncthreads();
  const int c4 = threadIdx.x & 15;
  const long hw = (long)H * W, npix = (long)B * hw;
  const long pend = (npix + 15) & ~15L;
  for (long pix0 = (long)blockIdx.x * 16 + (threadIdx.x >> 4); pix0 < pend; pix0 += (long)gridDim.x * 16) {
    const bool live = pix0 < npix;
    const long pix = live ? pix0 : npix - 1;
    const long b = pix / hw, r = pix - b * hw;
    const int h = (int)(r / W), wc = (int)(r - (long)h * W);
    const float* xb = x + b * hw * 64 + c4 * 4;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int kh = 0; kh < K9; ++kh) {
      const int gh = h + kh - 4;
      if (gh < 0 || gh >= H) continue;  // uniform within the 16-lane pixel group
#pragma unroll
      for (int kw = 0; kw < K9; ++kw) {
        const int gw = wc + kw - 4;
        const bool ok = gw >= 0 && gw < W;
        const f32x4 xv = sisr_keep_if(*reinterpret_cast<const f32x4*>(xb + ((long)gh * W + min(max(gw, 0), W - 1)) * 64), ok);
        const float* wt = wl + ((kh * K9 + kw) * 16 + c4) * 12;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wt), w1 = *reinterpret_cast<const f32x4*>(wt + 4),
                    w2 = *reinterpret_cast<const f32x4*>(wt + 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a0 += xv[e] * w0[e];
          a1 += xv[e] * w1[e];
          a2 += xv[e] * w2[e];
        }
      }
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      a0 += __shfl_xor(a0, o);
      a1 += __shfl_xor(a1, o);
      a2 += __shfl_xor(a2, o);
    }
    if (live && c4 < 3) {
      const float v = (c4 == 0 ? a0 : (c4 == 1 ? a1 : a2)) + (bias ? bias[c4] : 0.f);
      y[(b * 3 + c4) * hw + r] = v;
    }
  }
}

// input gradient: dx[b][h][w][ci] = sum_{co, kh, kw} dy[b][co][h + 4 - kh][w + 4 - kw] * w[co][ci][kh][kw], then the
// LeakyReLU(0.2) mask of the map x fed forward (mask nullable).  dy NCHW (already clamp-masked), dx NHWC.
__global__ __launch_bounds__(256) void conv9_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                          const float* __restrict__ mask, float* __restrict__ dx, int B, int H,
                                                          int W) {
  extern __shared__ __attribute__((aligned(16))) float wl[];
  conv9_stage_w(w, wl);
  __syncthreads();
  const int c4 = threadIdx.x & 15;
  const long hw = (long)H * W, npix = (long)B * hw;
  const long pend = (npix + 15) & ~15L;
  for (long pix0 = (long)blockIdx.x * 16 + (threadIdx.x >> 4); pix0 < pend; pix0 += (long)gridDim.x * 16) {
    const bool live = pix0 < npix;
    const long pix = live ? pix0 : npix - 1;
    const long b = pix / hw, r = pix - b * hw;
    const int h = (int)(r / W), wc = (int)(r - (long)h * W);
    const float* db = dy + b * 3 * hw;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int kh = 0; kh < K9; ++kh) {
      const int gh = h + 4 - kh;
      if (gh < 0 || gh >= H) continue;
#pragma unroll
      for (int kw = 0; kw < K9; ++kw) {
        const int gw = wc + 4 - kw;
        if (gw < 0 || gw >= W) continue;
        const long at = (long)gh * W + gw;
        const float d0 = db[at], d1 = db[hw + at], d2 = db[2 * hw + at];
        const float* wt = wl + ((kh * K9 + kw) * 16 + c4) * 12;
        acc += *reinterpret_cast<const f32x4*>(wt) * d0 + *reinterpret_cast<const f32x4*>(wt + 4) * d1 +
               *reinterpret_cast<const f32x4*>(wt + 8) * d2;
      }
    }
    if (live) {
      if (mask) {
        const f32x4 m = *reinterpret_cast<const f32x4*>(mask + pix * 64 + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = m[e] > 0.f ? acc[e] : 0.2f * acc[e];
      }
      *reinterpret_cast<f32x4*>(dx + pix * 64 + c4 * 4) = acc;
    }
  }
}

// weight gradient partials: block = 256 threads = (16 tap groups) x (16 c4); tap group g owns taps g, g + 16, ... (6 slots);
// a block walks `span` pixels and writes part[block][tap][co][ci] (ordered second stage: conv9_wgrad_reduce_kernel).
#define W9_SLOTS 6
__global__ __launch_bounds__(256) void conv9_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ part, int B, int H, int W, long span) {
  const int c4 = threadIdx.x & 15, tg = threadIdx.x >> 4;
  const long hw = (long)H * W, npix = (long)B * hw;
  f32x4 acc[W9_SLOTS][3];
#pragma unroll
  for (int s = 0; s < W9_SLOTS; ++s)
#pragma unroll
    for (int co = 0; co < 3; ++co) acc[s][co] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[3] = {0.f, 0.f, 0.f};
  const long p0 = (long)blockIdx.x * span, p1 = min(p0 + span, npix);
  for (long pix = p0; pix < p1; ++pix) {
    const long b = pix / hw, r = pix - b * hw;
    const int h = (int)(r / W), wc = (int)(r - (long)h * W);
    const float* db = dy + b * 3 * hw + r;
    const float d0 = db[0], d1 = db[hw], d2 = db[2 * hw];
    bsum[0] += d0;
    bsum[1] += d1;
    bsum[2] += d2;
    const float* xb = x + b * hw * 64 + c4 * 4;
#pragma unroll
    for (int s = 0; s < W9_SLOTS; ++s) {
      const int t = tg + 16 * s;
      if (t < T9) {
        const int gh = h + t / K9 - 4, gw = wc + t % K9 - 4;
        if (gh >= 0 && gh < H && gw >= 0 && gw < W) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(xb + ((long)gh * W + gw) * 64);
          acc[s][0] += xv * d0;
          acc[s][1] += xv * d1;
          acc[s][2] += xv * d2;
        }
      }
    }
  }
  float* out = part + (long)blockIdx.x * (T9 * 3 * 64 + 4);
#pragma unroll
  for (int s = 0; s < W9_SLOTS; ++s) {
    const int t = tg + 16 * s;
    if (t < T9) {
#pragma unroll
      for (int co = 0; co < 3; ++co) *reinterpret_cast<f32x4*>(out + ((long)t * 3 + co) * 64 + c4 * 4) = acc[s][co];
    }
  }
  if (threadIdx.x < 3) out[T9 * 3 * 64 + threadIdx.x] = bsum[threadIdx.x];  // every thread summed the same dy values
}

__global__ __launch_bounds__(256) void conv9_wgrad_reduce_kernel(const float* __restrict__ part, int nparts,
                                                                 float* __restrict__ dw, float* __restrict__ db) {
  const int stride = T9 * 3 * 64 + 4;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < T9 * 3 * 64) {
    float s = 0.f;
    for (int k = 0; k < nparts; ++k) s += part[(long)k * stride + i];
    const int ci = i & 63, co = (i >> 6) % 3, t = i / 192;
    dw[((long)co * 64 + ci) * T9 + t] = s;
  } else if (i < T9 * 3 * 64 + 3 && db) {
    float s = 0.f;
    for (int k = 0; k < nparts; ++k) s += part[(long)k * stride + i];
    db[i - T9 * 3 * 64] = s;
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

#define CONV9_LDS (T9 * 16 * 12 * sizeof(float))  // 62 208 B

extern "C" int sisr_conv9_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, void* stream) {
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0) return SISR_ERR_ARG;
  if (!sisr_aligned16(x)) return SISR_ERR_ALIGN;
  const long npix = (long)B * H * W;
  long blocks = (npix + 15) / 16;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(conv9_fwd_kernel, dim3((unsigned)blocks), dim3(256), CONV9_LDS, (hipStream_t)stream, x, w, bias, y, B, H, W);
  return sisr_check_launch();
}

extern "C" int sisr_conv9_dgrad(const float* dy, const float* w, const float* leaky_mask, float* dx, int B, int H, int W,
                                void* stream) {
  if (!dy || !w || !dx || B <= 0 || H <= 0 || W <= 0) return SISR_ERR_ARG;
  if (!sisr_aligned16(dx) || !sisr_aligned16(leaky_mask)) return SISR_ERR_ALIGN;
  const long npix = (long)B * H * W;
  long blocks = (npix + 15) / 16;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(conv9_dgrad_kernel, dim3((unsigned)blocks), dim3(256), CONV9_LDS, (hipStream_t)stream, dy, w, leaky_mask,
                     dx, B, H, W);
  return sisr_check_launch();
}

static int conv9_parts(long npix) {
  long n = (npix + 2047) / 2048;  // >= 2048 pixels per block
  if (n > 1024) n = 1024;
  return (int)(n < 1 ? 1 : n);
}
extern "C" size_t sisr_conv9_wgrad_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return (size_t)conv9_parts((long)B * H * W) * (T9 * 3 * 64 + 4) * sizeof(float);
}
extern "C" int sisr_conv9_wgrad(const float* x, const float* dy, float* dw, float* db, float* workspace, size_t workspace_bytes,
                                int B, int H, int W, void* stream) {
  if (!x || !dy || !dw || !workspace || B <= 0 || H <= 0 || W <= 0) return SISR_ERR_ARG;
  if (workspace_bytes < sisr_conv9_wgrad_workspace_bytes(B, H, W)) return SISR_ERR_ARG;
  if (!sisr_aligned16(x) || !sisr_aligned16(workspace)) return SISR_ERR_ALIGN;
  const long npix = (long)B * H * W;
  const int parts = conv9_parts(npix);
  const long span = (npix + parts - 1) / parts;
  hipLaunchKernelGGL(conv9_wgrad_kernel, dim3(parts), dim3(256), 0, (hipStream_t)stream, x, dy, workspace, B, H, W, span);
  int rc = sisr_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(conv9_wgrad_reduce_kernel, dim3((T9 * 3 * 64 + 3 + 255) / 256), dim3(256), 0, (hipStream_t)stream, workspace,
                     parts, dw, db);
  return sisr_check_launch();
}

extern "C" int sisr_clamp01(const float* a, const float* grad, float* out, long n, int backward, void* stream) {
  if (!a || !out || n <= 0 || (backward && !grad)) return SISR_ERR_ARG;
  hipLaunchKernelGGL(clamp01_kernel, dim3(sft_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, grad, out, n, backward);
  return sisr_check_launch();
}
