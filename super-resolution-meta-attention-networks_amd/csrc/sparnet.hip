// SPARNet / QSPARNet (ref: Code/SISR/models/SPARNet/blocks.py:10-243, architectures.py:7-155): everything around the 3x3
// convolutions, which run on the MFMA conv kernels of conv3x3_mfma.hip / wgrad3x3_mfma.hip.
//
// A reference ConvLayer (blocks.py:69-103) is  [nearest x2] -> ReflectionPad2d(1) -> Conv2d(3x3, stride 1 | 2, no padding)
// -> [BatchNorm2d] -> [LeakyReLU(0.2)].  Here:
//   pad_reflect_up     gathers the (optionally 2x nearest-upsampled) map with its reflected ring in one pass: (H, W) ->
//                      (up H + 2, up W + 2); the zero-padded "same" MFMA conv over that map equals the reference's unpadded
//                      conv on every pixel of its interior;
//   crop_stride        takes that interior, every `stride`-th pixel (a stride-2 conv is the stride-1 conv subsampled);
//   the adjoints       embed_stride (gradient placed back into a zero map of the padded size; the MFMA input-gradient and
//                      weight-gradient kernels then run on the padded geometry unchanged) and pad_reflect_up_bwd (folds the
//                      ring and the 2 x 2 replicas back onto the source pixel, in index order);
//   bn_*               BatchNorm2d over (B, H, W) per channel with the LeakyReLU folded in: two-pass statistics (mean, then
//                      centred squares: no E[x^2] - E[x]^2 cancellation), partial sums per workgroup added in index order
//                      (deterministic), running statistics updated as torch does (momentum, unbiased variance);
//   spar_combine_*     the spatial-attention product of HourGlassBlock.forward (blocks.py:236-243) with the block's residual
//                      sum (blocks.py:166):  out = identity + x * sigmoid(logit),  logit = channel 0 of the 64 -> 1 conv's
//                      zero-padded 64-channel result.
// Maps are channels-last with the channel count zero-padded to a multiple of 64 (C_real <= C: the padded channels stay zero
// through every kernel here).  HBM-bound passes over small maps (the network works at 128^2 ... 4^2 pixels): one thread per
// 16-byte piece, coalesced; nothing here is on the headline path.
#include "sisr_common.h"

static inline unsigned sp_blocks(long n, long cap = 65535) {
  long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

// padded index i (0 .. up*n + 1) -> source index in the un-upsampled map of n rows / columns
__device__ __forceinline__ int sp_src(int i, int n, int up) {
  const int nu = n * up;
  int u = i - 1;
  if (u < 0) u = -u;
  if (u >= nu) u = 2 * nu - 2 - u;
  return up == 2 ? (u >> 1) : u;
}

__global__ __launch_bounds__(256) void pad_reflect_up_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, int H, int W,
                                                             int c4n, int up, long total) {
  const int Hp = up * H + 2, Wp = up * W + 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % c4n);
    long t = i / c4n;
    const int pw = (int)(t % Wp);
    t /= Wp;
    const int ph = (int)(t % Hp);
    const long b = t / Hp;
    y[i] = x[((b * H + sp_src(ph, H, up)) * W + sp_src(pw, W, up)) * c4n + c4];
  }
}

// adjoint: dx[b][h][w] = sum of dy over the padded positions that read (h, w); rows then columns in ascending order
__global__ __launch_bounds__(256) void pad_reflect_up_bwd_kernel(const f32x4* __restrict__ dy, f32x4* __restrict__ dx, int H,
                                                                 int W, int c4n, int up, long total) {
  const int Hp = up * H + 2, Wp = up * W + 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % c4n);
    long t = i / c4n;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const long b = t / H;
    // candidates: the ring positions 0 and n_p - 1, and the up interior positions up*h + 1 .. up*h + up
    int ri[4], rj[4], nr = 0, nc = 0;
    if (sp_src(0, H, up) == h) ri[nr++] = 0;
    for (int k = 1; k <= up; ++k) ri[nr++] = up * h + k;
    if (sp_src(Hp - 1, H, up) == h) ri[nr++] = Hp - 1;
    if (sp_src(0, W, up) == w) rj[nc++] = 0;
    for (int k = 1; k <= up; ++k) rj[nc++] = up * w + k;
    if (sp_src(Wp - 1, W, up) == w) rj[nc++] = Wp - 1;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < nr; ++a)
      for (int c = 0; c < nc; ++c) acc += dy[((b * Hp + ri[a]) * Wp + rj[c]) * c4n + c4];
    dx[i] = acc;
  }
}

// y[b][h][w] = yf[b][1 + s h][1 + s w]  (embed == 0)   |   yf = 0 except those positions <- y  (embed == 1)
__global__ __launch_bounds__(256) void crop_stride_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int Hf,
                                                          int Wf, int Ho, int Wo, int s, int c4n, long total, int embed) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % c4n);
    long t = i / c4n;
    if (!embed) {
      const int w = (int)(t % Wo);
      t /= Wo;
      const int h = (int)(t % Ho);
      const long b = t / Ho;
      dst[i] = src[((b * Hf + 1 + s * h) * Wf + 1 + s * w) * c4n + c4];
    } else {
      const int j = (int)(t % Wf);
      t /= Wf;
      const int r = (int)(t % Hf);
      const long b = t / Hf;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const int h = (r - 1) / s, w = (j - 1) / s;
      if (r >= 1 && j >= 1 && (r - 1) % s == 0 && (j - 1) % s == 0 && h < Ho && w < Wo)
        v = src[((b * Ho + h) * Wo + w) * c4n + c4];
      dst[i] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------ BatchNorm (+ LeakyReLU)
// x: [npix][C] (C = 64 or 128 ... multiple of 64, at most SP_MAXC).  A workgroup = (C / 4) channel lanes x (1024 / C) pixel
// lanes; workgroup k owns pixels [k * chunk, (k + 1) * chunk).
#define SP_MAXC 256
#define SP_MAXBLK 256

// z = x * sc + sh, the pre-activation value of the forward.  The backward needs its SIGN (the LeakyReLU mask) and must get the
// forward's: a value recomputed as xhat * gamma + beta rounds differently, and an element within an ulp of zero then takes
// slope 1 one way and `slope` the other -- one such element in 16 k moves the block's input gradient by 1e-3 (found with
// the float64 oracle).  So forward and backward share these three explicit operations, fma included.
__device__ __forceinline__ void sp_bn_coeffs(float g, float b, float mean, float inv, float& sc, float& sh) {
  sc = g * inv;
  sh = __builtin_fmaf(-mean, sc, b);
}
__device__ __forceinline__ float sp_bn_z(float x, float sc, float sh) { return __builtin_fmaf(x, sc, sh); }

// sum over k < n of part[k * stride + c], added in index order; sixteen loads in flight (the kernels below do this once per
// workgroup on the way to their real work: dependent loads one at a time cost 64 memory round trips)
__device__ __forceinline__ float sp_sum_parts(const float* __restrict__ part, int n, long stride, int c) {
  float s = 0.f;
  int k = 0;
  for (; k + 16 <= n; k += 16) {
    float t[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) t[u] = part[(long)(k + u) * stride + c];
#pragma unroll
    for (int u = 0; u < 16; ++u) s += t[u];
  }
  for (; k < n; ++k) s += part[(long)k * stride + c];
  return s;
}

// The same sums for every channel of the map, by the whole 256-thread workgroup: C <= 256 channels x J = 256 / C slices, slice j
// adds partials j, j + J, ... (sixteen loads in flight each), the J slice sums are then added in slice order.  out[c] (LDS) holds
// the result after the closing barrier; tmp: 256 floats of LDS.  With up to 256 partials per launch a lone thread per channel
// would walk them in sixteen dependent rounds (~10 us before every consumer's real work).
__device__ __forceinline__ void sp_sum_parts_wg(const float* __restrict__ part, int n, long stride, int C, float* out, float* tmp) {
  const int J = 256 / C > 0 ? 256 / C : 1;
  for (int c0 = 0; c0 < C; c0 += 256) {  // (one trip: C <= SP_MAXC = 256)
    const int c = c0 + (int)threadIdx.x % (C < 256 ? C : 256), j = (int)threadIdx.x / (C < 256 ? C : 256);
    float s = 0.f;
    if (j < J && c < C) {
      int k = j;
      for (; k + 15 * J < n; k += 16 * J) {
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = part[(long)(k + u * J) * stride + c];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += t[u];
      }
      for (; k < n; k += J) s += part[(long)k * stride + c];
    }
    __syncthreads();
    tmp[threadIdx.x] = s;
    __syncthreads();
    if (j == 0 && c < C) {
      float r = tmp[threadIdx.x];
      for (int q = 1; q < J; ++q) r += tmp[threadIdx.x + q * C];
      out[c] = r;
    }
  }
  __syncthreads();
}

// MODE 0: sum of x.  MODE 1: sum of (x - mean)^2, mean from the MODE-0 partials (every workgroup adds them in index order).
// MODE 2 (backward): sums of dz and dz * xhat,  dz = dy * act'(xhat * gamma + beta).
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ mean_in, const float* __restrict__ invstd_in,
                                                         const float* __restrict__ part_in, float* __restrict__ part_out,
                                                         long npix, int C, int C_real, float slope, long chunk) {
  __shared__ float stat[2 * SP_MAXC];
  __shared__ __attribute__((aligned(16))) float red[2 * 256 * 4];
  const int c4n = C >> 2, rows = 256 / c4n;
  const int c4 = threadIdx.x % c4n, row = threadIdx.x / c4n;
  const int nblk = gridDim.x;
  if (MODE == 1) {
    sp_sum_parts_wg(part_in, nblk, C, C, stat, red);
    for (int c = threadIdx.x; c < C; c += 256) stat[c] = stat[c] / (float)npix;
    __syncthreads();
  }
  f32x4 m4 = {0.f, 0.f, 0.f, 0.f}, i4 = m4, sc4 = m4, sh4 = m4;
  if (MODE == 1) m4 = *reinterpret_cast<const f32x4*>(stat + c4 * 4);
  if (MODE == 2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = c4 * 4 + e;
      m4[e] = mean_in[c];
      i4[e] = invstd_in[c];
      float a, b;
      sp_bn_coeffs(c < C_real ? gamma[c] : 0.f, c < C_real ? beta[c] : 0.f, m4[e], i4[e], a, b);
      sc4[e] = a;
      sh4[e] = b;
    }
  }
  const long p0 = (long)blockIdx.x * chunk, p1 = min(npix, p0 + chunk);
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, a2 = a;
  for (long p = p0 + row; p < p1; p += rows) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + p * C + c4 * 4);
    if (MODE == 0) a += v;
    if (MODE == 1) {
      const f32x4 d = v - m4;
      a += d * d;
    }
    if (MODE == 2) {
      const f32x4 xh = (v - m4) * i4;
      f32x4 dz = *reinterpret_cast<const f32x4*>(dy + p * C + c4 * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (!(sp_bn_z(v[e], sc4[e], sh4[e]) > 0.f)) dz[e] *= slope;
      a += dz;
      a2 += dz * xh;
    }
  }
  *reinterpret_cast<f32x4*>(red + threadIdx.x * 4) = a;
  if (MODE == 2) *reinterpret_cast<f32x4*>(red + 1024 + threadIdx.x * 4) = a2;
  __syncthreads();
  for (int c = threadIdx.x; c < (MODE == 2 ? 2 : 1) * C; c += 256) {
    const int which = c / C, cc = c - which * C;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += red[which * 1024 + (r * c4n + (cc >> 2)) * 4 + (cc & 3)];
    part_out[((long)blockIdx.x * (MODE == 2 ? 2 : 1) + which) * C + cc] = s;
  }
}

// y = act(xhat * gamma + beta).  TRAIN: statistics from the partials (every workgroup adds them in index order; workgroup 0
// also writes mean / invstd and updates the running statistics); else from running_mean / running_var.
template <bool TRAIN>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ part_sum, const float* __restrict__ part_sq,
                                                       int nblk, float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                       float* __restrict__ running_mean, float* __restrict__ running_var,
                                                       float momentum, float eps, float slope, long npix, int C, int C_real) {
  __shared__ __attribute__((aligned(16))) float sc[SP_MAXC], sh[SP_MAXC];
  __shared__ float tmp[256];
  if (TRAIN) {  // sums of x and of the centred squares, parked in sc / sh until the coefficients replace them
    sp_sum_parts_wg(part_sum, nblk, C, C, sc, tmp);
    sp_sum_parts_wg(part_sq, nblk, C, C, sh, tmp);
  }
  for (int c = threadIdx.x; c < C; c += 256) {
    float mean, var;
    if (TRAIN) {
      mean = sc[c] / (float)npix;
      var = sh[c] / (float)npix;
    } else {
      mean = c < C_real ? running_mean[c] : 0.f;
      var = c < C_real ? running_var[c] : 1.f;
    }
    const float inv = 1.f / sqrtf(var + eps);
    if (TRAIN && blockIdx.x == 0) {
      mean_out[c] = mean;
      invstd_out[c] = inv;
      if (running_mean && c < C_real) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        const float unbiased = npix > 1 ? var * ((float)npix / (float)(npix - 1)) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
      }
    }
    sp_bn_coeffs(c < C_real ? gamma[c] : 0.f, c < C_real ? beta[c] : 0.f, mean, inv, sc[c], sh[c]);
  }
  __syncthreads();
  const int c4n = C >> 2;
  const long total = npix * c4n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % c4n);
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    const f32x4 s4 = *reinterpret_cast<const f32x4*>(sc + c4 * 4), t4 = *reinterpret_cast<const f32x4*>(sh + c4 * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float z = sp_bn_z(v[e], s4[e], t4[e]);
      v[e] = z > 0.f ? z : z * slope;
    }
    reinterpret_cast<f32x4*>(y)[i] = v;
  }
}

// dx = gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat));  workgroup 0 writes dgamma = sum(dz * xhat), dbeta = sum(dz)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean_in, const float* __restrict__ invstd_in,
                                                           const float* __restrict__ part, int nblk, float* __restrict__ dx,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, float slope,
                                                           long npix, int C, int C_real) {
  __shared__ __attribute__((aligned(16))) float s1[SP_MAXC], s2[SP_MAXC], mn[SP_MAXC], iv[SP_MAXC], gm[SP_MAXC], sc[SP_MAXC], sh[SP_MAXC];
  __shared__ float tmp[256];
  sp_sum_parts_wg(part, nblk, 2L * C, C, s1, tmp);
  sp_sum_parts_wg(part + C, nblk, 2L * C, C, s2, tmp);
  for (int c = threadIdx.x; c < C; c += 256) {
    const float a = s1[c], b = s2[c];
    if (blockIdx.x == 0 && c < C_real) {
      dbeta[c] = a;
      dgamma[c] = b;
    }
    s1[c] = a / (float)npix;
    s2[c] = b / (float)npix;
    mn[c] = mean_in[c];
    iv[c] = invstd_in[c];
    gm[c] = c < C_real ? gamma[c] : 0.f;
    sp_bn_coeffs(gm[c], c < C_real ? beta[c] : 0.f, mn[c], iv[c], sc[c], sh[c]);
  }
  __syncthreads();
  const int c4n = C >> 2;
  const long total = npix * c4n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c0 = (int)(i % c4n) * 4;
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 dz = reinterpret_cast<const f32x4*>(dy)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = c0 + e;
      const float xh = (v[e] - mn[c]) * iv[c];
      float d = dz[e];
      if (!(sp_bn_z(v[e], sc[c], sh[c]) > 0.f)) d *= slope;
      dz[e] = gm[c] * iv[c] * (d - s1[c] - xh * s2[c]);
    }
    reinterpret_cast<f32x4*>(dx)[i] = dz;
  }
}

// ---- maps of at most 4 K pixels (SPARNet at 16 images: everything from 16^2 pixels down, 149 of the 187 batch norms of a
// step), ONE launch per direction: a workgroup owns four channels and keeps its pixels (16 float4 per thread, NT = 256 or
// 1024 threads) in registers through both statistics passes and the apply pass -- the same two-pass arithmetic as the partial-sum
// kernels above without their three (two) dependent launches of 7 - 15 us each, no workspace, no cross-workgroup step.
// Sums: per thread in pixel order, lanes by xor-shuffles (32, 16, .., 1), waves in wave order: fixed, independent of timing.
template <int NT>
__device__ __forceinline__ f32x4 sp_block_sum(f32x4 v, float* red /* [NT / 64][4] */) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += __shfl_xor(v[e], o);
  }
  const int wave = threadIdx.x >> 6;
  __syncthreads();  // red is free again
  if ((threadIdx.x & 63) == 0) *reinterpret_cast<f32x4*>(red + wave * 4) = v;
  __syncthreads();
  f32x4 s = *reinterpret_cast<const f32x4*>(red);
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) s += *reinterpret_cast<const f32x4*>(red + w * 4);
  return s;
}

template <int NT, int PX>
__global__ __launch_bounds__(NT) void bn_small_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                          float momentum, float eps, float slope, int npix, int C, int C_real) {
  __shared__ __attribute__((aligned(16))) float red[(NT / 64) * 4];
  const int c0 = blockIdx.x * 4;
  // buffer addressing: a pixel's byte offset is two vector instructions where it is used, not a 64-bit address kept per pixel
  const sisr_rsrc_t rx = sisr_rsrc(x), ry = sisr_rsrc(y);
  auto pix_off = [&](int p) { return (unsigned)(min(p, npix - 1) * C + c0) * 4u; };
  f32x4 v[PX];
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    const int p = threadIdx.x + k * NT;
    v[k] = sisr_keep_if(sisr_buf_load4(rx, pix_off(p), 0u), p < npix);
  }
#pragma unroll
  for (int k = 0; k < PX; ++k) a += v[k];
  const f32x4 mean = sp_block_sum<NT>(a, red) / (float)npix;
  f32x4 q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    if (threadIdx.x + k * NT < npix) {
      const f32x4 d = v[k] - mean;
      q += d * d;
    }
  }
  const f32x4 var = sp_block_sum<NT>(q, red) / (float)npix;
  f32x4 sc, sh;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = c0 + e;
    const float inv = 1.f / sqrtf(var[e] + eps);
    if (threadIdx.x == 0) {
      mean_out[c] = mean[e];
      invstd_out[c] = inv;
      if (running_mean && c < C_real) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean[e];
        const float unbiased = npix > 1 ? var[e] * ((float)npix / (float)(npix - 1)) : var[e];
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
      }
    }
    float s_, t_;
    sp_bn_coeffs(c < C_real ? gamma[c] : 0.f, c < C_real ? beta[c] : 0.f, mean[e], inv, s_, t_);
    sc[e] = s_;
    sh[e] = t_;
  }
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    const int p = threadIdx.x + k * NT;
    if (p < npix) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float z = sp_bn_z(v[k][e], sc[e], sh[e]);
        o[e] = z > 0.f ? z : z * slope;
      }
      sisr_buf_store4(o, ry, pix_off(p), 0u);
    }
  }
}

// KEEP: dy stays in registers too (16 pixels per thread: 128 + 64 registers); else it is read again for the apply pass (32 pixels per
// thread)
template <int NT, int PX, bool KEEP>
__global__ __launch_bounds__(NT) void bn_small_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ mean_in, const float* __restrict__ invstd_in,
                                                          float* __restrict__ dx, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, float slope, int npix, int C, int C_real) {
  __shared__ __attribute__((aligned(16))) float red[(NT / 64) * 4];
  const int c0 = blockIdx.x * 4;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const sisr_rsrc_t rx = sisr_rsrc(x), rdy = sisr_rsrc(dy), rdx = sisr_rsrc(dx);
  auto pix_off = [&](int p) { return (unsigned)(min(p, npix - 1) * C + c0) * 4u; };
  f32x4 v[PX], d[KEEP ? PX : 1];
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    const int p = threadIdx.x + k * NT;
    v[k] = sisr_keep_if(sisr_buf_load4(rx, pix_off(p), 0u), p < npix);
    if (KEEP) d[k] = sisr_keep_if(sisr_buf_load4(rdy, pix_off(p), 0u), p < npix);
  }
  f32x4 m4, i4, g4, sc, sh;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = c0 + e;
    m4[e] = mean_in[c];
    i4[e] = invstd_in[c];
    g4[e] = c < C_real ? gamma[c] : 0.f;
    float s_, t_;
    sp_bn_coeffs(g4[e], c < C_real ? beta[c] : 0.f, m4[e], i4[e], s_, t_);
    sc[e] = s_;
    sh[e] = t_;
  }
  auto masked = [&](f32x4 dz, const f32x4& xv) {  // dy * LeakyReLU'(forward pre-activation), the forward's own sign test
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (!(sp_bn_z(xv[e], sc[e], sh[e]) > 0.f)) dz[e] *= slope;
    return dz;
  };
  f32x4 a = z4, a2 = z4;
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    const int p = threadIdx.x + k * NT;
    if (p < npix) {
      const f32x4 xh = (v[k] - m4) * i4;
      const f32x4 dz = masked(KEEP ? d[k] : sisr_buf_load4(rdy, pix_off(p), 0u), v[k]);
      if (KEEP) d[k] = dz;
      a += dz;
      a2 += dz * xh;
    }
    if (!KEEP && (k & 7) == 7) asm volatile("" ::: "memory");  // eight re-reads in flight, not all 32 (256 registers per wave)
  }
  const f32x4 sa = sp_block_sum<NT>(a, red);
  const f32x4 sb = sp_block_sum<NT>(a2, red);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c0 + e < C_real) {
        dbeta[c0 + e] = sa[e];
        dgamma[c0 + e] = sb[e];
      }
  }
  const f32x4 s1 = sa / (float)npix, s2 = sb / (float)npix;
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    const int p = threadIdx.x + k * NT;
    if (p < npix) {
      const f32x4 dz = KEEP ? d[k] : masked(sisr_buf_load4(rdy, pix_off(p), 0u), v[k]);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (v[k][e] - m4[e]) * i4[e];
        o[e] = g4[e] * i4[e] * (dz[e] - s1[e] - xh * s2[e]);
      }
      sisr_buf_store4(o, rdx, pix_off(p), 0u);
    }
    if (!KEEP && (k & 7) == 7) asm volatile("" ::: "memory");
  }
}

static inline void sp_bn_geometry(long npix, int C, int* nblk, long* chunk) {
  const int rows = 1024 / C;                  // pixel lanes of a workgroup
  long want = (npix + rows * 8 - 1) / (rows * 8);  // at least eight pixels per lane
  int n = (int)(want < 1 ? 1 : (want > SP_MAXBLK ? SP_MAXBLK : want));
  *nblk = n;
  *chunk = (npix + n - 1) / n;
}

extern "C" size_t sisr_bn_workspace_bytes(long npix, int C) {
  if (npix <= 0 || C <= 0 || (C & 63) || C > SP_MAXC) return 0;
  return (size_t)SP_MAXBLK * 2 * C * sizeof(float);
}

extern "C" int sisr_bn_act_fwd(const float* x, float* y, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, float* mean_out, float* invstd_out, long npix, int C, int C_real,
                               int training, float momentum, float eps, float slope, float* workspace,
                               size_t workspace_bytes, void* stream) {
  if (!x || !y || !gamma || !beta || npix <= 0 || C <= 0 || (C & 63) || C > SP_MAXC || C_real <= 0 || C_real > C)
    return SISR_ERR_ARG;
  if (!sisr_aligned16(x) || !sisr_aligned16(y)) return SISR_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const unsigned ga = sp_blocks(npix * (C >> 2), 2048);  // every workgroup first reduces the partials: not too many of them
  if (!training) {
    if (!running_mean || !running_var) return SISR_ERR_ARG;
    hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(ga), dim3(256), 0, st, x, y, gamma, beta, nullptr, nullptr, 0, nullptr,
                       nullptr, running_mean, running_var, 0.f, eps, slope, npix, C, C_real);
    return sisr_check_launch();
  }
  if (!mean_out || !invstd_out || !workspace || workspace_bytes < sisr_bn_workspace_bytes(npix, C)) return SISR_ERR_ARG;
  if (npix <= 256 * 16) {
    hipLaunchKernelGGL((bn_small_fwd_kernel<256, 16>), dim3(C / 4), dim3(256), 0, st, x, y, gamma, beta, running_mean, running_var,
                       mean_out, invstd_out, momentum, eps, slope, (int)npix, C, C_real);
    return sisr_check_launch();
  }
  int nblk;
  long chunk;
  sp_bn_geometry(npix, C, &nblk, &chunk);
  float* psum = workspace;
  float* psq = workspace + (size_t)SP_MAXBLK * C;
  hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(nblk), dim3(256), 0, st, x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                     psum, npix, C, C_real, 1.f, chunk);
  hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(nblk), dim3(256), 0, st, x, nullptr, nullptr, nullptr, nullptr, nullptr, psum,
                     psq, npix, C, C_real, 1.f, chunk);
  hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(ga), dim3(256), 0, st, x, y, gamma, beta, psum, psq, nblk, mean_out,
                     invstd_out, running_mean, running_var, momentum, eps, slope, npix, C, C_real);
  return sisr_check_launch();
}

extern "C" int sisr_bn_act_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean,
                               const float* invstd, float* dx, float* dgamma, float* dbeta, long npix, int C, int C_real,
                               float slope, float* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !dy || !gamma || !beta || !mean || !invstd || !dx || !dgamma || !dbeta || !workspace || npix <= 0 || C <= 0 ||
      (C & 63) || C > SP_MAXC || C_real <= 0 || C_real > C || workspace_bytes < sisr_bn_workspace_bytes(npix, C))
    return SISR_ERR_ARG;
  if (!sisr_aligned16(x) || !sisr_aligned16(dy) || !sisr_aligned16(dx)) return SISR_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (npix <= 256 * 16) {
    hipLaunchKernelGGL((bn_small_bwd_kernel<256, 16, true>), dim3(C / 4), dim3(256), 0, st, x, dy, gamma, beta, mean, invstd, dx, dgamma,
                       dbeta, slope, (int)npix, C, C_real);
    return sisr_check_launch();
  }
  int nblk;
  long chunk;
  sp_bn_geometry(npix, C, &nblk, &chunk);
  hipLaunchKernelGGL(bn_partial_kernel<2>, dim3(nblk), dim3(256), 0, st, x, dy, gamma, beta, mean, invstd, nullptr, workspace,
                     npix, C, C_real, slope, chunk);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(sp_blocks(npix * (C >> 2), 2048)), dim3(256), 0, st, x, dy, gamma, beta, mean, invstd,
                     workspace, nblk, dx, dgamma, dbeta, slope, npix, C, C_real);
  return sisr_check_launch();
}

// ------------------------------------------------------------------------------------------------ spatial attention product
// y = idn + x * a,  a = sigmoid(logits[p][0]);  att[p] = a.  One thread per 16-byte piece; the C / 4 threads of a pixel are
// consecutive lanes of one wave (C / 4 = 16 or 32 ...: a power of two up to 64).
__global__ __launch_bounds__(256) void spar_combine_fwd_kernel(const f32x4* __restrict__ x, const float* __restrict__ logits,
                                                               const f32x4* __restrict__ idn, f32x4* __restrict__ y,
                                                               float* __restrict__ att, int c4n, int Cl, long total) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long p = i / c4n;
    const float a = 1.f / (1.f + expf(-logits[p * Cl]));
    f32x4 v = x[i] * a;
    if (idn) v += idn[i];
    y[i] = v;
    if (i % c4n == 0) att[p] = a;
  }
}

// dx = dy * a;  dlogits[p][0] = (sum_c dy * x) * a * (1 - a), every other channel of dlogits 0.  total is a multiple of c4n
// and the grid-stride a multiple of 64, so the c4n lanes of a pixel stay together in one wave.
__global__ __launch_bounds__(256) void spar_combine_bwd_kernel(const f32x4* __restrict__ dy, const f32x4* __restrict__ x,
                                                               const float* __restrict__ att, f32x4* __restrict__ dx,
                                                               f32x4* __restrict__ dlogits, int c4n, int cl4n, long total) {
  for (long i0 = (long)blockIdx.x * 256; i0 < total; i0 += (long)gridDim.x * 256) {
    const long i = i0 + threadIdx.x;
    const bool on = i < total;
    const long p = on ? i / c4n : 0;
    const int c4 = on ? (int)(i % c4n) : 0;
    const float a = on ? att[p] : 0.f;
    f32x4 g = {0.f, 0.f, 0.f, 0.f}, v = g;
    if (on) {
      g = dy[i];
      v = x[i];
      dx[i] = g * a;
    }
    float s = (g[0] * v[0] + g[1] * v[1]) + (g[2] * v[2] + g[3] * v[3]);
    for (int o = 1; o < c4n; o <<= 1) s += __shfl_xor(s, o);
    if (on) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      for (int q = c4; q < cl4n; q += c4n) {
        f32x4 d = z;
        if (q == 0) d[0] = s * a * (1.f - a);
        dlogits[p * cl4n + q] = d;
      }
    }
  }
}

extern "C" int sisr_pad_reflect_up(const float* x, float* y, int B, int H, int W, int C, int up, int adjoint, void* stream) {
  // adjoint == 0: x (B, H, W, C) -> y (B, up H + 2, up W + 2, C);  adjoint == 1: x is the padded gradient, y (B, H, W, C)
  if (!x || !y || B <= 0 || H < 2 || W < 2 || C <= 0 || (C & 3) || (up != 1 && up != 2)) return SISR_ERR_ARG;
  if (!sisr_aligned16(x) || !sisr_aligned16(y)) return SISR_ERR_ALIGN;
  const int c4n = C >> 2;
  if (!adjoint) {
    const long total = (long)B * (up * H + 2) * (up * W + 2) * c4n;
    hipLaunchKernelGGL(pad_reflect_up_kernel, dim3(sp_blocks(total)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4*>(x), reinterpret_cast<f32x4*>(y), H, W, c4n, up, total);
  } else {
    const long total = (long)B * H * W * c4n;
    hipLaunchKernelGGL(pad_reflect_up_bwd_kernel, dim3(sp_blocks(total)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4*>(x), reinterpret_cast<f32x4*>(y), H, W, c4n, up, total);
  }
  return sisr_check_launch();
}

extern "C" int sisr_crop_stride(const float* src, float* dst, int B, int Hf, int Wf, int C, int stride, int embed,
                                void* stream) {
  // embed == 0: src (B, Hf, Wf, C) -> dst (B, Ho, Wo, C), Ho = (Hf - 3) / stride + 1;  embed == 1: the adjoint (src is the
  // small map, dst the zero-filled large one)
  if (!src || !dst || B <= 0 || Hf < 3 || Wf < 3 || C <= 0 || (C & 3) || (stride != 1 && stride != 2)) return SISR_ERR_ARG;
  if (!sisr_aligned16(src) || !sisr_aligned16(dst)) return SISR_ERR_ALIGN;
  const int Ho = (Hf - 3) / stride + 1, Wo = (Wf - 3) / stride + 1, c4n = C >> 2;
  const long total = embed ? (long)B * Hf * Wf * c4n : (long)B * Ho * Wo * c4n;
  hipLaunchKernelGGL(crop_stride_kernel, dim3(sp_blocks(total)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const f32x4*>(src), reinterpret_cast<f32x4*>(dst), Hf, Wf, Ho, Wo, stride, c4n, total,
                     embed);
  return sisr_check_launch();
}

// nearest-neighbour upsampling by an integer factor (ref: SRMD_blocks.py:58-63 nn.Upsample(scale_factor = 2 | 3 | 4, 'nearest')
// in front of SRMD's 'upconv' tail): y[b][i][j] = x[b][i / up][j / up]; adjoint: dx = sum of the up x up replicas, row-major.
__global__ __launch_bounds__(256) void nearest_up_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int H, int W,
                                                         int c4n, int up, long total, int adjoint) {
  const int Ho = H * up, Wo = W * up;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % c4n);
    long t = i / c4n;
    if (!adjoint) {
      const int j = (int)(t % Wo);
      t /= Wo;
      const int r = (int)(t % Ho);
      const long b = t / Ho;
      dst[i] = src[((b * H + r / up) * W + j / up) * c4n + c4];
    } else {
      const int w = (int)(t % W);
      t /= W;
      const int h = (int)(t % H);
      const long b = t / H;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int a = 0; a < up; ++a)
        for (int c = 0; c < up; ++c) acc += src[((b * Ho + h * up + a) * Wo + w * up + c) * c4n + c4];
      dst[i] = acc;
    }
  }
}

extern "C" int sisr_nearest_up(const float* src, float* dst, int B, int H, int W, int C, int up, int adjoint, void* stream) {
  // adjoint == 0: src (B, H, W, C) -> dst (B, up H, up W, C);  adjoint != 0: src is the gradient of the large map, dst (B, H, W, C)
  if (!src || !dst || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || up < 1 || up > 4) return SISR_ERR_ARG;
  if (!sisr_aligned16(src) || !sisr_aligned16(dst)) return SISR_ERR_ALIGN;
  const int c4n = C >> 2;
  const long total = adjoint ? (long)B * H * W * c4n : (long)B * H * up * W * up * c4n;
  hipLaunchKernelGGL(nearest_up_kernel, dim3(sp_blocks(total)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const f32x4*>(src), reinterpret_cast<f32x4*>(dst), H, W, c4n, up, total, adjoint);
  return sisr_check_launch();
}

static inline bool sp_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

extern "C" int sisr_spar_combine_fwd(const float* x, const float* logits, const float* identity, float* y, float* att,
                                     long npix, int C, int C_logits, void* stream) {
  if (!x || !logits || !y || !att || npix <= 0 || C <= 0 || (C & 3) || C_logits <= 0) return SISR_ERR_ARG;
  if (!sisr_aligned16(x) || !sisr_aligned16(y) || (identity && !sisr_aligned16(identity))) return SISR_ERR_ALIGN;
  const int c4n = C >> 2;
  const long total = npix * c4n;
  hipLaunchKernelGGL(spar_combine_fwd_kernel, dim3(sp_blocks(total)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const f32x4*>(x), logits, reinterpret_cast<const f32x4*>(identity),
                     reinterpret_cast<f32x4*>(y), att, c4n, C_logits, total);
  return sisr_check_launch();
}

extern "C" int sisr_spar_combine_bwd(const float* dy, const float* x, const float* att, float* dx, float* dlogits, long npix,
                                     int C, int C_logits, void* stream) {
  if (!dy || !x || !att || !dx || !dlogits || npix <= 0 || C <= 0 || (C & 3) || C_logits <= 0 || (C_logits & 3))
    return SISR_ERR_ARG;
  const int c4n = C >> 2;
  if (!sp_pow2(c4n) || c4n > 64) return SISR_ERR_UNSUPPORTED;
  if (!sisr_aligned16(dy) || !sisr_aligned16(x) || !sisr_aligned16(dx) || !sisr_aligned16(dlogits)) return SISR_ERR_ALIGN;
  const long total = npix * c4n;
  hipLaunchKernelGGL(spar_combine_bwd_kernel, dim3(sp_blocks(total)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const f32x4*>(dy), reinterpret_cast<const f32x4*>(x), att, reinterpret_cast<f32x4*>(dx),
                     reinterpret_cast<f32x4*>(dlogits), c4n, C_logits >> 2, total);
  return sisr_check_launch();
}

// ================================================================================================ non-default ConvLayer options
// ref: SPARNet/blocks.py:17-33 (norm_type 'in' = InstanceNorm2d(affine), 'gn' = GroupNorm(32, C), 'pixel' = F.normalize(x, p = 2,
// dim = 1)), :50-64 (relu_type 'prelu' = PReLU(C), 'selu'), :147-151 (att_name 'spar3d': one attention map per channel).  The
// reference's defaults (batch norm, LeakyReLU, one attention channel) are the fused kernels above; these options are plain
// HBM-bound passes, one launch per direction, written for correctness first (maps of this network are small).
//
// Group statistics (instance norm: groups of one channel).  x, y: [B][HW][C]; group g of sample b = channels [g cg, (g + 1) cg)
// over all HW pixels; C_real real channels (a multiple of cg), the padded ones are written as zero.  One workgroup per
// (group, sample): mean, then centred squares (two passes, as torch), then the affine apply.  mean / invstd: [B][C_real / cg].
__device__ __forceinline__ float sp_wg_sum(float v, float* red /* [4] */) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void group_norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ mean_out, float* __restrict__ invstd_out, long hw,
                                                             int C, int C_real, int cg, float eps) {
  __shared__ float red[4];
  const int g = blockIdx.x, b = blockIdx.y, groups = C_real / cg;
  const float* xb = x + (long)b * hw * C + g * cg;
  float* yb = y + (long)b * hw * C + g * cg;
  const long n = hw * cg;
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) s += xb[(i / cg) * C + (i % cg)];
  const float mean = sp_wg_sum(s, red) / (float)n;
  float q = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) {
    const float d = xb[(i / cg) * C + (i % cg)] - mean;
    q += d * d;
  }
  const float inv = 1.f / sqrtf(sp_wg_sum(q, red) / (float)n + eps);
  if (threadIdx.x == 0) {
    mean_out[b * groups + g] = mean;
    invstd_out[b * groups + g] = inv;
  }
  for (long i = threadIdx.x; i < n; i += 256) {
    const int j = (int)(i % cg), c = g * cg + j;
    const long o = (i / cg) * C + j;
    yb[o] = (xb[o] - mean) * inv * gamma[c] + beta[c];
  }
}

// dx = invstd (dxhat - mean_g(dxhat) - xhat mean_g(dxhat xhat)), dxhat = dy gamma[c];  per-sample partials dgamma_b[b][c] =
// sum_hw dy xhat, dbeta_b[b][c] = sum_hw dy (the caller adds them over the batch with sisr_sum_partials).
__global__ __launch_bounds__(256) void group_norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                             const float* __restrict__ invstd_in, float* __restrict__ dx,
                                                             float* __restrict__ dgamma_b, float* __restrict__ dbeta_b, long hw,
                                                             int C, int C_real, int cg) {
  __shared__ float red[4];
  const int g = blockIdx.x, b = blockIdx.y, groups = C_real / cg;
  const long base = (long)b * hw * C + g * cg;
  const float mean = mean_in[b * groups + g], inv = invstd_in[b * groups + g];
  const long n = hw * cg;
  float s1 = 0.f, s2 = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) {
    const int j = (int)(i % cg);
    const long o = base + (i / cg) * C + j;
    const float xh = (x[o] - mean) * inv, dh = dy[o] * gamma[g * cg + j];
    s1 += dh;
    s2 += dh * xh;
  }
  s1 = sp_wg_sum(s1, red) / (float)n;
  s2 = sp_wg_sum(s2, red) / (float)n;
  for (long i = threadIdx.x; i < n; i += 256) {
    const int j = (int)(i % cg);
    const long o = base + (i / cg) * C + j;
    const float xh = (x[o] - mean) * inv, dh = dy[o] * gamma[g * cg + j];
    dx[o] = inv * (dh - s1 - xh * s2);
  }
  for (int j = 0; j < cg; ++j) {  // per channel of the group: sums over this sample's pixels
    float a = 0.f, c2 = 0.f;
    for (long p = threadIdx.x; p < hw; p += 256) {
      const long o = base + p * C + j;
      const float d = dy[o];
      a += d * ((x[o] - mean) * inv);
      c2 += d;
    }
    a = sp_wg_sum(a, red);
    c2 = sp_wg_sum(c2, red);
    if (threadIdx.x == 0) {
      dgamma_b[(long)b * C_real + g * cg + j] = a;
      dbeta_b[(long)b * C_real + g * cg + j] = c2;
    }
  }
}

// channels >= C_real of y (dx): zero
__global__ __launch_bounds__(256) void zero_pad_channels_kernel(float* __restrict__ y, long npix, int C, int C_real) {
  const int w = C - C_real;
  const long total = npix * w;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) y[(i / w) * C + C_real + (i % w)] = 0.f;
}

extern "C" int sisr_group_norm_fwd(const float* x, float* y, const float* gamma, const float* beta, float* mean_out,
                                   float* invstd_out, int B, long hw, int C, int C_real, int cg, float eps, void* stream) {
  if (!x || !y || !gamma || !beta || !mean_out || !invstd_out || B <= 0 || hw <= 0 || C <= 0 || C_real <= 0 || C_real > C || cg <= 0 ||
      C_real % cg || B > 65535)
    return SISR_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(group_norm_fwd_kernel, dim3(C_real / cg, B), dim3(256), 0, st, x, y, gamma, beta, mean_out, invstd_out, hw, C,
                     C_real, cg, eps);
  if (C_real < C)
    hipLaunchKernelGGL(zero_pad_channels_kernel, dim3(sp_blocks((long)B * hw * (C - C_real))), dim3(256), 0, st, y, (long)B * hw, C, C_real);
  return sisr_check_launch();
}

extern "C" int sisr_group_norm_bwd(const float* x, const float* dy, const float* gamma, const float* mean, const float* invstd,
                                   float* dx, float* dgamma_b, float* dbeta_b, int B, long hw, int C, int C_real, int cg,
                                   void* stream) {
  if (!x || !dy || !gamma || !mean || !invstd || !dx || !dgamma_b || !dbeta_b || B <= 0 || hw <= 0 || C <= 0 || C_real <= 0 ||
      C_real > C || cg <= 0 || C_real % cg || B > 65535)
    return SISR_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(group_norm_bwd_kernel, dim3(C_real / cg, B), dim3(256), 0, st, x, dy, gamma, mean, invstd, dx, dgamma_b, dbeta_b,
                     hw, C, C_real, cg);
  if (C_real < C)
    hipLaunchKernelGGL(zero_pad_channels_kernel, dim3(sp_blocks((long)B * hw * (C - C_real))), dim3(256), 0, st, dx, (long)B * hw, C, C_real);
  return sisr_check_launch();
}

// Pixel norm: y = x / max(||x||_2 over the pixel's channels, 1e-12) (F.normalize).  The C / 4 lanes of a pixel are consecutive
// lanes of one wave (C / 4 a power of two <= 64); padded channels are zero and stay zero.  backward != 0: x, dy -> dx =
// (dy - y sum_c(dy y)) / max(||x||, eps)  (dy / eps where the norm is below eps).
__global__ __launch_bounds__(256) void pixel_norm_kernel(const f32x4* __restrict__ x, const f32x4* __restrict__ dy, f32x4* __restrict__ out,
                                                         int c4n, long total, int backward) {
  for (long i0 = (long)blockIdx.x * 256; i0 < total; i0 += (long)gridDim.x * 256) {
    const long i = i0 + threadIdx.x;
    const bool on = i < total;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 v = on ? x[i] : z;
    float n2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    for (int o = 1; o < c4n; o <<= 1) n2 += __shfl_xor(n2, o);
    const float nrm = sqrtf(n2), den = fmaxf(nrm, 1e-12f);
    if (!backward) {
      if (on) out[i] = v / den;
    } else {
      const f32x4 g = on ? dy[i] : z;
      const f32x4 yv = v / den;
      float dot = (g[0] * yv[0] + g[1] * yv[1]) + (g[2] * yv[2] + g[3] * yv[3]);
      for (int o = 1; o < c4n; o <<= 1) dot += __shfl_xor(dot, o);
      if (on) out[i] = nrm > 1e-12f ? (g - yv * dot) / den : g / den;
    }
  }
}

extern "C" int sisr_pixel_norm(const float* x, const float* dy, float* out, long npix, int C, int backward, void* stream) {
  if (!x || !out || (backward && !dy) || npix <= 0 || C <= 0 || (C & 3)) return SISR_ERR_ARG;
  const int c4n = C >> 2;
  if (!sp_pow2(c4n) || c4n > 64) return SISR_ERR_UNSUPPORTED;
  if (!sisr_aligned16(x) || !sisr_aligned16(out) || !sisr_aligned16(dy)) return SISR_ERR_ALIGN;
  const long total = npix * c4n;
  unsigned blocks = sp_blocks(total);
  hipLaunchKernelGGL(pixel_norm_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const f32x4*>(x),
                     reinterpret_cast<const f32x4*>(dy), reinterpret_cast<f32x4*>(out), c4n, total, backward);
  return sisr_check_launch();
}

// Activations with a parameter or a second branch.  mode 0: PReLU (slope a[c], c < C_real; padded channels pass zeros), mode 1:
// SELU.  backward: dx, and for PReLU also dyx = dy * min(x, 0), whose per-channel sum is the slope's gradient.
#define SP_SELU_ALPHA 1.6732632423543772848170429916717f
#define SP_SELU_SCALE 1.0507009873554804934193349852946f
__global__ __launch_bounds__(256) void act_kernel(const f32x4* __restrict__ x, const f32x4* __restrict__ dy, const float* __restrict__ a,
                                                  f32x4* __restrict__ out, f32x4* __restrict__ dyx, int c4n, int C_real, long total,
                                                  int mode, int backward) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c0 = (int)(i % c4n) * 4;
    const f32x4 v = x[i];
    f32x4 o, m = {0.f, 0.f, 0.f, 0.f};
    const f32x4 g = backward ? dy[i] : m;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xv = v[e];
      if (mode == 0) {
        const float s = c0 + e < C_real ? a[c0 + e] : 0.f;
        if (!backward) o[e] = xv > 0.f ? xv : s * xv;
        else {
          o[e] = xv > 0.f ? g[e] : s * g[e];
          m[e] = xv > 0.f ? 0.f : g[e] * xv;
        }
      } else {
        if (!backward) o[e] = SP_SELU_SCALE * (xv > 0.f ? xv : SP_SELU_ALPHA * (expf(xv) - 1.f));
        else o[e] = g[e] * SP_SELU_SCALE * (xv > 0.f ? 1.f : SP_SELU_ALPHA * expf(xv));
      }
    }
    out[i] = o;
    if (backward && mode == 0) dyx[i] = m;
  }
}

extern "C" int sisr_act(const float* x, const float* dy, const float* slope, float* out, float* dyx, long npix, int C, int C_real,
                        int mode, int backward, void* stream) {
  if (!x || !out || npix <= 0 || C <= 0 || (C & 3) || (mode != 0 && mode != 1) || (backward && !dy) || (mode == 0 && !slope) ||
      (mode == 0 && backward && !dyx) || C_real <= 0 || C_real > C)
    return SISR_ERR_ARG;
  if (!sisr_aligned16(x) || !sisr_aligned16(out) || !sisr_aligned16(dy) || !sisr_aligned16(dyx)) return SISR_ERR_ALIGN;
  const long total = npix * (C >> 2);
  hipLaunchKernelGGL(act_kernel, dim3(sp_blocks(total)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const f32x4*>(x),
                     reinterpret_cast<const f32x4*>(dy), slope, reinterpret_cast<f32x4*>(out), reinterpret_cast<f32x4*>(dyx), C >> 2,
                     C_real, total, mode, backward);
  return sisr_check_launch();
}

// 'spar3d': y = identity + x * sigmoid(logits), one logit per element.  backward: dx = dy a, dlogits = dy x a (1 - a).
__global__ __launch_bounds__(256) void spar3d_kernel(const f32x4* __restrict__ x, const f32x4* __restrict__ logits,
                                                     const f32x4* __restrict__ idn_or_dy, f32x4* __restrict__ out0, f32x4* __restrict__ out1,
                                                     long total, int backward) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const f32x4 v = x[i], l = logits[i];
    f32x4 a;
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = 1.f / (1.f + expf(-l[e]));
    if (!backward) {
      f32x4 r = v * a;
      if (idn_or_dy) r += idn_or_dy[i];
      out0[i] = r;
    } else {
      const f32x4 g = idn_or_dy[i];
      out0[i] = g * a;
      out1[i] = g * v * a * (1.f - a);
    }
  }
}

extern "C" int sisr_spar3d(const float* x, const float* logits, const float* identity_or_dy, float* out0, float* out1, long n,
                           int backward, void* stream) {
  if (!x || !logits || !out0 || n <= 0 || (n & 3) || (backward && (!identity_or_dy || !out1))) return SISR_ERR_ARG;
  if (!sisr_aligned16(x) || !sisr_aligned16(logits) || !sisr_aligned16(identity_or_dy) || !sisr_aligned16(out0) || !sisr_aligned16(out1))
    return SISR_ERR_ALIGN;
  hipLaunchKernelGGL(spar3d_kernel, dim3(sp_blocks(n >> 2)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const f32x4*>(x),
                     reinterpret_cast<const f32x4*>(logits), reinterpret_cast<const f32x4*>(identity_or_dy),
                     reinterpret_cast<f32x4*>(out0), reinterpret_cast<f32x4*>(out1), n >> 2, backward);
  return sisr_check_launch();
}
