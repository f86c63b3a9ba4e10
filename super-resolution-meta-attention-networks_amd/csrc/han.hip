// HAN attention modules, fp32: layer attention (LAM) and channel-spatial attention (CSAM).
//
// LAM  (ref: advanced/HAN_blocks.py:7-37): X = (B, N, K) with K = C*H*W flattened per layer map;
//        E = X X^T (B,N,N);  A = softmax_j(max_j E_ij - E_ij);  out = gamma * A X + X.
//      The K = 1M-long dot products are HBM-bound skinny GEMMs: one streaming pass builds all N(N+1)/2
//      Gram entries from N float4 loads per thread (ordered two-stage reduction), one pass applies A.
//      Because every map uses the same NHWC layout the flattening order is irrelevant to E and A, so the
//      stack stays [B][N][H][W][64] and the following 3x3 conv reads it as 64-channel chunks (no concat).
// CSAM (ref: advanced/HAN_blocks.py:40-76): att = sigmoid(Conv3d_{1->1,k3,p1}(x as a (C,H,W) volume));
//        out = x * (gamma * att) + x.  27 MAC/element stencil; the channel taps come from the neighbouring
//      lanes' float4 (channels-last), the spatial taps from nine coalesced row reads.
#include "sisr_common.h"

#define LAM_PARTS 128
#define LAM_MAXN 16

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// part[b][blockIdx.x][i*N + j] = sum over this block's K-slice of P[b][i][k] * X[b][j][k]
// SYM: P == X, only j >= i computed (and mirrored by the consumer).
template <int N, bool SYM>
__global__ __launch_bounds__(256) void lam_gram_kernel(const float* __restrict__ P, const float* __restrict__ X,
                                                       float* __restrict__ part, long k4) {
  __shared__ float red[4][N * N];
  const int b = blockIdx.y;
  const f32x4* xp = reinterpret_cast<const f32x4*>(X) + (long)b * N * k4;
  const f32x4* pp = reinterpret_cast<const f32x4*>(P) + (long)b * N * k4;
  float acc[N * N];
#pragma unroll
  for (int i = 0; i < N * N; ++i) acc[i] = 0.f;
  for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < k4; k += (long)gridDim.x * 256) {
    f32x4 xv[N], pv[N];
#pragma unroll
    for (int i = 0; i < N; ++i) xv[i] = xp[(long)i * k4 + k];
    if (!SYM) {
#pragma unroll
      for (int i = 0; i < N; ++i) pv[i] = pp[(long)i * k4 + k];
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int j = (SYM ? i : 0); j < N; ++j) {
        const f32x4 m = (SYM ? xv[i] : pv[i]) * xv[j];
        acc[i * N + j] += (m[0] + m[1]) + (m[2] + m[3]);
      }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = (SYM ? i : 0); j < N; ++j) {
      const float s = wsum(acc[i * N + j]);
      if (lane == 0) red[wv][i * N + j] = s;
    }
  __syncthreads();
  for (int e = threadIdx.x; e < N * N; e += 256) {
    const int i = e / N, j = e - i * N;
    if (!SYM || j >= i)
      part[((long)b * gridDim.x + blockIdx.x) * N * N + e] = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
  }
}

// One block per sample: E from the partials (mirrored), attention rows.  attn: [B][N][N]
__global__ void lam_softmax_kernel(const float* __restrict__ part, int parts, int N, float* __restrict__ attn) {
  __shared__ float E[LAM_MAXN * LAM_MAXN];
  const int b = blockIdx.x;
  for (int e = threadIdx.x; e < N * N; e += blockDim.x) {
    const int i = e / N, j = e - i * N;
    const int src = j >= i ? e : j * N + i;
    float s = 0.f;
    for (int k = 0; k < parts; ++k) s += part[((long)b * parts + k) * N * N + src];
    E[e] = s;
  }
  __syncthreads();
  if (threadIdx.x < N) {
    const int i = threadIdx.x;
    float mx = E[i * N];
    for (int j = 1; j < N; ++j) mx = fmaxf(mx, E[i * N + j]);
    float top = 0.f;  // max_j (mx - E_ij)
    for (int j = 0; j < N; ++j) top = fmaxf(top, mx - E[i * N + j]);
    float den = 0.f;
    for (int j = 0; j < N; ++j) den += expf((mx - E[i * N + j]) - top);
    for (int j = 0; j < N; ++j) attn[((long)b * N + i) * N + j] = expf((mx - E[i * N + j]) - top) / den;
  }
}

// y[b][i] = sum_j C1[b][i][j] * U[b][j] + (V ? sum_j C2[b][i][j] * V[b][j] : 0)
template <int N, bool TWO>
__global__ __launch_bounds__(256) void lam_apply_kernel(const float* __restrict__ U, const float* __restrict__ C1,
                                                        const float* __restrict__ V, const float* __restrict__ C2,
                                                        float* __restrict__ Y, long k4) {
  __shared__ float c1[N * N], c2[N * N];
  const int b = blockIdx.y;
  for (int e = threadIdx.x; e < N * N; e += 256) {
    c1[e] = C1[(long)b * N * N + e];
    if (TWO) c2[e] = C2[(long)b * N * N + e];
  }
  __syncthreads();
  const f32x4* up = reinterpret_cast<const f32x4*>(U) + (long)b * N * k4;
  const f32x4* vp = reinterpret_cast<const f32x4*>(V) + (long)b * N * k4;
  f32x4* yp = reinterpret_cast<f32x4*>(Y) + (long)b * N * k4;
  for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < k4; k += (long)gridDim.x * 256) {
    // memory clobber: the N*N (x2) coefficients are re-read from LDS (broadcast) every iteration; without
    // it hipcc hoists all of them into VGPRs and spills (up to 776 registers at N = 11)
    asm volatile("" ::: "memory");
    f32x4 uv[N], vv[N];
#pragma unroll
    for (int i = 0; i < N; ++i) uv[i] = up[(long)i * k4 + k];
    if (TWO) {
#pragma unroll
      for (int i = 0; i < N; ++i) vv[i] = vp[(long)i * k4 + k];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < N; ++j) {
        s += c1[i * N + j] * uv[j];
        if (TWO) s += c2[i * N + j] * vv[j];
      }
      yp[(long)i * k4 + k] = s;
    }
  }
}

// coef[b] = gamma*A[b] + I   (forward application matrix)
__global__ void lam_fwd_coef_kernel(const float* __restrict__ attn, const float* __restrict__ gamma_p, int N,
                                    float* __restrict__ coef) {
  const int b = blockIdx.x;
  const float gamma = *gamma_p;
  for (int e = threadIdx.x; e < N * N; e += blockDim.x)
    coef[(long)b * N * N + e] = gamma * attn[(long)b * N * N + e] + ((e / N) == (e % N) ? 1.f : 0.f);
}

// Backward small step, one block (loops over b so dgamma is summed in batch order).
//   G2[i][j] = sum_k dO[i][k] X[j][k];   dgamma += sum_ij A_ij G2_ij;   dA = gamma*G2
//   dE'_ij = A_ij (dA_ij - sum_l dA_il A_il);  dE = -dE'  (the row-max term cancels: softmax rows sum to 1)
//   C1 = I + gamma*A^T (applied to dO),  C2 = dE + dE^T (applied to X)
__global__ void lam_bwd_small_kernel(const float* __restrict__ part, int parts, const float* __restrict__ attn,
                                     const float* __restrict__ gamma_p, int B, int N, float* __restrict__ C1,
                                     float* __restrict__ C2, float* __restrict__ dgamma) {
  __shared__ float G[LAM_MAXN * LAM_MAXN], dE[LAM_MAXN * LAM_MAXN], rowdot[LAM_MAXN];
  __shared__ float dgs;
  const float gamma = *gamma_p;
  if (threadIdx.x == 0) dgs = 0.f;
  for (int b = 0; b < B; ++b) {
    const float* A = attn + (long)b * N * N;
    __syncthreads();
    for (int e = threadIdx.x; e < N * N; e += blockDim.x) {
      float s = 0.f;
      for (int k = 0; k < parts; ++k) s += part[((long)b * parts + k) * N * N + e];
      G[e] = s;
    }
    __syncthreads();
    if (threadIdx.x < N) {
      const int i = threadIdx.x;
      float d = 0.f;
      for (int l = 0; l < N; ++l) d += G[i * N + l] * A[i * N + l];
      rowdot[i] = d;  // sum_l G_il A_il  (dA = gamma*G)
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float d = 0.f;
      for (int i = 0; i < N; ++i) d += rowdot[i];
      dgs += d;
    }
    for (int e = threadIdx.x; e < N * N; e += blockDim.x) {
      const int i = e / N;
      dE[e] = -(A[e] * gamma * (G[e] - rowdot[i]));
    }
    __syncthreads();
    for (int e = threadIdx.x; e < N * N; e += blockDim.x) {
      const int i = e / N, j = e - i * N;
      C2[(long)b * N * N + e] = dE[e] + dE[j * N + i];
      C1[(long)b * N * N + e] = gamma * A[j * N + i] + (i == j ? 1.f : 0.f);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) *dgamma = dgs;
}

// ------------------------------------------------------------------------------------------ CSAM
// 16 lanes per pixel (lane = float4 of channels), 16 pixels per 256-thread block iteration.
// MODE 0: forward  y = x*(1 + gamma*sigmoid(z)),  z = bias + conv3d(x)
// MODE 1: backward A: recompute z; dz = dy*x*gamma*s*(1-s) -> ws; dx = dy*(1+gamma*s);
//         block partials of dgamma = sum dy*x*s and dbias = sum dz
// MODE 2: backward B: dx += conv3d^T(dz);  block partials of dw[27] = sum dz * x(shifted)
template <int MODE>
__global__ __launch_bounds__(256) void csam_kernel(const float* __restrict__ x, const float* __restrict__ w27,
                                                   const float* __restrict__ bias_p,
                                                   const float* __restrict__ gamma_p, const float* __restrict__ dy,
                                                   float* __restrict__ out, float* __restrict__ dz,
                                                   float* __restrict__ part, int B, int H, int W) {
  __shared__ float red[4][28];
  const float bias = *bias_p, gamma = *gamma_p;
  const int c4 = threadIdx.x & 15;
  const long npix = (long)B * H * W;
  const long pend = (npix + 15) & ~15L;
  float wl[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) wl[i] = w27[i];  // [dc][dh][dw]
  float pacc[MODE == 2 ? 27 : 2];
#pragma unroll
  for (int i = 0; i < (MODE == 2 ? 27 : 2); ++i) pacc[i] = 0.f;
  for (long pix0 = (long)blockIdx.x * 16 + (threadIdx.x >> 4); pix0 < pend; pix0 += (long)gridDim.x * 16) {
    const bool live = pix0 < npix;
    const long pix = live ? pix0 : npix - 1;
    const long b = pix / ((long)H * W);
    const long r = pix - b * H * W;
    const int h = (int)(r / W), w = (int)(r - (long)h * W);
    const float* src = (MODE == 2) ? dz : x;  // the volume the 27-tap stencil runs over
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 ctr_x = reinterpret_cast<const f32x4*>(x)[pix * 16 + c4];
    f32x4 ctr_dz = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 2) ctr_dz = reinterpret_cast<const f32x4*>(dz)[pix * 16 + c4];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int gh = h + t / 3 - 1, gw = w + t % 3 - 1;
      const bool ok = gh >= 0 && gh < H && gw >= 0 && gw < W;
      const long np_ = (b * H + min(max(gh, 0), H - 1)) * W + min(max(gw, 0), W - 1);
      const f32x4 v = sisr_keep_if(reinterpret_cast<const f32x4*>(src)[np_ * 16 + c4], ok);
      float lft = __shfl_up(v[3], 1), rgt = __shfl_down(v[0], 1);
      if (c4 == 0) lft = 0.f;
      if (c4 == 15) rgt = 0.f;
      const float ext[6] = {lft, v[0], v[1], v[2], v[3], rgt};
      if (MODE != 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) z[e] += wl[t] * ext[e] + wl[9 + t] * ext[e + 1] + wl[18 + t] * ext[e + 2];
      } else {
        // transposed stencil: dx[c] += w[dc][t'] * dz[c - dc + 1] at the mirrored spatial tap t' = 8 - t
#pragma unroll
        for (int e = 0; e < 4; ++e)
          z[e] += wl[8 - t] * ext[e + 2] + wl[9 + 8 - t] * ext[e + 1] + wl[18 + 8 - t] * ext[e];
        // dw[dc][t] = sum dz[c] * x[c + dc - 1] at spatial tap t: needs x's neighbourhood as well
        const f32x4 xv = sisr_keep_if(reinterpret_cast<const f32x4*>(x)[np_ * 16 + c4], ok);
        float xl = __shfl_up(xv[3], 1), xr = __shfl_down(xv[0], 1);
        if (c4 == 0) xl = 0.f;
        if (c4 == 15) xr = 0.f;
        const float xe[6] = {xl, xv[0], xv[1], xv[2], xv[3], xr};
        if (live) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            pacc[t] += ctr_dz[e] * xe[e];
            pacc[9 + t] += ctr_dz[e] * xe[e + 1];
            pacc[18 + t] += ctr_dz[e] * xe[e + 2];
          }
        }
      }
    }
    if (MODE == 0) {
      if (live) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = ctr_x[e] * (1.f + gamma / (1.f + expf(-(z[e] + bias))));
        reinterpret_cast<f32x4*>(out)[pix * 16 + c4] = o;
      }
    } else if (MODE == 1) {
      if (live) {
        const f32x4 g = reinterpret_cast<const f32x4*>(dy)[pix * 16 + c4];
        f32x4 o, d;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float s = 1.f / (1.f + expf(-(z[e] + bias)));
          o[e] = g[e] * (1.f + gamma * s);
          d[e] = g[e] * ctr_x[e] * gamma * s * (1.f - s);
          pacc[0] += g[e] * ctr_x[e] * s;
          pacc[1] += d[e];
        }
        reinterpret_cast<f32x4*>(out)[pix * 16 + c4] = o;
        reinterpret_cast<f32x4*>(dz)[pix * 16 + c4] = d;
      }
    } else {
      if (live) {
        f32x4 o = reinterpret_cast<f32x4*>(out)[pix * 16 + c4];
        reinterpret_cast<f32x4*>(out)[pix * 16 + c4] = o + z;
      }
    }
  }
  if (MODE != 0) {
    constexpr int NP = (MODE == 2) ? 27 : 2;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const float s = wsum(pacc[i]);
      if (lane == 0) red[wv][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < NP)
      part[(long)blockIdx.x * NP + threadIdx.x] =
          ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
  }
}

// out[i] = sum_k part[k][n + ...]: final ordered sums for dgamma, dbias (from pass A) and dw[27] (pass B)
__global__ void csam_finish_kernel(const float* __restrict__ pa, int na, const float* __restrict__ pb, int nb,
                                   float* __restrict__ dw27, float* __restrict__ dbias, float* __restrict__ dgamma) {
  const int i = threadIdx.x;
  if (i < 27) {
    float s = 0.f;
    for (int k = 0; k < nb; ++k) s += pb[(long)k * 27 + i];
    dw27[i] = s;
  } else if (i < 29) {
    float s = 0.f;
    for (int k = 0; k < na; ++k) s += pa[(long)k * 2 + (i - 27)];
    if (i == 27) *dgamma = s; else *dbias = s;
  }
}

// ------------------------------------------------------------------------------------------ C ABI
// Statement-level dispatch on the compile-time map count (hipLaunchKernelGGL is a do{}while(0) statement;
// the variadic tail re-joins the commas of the template argument lists).
#define LAM_DISPATCH(N_, ...)                                  \
  switch (N_) {                                                \
    case 2: { constexpr int NN = 2; __VA_ARGS__; } break;      \
    case 3: { constexpr int NN = 3; __VA_ARGS__; } break;      \
    case 4: { constexpr int NN = 4; __VA_ARGS__; } break;      \
    case 5: { constexpr int NN = 5; __VA_ARGS__; } break;      \
    case 6: { constexpr int NN = 6; __VA_ARGS__; } break;      \
    case 8: { constexpr int NN = 8; __VA_ARGS__; } break;      \
    case 11: { constexpr int NN = 11; __VA_ARGS__; } break;    \
    default: return SISR_ERR_UNSUPPORTED;                      \
  }

static int lam_parts(long k4) {
  long p = (k4 + 255) / 256;
  if (p > LAM_PARTS) p = LAM_PARTS;
  if (p < 1) p = 1;
  return (int)p;
}

// workspace: gram partials [B][parts][N*N] + two coefficient matrices [B][N*N]
extern "C" size_t sisr_lam_workspace_bytes(int B, int N, long chw) {
  if (B <= 0 || N < 2 || N > LAM_MAXN || chw <= 0 || (chw & 3)) return 0;
  return ((size_t)B * lam_parts(chw / 4) * N * N + 2 * (size_t)B * N * N) * sizeof(float);
}

extern "C" int sisr_lam_fwd(const float* x, const float* gamma, float* y, float* attn, float* workspace, int B, int N,
                            long chw, void* stream) {
  if (!x || !gamma || !y || !attn || !workspace || B <= 0 || chw <= 0 || (chw & 3)) return SISR_ERR_ARG;
  if (!sisr_aligned16(x) || !sisr_aligned16(y)) return SISR_ERR_ALIGN;
  const long k4 = chw / 4;
  const int parts = lam_parts(k4);
  float* part = workspace;
  float* coef = workspace + (size_t)B * parts * N * N;
  hipStream_t st = (hipStream_t)stream;
  LAM_DISPATCH(N, hipLaunchKernelGGL((lam_gram_kernel<NN, true>), dim3(parts, B), dim3(256), 0, st, x, x, part, k4));
  hipLaunchKernelGGL(lam_softmax_kernel, dim3(B), dim3(256), 0, st, part, parts, N, attn);
  hipLaunchKernelGGL(lam_fwd_coef_kernel, dim3(B), dim3(256), 0, st, attn, gamma, N, coef);
  const int ablocks = (int)((k4 + 255) / 256 < 1024 ? (k4 + 255) / 256 : 1024);
  LAM_DISPATCH(N, hipLaunchKernelGGL((lam_apply_kernel<NN, false>), dim3(ablocks, B), dim3(256), 0, st, x, coef,
                                     (const float*)nullptr, (const float*)nullptr, y, k4));
  return sisr_check_launch();
}

extern "C" int sisr_lam_bwd(const float* x, const float* attn, const float* gamma, const float* dy, float* dx, float* dgamma,
                            float* workspace, int B, int N, long chw, void* stream) {
  if (!x || !attn || !gamma || !dy || !dx || !dgamma || !workspace || B <= 0 || chw <= 0 || (chw & 3)) return SISR_ERR_ARG;
  if (!sisr_aligned16(x) || !sisr_aligned16(dy) || !sisr_aligned16(dx)) return SISR_ERR_ALIGN;
  const long k4 = chw / 4;
  const int parts = lam_parts(k4);
  float* part = workspace;
  float* c1 = workspace + (size_t)B * parts * N * N;
  float* c2 = c1 + (size_t)B * N * N;
  hipStream_t st = (hipStream_t)stream;
  LAM_DISPATCH(N, hipLaunchKernelGGL((lam_gram_kernel<NN, false>), dim3(parts, B), dim3(256), 0, st, dy, x, part, k4));
  hipLaunchKernelGGL(lam_bwd_small_kernel, dim3(1), dim3(256), 0, st, part, parts, attn, gamma, B, N, c1, c2, dgamma);
  const int ablocks = (int)((k4 + 255) / 256 < 1024 ? (k4 + 255) / 256 : 1024);
  LAM_DISPATCH(N, hipLaunchKernelGGL((lam_apply_kernel<NN, true>), dim3(ablocks, B), dim3(256), 0, st, dy, c1, x, c2, dx,
                                     k4));
  return sisr_check_launch();
}

static int csam_blocks(long npix) {
  long nb = (npix + 15) / 16;
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  return (int)nb;
}

extern "C" int sisr_csam_fwd(const float* x, const float* w27, const float* bias, const float* gamma, float* y, int B, int H, int W,
                             int C, void* stream) {
  if (!x || !w27 || !bias || !gamma || !y || B <= 0 || H <= 0 || W <= 0) return SISR_ERR_ARG;
  if (C != 64) return SISR_ERR_UNSUPPORTED;
  if (!sisr_aligned16(x) || !sisr_aligned16(y)) return SISR_ERR_ALIGN;
  hipLaunchKernelGGL(csam_kernel<0>, dim3(csam_blocks((long)B * H * W)), dim3(256), 0, (hipStream_t)stream, x, w27, bias,
                     gamma, (const float*)nullptr, y, (float*)nullptr, (float*)nullptr, B, H, W);
  return sisr_check_launch();
}

// workspace: dz [B*H*W*64] + pass-A partials [blocks][2] + pass-B partials [blocks][27]
extern "C" size_t sisr_csam_bwd_workspace_bytes(int B, int H, int W, int C) {
  if (B <= 0 || H <= 0 || W <= 0 || C != 64) return 0;
  const long npix = (long)B * H * W;
  return ((size_t)npix * 64 + (size_t)csam_blocks(npix) * 32) * sizeof(float);
}

extern "C" int sisr_csam_bwd(const float* x, const float* w27, const float* bias, const float* gamma, const float* dy, float* dx,
                             float* dw27, float* dbias, float* dgamma, float* workspace, int B, int H, int W, int C,
                             void* stream) {
  if (!x || !w27 || !bias || !gamma || !dy || !dx || !dw27 || !dbias || !dgamma || !workspace || B <= 0 || H <= 0 || W <= 0)
    return SISR_ERR_ARG;
  if (C != 64) return SISR_ERR_UNSUPPORTED;
  if (!sisr_aligned16(x) || !sisr_aligned16(dy) || !sisr_aligned16(dx) || !sisr_aligned16(workspace)) return SISR_ERR_ALIGN;
  const long npix = (long)B * H * W;
  const int nb = csam_blocks(npix);
  float* dz = workspace;
  float* pa = workspace + (size_t)npix * 64;
  float* pb = pa + (size_t)nb * 2;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(csam_kernel<1>, dim3(nb), dim3(256), 0, st, x, w27, bias, gamma, dy, dx, dz, pa, B, H, W);
  hipLaunchKernelGGL(csam_kernel<2>, dim3(nb), dim3(256), 0, st, x, w27, bias, gamma, dy, dx, dz, pb, B, H, W);
  hipLaunchKernelGGL(csam_finish_kernel, dim3(1), dim3(64), 0, st, pa, nb, pb, nb, dw27, dbias, dgamma);
  return sisr_check_launch();
}
