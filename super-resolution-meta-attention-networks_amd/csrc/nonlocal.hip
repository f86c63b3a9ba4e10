// The 1x1 projections, pooling and output projection around SAN's non-local attention.
// ref: advanced/SAN_blocks.py:104-148 (_NonLocalBlockND.forward: theta / phi / g = 1x1 convs 64 -> 8, phi and g max-pooled
// 2x2, y = softmax(theta^T phi) g [csrc/san.hip], z = W(y) + x with W a 1x1 conv 8 -> 64), :305-336 (Nonlocal_CA: the block
// applied to the four quadrants independently).  Maps are channels-last [npix][64]; an attention domain is a rectangle of
// the map (the whole map, or a quadrant), its rows [domain][position][8].
//   nl_project_fwd     proj[p][0:24] = (theta | phi | g)(x[p])          M = pixels, N = 24 -> 32, K = 64   fp32 MFMA
//   nl_project_dgrad   dx[p] = dproj[p] . Wp + dz[p] (the skip)         M = pixels, N = 64,       K = 24   fp32 MFMA
//   nl_project_wgrad   dWp[n][c] = sum_p dproj[p][n] x[p][c], db        M = 24 -> 32, N = 64, K = pixels   fp32 MFMA
//   nl_split_pool      proj rectangle -> theta rows, 2x2-max-pooled phi / g rows (floor mode) and the reverse scatter
//   nl_output          z = y . W^T + b + x per pixel (K = 8: VALU), and its backward: dy, ordered partial sums of dW, db
// MFMA operand maps as in conv3x3_mfma.hip: A lane = (row lane & 31, k = lane >> 5), B lane = (column lane & 31, k),
// D register r = row (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of column lane & 31.  K runs as 8-channel groups: a lane loads
// the float4 of channels 8j + 4k .. + 3 and feeds four instructions, so every load is 16 B wide.
#include "sisr_common.h"

#define NL_C 64
#define NL_CI 8
#define NL_P 24  // theta | phi | g

struct NlDomains {
  int B, H, W, y0, x0, hq, wq, nqy, nqx;
};
__device__ __forceinline__ long nl_pixel(const NlDomains& d, int dom, int ly, int lx) {
  const int ix = dom % d.nqx, r = dom / d.nqx, iy = r % d.nqy, b = r / d.nqy;
  return ((long)b * d.H + d.y0 + iy * d.hq + ly) * d.W + d.x0 + ix * d.wq + lx;
}
static bool nl_domains_ok(const NlDomains& d) {
  return d.B > 0 && d.H > 0 && d.W > 0 && d.hq >= 2 && d.wq >= 2 && d.nqy > 0 && d.nqx > 0 && d.y0 >= 0 && d.x0 >= 0 &&
         d.y0 + d.nqy * d.hq <= d.H && d.x0 + d.nqx * d.wq <= d.W;
}
static unsigned nl_blocks(long n) {
  long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 65535 ? 65535 : b));
}

// ------------------------------------------------------------------ projections 64 -> 24
__device__ __forceinline__ const float* nl_wrow(const float* w0, const float* w1, const float* w2, int n) {
  return n < 8 ? w0 + n * NL_C : (n < 16 ? w1 + (n - 8) * NL_C : w2 + (n - 16) * NL_C);
}

__global__ __launch_bounds__(256) void nl_project_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w0,
                                                             const float* __restrict__ b0, const float* __restrict__ w1,
                                                             const float* __restrict__ b1, const float* __restrict__ w2,
                                                             const float* __restrict__ b2, float* __restrict__ proj, long npix) {
  const int lane = threadIdx.x & 63, li = lane & 31, kk = lane >> 5;
  f32x4 bf[8];
  float bv = 0.f;
  if (li < NL_P) {
    const float* wr = nl_wrow(w0, w1, w2, li);
#pragma unroll
    for (int j = 0; j < 8; ++j) bf[j] = *reinterpret_cast<const f32x4*>(wr + 8 * j + 4 * kk);
    bv = li < 8 ? b0[li] : (li < 16 ? b1[li - 8] : b2[li - 16]);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) bf[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const long waves = (long)gridDim.x * 4, tiles = (npix + 31) / 32;
  for (long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6); t < tiles; t += waves) {
    const long p0 = t * 32;
    const float* xp = x + min(p0 + li, npix - 1) * NL_C + 4 * kk;
    f32x4 a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = *reinterpret_cast<const f32x4*>(xp + 8 * j);
    f32x16 acc = {0};
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][e], bf[j][e], acc, 0, 0, 0);
    if (li < NL_P) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long p = p0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (p < npix) proj[p * NL_P + li] = acc[r] + bv;
      }
    }
  }
}

// dx[p][c] = sum_n dproj[p][n] Wp[n][c] + dz[p][c]
__global__ __launch_bounds__(256) void nl_project_dgrad_kernel(const float* __restrict__ dproj, const float* __restrict__ dz,
                                                               const float* __restrict__ w0, const float* __restrict__ w1,
                                                               const float* __restrict__ w2, float* __restrict__ dx, long npix) {
  const int lane = threadIdx.x & 63, li = lane & 31, kk = lane >> 5;
  float bf[3][4][2];  // B[k][c][e] of K-group j = Wp[8 j + 4 k + e][c], c = li + 32 nt
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float* wr = nl_wrow(w0, w1, w2, 8 * j + 4 * kk + e);
      bf[j][e][0] = wr[li];
      bf[j][e][1] = wr[li + 32];
    }
  const long waves = (long)gridDim.x * 4, tiles = (npix + 31) / 32;
  for (long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6); t < tiles; t += waves) {
    const long p0 = t * 32;
    const float* ap = dproj + min(p0 + li, npix - 1) * NL_P + 4 * kk;
    f32x4 a[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) a[j] = *reinterpret_cast<const f32x4*>(ap + 8 * j);
    f32x16 acc0 = {0}, acc1 = {0};
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][e], bf[j][e][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][e], bf[j][e][1], acc1, 0, 0, 0);
      }
    float s0[16], s1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long p = min(p0 + (r & 3) + 8 * (r >> 2) + 4 * kk, npix - 1);
      s0[r] = dz[p * NL_C + li];
      s1[r] = dz[p * NL_C + 32 + li];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long p = p0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
      if (p < npix) {
        dx[p * NL_C + li] = acc0[r] + s0[r];
        dx[p * NL_C + 32 + li] = acc1[r] + s1[r];
      }
    }
  }
}

// per-wave partial sums part[wave][33][64]: rows 0..23 = dWp[n][c] over the wave's pixels, row 32 = db[n] (first 24 entries)
#define NLW_ROWS 33
__global__ __launch_bounds__(256) void nl_project_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dproj,
                                                               float* __restrict__ part, long npix, long span) {
  const int lane = threadIdx.x & 63, li = lane & 31, kk = lane >> 5;
  const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long p_begin = wv * span, p_end = min(p_begin + span, npix);  // span is even
  f32x16 acc0 = {0}, acc1 = {0};
  float bsum = 0.f;
  const bool row_ok = li < NL_P;
  for (long p0 = p_begin; p0 < p_end; p0 += 16) {
    float a[8], b0[8], b1[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const long p = p0 + 2 * q + kk;
      const bool ok = p < p_end;
      const long pc = ok ? p : p_end - 1;
      const float av = row_ok ? dproj[pc * NL_P + li] : 0.f;
      a[q] = ok ? av : 0.f;
      b0[q] = x[pc * NL_C + li];
      b1[q] = x[pc * NL_C + 32 + li];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      bsum += a[q];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b0[q], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b1[q], acc1, 0, 0, 0);
    }
  }
  float* out = part + wv * (NLW_ROWS * NL_C);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = (r & 3) + 8 * (r >> 2) + 4 * kk;
    out[n * NL_C + li] = acc0[r];
    out[n * NL_C + 32 + li] = acc1[r];
  }
  bsum += __shfl_xor(bsum, 32);
  if (kk == 0) {
    out[32 * NL_C + li] = bsum;
    out[32 * NL_C + 32 + li] = 0.f;
  }
}

static int nl_wgrad_blocks(long npix) {
  long b = (npix + 2047) / 2048;  // >= 512 pixels per wave
  return (int)(b < 1 ? 1 : (b > 256 ? 256 : b));
}

extern "C" int sisr_nl_project_fwd(const float* x, const float* w_theta, const float* b_theta, const float* w_phi,
                                   const float* b_phi, const float* w_g, const float* b_g, float* proj, long npix, void* stream) {
  if (!x || !w_theta || !b_theta || !w_phi || !b_phi || !w_g || !b_g || !proj || npix <= 0) return SISR_ERR_ARG;
  if (!sisr_aligned16(x) || !sisr_aligned16(w_theta) || !sisr_aligned16(w_phi) || !sisr_aligned16(w_g)) return SISR_ERR_ALIGN;
  long blocks = (npix + 127) / 128;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(nl_project_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, w_theta, b_theta, w_phi,
                     b_phi, w_g, b_g, proj, npix);
  return sisr_check_launch();
}

extern "C" int sisr_nl_project_bwd_parts(long npix) { return npix > 0 ? nl_wgrad_blocks(npix) * 4 : 0; }

// dx = dproj . Wp + dz; part[sisr_nl_project_bwd_parts(npix)][33][64]: ordered partial sums (rows 0..23 dWp, row 32 db)
extern "C" int sisr_nl_project_bwd(const float* x, const float* dproj, const float* dz, const float* w_theta, const float* w_phi,
                                   const float* w_g, float* dx, float* part, long npix, void* stream) {
  if (!x || !dproj || !dz || !w_theta || !w_phi || !w_g || !dx || !part || npix <= 0) return SISR_ERR_ARG;
  if (!sisr_aligned16(dproj)) return SISR_ERR_ALIGN;
  long blocks = (npix + 127) / 128;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(nl_project_dgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dproj, dz, w_theta, w_phi,
                     w_g, dx, npix);
  int rc = sisr_check_launch();
  if (rc) return rc;
  const int wb = nl_wgrad_blocks(npix);
  long span = (npix + wb * 4 - 1) / (wb * 4);
  span = (span + 1) & ~1L;
  hipLaunchKernelGGL(nl_project_wgrad_kernel, dim3(wb), dim3(256), 0, (hipStream_t)stream, x, dproj, part, npix, span);
  return sisr_check_launch();
}

// ------------------------------------------------------------------ rows of an attention domain, 2x2 max pooling
__global__ __launch_bounds__(256) void nl_split_pool_fwd_kernel(const float* __restrict__ proj, float* __restrict__ theta,
                                                                float* __restrict__ phi, float* __restrict__ g, NlDomains d) {
  const int npos = d.hq * d.wq, hp = d.hq >> 1, wp = d.wq >> 1, nwin = hp * wp;
  const long nd = (long)d.B * d.nqy * d.nqx;
  const long n_theta = nd * npos * 2, total = n_theta + nd * nwin * 4;  // float4 items
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    if (i < n_theta) {
      const int h = (int)(i & 1);
      const long r = i >> 1;
      const int pos = (int)(r % npos), dom = (int)(r / npos);
      const long p = nl_pixel(d, dom, pos / d.wq, pos % d.wq);
      *reinterpret_cast<f32x4*>(theta + r * NL_CI + 4 * h) = *reinterpret_cast<const f32x4*>(proj + p * NL_P + 4 * h);
    } else {
      const long k = i - n_theta;
      const int q = (int)(k & 3);  // float4 q of the 16 pooled channels: 0, 1 = phi, 2, 3 = g
      const long r = k >> 2;
      const int win = (int)(r % nwin), dom = (int)(r / nwin);
      const int wy = win / wp, wx = win - wy * wp;
      f32x4 m = *reinterpret_cast<const f32x4*>(proj + nl_pixel(d, dom, 2 * wy, 2 * wx) * NL_P + 8 + 4 * q);
#pragma unroll
      for (int s = 1; s < 4; ++s) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(proj + nl_pixel(d, dom, 2 * wy + (s >> 1), 2 * wx + (s & 1)) * NL_P + 8 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
      }
      float* dst = (q < 2 ? phi : g) + r * NL_CI + 4 * (q & 1);
      *reinterpret_cast<f32x4*>(dst) = m;
    }
  }
}

// dproj of every pixel of the domains: theta part copied, pooled parts routed to the window's first maximum (scan order
// (0,0), (0,1), (1,0), (1,1), like MaxPool2d's backward); pixels outside the floor-mode windows get zero
__global__ __launch_bounds__(256) void nl_split_pool_bwd_kernel(const float* __restrict__ proj, const float* __restrict__ dtheta,
                                                                const float* __restrict__ dphi, const float* __restrict__ dg,
                                                                float* __restrict__ dproj, NlDomains d) {
  const int npos = d.hq * d.wq, hp = d.hq >> 1, wp = d.wq >> 1, nwin = hp * wp;
  const long nd = (long)d.B * d.nqy * d.nqx;
  const long total = nd * npos * 6;  // float4 items: 2 theta + 4 pooled per pixel
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int q = (int)(i % 6);
    const long r = i / 6;
    const int pos = (int)(r % npos), dom = (int)(r / npos);
    const int ly = pos / d.wq, lx = pos - ly * d.wq;
    const long p = nl_pixel(d, dom, ly, lx);
    f32x4 out = {0.f, 0.f, 0.f, 0.f};
    if (q < 2) {
      out = *reinterpret_cast<const f32x4*>(dtheta + r * NL_CI + 4 * q);
    } else if ((ly >> 1) < hp && (lx >> 1) < wp) {
      const int pq = q - 2, wy = ly >> 1, wx = lx >> 1, me = (ly & 1) * 2 + (lx & 1);
      f32x4 m = *reinterpret_cast<const f32x4*>(proj + nl_pixel(d, dom, 2 * wy, 2 * wx) * NL_P + 8 + 4 * pq);
      int arg[4] = {0, 0, 0, 0};
#pragma unroll
      for (int s = 1; s < 4; ++s) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(proj + nl_pixel(d, dom, 2 * wy + (s >> 1), 2 * wx + (s & 1)) * NL_P + 8 + 4 * pq);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (v[e] > m[e]) {
            m[e] = v[e];
            arg[e] = s;
          }
      }
      const long w = (long)dom * nwin + wy * wp + wx;
      const f32x4 gsrc = *reinterpret_cast<const f32x4*>((pq < 2 ? dphi : dg) + w * NL_CI + 4 * (pq & 1));
#pragma unroll
      for (int e = 0; e < 4; ++e) out[e] = arg[e] == me ? gsrc[e] : 0.f;
    }
    *reinterpret_cast<f32x4*>(dproj + p * NL_P + 4 * q) = out;
  }
}

static int nl_unpack(const int* dom, NlDomains* d) {
  if (!dom) return SISR_ERR_ARG;
  *d = NlDomains{dom[0], dom[1], dom[2], dom[3], dom[4], dom[5], dom[6], dom[7], dom[8]};
  return nl_domains_ok(*d) ? 0 : SISR_ERR_ARG;
}

// domains: 9 ints (host memory) B, H, W, y0, x0, hq, wq, nqy, nqx: B * nqy * nqx rectangles of hq x wq positions
extern "C" int sisr_nl_split_pool_fwd(const float* proj, float* theta, float* phi, float* g, const int* domains, void* stream) {
  NlDomains d;
  if (!proj || !theta || !phi || !g) return SISR_ERR_ARG;
  if (int rc = nl_unpack(domains, &d)) return rc;
  if (!sisr_aligned16(proj) || !sisr_aligned16(theta) || !sisr_aligned16(phi) || !sisr_aligned16(g)) return SISR_ERR_ALIGN;
  const long nd = (long)d.B * d.nqy * d.nqx;
  const long total = nd * d.hq * d.wq * 2 + nd * (d.hq >> 1) * (d.wq >> 1) * 4;
  hipLaunchKernelGGL(nl_split_pool_fwd_kernel, dim3(nl_blocks(total)), dim3(256), 0, (hipStream_t)stream, proj, theta, phi, g, d);
  return sisr_check_launch();
}

extern "C" int sisr_nl_split_pool_bwd(const float* proj, const float* dtheta, const float* dphi, const float* dg, float* dproj,
                                      const int* domains, void* stream) {
  NlDomains d;
  if (!proj || !dtheta || !dphi || !dg || !dproj) return SISR_ERR_ARG;
  if (int rc = nl_unpack(domains, &d)) return rc;
  if (!sisr_aligned16(proj) || !sisr_aligned16(dtheta) || !sisr_aligned16(dphi) || !sisr_aligned16(dg) || !sisr_aligned16(dproj))
    return SISR_ERR_ALIGN;
  const long total = (long)d.B * d.nqy * d.nqx * d.hq * d.wq * 6;
  hipLaunchKernelGGL(nl_split_pool_bwd_kernel, dim3(nl_blocks(total)), dim3(256), 0, (hipStream_t)stream, proj, dtheta, dphi, dg,
                     dproj, d);
  return sisr_check_launch();
}

// ------------------------------------------------------------------ output projection 8 -> 64 + skip
// 16 lanes per pixel, a lane owns 4 output channels and their 4 x 8 weights in registers
__global__ __launch_bounds__(256) void nl_output_fwd_kernel(const float* __restrict__ y, const float* __restrict__ x,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ z, NlDomains d) {
  const int c4 = threadIdx.x & 15;
  float wr[4][8];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int k = 0; k < 8; ++k) wr[e][k] = w[(4 * c4 + e) * NL_CI + k];
  const f32x4 bv = {bias[4 * c4], bias[4 * c4 + 1], bias[4 * c4 + 2], bias[4 * c4 + 3]};
  const int npos = d.hq * d.wq;
  const long total = (long)d.B * d.nqy * d.nqx * npos;
  for (long r = (long)blockIdx.x * 16 + (threadIdx.x >> 4); r < total; r += (long)gridDim.x * 16) {
    const int pos = (int)(r % npos), dom = (int)(r / npos);
    const long p = nl_pixel(d, dom, pos / d.wq, pos % d.wq);
    const f32x4 y0 = *reinterpret_cast<const f32x4*>(y + r * NL_CI), y1 = *reinterpret_cast<const f32x4*>(y + r * NL_CI + 4);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + p * NL_C + 4 * c4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) s += y0[k] * wr[e][k];
#pragma unroll
      for (int k = 0; k < 4; ++k) s += y1[k] * wr[e][4 + k];
      o[e] = s + bv[e] + xv[e];
    }
    *reinterpret_cast<f32x4*>(z + p * NL_C + 4 * c4) = o;
  }
}

// dy[r][k] = sum_c dz[p][c] W[c][k]; per-block partial sums part[block][64 * 8 + 64]: dW[c][k] = sum dz[p][c] y[r][k], db[c]
#define NLO_PART (NL_C * NL_CI + NL_C)
__global__ __launch_bounds__(256) void nl_output_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                                                            const float* __restrict__ w, float* __restrict__ dy,
                                                            float* __restrict__ part, NlDomains d) {
  __shared__ float red[16][16 * 9 + 1];  // [pixel slot][lane c4][8 weight sums + 1 bias sum] of one channel e
  const int c4 = threadIdx.x & 15, slot = threadIdx.x >> 4;
  float wr[4][8];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int k = 0; k < 8; ++k) wr[e][k] = w[(4 * c4 + e) * NL_CI + k];
  float aw[4][8], ab[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ab[e] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) aw[e][k] = 0.f;
  }
  const int npos = d.hq * d.wq;
  const long total = (long)d.B * d.nqy * d.nqx * npos;
  const long rounds = (total + (long)gridDim.x * 16 - 1) / ((long)gridDim.x * 16);
  for (long it = 0; it < rounds; ++it) {
    const long r0 = (it * gridDim.x + blockIdx.x) * 16 + slot;
    const bool live = r0 < total;
    const long r = live ? r0 : total - 1;
    const int pos = (int)(r % npos), dom = (int)(r / npos);
    const long p = nl_pixel(d, dom, pos / d.wq, pos % d.wq);
    f32x4 g4 = *reinterpret_cast<const f32x4*>(dz + p * NL_C + 4 * c4);
    if (!live) g4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 y0 = *reinterpret_cast<const f32x4*>(y + r * NL_CI), y1 = *reinterpret_cast<const f32x4*>(y + r * NL_CI + 4);
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = g4[0] * wr[0][k] + g4[1] * wr[1][k] + g4[2] * wr[2][k] + g4[3] * wr[3][k];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1)
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += __shfl_xor(s[k], o);
    if (live && c4 < 8) dy[r * NL_CI + c4] = s[c4 & 7];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ab[e] += g4[e];
#pragma unroll
      for (int k = 0; k < 4; ++k) aw[e][k] += g4[e] * y0[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) aw[e][4 + k] += g4[e] * y1[k];
    }
  }
  // the 16 pixel slots of the block, in slot order
  float* out = part + (long)blockIdx.x * NLO_PART;
  for (int e = 0; e < 4; ++e) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) red[slot][c4 * 9 + k] = aw[e][k];
    red[slot][c4 * 9 + 8] = ab[e];
    __syncthreads();
    if (slot == 0) {
      for (int k = 0; k < 9; ++k) {
        float t = 0.f;
        for (int sl = 0; sl < 16; ++sl) t += red[sl][c4 * 9 + k];
        if (k < 8) out[(4 * c4 + e) * NL_CI + k] = t;
        else out[NL_C * NL_CI + 4 * c4 + e] = t;
      }
    }
  }
}

static int nl_out_blocks(long rows) {
  long b = (rows + 16 * 32 - 1) / (16 * 32);  // >= 32 pixels per slot
  return (int)(b < 1 ? 1 : (b > 256 ? 256 : b));
}

extern "C" int sisr_nl_output_fwd(const float* y, const float* x, const float* w, const float* bias, float* z, const int* domains,
                                  void* stream) {
  NlDomains d;
  if (!y || !x || !w || !bias || !z) return SISR_ERR_ARG;
  if (int rc = nl_unpack(domains, &d)) return rc;
  if (!sisr_aligned16(y) || !sisr_aligned16(x) || !sisr_aligned16(z)) return SISR_ERR_ALIGN;
  const long rows = (long)d.B * d.nqy * d.nqx * d.hq * d.wq;
  long blocks = (rows + 15) / 16;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(nl_output_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, x, w, bias, z, d);
  return sisr_check_launch();
}

extern "C" int sisr_nl_output_bwd_parts(const int* domains) {
  NlDomains d;
  if (nl_unpack(domains, &d)) return 0;
  return nl_out_blocks((long)d.B * d.nqy * d.nqx * d.hq * d.wq);
}

// part: [sisr_nl_output_bwd_parts(domains)][64 * 8 + 64] ordered partial sums (dW [64][8], then db [64])
extern "C" int sisr_nl_output_bwd(const float* dz, const float* y, const float* w, float* dy, float* part, const int* domains,
                                  void* stream) {
  NlDomains d;
  if (!dz || !y || !w || !dy || !part) return SISR_ERR_ARG;
  if (int rc = nl_unpack(domains, &d)) return rc;
  if (!sisr_aligned16(dz) || !sisr_aligned16(y)) return SISR_ERR_ALIGN;
  const int blocks = nl_out_blocks((long)d.B * d.nqy * d.nqx * d.hq * d.wq);
  hipLaunchKernelGGL(nl_output_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dz, y, w, dy, part, d);
  return sisr_check_launch();
}
