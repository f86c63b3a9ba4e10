// 3x3 same-convolutions with a 3-channel side (the RGB ends of the network), fp32 VALU.
//
//   conv3x3_cin3  : planar NCHW (B,3,H,W) -> chunked NHWC (B,H,W,64k).   Head conv forward
//                   (ref: advanced/architectures.py:141 head = default_conv(3, 64)) and the input
//                   gradient of the tail conv.
//   conv3x3_cout3 : chunked NHWC (B,H,W,64k) -> planar NCHW (B,3,H,W).   Tail conv forward
//                   (ref: advanced/architectures.py:150-152 tail[1] = default_conv(64, 3)) and the
//                   input gradient of the head conv.
//   corr3x3_c3    : out[a][tap][c] = sum_{b,p} P[b][a][p+off(tap)] * Q[b][p][c], P planar 3-channel,
//                   Q chunked NHWC -- the weight gradient of both of the above (+ bias sums).
// K = 27 (or N = 3): too thin for the matrix cores as they stand -- except 64 -> 3, which conv_rgb_out.h widens to 27
// columns (output row, channel, kw) and runs on them; the rest are HBM-bound streaming kernels
// (the 512x512x64 map at the tail is the largest tensor in the network), written for coalesced
// 256-B pixel rows.  Weights are addressed through generic (so, si, flip) strides so the same kernel
// serves a forward pass (OIHW as stored) and a gradient pass (roles swapped, taps flipped).
#include "sisr_common.h"
#include "conv_rgb_out.h"

static View view_from(const int64_t* v) {
  View r;
  r.sB = v[0];
  r.sH = v[1];
  r.sW = v[2];
  r.chi = v[3];
  r.clo = v[4];
  r.cdiv = (int)v[5];
  return r;
}

// ------------------------------------------------------------------ 3 -> 64k
struct Cin3Params {
  const float* x;  // [B][3][H][W]
  float* y;
  View yv;
  const float* w;
  const float* bias;
  long so, si;
  int flip, B, H, W, cout;
};

// thread = (pixel, 16 consecutive output channels): the 27 input taps are loaded once per 16 outputs, the
// weights come from LDS as float4 (4 distinct addresses per wave), the store is 64 B per lane / 256 B per pixel.
__global__ __launch_bounds__(256) void conv3x3_cin3_kernel(Cin3Params p) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [27][cout]
  for (int i = threadIdx.x; i < 27 * p.cout; i += 256) {
    const int k = i / p.cout, co = i - k * p.cout;
    const int ci = k / 9, t = k - ci * 9;
    wl[i] = p.w[(long)co * p.so + (long)ci * p.si + (p.flip ? 8 - t : t)];
  }
  __syncthreads();
  const int groups = p.cout >> 4;  // 16-channel groups per pixel
  const long hw = (long)p.H * p.W;
  const long total = (long)p.B * hw * groups;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long pix = i / groups;
    const int cg = (int)(i - pix * groups);
    const long b = pix / hw;
    const long r = pix - b * hw;
    const int h = (int)(r / p.W), w = (int)(r - (long)h * p.W);
    const int co = cg * 16;
    f32x4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      acc[u] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + co + 4 * u) : (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* xb = p.x + b * 3 * hw;
    float xv[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const int ci = k / 9, t = k % 9;
      const int gh = h + t / 3 - 1, gw = w + t % 3 - 1;
      const bool ok = gh >= 0 && gh < p.H && gw >= 0 && gw < p.W;
      const int ch_ = min(max(gh, 0), p.H - 1), cw_ = min(max(gw, 0), p.W - 1);
      const float v = xb[ci * hw + (long)ch_ * p.W + cw_];
      xv[k] = ok ? v : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const f32x4* wr = reinterpret_cast<const f32x4*>(wl + k * p.cout + co);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] += xv[k] * wr[u];
    }
    float* yo = p.y + b * p.yv.sB + (long)h * p.yv.sH + (long)w * p.yv.sW + p.yv.chunk(co >> 6) + (co & 63);
#pragma unroll
    for (int u = 0; u < 4; ++u) reinterpret_cast<f32x4*>(yo)[u] = acc[u];
  }
}

// ------------------------------------------------------------------ 64k -> 3
struct Cout3Params {
  const float* x;
  View xv;
  float* y;  // [B][3][H][W]
  const float* w;
  const float* bias;  // [3]
  long so, si;
  int flip, B, H, W, cin_chunks;
};

// 16 lanes per pixel; a lane owns input channels [4*c4, 4*c4+4) of each chunk and keeps its 108 weights
// (9 taps x 4 ch x 3 outputs) in VGPRs across the whole pixel loop; the 16 partial dots are combined with
// four xor-shuffles.  Per pixel: nine coalesced 256-B row reads, 108 FMAs per lane.
__global__ __launch_bounds__(256) void conv3x3_cout3_kernel(Cout3Params p) {
  const int c4 = threadIdx.x & 15;
  const long hw = (long)p.H * p.W;
  const long npix = (long)p.B * hw;
  const long gstride = (long)gridDim.x * 16;
  const long pend = (npix + 15) & ~15L;
  for (int c = 0; c < p.cin_chunks; ++c) {
    float wr[9][4][3];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int o = 0; o < 3; ++o)
          wr[t][e][o] = p.w[(long)o * p.so + ((long)c * 64 + c4 * 4 + e) * p.si + (p.flip ? 8 - t : t)];
    for (long pix0 = (long)blockIdx.x * 16 + (threadIdx.x >> 4); pix0 < pend; pix0 += gstride) {
      const bool live = pix0 < npix;
      const long pix = live ? pix0 : npix - 1;
      const long b = pix / hw;
      const long r = pix - b * hw;
      const int h = (int)(r / p.W), w = (int)(r - (long)h * p.W);
      const float* xb = p.x + b * p.xv.sB + p.xv.chunk(c) + c4 * 4;
      f32x4 xv[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int gh = h + t / 3 - 1, gw = w + t % 3 - 1;
        const bool ok = gh >= 0 && gh < p.H && gw >= 0 && gw < p.W;
        const int ch_ = min(max(gh, 0), p.H - 1), cw_ = min(max(gw, 0), p.W - 1);
        xv[t] = sisr_keep_if(*reinterpret_cast<const f32x4*>(xb + (long)ch_ * p.xv.sH + (long)cw_ * p.xv.sW), ok);
      }
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a0 += xv[t][e] * wr[t][e][0];
          a1 += xv[t][e] * wr[t][e][1];
          a2 += xv[t][e] * wr[t][e][2];
        }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        a0 += __shfl_xor(a0, o);
        a1 += __shfl_xor(a1, o);
        a2 += __shfl_xor(a2, o);
      }
      if (live && c4 < 3) {
        float v = c4 == 0 ? a0 : (c4 == 1 ? a1 : a2);
        float* yo = p.y + (b * 3 + c4) * hw + r;
        if (c == 0) {
          if (p.bias) v += p.bias[c4];
        } else {
          v += *yo;  // later chunks accumulate (same thread wrote it: no race)
        }
        *yo = v;
      }
    }
  }
}

// ------------------------------------------------------------------ correlation (weight gradients)
struct Corr3Params {
  const float* P;  // [B][3][H][W]
  const float* Q;
  View qv;
  float* part;  // [blocks][chunks][31][64]: rows 0..26 = a*9 + tap, 27 = sum Q, 28..30 = sum P[a]
  int B, H, W, chunks, blocks;
};
#define CORR_ROWS 31

// Persistent blocks walk 4x64-pixel tiles (tile = blockIdx.x, += gridDim.x).  Per tile the 3-channel halo
// (3 x 6 x 66 floats, zero padded) is staged in LDS; wave w owns tile row w, lane c owns channel c of the
// 64-channel chunk blockIdx.y: per pixel one coalesced 256-B Q row (8 rows in flight) and 27 broadcast LDS
// reads feed 27 FMAs into per-lane accumulators.  Waves are combined through LDS in wave order at the end.
#define CT_H 4
#define CT_W 64
#define CP_W (CT_W + 2)
#define CP_N (3 * (CT_H + 2) * CP_W)
__global__ __launch_bounds__(256) void corr3x3_c3_kernel(Corr3Params p) {
  __shared__ float ph[CP_N];
  __shared__ float red[3][CORR_ROWS][64];
  const int c = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = blockIdx.y;
  const long hw = (long)p.H * p.W;
  const int tiles_w = (p.W + CT_W - 1) / CT_W, tiles_h = (p.H + CT_H - 1) / CT_H;
  const int per_img = tiles_w * tiles_h;
  const int total = per_img * p.B;
  float acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.f;
  float qs = 0.f, ps0 = 0.f, ps1 = 0.f, ps2 = 0.f;
  for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
    const int b = tile / per_img;
    const int tr = tile - b * per_img;
    const int th = tr / tiles_w, tw = tr - th * tiles_w;
    const int h0 = th * CT_H, w0 = tw * CT_W;
    __syncthreads();
    for (int i = threadIdx.x; i < CP_N; i += 256) {
      const int a = i / ((CT_H + 2) * CP_W);
      const int r = i - a * ((CT_H + 2) * CP_W);
      const int pr = r / CP_W, pc = r - pr * CP_W;
      const int gh = h0 - 1 + pr, gw = w0 - 1 + pc;
      const bool ok = gh >= 0 && gh < p.H && gw >= 0 && gw < p.W;
      ph[i] = ok ? p.P[((long)b * 3 + a) * hw + (long)gh * p.W + gw] : 0.f;
    }
    __syncthreads();
    const int gh = h0 + wv;
    if (gh < p.H) {
      const float* qrow = p.Q + (long)b * p.qv.sB + (long)gh * p.qv.sH + p.qv.chunk(q) + c;
      const float* prow = ph + wv * CP_W;  // tap (0,0) of tile column 0, channel 0
#pragma unroll 1
      for (int x0 = 0; x0 < CT_W; x0 += 8) {
        float qv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int gw = w0 + x0 + u;
          const float v = qrow[(long)min(gw, p.W - 1) * p.qv.sW];
          qv[u] = gw < p.W ? v : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          qs += qv[u];
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
              const float pv = prow[a * ((CT_H + 2) * CP_W) + (t / 3) * CP_W + (t % 3) + x0 + u];
              acc[a * 9 + t] += pv * qv[u];
            }
          if (w0 + x0 + u < p.W) {  // centre tap = the pixel itself: bias sums of P
            ps0 += prow[0 * ((CT_H + 2) * CP_W) + CP_W + 1 + x0 + u];
            ps1 += prow[1 * ((CT_H + 2) * CP_W) + CP_W + 1 + x0 + u];
            ps2 += prow[2 * ((CT_H + 2) * CP_W) + CP_W + 1 + x0 + u];
          }
        }
      }
    }
  }
  const int pl = wv;
  __syncthreads();
  if (pl > 0) {
#pragma unroll
    for (int k = 0; k < 27; ++k) red[pl - 1][k][c] = acc[k];
    red[pl - 1][27][c] = qs;
    red[pl - 1][28][c] = ps0;
    red[pl - 1][29][c] = ps1;
    red[pl - 1][30][c] = ps2;
  }
  __syncthreads();
  if (pl == 0) {
    float* out = p.part + ((long)blockIdx.x * p.chunks + q) * CORR_ROWS * 64;
#pragma unroll
    for (int k = 0; k < 27; ++k) out[k * 64 + c] = ((acc[k] + red[0][k][c]) + red[1][k][c]) + red[2][k][c];
    out[27 * 64 + c] = ((qs + red[0][27][c]) + red[1][27][c]) + red[2][27][c];
    out[28 * 64 + c] = ((ps0 + red[0][28][c]) + red[1][28][c]) + red[2][28][c];
    out[29 * 64 + c] = ((ps1 + red[0][29][c]) + red[1][29][c]) + red[2][29][c];
    out[30 * 64 + c] = ((ps2 + red[0][30][c]) + red[1][30][c]) + red[2][30][c];
  }
}

struct Corr3Reduce {
  const float* part;
  float* dw;
  float* db;
  long so, si;
  float alpha;
  int blocks, chunks, flip, a_is_out;
};

// blockDim (64, 4): x = output element within a 64-run, y = block-partial group (partials split 4 ways with 8
// loads in flight, group sums added in group order).  Elements: chunks*27*64 weights, then the bias entries.
__global__ __launch_bounds__(256) void corr3_reduce_kernel(Corr3Reduce p) {
  __shared__ float red[4][64];
  const int i = blockIdx.x * 64 + threadIdx.x;
  const int grp = threadIdx.y;
  const int per_chunk = 27 * 64;
  const int nw = p.chunks * per_chunk;
  const long stride = (long)p.chunks * CORR_ROWS * 64;
  long src = -1;
  int j0 = i - nw;
  if (i < nw) {
    const int q = i / per_chunk, r = i - q * per_chunk;
    src = (long)q * CORR_ROWS * 64 + r;  // r = k*64 + c
  } else if (p.db) {
    if (p.a_is_out) {
      if (j0 < 3) src = (long)(28 + j0) * 64;  // sums of P: identical in every lane, take chunk 0 / c = 0
    } else if (j0 < p.chunks * 64) {
      src = (long)(j0 >> 6) * CORR_ROWS * 64 + 27 * 64 + (j0 & 63);
    }
  }
  float s = 0.f;
  if (src >= 0) {
    int k = grp;
    for (; k + 28 < p.blocks; k += 32) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = p.part[(long)(k + 4 * u) * stride + src];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; k < p.blocks; k += 4) s += p.part[(long)k * stride + src];
  }
  red[grp][threadIdx.x] = s;
  __syncthreads();
  if (grp != 0 || src < 0) return;
  s = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
  if (i < nw) {
    const int q = i / per_chunk, r = i - q * per_chunk;
    const int k = r >> 6, c = r & 63;
    const int a = k / 9, t = k - a * 9;
    const long chn = (long)q * 64 + c;
    const long o = p.a_is_out ? a : chn, ii = p.a_is_out ? chn : a;
    p.dw[o * p.so + ii * p.si + (p.flip ? 8 - t : t)] = s * p.alpha;
  } else {
    p.db[j0] = s * p.alpha;
  }
}

// ------------------------------------------------------------------ C ABI
extern "C" int sisr_conv3x3_cin3(const float* x, const float* w, int64_t so, int64_t si, int flip_taps, const float* bias,
                                 float* y, const int64_t* yview, int B, int H, int W, int cout, void* stream) {
  if (!x || !w || !y || !yview || B <= 0 || H <= 0 || W <= 0) return SISR_ERR_ARG;
  if (cout <= 0 || (cout & 63) || cout > 512) return SISR_ERR_UNSUPPORTED;
  if (!sisr_aligned16(bias)) return SISR_ERR_ALIGN;
  Cin3Params p;
  p.x = x;
  p.y = y;
  p.yv = view_from(yview);
  if ((p.yv.sB | p.yv.sH | p.yv.sW | p.yv.chi | p.yv.clo) & 3 || !sisr_aligned16(y)) return SISR_ERR_ALIGN;
  p.w = w;
  p.bias = bias;
  p.so = so;
  p.si = si;
  p.flip = flip_taps;
  p.B = B;
  p.H = H;
  p.W = W;
  p.cout = cout;
  const long total = (long)B * H * W * (cout >> 4);
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(conv3x3_cin3_kernel, dim3((unsigned)blocks), dim3(256), 27 * cout * sizeof(float),
                     (hipStream_t)stream, p);
  return sisr_check_launch();
}

extern "C" int sisr_conv3x3_cout3(const float* x, const int64_t* xview, const float* w, int64_t so, int64_t si,
                                  int flip_taps, const float* bias, float* y, int B, int H, int W, int cin,
                                  void* stream) {
  if (!x || !w || !y || !xview || B <= 0 || H <= 0 || W <= 0) return SISR_ERR_ARG;
  if (cin <= 0 || (cin & 63)) return SISR_ERR_UNSUPPORTED;
  Cout3Params p;
  p.x = x;
  p.xv = view_from(xview);
  if ((p.xv.sB | p.xv.sH | p.xv.sW | p.xv.chi | p.xv.clo) & 3 || !sisr_aligned16(x)) return SISR_ERR_ALIGN;
  p.y = y;
  p.w = w;
  p.bias = bias;
  p.so = so;
  p.si = si;
  p.flip = flip_taps;
  p.B = B;
  p.H = H;
  p.W = W;
  p.cin_chunks = cin / 64;
  if (cin == 64) {  // one chunk: the matrix-core kernel (conv_rgb_out.h); wider inputs stay on the streaming kernel below
    RgbOutParams q = {};
    q.x = x;
    q.sB = p.xv.sB;
    q.sH = p.xv.sH;
    q.sW = p.xv.sW;
    q.w = w;
    q.so = so;
    q.si = si;
    q.flip = flip_taps;
    q.bias = bias;
    q.y = y;
    q.B = B;
    q.H = H;
    q.W = W;
    return rgb_out_launch<3, 3>(q, stream);
  }
  const long npix = (long)B * H * W;
  long blocks = (npix + 15) / 16;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(conv3x3_cout3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  return sisr_check_launch();
}

static int corr3_blocks(int B, int H, int W) {
  long nb = (long)B * ((H + CT_H - 1) / CT_H) * ((W + CT_W - 1) / CT_W);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  return (int)nb;
}

extern "C" size_t sisr_corr3x3_c3_workspace_bytes(int B, int H, int W, int channels) {
  if (B <= 0 || H <= 0 || W <= 0 || channels <= 0 || (channels & 63)) return 0;
  return (size_t)corr3_blocks(B, H, W) * (channels / 64) * CORR_ROWS * 64 * sizeof(float);
}

// dw element for (3-channel index a, tap t, 64k-channel index c) goes to
//   dw[o*so + i*si + (flip ? 8-t : t)],  (o,i) = a_is_out ? (a,c) : (c,a)
// db: a_is_out ? 3 sums of P : `channels` sums of Q.
extern "C" int sisr_corr3x3_c3(const float* P, const float* Q, const int64_t* qview, float alpha, float* dw, int64_t so,
                               int64_t si, int flip_taps, int a_is_out, float* dbias, float* workspace,
                               size_t workspace_bytes, int B, int H, int W, int channels, void* stream) {
  if (!P || !Q || !qview || !dw || !workspace || B <= 0 || H <= 0 || W <= 0) return SISR_ERR_ARG;
  if (channels <= 0 || (channels & 63)) return SISR_ERR_UNSUPPORTED;
  if (workspace_bytes < sisr_corr3x3_c3_workspace_bytes(B, H, W, channels)) return SISR_ERR_ARG;
  Corr3Params p;
  p.P = P;
  p.Q = Q;
  p.qv = view_from(qview);
  p.part = workspace;
  p.B = B;
  p.H = H;
  p.W = W;
  p.chunks = channels / 64;
  p.blocks = corr3_blocks(B, H, W);
  hipLaunchKernelGGL(corr3x3_c3_kernel, dim3(p.blocks, p.chunks), dim3(256), 0, (hipStream_t)stream, p);
  int rc = sisr_check_launch();
  if (rc) return rc;
  Corr3Reduce r;
  r.part = workspace;
  r.dw = dw;
  r.db = dbias;
  r.so = so;
  r.si = si;
  r.alpha = alpha;
  r.blocks = p.blocks;
  r.chunks = p.chunks;
  r.flip = flip_taps;
  r.a_is_out = a_is_out;
  const int total = p.chunks * 27 * 64 + (dbias ? (a_is_out ? 3 : channels) : 0);
  hipLaunchKernelGGL(corr3_reduce_kernel, dim3((total + 63) / 64), dim3(64, 4), 0, (hipStream_t)stream, r);
  return sisr_check_launch();
}
