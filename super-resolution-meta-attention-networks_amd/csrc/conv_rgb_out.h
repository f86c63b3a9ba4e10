// 3-output-channel convolutions (KS x KS, 64 -> 3) on the fp32 matrix cores: the RGB ends of the networks.
//   KS = 9, OR = 1: SFTMD's conv_output (ref: SFTMD_variants/architectures.py:159)
//   KS = 3, OR = 3: the 64 -> 3 tail conv of RCAN / EDSR / HAN / SAN (ref: advanced/architectures.py:150-152) and, with
//                   role-swapped flipped weights, the input gradient of the 3 -> 64 head conv
// Three output channels alone would leave 29 of the 32 MFMA columns empty, so the columns are (output row, channel, kw):
//   Z[q][(orow, co, kw)] = sum_{r < KS + OR - 1, ci} x[row0 + r][q][ci] w[co][ci][kh = r - orow][kw]     (zero where kh is outside)
//   y[co][row0 + PAD + orow][p] = b[co] + sum_kw Z[p + kw - PAD][(orow, co, kw)]                     through LDS
// M = 32 consecutive pixels q of a row (32 - (KS - 1) of them are output columns), N = 3 KS OR = 27 -> 32, K = (KS + OR - 1) x 64.
// A fragments come straight from HBM / L2 as 16-B loads (channels 8j + 4k .. + 3 of the lane's pixel feed four
// v_mfma_f32_32x32x2_f32), B from LDS (staged once per workgroup), one row ahead in registers.
#pragma once
#include "sisr_common.h"

#define RGBO_WAVES 8
#define RGBO_ZLD 33

struct RgbOutParams {
  const float* x;
  long sB, sH, sW;  // floats: batch / row / pixel stride of the 64-channel map
  const float* w;
  long so, si;      // weight strides of (output channel, input channel); taps contiguous, optionally flipped
  int flip;
  const float* bias;
  float* y;         // [B][3][H][W]
  int B, H, W, tiles_w, tiles_h;
  long ntiles;
};

template <int KS, int OR>
__global__ __launch_bounds__(64 * RGBO_WAVES) void rgb_out_mfma_kernel(RgbOutParams p) {
  constexpr int KR = KS + OR - 1, PAD = (KS - 1) / 2, TCOL = 32 - (KS - 1), NV = 3 * KS * OR, NOUT = 3 * OR * TCOL;
  static_assert(NV <= 32, "columns (orow, co, kw) must fit one MFMA tile");
  extern __shared__ __attribute__((aligned(16))) float rgbo_lds[];
  float* wz = rgbo_lds;  // [r][octet j][k][n][e] = w[co(n)][8j + 4k + e][r - orow(n)][kw(n)]
  float* zt = rgbo_lds + KR * 2048 + (threadIdx.x >> 6) * (32 * RGBO_ZLD);
  for (int i = threadIdx.x; i < KR * 2048; i += 64 * RGBO_WAVES) {
    const int e = i & 3, n = (i >> 2) & 31, k = (i >> 7) & 1, j = (i >> 8) & 7, r = i >> 11;
    const int orow = n / (3 * KS), co = (n / KS) % 3, kw = n % KS, kh = r - orow, ci = 8 * j + 4 * k + e;
    float v = 0.f;
    if (n < NV && kh >= 0 && kh < KS) {
      const int t = kh * KS + kw;
      v = p.w[(long)co * p.so + (long)ci * p.si + (p.flip ? KS * KS - 1 - t : t)];
    }
    wz[i] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, kk = lane >> 5;
  const int H = p.H, W = p.W;
  const long hw = (long)H * W;
  for (long t = (long)blockIdx.x * RGBO_WAVES + wave; t < p.ntiles; t += (long)gridDim.x * RGBO_WAVES) {
    const int tw = (int)(t % p.tiles_w);
    const long rr = t / p.tiles_w;
    const int th = (int)(rr % p.tiles_h), b = (int)(rr / p.tiles_h);
    const int oy0 = th * OR, row0 = oy0 - PAD;  // first output row, first input row
    const int ox0 = tw * TCOL, qx = ox0 - PAD + li;
    const bool colok = qx >= 0 && qx < W;
    const float* xp = p.x + (long)b * p.sB + (long)min(max(qx, 0), W - 1) * p.sW + 4 * kk;
    f32x16 acc = {0};
    const int r0 = max(0, -row0), r_end = min(KR, H - row0);  // rows row0 + r inside the image
    const bool edge = ox0 - PAD < 0 || ox0 - PAD + 32 > W;    // uniform: only edge tiles pay for the column mask
    // two register sets, one row ahead: row r + 1 is in flight while row r feeds the matrix cores (no register moves)
    f32x4 ra[8], rb[8];
    auto load_row = [&](f32x4(&dst)[8], int r) {
      const float* row = xp + (long)(row0 + min(r, r_end - 1)) * p.sH;
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[j] = *reinterpret_cast<const f32x4*>(row + 8 * j);
    };
    auto mma_row = [&](f32x4(&src)[8], int r) {
      const float* wk = wz + r * 2048 + kk * 128 + li * 4;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f32x4 a = edge ? sisr_keep_if(src[j], colok) : src[j];
        const f32x4 bb = *reinterpret_cast<const f32x4*>(wk + j * 256);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], bb[e], acc, 0, 0, 0);
      }
    };
    load_row(ra, r0);
    for (int r = r0; r < r_end; r += 2) {
      load_row(rb, r + 1);
      mma_row(ra, r);
      if (r + 1 < r_end) {
        load_row(ra, r + 2);
        mma_row(rb, r + 1);
      }
    }
    // Z tile -> LDS (row = tile column, column = n); one wave owns zt, LDS operations of a wave complete in order
#pragma unroll
    for (int q = 0; q < 16; ++q) zt[((q & 3) + 8 * (q >> 2) + 4 * kk) * RGBO_ZLD + li] = acc[q];
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < (NOUT + 63) / 64; ++u) {
      const int idx = lane + 64 * u;
      if (idx < NOUT) {
        const int oc = idx / TCOL, o = idx - oc * TCOL;  // oc = orow * 3 + co
        const int orow = oc / 3, co = oc - orow * 3;
        float v = p.bias ? p.bias[co] : 0.f;
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) v += zt[(o + kw) * RGBO_ZLD + oc * KS + kw];
        if (ox0 + o < W && oy0 + orow < H) p.y[((long)b * 3 + co) * hw + (long)(oy0 + orow) * W + ox0 + o] = v;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
}

template <int KS, int OR>
static int rgb_out_launch(RgbOutParams p, void* stream) {
  constexpr int KR = KS + OR - 1, TCOL = 32 - (KS - 1);
  p.tiles_w = (p.W + TCOL - 1) / TCOL;
  p.tiles_h = (p.H + OR - 1) / OR;
  p.ntiles = (long)p.B * p.tiles_h * p.tiles_w;
  const size_t lds = (KR * 2048 + RGBO_WAVES * 32 * RGBO_ZLD) * sizeof(float);
  const int per_cu = lds > 80 * 1024 ? 1 : 2;
  long blocks = (p.ntiles + RGBO_WAVES - 1) / RGBO_WAVES;
  if (blocks > 256 * per_cu) blocks = 256 * per_cu;
  SISR_ALLOW_LDS((rgb_out_mfma_kernel<KS, OR>), lds);
  hipLaunchKernelGGL((rgb_out_mfma_kernel<KS, OR>), dim3((unsigned)blocks), dim3(64 * RGBO_WAVES), lds, (hipStream_t)stream, p);
  return sisr_check_launch();
}
