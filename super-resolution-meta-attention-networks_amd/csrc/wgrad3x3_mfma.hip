// Weight (and bias) gradient of the 3x3 same-convolution over 64-channel chunks, fp32 MFMA.
//
//   dW[co][ci][kh][kw] = alpha * sum_{b,h,w} X[b][h+kh-1][w+kw-1][ci] * dY'[b][h][w][co]
//   db[co]             = alpha * sum_{b,h,w} dY'[b][h][w][co]
//   dY' = dY * dy_scale[b][co] + dy_shift[b][co]   (optional; the RCAB backward never materialises
//                                                  dRes = dOut*g + c, it is rebuilt on load)
// This is autograd's convolution_backward weight/bias output for the reference's default_conv
// (advanced/common.py:5-8); GEMM shape M = 32 ci, N = 32 co, K = pixels per quadrant.
//
// Decomposition: blockIdx.y = ((cin_chunk*cout_chunks + cout_chunk)*4 + quadrant), quadrant =
// (ci half, co half); blockIdx.x = K-slice s of S.  A workgroup (4 waves) walks 8x32-pixel tiles
// s, s+S, ...; per tile it stages the 32-channel halves of the X halo (10x34 px) and of dY (8x32 px)
// in LDS, and wave w owns tile rows {2w,2w+1}: per pixel pair one B read (dY) and nine A reads (X at
// the nine taps) feed nine 32x32x2 MFMAs into nine persistent accumulators (144 VGPRs).  After the
// last tile the four waves' accumulators are summed through LDS in a fixed order and written as one
// slab; a second kernel adds the S slabs in index order (deterministic, no atomics) and scatters
// into the OIHW gradient.  Splitting the OUTPUT four ways instead of giving each wave a quadrant
// cuts the slab traffic 4x (36.9 KB per workgroup) at the price of re-reading the inputs from L2.
#include "sisr_common.h"
#include <string.h>
#include <stdlib.h>

#define WT_H 8
#define WT_W 32
#define WH_H (WT_H + 2)
#define WH_W (WT_W + 2)
#define WSTR 32                                   // floats per pixel in LDS (32-channel half)
#define X_ITEMS (WH_H * WH_W * 8)                 // float4 items: 2720
#define X_ITERS ((X_ITEMS + 255) / 256)           // 11
#define Y_ITEMS (WT_H * WT_W * 8)                 // 2048
#define Y_ITERS (Y_ITEMS / 256)                   // 8
#define LDS_X (WH_H * WH_W * WSTR)                // floats
#define LDS_Y (WT_H * WT_W * WSTR)
#define SLAB (9 * 16 * 64)                        // floats per workgroup slab: 9216
#define WG_MAX_UNITS 64                           // (cin / 64) * (cout / 64) * 4 quadrants of a maskable launch

struct WgradParams {
  const float* x;
  View xv;
  const float* dy;
  View yv;
  const float* dy_scale;
  const float* dy_shift;
  float* slabs;      // [S][gridDim.y][SLAB]
  float* bias_slabs; // [S][cout_chunks*2][32] (written by the first active unit of each (cout chunk, co half)) or null
  int B, H, W, cin_chunks, cout_chunks, tiles_w, tiles_h, S;
  // active units (blockIdx.y -> unit); all of them unless the caller knows blocks of the gradient to be unused
  int mapped;
  unsigned char unit_map[WG_MAX_UNITS];
  unsigned long long bias_units;  // bit blockIdx.y: this workgroup row also sums dY for the bias gradient
  // GEO kernels (SPARNet's ConvLayers): x is read through ReflectionPad2d(1) of the H x W map (geo_reflect: -1 -> 1, n -> n - 2
  // instead of zeros) which is stored subsampled by 2^geo_up (nearest upsampling read in place)
  int geo_reflect, geo_up;
  int geo_ysub;  // GEO: dY is stored subsampled by 2 and read ZERO-STUFFED (the gradient of a stride-2 conv seen at stride 1)
  int units;  // GEO batch launches: this job's workgroup rows (grid rows >= units and columns >= S leave at once); 0 = gridDim.y
};

template <bool GEO = false>
static __device__ __forceinline__ void wgrad3x3_c64_body(const WgradParams& p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ldx = lds;
  float* ldy = lds + LDS_X;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nrows = GEO && p.units ? p.units : (int)gridDim.y;
  if (GEO && ((int)blockIdx.x >= p.S || (int)blockIdx.y >= nrows)) return;  // a smaller job of a batched launch
  const int unit = p.mapped ? p.unit_map[blockIdx.y] : (int)blockIdx.y;
  const int quad = unit & 3, pair = unit >> 2;
  const int cq = pair % p.cout_chunks, cc = pair / p.cout_chunks;
  const int cih = quad >> 1, coh = quad & 1;
  const int i = lane & 31, kk = lane >> 5;
  const int H = p.H, W = p.W;
  const bool do_bias = p.bias_slabs && (p.mapped ? (int)((p.bias_units >> blockIdx.y) & 1) : (cc == 0 && cih == 0));
  const int Cout = p.cout_chunks * 64;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f32x16){0};
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};  // this thread's 4 channels (tid&7) of the dY column sums

  const int tiles_per_img = p.tiles_w * p.tiles_h;
  const int total = tiles_per_img * p.B;
  for (int tile = blockIdx.x; tile < total; tile += p.S) {
    const int b = tile / tiles_per_img;
    const int tr = tile - b * tiles_per_img;
    const int th = tr / p.tiles_w, tw = tr - th * p.tiles_w;
    const int h0 = th * WT_H, w0 = tw * WT_W;
    __syncthreads();  // previous tile fully consumed
    {  // Issue-lean staging (the fp32 MFMA owns the SIMD's issue port for its 64 cycles, so every staging
       // instruction of this wave is paid on top of the co-resident waves' MFMAs): thread (c4, pcol) owns chunk c4
       // of x halo column pcol + 1 and of dY column pcol; the two edge columns of the halo (10 rows x 2 x 8 chunks)
       // go one item per thread to tid < 160, so every load is unconditional: one global_load (scalar row base + lane
       // offset), one mask op and one ds_write per item.  Two batches of ten loads (144 accumulator VGPRs are live),
       // x and dY rows together, so a tile waits for memory twice.
      int tl = tid;
      asm volatile("" : "+v"(tl));
      const int c4 = tl & 7, pcol = tl >> 3;
      const int eidx = tl % 160, er = eidx >> 4, eside = (eidx >> 3) & 1, ec4 = eidx & 7;
      const float* xb = p.x + (long)b * p.xv.sB + p.xv.chunk(cc) + cih * 32;
      const float* yb = p.dy + (long)b * p.yv.sB + p.yv.chunk(cq) + coh * 32;
      const bool okc = w0 + pcol < W;
      const bool refl = GEO && p.geo_reflect;  // scalar
      const int gup = GEO ? p.geo_up : 0;
      auto xpix = [&](int g, int n) {  // stored row / column behind virtual coordinate g of an n-pixel axis
        if (refl) g = g < 0 ? -g : (g >= n ? 2 * n - 2 - g : g);
        return min(max(g, 0), n - 1) >> gup;
      };
      const unsigned gx = (unsigned)(xpix(w0 + pcol, W) * (int)p.xv.sW + c4 * 4), lx = (pcol + 1) * WSTR + c4 * 4;
      const int ysub = GEO ? p.geo_ysub : 0;  // scalar
      const unsigned gy = (unsigned)((min(w0 + pcol, W - 1) >> ysub) * (int)p.yv.sW + c4 * 4), ly = pcol * WSTR + c4 * 4;
      const bool oky = okc && !(ysub && ((w0 + pcol) & 1));
      f32x4 s4 = {1.f, 1.f, 1.f, 1.f}, t4 = {0.f, 0.f, 0.f, 0.f};
      if (p.dy_scale) s4 = *reinterpret_cast<const f32x4*>(p.dy_scale + (long)b * Cout + cq * 64 + coh * 32 + c4 * 4);
      if (p.dy_shift) t4 = *reinterpret_cast<const f32x4*>(p.dy_shift + (long)b * Cout + cq * 64 + coh * 32 + c4 * 4);
      // what a tile does not need is skipped by scalar branches, not computed with neutral constants (every vector
      // instruction here costs the co-resident workgroup's MFMA stream ~4.5 cycles): the zero-padding mask on tiles whose
      // halo lies inside the image, the affine rebuild when dY is taken as it is, the column sums where no bias gradient
      // is wanted from this quadrant
      const bool interior = !ysub && h0 >= 1 && h0 + WT_H + 1 <= H && w0 >= 1 && w0 + WT_W + 1 <= W;  // scalar
      const bool affine = p.dy_scale != nullptr || p.dy_shift != nullptr;                     // scalar
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        f32x4 v[5], u[4], e;
#pragma unroll
        for (int r = 0; r < 5; ++r)
          v[r] = *reinterpret_cast<const f32x4*>(xb + (long)xpix(h0 - 1 + 5 * half + r, H) * p.xv.sH + gx);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          u[r] = *reinterpret_cast<const f32x4*>(yb + (long)(min(h0 + 4 * half + r, H - 1) >> ysub) * p.yv.sH + gy);
        const int gwe = eside ? w0 + WT_W : w0 - 1;
        if (half == 0)
          e = *reinterpret_cast<const f32x4*>(xb + (long)xpix(h0 - 1 + er, H) * p.xv.sH + xpix(gwe, W) * (int)p.xv.sW + ec4 * 4);
        if (affine) {
#pragma unroll
          for (int r = 0; r < 4; ++r) u[r] = u[r] * s4 + t4;
        }
        if (!interior) {
          if (!refl) {  // reflected reads are values of the map; what pairs with pixels outside it is zeroed with dY below
#pragma unroll
            for (int r = 0; r < 5; ++r) {
              const int gh = h0 - 1 + 5 * half + r;
              v[r] = sisr_keep_if(v[r], gh >= 0 && gh < H && okc);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) u[r] = sisr_keep_if(u[r], oky && (h0 + 4 * half + r < H) && !(ysub && ((h0 + r) & 1)));
        }
#pragma unroll
        for (int r = 0; r < 5; ++r) *reinterpret_cast<f32x4*>(ldx + (5 * half + r) * (WH_W * WSTR) + lx) = v[r];
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(ldy + (4 * half + r) * (WT_W * WSTR) + ly) = u[r];
        if (do_bias) {
#pragma unroll
          for (int r = 0; r < 4; ++r) bsum += u[r];
        }
        if (half == 0) {
          const int ghe = h0 - 1 + er;
          if (!interior && !refl) e = sisr_keep_if(e, ghe >= 0 && ghe < H && gwe >= 0 && gwe < W);
          if (tl < 160) *reinterpret_cast<f32x4*>(ldx + er * (WH_W * WSTR) + (eside ? WH_W - 1 : 0) * WSTR + ec4 * 4) = e;
        }
      }
    }
    __syncthreads();

    // ---- wave w: rows 2w, 2w+1; 16 pixel pairs per row; K index (lane>>5) = pixel parity
#pragma unroll 1
    for (int r = 2 * wave; r < 2 * wave + 2; ++r) {
      const float* xa = ldx + (r * WH_W + kk) * WSTR + i;      // tap (0,0) of pair u=0
      const float* yb = ldy + (r * WT_W + kk) * WSTR + i;
#pragma unroll 4
      for (int u = 0; u < 16; ++u) {
        const float bv = yb[u * 2 * WSTR];
        float av[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) av[t] = xa[((t / 3) * WH_W + (t % 3) + 2 * u) * WSTR];
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv, acc[t], 0, 0, 0);
      }
    }
  }

  // ---- sum the four waves' accumulators through LDS (fixed order: ((w0+w2)+(w1+w3)) ), write slab
  __syncthreads();
  float* red = lds;  // 2 * SLAB floats = 73.7 KB <= LDS_X + LDS_Y (76.3 KB)
  if (wave >= 2) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(wave - 2) * SLAB + (t * 16 + r) * 64 + lane] = acc[t][r];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] += red[wave * SLAB + (t * 16 + r) * 64 + lane];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(t * 16 + r) * 64 + lane] = acc[t][r];
  }
  if (do_bias) *reinterpret_cast<f32x4*>(red + SLAB + tid * 4) = bsum;
  __syncthreads();
  if (wave == 0) {
    float* out = p.slabs + ((long)blockIdx.x * nrows + blockIdx.y) * SLAB;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[(t * 16 + r) * 64 + lane] = acc[t][r] + red[(t * 16 + r) * 64 + lane];
  }
  if (do_bias && tid < 32) {  // channel tid = (c4 = tid>>2, e = tid&3): threads with tid&7 == c4 hold it
    float s = 0.f;
    const int c4 = tid >> 2, e = tid & 3;
    for (int k = 0; k < 32; ++k) s += red[SLAB + (k * 8 + c4) * 4 + e];
    p.bias_slabs[((long)blockIdx.x * p.cout_chunks * 2 + cq * 2 + coh) * 32 + tid] = s;
  }
}

__global__ __launch_bounds__(256, 2) void wgrad3x3_c64_kernel(WgradParams p) { wgrad3x3_c64_body(p); }
__global__ __launch_bounds__(256, 2) void wgrad3x3_c64_geo_kernel(WgradParams p) { wgrad3x3_c64_body<true>(p); }

// Several weight gradients of one geometry in ONE launch (blockIdx.z = job).  At a few tiles per GPU a single gradient has
// two tiles per workgroup: slab epilogue, ramp and drain are as long as the work.  Eight of them share the 512 resident
// workgroups instead -- every workgroup walks 16 tiles of its own job, as at batch 32 -- and the second stage adds 16 slabs
// per job instead of 128.  Per-job results are those of a single launch with the same K-split (same kernel body).
#define WG_BATCH 8
struct WgradBatch {
  WgradParams job[WG_BATCH];
};
__global__ __launch_bounds__(256, 2) void wgrad3x3_c64_batch_kernel(WgradBatch bt) { wgrad3x3_c64_body(bt.job[blockIdx.z]); }
// ... and of DIFFERENT geometries (SPARNet: ~110 weight gradients per step on maps of 4^2 .. 32^2 pixels, 16 - 64 workgroups and
// ~20 us each when launched alone): the grid is sized for the largest job, the others' surplus workgroups leave at once.
__global__ __launch_bounds__(256, 2) void wgrad3x3_c64_geo_batch_kernel(WgradBatch bt) { wgrad3x3_c64_body<true>(bt.job[blockIdx.z]); }

// ------------------------------------------------------------------ fp32, one persistent workgroup per CU (round 3)
// The kernel above splits a tile's OUTPUT over four workgroups (each re-reads its half of x and dY: 1.45x the algorithmic
// traffic) and its pixels over the four waves (cross-wave reduction at the end), and it does not overlap a tile's loads with
// anything of its own: it relies on the second resident workgroup, whose K phase serialises with it on the MFMA pipe and
// leaves ~0.4 us holes at every hand-over (per-CU timelines of the conv kernel, tools/conv_timeline.py).  This form -- the
// geometry of the bf16 kernel below, in exact fp32 -- gives a workgroup the whole 64 ci x 64 co of a chunk pair:
//   wave w = quadrant (ci half w >> 1, co half w & 1), nine taps in nine persistent 32 x 32 accumulators (144 registers);
//   all four waves walk the same staged tile (8 x 32 pixels: the 10 x 34 x 64 halo of x and 8 x 32 x 64 of dY, 152.6 KB of
//   LDS), so every input byte is read from HBM once and no cross-wave reduction is needed;
//   one workgroup per CU (512 registers per wave): the 38 float4 per thread of tile i + 1 are requested before the K loop
//   of tile i and written to LDS after it, so a tile waits for memory only once per launch;
//   the K loop is MFMA + ds_read_b32 only (both free beside each other, tools/mfma_fill.py); staging skips the padding
//   mask on interior tiles, the affine rebuild when dY is taken as is, and the bias sums where none is wanted.
// Slab layout, bias slabs and the second-stage reduction are the shared ones (unit = pair * 4 + quadrant).
#define FW_X (WH_H * WH_W * 64)  // floats
#define FW_Y (WT_H * WT_W * 64)

template <bool GEO = false>
static __device__ __forceinline__ void wgrad3x3_c64_full_body(const WgradParams& p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ldx = lds;
  float* ldy = lds + FW_X;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pair = blockIdx.y;
  const int cq = pair % p.cout_chunks, cc = pair / p.cout_chunks;
  const int cih = __builtin_amdgcn_readfirstlane(wave >> 1), coh = __builtin_amdgcn_readfirstlane(wave & 1);
  const int i = lane & 31, kk = lane >> 5;
  const int H = p.H, W = p.W;
  const bool do_bias = p.bias_slabs && cc == 0;
  const bool affine = p.dy_scale != nullptr || p.dy_shift != nullptr;  // scalar
  const int Cout = p.cout_chunks * 64;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f32x16){0};
  f32x4 bsa = {0.f, 0.f, 0.f, 0.f}, bsb = bsa;

  const int tiles_per_img = p.tiles_w * p.tiles_h;
  const int total = tiles_per_img * p.B;
  // every XCD sweeps a contiguous eighth of the tiles (vertically adjacent tiles share two halo rows through one L2)
  int t_begin = blockIdx.x, t_end = total, t_step = p.S;
  if ((p.S & 7) == 0) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per = (total + 7) >> 3;
    t_begin = xcd * per + idx;
    t_end = min(total, (xcd + 1) * per);
    t_step = p.S >> 3;
  }
  struct Stage {
    f32x4 x[WH_H][2];
    f32x4 xe[2];
    f32x4 y[WT_H][2];
  };
  Stage st;
  const int c8 = tid & 7, pcol = tid >> 3;
  const int eidx = tid % 160, er = eidx >> 4, eside = (eidx >> 3) & 1, ec8 = eidx & 7;
  auto decode = [&](int tile, int& b, int& h0, int& w0) {
    b = tile / tiles_per_img;
    const int tr = tile - b * tiles_per_img;
    const int th = tr / p.tiles_w;
    h0 = th * WT_H;
    w0 = (tr - th * p.tiles_w) * WT_W;
  };
  const bool refl = GEO && p.geo_reflect;  // scalar
  const int gup = GEO ? p.geo_up : 0, ysub = GEO ? p.geo_ysub : 0;
  auto xpix = [&](int g, int n) {  // stored row / column behind virtual coordinate g of an n-pixel axis
    if (refl) g = g < 0 ? -g : (g >= n ? 2 * n - 2 - g : g);
    return min(max(g, 0), n - 1) >> gup;
  };
  auto issue = [&](int tile) {
    int b, h0, w0;
    decode(tile, b, h0, w0);
    const sisr_rsrc_t rx = sisr_rsrc(p.x + (long)b * p.xv.sB + p.xv.chunk(cc));
    const sisr_rsrc_t ry = sisr_rsrc(p.dy + (long)b * p.yv.sB + p.yv.chunk(cq));
    const unsigned vx = (unsigned)(xpix(w0 + pcol, W) * (int)p.xv.sW + c8 * 8) * 4u;
    const unsigned vy = (unsigned)((min(w0 + pcol, W - 1) >> ysub) * (int)p.yv.sW + c8 * 8) * 4u;
#pragma unroll
    for (int r = 0; r < WH_H; ++r) {
      const unsigned so = (unsigned)(xpix(h0 - 1 + r, H) * (int)p.xv.sH) * 4u;  // scalar
      st.x[r][0] = sisr_buf_load4(rx, vx, so);
      st.x[r][1] = sisr_buf_load4(rx, vx + 16u, so);
    }
    {
      const int gwe = xpix(eside ? w0 + WT_W : w0 - 1, W);
      const unsigned ve = (unsigned)(xpix(h0 - 1 + er, H) * (int)p.xv.sH + gwe * (int)p.xv.sW + ec8 * 8) * 4u;
      st.xe[0] = sisr_buf_load4(rx, ve, 0u);
      st.xe[1] = sisr_buf_load4(rx, ve + 16u, 0u);
    }
#pragma unroll
    for (int r = 0; r < WT_H; ++r) {
      const unsigned so = (unsigned)((min(h0 + r, H - 1) >> ysub) * (int)p.yv.sH) * 4u;  // scalar
      st.y[r][0] = sisr_buf_load4(ry, vy, so);
      st.y[r][1] = sisr_buf_load4(ry, vy + 16u, so);
    }
  };
  auto commit = [&](int tile) {
    int b, h0, w0;
    decode(tile, b, h0, w0);
    const bool interior = !ysub && h0 >= 1 && h0 + WT_H + 1 <= H && w0 >= 1 && w0 + WT_W + 1 <= W;  // scalar
    const bool okc = w0 + pcol < W;
    const bool oky = okc && !(ysub && ((w0 + pcol) & 1));
    float* lx = ldx + (pcol + 1) * 64 + c8 * 8;
    if (!interior && !refl) {  // (reflected reads are values of the map; what pairs with pixels outside it is zeroed with dY)
#pragma unroll
      for (int r = 0; r < WH_H; ++r) {
        const int gh = h0 - 1 + r;
        const bool ok = gh >= 0 && gh < H && okc;
        st.x[r][0] = sisr_keep_if(st.x[r][0], ok);
        st.x[r][1] = sisr_keep_if(st.x[r][1], ok);
      }
      const int gwe = eside ? w0 + WT_W : w0 - 1, ghe = h0 - 1 + er;
      const bool oke = ghe >= 0 && ghe < H && gwe >= 0 && gwe < W;
      st.xe[0] = sisr_keep_if(st.xe[0], oke);
      st.xe[1] = sisr_keep_if(st.xe[1], oke);
    }
#pragma unroll
    for (int r = 0; r < WH_H; ++r) {
      *reinterpret_cast<f32x4*>(lx + r * (WH_W * 64)) = st.x[r][0];
      *reinterpret_cast<f32x4*>(lx + r * (WH_W * 64) + 4) = st.x[r][1];
    }
    if (tid < 160) {
      float* le = ldx + (er * WH_W + (eside ? WH_W - 1 : 0)) * 64 + ec8 * 8;
      *reinterpret_cast<f32x4*>(le) = st.xe[0];
      *reinterpret_cast<f32x4*>(le + 4) = st.xe[1];
    }
    if (affine) {
      f32x4 s4a = {1.f, 1.f, 1.f, 1.f}, s4b = s4a, t4a = {0.f, 0.f, 0.f, 0.f}, t4b = t4a;
      if (p.dy_scale) {
        const float* sp = p.dy_scale + (long)b * Cout + cq * 64 + c8 * 8;
        s4a = *reinterpret_cast<const f32x4*>(sp);
        s4b = *reinterpret_cast<const f32x4*>(sp + 4);
      }
      if (p.dy_shift) {
        const float* tp = p.dy_shift + (long)b * Cout + cq * 64 + c8 * 8;
        t4a = *reinterpret_cast<const f32x4*>(tp);
        t4b = *reinterpret_cast<const f32x4*>(tp + 4);
      }
#pragma unroll
      for (int r = 0; r < WT_H; ++r) {
        st.y[r][0] = st.y[r][0] * s4a + t4a;
        st.y[r][1] = st.y[r][1] * s4b + t4b;
      }
    }
    if (!interior) {
#pragma unroll
      for (int r = 0; r < WT_H; ++r) {
        const bool ok = oky && (h0 + r < H) && !(ysub && ((h0 + r) & 1));
        st.y[r][0] = sisr_keep_if(st.y[r][0], ok);
        st.y[r][1] = sisr_keep_if(st.y[r][1], ok);
      }
    }
    float* ly = ldy + pcol * 64 + c8 * 8;
#pragma unroll
    for (int r = 0; r < WT_H; ++r) {
      *reinterpret_cast<f32x4*>(ly + r * (WT_W * 64)) = st.y[r][0];
      *reinterpret_cast<f32x4*>(ly + r * (WT_W * 64) + 4) = st.y[r][1];
    }
    if (do_bias) {
#pragma unroll
      for (int r = 0; r < WT_H; ++r) {
        bsa += st.y[r][0];
        bsb += st.y[r][1];
      }
    }
  };

  if (t_begin < t_end) {
    issue(t_begin);
    commit(t_begin);
  }
  __syncthreads();
  for (int tile = t_begin; tile < t_end; tile += t_step) {
    const bool has_next = tile + t_step < t_end;  // uniform
    if (has_next) issue(tile + t_step);
    // ---- K loop: 8 rows x 16 pixel pairs x 9 taps; K index (lane >> 5) = pixel parity.  One wave per SIMD: nobody else
    // covers an LDS round trip, so the ten operands of pair u + 1 (or of the next row's first pair) are requested before
    // the nine MFMAs of pair u are issued.
    {
      const float* xa = ldx + kk * 64 + cih * 32 + i;  // tap (0,0) of pair 0, row 0
      const float* yb = ldy + kk * 64 + coh * 32 + i;
      float bv = yb[0], av[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) av[t] = xa[((t / 3) * WH_W + (t % 3)) * 64];
#pragma unroll 1
      for (int r = 0; r < WT_H; ++r) {
        const int rn = r + 1 < WT_H ? 1 : 0;  // scalar: the last row re-reads itself (the values are not used)
        const float* xn = xa + rn * (WH_W * 64);
        const float* yn = yb + rn * (WT_W * 64);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          float bn, an[9];
          if (u < 15) {
            bn = yb[(u + 1) * 2 * 64];
#pragma unroll
            for (int t = 0; t < 9; ++t) an[t] = xa[((t / 3) * WH_W + (t % 3) + 2 * (u + 1)) * 64];
          } else {
            bn = yn[0];
#pragma unroll
            for (int t = 0; t < 9; ++t) an[t] = xn[((t / 3) * WH_W + (t % 3)) * 64];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv, acc[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          bv = bn;
#pragma unroll
          for (int t = 0; t < 9; ++t) av[t] = an[t];
        }
        xa = xn;
        yb = yn;
      }
    }
    __syncthreads();  // every wave is done with this tile's LDS image
    if (has_next) {
      commit(tile + t_step);
      __syncthreads();
    }
  }

  {
    float* out = p.slabs + ((long)blockIdx.x * ((long)gridDim.y * 4) + pair * 4 + cih * 2 + coh) * SLAB;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[(t * 16 + r) * 64 + lane] = acc[t][r];
  }
  if (do_bias) {
    __syncthreads();
    float* red = lds;
    *reinterpret_cast<f32x4*>(red + tid * 8) = bsa;
    *reinterpret_cast<f32x4*>(red + tid * 8 + 4) = bsb;
    __syncthreads();
    if (tid < 64) {  // channel tid = c8 * 8 + e lives in the threads with (tid & 7) == c8
      const int cc8 = tid >> 3, e = tid & 7;
      float s = 0.f;
      for (int k = 0; k < 32; ++k) s += red[(k * 8 + cc8) * 8 + e];
      p.bias_slabs[((long)blockIdx.x * p.cout_chunks + cq) * 64 + tid] = s;
    }
  }
}

__global__ __launch_bounds__(256, 1) void wgrad3x3_c64_full_kernel(WgradParams p) { wgrad3x3_c64_full_body(p); }
__global__ __launch_bounds__(256, 1) void wgrad3x3_c64_full_batch_kernel(WgradBatch bt) { wgrad3x3_c64_full_body(bt.job[blockIdx.z]); }
__global__ __launch_bounds__(256, 1) void wgrad3x3_c64_full_geo_kernel(WgradParams p) { wgrad3x3_c64_full_body<true>(p); }

// ------------------------------------------------------------------ bf16 matrix-core weight gradient
// Operands rounded to bf16 (RNE) as they are staged into LDS, products exact, fp32 accumulation; the bias
// gradient is summed from the unrounded fp32 dY'.  A workgroup owns all 64 ci x 64 co of one (cin chunk, cout
// chunk) pair: wave w accumulates quadrant (ci half w>>1, co half w&1) for the nine taps in nine persistent
// accumulators, all four waves walking the same staged tile, so the inputs are read from HBM once and no
// cross-wave reduction is needed.  The contraction index is the pixel, which is the SLOW index of the
// channels-last image, so both MFMA operands are fetched with the transposing LDS read ds_read_b64_tr_b16
// (4 pixels x 16 channels per 16-lane group, delivered channel-major): the LDS image stays the natural
// [pixel][64 ch] bf16 rows (128 B), a tap shift is a whole-pixel offset, and the 16-B chunk k of a pixel sits at
// slot k ^ 4*((col >> 1) & 1) so that the four pixels of a transposed block fall into four disjoint bank
// quarters.  Slab layout and second-stage reduction are shared with the fp32 kernel.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;
#define BW_PIX 128
#define BW_X_BYTES (WH_H * WH_W * BW_PIX)
#define BW_Y_BYTES (WT_H * WT_W * BW_PIX)

__device__ __forceinline__ u32x4 wg_pack_bf16x8(f32x4 a, f32x4 b) {
  bf16x8 r;
  r[0] = (__bf16)a[0]; r[1] = (__bf16)a[1]; r[2] = (__bf16)a[2]; r[3] = (__bf16)a[3];
  r[4] = (__bf16)b[0]; r[5] = (__bf16)b[1]; r[6] = (__bf16)b[2]; r[7] = (__bf16)b[3];
  return __builtin_bit_cast(u32x4, r);
}

__device__ __forceinline__ bf16x8 wg_tr_frag(const unsigned char* base, unsigned off0, unsigned off1) {
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(base + off0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(base + off1));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// X16 / Y16: x / dY is a bf16 map in HBM (same View, 2-byte elements; sisr_wgrad3x3_c64_bf16s): its 16-B pieces go to LDS
// as they are (x) or after the fp32 affine rebuild and re-rounding (dY with dy_scale / dy_shift).
// stamp (diagnostic library only, tools/bf16s_timeline.py --wgrad): per wave, the shader cycles spent in the K loops and in
// committing the next tile (wait for its prefetched loads, conversions, LDS writes, barriers), with the wave's lifetime.
template <bool X16, bool Y16>
static __device__ __forceinline__ void wgrad3x3_c64_bf16_body(const WgradParams& p, unsigned* stamp = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  unsigned char* ldx = ldsb;
  unsigned char* ldy = ldsb + BW_X_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pair = blockIdx.y;
  const int cq = pair % p.cout_chunks, cc = pair / p.cout_chunks;
  const int cih = __builtin_amdgcn_readfirstlane(wave >> 1), coh = __builtin_amdgcn_readfirstlane(wave & 1);
  const int H = p.H, W = p.W;
  const bool do_bias = p.bias_slabs && cc == 0;
  const int Cout = p.cout_chunks * 64;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f32x16){0};
  f32x4 bsa = {0.f, 0.f, 0.f, 0.f}, bsb = bsa;

  // transposed-read lane constants: lane = 16g + 4qq + pp supplies the address of block row (pixel) qq,
  // channels 4pp..4pp+3 of the 16-channel block (g & 1); K half (g >> 1) covers pixels 8(g>>1) .. +7
  unsigned aoff[3][2], yoff[2];
  {
    const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const int chunk = (g & 1) * 2 + (pp >> 1), sub = (pp & 1) * 8;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int pix = 8 * (g >> 1) + 4 * rd + qq;
      yoff[rd] = pix * BW_PIX + (((coh * 4 + chunk) ^ (((pix >> 1) & 1) << 2)) << 4) + sub;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
        aoff[kw][rd] = (pix + kw) * BW_PIX + (((cih * 4 + chunk) ^ ((((pix + kw) >> 1) & 1) << 2)) << 4) + sub;
    }
  }

  const int tiles_per_img = p.tiles_w * p.tiles_h;
  const int total = tiles_per_img * p.B;
  // Tile walk.  Workgroup s lands on XCD s % 8; give every XCD a contiguous eighth of the tiles and let its S / 8
  // workgroups sweep it together, so that vertically adjacent tiles (whose 10-row halos share two rows) are read
  // through the same L2 at about the same time.  (HBM-bound kernel: measured 316 MB read per launch with the
  // strided walk against 268 MB algorithmic.)  Falls back to the strided walk when S is not a multiple of 8.
  int t_begin = blockIdx.x, t_end = total, t_step = p.S;
  if ((p.S & 7) == 0) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per = (total + 7) >> 3;
    t_begin = xcd * per + idx;
    t_end = min(total, (xcd + 1) * per);
    t_step = p.S >> 3;
  }
  // ---- staging split into "issue the loads" and "convert + write LDS": one workgroup per CU (512 VGPRs per wave),
  // the loads of tile i+1 (38 float4 per thread) are in flight while tile i is in its K loop.  Thread (c8, pcol)
  // owns channels 8 c8 .. +7 of x halo column pcol + 1 (ten rows) and of dY column pcol (eight rows); the two edge
  // columns of the halo (10 x 2 x 8 = 160 items) go one per thread to tid < 160 -- every load is unconditional.
  struct Stage {  // a bf16 map's item is one 16-B piece (eight channels), held in [..][0] as raw bits
    f32x4 x[WH_H][X16 ? 1 : 2];
    f32x4 xe[X16 ? 1 : 2];
    f32x4 y[WT_H][Y16 ? 1 : 2];
  };
  Stage st;
  const int c8 = tid & 7, pcol = tid >> 3;
  const int eidx = tid % 160, er = eidx >> 4, eside = (eidx >> 3) & 1, ec8 = eidx & 7;
  auto unpack8 = [](f32x4 raw, f32x4& a, f32x4& b) {
    const u32x4 w = __builtin_bit_cast(u32x4, raw);
    a[0] = __uint_as_float(w[0] << 16); a[1] = __uint_as_float(w[0] & 0xffff0000u);
    a[2] = __uint_as_float(w[1] << 16); a[3] = __uint_as_float(w[1] & 0xffff0000u);
    b[0] = __uint_as_float(w[2] << 16); b[1] = __uint_as_float(w[2] & 0xffff0000u);
    b[2] = __uint_as_float(w[3] << 16); b[3] = __uint_as_float(w[3] & 0xffff0000u);
  };
  auto decode = [&](int tile, int& b, int& h0, int& w0) {
    b = tile / tiles_per_img;
    const int tr = tile - b * tiles_per_img;
    const int th = tr / p.tiles_w;
    h0 = th * WT_H;
    w0 = (tr - th * p.tiles_w) * WT_W;
  };
  auto issue = [&](int tile) {
    int b, h0, w0;
    decode(tile, b, h0, w0);
    const int gxi = min(w0 + pcol, W - 1) * (int)p.xv.sW + c8 * 8;
    const int gwe = min(max(eside ? w0 + WT_W : w0 - 1, 0), W - 1);
    if (X16) {
      const unsigned short* xb = reinterpret_cast<const unsigned short*>(p.x) + (long)b * p.xv.sB + p.xv.chunk(cc);
#pragma unroll
      for (int r = 0; r < WH_H; ++r)
        st.x[r][0] = *reinterpret_cast<const f32x4*>(xb + (long)min(max(h0 - 1 + r, 0), H - 1) * p.xv.sH + gxi);
      st.xe[0] = *reinterpret_cast<const f32x4*>(xb + (long)min(max(h0 - 1 + er, 0), H - 1) * p.xv.sH + gwe * (int)p.xv.sW + ec8 * 8);
    } else {
      const float* xb = p.x + (long)b * p.xv.sB + p.xv.chunk(cc);
#pragma unroll
      for (int r = 0; r < WH_H; ++r) {
        const float* a = xb + (long)min(max(h0 - 1 + r, 0), H - 1) * p.xv.sH + gxi;
        st.x[r][0] = *reinterpret_cast<const f32x4*>(a);
        st.x[r][X16 ? 0 : 1] = *reinterpret_cast<const f32x4*>(a + 4);
      }
      const float* e = xb + (long)min(max(h0 - 1 + er, 0), H - 1) * p.xv.sH + gwe * (int)p.xv.sW + ec8 * 8;
      st.xe[0] = *reinterpret_cast<const f32x4*>(e);
      st.xe[X16 ? 0 : 1] = *reinterpret_cast<const f32x4*>(e + 4);
    }
    const int gyi = min(w0 + pcol, W - 1) * (int)p.yv.sW + c8 * 8;
    if (Y16) {
      const unsigned short* yb = reinterpret_cast<const unsigned short*>(p.dy) + (long)b * p.yv.sB + p.yv.chunk(cq);
#pragma unroll
      for (int r = 0; r < WT_H; ++r) st.y[r][0] = *reinterpret_cast<const f32x4*>(yb + (long)min(h0 + r, H - 1) * p.yv.sH + gyi);
    } else {
      const float* yb = p.dy + (long)b * p.yv.sB + p.yv.chunk(cq);
#pragma unroll
      for (int r = 0; r < WT_H; ++r) {
        const float* a = yb + (long)min(h0 + r, H - 1) * p.yv.sH + gyi;
        st.y[r][0] = *reinterpret_cast<const f32x4*>(a);
        st.y[r][Y16 ? 0 : 1] = *reinterpret_cast<const f32x4*>(a + 4);
      }
    }
  };
  auto commit = [&](int tile) {
    int b, h0, w0;
    decode(tile, b, h0, w0);
    const int col = pcol + 1;
    const bool cokx = w0 + pcol < W;
    const unsigned lx = col * BW_PIX + ((c8 ^ (((col >> 1) & 1) << 2)) << 4);
#pragma unroll
    for (int r = 0; r < WH_H; ++r) {
      const int gh = h0 - 1 + r;
      u32x4 pk = X16 ? __builtin_bit_cast(u32x4, st.x[r][0]) : wg_pack_bf16x8(st.x[r][0], st.x[r][X16 ? 0 : 1]);
      const unsigned m = (gh >= 0 && gh < H && cokx) ? 0xffffffffu : 0u;
      pk &= (u32x4){m, m, m, m};
      *reinterpret_cast<u32x4*>(ldx + r * (WH_W * BW_PIX) + lx) = pk;
    }
    {
      const int ecol = eside ? WH_W - 1 : 0, gwe = eside ? w0 + WT_W : w0 - 1, ghe = h0 - 1 + er;
      u32x4 pk = X16 ? __builtin_bit_cast(u32x4, st.xe[0]) : wg_pack_bf16x8(st.xe[0], st.xe[X16 ? 0 : 1]);
      const unsigned m = (ghe >= 0 && ghe < H && gwe >= 0 && gwe < W) ? 0xffffffffu : 0u;
      pk &= (u32x4){m, m, m, m};
      if (tid < 160)
        *reinterpret_cast<u32x4*>(ldx + er * (WH_W * BW_PIX) + ecol * BW_PIX + ((ec8 ^ (((ecol >> 1) & 1) << 2)) << 4)) = pk;
    }
    f32x4 s4a = {1.f, 1.f, 1.f, 1.f}, s4b = s4a, t4a = {0.f, 0.f, 0.f, 0.f}, t4b = t4a;
    if (p.dy_scale) {
      const float* sp = p.dy_scale + (long)b * Cout + cq * 64 + c8 * 8;
      s4a = *reinterpret_cast<const f32x4*>(sp);
      s4b = *reinterpret_cast<const f32x4*>(sp + 4);
    }
    if (p.dy_shift) {
      const float* tp = p.dy_shift + (long)b * Cout + cq * 64 + c8 * 8;
      t4a = *reinterpret_cast<const f32x4*>(tp);
      t4b = *reinterpret_cast<const f32x4*>(tp + 4);
    }
    const bool coky = w0 + pcol < W;
    const unsigned ly = pcol * BW_PIX + ((c8 ^ (((pcol >> 1) & 1) << 2)) << 4);
#pragma unroll
    for (int r = 0; r < WT_H; ++r) {
      const bool ok = coky && (h0 + r < H);
      f32x4 ya, yb2;
      if (Y16) unpack8(st.y[r][0], ya, yb2);
      else { ya = st.y[r][0]; yb2 = st.y[r][Y16 ? 0 : 1]; }
      const f32x4 ta = sisr_keep_if(ya * s4a + t4a, ok), tb = sisr_keep_if(yb2 * s4b + t4b, ok);
      bsa += ta;
      bsb += tb;
      *reinterpret_cast<u32x4*>(ldy + r * (WT_W * BW_PIX) + ly) = wg_pack_bf16x8(ta, tb);
    }
  };

  unsigned long long st_life = 0, st_a = 0, st_k = 0, st_c = 0;
  unsigned st_tiles = 0;
  if (stamp) st_life = __builtin_amdgcn_s_memtime();
  if (t_begin < t_end) {
    issue(t_begin);
    commit(t_begin);
  }
  __syncthreads();
  for (int tile = t_begin; tile < t_end; tile += t_step) {
    const bool has_next = tile + t_step < t_end;  // uniform
    if (stamp) st_a = __builtin_amdgcn_s_memtime();
    if (has_next) issue(tile + t_step);

    // ---- 16 K-steps of 16 pixels (tile row r = ks >> 1, half hf = ks & 1), nine taps each.  One wave per SIMD: nobody
    // else covers an LDS round trip, so the ten fragments of K-step ks + 1 are requested before the nine MFMAs of step ks
    // (two fragment sets, 80 VGPRs; the accumulation order of every tap is unchanged: same bits).
    {
      bf16x8 af[2][9], bf[2];
      auto load_step = [&](int ks, bf16x8* a9, bf16x8& b1) {
        const int r = ks >> 1, hf = ks & 1;
        b1 = wg_tr_frag(ldy + (r * WT_W + 16 * hf) * BW_PIX, yoff[0], yoff[1]);
#pragma unroll
        for (int t = 0; t < 9; ++t)
          a9[t] = wg_tr_frag(ldx + ((r + t / 3) * WH_W + 16 * hf) * BW_PIX, aoff[t % 3][0], aoff[t % 3][1]);
      };
      load_step(0, af[0], bf[0]);
#pragma unroll
      for (int ks = 0; ks < 2 * WT_H; ++ks) {
        if (ks + 1 < 2 * WT_H) load_step(ks + 1, af[(ks + 1) & 1], bf[(ks + 1) & 1]);
#pragma unroll
        for (int t = 0; t < 9; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][t], bf[ks & 1], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);  // one step's MFMAs stay together behind the next step's fragment requests
      }
    }
    if (stamp) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      st_k += t - st_a;
      st_a = t;
      ++st_tiles;
    }
    __syncthreads();  // every wave is done with this tile's LDS image
    if (has_next) {
      commit(tile + t_step);
      __syncthreads();
    }
    if (stamp) st_c += __builtin_amdgcn_s_memtime() - st_a;
  }
  if (stamp && lane == 0) {
    unsigned* dbg = stamp + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
    dbg[0] = (unsigned)(__builtin_amdgcn_s_memtime() - st_life);
    dbg[1] = (unsigned)st_k;
    dbg[2] = 0;
    dbg[3] = (unsigned)st_c;
    dbg[4] = st_tiles;
    dbg[5] = __builtin_amdgcn_s_getreg(4 | (31 << 11));
    dbg[6] = __builtin_amdgcn_s_getreg(20 | (31 << 11));
    dbg[7] = (unsigned)(st_life & 0xffffffffu);
  }

  // ---- slabs: unit = pair*4 + (ci half, co half), the fp32 kernel's layout
  {
    float* out = p.slabs + ((long)blockIdx.x * ((long)gridDim.y * 4) + pair * 4 + cih * 2 + coh) * SLAB;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[(t * 16 + r) * 64 + lane] = acc[t][r];
  }
  if (do_bias) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(ldsb);
    *reinterpret_cast<f32x4*>(red + tid * 8) = bsa;
    *reinterpret_cast<f32x4*>(red + tid * 8 + 4) = bsb;
    __syncthreads();
    if (tid < 64) {  // channel tid = c8*8 + e lives in threads with (tid & 7) == c8
      const int c8 = tid >> 3, e = tid & 7;
      float s = 0.f;
      for (int k = 0; k < 32; ++k) s += red[(k * 8 + c8) * 8 + e];
      p.bias_slabs[((long)blockIdx.x * p.cout_chunks + cq) * 64 + tid] = s;
    }
  }
}
__global__ __launch_bounds__(256, 1) void wgrad3x3_c64_bf16_kernel(WgradParams p) { wgrad3x3_c64_bf16_body<false, false>(p); }
__global__ __launch_bounds__(256, 1) void wgrad3x3_c64_bf16_x16_kernel(WgradParams p) { wgrad3x3_c64_bf16_body<true, false>(p); }
__global__ __launch_bounds__(256, 1) void wgrad3x3_c64_bf16_xy16_kernel(WgradParams p) { wgrad3x3_c64_bf16_body<true, true>(p); }
#ifdef SISR_DIAG
static unsigned* g_diag_wgrad_stamp = nullptr;  // diagnostic library only (the product library keeps no state)
extern "C" void sisr_diag_wgrad_stamp(void* buf) { g_diag_wgrad_stamp = static_cast<unsigned*>(buf); }
__global__ __launch_bounds__(256, 1) void wgrad3x3_c64_bf16_xy16_stamp_kernel(WgradParams p, unsigned* stamp) {
  wgrad3x3_c64_bf16_body<true, true>(p, stamp);
}
#endif

// ------------------------------------------------------------------ bf16x3 weight gradient
// The bf16 kernel above with every fp32 operand split exactly into three bf16 numbers (conv3x3_mfma.hip, "bf16x3") and six
// products per tap and 16-pixel block; 4 x 32 pixel tiles so that the three planes of the x halo (78 KB) and of dY (49 KB)
// fit the 160 KB of LDS with one persistent workgroup per CU.  The six products of a tap go into ONE accumulator: its
// chain is K / 16 x 6 accumulations long against K / 2 in the fp32 kernel, so the rounding of the small terms against the
// running sum stays below the fp32 kernel's own accumulation error.
#define XW_TH 4
#define XW_HH (XW_TH + 2)
#define XW_X (XW_HH * WH_W * BW_PIX)
#define XW_Y (XW_TH * WT_W * BW_PIX)

__device__ __forceinline__ void wg_store_split3(unsigned char* dst, int plane, f32x4 a, f32x4 b, unsigned mask) {
  bf16x8 h, m, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = e < 4 ? a[e] : b[e - 4];
    const __bf16 vh = (__bf16)v;
    const float r1 = v - (float)vh;
    const __bf16 vm = (__bf16)r1;
    h[e] = vh;
    m[e] = vm;
    l[e] = (__bf16)(r1 - (float)vm);
  }
  const u32x4 mk = {mask, mask, mask, mask};
  *reinterpret_cast<u32x4*>(dst) = __builtin_bit_cast(u32x4, h) & mk;
  *reinterpret_cast<u32x4*>(dst + plane) = __builtin_bit_cast(u32x4, m) & mk;
  *reinterpret_cast<u32x4*>(dst + 2 * plane) = __builtin_bit_cast(u32x4, l) & mk;
}

__global__ __launch_bounds__(256, 1) void wgrad3x3_c64_x3_kernel(WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  unsigned char* ldx = ldsb;                 // three planes (hi, mid, lo) of XW_X bytes
  unsigned char* ldy = ldsb + 3 * XW_X;      // three planes of XW_Y bytes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pair = blockIdx.y;
  const int cq = pair % p.cout_chunks, cc = pair / p.cout_chunks;
  const int cih = __builtin_amdgcn_readfirstlane(wave >> 1), coh = __builtin_amdgcn_readfirstlane(wave & 1);
  const int H = p.H, W = p.W;
  const bool do_bias = p.bias_slabs && cc == 0;
  const int Cout = p.cout_chunks * 64;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f32x16){0};
  f32x4 bsa = {0.f, 0.f, 0.f, 0.f}, bsb = bsa;

  // transposed-read lane constants: lane = 16g + 4qq + pp supplies the address of block row (pixel) qq,
  // channels 4pp..4pp+3 of the 16-channel block (g & 1); K half (g >> 1) covers pixels 8(g>>1) .. +7
  unsigned aoff[3][2], yoff[2];
  {
    const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const int chunk = (g & 1) * 2 + (pp >> 1), sub = (pp & 1) * 8;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int pix = 8 * (g >> 1) + 4 * rd + qq;
      yoff[rd] = pix * BW_PIX + (((coh * 4 + chunk) ^ (((pix >> 1) & 1) << 2)) << 4) + sub;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
        aoff[kw][rd] = (pix + kw) * BW_PIX + (((cih * 4 + chunk) ^ ((((pix + kw) >> 1) & 1) << 2)) << 4) + sub;
    }
  }

  const int tiles_per_img = p.tiles_w * p.tiles_h;
  const int total = tiles_per_img * p.B;
  // Tile walk.  Workgroup s lands on XCD s % 8; give every XCD a contiguous eighth of the tiles and let its S / 8
  // workgroups sweep it together, so that vertically adjacent tiles (whose 10-row halos share two rows) are read
  // through the same L2 at about the same time.  (HBM-bound kernel: measured 316 MB read per launch with the
  // strided walk against 268 MB algorithmic.)  Falls back to the strided walk when S is not a multiple of 8.
  int t_begin = blockIdx.x, t_end = total, t_step = p.S;
  if ((p.S & 7) == 0) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per = (total + 7) >> 3;
    t_begin = xcd * per + idx;
    t_end = min(total, (xcd + 1) * per);
    t_step = p.S >> 3;
  }
  // ---- staging split into "issue the loads" and "convert + write LDS": one workgroup per CU (512 VGPRs per wave),
  // the loads of tile i+1 (38 float4 per thread) are in flight while tile i is in its K loop.  Thread (c8, pcol)
  // owns channels 8 c8 .. +7 of x halo column pcol + 1 (ten rows) and of dY column pcol (eight rows); the two edge
  // columns of the halo (10 x 2 x 8 = 160 items) go one per thread to tid < 160 -- every load is unconditional.
  struct Stage {
    f32x4 x[XW_HH][2];
    f32x4 xe[2];
    f32x4 y[XW_TH][2];
  };
  Stage st;
  const int c8 = tid & 7, pcol = tid >> 3;
  const int eidx = tid % (XW_HH * 16), er = eidx >> 4, eside = (eidx >> 3) & 1, ec8 = eidx & 7;
  auto decode = [&](int tile, int& b, int& h0, int& w0) {
    b = tile / tiles_per_img;
    const int tr = tile - b * tiles_per_img;
    const int th = tr / p.tiles_w;
    h0 = th * XW_TH;
    w0 = (tr - th * p.tiles_w) * WT_W;
  };
  auto issue = [&](int tile) {
    int b, h0, w0;
    decode(tile, b, h0, w0);
    const float* xb = p.x + (long)b * p.xv.sB + p.xv.chunk(cc);
    const int gxi = min(w0 + pcol, W - 1) * (int)p.xv.sW + c8 * 8;
#pragma unroll
    for (int r = 0; r < XW_HH; ++r) {
      const float* a = xb + (long)min(max(h0 - 1 + r, 0), H - 1) * p.xv.sH + gxi;
      st.x[r][0] = *reinterpret_cast<const f32x4*>(a);
      st.x[r][1] = *reinterpret_cast<const f32x4*>(a + 4);
    }
    {
      const int gwe = min(max(eside ? w0 + WT_W : w0 - 1, 0), W - 1);
      const float* e = xb + (long)min(max(h0 - 1 + er, 0), H - 1) * p.xv.sH + gwe * (int)p.xv.sW + ec8 * 8;
      st.xe[0] = *reinterpret_cast<const f32x4*>(e);
      st.xe[1] = *reinterpret_cast<const f32x4*>(e + 4);
    }
    const float* yb = p.dy + (long)b * p.yv.sB + p.yv.chunk(cq);
    const int gyi = min(w0 + pcol, W - 1) * (int)p.yv.sW + c8 * 8;
#pragma unroll
    for (int r = 0; r < XW_TH; ++r) {
      const float* a = yb + (long)min(h0 + r, H - 1) * p.yv.sH + gyi;
      st.y[r][0] = *reinterpret_cast<const f32x4*>(a);
      st.y[r][1] = *reinterpret_cast<const f32x4*>(a + 4);
    }
  };
  auto commit = [&](int tile) {
    int b, h0, w0;
    decode(tile, b, h0, w0);
    const int col = pcol + 1;
    const bool cokx = w0 + pcol < W;
    const unsigned lx = col * BW_PIX + ((c8 ^ (((col >> 1) & 1) << 2)) << 4);
#pragma unroll
    for (int r = 0; r < XW_HH; ++r) {
      const int gh = h0 - 1 + r;
      const unsigned m = (gh >= 0 && gh < H && cokx) ? 0xffffffffu : 0u;
      wg_store_split3(ldx + r * (WH_W * BW_PIX) + lx, XW_X, st.x[r][0], st.x[r][1], m);
    }
    {
      const int ecol = eside ? WH_W - 1 : 0, gwe = eside ? w0 + WT_W : w0 - 1, ghe = h0 - 1 + er;
      const unsigned m = (ghe >= 0 && ghe < H && gwe >= 0 && gwe < W) ? 0xffffffffu : 0u;
      if (tid < XW_HH * 16)
        wg_store_split3(ldx + er * (WH_W * BW_PIX) + ecol * BW_PIX + ((ec8 ^ (((ecol >> 1) & 1) << 2)) << 4), XW_X, st.xe[0], st.xe[1], m);
    }
    f32x4 s4a = {1.f, 1.f, 1.f, 1.f}, s4b = s4a, t4a = {0.f, 0.f, 0.f, 0.f}, t4b = t4a;
    if (p.dy_scale) {
      const float* sp = p.dy_scale + (long)b * Cout + cq * 64 + c8 * 8;
      s4a = *reinterpret_cast<const f32x4*>(sp);
      s4b = *reinterpret_cast<const f32x4*>(sp + 4);
    }
    if (p.dy_shift) {
      const float* tp = p.dy_shift + (long)b * Cout + cq * 64 + c8 * 8;
      t4a = *reinterpret_cast<const f32x4*>(tp);
      t4b = *reinterpret_cast<const f32x4*>(tp + 4);
    }
    const bool coky = w0 + pcol < W;
    const unsigned ly = pcol * BW_PIX + ((c8 ^ (((pcol >> 1) & 1) << 2)) << 4);
#pragma unroll
    for (int r = 0; r < XW_TH; ++r) {
      const bool ok = coky && (h0 + r < H);
      const f32x4 ta = sisr_keep_if(st.y[r][0] * s4a + t4a, ok), tb = sisr_keep_if(st.y[r][1] * s4b + t4b, ok);
      bsa += ta;
      bsb += tb;
      wg_store_split3(ldy + r * (WT_W * BW_PIX) + ly, XW_Y, ta, tb, 0xffffffffu);
    }
  };

  if (t_begin < t_end) {
    issue(t_begin);
    commit(t_begin);
  }
  __syncthreads();
  for (int tile = t_begin; tile < t_end; tile += t_step) {
    const bool has_next = tile + t_step < t_end;  // uniform
    if (has_next) issue(tile + t_step);

    // ---- 8 K-steps of 16 pixels (tile row r, half hf), nine taps each, six products per tap:
    // hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid (A = x planes, B = dY planes)
#pragma unroll 1
    for (int r = 0; r < XW_TH; ++r) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const unsigned char* yb = ldy + (r * WT_W + 16 * hf) * BW_PIX;
        const bf16x8 bh = wg_tr_frag(yb, yoff[0], yoff[1]);
        const bf16x8 bm = wg_tr_frag(yb + XW_Y, yoff[0], yoff[1]);
        const bf16x8 bl = wg_tr_frag(yb + 2 * XW_Y, yoff[0], yoff[1]);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const unsigned char* xa = ldx + ((r + t / 3) * WH_W + 16 * hf) * BW_PIX;
          const bf16x8 ah = wg_tr_frag(xa, aoff[t % 3][0], aoff[t % 3][1]);
          const bf16x8 am = wg_tr_frag(xa + XW_X, aoff[t % 3][0], aoff[t % 3][1]);
          const bf16x8 al = wg_tr_frag(xa + 2 * XW_X, aoff[t % 3][0], aoff[t % 3][1]);
          f32x16 c = acc[t];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);  // small terms first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
          acc[t] = c;
          __builtin_amdgcn_sched_barrier(0);  // one tap's fragments in flight at a time
        }
      }
    }
    __syncthreads();  // every wave is done with this tile's LDS image
    if (has_next) {
      commit(tile + t_step);
      __syncthreads();
    }
  }

  // ---- slabs: unit = pair*4 + (ci half, co half), the fp32 kernel's layout
  {
    float* out = p.slabs + ((long)blockIdx.x * ((long)gridDim.y * 4) + pair * 4 + cih * 2 + coh) * SLAB;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[(t * 16 + r) * 64 + lane] = acc[t][r];
  }
  if (do_bias) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(ldsb);
    *reinterpret_cast<f32x4*>(red + tid * 8) = bsa;
    *reinterpret_cast<f32x4*>(red + tid * 8 + 4) = bsb;
    __syncthreads();
    if (tid < 64) {  // channel tid = c8*8 + e lives in threads with (tid & 7) == c8
      const int c8 = tid >> 3, e = tid & 7;
      float s = 0.f;
      for (int k = 0; k < 32; ++k) s += red[(k * 8 + c8) * 8 + e];
      p.bias_slabs[((long)blockIdx.x * p.cout_chunks + cq) * 64 + tid] = s;
    }
  }
}

// Sum S slabs per output element in slab order and scatter to dW (generic strides / channel maps):
// element (unit, tap t, reg r, lane l): ci_local = (r&3) + 8*(r>>2) + 4*(l>>5), co_local = l&31.
struct ReduceParams {
  const float* slabs;
  const float* bias_slabs;
  float* dw;
  float* db;
  long so, si;
  float alpha;
  int S, units, cin_chunks, cout_chunks, flip, on, oq, in_, iq, bias_n, bias_q;
  int mapped;                             // != 0: slab row r holds unit unit_map[r] (launch with an active-unit mask)
  unsigned char unit_map[WG_MAX_UNITS];
  int o_lim, i_lim;                       // > 0: dW has only that many output / input channels (the rest of a 64-chunk is
                                          // zero padding of the caller's maps and is not written)
};

// blockDim = (64, RG): x = a run of FOUR consecutive elements (one 16-byte load per slab; 256 elements per workgroup), y = slab
// group (S split RG ways, up to 8 loads in flight per thread; with RG = 16 the 128 slabs of a 64 -> 64 gradient are one round
// of loads); the group sums are added in group order through LDS -- per element the same order as summing slab by slab
// within a group, then the groups.  Bound by the slabs' bytes and the memory latency (37.8 MB per 64 -> 64 gradient of the
// one-workgroup-per-CU kernel; dword loads needed four times the instructions: 15.7 -> ~9 us for a batch of eight at 4 tiles).
#define RG 16
static __device__ __forceinline__ void wgrad_reduce_body(const ReduceParams& p) {
  __shared__ __attribute__((aligned(16))) float red[RG][64][4];
  const long total = (long)p.units * SLAB;
  const long gid = ((long)blockIdx.x * 64 + threadIdx.x) * 4;
  const int grp = threadIdx.y;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const bool is_w = gid < total;
  const long j = gid - total;
  const bool is_b = !is_w && p.db && j < (long)p.cout_chunks * 64;
  if (is_w || is_b) {
    const float* src = is_w ? p.slabs + gid : p.bias_slabs + j;
    const long stride = is_w ? (long)p.units * SLAB : (long)p.cout_chunks * 64;
    int k = grp;
    for (; k + 7 * RG < p.S; k += 8 * RG) {
      f32x4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const f32x4*>(src + (long)(k + RG * u) * stride);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; k < p.S; k += RG) s += *reinterpret_cast<const f32x4*>(src + (long)k * stride);
  }
  *reinterpret_cast<f32x4*>(red[grp][threadIdx.x]) = s;
  __syncthreads();
  if (grp != 0) return;
  s = *reinterpret_cast<const f32x4*>(red[0][threadIdx.x]);
#pragma unroll
  for (int g = 1; g < RG; ++g) s += *reinterpret_cast<const f32x4*>(red[g][threadIdx.x]);
#pragma unroll
  for (int e4 = 0; e4 < 4; ++e4) {
    if (is_w) {
      const long ge = gid + e4;
      const int urow = (int)(ge / SLAB);
      const int e = (int)(ge - (long)urow * SLAB);
      const int unit = p.mapped ? p.unit_map[urow] : urow;
      const int l = e & 63, r = (e >> 6) & 15, t = e >> 10;
      const int quad = unit & 3, pair = unit >> 2;
      const int cq = pair % p.cout_chunks, cc = pair / p.cout_chunks;
      const int ci = (quad >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      const int co = (quad & 1) * 32 + (l & 31);
      const long o = (long)co * p.on + (long)cq * p.oq;
      const long ii = (long)ci * p.in_ + (long)cc * p.iq;
      if ((p.o_lim > 0 && o >= p.o_lim) || (p.i_lim > 0 && ii >= p.i_lim)) continue;
      p.dw[o * p.so + ii * p.si + (p.flip ? 8 - t : t)] = s[e4] * p.alpha;
    } else if (is_b) {
      const long je = j + e4;
      const int cq = (int)(je >> 6), co = (int)(je & 63);
      const long ob = (long)co * p.bias_n + (long)cq * p.bias_q;
      if (p.o_lim > 0 && ob >= p.o_lim) continue;
      p.db[ob] = s[e4] * p.alpha;
    }
  }
}

__global__ __launch_bounds__(64 * RG) void wgrad_reduce_kernel(ReduceParams p) { wgrad_reduce_body(p); }
struct ReduceBatch {
  ReduceParams job[WG_BATCH];
};
__global__ __launch_bounds__(64 * RG) void wgrad_reduce_batch_kernel(ReduceBatch bt) { wgrad_reduce_body(bt.job[blockIdx.y]); }

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

static int wgrad_split(int B, int H, int W, int units) {
  const long tiles = (long)B * ((H + WT_H - 1) / WT_H) * ((W + WT_W - 1) / WT_W);
  long S = 512 / units;  // two workgroups per CU resident across the whole grid
#ifdef SISR_DIAG
  if (const char* e = getenv("SISR_DIAG_WGRAD_WGS")) S = atol(e) / units;  // A/B of the K-split (tools/kbench.py)
#endif
  if (S < 1) S = 1;
  if (S > tiles) S = tiles;
  return (int)S;
}

// K-split of the dense form (wgrad3x3_c64_full_kernel): one workgroup per CU over all chunk pairs
static int wgrad_full_split(int B, int H, int W, int pairs) {
  const long tiles = (long)B * ((H + WT_H - 1) / WT_H) * ((W + WT_W - 1) / WT_W);
  long S = 256 / pairs;
  if (S < 1) S = 1;
  if (S > tiles) S = tiles;
  return (int)S;
}

extern "C" size_t sisr_wgrad3x3_c64_workspace_bytes(int B, int H, int W, int cin, int cout) {
  if (B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0 || (cin & 63) || (cout & 63)) return 0;
  const int units = (cin / 64) * (cout / 64) * 4;
  const int S = wgrad_split(B, H, W, units);
  const int Sf = wgrad_full_split(B, H, W, units / 4);
  // slabs for the larger of the two forms (dense: one workgroup per CU, four quadrant slabs each; masked: the K-slices go
  // to the active units, at most 512 workgroups); bias slabs sized for the finest split (512)
  size_t slabs = (size_t)(S * units > 512 ? S * units : 512);
  if ((size_t)Sf * units > slabs) slabs = (size_t)Sf * units;
  return (slabs * SLAB + (size_t)512 * cout) * sizeof(float);
}

static int wgrad_fp32_launch(const float* x, const int64_t* xview, const float* dy, const int64_t* dyview,
                             const float* dy_scale, const float* dy_shift, float alpha, float* dw, int64_t so,
                             int64_t si, int flip_taps, int out_perm_n, int out_perm_q, int in_perm_n,
                             int in_perm_q, float* dbias, int bias_n, int bias_q, float* workspace,
                             size_t workspace_bytes, int B, int H, int W, int cin, int cout,
                             unsigned long long active_units, int geo, int geo_up, int o_lim, int i_lim, void* stream);

extern "C" int sisr_wgrad3x3_c64(const float* x, const int64_t* xview, const float* dy, const int64_t* dyview,
                                 const float* dy_scale, const float* dy_shift, float alpha, float* dw, int64_t so,
                                 int64_t si, int flip_taps, int out_perm_n, int out_perm_q, int in_perm_n,
                                 int in_perm_q, float* dbias, int bias_n, int bias_q, float* workspace,
                                 size_t workspace_bytes, int B, int H, int W, int cin, int cout,
                                 unsigned long long active_units, void* stream) {
  return wgrad_fp32_launch(x, xview, dy, dyview, dy_scale, dy_shift, alpha, dw, so, si, flip_taps, out_perm_n, out_perm_q,
                           in_perm_n, in_perm_q, dbias, bias_n, bias_q, workspace, workspace_bytes, B, H, W, cin, cout,
                           active_units, 0, 0, 0, 0, stream);
}

// Weight gradient of SPARNet's ConvLayer conv (see sisr_conv3x3_c64_geo mode 1): x [B][H >> up][W >> up][cin] read through
// nearest upsampling and ReflectionPad2d(1), dy [B][H][W][cout]; dw [co_real][ci_real][3][3] plain OIHW (so = ci_real * 9,
// si = 9), rows / columns >= co_real / ci_real of the 64-chunks are not written; db [co_real] or null.  active_units as in
// sisr_wgrad3x3_c64 (a caller whose maps carry zero-padded channels masks the 32 x 32 blocks that only hold padding).
extern "C" int sisr_wgrad3x3_c64_geo(const float* x, const int64_t* xview, const float* dy, const int64_t* dyview, float* dw,
                                     int co_real, int ci_real, float* dbias, float* workspace, size_t workspace_bytes, int B, int H,
                                     int W, int cin, int cout, int up, unsigned long long active_units, void* stream) {
  // up: bit 0 = x stored at half size (nearest x2 read in place); bit 1 = dy stored at half size, ((H + 1) / 2, (W + 1) / 2), and
  // read zero-stuffed (the stride-2 ConvLayer: dy of the strided conv seen on the stride-1 grid)
  if (co_real <= 0 || ci_real <= 0 || co_real > cout || ci_real > cin || up < 0 || up > 2 || H < 2 || W < 2 ||
      ((up & 1) && ((H | W) & 1)))
    return SISR_ERR_ARG;
  return wgrad_fp32_launch(x, xview, dy, dyview, nullptr, nullptr, 1.f, dw, (int64_t)ci_real * 9, 9, 0, 1, 64, 1, 64, dbias, 1, 64,
                           workspace, workspace_bytes, B, H, W, cin, cout, active_units, 1, up, co_real, ci_real, stream);
}

static int wgrad_fp32_launch(const float* x, const int64_t* xview, const float* dy, const int64_t* dyview,
                             const float* dy_scale, const float* dy_shift, float alpha, float* dw, int64_t so,
                             int64_t si, int flip_taps, int out_perm_n, int out_perm_q, int in_perm_n,
                             int in_perm_q, float* dbias, int bias_n, int bias_q, float* workspace,
                             size_t workspace_bytes, int B, int H, int W, int cin, int cout,
                             unsigned long long active_units, int geo, int geo_up, int o_lim, int i_lim, void* stream) {
  if (!x || !dy || !dw || !xview || !dyview || !workspace || B <= 0 || H <= 0 || W <= 0) return SISR_ERR_ARG;
  if ((cin & 63) || (cout & 63) || cin <= 0 || cout <= 0) return SISR_ERR_UNSUPPORTED;
  if (workspace_bytes < sisr_wgrad3x3_c64_workspace_bytes(B, H, W, cin, cout)) return SISR_ERR_ARG;
  if (!sisr_aligned16(x) || !sisr_aligned16(dy) || !sisr_aligned16(workspace) || !sisr_aligned16(dy_scale) ||
      !sisr_aligned16(dy_shift))
    return SISR_ERR_ALIGN;
  WgradParams p = {};
  p.x = x;
  p.xv = view_from(xview);
  p.dy = dy;
  p.yv = view_from(dyview);
  if ((p.xv.sB | p.xv.sH | p.xv.sW | p.xv.chi | p.xv.clo | p.yv.sB | p.yv.sH | p.yv.sW | p.yv.chi | p.yv.clo) & 3)
    return SISR_ERR_ALIGN;
  p.dy_scale = dy_scale;
  p.dy_shift = dy_shift;
  p.B = B;
  p.H = H;
  p.W = W;
  p.cin_chunks = cin / 64;
  p.cout_chunks = cout / 64;
  p.tiles_w = (W + WT_W - 1) / WT_W;
  p.tiles_h = (H + WT_H - 1) / WT_H;
  p.geo_reflect = geo;
  p.geo_up = geo_up & 1;
  p.geo_ysub = (geo_up >> 1) & 1;
  // Units = (cin chunk, cout chunk, ci half, co half) blocks of the gradient, bit ((cc * cout_chunks + cq) * 4 + cih * 2 + coh)
  // of active_units (0 = all).  A caller whose weight is structurally sparse (SFTMD's merged convs) masks the blocks it
  // never reads: they are neither computed nor written, and the K-slices of the launch go to the rest.
  const int all_units = p.cin_chunks * p.cout_chunks * 4;
  if (active_units && all_units > WG_MAX_UNITS) return SISR_ERR_UNSUPPORTED;
  ReduceParams r = {};
  int units = all_units;
  p.bias_units = 0;
  if (active_units) {
    units = 0;
    unsigned seen_bias = 0;  // bit (cq * 2 + coh): a unit already sums this bias half
    for (int u = 0; u < all_units; ++u) {
      if (!((active_units >> u) & 1)) continue;
      p.unit_map[units] = r.unit_map[units] = (unsigned char)u;
      const int cq = (u >> 2) % p.cout_chunks, coh = u & 1;
      if (!((seen_bias >> (cq * 2 + coh)) & 1)) {
        p.bias_units |= 1ull << units;
        seen_bias |= 1u << (cq * 2 + coh);
      }
      ++units;
    }
    if (units == 0 || (all_units < 64 && (active_units >> all_units) != 0)) return SISR_ERR_ARG;
    unsigned need_bias = (1u << (p.cout_chunks * 2)) - 1;  // every (cout chunk, co half) ...
    if (o_lim > 0) need_bias = (1u << ((o_lim + 31) / 32)) - 1;  // ... that holds channels the caller reads
    if (dbias && (seen_bias & need_bias) != need_bias) return SISR_ERR_ARG;  // a bias half nobody would sum
  }
  p.mapped = active_units != 0;
  r.mapped = active_units != 0;
  // The dense form gives a whole tile (8 rows x 32 pixels, all four channel quadrants: 31 us of MFMA work on one CU) to one
  // workgroup.  A launch with fewer tiles than half the CUs (SPARNet's maps of 4^2 .. 32^2 pixels) leaves the chip idle for
  // that long; the quadrant form splits each tile over four workgroups: 62 -> ~35 us per launch there.
  const long tiles_all = (long)B * p.tiles_h * p.tiles_w * (units / 4);
  const bool dense = !active_units && !getenv("SISR_WGRAD_QUADRANT_KERNEL") && tiles_all >= 128;  // (env: A/B of the forms)
  p.S = dense ? wgrad_full_split(B, H, W, units / 4) : wgrad_split(B, H, W, units);
  p.slabs = workspace;
  p.bias_slabs = dbias ? workspace + (size_t)p.S * units * SLAB : nullptr;
  if (dense) {  // one persistent workgroup per CU owning all four quadrants of its chunk pair
    const size_t lds_full = (size_t)(FW_X + FW_Y) * sizeof(float);
    if (geo) {
      SISR_ALLOW_LDS(wgrad3x3_c64_full_geo_kernel, lds_full);
      hipLaunchKernelGGL(wgrad3x3_c64_full_geo_kernel, dim3(p.S, units / 4), dim3(256), lds_full, (hipStream_t)stream, p);
    } else {
      SISR_ALLOW_LDS(wgrad3x3_c64_full_kernel, lds_full);
      hipLaunchKernelGGL(wgrad3x3_c64_full_kernel, dim3(p.S, units / 4), dim3(256), lds_full, (hipStream_t)stream, p);
    }
  } else {
    const size_t lds_bytes = (size_t)(LDS_X + LDS_Y) * sizeof(float);
    if (geo) {
      SISR_ALLOW_LDS(wgrad3x3_c64_geo_kernel, lds_bytes);
      hipLaunchKernelGGL(wgrad3x3_c64_geo_kernel, dim3(p.S, units), dim3(256), lds_bytes, (hipStream_t)stream, p);
    } else {
      SISR_ALLOW_LDS(wgrad3x3_c64_kernel, lds_bytes);
      hipLaunchKernelGGL(wgrad3x3_c64_kernel, dim3(p.S, units), dim3(256), lds_bytes, (hipStream_t)stream, p);
    }
  }
  int rc = sisr_check_launch();
  if (rc) return rc;
  r.slabs = p.slabs;
  r.bias_slabs = p.bias_slabs;
  r.dw = dw;
  r.db = dbias;
  r.so = so;
  r.si = si;
  r.alpha = alpha;
  r.S = p.S;
  r.units = units;
  r.cin_chunks = p.cin_chunks;
  r.cout_chunks = p.cout_chunks;
  r.flip = flip_taps;
  r.on = out_perm_n;
  r.oq = out_perm_q;
  r.in_ = in_perm_n;
  r.iq = in_perm_q;
  r.bias_n = bias_n;
  r.bias_q = bias_q;
  r.o_lim = o_lim;
  r.i_lim = i_lim;
  const long total = (long)units * SLAB + (dbias ? cout : 0);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(64, RG), 0, (hipStream_t)stream, r);
  return sisr_check_launch();
}

// ---- batched launch of sisr_wgrad3x3_c64_geo jobs of different geometries (<= 8 per launch, any number per call)
struct sisr_wgrad_geo_job_host {
  const float* x;
  const float* dy;
  float* dw;
  float* dbias;
  int B, H, W, cin, cout, up, co_real, ci_real;
  unsigned long long active_units;
};
extern "C" size_t sisr_wgrad_geo_job_bytes(void) { return sizeof(sisr_wgrad_geo_job_host); }
static int wgrad_geo_units(const sisr_wgrad_geo_job_host& j) {
  const int all = (j.cin / 64) * (j.cout / 64) * 4;
  if (!j.active_units) return all;
  int n = 0;
  for (int u = 0; u < all; ++u) n += (int)((j.active_units >> u) & 1);
  return n;
}
// workgroups of one job inside a batch of `nj` jobs: the K-split that leaves the whole launch about two workgroups per CU
static int wgrad_geo_batch_split(const sisr_wgrad_geo_job_host& j, int units, int nj) {
  const long tiles = (long)j.B * ((j.H + WT_H - 1) / WT_H) * ((j.W + WT_W - 1) / WT_W);
  long S = 512 / ((long)units * nj);
  if (S < 1) S = 1;
  if (S > tiles) S = tiles;
  return (int)S;
}
static bool wgrad_geo_job_ok(const sisr_wgrad_geo_job_host& j) {
  return j.x && j.dy && j.dw && j.B > 0 && j.H >= 2 && j.W >= 2 && j.cin > 0 && j.cout > 0 && !(j.cin & 63) && !(j.cout & 63) &&
         (j.cin / 64) * (j.cout / 64) * 4 <= WG_MAX_UNITS && j.up >= 0 && j.up <= 2 && !((j.up & 1) && ((j.H | j.W) & 1)) &&
         j.co_real > 0 && j.co_real <= j.cout && j.ci_real > 0 && j.ci_real <= j.cin;
}
extern "C" size_t sisr_wgrad3x3_c64_geo_batch_workspace_bytes(const void* jobs_host, int njobs) {
  if (!jobs_host || njobs <= 0) return 0;
  const sisr_wgrad_geo_job_host* jobs = static_cast<const sisr_wgrad_geo_job_host*>(jobs_host);
  size_t worst = 0;
  for (int k0 = 0; k0 < njobs; k0 += WG_BATCH) {
    const int nj = njobs - k0 < WG_BATCH ? njobs - k0 : WG_BATCH;
    size_t need = 0;
    for (int k = k0; k < k0 + nj; ++k) {
      if (!wgrad_geo_job_ok(jobs[k])) return 0;
      const int units = wgrad_geo_units(jobs[k]);
      if (units <= 0) return 0;
      const int S = wgrad_geo_batch_split(jobs[k], units, nj);
      need += ((size_t)S * units * SLAB + (size_t)S * jobs[k].cout + 64) * sizeof(float);
    }
    if (need > worst) worst = need;
  }
  return worst;
}
extern "C" int sisr_wgrad3x3_c64_geo_batch(const void* jobs_host, int njobs, float* workspace, size_t workspace_bytes,
                                           void* stream) {
  if (!jobs_host || njobs <= 0 || !workspace || !sisr_aligned16(workspace)) return SISR_ERR_ARG;
  const size_t need = sisr_wgrad3x3_c64_geo_batch_workspace_bytes(jobs_host, njobs);  // 0: some job is not a valid one
  if (need == 0 || workspace_bytes < need) return SISR_ERR_ARG;
  const sisr_wgrad_geo_job_host* jobs = static_cast<const sisr_wgrad_geo_job_host*>(jobs_host);
  for (int k0 = 0; k0 < njobs; k0 += WG_BATCH) {
    const int nj = njobs - k0 < WG_BATCH ? njobs - k0 : WG_BATCH;
    WgradBatch wb;
    ReduceBatch rb;
    memset(&wb, 0, sizeof(wb));
    memset(&rb, 0, sizeof(rb));
    float* ws = workspace;
    int smax = 1, umax = 1;
    long tmax = 0;
    for (int k = 0; k < nj; ++k) {
      const sisr_wgrad_geo_job_host& j = jobs[k0 + k];
      if (!sisr_aligned16(j.x) || !sisr_aligned16(j.dy)) return SISR_ERR_ALIGN;
      WgradParams& p = wb.job[k];
      ReduceParams& r = rb.job[k];
      const int Hs = j.H >> (j.up & 1), Ws = j.W >> (j.up & 1);
      const int Hy = (j.up & 2) ? (j.H + 1) / 2 : j.H, Wy = (j.up & 2) ? (j.W + 1) / 2 : j.W;
      p.x = j.x;
      p.xv.sB = (long)Hs * Ws * j.cin; p.xv.sH = (long)Ws * j.cin; p.xv.sW = j.cin; p.xv.chi = 0; p.xv.clo = 64; p.xv.cdiv = 1 << 30;
      p.dy = j.dy;
      p.yv.sB = (long)Hy * Wy * j.cout; p.yv.sH = (long)Wy * j.cout; p.yv.sW = j.cout; p.yv.chi = 0; p.yv.clo = 64; p.yv.cdiv = 1 << 30;
      p.B = j.B;
      p.H = j.H;
      p.W = j.W;
      p.cin_chunks = j.cin / 64;
      p.cout_chunks = j.cout / 64;
      p.tiles_w = (j.W + WT_W - 1) / WT_W;
      p.tiles_h = (j.H + WT_H - 1) / WT_H;
      p.geo_reflect = 1;
      p.geo_up = j.up & 1;
      p.geo_ysub = (j.up >> 1) & 1;
      const int all_units = p.cin_chunks * p.cout_chunks * 4;
      int units = all_units;
      if (j.active_units) {
        units = 0;
        unsigned seen_bias = 0;
        for (int u = 0; u < all_units; ++u) {
          if (!((j.active_units >> u) & 1)) continue;
          p.unit_map[units] = r.unit_map[units] = (unsigned char)u;
          const int cq = (u >> 2) % p.cout_chunks, coh = u & 1;
          if (!((seen_bias >> (cq * 2 + coh)) & 1)) {
            p.bias_units |= 1ull << units;
            seen_bias |= 1u << (cq * 2 + coh);
          }
          ++units;
        }
        if (units == 0 || (all_units < 64 && (j.active_units >> all_units) != 0)) return SISR_ERR_ARG;
        const unsigned need_bias = (1u << ((j.co_real + 31) / 32)) - 1;
        if (j.dbias && (seen_bias & need_bias) != need_bias) return SISR_ERR_ARG;
      }
      p.mapped = r.mapped = j.active_units != 0;
      p.units = units;
      p.S = wgrad_geo_batch_split(j, units, nj);
      p.slabs = ws;
      ws += (size_t)p.S * units * SLAB;
      p.bias_slabs = j.dbias ? ws : nullptr;
      ws += (size_t)p.S * j.cout + 64;
      ws = reinterpret_cast<float*>(((uintptr_t)ws + 15) & ~(uintptr_t)15);
      r.slabs = p.slabs;
      r.bias_slabs = p.bias_slabs;
      r.dw = j.dw;
      r.db = j.dbias;
      r.so = (long)j.ci_real * 9;
      r.si = 9;
      r.alpha = 1.f;
      r.S = p.S;
      r.units = units;
      r.cin_chunks = p.cin_chunks;
      r.cout_chunks = p.cout_chunks;
      r.flip = 0;
      r.on = 1;
      r.oq = 64;
      r.in_ = 1;
      r.iq = 64;
      r.bias_n = 1;
      r.bias_q = 64;
      r.o_lim = j.co_real;
      r.i_lim = j.ci_real;
      if (p.S > smax) smax = p.S;
      if (units > umax) umax = units;
      const long total = (long)units * SLAB + (j.dbias ? j.cout : 0);
      if (total > tmax) tmax = total;
    }
    const size_t lds_bytes = (size_t)(LDS_X + LDS_Y) * sizeof(float);
    SISR_ALLOW_LDS(wgrad3x3_c64_geo_batch_kernel, lds_bytes);
    hipLaunchKernelGGL(wgrad3x3_c64_geo_batch_kernel, dim3(smax, umax, nj), dim3(256), lds_bytes, (hipStream_t)stream, wb);
    int rc = sisr_check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)((tmax + 255) / 256), nj), dim3(64, RG), 0, (hipStream_t)stream, rb);
    rc = sisr_check_launch();
    if (rc) return rc;
  }
  return SISR_OK;
}

// ---- batched launch: njobs (<= 8) 64 -> 64 weight gradients of one (B, H, W), plain OIHW outputs
struct sisr_wgrad_job_host {
  const float* x;
  const float* dy;
  const float* dy_scale;
  const float* dy_shift;
  float* dw;
  float* dbias;
};
static int wgrad_batch_split(int njobs, int B, int H, int W) {
  const long tiles = (long)B * ((H + WT_H - 1) / WT_H) * ((W + WT_W - 1) / WT_W);
  long S = 256 / njobs;  // one persistent workgroup per CU across the whole grid (wgrad3x3_c64_full_batch_kernel)
  if (S < 1) S = 1;
  if (S > tiles) S = tiles;
  return (int)S;
}
extern "C" size_t sisr_wgrad_job_bytes(void) { return sizeof(sisr_wgrad_job_host); }
extern "C" int sisr_wgrad3x3_c64_batch_max(void) { return WG_BATCH; }
extern "C" size_t sisr_wgrad3x3_c64_batch_workspace_bytes(int njobs, int B, int H, int W) {
  if (njobs <= 0 || njobs > WG_BATCH || B <= 0 || H <= 0 || W <= 0) return 0;
  const int S = wgrad_batch_split(njobs, B, H, W);
  return (size_t)njobs * ((size_t)S * 4 * SLAB + (size_t)S * 64) * sizeof(float);
}
extern "C" int sisr_wgrad3x3_c64_batch(const void* jobs_host, int njobs, const int64_t* xview, const int64_t* dyview,
                                       float* workspace, size_t workspace_bytes, int B, int H, int W, void* stream) {
  if (!jobs_host || !xview || !dyview || !workspace || njobs <= 0 || njobs > WG_BATCH || B <= 0 || H <= 0 || W <= 0)
    return SISR_ERR_ARG;
  if (workspace_bytes < sisr_wgrad3x3_c64_batch_workspace_bytes(njobs, B, H, W) || !sisr_aligned16(workspace)) return SISR_ERR_ARG;
  const sisr_wgrad_job_host* jobs = static_cast<const sisr_wgrad_job_host*>(jobs_host);
  const int S = wgrad_batch_split(njobs, B, H, W);
  const size_t per_job = (size_t)S * 4 * SLAB + (size_t)S * 64;
  WgradBatch wb;
  ReduceBatch rb;
  memset(&wb, 0, sizeof(wb));
  memset(&rb, 0, sizeof(rb));
  for (int k = 0; k < njobs; ++k) {
    const sisr_wgrad_job_host& j = jobs[k];
    if (!j.x || !j.dy || !j.dw) return SISR_ERR_ARG;
    if (!sisr_aligned16(j.x) || !sisr_aligned16(j.dy) || !sisr_aligned16(j.dy_scale) || !sisr_aligned16(j.dy_shift))
      return SISR_ERR_ALIGN;
    WgradParams& p = wb.job[k];
    p.x = j.x;
    p.xv = view_from(xview);
    p.dy = j.dy;
    p.yv = view_from(dyview);
    if ((p.xv.sB | p.xv.sH | p.xv.sW | p.xv.chi | p.xv.clo | p.yv.sB | p.yv.sH | p.yv.sW | p.yv.chi | p.yv.clo) & 3)
      return SISR_ERR_ALIGN;
    p.dy_scale = j.dy_scale;
    p.dy_shift = j.dy_shift;
    p.B = B;
    p.H = H;
    p.W = W;
    p.cin_chunks = p.cout_chunks = 1;
    p.tiles_w = (W + WT_W - 1) / WT_W;
    p.tiles_h = (H + WT_H - 1) / WT_H;
    p.S = S;
    p.slabs = workspace + (size_t)k * per_job;
    p.bias_slabs = j.dbias ? p.slabs + (size_t)S * 4 * SLAB : nullptr;
    p.mapped = 0;
    ReduceParams& r = rb.job[k];
    r.slabs = p.slabs;
    r.bias_slabs = p.bias_slabs;
    r.dw = j.dw;
    r.db = j.dbias;
    r.so = 64 * 9;
    r.si = 9;
    r.alpha = 1.f;
    r.S = S;
    r.units = 4;
    r.cin_chunks = r.cout_chunks = 1;
    r.flip = 0;
    r.on = 1;
    r.oq = 64;
    r.in_ = 1;
    r.iq = 64;
    r.bias_n = 1;
    r.bias_q = 64;
    r.mapped = 0;
  }
  const size_t lds_bytes = (size_t)(FW_X + FW_Y) * sizeof(float);
  SISR_ALLOW_LDS(wgrad3x3_c64_full_batch_kernel, lds_bytes);
  hipLaunchKernelGGL(wgrad3x3_c64_full_batch_kernel, dim3(S, 1, njobs), dim3(256), lds_bytes, (hipStream_t)stream, wb);
  int rc = sisr_check_launch();
  if (rc) return rc;
  const long total = 4L * SLAB + 64;  // bias rows past the weights; jobs without a bias skip them inside (db == null)
  hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)((total + 255) / 256), njobs), dim3(64, RG), 0, (hipStream_t)stream, rb);
  return sisr_check_launch();
}

static int wgrad_bf16_split(int B, int H, int W, int pairs) {
  const long tiles = (long)B * ((H + WT_H - 1) / WT_H) * ((W + WT_W - 1) / WT_W);
  long S = 256 / pairs;  // one persistent workgroup per CU (it prefetches its next tile: 152 KB in flight per CU)
  if (S < 1) S = 1;
  if (S > tiles) S = tiles;
  return (int)S;
}

extern "C" size_t sisr_wgrad3x3_c64_bf16_workspace_bytes(int B, int H, int W, int cin, int cout) {
  if (B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0 || (cin & 63) || (cout & 63)) return 0;
  const int pairs = (cin / 64) * (cout / 64);
  const int S = wgrad_bf16_split(B, H, W, pairs);
  return ((size_t)S * pairs * 4 * SLAB + (size_t)S * cout) * sizeof(float);
}

static int wgrad3x3_c64_bf16_launch(const float* x, const int64_t* xview, const float* dy, const int64_t* dyview,
                                    const float* dy_scale, const float* dy_shift, float alpha, float* dw, int64_t so,
                                    int64_t si, int flip_taps, int out_perm_n, int out_perm_q, int in_perm_n,
                                    int in_perm_q, float* dbias, int bias_n, int bias_q, float* workspace,
                                    size_t workspace_bytes, int B, int H, int W, int cin, int cout, int storage, void* stream) {
  if (!x || !dy || !dw || !xview || !dyview || !workspace || B <= 0 || H <= 0 || W <= 0) return SISR_ERR_ARG;
  if (storage != 0 && storage != 1 && storage != 3) return SISR_ERR_UNSUPPORTED;
  if ((cin & 63) || (cout & 63) || cin <= 0 || cout <= 0) return SISR_ERR_UNSUPPORTED;
  if (workspace_bytes < sisr_wgrad3x3_c64_bf16_workspace_bytes(B, H, W, cin, cout)) return SISR_ERR_ARG;
  if (!sisr_aligned16(x) || !sisr_aligned16(dy) || !sisr_aligned16(workspace) || !sisr_aligned16(dy_scale) ||
      !sisr_aligned16(dy_shift))
    return SISR_ERR_ALIGN;
  WgradParams p = {};
  p.x = x;
  p.xv = view_from(xview);
  p.dy = dy;
  p.yv = view_from(dyview);
  if ((p.xv.sB | p.xv.sH | p.xv.sW | p.xv.chi | p.xv.clo | p.yv.sB | p.yv.sH | p.yv.sW | p.yv.chi | p.yv.clo) & (storage ? 7 : 3))
    return SISR_ERR_ALIGN;  // 16-B pieces: four fp32 or eight bf16 elements
  p.dy_scale = dy_scale;
  p.dy_shift = dy_shift;
  p.B = B;
  p.H = H;
  p.W = W;
  p.cin_chunks = cin / 64;
  p.cout_chunks = cout / 64;
  p.tiles_w = (W + WT_W - 1) / WT_W;
  p.tiles_h = (H + WT_H - 1) / WT_H;
  const int pairs = p.cin_chunks * p.cout_chunks;
  const int units = pairs * 4;
  p.S = wgrad_bf16_split(B, H, W, pairs);
  p.slabs = workspace;
  p.bias_slabs = dbias ? workspace + (size_t)p.S * units * SLAB : nullptr;
  const size_t lds_bytes = BW_X_BYTES + BW_Y_BYTES;
  if (storage == 0) {
    SISR_ALLOW_LDS(wgrad3x3_c64_bf16_kernel, lds_bytes);
    hipLaunchKernelGGL(wgrad3x3_c64_bf16_kernel, dim3(p.S, pairs), dim3(256), lds_bytes, (hipStream_t)stream, p);
  } else if (storage == 1) {
    SISR_ALLOW_LDS(wgrad3x3_c64_bf16_x16_kernel, lds_bytes);
    hipLaunchKernelGGL(wgrad3x3_c64_bf16_x16_kernel, dim3(p.S, pairs), dim3(256), lds_bytes, (hipStream_t)stream, p);
  } else {
#ifdef SISR_DIAG
    if (g_diag_wgrad_stamp) {
      SISR_ALLOW_LDS(wgrad3x3_c64_bf16_xy16_stamp_kernel, lds_bytes);
      hipLaunchKernelGGL(wgrad3x3_c64_bf16_xy16_stamp_kernel, dim3(p.S, pairs), dim3(256), lds_bytes, (hipStream_t)stream, p,
                         g_diag_wgrad_stamp);
    } else
#endif
    {
      SISR_ALLOW_LDS(wgrad3x3_c64_bf16_xy16_kernel, lds_bytes);
      hipLaunchKernelGGL(wgrad3x3_c64_bf16_xy16_kernel, dim3(p.S, pairs), dim3(256), lds_bytes, (hipStream_t)stream, p);
    }
  }
  int rc = sisr_check_launch();
  if (rc) return rc;
  ReduceParams r = {};
  r.mapped = 0;
  r.slabs = p.slabs;
  r.bias_slabs = p.bias_slabs;
  r.dw = dw;
  r.db = dbias;
  r.so = so;
  r.si = si;
  r.alpha = alpha;
  r.S = p.S;
  r.units = units;
  r.cin_chunks = p.cin_chunks;
  r.cout_chunks = p.cout_chunks;
  r.flip = flip_taps;
  r.on = out_perm_n;
  r.oq = out_perm_q;
  r.in_ = in_perm_n;
  r.iq = in_perm_q;
  r.bias_n = bias_n;
  r.bias_q = bias_q;
  const long total = (long)units * SLAB + (dbias ? cout : 0);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(64, RG), 0, (hipStream_t)stream, r);
  return sisr_check_launch();
}

extern "C" int sisr_wgrad3x3_c64_bf16(const float* x, const int64_t* xview, const float* dy, const int64_t* dyview,
                                      const float* dy_scale, const float* dy_shift, float alpha, float* dw, int64_t so,
                                      int64_t si, int flip_taps, int out_perm_n, int out_perm_q, int in_perm_n,
                                      int in_perm_q, float* dbias, int bias_n, int bias_q, float* workspace,
                                      size_t workspace_bytes, int B, int H, int W, int cin, int cout, void* stream) {
  return wgrad3x3_c64_bf16_launch(x, xview, dy, dyview, dy_scale, dy_shift, alpha, dw, so, si, flip_taps, out_perm_n,
                                  out_perm_q, in_perm_n, in_perm_q, dbias, bias_n, bias_q, workspace, workspace_bytes, B, H,
                                  W, cin, cout, 0, stream);
}

// The same with bf16 STORAGE of its inputs: storage 1 = x is a bf16 map (a saved activation), 3 = x and dY are (same Views,
// 2-byte elements, pointers passed as float*; workspace from sisr_wgrad3x3_c64_bf16_workspace_bytes).
extern "C" int sisr_wgrad3x3_c64_bf16s(const float* x, const int64_t* xview, const float* dy, const int64_t* dyview,
                                       const float* dy_scale, const float* dy_shift, float alpha, float* dw, int64_t so,
                                       int64_t si, int flip_taps, int out_perm_n, int out_perm_q, int in_perm_n,
                                       int in_perm_q, float* dbias, int bias_n, int bias_q, float* workspace,
                                       size_t workspace_bytes, int B, int H, int W, int cin, int cout, int storage,
                                       void* stream) {
  if (storage != 1 && storage != 3) return SISR_ERR_ARG;
  return wgrad3x3_c64_bf16_launch(x, xview, dy, dyview, dy_scale, dy_shift, alpha, dw, so, si, flip_taps, out_perm_n,
                                  out_perm_q, in_perm_n, in_perm_q, dbias, bias_n, bias_q, workspace, workspace_bytes, B, H,
                                  W, cin, cout, storage, stream);
}

static int wgrad_x3_split(int B, int H, int W, int pairs) {
  const long tiles = (long)B * ((H + XW_TH - 1) / XW_TH) * ((W + WT_W - 1) / WT_W);
  long S = 256 / pairs;  // one persistent workgroup per CU
  if (S < 1) S = 1;
  if (S > tiles) S = tiles;
  return (int)S;
}

extern "C" size_t sisr_wgrad3x3_c64_x3_workspace_bytes(int B, int H, int W, int cin, int cout) {
  if (B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0 || (cin & 63) || (cout & 63)) return 0;
  const int pairs = (cin / 64) * (cout / 64);
  const int S = wgrad_x3_split(B, H, W, pairs);
  return ((size_t)S * pairs * 4 * SLAB + (size_t)S * cout) * sizeof(float);
}

extern "C" int sisr_wgrad3x3_c64_x3(const float* x, const int64_t* xview, const float* dy, const int64_t* dyview,
                                    const float* dy_scale, const float* dy_shift, float alpha, float* dw, int64_t so,
                                    int64_t si, int flip_taps, int out_perm_n, int out_perm_q, int in_perm_n,
                                    int in_perm_q, float* dbias, int bias_n, int bias_q, float* workspace,
                                    size_t workspace_bytes, int B, int H, int W, int cin, int cout, void* stream) {
  if (!x || !dy || !dw || !xview || !dyview || !workspace || B <= 0 || H <= 0 || W <= 0) return SISR_ERR_ARG;
  if ((cin & 63) || (cout & 63) || cin <= 0 || cout <= 0) return SISR_ERR_UNSUPPORTED;
  if (workspace_bytes < sisr_wgrad3x3_c64_x3_workspace_bytes(B, H, W, cin, cout)) return SISR_ERR_ARG;
  if (!sisr_aligned16(x) || !sisr_aligned16(dy) || !sisr_aligned16(workspace) || !sisr_aligned16(dy_scale) ||
      !sisr_aligned16(dy_shift))
    return SISR_ERR_ALIGN;
  WgradParams p = {};
  p.x = x;
  p.xv = view_from(xview);
  p.dy = dy;
  p.yv = view_from(dyview);
  if ((p.xv.sB | p.xv.sH | p.xv.sW | p.xv.chi | p.xv.clo | p.yv.sB | p.yv.sH | p.yv.sW | p.yv.chi | p.yv.clo) & 3)
    return SISR_ERR_ALIGN;
  p.dy_scale = dy_scale;
  p.dy_shift = dy_shift;
  p.B = B;
  p.H = H;
  p.W = W;
  p.cin_chunks = cin / 64;
  p.cout_chunks = cout / 64;
  p.tiles_w = (W + WT_W - 1) / WT_W;
  p.tiles_h = (H + XW_TH - 1) / XW_TH;
  const int pairs = p.cin_chunks * p.cout_chunks;
  const int units = pairs * 4;
  p.S = wgrad_x3_split(B, H, W, pairs);
  p.slabs = workspace;
  p.bias_slabs = dbias ? workspace + (size_t)p.S * units * SLAB : nullptr;
  const size_t lds_bytes = 3 * (size_t)(XW_X + XW_Y);
  SISR_ALLOW_LDS(wgrad3x3_c64_x3_kernel, lds_bytes);
  hipLaunchKernelGGL(wgrad3x3_c64_x3_kernel, dim3(p.S, pairs), dim3(256), lds_bytes, (hipStream_t)stream, p);
  int rc = sisr_check_launch();
  if (rc) return rc;
  ReduceParams r = {};
  r.mapped = 0;
  r.slabs = p.slabs;
  r.bias_slabs = p.bias_slabs;
  r.dw = dw;
  r.db = dbias;
  r.so = so;
  r.si = si;
  r.alpha = alpha;
  r.S = p.S;
  r.units = units;
  r.cin_chunks = p.cin_chunks;
  r.cout_chunks = p.cout_chunks;
  r.flip = flip_taps;
  r.on = out_perm_n;
  r.oq = out_perm_q;
  r.in_ = in_perm_n;
  r.iq = in_perm_q;
  r.bias_n = bias_n;
  r.bias_q = bias_q;
  const long total = (long)units * SLAB + (dbias ? cout : 0);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(64, RG), 0, (hipStream_t)stream, r);
  return sisr_check_launch();
}
