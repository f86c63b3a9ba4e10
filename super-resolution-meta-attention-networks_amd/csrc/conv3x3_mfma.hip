// 3x3 same-convolution over 64-channel chunks, fp32, on the gfx950 matrix cores.
//
// Replaces, for every Cin,Cout multiple of 64, the reference's default_conv forward
// (advanced/common.py:5-8 -> nn.Conv2d(k=3,pad=1)) and -- with flipped / role-swapped packed
// weights -- its input gradient.  Implicit GEMM: M = pixels, N = 64 output channels per block,
// K = 9 taps x 64 input channels per chunk, on v_mfma_f32_32x32x2_f32 (exact fp32: bit-identical
// to a k-ordered fmaf chain, MI355X guide "FP32-input MFMA").
//
// Work decomposition (one workgroup = 256 threads = 4 waves):
//   tile   : 4 rows x 32 cols of output pixels, one 64-wide output-channel chunk (blockIdx.y)
//   wave   : (ph, ch) -> rows {2ph, 2ph+1} x channels [32ch, 32ch+32): two 32x32 accumulators
//   LDS    : the 6 x 34 pixel halo of the input chunk, 68-float pixel stride (272 B) so the 16
//            lanes of a ds_read_b128 group land on 16 distinct 16-B bank slots
//   A frag : lane (pixel p = l&31, half h = l>>5) reads channels [8j+4h, 8j+4h+4) of its pixel as ONE
//            ds_read_b128 and feeds four consecutive MFMAs (k=0 <-> ci 8j+e, k=1 <-> ci 8j+4+e)
//   B frag : the matching 4 weights for output channel n = l&31, pre-packed so a wave's read is two
//            contiguous 512-B runs; fetched global->VGPR two steps ahead (weights are L2-resident:
//            147 KB per chunk pair, shared by every workgroup)
// Fused prologue: per-(b,ci) affine on the input (used for dRes = dOut*g + c in the RCAB backward).
// Fused epilogue: +bias, ReLU, scalar and per-(b,co) scale, ReLU-mask, residual add, pixel-shuffle
// address map, and per-wave partial sums for the global average pool.
#include "sisr_common.h"
#include "ca_gate.h"
#include <string.h>
#include <stdlib.h>
#include <type_traits>

#ifdef SISR_WT_STORES
#define SISR_Y_STORE1 sisr_buf_store1_wt
#define SISR_Y_STORE4 sisr_buf_store4_wt
#else
#define SISR_Y_STORE1 sisr_buf_store1
#define SISR_Y_STORE4 sisr_buf_store4
#endif
#ifndef SISR_GATE_WGS
#define SISR_GATE_WGS 4   // resident workgroups per CU of the 2-row GATE build (3: single-batch staging, see the kernel)
#endif
#ifndef SISR_GATE_PRIO
#define SISR_GATE_PRIO 0  // 1: the GATE prologue (two maps, arithmetic, a store) runs at wave priority 3
#endif
#define TH 4
#define TW 32
#define HALO_H (TH + 2)
#define HALO_W (TW + 2)
#define HALO_ITEMS (HALO_H * HALO_W * 16)            // float4 items in the halo
#define STAGE_ITERS ((HALO_ITEMS + 255) / 256)        // 13

struct ConvParams {
  const float* x;
  View xv;
  float* y;
  View yv;
  const float* res;
  const float* mask;
  const float* w;
  const float* bias;
  const float* in_scale;
  const float* in_shift;
  const float* out_scale;
  float* gap;
  const float* gate_add;  // GATE prologue: input = x * in_scale[b,c] + gate_add (x's layout), ...
  float* gate_out;        // ... whose in-image value is also written here once (by the tile that owns the pixel)
  const float* dot;       // DOT epilogue: gap partials hold sum(v * dot) instead of sum(v) (y's layout)
  float alpha;
  int bias_n, bias_q;
  int B, H, W, cin_chunks, cout_chunks, relu, tiles_w, tiles_h;
  int mask_leaky;  // relu: 0 none, 1 ReLU, 2 LeakyReLU(0.2); mask_leaky: the mask is LeakyReLU's derivative, not ReLU's
  // Channel-attention tails (64 -> 64, fp32 kernels): the workgroup that finishes a sample LAST (device-scope counter,
  // returned to zero) turns the GAP partial sums this launch wrote into the gate (forward: `gap` of a plain conv), or the
  // DOT partial sums into the gate's backward (per sample; the last sample's finisher then sums the parameter gradients
  // over the batch).  Same arithmetic and summation order as ca_gate_fwd / ca_gate_bwd (ca_gate.h): which workgroup
  // does it does not change a bit.  Saves one launch on the serial chain per block and direction.
  struct {
    const float *w1, *b1, *w2, *b2, *mul;
    float *s, *hid, *ca, *g;
    unsigned* counter;  // [B]
    float inv_hw;
    int R;
  } fwd_tail;  // active when g != nullptr
  struct {
    const float *w1, *w2, *s, *hid, *ca, *mul;
    float *shift, *dmul, *dz2, *dz1, *dw1, *db1, *dw2, *db2;
    unsigned* counter;  // [B + 1]: per sample, then one for the batch
    float inv_hw;
    int R;
  } bwd_tail;  // active when shift != nullptr
  // Gate HEADS (template HEAD of conv3x3_c64_v4_kernel): the CONSUMER of a gate computes it -- every workgroup, for its own
  // sample, from the partial sums the previous launch wrote -- instead of a launch of its own on the serial chain.  The field
  // blocks above carry the operands (fwd_tail: HEAD 1, bwd_tail: HEAD 2; counters and parameter-gradient fields unused).
  const float* head_part;  // [B][head_parts][64]
  int head_parts;
  // Generalised input geometry (template GEO of conv3x3_c64_v4_kernel; all zero elsewhere).  Output pixel (h, w) of the H x W
  // output reads the VIRTUAL input map of geo_h x geo_w pixels at (h + kh - 1 - geo_off, w + kw - 1 - geo_off):
  //   geo_reflect: coordinates outside the virtual map are reflected (-1 -> 1, n -> n - 2: ReflectionPad2d(1)) instead of
  //                reading zeros;  geo_up: the stored map is the virtual one subsampled by 2^geo_up (nearest upsampling read
  //                in place: stored pixel = virtual >> geo_up);  geo_off = 1 with geo_h = H - 2: the transposed ("full")
  //                form, an input-gradient map two pixels larger than the gradient it is computed from.
  int geo_reflect, geo_up, geo_off, geo_h, geo_w;
  int geo_sub;  // with geo_up = 1 and zero padding: the virtual map is the stored one ZERO-STUFFED (value at even coordinates only:
                // the gradient of a stride-2 conv seen at stride 1), not replicated
#ifdef SISR_DIAG
  unsigned* stamp;  // diagnostic library only: per wave {start lo, start hi, staging, K loop, epilogue, HW_ID, XCC_ID, 0}
#endif
};

// General kernel: every prologue / epilogue combination, XOR-swizzled LDS (16-B chunk k of halo pixel p lives
// at slot k ^ (p & 15)), 52 KB per workgroup so three fit a CU (register budget 168), rolled tap loop with A two
// and B six K-steps ahead.  The hot combinations run on the leaner conv3x3_c64_v4_kernel below; this one stays
// as the fallback and as the carrier of the diagnostic builds that led to it:
// ABL (selected with sisr_conv3x3_c64_set_variant 13 / 16): 3 = no A and no B loads (pure MFMA + staging + stores),
// 6 = phase stamps (s_memtime at start / after staging / after the K loop / at the end, written per wave into the
// gap buffer as uint32 cycles).  Outputs of ABL builds are meaningless.
template <int ABL = 0>
__global__ __launch_bounds__(256, 3) void conv3x3_c64_kernel(ConvParams p) {
  constexpr int V = 2;
  constexpr int PSTR = 64;
  constexpr int AD = 2;  // A prefetch distance in K-steps
  constexpr int BD = 6;  // B prefetch distance in K-steps (8-slot ring)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int q = blockIdx.y;
  // Workgroups are dealt round-robin over the 8 XCDs (block b -> b % 8, a speed-only assumption): give each
  // XCD a contiguous run of tiles so vertically adjacent tiles, which share two halo rows, meet in one L2.
  int bid;
  {
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const unsigned qn = nb >> 3, rn = nb & 7;
    bid = (int)((xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx);
  }
  const int tw = bid % p.tiles_w;
  bid /= p.tiles_w;
  const int th = bid % p.tiles_h;
  const int b = bid / p.tiles_h;
  const int h0 = th * TH, w0 = tw * TW;
  const int ph = wave >> 1, ch = wave & 1;
  const int n = lane & 31, hh = lane >> 5;
  const int H = p.H, W = p.W;

  f32x16 acc0 = {0}, acc1 = {0};
  unsigned long long st0 = 0, st1 = 0, st2 = 0;
  if (ABL == 6) st0 = __builtin_amdgcn_s_memtime();

  for (int c = 0; c < p.cin_chunks; ++c) {
    if (c) __syncthreads();  // previous chunk's A reads are done before the halo is overwritten
    // ---- stage the halo of input chunk c: global -> VGPR -> LDS (16 lanes x 16 B per pixel)
    {
      // Branch-free: out-of-image taps load a clamped (valid) address and are zeroed by a select, so
      // all 13 loads of a thread are in flight together.  c4 = tid & 15 is the same for every item.
      // `tl` launders tid once per chunk: without it hipcc hoists the 13 chunk-invariant halo addresses,
      // clamps and masks out of the chunk loop and keeps ~60 VGPRs live across the whole MFMA loop.
      int tl = tid;
      asm volatile("" : "+v"(tl));
      const float* xb = p.x + (long)b * p.xv.sB + p.xv.chunk(c);
      const int c4 = tl & 15;
      f32x4 s4 = {1.f, 1.f, 1.f, 1.f}, t4 = {0.f, 0.f, 0.f, 0.f};
      if (p.in_scale) s4 = *reinterpret_cast<const f32x4*>(p.in_scale + ((long)b * p.cin_chunks + c) * 64 + c4 * 4);
      if (p.in_shift) t4 = *reinterpret_cast<const f32x4*>(p.in_shift + ((long)b * p.cin_chunks + c) * 64 + c4 * 4);
      f32x4 v[STAGE_ITERS];
#pragma unroll
      for (int it = 0; it < STAGE_ITERS; ++it) {
        const int pix = (it * 256 + tl) >> 4;
        const int pr = pix / HALO_W, pc = pix - pr * HALO_W;
        const int gh = h0 - 1 + pr, gw = w0 - 1 + pc;
        const bool ok = gh >= 0 && gh < H && gw >= 0 && gw < W;
        const int ch_ = min(max(gh, 0), H - 1), cw_ = min(max(gw, 0), W - 1);
        f32x4 t;
        t = *reinterpret_cast<const f32x4*>(xb + (long)ch_ * p.xv.sH + (long)cw_ * p.xv.sW + c4 * 4);
        v[it] = sisr_keep_if(t * s4 + t4, ok);  // zero padding stays zero: the affine is for in-image pixels only
      }
#pragma unroll
      for (int it = 0; it < STAGE_ITERS; ++it) {
        const int idx = it * 256 + tl;
        const int pix = idx >> 4;
        const int slot = (V == 2) ? (c4 ^ (pix & 15)) : c4;
        if (idx < HALO_ITEMS) *reinterpret_cast<f32x4*>(lds + pix * PSTR + slot * 4) = v[it];
      }
    }
    __syncthreads();
    if (ABL == 6) st1 = __builtin_amdgcn_s_memtime();

    // ---- 72 K-steps (9 taps x 8 octets of input channels), 8 MFMAs each
    const float* wq = p.w + ((long)q * p.cin_chunks + c) * (9 * 64 * 64) + hh * 256 + (ch * 32 + n) * 4;
    const int pix0 = (2 * ph) * HALO_W + n;  // halo pixel of (row 2ph, col n) at tap (0,0)
    // A fragment of M-tile m, tap t, octet j: channels [8j+4hh, 8j+4hh+4) of halo pixel pix0 + m*34 + offset(t)
    auto load_a = [&](int m, int t, int j) -> f32x4 {
      const int pix = pix0 + (t / 3 + m) * HALO_W + (t % 3);
      const int slot = (V == 2) ? ((2 * j + hh) ^ (pix & 15)) : (2 * j + hh);
      if (ABL == 3) return (f32x4){(float)pix, 1.f, (float)slot, 2.f};
      return *reinterpret_cast<const f32x4*>(lds + pix * PSTR + slot * 4);
    };
    auto load_b = [&](int s) -> f32x4 {  // K-step s = tap*8 + j; clamped so the run-ahead never leaves the buffer
      if (ABL == 3) return (f32x4){(float)s, 1.f, (float)hh, 3.f};
      return *reinterpret_cast<const f32x4*>(wq + min(s, 71) * 512);
    };
    // Software pipeline over K-steps s = tap*8 + j (8 MFMAs each): B (global/L2) BD steps ahead in an
    // 8-slot ring (BD steps), A (LDS) AD steps ahead in a 4-slot ring; both ring indices depend on j only, so the tap
    // loop stays rolled (64-MFMA body, addresses recomputed per tap instead of 144 live address registers).
    // sched_barrier(0) after every step keeps hipcc from sinking the prefetches down to their first use
    // (it otherwise emits load; s_waitcnt 0; mfma).
    f32x4 bq[8];
    f32x4 aq[4][2];
#pragma unroll
    for (int s = 0; s < BD; ++s) bq[s] = load_b(s);
#pragma unroll
    for (int s = 0; s < AD; ++s) {
      aq[s][0] = load_a(0, 0, s);
      aq[s][1] = load_a(1, 0, s);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const int tnext = min(tap + 1, 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        bq[(j + BD) & 7] = load_b(tap * 8 + j + BD);
        {
          const int ja = j + AD;  // step s + AD: same tap while ja < 8, else the next tap's first octets
          aq[ja & 3][0] = load_a(0, ja < 8 ? tap : tnext, ja & 7);
          aq[ja & 3][1] = load_a(1, ja < 8 ? tap : tnext, ja & 7);
        }
        const f32x4 bb = bq[j];
        const f32x4 a0 = aq[j & 3][0];
        const f32x4 a1 = aq[j & 3][1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], bb[e], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], bb[e], acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  if (ABL == 6) st2 = __builtin_amdgcn_s_memtime();
  // ---- epilogue.  C/D map of the 32x32 tile: column (= output channel) on the lane, pixel
  // (r&3) + 8*(r>>2) + 4*(lane>>5) in register r.
  const int co = ch * 32 + n;
  const int Cout = p.cout_chunks * 64;
  const float bv = p.bias ? p.bias[co * p.bias_n + q * p.bias_q] : 0.f;
  float os = p.alpha;
  if (p.out_scale) os *= p.out_scale[(long)b * Cout + q * 64 + co];
  const long ybase = (long)b * p.yv.sB + p.yv.chunk(q) + co;
  float grow[2];  // GAP partial of the 2-row strip = (row 0) + (row 1), the order every conv kernel uses
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    float gsum = 0.f;
    const int row = h0 + 2 * ph + m;
    const f32x16 acc = m ? acc1 : acc0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int col = w0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      if (row < H && col < W) {
        float v = acc[r] + bv;
        if (p.relu == 1) v = fmaxf(v, 0.f);
        if (p.relu == 2) v = v > 0.f ? v : 0.2f * v;
        v *= os;
        const long off = ybase + (long)row * p.yv.sH + (long)col * p.yv.sW;
        if (p.mask) v = p.mask[off] > 0.f ? v : (p.mask_leaky ? 0.2f * v : 0.f);
        if (p.res) v += p.res[off];
        p.y[off] = v;
        gsum += v;
      }
    }
    grow[m] = gsum + __shfl_xor(gsum, 32);
  }
  if (ABL == 6) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long st3 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && p.gap) {
      unsigned* dbg = reinterpret_cast<unsigned*>(p.gap) + ((long)blockIdx.x * 4 + wave) * 4;
      dbg[0] = (unsigned)(st1 - st0);
      dbg[1] = (unsigned)(st2 - st1);
      dbg[2] = (unsigned)(st3 - st2);
      dbg[3] = (unsigned)(st0 & 0xffffffffu);
    }
    return;
  }
  if (p.gap) {
    if (hh == 0) {
      const int tile = th * p.tiles_w + tw;
      const long parts = (long)p.tiles_w * p.tiles_h * 2;
      p.gap[(((long)b * parts) + tile * 2 + ph) * Cout + q * 64 + co] = grow[0] + grow[1];
    }
  }
}

// ------------------------------------------------------------------ issue-lean kernel ("v4")
// In-kernel stamps and the pure-MFMA ceiling (tools/conv_phases.py, tools/mfma_peak.py) showed that the fp32
// MFMA holds its SIMD's issue port for the whole 64 cycles: every other instruction a co-resident wave
// issues is paid on top, so throughput = 64 / (64 + 4 * non-MFMA instructions per MFMA).  conv3x3_c64_kernel
// spent ~4.6 such instructions per MFMA (82 % ceiling).  This variant keeps the same tile / wave / LDS
// geometry and cuts them to ~1.3:
//   * halo staging walks (row, 16-column block) pairs: row base is scalar, the lane's column offset is one
//     of three precomputed VGPRs -> one global_load (saddr+voffset) + one mask op + one ds_write per item;
//   * the LDS swizzle uses the halo COLUMN (slot = chunk ^ (col & 15)), so the 24 (kw, octet) A addresses
//     are lane constants computed once; tap row and M-tile enter as ds_read immediates (fully unrolled K loop);
//   * B fragments: scalar base + one lane offset VGPR + immediate;
//   * bias is the accumulator's initial value; stores use scalar row/column bases + one lane offset; interior
//     tiles take a predicate-free path.
// Covers the block-hot combinations (plain/ReLU/scale, +GAP, +residual, +mask, +mask+affine prologue); anything
// else is routed to conv3x3_c64_kernel.  Results are bit-identical to it (same MFMA order).
// MT = output rows (32-pixel M-tiles) per wave: 2 -> the 4-row tile described above; 1 -> 2-row tiles (four rows
// of halo, 34.8 KB, one accumulator) for launches whose 4-row grid has fewer workgroups than the chip has CUs (one 128x128 sample: 22.7 -> 14.3 us).
// Both produce bit-identical outputs and GAP partials (same MFMA order per output element; partials are per
// 2-row strip, summed row by row).
// GATE (forward of a gated residual chain): the conv's input is the previous block's output
//   y = t * gate[b,c] + skip,  built while the halo is staged instead of by a separate pass; the tile that owns a
// pixel also writes y out (the next skip / weight-gradient operand), so the map is read and written exactly once
// more than by a plain conv, hidden under the MFMA-bound K loop.  DOT (backward of the same chain): the GAP
// partial slot receives sum(v * dot) -- the gate gradient sum(dY * t) of the block that produced this conv's
// input -- saving that block's separate reduction pass over two maps.
// LEAKY (SFTMD, csrc/sft.hip): without MASK the activation is LeakyReLU(0.2) (p.relu == 2); with MASK the mask is that
// activation's derivative (slope 0.2 where the masking map is <= 0) instead of ReLU's.
// KSEL (SFTMD's merged convs, whose weights are structurally sparse; every other caller: 0 = the dense code path, untouched):
//   1  block-diagonal 64 -> 128: output chunk q contracts input channels 32q .. 32q+31 only (4 of 8 octets per tap)
//   2  128 -> 64 whose second input chunk carries at most 16 channels (2 of 8 octets per tap)
//   3  the transpose of 1, 128 -> 64: input chunk c feeds output channels 32c .. 32c+31 only.  2-row tiles (MT = 1) with
//      BOTH chunks' halos resident (2 x 34.8 KB): wave (row, half) runs one K loop over its own half's chunk -- no
//      branch ("the other two waves skip the chunk" was a wave-uniform branch around the unrolled K loop: 118 spilled VGPRs)
// The skipped products are exact zeros, so results are those of the dense kernel on the zero-padded weights.
// HEAD (small launches, where a 6 us gate kernel between two convs is a tenth of a conv): 1 = with GATE, the gate g of the
// skip being rebuilt is computed here from the previous conv's GAP partial sums (ca_gate_fwd_sample, every workgroup for
// its own sample; all write the same s / hid / ca / g); 2 = with AFFINE + MASK, the GAP-backward shift is computed here from
// the previous conv's DOT partial sums (ca_gate_bwd_sample).  Same device functions as the stand-alone gate kernels.
#define SISR_HEAD_LDS 256  // bytes behind the halo in the launches with a gate head: the 64 gate / shift values of the sample
// GEO (SPARNet's ConvLayers, ref SPARNet/blocks.py:69-103: [nearest x2] -> ReflectionPad2d(1) -> Conv2d(3x3)): the halo is
// staged through the coordinate map of ConvParams::geo_* -- reflection and nearest upsampling are address arithmetic on the
// scalar row offsets and the three per-lane column offsets, so the padded / upsampled map is never materialised and no ring of
// throw-away outputs is computed; with geo_off the same kernel produces the (H + 2) x (W + 2) transposed-conv result from an
// H x W gradient.  KSEL 4 / 5 (one input chunk whose channels >= 32 / >= 8 are zero padding: 32-feature layers, RGB and
// attention-logit ends): 4 / 1 of the 8 octets per tap.
template <bool AFFINE, bool MASK, bool RES, int MT, bool GATE = false, bool DOT = false, bool LEAKY = false, int KSEL = 0, int HEAD = 0,
          bool GEO = false, bool S2 = false>
__global__ __launch_bounds__(256, KSEL == 3 ? 2 : (MT == 1 ? (GATE ? SISR_GATE_WGS : 4) : 3)) void conv3x3_c64_v4_kernel(ConvParams p) {
  // S2 (GEO, 2-row tiles): the stride-2 ConvLayer conv computed at its OUTPUT pixels -- lane n of an M-tile reads halo column
  // 2 n + kw of a 5-row x 66-column halo (84.5 KB) -- instead of the stride-1 conv subsampled (a quarter of the arithmetic)
  static_assert(!S2 || (GEO && MT == 1 && !GATE && !AFFINE && HEAD == 0 && (KSEL == 0 || KSEL == 4)), "S2: plain GEO form only");
  constexpr int SS = S2 ? 2 : 1, HWv = S2 ? 66 : HALO_W, NK = S2 ? 5 : 3;
  constexpr int THv = 2 * MT, HHv = S2 ? 2 * THv + 1 : THv + 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // (Static priorities per workgroup -- the four workgroups of a CU, ids 256 apart, at s_setprio 3 / 2 / 1 / 0 so that their K loops
  // run one after the other and only the last epilogue is exposed -- were measured at 4 tiles in round 4: within +-1 % on three
  // forms, 3 % slower on the ReLU + GAP form.  Left out.)
  // (Wave priorities were measured and left out: with staging and epilogue at s_setprio 3 the staging phase shrinks from
  // 42 k to 7 k cycles -- the unbroken MFMA stream of an older wave otherwise starves it -- but launches take the same time
  // or 1-2 % longer: the pipe is busy either way.)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = blockIdx.y;
  int bid;
  {
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const unsigned qn = nb >> 3, rn = nb & 7;
    bid = (int)((xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx);
  }
  const int tw = bid % p.tiles_w;
  bid /= p.tiles_w;
  const int th = bid % p.tiles_h;
  const int b = bid / p.tiles_h;
  const int h0 = th * THv, w0 = tw * TW;
  const int ph = __builtin_amdgcn_readfirstlane(wave >> 1), ch = __builtin_amdgcn_readfirstlane(wave & 1);
  const int n = lane & 31, hh = lane >> 5;
  const int H = p.H, W = p.W;
  const int co = ch * 32 + n;
  const int Cout = p.cout_chunks * 64;

#ifdef SISR_DIAG
  unsigned long long st0 = 0, st1 = 0, st2 = 0, rt0 = 0;
  if (p.stamp) {
    st0 = __builtin_amdgcn_s_memtime();
    rt0 = __builtin_amdgcn_s_memrealtime();
  }
#endif
  const float bv = p.bias ? p.bias[co * p.bias_n + q * p.bias_q] : 0.f;
  f32x16 acc0, acc1;  // acc1 is dead code for MT == 1
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = bv;

  // lane constants for the A reads: float offset of (halo row 2ph, col n+kw, slot (2j+hh)^((n+kw)&15))
  unsigned aoff[3][8];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      aoff[kw][j] = ((SS * MT * ph) * HWv + SS * n + kw) * 64 +
                    (((2 * (KSEL == 1 ? (j & 3) + 4 * q : j) + hh) ^ ((SS * n + kw) & 15)) << 2) +
                    (KSEL == 3 ? ch * (HHv * HWv * 64) : 0);
  const unsigned boff = hh * 256 + co * 4;

  if constexpr (HEAD == 1) {
    ca_gate_fwd_sample<false>(p.head_part, p.head_parts, p.fwd_tail.inv_hw, b, p.fwd_tail.w1, p.fwd_tail.b1, p.fwd_tail.w2,
                              p.fwd_tail.b2, p.fwd_tail.R, p.fwd_tail.mul, p.fwd_tail.s, p.fwd_tail.hid, p.fwd_tail.ca,
                              p.fwd_tail.g, lds, lds + HHv * HWv * 64);
    __syncthreads();  // g (= p.in_scale of this launch): 64 floats of LDS behind the halo (launched with SISR_HEAD_LDS more
                      // bytes) hand it to the staging below -- no store -> load round trip; the scratch is free again
  } else if constexpr (HEAD == 2) {
    ca_gate_bwd_sample<false>(p.head_part, p.head_parts, p.bwd_tail.inv_hw, b, p.bwd_tail.w1, p.bwd_tail.w2, p.bwd_tail.R,
                              p.bwd_tail.hid, p.bwd_tail.ca, p.bwd_tail.mul, p.bwd_tail.shift, p.bwd_tail.dmul,
                              p.bwd_tail.dz2, p.bwd_tail.dz1, lds, lds + HHv * HWv * 64);
    __syncthreads();  // shift (= p.in_shift of this launch), handed over the same way
  }
  int c_begin = 0;
  if constexpr (KSEL == 3) {
    // chunk 0's halo goes to LDS here, chunk 1's in the (single) pass of the loop below into the second half: then one
    // barrier and one K loop per wave, over the chunk of its own channel half
    const int c4 = tid & 15, pcol = tid >> 4;
    const float* xb = p.x + (long)b * p.xv.sB + p.xv.chunk(0);
    f32x4 v[HHv][3];
#pragma unroll
    for (int r = 0; r < HHv; ++r) {
      const float* xrow = xb + (long)min(max(h0 - 1 + r, 0), H - 1) * p.xv.sH;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int gw = w0 - 1 + pcol + 16 * k;
        if (k < 2 || pcol < 2) v[r][k] = *reinterpret_cast<const f32x4*>(xrow + (long)min(max(gw, 0), W - 1) * p.xv.sW + c4 * 4);
      }
    }
#pragma unroll
    for (int r = 0; r < HHv; ++r) {
      const int gh = h0 - 1 + r;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int col = pcol + 16 * k, gw = w0 - 1 + col;
        if (k < 2 || pcol < 2)
          *reinterpret_cast<f32x4*>(lds + r * (HWv * 64) + col * 64 + ((c4 ^ (col & 15)) << 2)) =
              sisr_keep_if(v[r][k], gh >= 0 && gh < H && gw >= 0 && gw < W);
      }
    }
    c_begin = 1;
  }
  for (int c = c_begin; c < p.cin_chunks; ++c) {
    if (KSEL != 3 && c) __syncthreads();
    float* ldsc = lds + (KSEL == 3 ? c * (HHv * HWv * 64) : 0);  // KSEL 3: both chunks' halos are resident
    // The first six weight fragments of the K loop are requested BEFORE the halo is staged: they do not depend on it, and a
    // K loop that asks for them after the staging barrier opens with an L2 round trip (0.4-0.6 us per tile in which the
    // matrix pipe of this workgroup's SIMDs idles unless another workgroup happens to be mid-loop; per-CU timelines from
    // in-kernel stamps, tools/conv_timeline.py).
    f32x4 bq[8];
    const sisr_rsrc_t rw = sisr_rsrc(p.w + ((long)q * p.cin_chunks + c) * (9 * 64 * 64));
    // (Not in the builds with an arithmetic staging prologue: there the 24 registers spill and the launch gets 2-3 % slower.)
    constexpr bool EARLY_B = KSEL == 0 && !AFFINE && !GATE;
    if constexpr (EARLY_B) {
#pragma unroll
      for (int s = 0; s < 6; ++s) bq[s] = sisr_buf_load4(rw, boff * 4u, (unsigned)(s * 2048));
    }
    {  // ---- halo staging: thread = (chunk c4, column pcol + {0,16,32}), rows 0..5.  Every access is {scalar resource of
       // the sample's chunk, lane byte offset of (column, 16-B piece), scalar row offset}: no per-access address arithmetic on
       // the vector unit, which the MFMA stream of the co-resident waves would pay for (sisr_common.h, buffer addressing)
      int tl = tid;
      asm volatile("" : "+v"(tl));  // keep the per-chunk address math out of the K loop's live ranges
      const int c4 = tl & 15, pcol = tl >> 4;
      const sisr_rsrc_t rx = sisr_rsrc(p.x + (long)b * p.xv.sB + p.xv.chunk(c));  // scalar
      f32x4 s4 = {1.f, 1.f, 1.f, 1.f}, t4 = {0.f, 0.f, 0.f, 0.f};
      if (AFFINE) {
        s4 = *reinterpret_cast<const f32x4*>(p.in_scale + ((long)b * p.cin_chunks + c) * 64 + c4 * 4);
        if constexpr (HEAD == 2) t4 = *reinterpret_cast<const f32x4*>(lds + HHv * HWv * 64 + c4 * 4);
        else if (p.in_shift) t4 = *reinterpret_cast<const f32x4*>(p.in_shift + ((long)b * p.cin_chunks + c) * 64 + c4 * 4);
      }
      unsigned goff[NK], loff[NK];
      bool cok[NK];
      // GEO: virtual input size, origin shift, reflection and subsampling of the stored map (all scalar)
      const int Hv = GEO ? p.geo_h : H, Wv = GEO ? p.geo_w : W, goffs = GEO ? p.geo_off : 0;
      const bool refl = GEO && p.geo_reflect;
      const int gup = GEO ? p.geo_up : 0;
      const bool sub = GEO && p.geo_sub;  // scalar
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const int col = pcol + 16 * k;
        int gw = SS * w0 - 1 + col - goffs;
        cok[k] = (refl || (gw >= 0 && gw < Wv)) && col < HWv && !(sub && (gw & 1));
        if (refl) gw = gw < 0 ? -gw : (gw >= Wv ? 2 * Wv - 2 - gw : gw);
        goff[k] = (unsigned)((min(max(gw, 0), Wv - 1) >> gup) * (int)p.xv.sW + c4 * 4) * 4u;  // bytes
        loff[k] = col * 64 + ((c4 ^ (col & 15)) << 2);
      }
      // a tile whose halo lies inside the image needs no zero padding: its staging stores skip the masking (4 VALU per piece)
      const bool interior = refl || (!sub && h0 - goffs >= 1 && h0 - goffs + THv + 1 <= Hv && w0 - goffs >= 1 && w0 - goffs + TW + 1 <= Wv);  // scalar
      auto row_bytes = [&](int gh) -> unsigned {  // scalar: byte offset of the stored row behind (shifted) halo row gh
        if (refl) gh = gh < 0 ? -gh : (gh >= Hv ? 2 * Hv - 2 - gh : gh);
        return (unsigned)((min(max(gh, 0), Hv - 1) >> gup) * (int)p.xv.sH) * 4u;
      };
      if (!GATE) {
        f32x4 v[HHv][NK];
#pragma unroll
        for (int r = 0; r < HHv; ++r) {
          const unsigned ro = row_bytes(SS * h0 - 1 + r - goffs);  // scalar, bytes
#pragma unroll
          for (int k = 0; k < NK; ++k)
            if (k < NK - 1 || pcol < 2) v[r][k] = sisr_buf_load4(rx, goff[k], ro);
        }
        if (interior) {
#pragma unroll
          for (int r = 0; r < HHv; ++r)
#pragma unroll
            for (int k = 0; k < NK; ++k)
              if (k < NK - 1 || pcol < 2) {
                f32x4 t = v[r][k];
                if (AFFINE) t = t * s4 + t4;
                *reinterpret_cast<f32x4*>(ldsc + r * (HWv * 64) + loff[k]) = t;
              }
        } else {
#pragma unroll
          for (int r = 0; r < HHv; ++r) {
            const int gh = SS * h0 - 1 + r - goffs;
            const bool rok = gh >= 0 && gh < Hv && !(sub && (gh & 1));  // scalar
#pragma unroll
            for (int k = 0; k < NK; ++k)
              if (k < NK - 1 || pcol < 2) {
                f32x4 t = v[r][k];
                if (AFFINE) t = t * s4 + t4;
                *reinterpret_cast<f32x4*>(ldsc + r * (HWv * 64) + loff[k]) = sisr_keep_if(t, rok && cok[k]);
              }
          }
        }
      } else {  // y = t * gate + skip on the fly; rows in batches of HHv / 2 (two operand tensors in flight)
        const f32x4 g4 = HEAD == 1 ? *reinterpret_cast<const f32x4*>(lds + HHv * HWv * 64 + c4 * 4)
                                   : *reinterpret_cast<const f32x4*>(p.in_scale + (long)b * 64 + c4 * 4);
        const sisr_rsrc_t ru = sisr_rsrc(p.gate_add + (long)b * p.xv.sB);
        const sisr_rsrc_t ro_ = sisr_rsrc(p.gate_out + (long)b * p.xv.sB);
        // 2-row tiles with three resident workgroups (168 registers): both operand maps of all four halo rows are requested
        // at once, one memory round trip per tile instead of two
        constexpr int RB = (MT == 1 && SISR_GATE_WGS == 3) ? HHv : HHv / 2;
        if (SISR_GATE_PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int r0 = 0; r0 < HHv; r0 += RB) {
          f32x4 v[RB][3], u[RB][3];
#pragma unroll
          for (int r = 0; r < RB; ++r) {
            const unsigned ro = (unsigned)(min(max(h0 - 1 + r0 + r, 0), H - 1) * (int)p.xv.sH) * 4u;  // scalar, bytes
#pragma unroll
            for (int k = 0; k < 3; ++k)
              if (k < 2 || pcol < 2) {
                v[r][k] = sisr_buf_load4(rx, goff[k], ro);
                u[r][k] = sisr_buf_load4(ru, goff[k], ro);
              }
          }
#pragma unroll
          for (int r = 0; r < RB; ++r) {
            const int hr = r0 + r, gh = h0 - 1 + hr;
            const bool rok = gh >= 0 && gh < H;               // scalar
            const bool rown = hr >= 1 && hr <= THv && gh < H;  // scalar: a row this tile owns
            const unsigned ro = (unsigned)(min(max(gh, 0), H - 1) * (int)p.xv.sH) * 4u;
#pragma unroll
            for (int k = 0; k < 3; ++k)
              if (k < 2 || pcol < 2) {
                const f32x4 t = sisr_mul_add4(v[r][k], g4, u[r][k]);
                if (interior) *reinterpret_cast<f32x4*>(lds + hr * (HWv * 64) + loff[k]) = t;
                else *reinterpret_cast<f32x4*>(lds + hr * (HWv * 64) + loff[k]) = sisr_keep_if(t, rok && cok[k]);
                const int col = pcol + 16 * k;
                if (rown && cok[k] && col >= 1 && col <= TW) SISR_Y_STORE4(t, ro_, goff[k], ro);
              }
          }
        }
        if (SISR_GATE_PRIO) __builtin_amdgcn_s_setprio(0);
      }
    }
    __syncthreads();
#ifdef SISR_DIAG
    if (p.stamp) st1 = __builtin_amdgcn_s_memtime();
#endif

    // ---- K loop, fully unrolled: 72 steps x 8 MFMAs; every address is lane constant + immediate
    const float* wq = p.w + ((long)q * p.cin_chunks + c) * (9 * 64 * 64);  // scalar
    if constexpr (KSEL != 0) {
      // NJ octets per tap starting at octet J0 (KSEL 1: J0 = 4q, folded into aoff and the weight base); same pipeline
      auto kloop = [&](auto nj_tag, const float* wb) {
        constexpr int NJ = decltype(nj_tag)::value, NS = 9 * NJ;
        const sisr_rsrc_t rwb = sisr_rsrc(wb);
        auto ldb = [&](int s) { return sisr_buf_load4(rwb, boff * 4u, (unsigned)(((s / NJ) * 8 + (s % NJ)) * 2048)); };
        auto lda = [&](int m, int s) {
          return *reinterpret_cast<const f32x4*>(lds + (((s / NJ) / 3 + m) * (HWv * 64)) + aoff[(s / NJ) % 3][s % NJ]);
        };
        f32x4 bq[8];
        f32x4 aq[4][2];
#pragma unroll
        for (int s = 0; s < 6; ++s) bq[s] = ldb(s);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          aq[s][0] = lda(0, s);
          if (MT == 2) aq[s][1] = lda(1, s);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          if (s + 6 < NS) bq[(s + 6) & 7] = ldb(s + 6);
          if (s + 2 < NS) {
            aq[(s + 2) & 3][0] = lda(0, s + 2);
            if (MT == 2) aq[(s + 2) & 3][1] = lda(1, s + 2);
          }
          const f32x4 bb = bq[s & 7];
          const f32x4 a0 = aq[s & 3][0];
          const f32x4 a1 = aq[s & 3][1];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], bb[e], acc0, 0, 0, 0);
            if (MT == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], bb[e], acc1, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      if constexpr (KSEL == 1) {
        kloop(std::integral_constant<int, 4>{}, wq + (long)q * (4 * 512));
      } else if constexpr (KSEL == 4) {
        kloop(std::integral_constant<int, 4>{}, wq);
      } else if constexpr (KSEL == 5) {
        kloop(std::integral_constant<int, 1>{}, wq);
      } else if constexpr (KSEL == 3) {
        kloop(std::integral_constant<int, 8>{}, p.w + ((long)q * p.cin_chunks + ch) * (9 * 64 * 64));
      } else {
        if (c == 0) kloop(std::integral_constant<int, 8>{}, wq);
        else kloop(std::integral_constant<int, 2>{}, wq);
      }
    } else {
#define V4_LOAD_B(s) sisr_buf_load4(rw, boff * 4u, (unsigned)((s) * 2048))
#define V4_LOAD_A(m, s) \
  (*reinterpret_cast<const f32x4*>(lds + ((((s) >> 3) / 3 + (m)) * (HWv * 64)) + aoff[((s) >> 3) % 3][(s) & 7]))
    f32x4 aq[4][2];
    if constexpr (!EARLY_B) {
#pragma unroll
      for (int s = 0; s < 6; ++s) bq[s] = V4_LOAD_B(s);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      aq[s][0] = V4_LOAD_A(0, s);
      if (MT == 2) aq[s][1] = V4_LOAD_A(1, s);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 72; ++s) {
      if (s + 6 < 72) bq[(s + 6) & 7] = V4_LOAD_B(s + 6);
      if (s + 2 < 72) {
        aq[(s + 2) & 3][0] = V4_LOAD_A(0, s + 2);
        if (MT == 2) aq[(s + 2) & 3][1] = V4_LOAD_A(1, s + 2);
      }
      const f32x4 bb = bq[s & 7];
      const f32x4 a0 = aq[s & 3][0];
      const f32x4 a1 = aq[s & 3][1];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], bb[e], acc0, 0, 0, 0);
        if (MT == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], bb[e], acc1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef V4_LOAD_A
#undef V4_LOAD_B
    }
  }

  // ---- epilogue
#ifdef SISR_DIAG
  if (p.stamp) st2 = __builtin_amdgcn_s_memtime();
#endif
  // Every VALU instruction here is paid in matrix-pipe time by the co-resident waves (~4.5 cycles each), so the optional
  // steps are skipped by scalar branches rather than computed with neutral constants, and all addresses are {scalar resource
  // of the tile, lane byte offset, scalar byte offset}.
  float os = p.alpha;
  if (p.out_scale) os *= p.out_scale[(long)b * Cout + q * 64 + co];
  const bool scaled = p.out_scale != nullptr || p.alpha != 1.0f;                    // scalar
  const bool want_sum = p.gap != nullptr;                                           // scalar
  const unsigned loff_y = (unsigned)(co + 4 * hh * (int)p.yv.sW) * 4u;              // lane part of every output address, bytes
  const long tile_base = (long)b * p.yv.sB + p.yv.chunk(q) + (long)w0 * p.yv.sW;    // scalar
  const bool full = (h0 + THv <= H) && (w0 + TW <= W);                              // scalar
  const unsigned swb = (unsigned)p.yv.sW * 4u;                                      // bytes per pixel column
  const sisr_rsrc_t ry = sisr_rsrc(p.y + tile_base);
  const sisr_rsrc_t rmk = sisr_rsrc(MASK ? p.mask + tile_base : p.y);
  const sisr_rsrc_t rrs = sisr_rsrc(RES ? p.res + tile_base : p.y);
  const sisr_rsrc_t rdt = sisr_rsrc(DOT ? p.dot + tile_base : p.y);
  float grow[2] = {0.f, 0.f};  // per output row: the GAP partial is (row 0) + (row 1) of a 2-row strip
  // Epilogue operands (ReLU mask / residual / DOT map) of BOTH output rows are requested before the first store: y, mask,
  // res and dot are distinct buffers, but the compiler cannot know that, and with loads and stores interleaved it emitted
  // load -> s_waitcnt vmcnt(0) -> store per element (a full memory round trip each).  One round trip per tile remains.
  // (With two operand sets -- DOT + residual -- the second set is fetched per row: 64 values in flight measured slower.)
  constexpr bool TWO_SETS = DOT && RES;
  float mk[MT][16], rs[MT][16], dt[MT][16];
  auto fetch = [&](int m, bool first_set, bool second_set) {
    const int row = h0 + MT * ph + m;
    const unsigned row_off = (unsigned)(min(row, H - 1) * (int)p.yv.sH) * 4u;  // scalar, bytes
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cr = (r & 3) + 8 * (r >> 2);
      // partial tiles: unconditional loads from a clamped (in-image) column instead of a branch + wait per element
      const unsigned vo = full ? loff_y : (unsigned)(co + (min(w0 + cr + 4 * hh, W - 1) - w0) * (int)p.yv.sW) * 4u;
      const unsigned so = full ? row_off + (unsigned)cr * swb : row_off;
      if (MASK && first_set) mk[m][r] = sisr_buf_load1(rmk, vo, so);
      if (RES && first_set) rs[m][r] = sisr_buf_load1(rrs, vo, so);
      if (DOT && (TWO_SETS ? second_set : first_set)) dt[m][r] = sisr_buf_load1(rdt, vo, so);
    }
  };
#pragma unroll
  for (int m = 0; m < MT; ++m) fetch(m, true, false);
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    float gsum = 0.f;
    const int row = h0 + MT * ph + m;
    f32x16 acc = m ? acc1 : acc0;
    const unsigned row_off = (unsigned)(row * (int)p.yv.sH) * 4u;  // scalar, bytes (used for in-image rows only)
    if (TWO_SETS) fetch(m, false, true);
    if (LEAKY && !MASK) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = acc[r] > 0.f ? acc[r] : 0.2f * acc[r];
    } else if (p.relu) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
    }
    if (scaled) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] *= os;
    }
    if (MASK) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = mk[m][r] > 0.f ? acc[r] : (LEAKY ? 0.2f * acc[r] : 0.f);
    }
    if (RES) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += rs[m][r];
    }
    if (full) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        SISR_Y_STORE1(acc[r], ry, loff_y, row_off + (unsigned)((r & 3) + 8 * (r >> 2)) * swb);
      if (want_sum) {
#pragma unroll
        for (int r = 0; r < 16; ++r) gsum = DOT ? __builtin_fmaf(acc[r], dt[m][r], gsum) : gsum + acc[r];  // explicit fma: same bits in every build
      }
    } else if (row < H) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cr = (r & 3) + 8 * (r >> 2);
        if (w0 + cr + 4 * hh < W) {
          SISR_Y_STORE1(acc[r], ry, loff_y, row_off + (unsigned)cr * swb);
          gsum = DOT ? __builtin_fmaf(acc[r], dt[m][r], gsum) : gsum + acc[r];  // explicit fma: same bits in every build
        }
      }
    }
    grow[m] = gsum + __shfl_xor(gsum, 32);
  }
  if (p.gap) {
    const long parts = (long)p.tiles_w * ((H + 3) / 4) * 2;  // one partial per (2-row strip, 32 columns)
    if (MT == 2) {
      if (hh == 0) p.gap[(((long)b * parts) + (th * p.tiles_w + tw) * 2 + ph) * Cout + q * 64 + co] = grow[0] + grow[1];
    } else {  // the strip's two rows live in waves ph = 0 and ph = 1: add them through LDS in row order
      __syncthreads();
      if (ph == 1 && hh == 0) lds[co] = grow[0];
      __syncthreads();
      if (ph == 0 && hh == 0) {
        float* g = p.gap + (((long)b * parts) + ((th >> 1) * p.tiles_w + tw) * 2 + (th & 1)) * Cout + q * 64 + co;
        g[0] = grow[0] + lds[co];
        if ((th & 1) == 0 && th + 1 >= p.tiles_h) g[Cout] = 0.f;  // H % 4 in {1, 2}: the tile's second strip is empty
      }
    }
  }
#ifdef SISR_DIAG
  if (p.stamp) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long st3 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
      unsigned* dbg = p.stamp + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
      const unsigned long long rt3 = __builtin_amdgcn_s_memrealtime();
      dbg[0] = (unsigned)(st0 & 0xffffffffu);
      dbg[1] = (unsigned)(rt0 & 0xffffffffu);  // 100 MHz, chip-wide: aligns the workgroups of different XCDs
      dbg[2] = (unsigned)(st1 - st0);
      dbg[3] = (unsigned)(st2 - st1);
      dbg[4] = (unsigned)(st3 - st2);
      dbg[5] = __builtin_amdgcn_s_getreg(4 | (31 << 11));   // HW_REG_HW_ID
      dbg[6] = __builtin_amdgcn_s_getreg(20 | (31 << 11));  // HW_REG_XCC_ID
      dbg[7] = (unsigned)(rt3 - rt0);
    }
  }
#endif
  if (HEAD == 0 && (p.fwd_tail.g || p.bwd_tail.shift)) {  // uniform
    const long parts = (long)p.tiles_w * ((H + 3) / 4) * 2;
    const unsigned per_sample = gridDim.x / (unsigned)p.B;  // workgroups of one sample (cout_chunks == 1 here)
    unsigned* cnt = p.fwd_tail.g ? p.fwd_tail.counter : p.bwd_tail.counter;
    int* flag = reinterpret_cast<int*>(lds);
    float* red = lds + 64;
    __threadfence();  // this workgroup's partial sums are visible device-wide before it is counted
    __syncthreads();  // ... and every wave is done with the halo / strip exchange in LDS
    if (tid == 0) flag[0] = atomicAdd(cnt + b, 1u) == per_sample - 1;
    __syncthreads();
    if (!flag[0]) return;
    __threadfence();
    if (tid == 0) cnt[b] = 0u;
    if (p.fwd_tail.g) {
      ca_gate_fwd_sample<true>(p.gap, (int)parts, p.fwd_tail.inv_hw, b, p.fwd_tail.w1, p.fwd_tail.b1, p.fwd_tail.w2,
                               p.fwd_tail.b2, p.fwd_tail.R, p.fwd_tail.mul, p.fwd_tail.s, p.fwd_tail.hid, p.fwd_tail.ca,
                               p.fwd_tail.g, red);
    } else {
      ca_gate_bwd_sample<true>(p.gap, (int)parts, p.bwd_tail.inv_hw, b, p.bwd_tail.w1, p.bwd_tail.w2, p.bwd_tail.R,
                               p.bwd_tail.hid, p.bwd_tail.ca, p.bwd_tail.mul, p.bwd_tail.shift, p.bwd_tail.dmul,
                               p.bwd_tail.dz2, p.bwd_tail.dz1, red);
      __threadfence();  // this sample's dz2 / dz1 before it is counted on the batch counter
      __syncthreads();
      if (tid == 0) flag[0] = atomicAdd(cnt + p.B, 1u) == (unsigned)(p.B - 1);
      __syncthreads();
      if (!flag[0]) return;
      __threadfence();
      if (tid == 0) cnt[p.B] = 0u;
      ca_gate_bwd_params(p.bwd_tail.dz2, p.bwd_tail.dz1, p.bwd_tail.hid, p.bwd_tail.s, p.bwd_tail.R, p.B, p.bwd_tail.dw1,
                         p.bwd_tail.db1, p.bwd_tail.dw2, p.bwd_tail.db2);
    }
  }
}

// ------------------------------------------------------------------ persistent form of the fp32 kernel (round 3)
// conv3x3_c64_v4_kernel leaves the overlap of one tile's staging / epilogue with another tile's K loop to whatever
// workgroups happen to share the CU: per-CU timelines (tools/conv_timeline.py) show K phases running back to back at 97 % of
// the MFMA rate with a 0.3 - 1.6 us hole at every hand-over (the waiting workgroup's last staging instructions only issue
// once the older wave's MFMA stream stops), a ramp per launch and a ragged last round.  Here 512 workgroups (two per CU, 256
// registers each) walk the tiles of a launch: the halo of tile i + 1 is REQUESTED before the K loop of tile i (18 float4 per
// thread in registers, 36 with the GATE prologue's second map) and written to LDS after it, so a workgroup waits for memory
// once per launch and its only non-MFMA time per tile is epilogue + LDS writes, which the other resident workgroup's K loop
// covers.  Same tile geometry, LDS image, K loop, prologue / epilogue arithmetic and summation order as the per-tile kernel:
// bit-identical results.  64 -> 64 maps only (one input chunk); grids too small to give every workgroup two tiles, gate
// heads / tails, LeakyReLU and the sparse selections stay on the per-tile kernel.
template <bool AFFINE, bool MASK, bool RES, bool GATE, bool DOT>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_p4_kernel(ConvParams p, int total_tiles) {
  constexpr int MT = 2, THv = 4, HHv = 6;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ph = __builtin_amdgcn_readfirstlane(wave >> 1), ch = __builtin_amdgcn_readfirstlane(wave & 1);
  const int n = lane & 31, hh = lane >> 5;
  const int H = p.H, W = p.W;
  const int co = ch * 32 + n;
  const float bv = p.bias ? p.bias[co * p.bias_n] : 0.f;

  unsigned aoff[3][8];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      aoff[kw][j] = ((MT * ph) * HALO_W + n + kw) * 64 + (((2 * j + hh) ^ ((n + kw) & 15)) << 2);
  const unsigned boff = hh * 256 + co * 4;
  const sisr_rsrc_t rw = sisr_rsrc(p.w);

  // tile walk: every XCD sweeps a contiguous eighth of the tiles (vertically adjacent tiles share halo rows through one L2)
  const int G = gridDim.x;
  int t_begin = blockIdx.x, t_end = total_tiles, t_step = G;
  if ((G & 7) == 0) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per = (total_tiles + 7) >> 3;
    t_begin = xcd * per + idx;
    t_end = min(total_tiles, (xcd + 1) * per);
    t_step = G >> 3;
  }
  auto decode = [&](int tile, int& b, int& h0, int& w0) {
    const int tw = tile % p.tiles_w;
    const int t2 = tile / p.tiles_w;
    b = t2 / p.tiles_h;
    h0 = (t2 - b * p.tiles_h) * THv;
    w0 = tw * TW;
  };
  // staging thread map: (16-B piece c4 of a pixel, halo column pcol + {0, 16, 32}), rows 0..5
  const int c4 = tid & 15, pcol = tid >> 4;
  f32x4 v[HHv][3];
  auto issue = [&](int tile) {
    int b, h0, w0;
    decode(tile, b, h0, w0);
    const sisr_rsrc_t rx = sisr_rsrc(p.x + (long)b * p.xv.sB);
#pragma unroll
    for (int r = 0; r < HHv; ++r) {
      const unsigned ro = (unsigned)(min(max(h0 - 1 + r, 0), H - 1) * (int)p.xv.sH) * 4u;  // scalar, bytes
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k < 2 || pcol < 2) {
          const unsigned go = (unsigned)(min(max(w0 - 1 + pcol + 16 * k, 0), W - 1) * (int)p.xv.sW + c4 * 4) * 4u;
          v[r][k] = sisr_buf_load4(rx, go, ro);
        }
    }
  };
  auto commit = [&](int tile) {
    int b, h0, w0;
    decode(tile, b, h0, w0);
    const bool interior = h0 >= 1 && h0 + THv + 1 <= H && w0 >= 1 && w0 + TW + 1 <= W;  // scalar
    f32x4 s4 = {1.f, 1.f, 1.f, 1.f}, t4 = {0.f, 0.f, 0.f, 0.f};
    if (AFFINE || GATE) s4 = *reinterpret_cast<const f32x4*>(p.in_scale + (long)b * 64 + c4 * 4);
    if (AFFINE && p.in_shift) t4 = *reinterpret_cast<const f32x4*>(p.in_shift + (long)b * 64 + c4 * 4);
    const sisr_rsrc_t ro_ = sisr_rsrc(GATE ? p.gate_out + (long)b * p.xv.sB : p.y);
    // GATE: the skip map is fetched here, not a tile ahead (two prefetched maps would not fit 256 registers beside the K
    // loop's); its round trip is covered by the other resident workgroup's K loop
    f32x4 u[GATE ? HHv : 1][3];
    if (GATE) {
      const sisr_rsrc_t ru = sisr_rsrc(p.gate_add + (long)b * p.xv.sB);
#pragma unroll
      for (int r = 0; r < HHv; ++r) {
        const unsigned ro = (unsigned)(min(max(h0 - 1 + r, 0), H - 1) * (int)p.xv.sH) * 4u;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (k < 2 || pcol < 2)
            u[r][k] = sisr_buf_load4(ru, (unsigned)(min(max(w0 - 1 + pcol + 16 * k, 0), W - 1) * (int)p.xv.sW + c4 * 4) * 4u, ro);
      }
    }
#pragma unroll
    for (int r = 0; r < HHv; ++r) {
      const int gh = h0 - 1 + r;
      const bool rok = gh >= 0 && gh < H;                 // scalar
      const bool rown = r >= 1 && r <= THv && gh < H;      // scalar: a row this tile owns
      const unsigned ro = (unsigned)(min(max(gh, 0), H - 1) * (int)p.xv.sH) * 4u;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k < 2 || pcol < 2) {
          const int col = pcol + 16 * k, gw = w0 - 1 + col;
          const bool cok = gw >= 0 && gw < W && col < HALO_W;
          f32x4 t = v[r][k];
          if (AFFINE) t = t * s4 + t4;
          if (GATE) {
            t = sisr_mul_add4(t, s4, u[r][k]);
            if (rown && cok && col >= 1 && col <= TW)
              SISR_Y_STORE4(t, ro_, (unsigned)(min(max(gw, 0), W - 1) * (int)p.xv.sW + c4 * 4) * 4u, ro);
          }
          if (!interior) t = sisr_keep_if(t, rok && cok);
          *reinterpret_cast<f32x4*>(lds + r * (HALO_W * 64) + col * 64 + ((c4 ^ (col & 15)) << 2)) = t;
        }
    }
  };

  if (t_begin < t_end) {
    issue(t_begin);
    commit(t_begin);
  }
  __syncthreads();
  const int Cout = 64;
  for (int tile = t_begin; tile < t_end; tile += t_step) {
    const bool has_next = tile + t_step < t_end;  // uniform
    int b, h0, w0;
    decode(tile, b, h0, w0);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = bv;
    f32x4 bq[8];
#pragma unroll
    for (int s = 0; s < 6; ++s) bq[s] = sisr_buf_load4(rw, boff * 4u, (unsigned)(s * 2048));
    if (has_next) issue(tile + t_step);  // in flight across the K loop and the epilogue of this tile
    {
#define P4_LOAD_B(s) sisr_buf_load4(rw, boff * 4u, (unsigned)((s) * 2048))
#define P4_LOAD_A(m, s) \
  (*reinterpret_cast<const f32x4*>(lds + ((((s) >> 3) / 3 + (m)) * (HALO_W * 64)) + aoff[((s) >> 3) % 3][(s) & 7]))
      f32x4 aq[4][2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        aq[s][0] = P4_LOAD_A(0, s);
        aq[s][1] = P4_LOAD_A(1, s);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 72; ++s) {
        if (s + 6 < 72) bq[(s + 6) & 7] = P4_LOAD_B(s + 6);
        if (s + 2 < 72) {
          aq[(s + 2) & 3][0] = P4_LOAD_A(0, s + 2);
          aq[(s + 2) & 3][1] = P4_LOAD_A(1, s + 2);
        }
        const f32x4 bb = bq[s & 7];
        const f32x4 a0 = aq[s & 3][0];
        const f32x4 a1 = aq[s & 3][1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], bb[e], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], bb[e], acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#undef P4_LOAD_A
#undef P4_LOAD_B
    }
    __syncthreads();  // every wave is done with this tile's halo
    // ---- epilogue (the per-tile kernel's, see there)
    {
      float os = p.alpha;
      if (p.out_scale) os *= p.out_scale[(long)b * Cout + co];
      const bool scaled = p.out_scale != nullptr || p.alpha != 1.0f;
      const bool want_sum = p.gap != nullptr;
      const unsigned loff_y = (unsigned)(co + 4 * hh * (int)p.yv.sW) * 4u;
      const long tile_base = (long)b * p.yv.sB + (long)w0 * p.yv.sW;
      const bool full = (h0 + THv <= H) && (w0 + TW <= W);
      const unsigned swb = (unsigned)p.yv.sW * 4u;
      const sisr_rsrc_t ry = sisr_rsrc(p.y + tile_base);
      const sisr_rsrc_t rmk = sisr_rsrc(MASK ? p.mask + tile_base : p.y);
      const sisr_rsrc_t rrs = sisr_rsrc(RES ? p.res + tile_base : p.y);
      const sisr_rsrc_t rdt = sisr_rsrc(DOT ? p.dot + tile_base : p.y);
      float grow[2] = {0.f, 0.f};
      constexpr bool TWO_SETS = DOT && RES;
      float mk[MT][16], rs[MT][16], dt[MT][16];
      auto fetch = [&](int m, bool first_set, bool second_set) {
        const int row = h0 + MT * ph + m;
        const unsigned row_off = (unsigned)(min(row, H - 1) * (int)p.yv.sH) * 4u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cr = (r & 3) + 8 * (r >> 2);
          const unsigned vo = full ? loff_y : (unsigned)(co + (min(w0 + cr + 4 * hh, W - 1) - w0) * (int)p.yv.sW) * 4u;
          const unsigned so = full ? row_off + (unsigned)cr * swb : row_off;
          if (MASK && first_set) mk[m][r] = sisr_buf_load1(rmk, vo, so);
          if (RES && first_set) rs[m][r] = sisr_buf_load1(rrs, vo, so);
          if (DOT && (TWO_SETS ? second_set : first_set)) dt[m][r] = sisr_buf_load1(rdt, vo, so);
        }
      };
#pragma unroll
      for (int m = 0; m < MT; ++m) fetch(m, true, false);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        float gsum = 0.f;
        const int row = h0 + MT * ph + m;
        f32x16 acc = m ? acc1 : acc0;
        const unsigned row_off = (unsigned)(row * (int)p.yv.sH) * 4u;
        if (TWO_SETS) fetch(m, false, true);
        if (p.relu) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
        }
        if (scaled) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] *= os;
        }
        if (MASK) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = mk[m][r] > 0.f ? acc[r] : 0.f;
        }
        if (RES) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] += rs[m][r];
        }
        if (full) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            SISR_Y_STORE1(acc[r], ry, loff_y, row_off + (unsigned)((r & 3) + 8 * (r >> 2)) * swb);
          if (want_sum) {
#pragma unroll
            for (int r = 0; r < 16; ++r) gsum = DOT ? __builtin_fmaf(acc[r], dt[m][r], gsum) : gsum + acc[r];
          }
        } else if (row < H) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int cr = (r & 3) + 8 * (r >> 2);
            if (w0 + cr + 4 * hh < W) {
              SISR_Y_STORE1(acc[r], ry, loff_y, row_off + (unsigned)cr * swb);
              gsum = DOT ? __builtin_fmaf(acc[r], dt[m][r], gsum) : gsum + acc[r];
            }
          }
        }
        grow[m] = gsum + __shfl_xor(gsum, 32);
      }
      if (p.gap) {
        const long parts = (long)p.tiles_w * ((H + 3) / 4) * 2;  // one partial per (2-row strip, 32 columns)
        const int th = h0 / THv, tw = w0 / TW;
        if (hh == 0) p.gap[(((long)b * parts) + (th * p.tiles_w + tw) * 2 + ph) * Cout + co] = grow[0] + grow[1];
      }
    }
    if (has_next) {
      commit(tile + t_step);
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------ bf16 matrix-core kernel (fp32 in HBM)
// Same tile / wave geometry, View addressing and epilogue as the fp32 kernels, but the contraction runs on
// v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate): activations are rounded to bf16 (RNE, v_cvt_pk_bf16_f32)
// when the halo is staged into LDS -- after the optional per-(b, c) affine prologue -- weights are rounded once
// at pack time, products are exact and accumulation stays fp32.  Feature maps remain fp32 in HBM, so every other
// kernel of the step is unchanged; the conv itself becomes HBM-bound (268 MB per 64->64 launch at B = 32 against
// 15 us of MFMA time).
//   LDS image: halo pixel = 64 bf16 = 128 B; its 16-B chunk k (channels 8k..8k+7) sits at slot
//   k ^ ((col >> 1) & 7): the A fragment of lane (pixel column n, half h) for K-step (tap, 16-channel block kb)
//   is the single ds_read_b128 of chunk 2kb+h, and 16 consecutive columns then cover all 64 banks once.
//   B fragments ([q][c][36 K-steps][h][co][8] bf16, 2 KB per K-step) stream from L2 one global_load_dwordx4 per
//   K-step, prefetched four steps ahead.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ u32x4 sisr_pack_bf16x8(f32x4 a, f32x4 b) {
  bf16x8 r;
  r[0] = (__bf16)a[0]; r[1] = (__bf16)a[1]; r[2] = (__bf16)a[2]; r[3] = (__bf16)a[3];
  r[4] = (__bf16)b[0]; r[5] = (__bf16)b[1]; r[6] = (__bf16)b[2]; r[7] = (__bf16)b[3];
  return __builtin_bit_cast(u32x4, r);
}

__device__ __forceinline__ bf16x8 sisr_buf_load_bf16x8(sisr_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
  return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff_bytes, (int)soff_bytes, 0));
}

#define BH_PIX 128  // bytes per halo pixel
#define BE_LD 68    // floats per pixel row of the epilogue transpose buffer (64 + 4: rows 272 B apart)

template <bool AFFINE, bool MASK, bool RES, bool GATE = false, bool DOT = false>
__global__ __launch_bounds__(256, 3) void conv3x3_c64_bf16_kernel(ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = blockIdx.y;
  int bid;
  {
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const unsigned qn = nb >> 3, rn = nb & 7;
    bid = (int)((xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx);
  }
  const int tw = bid % p.tiles_w;
  bid /= p.tiles_w;
  const int th = bid % p.tiles_h;
  const int b = bid / p.tiles_h;
  const int h0 = th * TH, w0 = tw * TW;
  const int ph = __builtin_amdgcn_readfirstlane(wave >> 1), ch = __builtin_amdgcn_readfirstlane(wave & 1);
  const int n = lane & 31, hh = lane >> 5;
  const int H = p.H, W = p.W;
  const int co = ch * 32 + n;
  const int Cout = p.cout_chunks * 64;

  const float bv = p.bias ? p.bias[co * p.bias_n + q * p.bias_q] : 0.f;
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = bv;

  // lane constants of the A reads: byte offset of (halo row 2ph, column n+kw, chunk 2kb+hh)
  unsigned aoff[3][4];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
      aoff[kw][kb] = ((2 * ph) * HALO_W + n + kw) * BH_PIX + (((2 * kb + hh) ^ (((n + kw) >> 1) & 7)) << 4);
  const unsigned boff = (hh * 64 + co) * 16;  // bytes

  const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.w);
  for (int c = 0; c < p.cin_chunks; ++c) {
    if (c) __syncthreads();
    {  // ---- halo staging: thread = (8-channel chunk c8, column pcol [+32]); rows in two batches of three
      int tl = tid;
      asm volatile("" : "+v"(tl));
      const int c8 = tl & 7, pcol = tl >> 3;
      const float* xb = p.x + (long)b * p.xv.sB + p.xv.chunk(c);
      f32x4 s4a = {1.f, 1.f, 1.f, 1.f}, s4b = s4a, t4a = {0.f, 0.f, 0.f, 0.f}, t4b = t4a;
      if (AFFINE) {
        const float* sp = p.in_scale + ((long)b * p.cin_chunks + c) * 64 + c8 * 8;
        s4a = *reinterpret_cast<const f32x4*>(sp);
        s4b = *reinterpret_cast<const f32x4*>(sp + 4);
        if (p.in_shift) {
          const float* tp = p.in_shift + ((long)b * p.cin_chunks + c) * 64 + c8 * 8;
          t4a = *reinterpret_cast<const f32x4*>(tp);
          t4b = *reinterpret_cast<const f32x4*>(tp + 4);
        }
      }
      unsigned goff[2], loff[2];
      bool cok[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int col = pcol + 32 * k;
        const int gw = w0 - 1 + col;
        cok[k] = gw >= 0 && gw < W && col < HALO_W;
        goff[k] = (unsigned)(min(max(gw, 0), W - 1) * (int)p.xv.sW + c8 * 8);
        loff[k] = col * BH_PIX + ((c8 ^ ((col >> 1) & 7)) << 4);
      }
      if (!GATE) {
#pragma unroll
        for (int r0 = 0; r0 < HALO_H; r0 += 3) {
          f32x4 v[3][2][2];
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const int gh = h0 - 1 + r0 + r;
            const float* xrow = xb + (long)min(max(gh, 0), H - 1) * p.xv.sH;  // scalar
#pragma unroll
            for (int k = 0; k < 2; ++k)
              if (k == 0 || pcol < 2) {
                v[r][k][0] = *reinterpret_cast<const f32x4*>(xrow + goff[k]);
                v[r][k][1] = *reinterpret_cast<const f32x4*>(xrow + goff[k] + 4);
              }
          }
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const int gh = h0 - 1 + r0 + r;
            const bool rok = gh >= 0 && gh < H;  // scalar
#pragma unroll
            for (int k = 0; k < 2; ++k)
              if (k == 0 || pcol < 2) {
                f32x4 ta = v[r][k][0], tb = v[r][k][1];
                if (AFFINE) {
                  ta = ta * s4a + t4a;
                  tb = tb * s4b + t4b;
                }
                u32x4 pk = sisr_pack_bf16x8(ta, tb);
                const unsigned m = (rok && cok[k]) ? 0xffffffffu : 0u;
                pk &= (u32x4){m, m, m, m};
                *reinterpret_cast<u32x4*>(ldsb + (r0 + r) * (HALO_W * BH_PIX) + loff[k]) = pk;
              }
          }
        }
      } else {  // GATE: u = t * gate + skip in fp32, written out once by the owning tile, then rounded for the MFMA
        const float* gp = p.in_scale + (long)b * 64 + c8 * 8;
        const f32x4 g4a = *reinterpret_cast<const f32x4*>(gp), g4b = *reinterpret_cast<const f32x4*>(gp + 4);
        const long boffs = (long)b * p.xv.sB;
#pragma unroll
        for (int r0 = 0; r0 < HALO_H; r0 += 2) {
          f32x4 v[2][2][2], u[2][2][2];
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const long ro = boffs + (long)min(max(h0 - 1 + r0 + r, 0), H - 1) * p.xv.sH;  // scalar
#pragma unroll
            for (int k = 0; k < 2; ++k)
              if (k == 0 || pcol < 2) {
                v[r][k][0] = *reinterpret_cast<const f32x4*>(p.x + ro + goff[k]);
                v[r][k][1] = *reinterpret_cast<const f32x4*>(p.x + ro + goff[k] + 4);
                u[r][k][0] = *reinterpret_cast<const f32x4*>(p.gate_add + ro + goff[k]);
                u[r][k][1] = *reinterpret_cast<const f32x4*>(p.gate_add + ro + goff[k] + 4);
              }
          }
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int hr = r0 + r, gh = h0 - 1 + hr;
            const bool rok = gh >= 0 && gh < H;              // scalar
            const bool rown = hr >= 1 && hr <= TH && gh < H;  // scalar
            const long ro = boffs + (long)min(max(gh, 0), H - 1) * p.xv.sH;
#pragma unroll
            for (int k = 0; k < 2; ++k)
              if (k == 0 || pcol < 2) {
                const f32x4 ta = sisr_mul_add4(v[r][k][0], g4a, u[r][k][0]), tb = sisr_mul_add4(v[r][k][1], g4b, u[r][k][1]);
                const int col = pcol + 32 * k;
                if (rown && cok[k] && col >= 1 && col <= TW) {
                  *reinterpret_cast<f32x4*>(p.gate_out + ro + goff[k]) = ta;
                  *reinterpret_cast<f32x4*>(p.gate_out + ro + goff[k] + 4) = tb;
                }
                u32x4 pk = sisr_pack_bf16x8(ta, tb);
                const unsigned m = (rok && cok[k]) ? 0xffffffffu : 0u;
                pk &= (u32x4){m, m, m, m};
                *reinterpret_cast<u32x4*>(ldsb + hr * (HALO_W * BH_PIX) + loff[k]) = pk;
              }
          }
        }
      }
    }
    __syncthreads();

    // ---- K loop: 36 steps (tap t = s >> 2, 16-channel block kb = s & 3) x 2 MFMAs
    const unsigned char* wq = wbase + ((long)q * p.cin_chunks + c) * (36 * 2048);  // scalar
    const sisr_rsrc_t rwq = sisr_rsrc(wq);  // weight fragments by {resource, lane offset, scalar offset}: sisr_common.h
#define BF_LOAD_B(s) sisr_buf_load_bf16x8(rwq, boff, (unsigned)((s) * 2048))
#define BF_LOAD_A(m, s) \
  (*reinterpret_cast<const bf16x8*>(ldsb + ((((s) >> 2) / 3 + (m)) * (HALO_W * BH_PIX)) + aoff[((s) >> 2) % 3][(s) & 3]))
    bf16x8 bq[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) bq[s] = BF_LOAD_B(s);
#pragma unroll
    for (int s = 0; s < 36; ++s) {
      const bf16x8 bb = bq[s & 3];
      if (s + 4 < 36) bq[s & 3] = BF_LOAD_B(s + 4);
      const bf16x8 a0 = BF_LOAD_A(0, s);
      const bf16x8 a1 = BF_LOAD_A(1, s);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bb, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bb, acc1, 0, 0, 0);
    }
#undef BF_LOAD_A
#undef BF_LOAD_B
  }

  // ---- epilogue.  The kernel is HBM-bound, so output / mask / residual traffic must move as whole 256-B pixel
  // rows: the accumulators (column = cout on the lane, 16 pixel columns in registers) are transposed through LDS
  // and every thread then handles float4 pieces of eight pixels, all of its loads in flight at once.
  float os = p.alpha;
  if (p.out_scale) os *= p.out_scale[(long)b * Cout + q * 64 + co];
  const float lo = p.relu ? 0.f : -3.402823466e38f;
  float* ot = reinterpret_cast<float*>(ldsb);  // [TH*TW pixels][64] fp32, row stride BE_LD
  __syncthreads();                              // every wave is done reading the halo
  float grow[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    float gsum = 0.f;
    const f32x16 acc = m ? acc1 : acc0;
    const int prow = (2 * ph + m) * TW;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pc = (r & 3) + 8 * (r >> 2) + 4 * hh;
      const float v = fmaxf(acc[r], lo) * os;
      ot[(prow + pc) * BE_LD + co] = v;
      if (h0 + 2 * ph + m < H && w0 + pc < W) gsum += v;
    }
    grow[m] = gsum + __shfl_xor(gsum, 32);
  }
  if (!DOT && p.gap) {  // only offered without mask / residual (host checks): v above is the final value
    if (hh == 0) {
      const int tile = th * p.tiles_w + tw;
      const long parts = (long)p.tiles_w * p.tiles_h * 2;
      p.gap[(((long)b * parts) + tile * 2 + ph) * Cout + q * 64 + co] = grow[0] + grow[1];
    }
  }
  __syncthreads();
  {
    const int c4 = tid & 15, pr = tid >> 4;  // pixel pr + 16 i  ->  tile row i >> 1, column pr + 16 (i & 1)
    const long tile_base = (long)b * p.yv.sB + p.yv.chunk(q) + (long)w0 * p.yv.sW + c4 * 4;
    f32x4 rv[8], mv[8], dv[8];
    bool ok[8];
    long off[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = h0 + (i >> 1), col = pr + 16 * (i & 1);
      ok[i] = row < H && w0 + col < W;
      off[i] = tile_base + (long)min(row, H - 1) * p.yv.sH + (long)min(col, W - 1 - w0) * p.yv.sW;
      if (RES) rv[i] = *reinterpret_cast<const f32x4*>(p.res + off[i]);
      if (MASK) mv[i] = *reinterpret_cast<const f32x4*>(p.mask + off[i]);
      if (DOT) dv[i] = *reinterpret_cast<const f32x4*>(p.dot + off[i]);
    }
    f32x4 dsum[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};  // DOT: per 2-row strip (i < 4 / i >= 4)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f32x4 v = *reinterpret_cast<const f32x4*>(ot + (pr + 16 * i) * BE_LD + c4 * 4);
      if (MASK) {
        v[0] = mv[i][0] > 0.f ? v[0] : 0.f;
        v[1] = mv[i][1] > 0.f ? v[1] : 0.f;
        v[2] = mv[i][2] > 0.f ? v[2] : 0.f;
        v[3] = mv[i][3] > 0.f ? v[3] : 0.f;
      }
      if (RES) v += rv[i];
      if (ok[i]) {
        *reinterpret_cast<f32x4*>(p.y + off[i]) = v;
        if (DOT) dsum[i >> 2] += v * dv[i];
      }
    }
    if (DOT) {  // sum(v * dot) per strip and channel: the 16 pixel-column threads of a channel quad, in order
      __syncthreads();  // everyone is done with the transpose buffer
      float* red = ot;  // [2 strips][16][64]
      *reinterpret_cast<f32x4*>(red + (0 * 16 + pr) * 64 + c4 * 4) = dsum[0];
      *reinterpret_cast<f32x4*>(red + (1 * 16 + pr) * 64 + c4 * 4) = dsum[1];
      __syncthreads();
      if (tid < 128) {
        const int strip = tid >> 6, chn = tid & 63;
        float sacc = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) sacc += red[(strip * 16 + j) * 64 + chn];
        const int tile = th * p.tiles_w + tw;
        const long parts = (long)p.tiles_w * p.tiles_h * 2;
        p.gap[(((long)b * parts) + tile * 2 + strip) * Cout + q * 64 + chn] = sacc;
      }
    }
  }
}

// ------------------------------------------------------------------ fp32 through the bf16 matrix cores ("bf16x3")
// Every fp32 value is the exact sum of three bf16 numbers  x = hi + mid + lo  (24 = 3 x 8 significand bits:
// hi = rne(x), mid = rne(x - hi), lo = x - hi - mid, each difference exact in fp32).  A product x*w is then the sum
// of nine exact bf16 x bf16 products; the six with weight >= 2^-16 relative -- hi*hi, hi*mid, mid*hi, hi*lo, lo*hi,
// mid*mid -- run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (the three dropped ones are <= 2^-24 |x||w| each,
// the size of fp32's own rounding of the product), at 6/16 of the fp32 MFMA's cycle count.  hi*hi accumulates into
// its own register set and the five corrections into another, added once at the end, so the small terms are not
// rounded against the large running sum.  Opt-in (SISR_PRECISION=bf16x3); the headline path stays exact fp32.
// Same tile / wave geometry, staging thread map, LDS swizzle and epilogue as conv3x3_c64_bf16_kernel; the halo is
// staged as three bf16 planes (78 KB, two workgroups per CU) and the packed weights as three planes `wplane` bytes apart.
#define X3_PLANE (HALO_H * HALO_W * BH_PIX)

__device__ __forceinline__ void sisr_split3(f32x4 a, f32x4 b, u32x4& hi, u32x4& mid, u32x4& lo) {
  bf16x8 h, m, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = e < 4 ? a[e] : b[e - 4];
    const __bf16 xh = (__bf16)x;
    const float r1 = x - (float)xh;
    const __bf16 xm = (__bf16)r1;
    const float r2 = r1 - (float)xm;
    h[e] = xh;
    m[e] = xm;
    l[e] = (__bf16)r2;
  }
  hi = __builtin_bit_cast(u32x4, h);
  mid = __builtin_bit_cast(u32x4, m);
  lo = __builtin_bit_cast(u32x4, l);
}

__device__ __forceinline__ void sisr_store_split3(unsigned char* dst, f32x4 a, f32x4 b, unsigned mask) {
  u32x4 hi, mid, lo;
  sisr_split3(a, b, hi, mid, lo);
  const u32x4 mk = {mask, mask, mask, mask};
  *reinterpret_cast<u32x4*>(dst) = hi & mk;
  *reinterpret_cast<u32x4*>(dst + X3_PLANE) = mid & mk;
  *reinterpret_cast<u32x4*>(dst + 2 * X3_PLANE) = lo & mk;
}

// BD = B-fragment prefetch distance in K-steps (ring of BD + 1 slots); the A fragments of step s + 1 are requested before
// the MFMAs of step s (double buffer).
template <bool AFFINE, bool MASK, bool RES, bool GATE = false, bool DOT = false, int BD = 4, bool STAMP = false>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_x3_kernel(ConvParams p, long wplane) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  // Wave priority: the two resident workgroups take turns in the K loop (the older wave's MFMA stream wins the SIMD's
  // arbiter), and per-CU timelines from in-kernel stamps showed the waiting workgroup's staging -- ~1 500 vector
  // instructions of operand splitting -- crawling in the leftover slots and finishing ~2 us AFTER the pipe had freed up
  // (K phases of 8.8 us with 2 us holes).  Unlike the fp32 MFMA, the bf16 MFMA holds the issue port for 8 of its 32 cycles
  // only, so staging and epilogue at raised priority fit into the gaps of the other workgroup's MFMA stream.
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = blockIdx.y;
  int bid;
  {
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const unsigned qn = nb >> 3, rn = nb & 7;
    bid = (int)((xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx);
  }
  const int tw = bid % p.tiles_w;
  bid /= p.tiles_w;
  const int th = bid % p.tiles_h;
  const int b = bid / p.tiles_h;
  const int h0 = th * TH, w0 = tw * TW;
  const int ph = __builtin_amdgcn_readfirstlane(wave >> 1), ch = __builtin_amdgcn_readfirstlane(wave & 1);
  const int n = lane & 31, hh = lane >> 5;
  const int H = p.H, W = p.W;
  const int co = ch * 32 + n;
  const int Cout = p.cout_chunks * 64;

  const float bv = p.bias ? p.bias[co * p.bias_n + q * p.bias_q] : 0.f;
  unsigned long long st0 = 0, st1 = 0, st2 = 0, rt0 = 0;  // STAMP: diagnostic build (tools/conv_timeline.py x3), outputs meaningless
  if (STAMP) {
    st0 = __builtin_amdgcn_s_memtime();
    rt0 = __builtin_amdgcn_s_memrealtime();
  }
  f32x16 acc0, acc1, cor0 = {0}, cor1 = {0};  // hi*hi products / the five correction products
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = bv;

  // lane constants of the A reads: byte offset of (halo row 2ph, column n+kw, chunk 2kb+hh)
  unsigned aoff[3][4];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
      aoff[kw][kb] = ((2 * ph) * HALO_W + n + kw) * BH_PIX + (((2 * kb + hh) ^ (((n + kw) >> 1) & 7)) << 4);
  const unsigned boff = (hh * 64 + co) * 16;  // bytes

  const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.w);
  for (int c = 0; c < p.cin_chunks; ++c) {
    if (c) __syncthreads();
    {  // ---- halo staging, balanced over the waves (the split costs ~7 VALU per element, and VALU issue is what a
       // co-resident wave's MFMA stream leaves over): thread (c8, pcol) owns channels 8 c8 .. +7 of INTERIOR column
       // pcol + 1 in all six rows; the two edge columns (6 rows x 2 x 8 chunks = 96 items) go one per thread to
       // tid < 96.  Every load is unconditional (clamped address, masked value).
      int tl = tid;
      asm volatile("" : "+v"(tl));
      const int c8 = tl & 7, pcol = tl >> 3;
      const int eidx = tl % 96, er = eidx >> 4, eside = (eidx >> 3) & 1, ec8 = eidx & 7;
      const float* xb = p.x + (long)b * p.xv.sB + p.xv.chunk(c);
      const long boffs = (long)b * p.xv.sB;
      f32x4 s4a = {1.f, 1.f, 1.f, 1.f}, s4b = s4a, t4a = {0.f, 0.f, 0.f, 0.f}, t4b = t4a, e4a = s4a, e4b = s4a,
            f4a = t4a, f4b = t4a;
      if (AFFINE || GATE) {
        const float* sp = p.in_scale + ((long)b * p.cin_chunks + c) * 64;
        s4a = *reinterpret_cast<const f32x4*>(sp + c8 * 8);
        s4b = *reinterpret_cast<const f32x4*>(sp + c8 * 8 + 4);
        e4a = *reinterpret_cast<const f32x4*>(sp + ec8 * 8);
        e4b = *reinterpret_cast<const f32x4*>(sp + ec8 * 8 + 4);
        if (AFFINE && p.in_shift) {
          const float* tp = p.in_shift + ((long)b * p.cin_chunks + c) * 64;
          t4a = *reinterpret_cast<const f32x4*>(tp + c8 * 8);
          t4b = *reinterpret_cast<const f32x4*>(tp + c8 * 8 + 4);
          f4a = *reinterpret_cast<const f32x4*>(tp + ec8 * 8);
          f4b = *reinterpret_cast<const f32x4*>(tp + ec8 * 8 + 4);
        }
      }
      const int gw = w0 + pcol;  // interior column pcol + 1
      const bool cok = gw < W;
      const unsigned goff = (unsigned)(min(gw, W - 1) * (int)p.xv.sW + c8 * 8);
      const unsigned loff = (pcol + 1) * BH_PIX + ((c8 ^ (((pcol + 1) >> 1) & 7)) << 4);
      const int ecol = eside ? HALO_W - 1 : 0, gwe = eside ? w0 + TW : w0 - 1, ghe = h0 - 1 + er;
      const unsigned egoff = (unsigned)(min(max(gwe, 0), W - 1) * (int)p.xv.sW + ec8 * 8);
      const long erow = (long)min(max(ghe, 0), H - 1) * p.xv.sH;
      const unsigned eloff = er * (HALO_W * BH_PIX) + ecol * BH_PIX + ((ec8 ^ ((ecol >> 1) & 7)) << 4);
      const unsigned emask = (ghe >= 0 && ghe < H && gwe >= 0 && gwe < W) ? 0xffffffffu : 0u;
      // edge item: requested first, converted with the first batch
      f32x4 ea = *reinterpret_cast<const f32x4*>(xb + erow + egoff), eb = *reinterpret_cast<const f32x4*>(xb + erow + egoff + 4);
      f32x4 ua = {0.f, 0.f, 0.f, 0.f}, ub = ua;
      if (GATE) {
        ua = *reinterpret_cast<const f32x4*>(p.gate_add + boffs + erow + egoff);
        ub = *reinterpret_cast<const f32x4*>(p.gate_add + boffs + erow + egoff + 4);
      }
      constexpr int RB = GATE ? 2 : 3;  // rows per batch (GATE holds two operand tensors)
#pragma unroll
      for (int r0 = 0; r0 < HALO_H; r0 += RB) {
        f32x4 v[RB][2], u[RB][2];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const long ro = (long)min(max(h0 - 1 + r0 + r, 0), H - 1) * p.xv.sH;  // scalar
          v[r][0] = *reinterpret_cast<const f32x4*>(xb + ro + goff);
          v[r][1] = *reinterpret_cast<const f32x4*>(xb + ro + goff + 4);
          if (GATE) {
            u[r][0] = *reinterpret_cast<const f32x4*>(p.gate_add + boffs + ro + goff);
            u[r][1] = *reinterpret_cast<const f32x4*>(p.gate_add + boffs + ro + goff + 4);
          }
        }
        if (r0 == 0) {
          if (AFFINE) {
            ea = ea * e4a + f4a;
            eb = eb * e4b + f4b;
          }
          if (GATE) {
            ea = sisr_mul_add4(ea, e4a, ua);
            eb = sisr_mul_add4(eb, e4b, ub);
          }
          if (tl < 96) sisr_store_split3(ldsb + eloff, ea, eb, emask);
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const int hr = r0 + r, gh = h0 - 1 + hr;
          const bool rok = gh >= 0 && gh < H;  // scalar
          f32x4 ta = v[r][0], tb = v[r][1];
          if (AFFINE) {
            ta = ta * s4a + t4a;
            tb = tb * s4b + t4b;
          }
          if (GATE) {  // u = t * gate + skip in fp32, written out once by the owning tile, then split for the MFMA
            ta = sisr_mul_add4(ta, s4a, u[r][0]);
            tb = sisr_mul_add4(tb, s4b, u[r][1]);
            if (hr >= 1 && hr <= TH && gh < H && cok) {  // interior columns are exactly the pixels this tile owns
              float* o = p.gate_out + boffs + (long)gh * p.xv.sH + goff;
              *reinterpret_cast<f32x4*>(o) = ta;
              *reinterpret_cast<f32x4*>(o + 4) = tb;
            }
          }
          sisr_store_split3(ldsb + hr * (HALO_W * BH_PIX) + loff, ta, tb, (rok && cok) ? 0xffffffffu : 0u);
        }
      }
    }
    __syncthreads();

    if (STAMP) st1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_setprio(0);
    // ---- K loop: 36 steps (tap t = s >> 2, 16-channel block kb = s & 3); per step and M-tile six MFMAs:
    // hi*hi into the main accumulator, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid into the correction accumulator
    const unsigned char* wq = wbase + ((long)q * p.cin_chunks + c) * (36 * 2048);  // scalar
    const sisr_rsrc_t rwq = sisr_rsrc(wq);  // weight fragments by {resource, lane offset, scalar offset}: sisr_common.h
#define BF_LOAD_B(s, pl) sisr_buf_load_bf16x8(rwq, boff, (unsigned)((pl) * (unsigned)wplane + (s) * 2048))
#define BF_LOAD_A(m, s, pl) \
  (*reinterpret_cast<const bf16x8*>(ldsb + (pl) * X3_PLANE + ((((s) >> 2) / 3 + (m)) * (HALO_W * BH_PIX)) + aoff[((s) >> 2) % 3][(s) & 3]))
    constexpr int RING = BD + 1;
    bf16x8 bq[RING][3];
    bf16x8 aq[2][2][3];  // [step parity][M-tile][plane]
#pragma unroll
    for (int s = 0; s < BD; ++s)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) bq[s][pl] = BF_LOAD_B(s, pl);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      aq[0][0][pl] = BF_LOAD_A(0, 0, pl);
      aq[0][1][pl] = BF_LOAD_A(1, 0, pl);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 36; ++s) {
      if (s + BD < 36) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bq[(s + BD) % RING][pl] = BF_LOAD_B(s + BD, pl);
      }
      if (s + 1 < 36) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          aq[(s + 1) & 1][0][pl] = BF_LOAD_A(0, s + 1, pl);
          aq[(s + 1) & 1][1][pl] = BF_LOAD_A(1, s + 1, pl);
        }
      }
      const bf16x8 bh = bq[s % RING][0], bm = bq[s % RING][1], bl = bq[s % RING][2];
      const bf16x8 a0h = aq[s & 1][0][0], a0m = aq[s & 1][0][1], a0l = aq[s & 1][0][2];
      const bf16x8 a1h = aq[s & 1][1][0], a1m = aq[s & 1][1][1], a1l = aq[s & 1][1][2];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, bh, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, bh, acc1, 0, 0, 0);
      cor0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, bm, cor0, 0, 0, 0);
      cor1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, bm, cor1, 0, 0, 0);
      cor0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0m, bh, cor0, 0, 0, 0);
      cor1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1m, bh, cor1, 0, 0, 0);
      cor0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, bl, cor0, 0, 0, 0);
      cor1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, bl, cor1, 0, 0, 0);
      cor0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0l, bh, cor0, 0, 0, 0);
      cor1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l, bh, cor1, 0, 0, 0);
      cor0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0m, bm, cor0, 0, 0, 0);
      cor1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1m, bm, cor1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);  // keep each step's fragment loads next to it (register budget)
    }
#undef BF_LOAD_A
#undef BF_LOAD_B
  }

  if (STAMP) st2 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_setprio(3);
  acc0 += cor0;
  acc1 += cor1;
  // ---- epilogue.  The kernel is HBM-bound, so output / mask / residual traffic must move as whole 256-B pixel
  // rows: the accumulators (column = cout on the lane, 16 pixel columns in registers) are transposed through LDS
  // and every thread then handles float4 pieces of eight pixels, all of its loads in flight at once.
  float os = p.alpha;
  if (p.out_scale) os *= p.out_scale[(long)b * Cout + q * 64 + co];
  const float lo = p.relu ? 0.f : -3.402823466e38f;
  float* ot = reinterpret_cast<float*>(ldsb);  // [TH*TW pixels][64] fp32, row stride BE_LD
  __syncthreads();                              // every wave is done reading the halo
  float grow[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    float gsum = 0.f;
    const f32x16 acc = m ? acc1 : acc0;
    const int prow = (2 * ph + m) * TW;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pc = (r & 3) + 8 * (r >> 2) + 4 * hh;
      const float v = fmaxf(acc[r], lo) * os;
      ot[(prow + pc) * BE_LD + co] = v;
      if (h0 + 2 * ph + m < H && w0 + pc < W) gsum += v;
    }
    grow[m] = gsum + __shfl_xor(gsum, 32);
  }
  if (!DOT && p.gap) {  // only offered without mask / residual (host checks): v above is the final value
    if (hh == 0) {
      const int tile = th * p.tiles_w + tw;
      const long parts = (long)p.tiles_w * p.tiles_h * 2;
      p.gap[(((long)b * parts) + tile * 2 + ph) * Cout + q * 64 + co] = grow[0] + grow[1];
    }
  }
  __syncthreads();
  {
    const int c4 = tid & 15, pr = tid >> 4;  // pixel pr + 16 i  ->  tile row i >> 1, column pr + 16 (i & 1)
    const long tile_base = (long)b * p.yv.sB + p.yv.chunk(q) + (long)w0 * p.yv.sW + c4 * 4;
    f32x4 rv[8], mv[8], dv[8];
    bool ok[8];
    long off[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = h0 + (i >> 1), col = pr + 16 * (i & 1);
      ok[i] = row < H && w0 + col < W;
      off[i] = tile_base + (long)min(row, H - 1) * p.yv.sH + (long)min(col, W - 1 - w0) * p.yv.sW;
      if (RES) rv[i] = *reinterpret_cast<const f32x4*>(p.res + off[i]);
      if (MASK) mv[i] = *reinterpret_cast<const f32x4*>(p.mask + off[i]);
      if (DOT) dv[i] = *reinterpret_cast<const f32x4*>(p.dot + off[i]);
    }
    f32x4 dsum[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};  // DOT: per 2-row strip (i < 4 / i >= 4)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f32x4 v = *reinterpret_cast<const f32x4*>(ot + (pr + 16 * i) * BE_LD + c4 * 4);
      if (MASK) {
        v[0] = mv[i][0] > 0.f ? v[0] : 0.f;
        v[1] = mv[i][1] > 0.f ? v[1] : 0.f;
        v[2] = mv[i][2] > 0.f ? v[2] : 0.f;
        v[3] = mv[i][3] > 0.f ? v[3] : 0.f;
      }
      if (RES) v += rv[i];
      if (ok[i]) {
        *reinterpret_cast<f32x4*>(p.y + off[i]) = v;
        if (DOT) dsum[i >> 2] += v * dv[i];
      }
    }
    if (DOT) {  // sum(v * dot) per strip and channel: the 16 pixel-column threads of a channel quad, in order
      __syncthreads();  // everyone is done with the transpose buffer
      float* red = ot;  // [2 strips][16][64]
      *reinterpret_cast<f32x4*>(red + (0 * 16 + pr) * 64 + c4 * 4) = dsum[0];
      *reinterpret_cast<f32x4*>(red + (1 * 16 + pr) * 64 + c4 * 4) = dsum[1];
      __syncthreads();
      if (tid < 128) {
        const int strip = tid >> 6, chn = tid & 63;
        float sacc = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) sacc += red[(strip * 16 + j) * 64 + chn];
        const int tile = th * p.tiles_w + tw;
        const long parts = (long)p.tiles_w * p.tiles_h * 2;
        p.gap[(((long)b * parts) + tile * 2 + strip) * Cout + q * 64 + chn] = sacc;
      }
    }
  }
  if (STAMP) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long st3 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && p.dot) {  // the stamp buffer rides in p.dot (unused by this instantiation)
      // same record as the fp32 kernel's (sisr_diag_conv_stamp): shader-clock phases, 100 MHz start and lifetime, placement
      const unsigned long long rt3 = __builtin_amdgcn_s_memrealtime();
      unsigned* dbg = reinterpret_cast<unsigned*>(const_cast<float*>(p.dot)) + ((long)blockIdx.x * 4 + wave) * 8;
      dbg[0] = (unsigned)(st0 & 0xffffffffu);
      dbg[1] = (unsigned)(rt0 & 0xffffffffu);
      dbg[2] = (unsigned)(st1 - st0);
      dbg[3] = (unsigned)(st2 - st1);
      dbg[4] = (unsigned)(st3 - st2);
      dbg[5] = __builtin_amdgcn_s_getreg(4 | (31 << 11));
      dbg[6] = __builtin_amdgcn_s_getreg(20 | (31 << 11));
      dbg[7] = (unsigned)(rt3 - rt0);
    }
  }
}

// ------------------------------------------------------------------ persistent bf16 kernel (64 -> 64)
// PMC showed the kernel above to be bound by memory-level parallelism: a workgroup has its halo loads in flight
// for ~15 % of its life, so a CU keeps ~22 KB outstanding -- enough for ~3 TB/s.  Here a workgroup walks tiles
// g, g + G, ... and loads tile i+1 (96 VGPRs of prefetch per thread) while tile i is in its K loop: two workgroups
// per CU keep ~100 KB in flight nearly all the time.  LDS: two buffers of 34.8 KB, each first the bf16 halo of a
// tile (26 KB) and then, once its K loop is done, that tile's fp32 transpose buffer for the float4 epilogue.
// Arithmetic, fragment order and results are those of conv3x3_c64_bf16_kernel.
#define PB_BUF (TH * TW * BE_LD * 4)

// bf16 STORAGE of maps (sisr_conv3x3_c64_bf16s): a map kept in HBM as bf16 has the same View (strides in elements) and is
// addressed through a `const float*` field of ConvParams reinterpreted as 2-byte elements.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sisr_unpack_bf16x8(u32x4 w, f32x4& a, f32x4& b) {  // exact: a bf16 is the top half of a float
  a[0] = __uint_as_float(w[0] << 16); a[1] = __uint_as_float(w[0] & 0xffff0000u);
  a[2] = __uint_as_float(w[1] << 16); a[3] = __uint_as_float(w[1] & 0xffff0000u);
  b[0] = __uint_as_float(w[2] << 16); b[1] = __uint_as_float(w[2] & 0xffff0000u);
  b[2] = __uint_as_float(w[3] << 16); b[3] = __uint_as_float(w[3] & 0xffff0000u);
}
__device__ __forceinline__ f32x4 sisr_load_bf16x4(const float* base, long elem) {  // four consecutive bf16 elements -> fp32
  const u32x2 w = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(base) + elem);
  f32x4 r;
  r[0] = __uint_as_float(w[0] << 16); r[1] = __uint_as_float(w[0] & 0xffff0000u);
  r[2] = __uint_as_float(w[1] << 16); r[3] = __uint_as_float(w[1] & 0xffff0000u);
  return r;
}
__device__ __forceinline__ void sisr_store_bf16x4(float* base, long elem, f32x4 v) {  // RNE
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  bf16x4_t r;
  r[0] = (__bf16)v[0]; r[1] = (__bf16)v[1]; r[2] = (__bf16)v[2]; r[3] = (__bf16)v[3];
  *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(base) + elem) = __builtin_bit_cast(u32x2, r);
}

// ... and the same through a buffer resource (scalar base, 32-bit lane offset, scalar offset: sisr_common.h) -- beside an
// MFMA stream a 64-bit lane address costs its SIMD tens of cycles of issue per access
__device__ __forceinline__ f32x4 sisr_buf_load_bf16x4(sisr_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
  const u32x2 w = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff_bytes, (int)soff_bytes, 0));
  f32x4 o;
  o[0] = __uint_as_float(w[0] << 16); o[1] = __uint_as_float(w[0] & 0xffff0000u);
  o[2] = __uint_as_float(w[1] << 16); o[3] = __uint_as_float(w[1] & 0xffff0000u);
  return o;
}
__device__ __forceinline__ void sisr_buf_store_bf16x4(f32x4 v, sisr_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  bf16x4_t t;
  t[0] = (__bf16)v[0]; t[1] = (__bf16)v[1]; t[2] = (__bf16)v[2]; t[3] = (__bf16)v[3];
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, t), r, (int)voff_bytes, (int)soff_bytes, 0);
}

// IN16: x, gate_add and gate_out are bf16 maps (the halo goes to LDS without conversion unless a prologue touches it);
// OUT16: y is a bf16 map (rounded once, in the epilogue's store); AUX16: mask and dot are bf16 maps; RES16: res is.
// STAMP (diagnostic library only, tools/bf16s_timeline.py): every wave sums the shader cycles it spends in the K loop, in the
// epilogue and in committing the next halo, and writes them with its lifetime to the record p.dot points to.
template <bool AFFINE, bool MASK, bool RES, bool GATE, bool DOT, bool IN16 = false, bool OUT16 = false, bool AUX16 = false,
          bool RES16 = false, bool STAMP = false>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_bf16_persist_kernel(ConvParams p, int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = gridDim.x;
  int g;
  {
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const unsigned qn = nb >> 3, rn = nb & 7;
    g = (int)((xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx);
  }
  const int ph = __builtin_amdgcn_readfirstlane(wave >> 1), ch = __builtin_amdgcn_readfirstlane(wave & 1);
  const int n = lane & 31, hh = lane >> 5;
  const int H = p.H, W = p.W;
  const int co = ch * 32 + n;
  const float bv = p.bias ? p.bias[co * p.bias_n] : 0.f;
  unsigned aoff[3][4];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
      aoff[kw][kb] = ((2 * ph) * HALO_W + n + kw) * BH_PIX + (((2 * kb + hh) ^ (((n + kw) >> 1) & 7)) << 4);
  const unsigned boff = (hh * 64 + co) * 16;
  const unsigned char* wq0 = reinterpret_cast<const unsigned char*>(p.w);
  const int c8 = tid & 7, pcol = tid >> 3;
  const int tiles_per_img = p.tiles_w * p.tiles_h;

  // ---- halo staging, split into "issue the loads" and "convert + write LDS".  Thread (c8, pcol) owns channels
  // 8 c8 .. +7 of halo column pcol + 1 in all six rows; the two edge columns (0 and 33: 6 rows x 2 x 8 chunks = 96
  // items) go one per thread to tid % 96 (threads 96.. load a duplicate and do not write), so every load is
  // unconditional and a thread holds 14 float4 pairs per tile.
  struct Halo {  // IN16: one 16-B piece (eight bf16) per item, held in in[r][0] / ed[0] as raw bits
    f32x4 in[HALO_H][IN16 ? 1 : 2];
    f32x4 ed[IN16 ? 1 : 2];
  };
  Halo v;
  const int eidx = tid % 96, er = eidx >> 4, eside = (eidx >> 3) & 1, ec8 = eidx & 7;
  auto decode = [&](int tile, int& b, int& h0, int& w0) {
    b = tile / tiles_per_img;
    const int r = tile - b * tiles_per_img;
    const int th = r / p.tiles_w;
    h0 = th * TH;
    w0 = (r - th * p.tiles_w) * TW;
  };
  // Every map access below is {resource on a wave-uniform base, 32-bit lane byte offset, scalar byte offset}: beside the
  // other resident workgroup's MFMA stream a vector-memory instruction with a 64-bit lane address costs the SIMD tens of
  // issue cycles, and the 64-bit arithmetic that builds it more (sisr_common.h "buffer addressing").
  constexpr unsigned EI = IN16 ? 2u : 4u;  // bytes per element of x / gate_add / gate_out
  auto issue = [&](const float* src, int tile, Halo& dst) {
    int b, h0, w0;
    decode(tile, b, h0, w0);
    const sisr_rsrc_t rx = sisr_rsrc(reinterpret_cast<const unsigned char*>(src) + (long)b * p.xv.sB * EI);
    const unsigned vin = (unsigned)(min(w0 + pcol, W - 1) * (int)p.xv.sW + c8 * 8) * EI;
    const int gwe = min(max(eside ? w0 + TW : w0 - 1, 0), W - 1);
    const unsigned ved = (unsigned)(min(max(h0 - 1 + er, 0), H - 1) * (int)p.xv.sH + gwe * (int)p.xv.sW + ec8 * 8) * EI;
#pragma unroll
    for (int r = 0; r < HALO_H; ++r) {
      const unsigned ro = (unsigned)(min(max(h0 - 1 + r, 0), H - 1) * (int)p.xv.sH) * EI;  // scalar
      dst.in[r][0] = sisr_buf_load4(rx, vin, ro);
      if (!IN16) dst.in[r][IN16 ? 0 : 1] = sisr_buf_load4(rx, vin + 16u, ro);
    }
    dst.ed[0] = sisr_buf_load4(rx, ved, 0u);
    if (!IN16) dst.ed[IN16 ? 0 : 1] = sisr_buf_load4(rx, ved + 16u, 0u);
  };
  auto commit = [&](unsigned char* buf, int tile) {
    int b, h0, w0;
    decode(tile, b, h0, w0);
    f32x4 s4a = {1.f, 1.f, 1.f, 1.f}, s4b = s4a, t4a = {0.f, 0.f, 0.f, 0.f}, t4b = t4a, e4a = s4a, e4b = s4a,
          f4a = t4a, f4b = t4a;
    if (AFFINE || GATE) {
      const float* sp = p.in_scale + (long)b * 64;
      s4a = *reinterpret_cast<const f32x4*>(sp + c8 * 8);
      s4b = *reinterpret_cast<const f32x4*>(sp + c8 * 8 + 4);
      e4a = *reinterpret_cast<const f32x4*>(sp + ec8 * 8);
      e4b = *reinterpret_cast<const f32x4*>(sp + ec8 * 8 + 4);
      if (AFFINE && p.in_shift) {
        const float* tp = p.in_shift + (long)b * 64;
        t4a = *reinterpret_cast<const f32x4*>(tp + c8 * 8);
        t4b = *reinterpret_cast<const f32x4*>(tp + c8 * 8 + 4);
        f4a = *reinterpret_cast<const f32x4*>(tp + ec8 * 8);
        f4b = *reinterpret_cast<const f32x4*>(tp + ec8 * 8 + 4);
      }
    }
    Halo u;
    if (GATE) issue(p.gate_add, tile, u);
    const int col = pcol + 1, gw = w0 + pcol;
    const bool cok = gw < W;
    const sisr_rsrc_t rgo = sisr_rsrc(GATE ? reinterpret_cast<unsigned char*>(p.gate_out) + (long)b * p.xv.sB * EI
                                           : reinterpret_cast<unsigned char*>(p.y));
    const unsigned vgo = (unsigned)(min(gw, W - 1) * (int)p.xv.sW + c8 * 8) * EI;
    const unsigned lo_in = col * BH_PIX + ((c8 ^ ((col >> 1) & 7)) << 4);
#pragma unroll
    for (int r = 0; r < HALO_H; ++r) {
      const int gh = h0 - 1 + r;
      const bool rok = gh >= 0 && gh < H;
      u32x4 pk;
      if (IN16 && !AFFINE && !GATE) {
        pk = __builtin_bit_cast(u32x4, v.in[r][0]);  // stored bf16 -> LDS as it is
      } else {
        f32x4 ta, tb;
        if (IN16) sisr_unpack_bf16x8(__builtin_bit_cast(u32x4, v.in[r][0]), ta, tb);
        else { ta = v.in[r][0]; tb = v.in[r][IN16 ? 0 : 1]; }
        if (AFFINE) {
          ta = ta * s4a + t4a;
          tb = tb * s4b + t4b;
        }
        if (GATE) {
          f32x4 ua, ub;
          if (IN16) sisr_unpack_bf16x8(__builtin_bit_cast(u32x4, u.in[r][0]), ua, ub);
          else { ua = u.in[r][0]; ub = u.in[r][IN16 ? 0 : 1]; }
          ta = sisr_mul_add4(ta, s4a, ua);
          tb = sisr_mul_add4(tb, s4b, ub);
          if (!IN16 && r >= 1 && r <= TH && gh < H && cok) {  // interior columns are exactly the pixels this tile owns
            // (row offset folded into the lane offset, scalar offset 0: see the epilogue's stores)
            const unsigned ro = (unsigned)(min(gh, H - 1) * (int)p.xv.sH) * EI;
            sisr_buf_store4(ta, rgo, vgo + ro, 0u);
            sisr_buf_store4(tb, rgo, vgo + ro + 16u, 0u);
          }
        }
        pk = sisr_pack_bf16x8(ta, tb);
        if (IN16 && GATE && r >= 1 && r <= TH && gh < H && cok)  // the gated skip is stored as the bf16 the MFMA reads
          sisr_buf_store4(__builtin_bit_cast(f32x4, pk), rgo, vgo + (unsigned)(min(gh, H - 1) * (int)p.xv.sH) * EI, 0u);
      }
      const unsigned m = (rok && cok) ? 0xffffffffu : 0u;
      pk &= (u32x4){m, m, m, m};
      *reinterpret_cast<u32x4*>(buf + r * (HALO_W * BH_PIX) + lo_in) = pk;
    }
    {
      const int ecol = eside ? HALO_W - 1 : 0, gwe = eside ? w0 + TW : w0 - 1, ghe = h0 - 1 + er;
      u32x4 pk;
      if (IN16 && !AFFINE && !GATE) {
        pk = __builtin_bit_cast(u32x4, v.ed[0]);
      } else {
        f32x4 ta, tb;
        if (IN16) sisr_unpack_bf16x8(__builtin_bit_cast(u32x4, v.ed[0]), ta, tb);
        else { ta = v.ed[0]; tb = v.ed[IN16 ? 0 : 1]; }
        if (AFFINE) {
          ta = ta * e4a + f4a;
          tb = tb * e4b + f4b;
        }
        if (GATE) {
          f32x4 ua, ub;
          if (IN16) sisr_unpack_bf16x8(__builtin_bit_cast(u32x4, u.ed[0]), ua, ub);
          else { ua = u.ed[0]; ub = u.ed[IN16 ? 0 : 1]; }
          ta = sisr_mul_add4(ta, e4a, ua);
          tb = sisr_mul_add4(tb, e4b, ub);
        }
        pk = sisr_pack_bf16x8(ta, tb);
      }
      const unsigned m = (ghe >= 0 && ghe < H && gwe >= 0 && gwe < W) ? 0xffffffffu : 0u;
      pk &= (u32x4){m, m, m, m};
      if (tid < 96)
        *reinterpret_cast<u32x4*>(buf + er * (HALO_W * BH_PIX) + ecol * BH_PIX + ((ec8 ^ ((ecol >> 1) & 7)) << 4)) = pk;
    }
  };

  int tile = g;
  if (tile >= total_tiles) return;
  unsigned long long st_life = 0, st_a = 0, st_k = 0, st_e = 0, st_c = 0;
  unsigned st_tiles = 0;
  if (STAMP) st_life = __builtin_amdgcn_s_memtime();
  issue(p.x, tile, v);
  commit(ldsb, tile);
  __syncthreads();
  for (int it = 0;; ++it) {
    if (STAMP) st_a = __builtin_amdgcn_s_memtime();
    unsigned char* cur = ldsb + (it & 1) * PB_BUF;
    unsigned char* nxt = ldsb + ((it + 1) & 1) * PB_BUF;
    const int next = tile + G;
    const bool has_next = next < total_tiles;  // uniform
    if (has_next) issue(p.x, next, v);         // in flight across the K loop and the epilogue of this tile
    int b, h0, w0;
    decode(tile, b, h0, w0);

    // ---- K loop (36 steps x 2 MFMAs), B fragments eight steps ahead
    const sisr_rsrc_t rwq = sisr_rsrc(wq0);  // weight fragments by {resource, lane offset, scalar offset}: no address VGPRs
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = bv;
#define PB_LOAD_B(s) sisr_buf_load_bf16x8(rwq, boff, (unsigned)((s) * 2048))
#define PB_LOAD_A(m, s) \
  (*reinterpret_cast<const bf16x8*>(cur + ((((s) >> 2) / 3 + (m)) * (HALO_W * BH_PIX)) + aoff[((s) >> 2) % 3][(s) & 3]))
    {
      bf16x8 bq[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) bq[s] = PB_LOAD_B(s);
#pragma unroll
      for (int s = 0; s < 36; ++s) {
        const bf16x8 bb = bq[s & 7];
        if (s + 8 < 36) bq[s & 7] = PB_LOAD_B(s + 8);
        const bf16x8 a0 = PB_LOAD_A(0, s);
        const bf16x8 a1 = PB_LOAD_A(1, s);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bb, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bb, acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);  // keep the fragment loads next to their step: 96 prefetch VGPRs are live
      }
    }
#undef PB_LOAD_A
#undef PB_LOAD_B
    if (STAMP) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      st_k += t - st_a;
      st_a = t;
      ++st_tiles;
    }

    // ---- epilogue through `cur` (its halo is dead once every wave has left the K loop)
    float os = p.alpha;
    if (p.out_scale) os *= p.out_scale[(long)b * 64 + co];
    const float lo = p.relu ? 0.f : -3.402823466e38f;
    float* ot = reinterpret_cast<float*>(cur);
    const int c4 = tid & 15, pr = tid >> 4;
    const long tile_base = (long)b * p.yv.sB + (long)w0 * p.yv.sW;  // wave-uniform, elements
    constexpr unsigned EY = OUT16 ? 2u : 4u, EA = AUX16 ? 2u : 4u, ER = RES16 ? 2u : 4u;
    const sisr_rsrc_t ry = sisr_rsrc(reinterpret_cast<unsigned char*>(p.y) + tile_base * EY);
    const sisr_rsrc_t rrs = sisr_rsrc(RES ? reinterpret_cast<const unsigned char*>(p.res) + tile_base * ER : (const unsigned char*)p.y);
    const sisr_rsrc_t rmk = sisr_rsrc(MASK ? reinterpret_cast<const unsigned char*>(p.mask) + tile_base * EA : (const unsigned char*)p.y);
    const sisr_rsrc_t rdt = sisr_rsrc(DOT ? reinterpret_cast<const unsigned char*>(p.dot) + tile_base * EA : (const unsigned char*)p.y);
    __syncthreads();
    // (beside the other workgroup's MFMA stream every instruction here costs issue time: ReLU, scaling and the pooled sums sit
    // behind wave-uniform branches, the transposing writes use one base register and immediate offsets)
    const bool do_relu = p.relu != 0, scaled = p.out_scale != nullptr || p.alpha != 1.0f;  // uniform
    const bool want_sum = !DOT && p.gap != nullptr, full_tile = (h0 + TH <= H) && (w0 + TW <= W);
    (void)lo;
    float grow[2] = {0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      f32x16 acc = m ? acc1 : acc0;
      if (do_relu) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
      }
      if (scaled) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] *= os;
      }
      float* wb = ot + ((2 * ph + m) * TW + 4 * hh) * BE_LD + co;
#pragma unroll
      for (int r = 0; r < 16; ++r) wb[((r & 3) + 8 * (r >> 2)) * BE_LD] = acc[r];
      if (want_sum) {
        float gsum = 0.f;
        if (full_tile) {
#pragma unroll
          for (int r = 0; r < 16; ++r) gsum += acc[r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (h0 + 2 * ph + m < H && w0 + (r & 3) + 8 * (r >> 2) + 4 * hh < W) gsum += acc[r];
        }
        grow[m] = gsum + __shfl_xor(gsum, 32);
      }
    }
    const int tile_in_img = tile - b * tiles_per_img;
    const long parts = (long)tiles_per_img * 2;
    if (!DOT && p.gap && hh == 0) p.gap[(((long)b * parts) + tile_in_img * 2 + ph) * 64 + co] = grow[0] + grow[1];
    // Whole tiles (every tile of a 128 x 128 map) take the straight-line form: no clamps, no per-store predicates -- the
    // divergent branch around each store of the general form costs six scalar instructions.
    f32x4 rv[8], mv[8], dv[8];
    bool ok[8];
    unsigned vo[8], so[8];  // element offsets inside the tile's resource: lane part, scalar (row) part
    if (full_tile) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        ok[i] = true;
        vo[i] = (unsigned)((pr + 16 * (i & 1)) * (int)p.yv.sW + c4 * 4);
        so[i] = (unsigned)((h0 + (i >> 1)) * (int)p.yv.sH);  // scalar
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = h0 + (i >> 1), col = pr + 16 * (i & 1);
        ok[i] = row < H && w0 + col < W;
        vo[i] = (unsigned)(min(col, W - 1 - w0) * (int)p.yv.sW + c4 * 4);
        so[i] = (unsigned)(min(row, H - 1) * (int)p.yv.sH);  // scalar
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // epilogue operands (the accumulators are dead by now)
      if (RES) rv[i] = RES16 ? sisr_buf_load_bf16x4(rrs, vo[i] * 2u, so[i] * 2u) : sisr_buf_load4(rrs, vo[i] * 4u, so[i] * 4u);
      if (MASK) mv[i] = AUX16 ? sisr_buf_load_bf16x4(rmk, vo[i] * 2u, so[i] * 2u) : sisr_buf_load4(rmk, vo[i] * 4u, so[i] * 4u);
      if (DOT) dv[i] = AUX16 ? sisr_buf_load_bf16x4(rdt, vo[i] * 2u, so[i] * 2u) : sisr_buf_load4(rdt, vo[i] * 4u, so[i] * 4u);
    }
    __syncthreads();
    f32x4 dsum[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // Stores: a 128-bit buffer store carries its whole offset in the lane operand, scalar offset 0.  Found the hard way: a
    // 128-bit buffer store with an SGPR scalar offset whose data registers the next VALU instruction overwrites loses that
    // race on gfx950 (the compiler's hazard table covers the immediate-offset form only): 4 channels of every other row came
    // out wrong.  The 64-bit (bf16) stores are outside that hazard and keep the row in the scalar offset.
    auto put = [&](int i, f32x4 val) {
      if (OUT16) sisr_buf_store_bf16x4(val, ry, vo[i] * 2u, so[i] * 2u);
      else sisr_buf_store4(val, ry, (vo[i] + so[i]) * 4u, 0u);
    };
    auto finish = [&](int i) {
      f32x4 val = *reinterpret_cast<const f32x4*>(ot + (pr + 16 * i) * BE_LD + c4 * 4);
      if (MASK) {
        val[0] = mv[i][0] > 0.f ? val[0] : 0.f;
        val[1] = mv[i][1] > 0.f ? val[1] : 0.f;
        val[2] = mv[i][2] > 0.f ? val[2] : 0.f;
        val[3] = mv[i][3] > 0.f ? val[3] : 0.f;
      }
      if (RES) val += rv[i];
      return val;
    };
    if (full_tile) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x4 val = finish(i);
        put(i, val);
        if (DOT) dsum[i >> 2] += val * dv[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x4 val = finish(i);
        if (ok[i]) {
          put(i, val);
          if (DOT) dsum[i >> 2] += val * dv[i];
        }
      }
    }
    if (DOT) {
      __syncthreads();
      float* red = ot;
      *reinterpret_cast<f32x4*>(red + (0 * 16 + pr) * 64 + c4 * 4) = dsum[0];
      *reinterpret_cast<f32x4*>(red + (1 * 16 + pr) * 64 + c4 * 4) = dsum[1];
      __syncthreads();
      if (tid < 128) {
        const int strip = tid >> 6, chn = tid & 63;
        float sacc = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) sacc += red[(strip * 16 + j) * 64 + chn];
        p.gap[(((long)b * parts) + tile_in_img * 2 + strip) * 64 + chn] = sacc;
      }
    }
    if (STAMP) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      st_e += t - st_a;
      st_a = t;
    }
    if (!has_next) break;
    commit(nxt, next);  // the prefetched halo of the next tile (nxt's previous reader finished one iteration ago)
    __syncthreads();
    if (STAMP) st_c += __builtin_amdgcn_s_memtime() - st_a;
    tile = next;
  }
  if (STAMP && lane == 0) {
    unsigned* dbg = reinterpret_cast<unsigned*>(const_cast<float*>(p.dot)) + ((long)blockIdx.x * 4 + wave) * 8;
    dbg[0] = (unsigned)(__builtin_amdgcn_s_memtime() - st_life);
    dbg[1] = (unsigned)st_k;
    dbg[2] = (unsigned)st_e;
    dbg[3] = (unsigned)st_c;
    dbg[4] = st_tiles;
    dbg[5] = __builtin_amdgcn_s_getreg(4 | (31 << 11));   // HW_REG_HW_ID
    dbg[6] = __builtin_amdgcn_s_getreg(20 | (31 << 11));  // HW_REG_XCC_ID
    dbg[7] = (unsigned)(st_life & 0xffffffffu);
  }
}

// bf16 packings of one OIHW weight (forward and input-gradient orders) in one launch:
// packed[q][c][s][h][n][j] = bf16(w[o][i][t]),  t = s >> 2,  i_local = 16*(s & 3) + 8h + j,  o_local = n.
__global__ void pack_conv3x3_bf16_both_kernel(const float* __restrict__ w, __bf16* __restrict__ pf,
                                              __bf16* __restrict__ pd, int cout, int cin, int r) {
  const long total = (long)cout * cin * 9;
  const int rr = r * r;
  const int oc = cout >> 6, ic = cin >> 6;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long t_ = idx;
    const int j = t_ & 7;
    t_ >>= 3;
    const int n = t_ & 63;
    t_ >>= 6;
    const int h = t_ & 1;
    t_ >>= 1;
    const int s = t_ % 36;
    t_ /= 36;
    const int t = s >> 2;
    const int k = 16 * (s & 3) + 8 * h + j;
    {
      const int c = t_ % ic, q = t_ / ic;
      const long o = r > 1 ? (long)n * rr + q : (long)q * 64 + n;
      const long i = (long)c * 64 + k;
      pf[idx] = (__bf16)w[(o * cin + i) * 9 + t];
    }
    {
      const int c = t_ % oc, q = t_ / oc;
      const long i = (long)q * 64 + n;
      const long o = r > 1 ? (long)k * rr + c : (long)c * 64 + k;
      pd[idx] = (__bf16)w[(o * cin + i) * 9 + (8 - t)];
    }
  }
}

__device__ __forceinline__ void sisr_store_w3(__bf16* dst, long idx, long total, float x) {
  const __bf16 xh = (__bf16)x;
  const float r1 = x - (float)xh;
  const __bf16 xm = (__bf16)r1;
  dst[idx] = xh;
  dst[idx + total] = xm;
  dst[idx + 2 * total] = (__bf16)(r1 - (float)xm);
}

// bf16x3 packings: the bf16 element order of pack_conv3x3_bf16_both_kernel, three planes (hi, mid, lo) `total`
// elements apart, for the forward and the input-gradient order.
__global__ void pack_conv3x3_x3_both_kernel(const float* __restrict__ w, __bf16* __restrict__ pf, __bf16* __restrict__ pd,
                                            int cout, int cin, int r) {
  const long total = (long)cout * cin * 9;
  const int rr = r * r;
  const int oc = cout >> 6, ic = cin >> 6;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long t_ = idx;
    const int j = t_ & 7;
    t_ >>= 3;
    const int n = t_ & 63;
    t_ >>= 6;
    const int h = t_ & 1;
    t_ >>= 1;
    const int s = t_ % 36;
    t_ /= 36;
    const int t = s >> 2;
    const int k = 16 * (s & 3) + 8 * h + j;
    {
      const int c = t_ % ic, q = t_ / ic;
      const long o = r > 1 ? (long)n * rr + q : (long)q * 64 + n;
      const long i = (long)c * 64 + k;
      sisr_store_w3(pf, idx, total, w[(o * cin + i) * 9 + t]);
    }
    {
      const int c = t_ % oc, q = t_ / oc;
      const long i = (long)q * 64 + n;
      const long o = r > 1 ? (long)k * rr + c : (long)c * 64 + k;
      sisr_store_w3(pd, idx, total, w[(o * cin + i) * 9 + (8 - t)]);
    }
  }
}

// ------------------------------------------------------------------ weight packing
// packed[q][c][t][j][h][co][e] = w[o*so + i*si + t'],  o = co*on + q*oq,  i = (8j+4h+e)*in_ + c*iq,
// t' = flip ? 8-t : t.  One thread per packed element.
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, float* __restrict__ packed, int cout_chunks,
                                    int cin_chunks, long so, long si, int flip, int on, int oq, int in_, int iq) {
  const long total = (long)cout_chunks * cin_chunks * 9 * 4096;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long r = idx;
    const int e = r & 3;
    r >>= 2;
    const int co = r & 63;
    r >>= 6;
    const int h = r & 1;
    r >>= 1;
    const int j = r & 7;
    r >>= 3;
    const int t = r % 9;
    r /= 9;
    const int c = r % cin_chunks;
    const int q = r / cin_chunks;
    const long o = (long)co * on + (long)q * oq;
    const long i = (long)(8 * j + 4 * h + e) * in_ + (long)c * iq;
    packed[idx] = w[o * so + i * si + (flip ? 8 - t : t)];
  }
}

// Both packings of one OIHW weight in a single launch: the forward B-fragment order (q = output chunk)
// and the input-gradient order (roles swapped, taps flipped).  r > 1: the conv feeds PixelShuffle(r), so
// its output channels are regrouped as chunk q = i*r+j <- {c*r*r + q}.
__global__ void pack_conv3x3_both_kernel(const float* __restrict__ w, float* __restrict__ pf, float* __restrict__ pd,
                                         int cout, int cin, int r) {
  const long total = (long)cout * cin * 9;
  const int rr = r * r;
  const int oc = cout >> 6, ic = cin >> 6;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long t_ = idx;
    const int e = t_ & 3;
    t_ >>= 2;
    const int n = t_ & 63;
    t_ >>= 6;
    const int h = t_ & 1;
    t_ >>= 1;
    const int j = t_ & 7;
    t_ >>= 3;
    const int t = t_ % 9;
    t_ /= 9;
    const int k8 = 8 * j + 4 * h + e;
    {  // forward: [q < oc][c < ic]; o = (n, q) through the shuffle map, i = c*64 + k8
      const int c = t_ % ic, q = t_ / ic;
      const long o = r > 1 ? (long)n * rr + q : (long)q * 64 + n;
      const long i = (long)c * 64 + k8;
      pf[idx] = w[(o * cin + i) * 9 + t];
    }
    {  // dgrad: [q < ic][c < oc]; output = original input channel q*64+n, input = original output (k8, c)
      const int c = t_ % oc, q = t_ / oc;
      const long i = (long)q * 64 + n;
      const long o = r > 1 ? (long)k8 * rr + c : (long)c * 64 + k8;
      pd[idx] = w[(o * cin + i) * 9 + (8 - t)];
    }
  }
}

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

extern "C" int sisr_pack_conv3x3(const float* w, float* packed, int cout, int cin, int64_t so, int64_t si,
                                 int flip_taps, int out_perm_n, int out_perm_q, int in_perm_n, int in_perm_q,
                                 void* stream) {
  if (!w || !packed || cout <= 0 || cin <= 0 || (cout & 63) || (cin & 63)) return SISR_ERR_ARG;
  const long total = (long)cout * cin * 9;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_conv3x3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, cout / 64,
                     cin / 64, (long)so, (long)si, flip_taps, out_perm_n, out_perm_q, in_perm_n, in_perm_q);
  return sisr_check_launch();
}

extern "C" int sisr_pack_conv3x3_both(const float* w, float* packed_fwd, float* packed_dgrad, int cout, int cin,
                                      int shuffle_r, void* stream) {
  if (!w || !packed_fwd || !packed_dgrad || cout <= 0 || cin <= 0 || (cout & 63) || (cin & 63) || shuffle_r < 1)
    return SISR_ERR_ARG;
  if (shuffle_r > 1 && cout != 64 * shuffle_r * shuffle_r) return SISR_ERR_UNSUPPORTED;
  const long total = (long)cout * cin * 9;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_conv3x3_both_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed_fwd,
                     packed_dgrad, cout, cin, shuffle_r);
  return sisr_check_launch();
}

// Every conv weight of a network packed by ONE launch (a training step repacks all of them after the optimiser
// moved them: 413 launches of ~5 us each for RCAN).  jobs[] lives in device memory (built once per network by the
// caller): job j owns blocks [first_block, first_block of j+1); a block packs 256 consecutive packed elements in
// both orders, element maps as in pack_conv3x3_both_kernel / pack_conv3x3_bf16_both_kernel.
struct PackJob {
  const float* w;
  void* pf;
  void* pd;
  int cout, cin, r, first_block;
  int co_real, ci_real;  // > 0: w is (co_real, ci_real, 3, 3) and is packed as its zero-padded (cout, cin, 3, 3) twin (SPARNet)
};

template <int MODE>  // 0 fp32, 1 bf16, 2 bf16x3 (three planes `total` elements apart)
__global__ __launch_bounds__(256) void pack_conv3x3_many_kernel(const PackJob* __restrict__ jobs, int n_jobs) {
  __shared__ int job_s;
  if (threadIdx.x == 0) {  // largest j with first_block <= blockIdx.x
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    job_s = lo;
  }
  __syncthreads();
  const PackJob jb = jobs[job_s];
  const float* __restrict__ wsrc = jb.w;
  const int cout = jb.cout, cin = jb.cin, r = jb.r;
  const int cor = jb.co_real > 0 ? jb.co_real : cout, cir = jb.ci_real > 0 ? jb.ci_real : cin;
  // element (o, i, tap) of the (zero-padded) weight; the index expression the packings below were written with
  struct Src {
    const float* __restrict__ p;
    int cor, cir;
    __device__ float at(long o, long i, int t) const { return (o < cor && i < cir) ? p[(o * cir + i) * 9 + t] : 0.f; }
  };
  const Src w{wsrc, cor, cir};
  const long total = (long)cout * cin * 9;
  const long idx = (long)(blockIdx.x - jb.first_block) * 256 + threadIdx.x;
  if (idx >= total) return;
  const int rr = r * r;
  const int oc = cout >> 6, ic = cin >> 6;
  if (MODE == 0) {
    float* pf = static_cast<float*>(jb.pf);
    float* pd = static_cast<float*>(jb.pd);
    long t_ = idx;
    const int e = t_ & 3;
    t_ >>= 2;
    const int n = t_ & 63;
    t_ >>= 6;
    const int h = t_ & 1;
    t_ >>= 1;
    const int j = t_ & 7;
    t_ >>= 3;
    const int t = t_ % 9;
    t_ /= 9;
    const int k8 = 8 * j + 4 * h + e;
    {
      const int c = t_ % ic, q = t_ / ic;
      const long o = r > 1 ? (long)n * rr + q : (long)q * 64 + n;
      const long i = (long)c * 64 + k8;
      pf[idx] = w.at(o, i, t);
    }
    {
      const int c = t_ % oc, q = t_ / oc;
      const long i = (long)q * 64 + n;
      const long o = r > 1 ? (long)k8 * rr + c : (long)c * 64 + k8;
      pd[idx] = w.at(o, i, 8 - t);
    }
  } else {
    __bf16* pf = static_cast<__bf16*>(jb.pf);
    __bf16* pd = static_cast<__bf16*>(jb.pd);
    long t_ = idx;
    const int j = t_ & 7;
    t_ >>= 3;
    const int n = t_ & 63;
    t_ >>= 6;
    const int h = t_ & 1;
    t_ >>= 1;
    const int s = t_ % 36;
    t_ /= 36;
    const int t = s >> 2;
    const int k = 16 * (s & 3) + 8 * h + j;
    {
      const int c = t_ % ic, q = t_ / ic;
      const long o = r > 1 ? (long)n * rr + q : (long)q * 64 + n;
      const long i = (long)c * 64 + k;
      if (MODE == 2) sisr_store_w3(pf, idx, total, w.at(o, i, t));
      else pf[idx] = (__bf16)w.at(o, i, t);
    }
    {
      const int c = t_ % oc, q = t_ / oc;
      const long i = (long)q * 64 + n;
      const long o = r > 1 ? (long)k * rr + c : (long)c * 64 + k;
      if (MODE == 2) sisr_store_w3(pd, idx, total, w.at(o, i, 8 - t));
      else pd[idx] = (__bf16)w.at(o, i, 8 - t);
    }
  }
}

extern "C" size_t sisr_pack_job_bytes() { return sizeof(PackJob); }

extern "C" int sisr_pack_conv3x3_many(const void* jobs_device, int n_jobs, int total_blocks, int bf16, void* stream) {
  if (!jobs_device || n_jobs <= 0 || total_blocks <= 0) return SISR_ERR_ARG;
  if (bf16 == 2)
    hipLaunchKernelGGL(pack_conv3x3_many_kernel<2>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const PackJob*>(jobs_device), n_jobs);
  else if (bf16 == 1)
    hipLaunchKernelGGL(pack_conv3x3_many_kernel<1>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const PackJob*>(jobs_device), n_jobs);
  else if (bf16 == 0)
    hipLaunchKernelGGL(pack_conv3x3_many_kernel<0>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const PackJob*>(jobs_device), n_jobs);
  else
    return SISR_ERR_ARG;
  return sisr_check_launch();
}

// Kernel selection (process-wide, read-only during launches): 4 = issue-lean kernel (2-row tiles on small grids,
// 4-row tiles otherwise) with the general kernel as fallback (default); 5 / 6 = the same with the 4-row / 2-row
// tile forced; 2 = general kernel only; 13 / 16 = diagnostic builds of the general kernel (see above).
// 4-row-tile workgroups below which the 2-row kernel is used.  Round 2 (weights fetched with 64-bit lane addresses: the
// 2-row tile's twice-as-many weight loads per MFMA cost matrix-pipe time): 1.6x faster at 128 workgroups, a tie or worse from
// 256 up, so 200.  With buffer-addressed weight loads (free beside the MFMAs) the 2-row tile wins at every size measured
// (64 -> 64, 128 x 128 maps: 8 samples 85 vs 101 us, 16: 158 vs 170, 32: 296 vs 307): four resident workgroups per CU
// instead of three and a finer last round.  SISR_CONV_TILE_ROWS=4 (read per call, no state kept) restores the old rule.
#define SMALL_GRID_BLOCKS (sisr_small_grid_blocks())
// the persistent form (conv3x3_c64_p4_kernel) is used from two tiles per workgroup on (select 7 forces it, 5 / 6 the
// per-tile kernels; SISR_CONV_PERSISTENT=0 switches it off per call, for A/B measurements)
static inline bool sisr_use_persistent(int variant, long nblk) {
  if (variant == 7) return true;
  if (variant != 4) return false;
  const char* e = getenv("SISR_CONV_PERSISTENT");
  if (e && e[0] == '0') return false;
  return nblk >= 1024;
}
static inline long sisr_small_grid_blocks() {
  const char* e = getenv("SISR_CONV_TILE_ROWS");
  if (e && e[0] == '4') return 200;
  if (e && e[0] == '2') return 0x7fffffffL;
  if (const char* n = getenv("SISR_CONV_SMALL_BLOCKS")) return atol(n);
  return 0x7fffffffL;
}
// Host-side description of a channel-attention tail (include/sisr_hip.h: sisr_ca_tail).
struct sisr_ca_tail_host {
  int backward, hidden;
  float inv_hw;
  const float *w1, *b1, *w2, *b2, *mul;
  const float *s, *hid, *ca;
  float *s_out, *hid_out, *ca_out, *g_out;
  float *shift, *dmul, *dw1, *db1, *dw2, *db2;
  float* workspace;
  unsigned* counter;
  const float* head_part;  // head != 0: partial sums of the PREVIOUS launch, [B][head_parts][64]
  int head_parts, head;
};
extern "C" size_t sisr_ca_tail_bytes() { return sizeof(sisr_ca_tail_host); }

extern "C" int sisr_conv3x3_c64_gap_parts(int H, int W) { return ((H + TH - 1) / TH) * ((W + TW - 1) / TW) * 2; }

#ifdef SISR_DIAG
static unsigned* g_diag_conv_stamp = nullptr;  // diagnostic library only (the product library keeps no state)
extern "C" void sisr_diag_conv_stamp(void* buf) { g_diag_conv_stamp = static_cast<unsigned*>(buf); }
#endif

extern "C" int sisr_conv3x3_c64(const float* x, const int64_t* xview, const float* wpacked, const float* bias,
                                int bias_n, int bias_q, float* y, const int64_t* yview, const float* res,
                                const float* mask, const float* in_scale, const float* in_shift,
                                const float* out_scale, float alpha, int relu, float* gap_partial,
                                const float* gate_add, float* gate_out, const float* dot, int B, int H, int W, int cin,
                                int cout, const void* ca_tail, int select, void* stream) {
  if (!x || !wpacked || !y || !xview || !yview || B <= 0 || H <= 0 || W <= 0) return SISR_ERR_ARG;
  const sisr_ca_tail_host* tail = static_cast<const sisr_ca_tail_host*>(ca_tail);
  const sisr_ca_tail_host* head = nullptr;
  if (tail && tail->head) {  // gate head: computed by this launch's workgroups from the previous launch's partial sums
    head = tail;
    tail = nullptr;
    if (cin != 64 || cout != 64 || !head->head_part || head->head_parts <= 0 || head->hidden < 1 || head->hidden > 16 || select != 0)
      return SISR_ERR_UNSUPPORTED;
    if (head->backward ? (!mask || !in_scale || in_shift != head->shift || !head->w1 || !head->w2 || !head->hid || !head->ca ||
                          !head->shift || !head->workspace || (head->mul && !head->dmul) || gate_add || dot)
                       : (!gate_add || !gate_out || in_scale != head->g_out || !head->w1 || !head->b1 || !head->w2 || !head->b2 ||
                          !head->s_out || !head->hid_out || !head->ca_out || !head->g_out))
      return SISR_ERR_ARG;
  }
  if (tail) {  // only on the issue-lean 64 -> 64 kernels, from the partial sums this launch writes
    if (cin != 64 || cout != 64 || !gap_partial || !tail->counter || tail->hidden < 1 || tail->hidden > 16 || select == 2)
      return SISR_ERR_UNSUPPORTED;
    if (tail->backward ? (!dot || !tail->w1 || !tail->w2 || !tail->s || !tail->hid || !tail->ca || !tail->shift ||
                          !tail->dw1 || !tail->db1 || !tail->dw2 || !tail->db2 || !tail->workspace ||
                          (tail->mul && !tail->dmul))
                       : (dot || gate_add || mask || res || in_scale || out_scale || !tail->w1 || !tail->b1 || !tail->w2 ||
                          !tail->b2 || !tail->s_out || !tail->hid_out || !tail->ca_out || !tail->g_out))
      return SISR_ERR_ARG;
  }
  // select (per call; the library keeps no state): 0 / 4 = issue-lean kernel, tile height by grid size, general kernel
  // as fallback; 5 / 6 = the same with the 4-row / 2-row tile forced (A/B measurements, bit-identical results);
  // 2 = general kernel only.  Diagnostic builds (-DSISR_DIAG) add 13 / 16.
  // 8 / 9 / 10 = structurally sparse weights (SFTMD's merged convs; KSEL 1 / 2 / 3 of conv3x3_c64_v4_kernel): the caller
  // asserts that the skipped blocks of the packed weight are zero
  const int ksel = (select >= 8 && select <= 10) ? select - 7 : 0;
  const int variant = (select == 0 || ksel) ? 4 : select;
#ifdef SISR_DIAG
  if (variant != 4 && variant != 5 && variant != 6 && variant != 7 && variant != 2 && variant != 13 &&
      variant != 16)
    return SISR_ERR_ARG;
#else
  if (variant != 4 && variant != 5 && variant != 6 && variant != 7 && variant != 2) return SISR_ERR_ARG;
#endif
  if (ksel) {
    const bool shape_ok = ksel == 1 ? (cin == 64 && cout == 128) : (cin == 128 && cout == 64);
    if (!shape_ok || ca_tail || gate_add || gate_out || dot || in_scale || in_shift || out_scale || res) return SISR_ERR_UNSUPPORTED;
  }
  if ((cin & 63) || (cout & 63) || cin <= 0 || cout <= 0) return SISR_ERR_UNSUPPORTED;
  if (!sisr_aligned16(x) || !sisr_aligned16(wpacked) || !sisr_aligned16(in_scale) || !sisr_aligned16(in_shift))
    return SISR_ERR_ALIGN;
  ConvParams p;
  memset(&p, 0, sizeof(p));
#ifdef SISR_DIAG
  p.stamp = g_diag_conv_stamp;
#endif
  p.x = x;
  p.xv = view_from(xview);
  p.y = y;
  p.yv = view_from(yview);
  if ((p.xv.sB | p.xv.sH | p.xv.sW | p.xv.chi | p.xv.clo) & 3) return SISR_ERR_ALIGN;  // 16-B pixel rows
  p.res = res;
  p.mask = mask;
  p.w = wpacked;
  p.bias = bias;
  p.in_scale = in_scale;
  p.in_shift = in_shift;
  p.out_scale = out_scale;
  p.gap = gap_partial;
  p.gate_add = gate_add;
  p.gate_out = gate_out;
  p.dot = dot;
  p.alpha = alpha;
  p.bias_n = bias_n;
  p.bias_q = bias_q;
  p.B = B;
  p.H = H;
  p.W = W;
  p.cin_chunks = cin / 64;
  p.cout_chunks = cout / 64;
  // relu: 0 none, 1 ReLU, 2 LeakyReLU(0.2); + 4: `mask` carries LeakyReLU's derivative (slope 0.2 where the map is <= 0)
  if (relu < 0 || relu > 6 || (relu & 3) == 3 || ((relu & 4) && !mask)) return SISR_ERR_ARG;
  p.relu = relu & 3;
  p.mask_leaky = (relu & 4) != 0;
  const bool leaky = p.relu == 2 || p.mask_leaky;
  if (leaky && (tail || gate_add || gate_out || dot || in_scale || in_shift || (p.relu == 2 && (mask || res)) ||
                (p.mask_leaky && res)))
    return SISR_ERR_UNSUPPORTED;
  if (tail && !tail->backward) {
    p.fwd_tail.w1 = tail->w1; p.fwd_tail.b1 = tail->b1; p.fwd_tail.w2 = tail->w2; p.fwd_tail.b2 = tail->b2;
    p.fwd_tail.mul = tail->mul; p.fwd_tail.s = tail->s_out; p.fwd_tail.hid = tail->hid_out; p.fwd_tail.ca = tail->ca_out;
    p.fwd_tail.g = tail->g_out; p.fwd_tail.counter = tail->counter; p.fwd_tail.inv_hw = tail->inv_hw;
    p.fwd_tail.R = tail->hidden;
  } else if (tail) {
    p.bwd_tail.w1 = tail->w1; p.bwd_tail.w2 = tail->w2; p.bwd_tail.s = tail->s; p.bwd_tail.hid = tail->hid;
    p.bwd_tail.ca = tail->ca; p.bwd_tail.mul = tail->mul; p.bwd_tail.shift = tail->shift; p.bwd_tail.dmul = tail->dmul;
    p.bwd_tail.dz2 = tail->workspace; p.bwd_tail.dz1 = tail->workspace + 64;
    p.bwd_tail.dw1 = tail->dw1; p.bwd_tail.db1 = tail->db1; p.bwd_tail.dw2 = tail->dw2; p.bwd_tail.db2 = tail->db2;
    p.bwd_tail.counter = tail->counter; p.bwd_tail.inv_hw = tail->inv_hw; p.bwd_tail.R = tail->hidden;
  }
  if (head) {
    p.head_part = head->head_part;
    p.head_parts = head->head_parts;
    if (!head->backward) {
      p.fwd_tail.w1 = head->w1; p.fwd_tail.b1 = head->b1; p.fwd_tail.w2 = head->w2; p.fwd_tail.b2 = head->b2;
      p.fwd_tail.mul = head->mul; p.fwd_tail.s = head->s_out; p.fwd_tail.hid = head->hid_out; p.fwd_tail.ca = head->ca_out;
      p.fwd_tail.g = head->g_out; p.fwd_tail.inv_hw = head->inv_hw; p.fwd_tail.R = head->hidden;
    } else {
      p.bwd_tail.w1 = head->w1; p.bwd_tail.w2 = head->w2; p.bwd_tail.hid = head->hid; p.bwd_tail.ca = head->ca;
      p.bwd_tail.mul = head->mul; p.bwd_tail.shift = head->shift; p.bwd_tail.dmul = head->dmul;
      p.bwd_tail.dz2 = head->workspace; p.bwd_tail.dz1 = head->workspace + 64;
      p.bwd_tail.inv_hw = head->inv_hw; p.bwd_tail.R = head->hidden;
    }
  }
  p.tiles_w = (W + TW - 1) / TW;
  p.tiles_h = (H + TH - 1) / TH;
  const long nblk = (long)p.tiles_w * p.tiles_h * B;
  if (nblk > 0x7fffffffL) return SISR_ERR_ARG;
  const dim3 grid((unsigned)nblk, p.cout_chunks);
  if (gate_add || gate_out || dot) {
    // fused gated-residual chain (64 -> 64 only): GATE prologue [+ residual], or DOT epilogue [+ residual]
    const bool gate = gate_add != nullptr;
    if (gate != (gate_out != nullptr) || (gate && !in_scale) || (gate && dot) || (dot && !gap_partial) ||
        (!gate && in_scale) || in_shift || mask || out_scale || cin != 64 || cout != 64 ||
        !sisr_aligned16(gate_add) || !sisr_aligned16(gate_out))
      return SISR_ERR_UNSUPPORTED;
    if (memcmp(xview, yview, 6 * sizeof(int64_t)) != 0) return SISR_ERR_UNSUPPORTED;  // skip / dot share one layout
    hipStream_t st = (hipStream_t)stream;
    const bool rs = res != nullptr;
    if (!head && !tail && sisr_use_persistent(variant, nblk)) {
      const dim3 gp((unsigned)(nblk < 512 ? nblk : 512));
      const size_t lbp = HALO_H * HALO_W * 64 * sizeof(float);
#define P4X(RS, GT, DT) hipLaunchKernelGGL((conv3x3_c64_p4_kernel<false, false, RS, GT, DT>), gp, dim3(256), lbp, st, p, (int)nblk)
      if (gate) { if (rs) P4X(true, true, false); else P4X(false, true, false); }
      else      { if (rs) P4X(true, false, true); else P4X(false, false, true); }
#undef P4X
      return sisr_check_launch();
    }
    const bool small = variant == 6 || (variant != 5 && nblk < SMALL_GRID_BLOCKS);
    dim3 g = grid;
    size_t lb = HALO_H * HALO_W * 64 * sizeof(float);
    if (small) {
      p.tiles_h = (H + 1) / 2;
      g = dim3((unsigned)((long)p.tiles_w * p.tiles_h * B), 1);
      lb = 4 * HALO_W * 64 * sizeof(float);
    }
#define V4X(RS, MTV, GT, DT) hipLaunchKernelGGL((conv3x3_c64_v4_kernel<false, false, RS, MTV, GT, DT>), g, dim3(256), lb, st, p)
    if (head && (!gate || head->backward)) return SISR_ERR_ARG;
#define V4H(RS, MTV) hipLaunchKernelGGL((conv3x3_c64_v4_kernel<false, false, RS, MTV, true, false, false, 0, 1>), g, dim3(256), lb + SISR_HEAD_LDS, st, p)
    if (gate && head) {
      if (small) { if (rs) V4H(true, 1); else V4H(false, 1); }
      else       { if (rs) V4H(true, 2); else V4H(false, 2); }
    } else if (gate) {
      if (small) { if (rs) V4X(true, 1, true, false); else V4X(false, 1, true, false); }
      else       { if (rs) V4X(true, 2, true, false); else V4X(false, 2, true, false); }
    } else {
      if (small) { if (rs) V4X(true, 1, false, true); else V4X(false, 1, false, true); }
      else       { if (rs) V4X(true, 2, false, true); else V4X(false, 2, false, true); }
    }
#undef V4X
#undef V4H
    return sisr_check_launch();
  }
  if (variant == 4 || variant == 5 || variant == 6 || variant == 7) {
    const bool aff = in_scale != nullptr, msk = mask != nullptr, rs = res != nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (ksel) {
      // 1: plain (bias only); 2: LeakyReLU epilogue; 3: LeakyReLU' mask -- the forms SFTMD needs
      const bool form_ok = ksel == 1 ? (!msk && p.relu == 0) : (ksel == 2 ? (!msk && p.relu == 2) : (msk && p.mask_leaky && p.relu == 0 && !p.bias));
      if (!form_ok) return SISR_ERR_UNSUPPORTED;
      if (ksel == 3) {  // always the 2-row tile, both input chunks resident in LDS
        p.tiles_h = (H + 1) / 2;
        const dim3 g3((unsigned)((long)p.tiles_w * p.tiles_h * B), 1);
        const size_t lb3 = 2 * 4 * HALO_W * 64 * sizeof(float);
        SISR_ALLOW_LDS((conv3x3_c64_v4_kernel<false, true, false, 1, false, false, true, 3>), lb3);
        hipLaunchKernelGGL((conv3x3_c64_v4_kernel<false, true, false, 1, false, false, true, 3>), g3, dim3(256), lb3, st, p);
        return sisr_check_launch();
      }
      const bool small = nblk * p.cout_chunks < SMALL_GRID_BLOCKS;
      dim3 g = grid;
      size_t lb = HALO_H * HALO_W * 64 * sizeof(float);
      if (small) {
        p.tiles_h = (H + 1) / 2;
        g = dim3((unsigned)((long)p.tiles_w * p.tiles_h * B), p.cout_chunks);
        lb = 4 * HALO_W * 64 * sizeof(float);
      }
#define V4K(MK, MTV, LK, KS) hipLaunchKernelGGL((conv3x3_c64_v4_kernel<false, MK, false, MTV, false, false, LK, KS>), g, dim3(256), lb, st, p)
      if (ksel == 1) { if (small) V4K(false, 1, false, 1); else V4K(false, 2, false, 1); }
      else { if (small) V4K(false, 1, true, 2); else V4K(false, 2, true, 2); }
#undef V4K
      return sisr_check_launch();
    }
    if (!(in_shift && !in_scale) && !(aff && !msk) && !(msk && rs) && !leaky && !head && !tail && cin == 64 && cout == 64 &&
        memcmp(xview, yview, 6 * sizeof(int64_t)) == 0 && p.xv.clo == 64 && p.xv.chi == 0 && sisr_use_persistent(variant, nblk)) {
      // persistent form (plain 64 -> 64 maps): see conv3x3_c64_p4_kernel
      const dim3 gp((unsigned)(nblk < 512 ? nblk : 512));
      const size_t lbp = HALO_H * HALO_W * 64 * sizeof(float);
#define P4K(AF, MK, RS) hipLaunchKernelGGL((conv3x3_c64_p4_kernel<AF, MK, RS, false, false>), gp, dim3(256), lbp, st, p, (int)nblk)
      if (aff) P4K(true, true, false);
      else if (msk) P4K(false, true, false);
      else if (rs) P4K(false, false, true);
      else P4K(false, false, false);
#undef P4K
      return sisr_check_launch();
    }
    if (!(in_shift && !in_scale) && !(aff && !msk) && !(msk && rs)) {
      // Grids with fewer 4-row workgroups than CUs (a single 128x128 sample) leave half the chip idle: halve the tile.  Variant 5 / 6 force
      // the 4-row / 2-row kernel (A/B measurements); results are bit-identical either way.
      const bool small = variant == 6 || (variant == 4 && nblk * p.cout_chunks < SMALL_GRID_BLOCKS);
      if (small) {
        p.tiles_h = (H + 1) / 2;
        const dim3 grid2((unsigned)((long)p.tiles_w * p.tiles_h * B), p.cout_chunks);
        const size_t lb2 = 4 * HALO_W * 64 * sizeof(float);
        if (leaky && msk)
          hipLaunchKernelGGL((conv3x3_c64_v4_kernel<false, true, false, 1, false, false, true>), grid2, dim3(256), lb2, st, p);
        else if (leaky)
          hipLaunchKernelGGL((conv3x3_c64_v4_kernel<false, false, false, 1, false, false, true>), grid2, dim3(256), lb2, st, p);
        else if (aff && head)
          hipLaunchKernelGGL((conv3x3_c64_v4_kernel<true, true, false, 1, false, false, false, 0, 2>), grid2, dim3(256), lb2 + SISR_HEAD_LDS, st, p);
        else if (aff)
          hipLaunchKernelGGL((conv3x3_c64_v4_kernel<true, true, false, 1>), grid2, dim3(256), lb2, st, p);
        else if (msk)
          hipLaunchKernelGGL((conv3x3_c64_v4_kernel<false, true, false, 1>), grid2, dim3(256), lb2, st, p);
        else if (rs)
          hipLaunchKernelGGL((conv3x3_c64_v4_kernel<false, false, true, 1>), grid2, dim3(256), lb2, st, p);
        else
          hipLaunchKernelGGL((conv3x3_c64_v4_kernel<false, false, false, 1>), grid2, dim3(256), lb2, st, p);
        return sisr_check_launch();
      }
      const size_t lb = HALO_H * HALO_W * 64 * sizeof(float);
      if (leaky && msk)
        hipLaunchKernelGGL((conv3x3_c64_v4_kernel<false, true, false, 2, false, false, true>), grid, dim3(256), lb, st, p);
      else if (leaky)
        hipLaunchKernelGGL((conv3x3_c64_v4_kernel<false, false, false, 2, false, false, true>), grid, dim3(256), lb, st, p);
      else if (aff && head)
        hipLaunchKernelGGL((conv3x3_c64_v4_kernel<true, true, false, 2, false, false, false, 0, 2>), grid, dim3(256), lb + SISR_HEAD_LDS, st, p);
      else if (aff)
        hipLaunchKernelGGL((conv3x3_c64_v4_kernel<true, true, false, 2>), grid, dim3(256), lb, st, p);
      else if (msk)
        hipLaunchKernelGGL((conv3x3_c64_v4_kernel<false, true, false, 2>), grid, dim3(256), lb, st, p);
      else if (rs)
        hipLaunchKernelGGL((conv3x3_c64_v4_kernel<false, false, true, 2>), grid, dim3(256), lb, st, p);
      else
        hipLaunchKernelGGL((conv3x3_c64_v4_kernel<false, false, false, 2>), grid, dim3(256), lb, st, p);
      return sisr_check_launch();
    }
    if (tail || head) return SISR_ERR_UNSUPPORTED;  // the general kernel has neither tails nor heads
    const size_t lb = HALO_H * HALO_W * 64 * sizeof(float);
    hipLaunchKernelGGL(conv3x3_c64_kernel<0>, grid, dim3(256), lb, st, p);
    return sisr_check_launch();
  }
  if (tail || head) return SISR_ERR_UNSUPPORTED;
  const size_t lb = HALO_H * HALO_W * 64 * sizeof(float);
#ifdef SISR_DIAG
  if (variant == 13)
    hipLaunchKernelGGL(conv3x3_c64_kernel<3>, grid, dim3(256), lb, (hipStream_t)stream, p);
  else if (variant == 16)
    hipLaunchKernelGGL(conv3x3_c64_kernel<6>, grid, dim3(256), lb, (hipStream_t)stream, p);
  else
#endif
    hipLaunchKernelGGL(conv3x3_c64_kernel<0>, grid, dim3(256), lb, (hipStream_t)stream, p);
  return sisr_check_launch();
}

// SPARNet's ConvLayer convs on the issue-lean kernel's GEO form (see conv3x3_c64_v4_kernel): plain bias epilogue [+ per-strip
// channel sums of the output in `gap_partial`, the layout of sisr_conv3x3_c64's].
//   mode 1  y[B][H][W][cout] = Conv2d(3x3, no padding)(ReflectionPad2d(1)(nearest-upsample^up(x))), x: [B][H >> up][W >> up][cin]
//   mode 2  y[B][H][W][cout] = the zero-padded 3x3 conv of x: [B][H - 2][W - 2][cin] placed at offset (1, 1) of an H x W zero
//           map: with the flipped / role-swapped packed weight, the input gradient of mode 1's valid conv before the
//           reflection is folded back
// kreal: the caller's promise that input channels >= kreal are zero (cin == 64 only): 8 / 32 select the 1- / 4-octet K loops.
extern "C" int sisr_conv3x3_c64_geo(const float* x, const int64_t* xview, const float* wpacked, const float* bias, float* y,
                                    const int64_t* yview, float* gap_partial, int B, int H, int W, int cin, int cout, int mode,
                                    int up, int kreal, void* stream) {
  if (!x || !wpacked || !y || !xview || !yview || B <= 0 || H <= 0 || W <= 0) return SISR_ERR_ARG;
  if ((cin & 63) || (cout & 63) || cin <= 0 || cout <= 0) return SISR_ERR_UNSUPPORTED;
  if (mode < 1 || mode > 4) return SISR_ERR_ARG;
  if (up < 0 || up > 1 || (mode != 1 && up) || kreal < 0) return SISR_ERR_UNSUPPORTED;
  const bool sub = mode == 4;  // mode 2 on the zero-stuffed gradient of a stride-2 conv: x is [B][(H - 1) / 2][(W - 1) / 2][cin]
  if (sub) mode = 2;
  if (mode != 2 && (H < 2 || W < 2 || ((H | W) & ((1 << up) - 1)))) return SISR_ERR_ARG;  // ReflectionPad2d(1) needs 2 pixels
  if (mode == 2 && (H < 3 || W < 3)) return SISR_ERR_ARG;
  if (!sisr_aligned16(x) || !sisr_aligned16(wpacked) || !sisr_aligned16(y)) return SISR_ERR_ALIGN;
  ConvParams p;
  memset(&p, 0, sizeof(p));
#ifdef SISR_DIAG
  p.stamp = nullptr;
#endif
  p.x = x;
  p.xv = view_from(xview);
  p.y = y;
  p.yv = view_from(yview);
  if ((p.xv.sB | p.xv.sH | p.xv.sW | p.xv.chi | p.xv.clo) & 3) return SISR_ERR_ALIGN;
  p.w = wpacked;
  p.bias = bias;
  p.gap = gap_partial;
  p.alpha = 1.f;
  p.bias_n = 1;
  p.bias_q = 64;
  p.B = B;
  p.cin_chunks = cin / 64;
  p.cout_chunks = cout / 64;
  p.geo_reflect = mode != 2;
  p.geo_up = sub ? 1 : up;
  p.geo_sub = sub;
  p.geo_off = mode == 2;
  p.geo_h = mode == 2 ? H - 2 : H;
  p.geo_w = mode == 2 ? W - 2 : W;
  if (mode == 3) {  // H, W are the INPUT's; the output has every second pixel of the stride-1 result
    H = (H + 1) / 2;
    W = (W + 1) / 2;
  }
  p.H = H;
  p.W = W;
  p.tiles_w = (W + TW - 1) / TW;
  p.tiles_h = (H + TH - 1) / TH;
  const long nblk = (long)p.tiles_w * p.tiles_h * B;
  if (nblk > 0x7fffffffL) return SISR_ERR_ARG;
  const int ksel = (cin == 64 && kreal > 0 && kreal <= 8) ? 5 : ((cin == 64 && kreal > 0 && kreal <= 32) ? 4 : 0);
  const bool small = nblk * p.cout_chunks < SMALL_GRID_BLOCKS;
  dim3 g((unsigned)nblk, p.cout_chunks);
  size_t lb = HALO_H * HALO_W * 64 * sizeof(float);
  if (small) {
    p.tiles_h = (H + 1) / 2;
    g = dim3((unsigned)((long)p.tiles_w * p.tiles_h * B), p.cout_chunks);
    lb = 4 * HALO_W * 64 * sizeof(float);
  }
  hipStream_t st = (hipStream_t)stream;
  if (mode == 3) {  // 2-row tiles, 5 x 66-pixel halo
    p.tiles_h = (H + 1) / 2;
    g = dim3((unsigned)((long)p.tiles_w * p.tiles_h * B), p.cout_chunks);
    lb = 5 * 66 * 64 * sizeof(float);
    if (ksel == 4) {
      SISR_ALLOW_LDS((conv3x3_c64_v4_kernel<false, false, false, 1, false, false, false, 4, 0, true, true>), lb);
      hipLaunchKernelGGL((conv3x3_c64_v4_kernel<false, false, false, 1, false, false, false, 4, 0, true, true>), g, dim3(256), lb, st, p);
    } else {
      SISR_ALLOW_LDS((conv3x3_c64_v4_kernel<false, false, false, 1, false, false, false, 0, 0, true, true>), lb);
      hipLaunchKernelGGL((conv3x3_c64_v4_kernel<false, false, false, 1, false, false, false, 0, 0, true, true>), g, dim3(256), lb, st, p);
    }
    return sisr_check_launch();
  }
#define V4G(MTV, KS) hipLaunchKernelGGL((conv3x3_c64_v4_kernel<false, false, false, MTV, false, false, false, KS, 0, true>), g, dim3(256), lb, st, p)
  if (ksel == 5) { if (small) V4G(1, 5); else V4G(2, 5); }
  else if (ksel == 4) { if (small) V4G(1, 4); else V4G(2, 4); }
  else { if (small) V4G(1, 0); else V4G(2, 0); }
#undef V4G
  return sisr_check_launch();
}

extern "C" int sisr_pack_conv3x3_bf16_both(const float* w, void* packed_fwd, void* packed_dgrad, int cout, int cin,
                                           int shuffle_r, void* stream) {
  if (!w || !packed_fwd || !packed_dgrad || cout <= 0 || cin <= 0 || shuffle_r < 1) return SISR_ERR_ARG;
  if ((cin & 63) || (cout & 63)) return SISR_ERR_UNSUPPORTED;
  if (shuffle_r > 1 && cout != 64 * shuffle_r * shuffle_r) return SISR_ERR_UNSUPPORTED;
  const long total = (long)cout * cin * 9;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_conv3x3_bf16_both_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w,
                     (__bf16*)packed_fwd, (__bf16*)packed_dgrad, cout, cin, shuffle_r);
  return sisr_check_launch();
}

extern "C" int sisr_conv3x3_c64_bf16(const float* x, const int64_t* xview, const void* wpacked_bf16, const float* bias,
                                     int bias_n, int bias_q, float* y, const int64_t* yview, const float* res,
                                     const float* mask, const float* in_scale, const float* in_shift,
                                     const float* out_scale, float alpha, int relu, float* gap_partial,
                                     const float* gate_add, float* gate_out, const float* dot, int B, int H, int W,
                                     int cin, int cout, int select, void* stream) {
  if (!x || !wpacked_bf16 || !y || !xview || !yview || B <= 0 || H <= 0 || W <= 0) return SISR_ERR_ARG;
  if (select != 0 && select != 1) return SISR_ERR_ARG;
  const bool persist = select != 1;  // per call: 0 = persistent tile loop where it applies, 1 = per-tile kernel
  const bool gate = gate_add != nullptr;
  if (gate_add || gate_out || dot) {  // fused gated-residual chain, same contract as the fp32 entry
    if (gate != (gate_out != nullptr) || (gate && !in_scale) || (gate && dot) || (dot && !gap_partial) ||
        (!gate && in_scale) || in_shift || mask || out_scale || cin != 64 || cout != 64 || !sisr_aligned16(gate_add) ||
        !sisr_aligned16(gate_out) || memcmp(xview, yview, 6 * sizeof(int64_t)) != 0)
      return SISR_ERR_UNSUPPORTED;
  }
  if ((cin & 63) || (cout & 63) || cin <= 0 || cout <= 0) return SISR_ERR_UNSUPPORTED;
  if (in_shift && !in_scale) return SISR_ERR_UNSUPPORTED;
  if (!sisr_aligned16(x) || !sisr_aligned16(wpacked_bf16) || !sisr_aligned16(in_scale) || !sisr_aligned16(in_shift))
    return SISR_ERR_ALIGN;
  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.x = x;
  p.xv = view_from(xview);
  p.y = y;
  p.yv = view_from(yview);
  if ((p.xv.sB | p.xv.sH | p.xv.sW | p.xv.chi | p.xv.clo) & 3) return SISR_ERR_ALIGN;
  p.res = res;
  p.mask = mask;
  p.w = reinterpret_cast<const float*>(wpacked_bf16);
  p.bias = bias;
  p.in_scale = in_scale;
  p.in_shift = in_shift;
  p.out_scale = out_scale;
  p.gap = gap_partial;
  p.gate_add = gate_add;
  p.gate_out = gate_out;
  p.dot = dot;
  p.alpha = alpha;
  p.bias_n = bias_n;
  p.bias_q = bias_q;
  p.B = B;
  p.H = H;
  p.W = W;
  p.cin_chunks = cin / 64;
  p.cout_chunks = cout / 64;
  if (relu != 0 && relu != 1) return SISR_ERR_UNSUPPORTED;  // LeakyReLU codes: fp32 kernels only
  p.relu = relu;
  p.tiles_w = (W + TW - 1) / TW;
  p.tiles_h = (H + TH - 1) / TH;
  const long nblk = (long)p.tiles_w * p.tiles_h * B;
  if (nblk > 0x7fffffffL) return SISR_ERR_ARG;
  const dim3 grid((unsigned)nblk, p.cout_chunks);
  if (gap_partial && !dot && (mask || res)) return SISR_ERR_UNSUPPORTED;
  const size_t lb_halo = HALO_H * HALO_W * BH_PIX, lb_out = TH * TW * BE_LD * sizeof(float);
  const size_t lb = lb_halo > lb_out ? lb_halo : lb_out;
  hipStream_t st = (hipStream_t)stream;
  if (persist && cin == 64 && cout == 64 && nblk >= 1024) {
    // 64 -> 64 with enough tiles to give every persistent workgroup at least two: the double-buffered tile loop
    const int G = 512;  // two workgroups per CU
    const size_t plb = 2 * (size_t)PB_BUF;
    const dim3 pg(G);
    const int total = (int)nblk;
#define PBX(AF, MK, RS, GT, DT)                                                                              \
  do {                                                                                                       \
    SISR_ALLOW_LDS((conv3x3_c64_bf16_persist_kernel<AF, MK, RS, GT, DT>), plb);                              \
    hipLaunchKernelGGL((conv3x3_c64_bf16_persist_kernel<AF, MK, RS, GT, DT>), pg, dim3(256), plb, st, p, total); \
  } while (0)
    if (gate) { if (res) PBX(false, false, true, true, false); else PBX(false, false, false, true, false); }
    else if (dot) { if (res) PBX(false, false, true, false, true); else PBX(false, false, false, false, true); }
    else {
      const int sel = (in_scale ? 4 : 0) | (mask ? 2 : 0) | (res ? 1 : 0);
      switch (sel) {
        case 0: PBX(false, false, false, false, false); break;
        case 1: PBX(false, false, true, false, false); break;
        case 2: PBX(false, true, false, false, false); break;
        case 3: PBX(false, true, true, false, false); break;
        case 4: PBX(true, false, false, false, false); break;
        case 5: PBX(true, false, true, false, false); break;
        case 6: PBX(true, true, false, false, false); break;
        case 7: PBX(true, true, true, false, false); break;
      }
    }
#undef PBX
    return sisr_check_launch();
  }
  if (gate || dot) {
#define BFX(RS, GT, DT) hipLaunchKernelGGL((conv3x3_c64_bf16_kernel<false, false, RS, GT, DT>), grid, dim3(256), lb, st, p)
    if (gate) { if (res) BFX(true, true, false); else BFX(false, true, false); }
    else      { if (res) BFX(true, false, true); else BFX(false, false, true); }
#undef BFX
    return sisr_check_launch();
  }
  const int sel = (in_scale ? 4 : 0) | (mask ? 2 : 0) | (res ? 1 : 0);
  switch (sel) {
#define BF_CASE(i, a, m, r) \
  case i: hipLaunchKernelGGL((conv3x3_c64_bf16_kernel<a, m, r>), grid, dim3(256), lb, st, p); break;
    BF_CASE(0, false, false, false)
    BF_CASE(1, false, false, true)
    BF_CASE(2, false, true, false)
    BF_CASE(3, false, true, true)
    BF_CASE(4, true, false, false)
    BF_CASE(5, true, false, true)
    BF_CASE(6, true, true, false)
    BF_CASE(7, true, true, true)
#undef BF_CASE
  }
  return sisr_check_launch();
}

// bf16 operands AND bf16 storage of selected maps (64 -> 64 only; always the persistent tile loop).  storage bits:
//   1  x, gate_add and gate_out are bf16 maps      2  y is a bf16 map (rounded to nearest even in the epilogue's store)
//   4  mask and dot are bf16 maps                   8  res is a bf16 map
// A bf16 map has the View of its fp32 twin (strides in ELEMENTS) and 2-byte elements; pointers are passed as float* and
// must be 16-byte aligned.  Everything else (bias, scales, partial sums, arithmetic: bf16 operands, fp32 accumulate) is
// sisr_conv3x3_c64_bf16's.  Only the combinations the fused residual-group node launches are built; others return
// SISR_ERR_UNSUPPORTED.
extern "C" int sisr_conv3x3_c64_bf16s(const float* x, const int64_t* xview, const void* wpacked_bf16, const float* bias,
                                      int bias_n, int bias_q, float* y, const int64_t* yview, const float* res,
                                      const float* mask, const float* in_scale, const float* in_shift, float alpha, int relu,
                                      float* gap_partial, const float* gate_add, float* gate_out, const float* dot, int B,
                                      int H, int W, int storage, void* stream) {
  if (!x || !wpacked_bf16 || !y || !xview || !yview || B <= 0 || H <= 0 || W <= 0 || storage < 0 || storage > 15) return SISR_ERR_ARG;
#ifdef SISR_DIAG
  const float* stamp_buf = nullptr;  // tools/bf16s_timeline.py: the plain bf16 -> bf16 form with in-kernel cycle sums
  if (getenv("SISR_BF16S_STAMP") && storage == 3 && dot && !gap_partial && !gate_add && !mask && !res && !in_scale) {
    stamp_buf = dot;
    dot = nullptr;
  }
#endif
  const bool gate = gate_add != nullptr;
  if (gate != (gate_out != nullptr) || (gate && !in_scale) || (gate && dot) || (dot && !gap_partial) ||
      (gate && (in_shift || mask)) || memcmp(xview, yview, 6 * sizeof(int64_t)) != 0)
    return SISR_ERR_UNSUPPORTED;
  if (in_shift && !in_scale) return SISR_ERR_UNSUPPORTED;
  if (gap_partial && !dot && (mask || res)) return SISR_ERR_UNSUPPORTED;
  if (relu != 0 && relu != 1) return SISR_ERR_UNSUPPORTED;
  if (!sisr_aligned16(x) || !sisr_aligned16(y) || !sisr_aligned16(wpacked_bf16) || !sisr_aligned16(in_scale) ||
      !sisr_aligned16(in_shift) || !sisr_aligned16(gate_add) || !sisr_aligned16(gate_out) || !sisr_aligned16(res) ||
      !sisr_aligned16(mask) || !sisr_aligned16(dot))
    return SISR_ERR_ALIGN;
  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.x = x;
  p.xv = view_from(xview);
  p.y = y;
  p.yv = view_from(yview);
  if ((p.xv.sB | p.xv.sH | p.xv.sW | p.xv.chi | p.xv.clo) & 7) return SISR_ERR_ALIGN;  // 16-B pieces of 2-byte elements
  p.res = res;
  p.mask = mask;
  p.w = reinterpret_cast<const float*>(wpacked_bf16);
  p.bias = bias;
  p.in_scale = in_scale;
  p.in_shift = in_shift;
  p.gap = gap_partial;
  p.gate_add = gate_add;
  p.gate_out = gate_out;
  p.dot = dot;
  p.alpha = alpha;
  p.bias_n = bias_n;
  p.bias_q = bias_q;
  p.B = B;
  p.H = H;
  p.W = W;
  p.cin_chunks = p.cout_chunks = 1;
  p.relu = relu;
  p.tiles_w = (W + TW - 1) / TW;
  p.tiles_h = (H + TH - 1) / TH;
  const long nblk = (long)p.tiles_w * p.tiles_h * B;
  if (nblk > 0x7fffffffL) return SISR_ERR_ARG;
  const int G = (int)(nblk < 512 ? nblk : 512);  // two workgroups per CU; fewer tiles than that: one tile each
  const size_t plb = 2 * (size_t)PB_BUF;
  const dim3 pg(G);
  const int total = (int)nblk;
  hipStream_t st = (hipStream_t)stream;
#define PSX(AF, MK, RS, GT, DT, I16, O16, A16, R16)                                                                     \
  do {                                                                                                                  \
    SISR_ALLOW_LDS((conv3x3_c64_bf16_persist_kernel<AF, MK, RS, GT, DT, I16, O16, A16, R16>), plb);                     \
    hipLaunchKernelGGL((conv3x3_c64_bf16_persist_kernel<AF, MK, RS, GT, DT, I16, O16, A16, R16>), pg, dim3(256), plb, st, \
                       p, total);                                                                                       \
    return sisr_check_launch();                                                                                         \
  } while (0)
  const int form = (in_scale && !gate ? 16 : 0) | (mask ? 8 : 0) | (res ? 4 : 0) | (gate ? 2 : 0) | (dot ? 1 : 0);
#ifdef SISR_DIAG
  if (stamp_buf) {
    p.dot = stamp_buf;
    SISR_ALLOW_LDS((conv3x3_c64_bf16_persist_kernel<false, false, false, false, false, true, true, false, false, true>), plb);
    hipLaunchKernelGGL((conv3x3_c64_bf16_persist_kernel<false, false, false, false, false, true, true, false, false, true>), pg,
                       dim3(256), plb, st, p, total);
    return sisr_check_launch();
  }
#endif
  // forward of the group node: bf16 activations in and out
  if (storage == 3 && form == 0) PSX(false, false, false, false, false, true, true, false, false);   // conv (+ ReLU / GAP sums)
  if (storage == 3 && form == 2) PSX(false, false, false, true, false, true, true, false, false);    // GATE prologue
  if (storage == 1 && form == 6) PSX(false, false, true, true, false, true, false, false, false);    // group tail: + fp32 residual, fp32 out
  // backward with fp32 gradient maps: only the saved activations (mask, dot) are bf16
  if (storage == 4 && form == 1) PSX(false, false, false, false, true, false, false, true, false);   // first conv, DOT
  if (storage == 4 && form == 5) PSX(false, false, true, false, true, false, false, true, false);    // dgrad + residual, DOT
  if (storage == 4 && form == 24) PSX(true, true, false, false, false, false, false, true, false);   // dgrad, ReLU mask + affine
  // backward with bf16 gradient maps as well
  if (storage == 6 && form == 1) PSX(false, false, false, false, true, false, true, true, false);    // fp32 dOut -> bf16 dU, DOT
  if (storage == 15 && form == 5) PSX(false, false, true, false, true, true, true, true, true);
  if (storage == 7 && form == 24) PSX(true, true, false, false, false, true, true, true, false);
  if (storage == 11 && form == 4) PSX(false, false, true, false, false, true, true, false, true);    // dgrad + residual (bf16 out)
  if (storage == 9 && form == 4) PSX(false, false, true, false, false, true, false, false, true);    // block 0: ... -> fp32 dX of the group
#undef PSX
  return SISR_ERR_UNSUPPORTED;
}

// bf16x3: fp32 through the bf16 matrix cores (three-way operand split, six products); same contract as
// sisr_conv3x3_c64_bf16, packed weights from sisr_pack_conv3x3_x3_both (three bf16 planes).
extern "C" int sisr_conv3x3_c64_x3(const float* x, const int64_t* xview, const void* wpacked_bf16, const float* bias,
                                     int bias_n, int bias_q, float* y, const int64_t* yview, const float* res,
                                     const float* mask, const float* in_scale, const float* in_shift,
                                     const float* out_scale, float alpha, int relu, float* gap_partial,
                                     const float* gate_add, float* gate_out, const float* dot, int B, int H, int W,
                                     int cin, int cout, int select, void* stream) {
  if (!x || !wpacked_bf16 || !y || !xview || !yview || B <= 0 || H <= 0 || W <= 0) return SISR_ERR_ARG;
  if (select != 0) return SISR_ERR_ARG;  // one kernel family; the argument keeps the three conv entries call-compatible
  const bool gate = gate_add != nullptr;
  if (gate_add || gate_out || dot) {  // fused gated-residual chain, same contract as the fp32 entry
    if (gate != (gate_out != nullptr) || (gate && !in_scale) || (gate && dot) || (dot && !gap_partial) ||
        (!gate && in_scale) || in_shift || mask || out_scale || cin != 64 || cout != 64 || !sisr_aligned16(gate_add) ||
        !sisr_aligned16(gate_out) || memcmp(xview, yview, 6 * sizeof(int64_t)) != 0)
      return SISR_ERR_UNSUPPORTED;
  }
  if ((cin & 63) || (cout & 63) || cin <= 0 || cout <= 0) return SISR_ERR_UNSUPPORTED;
  if (in_shift && !in_scale) return SISR_ERR_UNSUPPORTED;
  if (!sisr_aligned16(x) || !sisr_aligned16(wpacked_bf16) || !sisr_aligned16(in_scale) || !sisr_aligned16(in_shift))
    return SISR_ERR_ALIGN;
  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.x = x;
  p.xv = view_from(xview);
  p.y = y;
  p.yv = view_from(yview);
  if ((p.xv.sB | p.xv.sH | p.xv.sW | p.xv.chi | p.xv.clo) & 3) return SISR_ERR_ALIGN;
  p.res = res;
  p.mask = mask;
  p.w = reinterpret_cast<const float*>(wpacked_bf16);
  p.bias = bias;
  p.in_scale = in_scale;
  p.in_shift = in_shift;
  p.out_scale = out_scale;
  p.gap = gap_partial;
  p.gate_add = gate_add;
  p.gate_out = gate_out;
  p.dot = dot;
  p.alpha = alpha;
  p.bias_n = bias_n;
  p.bias_q = bias_q;
  p.B = B;
  p.H = H;
  p.W = W;
  p.cin_chunks = cin / 64;
  p.cout_chunks = cout / 64;
  if (relu != 0 && relu != 1) return SISR_ERR_UNSUPPORTED;  // LeakyReLU codes: fp32 kernels only
  p.relu = relu;
  p.tiles_w = (W + TW - 1) / TW;
  p.tiles_h = (H + TH - 1) / TH;
  const long nblk = (long)p.tiles_w * p.tiles_h * B;
  if (nblk > 0x7fffffffL) return SISR_ERR_ARG;
  const dim3 grid((unsigned)nblk, p.cout_chunks);
  if (gap_partial && !dot && (mask || res)) return SISR_ERR_UNSUPPORTED;
  const size_t lb_halo = 3 * (size_t)X3_PLANE, lb_out = TH * TW * BE_LD * sizeof(float);
  const size_t lb = lb_halo > lb_out ? lb_halo : lb_out;
  hipStream_t st = (hipStream_t)stream;
  const long wplane = (long)cin * cout * 9 * 2;  // bytes between the hi / mid / lo planes of the packed weights
#define X3L(AF, MK, RS, GT, DT)                                                                        \
  do {                                                                                                 \
    SISR_ALLOW_LDS((conv3x3_c64_x3_kernel<AF, MK, RS, GT, DT>), lb);                                   \
    hipLaunchKernelGGL((conv3x3_c64_x3_kernel<AF, MK, RS, GT, DT>), grid, dim3(256), lb, st, p, wplane); \
  } while (0)
#ifdef SISR_DIAG
  if (getenv("SISR_X3_STAMP") && dot && !gate && !in_scale && !mask && !res) {  // tools/x3_phases.py
    SISR_ALLOW_LDS((conv3x3_c64_x3_kernel<false, false, false, false, false, 4, true>), lb);
    hipLaunchKernelGGL((conv3x3_c64_x3_kernel<false, false, false, false, false, 4, true>), grid, dim3(256), lb, st, p, wplane);
    return sisr_check_launch();
  }
  static const int x3_bd = getenv("SISR_X3_BD") ? atoi(getenv("SISR_X3_BD")) : 4;  // diagnostic A/B of the plain form
  if (x3_bd == 6 && !gate && !dot && !in_scale && !mask && !res) {
    SISR_ALLOW_LDS((conv3x3_c64_x3_kernel<false, false, false, false, false, 6>), lb);
    hipLaunchKernelGGL((conv3x3_c64_x3_kernel<false, false, false, false, false, 6>), grid, dim3(256), lb, st, p, wplane);
    return sisr_check_launch();
  }
#endif
  if (gate) { if (res) X3L(false, false, true, true, false); else X3L(false, false, false, true, false); }
  else if (dot) { if (res) X3L(false, false, true, false, true); else X3L(false, false, false, false, true); }
  else {
    const int sel = (in_scale ? 4 : 0) | (mask ? 2 : 0) | (res ? 1 : 0);
    switch (sel) {
      case 0: X3L(false, false, false, false, false); break;
      case 1: X3L(false, false, true, false, false); break;
      case 2: X3L(false, true, false, false, false); break;
      case 3: X3L(false, true, true, false, false); break;
      case 4: X3L(true, false, false, false, false); break;
      case 5: X3L(true, false, true, false, false); break;
      case 6: X3L(true, true, false, false, false); break;
      case 7: X3L(true, true, true, false, false); break;
    }
  }
#undef X3L
  return sisr_check_launch();
}

extern "C" int sisr_pack_conv3x3_x3_both(const float* w, void* packed_fwd, void* packed_dgrad, int cout, int cin,
                                         int shuffle_r, void* stream) {
  if (!w || !packed_fwd || !packed_dgrad || cout <= 0 || cin <= 0 || (cout & 63) || (cin & 63) || shuffle_r < 1)
    return SISR_ERR_ARG;
  if (shuffle_r > 1 && cout != 64 * shuffle_r * shuffle_r) return SISR_ERR_UNSUPPORTED;
  const long total = (long)cout * cin * 9;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_conv3x3_x3_both_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w,
                     static_cast<__bf16*>(packed_fwd), static_cast<__bf16*>(packed_dgrad), cout, cin, shuffle_r);
  return sisr_check_launch();
}

#ifdef SISR_DIAG
// Diagnostic: resident workgroups per CU the runtime computes for the plain bf16x3 / fp32 kernels with their LDS sizes.
extern "C" int sisr_diag_conv_occupancy(int which) {
  int n = -1;
  if (which == 0) {
    const size_t lb = 3 * (size_t)X3_PLANE;
    SISR_ALLOW_LDS((conv3x3_c64_x3_kernel<false, false, false, false, false>), lb);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv3x3_c64_x3_kernel<false, false, false, false, false>, 256, lb) != hipSuccess) return -2;
  } else {
    const size_t lb = HALO_H * HALO_W * 64 * sizeof(float);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv3x3_c64_v4_kernel<false, false, false, 2>, 256, lb) != hipSuccess) return -2;
  }
  return n;
}
#endif
