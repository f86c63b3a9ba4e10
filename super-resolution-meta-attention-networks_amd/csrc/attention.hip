// Channel-attention / meta-attention gates and the gated residual, fp32.
//
//   CA gate   (ref: advanced/architectures.py:13-32 CALayer; attention_manipulators/architectures.py:125
//              QCALayer style 'standard'):  s = mean_hw(t);  h = relu(W1 s + b1);  ca = sigmoid(W2 h + b2)
//   meta gate (ref: attention_manipulators/q_layer.py:20-43 ParaCALayer):
//              m = sigmoid(V2 act(V1 md + c1) + c2),  act = ReLU or identity
//   block out (ref: advanced/architectures.py:68-71 RCAB, attention_manipulators/architectures.py:172-180
//              QRCAB, :348-356 ParamResBlock):       y = t * g[b,c] + x,   g = ca, ca*m or m
// The GAP itself is produced by the conv epilogue as per-wave partial sums; these kernels finish
// it.  All reductions run in a fixed order (no atomics), so results are run-to-run reproducible.
// These are M = batch sized contractions (64x4, 10x32x64): VALU + wave shuffles, not MFMA.
#include "sisr_common.h"
#include "ca_gate.h"
#include <string.h>

// ---------------------------------------------------------------- CA gate forward (C = 64, one block per sample)
// part: [B][parts][64] partial sums.  Outputs s, ca, g: [B][64]; hid: [B][R].
__global__ __launch_bounds__(256) void ca_gate_fwd_kernel(const float* __restrict__ part, int parts, float inv_hw,
                                                          const float* __restrict__ w1, const float* __restrict__ b1,
                                                          const float* __restrict__ w2, const float* __restrict__ b2,
                                                          int R, const float* __restrict__ mul, float* __restrict__ s_out,
                                                          float* __restrict__ hid_out, float* __restrict__ ca_out,
                                                          float* __restrict__ g_out) {
  __shared__ __attribute__((aligned(16))) float red[16 * 64];
  ca_gate_fwd_sample<false>(part, parts, inv_hw, blockIdx.x, w1, b1, w2, b2, R, mul, s_out, hid_out, ca_out, g_out, red);
}

// ---------------------------------------------------------------- CA gate backward
// One launch (it sits on the serial backward chain of every block).  Per sample (one block each):
//   dg = sum of the partials of sum_hw dOut*t;  dca = dg*mul; dz2 = dca*ca*(1-ca); dh = W2^T dz2;
//   dz1 = dh*[hid>0]; ds = W1^T dz1  ->  shift[b][c] = ds*inv_hw (the GAP backward broadcast, consumed as the
//   dgrad / wgrad prologue shift), dmul = dg*ca, and dz2 / dz1 into the workspace ([B][80] rows).
// The block that finishes last (device-scope counter, reset for the next launch) then sums the parameter
// gradients over the batch in batch order -- which block does it does not change the result.
__global__ __launch_bounds__(256) void ca_gate_bwd_kernel(const float* __restrict__ dgpart, int parts, float inv_hw,
                                                          const float* __restrict__ w1, const float* __restrict__ w2,
                                                          int R, const float* __restrict__ s_in,
                                                          const float* __restrict__ hid, const float* __restrict__ ca_in,
                                                          const float* __restrict__ mul, float* __restrict__ shift,
                                                          float* __restrict__ dmul, float* dz2_out, float* dz1_out,
                                                          float* __restrict__ dw1, float* __restrict__ db1,
                                                          float* __restrict__ dw2, float* __restrict__ db2,
                                                          unsigned* counter, int B) {
  __shared__ __attribute__((aligned(16))) float red[16 * 64];
  __shared__ int is_last;
  ca_gate_bwd_sample<false>(dgpart, parts, inv_hw, blockIdx.x, w1, w2, R, hid, ca_in, mul, shift, dmul, dz2_out, dz1_out, red);
  if (!dw1) return;  // parameter gradients deferred to sisr_ca_gate_bwd_params_batch (dz2 / dz1 stay in the workspace)
  __threadfence();  // this block's dz2 / dz1 are visible device-wide before it is counted
  __syncthreads();
  if (threadIdx.x == 0) is_last = atomicAdd(counter, 1u) == (unsigned)(B - 1);
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  if (threadIdx.x == 0) *counter = 0u;
  ca_gate_bwd_params(dz2_out, dz1_out, hid, s_in, R, B, dw1, db1, dw2, db2);
}

// Parameter gradients of up to CA_PB gates in one launch (blockIdx.x = job): the chain kernel above then only produces what
// the next conv waits for (shift, dz2 / dz1), and a residual group's 20 gates pay one launch for their weight gradients.
// Same ca_gate_bwd_params code, so the sums are those of the in-kernel form.
#define CA_PB 32
struct CaParamJob {
  const float *dz, *hid, *s;  // dz: [B][80] workspace of that gate (per sample: dz2 [64], then dz1 [R <= 16])
  float *dw1, *db1, *dw2, *db2;
};
struct CaParamBatch {
  CaParamJob job[CA_PB];
};
__global__ __launch_bounds__(256) void ca_gate_bwd_params_batch_kernel(CaParamBatch bt, int B, int R) {
  const CaParamJob& j = bt.job[blockIdx.x];
  ca_gate_bwd_params(j.dz, j.dz + 64, j.hid, j.s, R, B, j.dw1, j.db1, j.dw2, j.db2);
}

// ---------------------------------------------------------------- meta gate (ParaCALayer) forward / backward
// md [B][M]; v1 [Hd][M]; c1 [Hd]; v2 [C][Hd]; c2 [C].  One block per sample, blockDim = 256.
__device__ __forceinline__ void meta_gate_fwd_body(const float* __restrict__ md, int M, int Hd, int C,
                                                   const float* __restrict__ v1, const float* __restrict__ c1,
                                                   const float* __restrict__ v2, const float* __restrict__ c2,
                                                   int relu, float* __restrict__ hid, float* __restrict__ m) {
  extern __shared__ float sm[];  // Hd floats
  const int b = blockIdx.x;
  for (int j = threadIdx.x; j < Hd; j += blockDim.x) {
    float z = c1[j];
    for (int k = 0; k < M; ++k) z += v1[(long)j * M + k] * md[(long)b * M + k];
    if (relu) z = fmaxf(z, 0.f);
    sm[j] = z;
    hid[(long)b * Hd + j] = z;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float z = c2[c];
    for (int j = 0; j < Hd; ++j) z += v2[(long)c * Hd + j] * sm[j];
    m[(long)b * C + c] = sigmoidf(z);
  }
}

__global__ __launch_bounds__(256) void meta_gate_fwd_kernel(const float* __restrict__ md, int M, int Hd, int C,
                                                            const float* __restrict__ v1, const float* __restrict__ c1,
                                                            const float* __restrict__ v2, const float* __restrict__ c2,
                                                            int relu, float* __restrict__ hid, float* __restrict__ m) {
  meta_gate_fwd_body(md, M, Hd, C, v1, c1, v2, c2, relu, hid, m);
}

// All L meta-attention layers of a network in one launch (blockIdx.y = layer): the gates depend on the metadata and
// on the layers' own weights only, never on features, so the 200 ParaCALayers of a QRCAN need not be 200 launches.
// Weight pointers come from device tables (the layers' parameters live wherever nn.Module put them); hid and m are
// [L][B][Hd] and [L][B][C].
struct MetaTables {
  const float* const* v1;
  const float* const* c1;
  const float* const* v2;
  const float* const* c2;
};

__global__ __launch_bounds__(256) void meta_gate_many_fwd_kernel(const float* __restrict__ md, int B, int M, int Hd, int C,
                                                                 MetaTables t, int relu, float* __restrict__ hid,
                                                                 float* __restrict__ m) {
  const int l = blockIdx.y;
  meta_gate_fwd_body(md, M, Hd, C, t.v1[l], t.c1[l], t.v2[l], t.c2[l], relu, hid + (long)l * B * Hd, m + (long)l * B * C);
}

// Backward in two stages (the single-block batch loop it replaces took ~0.5 ms per layer at B = 32):
//  sample stage (one block per sample): dz2 = dm*m*(1-m); dz1 = act'(hid) * V2^T dz2; dmd = V1^T dz1
//  param stage  (one thread per parameter element): dV2 = sum_b dz2 (x) hid, dc2 = sum_b dz2,
//                                                   dV1 = sum_b dz1 (x) md,  dc1 = sum_b dz1   (batch order)
__device__ __forceinline__ void meta_gate_bwd_sample_body(const float* __restrict__ dm, const float* __restrict__ m,
                                                          const float* __restrict__ hid, int M, int Hd, int C,
                                                          const float* __restrict__ v1, const float* __restrict__ v2,
                                                          int relu, float* __restrict__ dz2_out,
                                                          float* __restrict__ dz1_out, float* __restrict__ dmd) {
  extern __shared__ float sm[];  // dz2 [C] | dz1 [Hd] | v2 [C*Hd] | v1 [Hd*M]   (weights staged coalesced)
  float* dz2 = sm;
  float* dz1 = sm + C;
  float* v2s = dz1 + Hd;
  float* v1s = v2s + C * Hd;
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < C * Hd; i += 256) v2s[i] = v2[i];
  if (dmd)
    for (int i = tid; i < Hd * M; i += 256) v1s[i] = v1[i];
  for (int c = tid; c < C; c += 256) {
    const float mv = m[(long)b * C + c];
    const float d = dm[(long)b * C + c] * mv * (1.f - mv);
    dz2[c] = d;
    dz2_out[(long)b * C + c] = d;
  }
  __syncthreads();
  for (int j = tid; j < Hd; j += 256) {
    float dh = 0.f;
    for (int c = 0; c < C; ++c) dh += v2s[c * Hd + j] * dz2[c];
    if (relu && !(hid[(long)b * Hd + j] > 0.f)) dh = 0.f;
    dz1[j] = dh;
    dz1_out[(long)b * Hd + j] = dh;
  }
  if (dmd) {
    __syncthreads();
    for (int k = tid; k < M; k += 256) {
      float d = 0.f;
      for (int j = 0; j < Hd; ++j) d += v1s[j * M + k] * dz1[j];
      dmd[(long)b * M + k] = d;
    }
  }
}

__global__ __launch_bounds__(256) void meta_gate_bwd_sample_kernel(const float* __restrict__ dm, const float* __restrict__ m,
                                                                   const float* __restrict__ hid, int M, int Hd, int C,
                                                                   const float* __restrict__ v1,
                                                                   const float* __restrict__ v2, int relu,
                                                                   float* __restrict__ dz2_out, float* __restrict__ dz1_out,
                                                                   float* __restrict__ dmd) {
  meta_gate_bwd_sample_body(dm, m, hid, M, Hd, C, v1, v2, relu, dz2_out, dz1_out, dmd);
}

// blockIdx.y = layer; dm, m, dz2: [L][B][C]; hid, dz1: [L][B][Hd]; no metadata gradient (it would be a sum over layers)
__global__ __launch_bounds__(256) void meta_gate_many_bwd_sample_kernel(const float* __restrict__ dm,
                                                                        const float* __restrict__ m,
                                                                        const float* __restrict__ hid, int B, int M, int Hd,
                                                                        int C, MetaTables t, int relu,
                                                                        float* __restrict__ dz2_out,
                                                                        float* __restrict__ dz1_out) {
  const long l = blockIdx.y;
  meta_gate_bwd_sample_body(dm + l * B * C, m + l * B * C, hid + l * B * Hd, M, Hd, C, t.v1[l], t.v2[l], relu,
                            dz2_out + l * B * C, dz1_out + l * B * Hd, nullptr);
}

__device__ __forceinline__ void meta_gate_bwd_param_body(const float* __restrict__ dz2, const float* __restrict__ dz1,
                                                         const float* __restrict__ hid, const float* __restrict__ md,
                                                         int B, int M, int Hd, int C, float* __restrict__ dv1,
                                                         float* __restrict__ dc1, float* __restrict__ dv2,
                                                         float* __restrict__ dc2) {
  const long n2 = (long)C * Hd, n1 = (long)Hd * M;
  const long total = n2 + n1 + C + Hd;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    float acc = 0.f;
    if (i < n2) {
      const int c = (int)(i / Hd), j = (int)(i - (long)c * Hd);
      for (int b = 0; b < B; ++b) acc += dz2[(long)b * C + c] * hid[(long)b * Hd + j];
      dv2[i] = acc;
    } else if (i < n2 + n1) {
      const long e = i - n2;
      const int j = (int)(e / M), k = (int)(e - (long)j * M);
      for (int b = 0; b < B; ++b) acc += dz1[(long)b * Hd + j] * md[(long)b * M + k];
      dv1[e] = acc;
    } else if (i < n2 + n1 + C) {
      const int c = (int)(i - n2 - n1);
      for (int b = 0; b < B; ++b) acc += dz2[(long)b * C + c];
      dc2[c] = acc;
    } else {
      const int j = (int)(i - n2 - n1 - C);
      for (int b = 0; b < B; ++b) acc += dz1[(long)b * Hd + j];
      dc1[j] = acc;
    }
  }
}

__global__ __launch_bounds__(256) void meta_gate_bwd_param_kernel(const float* __restrict__ dz2, const float* __restrict__ dz1,
                                                                  const float* __restrict__ hid, const float* __restrict__ md,
                                                                  int B, int M, int Hd, int C, float* __restrict__ dv1,
                                                                  float* __restrict__ dc1, float* __restrict__ dv2,
                                                                  float* __restrict__ dc2) {
  meta_gate_bwd_param_body(dz2, dz1, hid, md, B, M, Hd, C, dv1, dc1, dv2, dc2);
}

// blockIdx.y = layer; gradients are written as [L][Hd*M], [L][Hd], [L][C*Hd], [L][C] (the caller hands out row views)
__global__ __launch_bounds__(256) void meta_gate_many_bwd_param_kernel(const float* __restrict__ dz2,
                                                                       const float* __restrict__ dz1,
                                                                       const float* __restrict__ hid,
                                                                       const float* __restrict__ md, int B, int M, int Hd,
                                                                       int C, float* __restrict__ dv1, float* __restrict__ dc1,
                                                                       float* __restrict__ dv2, float* __restrict__ dc2) {
  const long l = blockIdx.y;
  meta_gate_bwd_param_body(dz2 + l * B * C, dz1 + l * B * Hd, hid + l * B * Hd, md, B, M, Hd, C, dv1 + l * Hd * M,
                           dc1 + l * Hd, dv2 + l * C * Hd, dc2 + l * C);
}

// the same, every layer's four gradients written where the caller's tables point (the optimiser's gradient arena: the 800
// small gradients of a QRCAN then need no gather before the update)
struct MetaGradTables {
  float* const* dv1;
  float* const* dc1;
  float* const* dv2;
  float* const* dc2;
};
__global__ __launch_bounds__(256) void meta_gate_many_bwd_param_scatter_kernel(const float* __restrict__ dz2,
                                                                               const float* __restrict__ dz1,
                                                                               const float* __restrict__ hid,
                                                                               const float* __restrict__ md, int B, int M,
                                                                               int Hd, int C, MetaGradTables g) {
  const long l = blockIdx.y;
  meta_gate_bwd_param_body(dz2 + l * B * C, dz1 + l * B * Hd, hid + l * B * Hd, md, B, M, Hd, C, g.dv1[l], g.dc1[l], g.dv2[l],
                           g.dc2[l]);
}

// ---------------------------------------------------------------- generic gate MLP (the metadata-mixing QCALayer styles)
// ref: attention_manipulators/architectures.py:105-127.  After the global average pool every style is a 2..4 layer MLP
// on a <= 74-element vector per sample, with the metadata vector concatenated to some layers' inputs:
//   modulate            [64 -> 4 relu][4 -> 64 sigmoid]                       then y *= metadata (M == C)
//   max_concat / softmax [64+M -> 4 relu][4 -> 64 sigmoid]                    (softmax: then softmax over channels)
//   mini_concat         [64 -> 4][relu(cat(., md)) -> 64 sigmoid]             (the ReLU also hits the metadata)
//   extended_attention  [64+M -> 32 relu][32+M -> 16 relu][16+M -> 4 relu][4 -> 64 sigmoid]
// One workgroup per sample, everything in LDS, sums in index order.  An optional per-(b,c) factor `mul` (the block's
// meta-attention gate) is applied last.  The forward keeps every layer's output (acts) for the backward, which
// returns d pool, d metadata, d mul and, summed over the batch in batch order, the parameter gradients.
#define GM_MAXL 4
#define GM_MAXW 544  // widest layer input: 512 pooled channels + 32 metadata values
struct GateMlp {
  const float* w[GM_MAXL];
  const float* b[GM_MAXL];
  int nin[GM_MAXL], nout[GM_MAXL], cat[GM_MAXL], relu_in[GM_MAXL], act[GM_MAXL];
  int L, M, C, final_mode;  // final_mode: 0 none, 1 softmax over channels, 2 multiply by the metadata
};

__device__ __forceinline__ int gm_acts_width(const GateMlp& d) {
  int w = d.nin[0];
  for (int k = 0; k < d.L; ++k) w += d.nout[k];
  return w;
}

__global__ __launch_bounds__(256) void gate_mlp_fwd_kernel(const float* __restrict__ pool, const float* __restrict__ md,
                                                           const float* __restrict__ mul, GateMlp d,
                                                           float* __restrict__ acts, float* __restrict__ yfin,
                                                           float* __restrict__ y) {
  __shared__ float cur[GM_MAXW], nxt[GM_MAXW], mdv[GM_MAXW];
  const int b = blockIdx.x, t = threadIdx.x;
  const int AW = gm_acts_width(d);
  float* ab = acts + (long)b * AW;
  for (int j = t; j < d.nin[0]; j += 256) {
    const float v = pool[(long)b * d.nin[0] + j];
    cur[j] = v;
    ab[j] = v;
  }
  for (int j = t; j < d.M; j += 256) mdv[j] = md[(long)b * d.M + j];
  __syncthreads();
  int off = d.nin[0];
  for (int k = 0; k < d.L; ++k) {
    const int nin = d.nin[k], inw = nin + (d.cat[k] ? d.M : 0), nout = d.nout[k];
    for (int o = t; o < nout; o += 256) {
      float z = d.b[k] ? d.b[k][o] : 0.f;
      const float* wr = d.w[k] + (long)o * inw;
      for (int j = 0; j < inw; ++j) {
        float v = j < nin ? cur[j] : mdv[j - nin];
        if (d.relu_in[k]) v = fmaxf(v, 0.f);
        z += wr[j] * v;
      }
      if (d.act[k] == 1) z = fmaxf(z, 0.f);
      else if (d.act[k] == 2) z = sigmoidf(z);
      nxt[o] = z;
      ab[off + o] = z;
    }
    __syncthreads();
    for (int o = t; o < nout; o += 256) cur[o] = nxt[o];
    __syncthreads();
    off += nout;
  }
  const int C = d.C;
  if (d.final_mode == 1) {
    float mx = -3.402823466e38f;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, cur[c]);
    float sum = 0.f;
    for (int c = 0; c < C; ++c) sum += expf(cur[c] - mx);
    for (int c = t; c < C; c += 256) nxt[c] = expf(cur[c] - mx) / sum;
  } else if (d.final_mode == 2) {
    for (int c = t; c < C; c += 256) nxt[c] = cur[c] * mdv[c];
  } else {
    for (int c = t; c < C; c += 256) nxt[c] = cur[c];
  }
  for (int c = t; c < C; c += 256) {
    const float v = nxt[c];
    yfin[(long)b * C + c] = v;
    y[(long)b * C + c] = mul ? v * mul[(long)b * C + c] : v;
  }
}

// per sample: dz of every layer into ws [B][ZW] (ZW = sum of nout), d pool, d metadata, d mul
__global__ __launch_bounds__(256) void gate_mlp_bwd_sample_kernel(const float* __restrict__ dy, const float* __restrict__ md,
                                                                  const float* __restrict__ mul, GateMlp d,
                                                                  const float* __restrict__ acts,
                                                                  const float* __restrict__ yfin, float* __restrict__ ws,
                                                                  float* __restrict__ dpool, float* __restrict__ dmd,
                                                                  float* __restrict__ dmul) {
  __shared__ float dv[GM_MAXW], dz[GM_MAXW], mdv[GM_MAXW], dmda[GM_MAXW], red[2];
  const int b = blockIdx.x, t = threadIdx.x;
  const int AW = gm_acts_width(d), C = d.C;
  const int ZW = AW - d.nin[0];
  const float* ab = acts + (long)b * AW;
  for (int j = t; j < d.M; j += 256) {
    mdv[j] = md[(long)b * d.M + j];
    dmda[j] = 0.f;
  }
  for (int c = t; c < C; c += 256) {
    float g = dy[(long)b * C + c];
    if (mul) {
      if (dmul) dmul[(long)b * C + c] = g * yfin[(long)b * C + c];
      g *= mul[(long)b * C + c];
    }
    dv[c] = g;
  }
  __syncthreads();
  const int vL = AW - d.nout[d.L - 1];  // offset of the last layer's output in acts
  if (d.final_mode == 1) {
    if (t == 0) {
      float s = 0.f;
      for (int c = 0; c < C; ++c) s += dv[c] * yfin[(long)b * C + c];
      red[0] = s;
    }
    __syncthreads();
    for (int c = t; c < C; c += 256) dz[c] = yfin[(long)b * C + c] * (dv[c] - red[0]);
    __syncthreads();
    for (int c = t; c < C; c += 256) dv[c] = dz[c];
  } else if (d.final_mode == 2) {
    for (int c = t; c < C; c += 256) {
      dmda[c] += dv[c] * ab[vL + c];
      dv[c] = dv[c] * mdv[c];
    }
  }
  __syncthreads();
  int voff = vL, zoff = ZW;
  for (int k = d.L - 1; k >= 0; --k) {
    const int nin = d.nin[k], inw = nin + (d.cat[k] ? d.M : 0), nout = d.nout[k];
    zoff -= nout;
    const int inoff = voff - nin;  // offset of this layer's (previous layer's output) input in acts
    for (int o = t; o < nout; o += 256) {
      const float v = ab[voff + o];
      float g = dv[o];
      if (d.act[k] == 1) g = v > 0.f ? g : 0.f;
      else if (d.act[k] == 2) g = g * v * (1.f - v);
      dz[o] = g;
      ws[(long)b * ZW + zoff + o] = g;
    }
    __syncthreads();
    for (int j = t; j < inw; j += 256) {
      float g = 0.f;
      for (int o = 0; o < nout; ++o) g += d.w[k][(long)o * inw + j] * dz[o];
      const float raw = j < nin ? ab[inoff + j] : mdv[j - nin];
      if (d.relu_in[k] && !(raw > 0.f)) g = 0.f;
      if (j < nin) dv[j] = g;
      else dmda[j - nin] += g;
    }
    __syncthreads();
    voff = inoff;
  }
  for (int j = t; j < d.nin[0]; j += 256) dpool[(long)b * d.nin[0] + j] = dv[j];
  if (dmd)
    for (int j = t; j < d.M; j += 256) dmd[(long)b * d.M + j] = dmda[j];
}

// one thread per parameter element of every layer: sum over the batch (batch order) of dz (x) layer input
struct GateMlpGrads {
  float* dw[GM_MAXL];
  float* db[GM_MAXL];
};
__global__ __launch_bounds__(256) void gate_mlp_bwd_param_kernel(const float* __restrict__ md, GateMlp d,
                                                                 const float* __restrict__ acts,
                                                                 const float* __restrict__ ws, GateMlpGrads g, int B) {
  const int AW = gm_acts_width(d);
  const int ZW = AW - d.nin[0];
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  int voff = 0, zoff = 0;
  for (int k = 0; k < d.L; ++k) {
    const int nin = d.nin[k], inw = nin + (d.cat[k] ? d.M : 0), nout = d.nout[k];
    const long nw = (long)nout * inw;
    if (i < nw + nout) {
      float acc = 0.f;
      if (i < nw) {
        const int o = (int)(i / inw), j = (int)(i - (long)o * inw);
        for (int b = 0; b < B; ++b) {
          float v = j < nin ? acts[(long)b * AW + voff + j] : md[(long)b * d.M + (j - nin)];
          if (d.relu_in[k]) v = fmaxf(v, 0.f);
          acc += ws[(long)b * ZW + zoff + o] * v;
        }
        g.dw[k][i] = acc;
      } else {
        const int o = (int)(i - nw);
        for (int b = 0; b < B; ++b) acc += ws[(long)b * ZW + zoff + o];
        if (g.db[k]) g.db[k][o] = acc;
      }
      return;
    }
    i -= nw + nout;
    voff += nin;
    zoff += nout;
  }
}

// ---------------------------------------------------------------- gated residual: y = t*g[b,c] + shift[b,c] + x
// t, x, y: contiguous [B][HW][C]; g, shift: [B][C] (nullable -> 1, 0); x nullable.  C % 4 == 0.
__global__ __launch_bounds__(256) void gate_residual_fwd_kernel(const float* __restrict__ t, const float* __restrict__ g,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ x, float* __restrict__ y,
                                                                long hw, int C, long total4) {
  const int c4n = C >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const long pix = i / c4n;
    const int c4 = (int)(i - pix * c4n);
    const long b = pix / hw;
    f32x4 v = reinterpret_cast<const f32x4*>(t)[i];
    if (g && x && !shift) {  // the gated skip: product rounded, then the sum -- as the conv kernels' GATE prologue forms it
      v = sisr_mul_add4(v, *reinterpret_cast<const f32x4*>(g + b * C + c4 * 4), reinterpret_cast<const f32x4*>(x)[i]);
    } else {
      if (g) v = v * *reinterpret_cast<const f32x4*>(g + b * C + c4 * 4);
      if (shift) v = v + *reinterpret_cast<const f32x4*>(shift + b * C + c4 * 4);
      if (x) v = v + reinterpret_cast<const f32x4*>(x)[i];
    }
    reinterpret_cast<f32x4*>(y)[i] = v;
  }
}

// dg partials: part[b][k][c] = sum over pixel slice k of dy*t (t == nullptr: of dy -- the plain GAP)
// (C = 64, 256 threads = 16 px x 16 float4)
// Maps wider than 64 channels (gated blocks at n_feats = 128, 256, ...): blockIdx.z = 64-channel chunk of a pixel of C
// channels; part is [B][parts][C].  For C = 64 this is the kernel it always was.
__global__ __launch_bounds__(256) void gate_dg_partial_kernel(const float* __restrict__ dy, const float* __restrict__ t,
                                                              float* __restrict__ part, long hw, int parts, int C) {
  __shared__ f32x4 red[256];
  const int b = blockIdx.y, k = blockIdx.x, chunk = blockIdx.z;
  const int c4 = threadIdx.x & 15, pr = threadIdx.x >> 4;
  const int ps = C >> 2;  // float4 per pixel
  const long per = (hw + parts - 1) / parts;
  const long p0 = (long)k * per;
  const long p1 = p0 + per < hw ? p0 + per : hw;
  const f32x4* d4 = reinterpret_cast<const f32x4*>(dy) + (long)b * hw * ps + chunk * 16;
  const f32x4* t4 = reinterpret_cast<const f32x4*>(t) + (long)b * hw * ps + chunk * 16;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (t) {
    for (long p = p0 + pr; p < p1; p += 16) acc += d4[p * ps + c4] * t4[p * ps + c4];
  } else {
    for (long p = p0 + pr; p < p1; p += 16) acc += d4[p * ps + c4];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < 16) {
    f32x4 s = red[threadIdx.x];
    for (int r = 1; r < 16; ++r) s += red[r * 16 + threadIdx.x];
    reinterpret_cast<f32x4*>(part + ((long)b * parts + k) * C + chunk * 64)[threadIdx.x] = s;
  }
}

// out[b][c] = scale * sum_k part[b][k][c], C a multiple of 4: one workgroup per (64-channel chunk, b); 16 lanes cover the chunk with
// one float4 each, the 16 lane groups take k = r, r + 16, ... (coalesced 256-B rows, 16 loads in flight per wave instead of
// one thread walking all `parts` rows), and the 16 group sums are added in index order: the result does not depend on timing.
__global__ __launch_bounds__(256) void sum_partials_c4_kernel(const float* __restrict__ part, int parts, int C, float scale,
                                                              float* __restrict__ out) {
  __shared__ f32x4 red[256];
  const int lane = threadIdx.x & 15, r = threadIdx.x >> 4, b = blockIdx.y;
  const int c = blockIdx.x * 64 + lane * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c < C)
    for (int k = r; k < parts; k += 16) s += *reinterpret_cast<const f32x4*>(part + ((long)b * parts + k) * C + c);
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < 16 && c < C) {
    f32x4 t = red[threadIdx.x];
    for (int q = 1; q < 16; ++q) t += red[q * 16 + threadIdx.x];
    *reinterpret_cast<f32x4*>(out + (long)b * C + c) = t * scale;
  }
}

// the same for any C (one thread per output, rows in order)
__global__ void sum_partials_kernel(const float* __restrict__ part, int parts, int C, float scale,
                                    float* __restrict__ out, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long b = i / C;
  const int c = (int)(i - b * C);
  float s = 0.f;
  for (int k = 0; k < parts; ++k) s += part[((long)b * parts + k) * C + c];
  out[i] = s * scale;
}

// ---------------------------------------------------------------- pixel attention (PALayer), C = 64, hidden = 8
// ref: attention_manipulators/architectures.py:13-26:  y = x * sigmoid(w2 . relu(W1 x + b1) + b2) per pixel.
// 16 lanes per pixel, a lane owns 4 channels and the matching 8x4 slice of W1 in registers; the 8 hidden sums
// are completed with four xor-shuffles.  HBM-bound (one read + one write of the map); the backward pass
// recomputes the gate instead of storing it and emits ordered per-block partial sums of the parameter grads.
#define PA_H 8
#define PA_NP (PA_H * 64 + PA_H + PA_H + 1)  // dW1[8][64], db1[8], dw2[8], db2

template <bool BWD>
__global__ __launch_bounds__(256) void pa_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                 const float* __restrict__ b1, const float* __restrict__ w2,
                                                 const float* __restrict__ b2, const float* __restrict__ dy,
                                                 float* __restrict__ out, float* __restrict__ part, long npix) {
  __shared__ float red[16][PA_NP];
  const int c4 = threadIdx.x & 15, grp = threadIdx.x >> 4;
  f32x4 wr[PA_H];
  float bb1[PA_H], ww2[PA_H];
#pragma unroll
  for (int j = 0; j < PA_H; ++j) {
    wr[j] = *reinterpret_cast<const f32x4*>(w1 + j * 64 + c4 * 4);
    bb1[j] = b1[j];
    ww2[j] = w2[j];
  }
  const float bb2 = b2[0];
  f32x4 aw1[PA_H];
  float ab1[PA_H], aw2[PA_H], ab2 = 0.f;
  if (BWD) {
#pragma unroll
    for (int j = 0; j < PA_H; ++j) {
      aw1[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      ab1[j] = aw2[j] = 0.f;
    }
  }
  const long pend = (npix + 15) & ~15L;
  for (long p0 = (long)blockIdx.x * 16 + grp; p0 < pend; p0 += (long)gridDim.x * 16) {
    const bool live = p0 < npix;
    const long pix = live ? p0 : npix - 1;
    const f32x4 xv = reinterpret_cast<const f32x4*>(x)[pix * 16 + c4];
    float h[PA_H];
#pragma unroll
    for (int j = 0; j < PA_H; ++j) {
      const f32x4 t = wr[j] * xv;
      h[j] = (t[0] + t[1]) + (t[2] + t[3]);
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1)
#pragma unroll
      for (int j = 0; j < PA_H; ++j) h[j] += __shfl_xor(h[j], o);
    float z = bb2, a[PA_H];
#pragma unroll
    for (int j = 0; j < PA_H; ++j) {
      a[j] = fmaxf(h[j] + bb1[j], 0.f);
      z += ww2[j] * a[j];
    }
    const float g = 1.f / (1.f + expf(-z));
    if (!BWD) {
      if (live) reinterpret_cast<f32x4*>(out)[pix * 16 + c4] = xv * g;
    } else {
      const f32x4 dv = reinterpret_cast<const f32x4*>(dy)[pix * 16 + c4];
      const f32x4 t = dv * xv;
      float dot = (t[0] + t[1]) + (t[2] + t[3]);
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) dot += __shfl_xor(dot, o);
      const float dz = live ? dot * g * (1.f - g) : 0.f;
      f32x4 dx = dv * g;
#pragma unroll
      for (int j = 0; j < PA_H; ++j) {
        const float da = a[j] > 0.f ? dz * ww2[j] : 0.f;
        dx += wr[j] * da;
        aw1[j] += xv * da;
        if (c4 == 0) {
          ab1[j] += da;
          aw2[j] += dz * a[j];
        }
      }
      if (c4 == 0) ab2 += dz;
      if (live) reinterpret_cast<f32x4*>(out)[pix * 16 + c4] = dx;
    }
  }
  if (BWD) {
    // ordered reduction over the 16 pixel groups of the block: red[grp][element]
#pragma unroll
    for (int j = 0; j < PA_H; ++j) {
#pragma unroll
      for (int e = 0; e < 4; ++e) red[grp][j * 64 + c4 * 4 + e] = aw1[j][e];
      if (c4 == 0) {
        red[grp][PA_H * 64 + j] = ab1[j];
        red[grp][PA_H * 64 + PA_H + j] = aw2[j];
      }
    }
    if (c4 == 0) red[grp][PA_NP - 1] = ab2;
    __syncthreads();
    for (int e = threadIdx.x; e < PA_NP; e += 256) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) s += red[k][e];
      part[(long)blockIdx.x * PA_NP + e] = s;
    }
  }
}

__global__ void pa_reduce_kernel(const float* __restrict__ part, int nblocks, float* __restrict__ dw1,
                                 float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= PA_NP) return;
  float s = 0.f;
  for (int k = 0; k < nblocks; ++k) s += part[(long)k * PA_NP + e];
  if (e < PA_H * 64) dw1[e] = s;
  else if (e < PA_H * 64 + PA_H) db1[e - PA_H * 64] = s;
  else if (e < PA_H * 64 + 2 * PA_H) dw2[e - PA_H * 64 - PA_H] = s;
  else db2[0] = s;
}

static int pa_blocks(long npix) {
  long nb = (npix + 15) / 16;
  if (nb > 1024) nb = 1024;
  return nb < 1 ? 1 : (int)nb;
}

// ---------------------------------------------------------------- C ABI
extern "C" int sisr_ca_gate_fwd(const float* gap_partial, int parts, int B, float inv_hw, const float* w1,
                                const float* b1, const float* w2, const float* b2, int channels, int hidden,
                                const float* mul, float* s, float* hid, float* ca, float* g, void* stream) {
  if (!gap_partial || !w1 || !b1 || !w2 || !b2 || !s || !hid || !ca || !g || B <= 0 || parts <= 0) return SISR_ERR_ARG;
  if (channels != 64 || hidden < 1 || hidden > 16) return SISR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(ca_gate_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, gap_partial, parts, inv_hw, w1, b1,
                     w2, b2, hidden, mul, s, hid, ca, g);
  return sisr_check_launch();
}

extern "C" size_t sisr_ca_gate_bwd_workspace_bytes(int B) { return B > 0 ? (size_t)B * 80 * sizeof(float) : 0; }

extern "C" int sisr_ca_gate_bwd(const float* dg_partial, int parts, int B, float inv_hw, const float* w1,
                                const float* w2, int channels, int hidden, const float* s, const float* hid,
                                const float* ca, const float* mul, float* shift, float* dmul, float* dw1, float* db1,
                                float* dw2, float* db2, float* workspace, unsigned* counter, void* stream) {
  const bool defer = !dw1 && !db1 && !dw2 && !db2;  // parameter gradients later, by sisr_ca_gate_bwd_params_batch
  if (!dg_partial || !w1 || !w2 || !s || !hid || !ca || !shift || !workspace || B <= 0 || parts <= 0) return SISR_ERR_ARG;
  if (!defer && (!dw1 || !db1 || !dw2 || !db2 || !counter)) return SISR_ERR_ARG;
  if (mul && !dmul) return SISR_ERR_ARG;
  if (channels != 64 || hidden < 1 || hidden > 16) return SISR_ERR_UNSUPPORTED;
  float* dz2 = workspace;  // [B][80] rows: dz2 | dz1 (ca_gate.h)
  float* dz1 = workspace + 64;
  hipLaunchKernelGGL(ca_gate_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dg_partial, parts, inv_hw, w1, w2,
                     hidden, s, hid, ca, mul, shift, dmul, dz2, dz1, dw1, db1, dw2, db2, counter, B);
  return sisr_check_launch();
}

extern "C" int sisr_ca_gate_bwd_params_batch_max(void) { return CA_PB; }
extern "C" size_t sisr_ca_param_job_bytes(void) { return sizeof(CaParamJob); }
// jobs: HOST array of njobs records { workspace of that gate's sisr_ca_gate_bwd call, hid, s, dw1, db1, dw2, db2 }
extern "C" int sisr_ca_gate_bwd_params_batch(const void* jobs_host, int njobs, int B, int hidden, void* stream) {
  if (!jobs_host || njobs <= 0 || njobs > CA_PB || B <= 0 || hidden < 1 || hidden > 16) return SISR_ERR_ARG;
  CaParamBatch bt;
  memset(&bt, 0, sizeof(bt));
  const CaParamJob* jobs = static_cast<const CaParamJob*>(jobs_host);
  for (int k = 0; k < njobs; ++k) {
    if (!jobs[k].dz || !jobs[k].hid || !jobs[k].s || !jobs[k].dw1 || !jobs[k].db1 || !jobs[k].dw2 || !jobs[k].db2) return SISR_ERR_ARG;
    bt.job[k] = jobs[k];
  }
  hipLaunchKernelGGL(ca_gate_bwd_params_batch_kernel, dim3(njobs), dim3(256), 0, (hipStream_t)stream, bt, B, hidden);
  return sisr_check_launch();
}

extern "C" int sisr_meta_gate_fwd(const float* md, int B, int M, int hidden, int channels, const float* v1,
                                  const float* c1, const float* v2, const float* c2, int relu, float* hid, float* m,
                                  void* stream) {
  if (!md || !v1 || !c1 || !v2 || !c2 || !hid || !m || B <= 0 || M <= 0 || hidden <= 0 || channels <= 0) return SISR_ERR_ARG;
  if (hidden > 4096) return SISR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(meta_gate_fwd_kernel, dim3(B), dim3(256), hidden * sizeof(float), (hipStream_t)stream, md, M, hidden,
                     channels, v1, c1, v2, c2, relu, hid, m);
  return sisr_check_launch();
}

extern "C" size_t sisr_meta_gate_bwd_workspace_bytes(int B, int hidden, int channels) {
  return (B > 0 && hidden > 0 && channels > 0) ? (size_t)B * (hidden + channels) * sizeof(float) : 0;
}

extern "C" int sisr_meta_gate_bwd(const float* dm, const float* m, const float* hid, const float* md, int B, int M,
                                  int hidden, int channels, const float* v1, const float* v2, int relu, float* dv1,
                                  float* dc1, float* dv2, float* dc2, float* dmd, float* workspace, void* stream) {
  if (!dm || !m || !hid || !md || !v1 || !v2 || !dv1 || !dc1 || !dv2 || !dc2 || !workspace || B <= 0) return SISR_ERR_ARG;
  const size_t lds = ((size_t)hidden + channels + (size_t)channels * hidden + (size_t)hidden * M) * sizeof(float);
  if (lds > 60000) return SISR_ERR_UNSUPPORTED;
  float* dz2 = workspace;
  float* dz1 = workspace + (size_t)B * channels;
  hipLaunchKernelGGL(meta_gate_bwd_sample_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, dm, m, hid, M, hidden,
                     channels, v1, v2, relu, dz2, dz1, dmd);
  int rc = sisr_check_launch();
  if (rc) return rc;
  const long total = (long)channels * hidden + (long)hidden * M + channels + hidden;
  hipLaunchKernelGGL(meta_gate_bwd_param_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     dz2, dz1, hid, md, B, M, hidden, channels, dv1, dc1, dv2, dc2);
  return sisr_check_launch();
}

extern "C" int sisr_meta_gate_many_fwd(const float* md, int B, int M, int hidden, int channels, int layers,
                                       const float* const* v1_table, const float* const* c1_table,
                                       const float* const* v2_table, const float* const* c2_table, int relu, float* hid,
                                       float* m, void* stream) {
  if (!md || !v1_table || !c1_table || !v2_table || !c2_table || !hid || !m || B <= 0 || M <= 0 || hidden <= 0 ||
      channels <= 0 || layers <= 0)
    return SISR_ERR_ARG;
  if (hidden > 4096 || layers > 65535) return SISR_ERR_UNSUPPORTED;
  const MetaTables t = {v1_table, c1_table, v2_table, c2_table};
  hipLaunchKernelGGL(meta_gate_many_fwd_kernel, dim3(B, layers), dim3(256), hidden * sizeof(float), (hipStream_t)stream,
                     md, B, M, hidden, channels, t, relu, hid, m);
  return sisr_check_launch();
}

extern "C" size_t sisr_meta_gate_many_bwd_workspace_bytes(int B, int hidden, int channels, int layers) {
  return (B > 0 && hidden > 0 && channels > 0 && layers > 0)
             ? (size_t)layers * B * (hidden + channels) * sizeof(float)
             : 0;
}

extern "C" int sisr_meta_gate_many_bwd(const float* dm, const float* m, const float* hid, const float* md, int B, int M,
                                       int hidden, int channels, int layers, const float* const* v1_table,
                                       const float* const* v2_table, int relu, float* dv1, float* dc1, float* dv2,
                                       float* dc2, float* workspace, void* stream) {
  if (!dm || !m || !hid || !md || !v1_table || !v2_table || !dv1 || !dc1 || !dv2 || !dc2 || !workspace || B <= 0 ||
      layers <= 0)
    return SISR_ERR_ARG;
  const size_t lds = ((size_t)hidden + channels + (size_t)channels * hidden + (size_t)hidden * M) * sizeof(float);
  if (lds > 60000 || layers > 65535) return SISR_ERR_UNSUPPORTED;
  float* dz2 = workspace;
  float* dz1 = workspace + (size_t)layers * B * channels;
  const MetaTables t = {v1_table, nullptr, v2_table, nullptr};
  hipLaunchKernelGGL(meta_gate_many_bwd_sample_kernel, dim3(B, layers), dim3(256), lds, (hipStream_t)stream, dm, m, hid, B,
                     M, hidden, channels, t, relu, dz2, dz1);
  int rc = sisr_check_launch();
  if (rc) return rc;
  const long total = (long)channels * hidden + (long)hidden * M + channels + hidden;
  hipLaunchKernelGGL(meta_gate_many_bwd_param_kernel, dim3((unsigned)((total + 255) / 256), layers), dim3(256), 0,
                     (hipStream_t)stream, dz2, dz1, hid, md, B, M, hidden, channels, dv1, dc1, dv2, dc2);
  return sisr_check_launch();
}

extern "C" int sisr_meta_gate_many_bwd_scatter(const float* dm, const float* m, const float* hid, const float* md, int B,
                                               int M, int hidden, int channels, int layers, const float* const* v1_table,
                                               const float* const* v2_table, int relu, float* const* dv1_table,
                                               float* const* dc1_table, float* const* dv2_table, float* const* dc2_table,
                                               float* workspace, void* stream) {
  if (!dm || !m || !hid || !md || !v1_table || !v2_table || !dv1_table || !dc1_table || !dv2_table || !dc2_table ||
      !workspace || B <= 0 || layers <= 0)
    return SISR_ERR_ARG;
  const size_t lds = ((size_t)hidden + channels + (size_t)channels * hidden + (size_t)hidden * M) * sizeof(float);
  if (lds > 60000 || layers > 65535) return SISR_ERR_UNSUPPORTED;
  float* dz2 = workspace;
  float* dz1 = workspace + (size_t)layers * B * channels;
  const MetaTables t = {v1_table, nullptr, v2_table, nullptr};
  hipLaunchKernelGGL(meta_gate_many_bwd_sample_kernel, dim3(B, layers), dim3(256), lds, (hipStream_t)stream, dm, m, hid, B,
                     M, hidden, channels, t, relu, dz2, dz1);
  int rc = sisr_check_launch();
  if (rc) return rc;
  const long total = (long)channels * hidden + (long)hidden * M + channels + hidden;
  const MetaGradTables g = {dv1_table, dc1_table, dv2_table, dc2_table};
  hipLaunchKernelGGL(meta_gate_many_bwd_param_scatter_kernel, dim3((unsigned)((total + 255) / 256), layers), dim3(256), 0,
                     (hipStream_t)stream, dz2, dz1, hid, md, B, M, hidden, channels, g);
  return sisr_check_launch();
}

// ---- generic gate MLP.  `desc` is a HOST pointer to the struct below (copied into the launch); all arrays device.
struct sisr_gate_mlp_host {
  const float* w[GM_MAXL];
  const float* b[GM_MAXL];
  int nin[GM_MAXL], nout[GM_MAXL], cat[GM_MAXL], relu_in[GM_MAXL], act[GM_MAXL];
  int L, M, C, final_mode;
};
static int gate_mlp_check(const sisr_gate_mlp_host* h, GateMlp* d) {
  if (!h || h->L < 1 || h->L > GM_MAXL || h->M < 0 || h->C < 1) return SISR_ERR_ARG;
  memcpy(d, h, sizeof(GateMlp));
  int prev = h->nin[0];
  for (int k = 0; k < h->L; ++k) {
    if (!h->w[k] || h->nin[k] != prev || h->nout[k] < 1) return SISR_ERR_ARG;
    if (h->nin[k] + (h->cat[k] ? h->M : 0) > GM_MAXW || h->nout[k] > GM_MAXW) return SISR_ERR_UNSUPPORTED;
    prev = h->nout[k];
  }
  if (prev != h->C || h->M > GM_MAXW) return SISR_ERR_UNSUPPORTED;
  if (h->final_mode == 2 && h->M != h->C) return SISR_ERR_UNSUPPORTED;
  return SISR_OK;
}
extern "C" size_t sisr_gate_mlp_desc_bytes() { return sizeof(sisr_gate_mlp_host); }
extern "C" int sisr_gate_mlp_fwd(const float* pool, const float* md, const float* mul, int B, const void* desc,
                                 float* acts, float* yfin, float* y, void* stream) {
  GateMlp d;
  int rc = gate_mlp_check(static_cast<const sisr_gate_mlp_host*>(desc), &d);
  if (rc) return rc;
  if (!pool || !acts || !yfin || !y || B <= 0 || (d.M > 0 && !md)) return SISR_ERR_ARG;
  hipLaunchKernelGGL(gate_mlp_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, pool, md, mul, d, acts, yfin, y);
  return sisr_check_launch();
}
extern "C" int sisr_gate_mlp_bwd(const float* dy, const float* md, const float* mul, int B, const void* desc,
                                 const float* acts, const float* yfin, float* workspace, float* dpool, float* dmd,
                                 float* dmul, float* const* dw, float* const* db, void* stream) {
  GateMlp d;
  int rc = gate_mlp_check(static_cast<const sisr_gate_mlp_host*>(desc), &d);
  if (rc) return rc;
  if (!dy || !acts || !yfin || !workspace || !dpool || !dw || !db || B <= 0) return SISR_ERR_ARG;
  hipLaunchKernelGGL(gate_mlp_bwd_sample_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dy, md, mul, d, acts, yfin,
                     workspace, dpool, dmd, dmul);
  rc = sisr_check_launch();
  if (rc) return rc;
  GateMlpGrads g;
  long total = 0;
  for (int k = 0; k < GM_MAXL; ++k) {
    g.dw[k] = k < d.L ? dw[k] : nullptr;
    g.db[k] = k < d.L ? db[k] : nullptr;
    if (k < d.L) {
      if (!dw[k]) return SISR_ERR_ARG;
      total += (long)d.nout[k] * (d.nin[k] + (d.cat[k] ? d.M : 0)) + d.nout[k];
    }
  }
  hipLaunchKernelGGL(gate_mlp_bwd_param_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     md, d, acts, workspace, g, B);
  return sisr_check_launch();
}

extern "C" int sisr_gate_residual_fwd(const float* t, const float* g, const float* shift, const float* x, float* y,
                                      int B, long hw, int channels, void* stream) {
  if (!t || !y || B <= 0 || hw <= 0 || channels <= 0 || (channels & 3)) return SISR_ERR_ARG;
  if (!sisr_aligned16(t) || !sisr_aligned16(g) || !sisr_aligned16(shift) || !sisr_aligned16(x) || !sisr_aligned16(y))
    return SISR_ERR_ALIGN;
  const long total4 = (long)B * hw * (channels >> 2);
  long blocks = (total4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gate_residual_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t, g, shift, x, y,
                     hw, channels, total4);
  return sisr_check_launch();
}

extern "C" int sisr_gate_dg_parts(long hw) {
  long p = (hw + 511) / 512;  // >= 512 pixels per block
  if (p > 128) p = 128;
  if (p < 1) p = 1;
  return (int)p;
}

extern "C" int sisr_gate_dg_partial(const float* dy, const float* t, float* part, int B, long hw, int channels,
                                    void* stream) {
  if (!dy || !part || B <= 0 || hw <= 0) return SISR_ERR_ARG;  /* t may be NULL: plain pixel sums */
  if (channels <= 0 || (channels & 63) || channels > 64 * 64) return SISR_ERR_UNSUPPORTED;
  if (!sisr_aligned16(dy) || !sisr_aligned16(t) || !sisr_aligned16(part)) return SISR_ERR_ALIGN;
  const int parts = sisr_gate_dg_parts(hw);
  hipLaunchKernelGGL(gate_dg_partial_kernel, dim3(parts, B, channels / 64), dim3(256), 0, (hipStream_t)stream, dy, t, part, hw,
                     parts, channels);
  return sisr_check_launch();
}

extern "C" int sisr_sum_partials(const float* part, int parts, int B, int channels, float scale, float* out,
                                 void* stream) {
  if (!part || !out || parts <= 0 || B <= 0 || channels <= 0) return SISR_ERR_ARG;
  if (!(channels & 3) && sisr_aligned16(part) && sisr_aligned16(out) && B <= 65535) {
    hipLaunchKernelGGL(sum_partials_c4_kernel, dim3((unsigned)((channels + 63) / 64), (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, part, parts, channels, scale, out);
    return sisr_check_launch();
  }
  const long total = (long)B * channels;
  hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part,
                     parts, channels, scale, out, total);
  return sisr_check_launch();
}

extern "C" int sisr_pa_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y,
                           long npix, int channels, int hidden, void* stream) {
  if (!x || !w1 || !b1 || !w2 || !b2 || !y || npix <= 0) return SISR_ERR_ARG;
  if (channels != 64 || hidden != PA_H) return SISR_ERR_UNSUPPORTED;
  if (!sisr_aligned16(x) || !sisr_aligned16(y) || !sisr_aligned16(w1)) return SISR_ERR_ALIGN;
  hipLaunchKernelGGL(pa_kernel<false>, dim3(pa_blocks(npix)), dim3(256), 0, (hipStream_t)stream, x, w1, b1, w2, b2,
                     (const float*)nullptr, y, (float*)nullptr, npix);
  return sisr_check_launch();
}

extern "C" size_t sisr_pa_bwd_workspace_bytes(long npix) {
  return npix > 0 ? (size_t)pa_blocks(npix) * PA_NP * sizeof(float) : 0;
}

extern "C" int sisr_pa_bwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                           const float* dy, float* dx, float* dw1, float* db1, float* dw2, float* db2, float* workspace,
                           long npix, int channels, int hidden, void* stream) {
  if (!x || !w1 || !b1 || !w2 || !b2 || !dy || !dx || !dw1 || !db1 || !dw2 || !db2 || !workspace || npix <= 0)
    return SISR_ERR_ARG;
  if (channels != 64 || hidden != PA_H) return SISR_ERR_UNSUPPORTED;
  if (!sisr_aligned16(x) || !sisr_aligned16(dy) || !sisr_aligned16(dx) || !sisr_aligned16(w1)) return SISR_ERR_ALIGN;
  const int nb = pa_blocks(npix);
  hipLaunchKernelGGL(pa_kernel<true>, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, w1, b1, w2, b2, dy, dx, workspace,
                     npix);
  int rc = sisr_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(pa_reduce_kernel, dim3((PA_NP + 255) / 256), dim3(256), 0, (hipStream_t)stream, workspace, nb, dw1,
                     db1, dw2, db2);
  return sisr_check_launch();
}
