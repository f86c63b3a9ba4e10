// Channel-attention gate arithmetic shared by the stand-alone gate kernels (attention.hip) and by the conv kernels'
// last-arriving-workgroup tails (conv3x3_mfma.hip): one code path, one summation order, identical bits either way.
//   forward  (ref: advanced/architectures.py:13-32):  s = mean_hw(t); h = relu(W1 s + b1); ca = sigmoid(W2 h + b2); g = ca [* mul]
//   backward: see ca_gate_bwd_sample / ca_gate_bwd_params
#pragma once
#include "sisr_common.h"

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float sigmoidf(float z) { return 1.f / (1.f + expf(-z)); }

// Sum `parts` rows of 64 floats with the whole 256-thread block: thread (c4 = t & 15, group = t >> 4) adds rows
// group, group + 16, ... of its float4 column, up to sixteen 16-byte loads in flight at once (the kernels that call
// this sit on the serial chain between two convs and are pure latency: 256 rows used to be eight dependent load
// rounds); the sixteen group sums are then added in group order.  Result valid in threads 0..63 (channel = t).
// `red` must hold 16 * 64 floats.
__device__ __forceinline__ float block_sum_parts(const float* __restrict__ pp, int parts, float* red) {
  const int c4 = threadIdx.x & 15, grp = threadIdx.x >> 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int k = grp;
  for (; k + 15 * 16 < parts; k += 16 * 16) {
    f32x4 t[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) t[u] = *reinterpret_cast<const f32x4*>(pp + (long)(k + 16 * u) * 64 + c4 * 4);
#pragma unroll
    for (int u = 0; u < 16; ++u) s += t[u];
  }
  for (; k < parts; k += 16) s += *reinterpret_cast<const f32x4*>(pp + (long)k * 64 + c4 * 4);
  *reinterpret_cast<f32x4*>(red + grp * 64 + c4 * 4) = s;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x < 64) {
    r = red[threadIdx.x];
#pragma unroll
    for (int g = 1; g < 16; ++g) r += red[g * 64 + threadIdx.x];
  }
  return r;
}


// ---- forward for ONE sample b, executed by a whole 256-thread workgroup.  red: 16 * 64 floats of LDS.
// part: that sample's [parts][64] partial sums.  NT: read the partials with cache-bypassing loads (they were written by
// other workgroups of the SAME launch).
template <bool NT>
__device__ __forceinline__ float block_sum_parts_t(const float* __restrict__ pp, int parts, float* red) {
  const int c4 = threadIdx.x & 15, grp = threadIdx.x >> 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int k = grp;
  for (; k + 15 * 16 < parts; k += 16 * 16) {
    f32x4 t[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const f32x4* q = reinterpret_cast<const f32x4*>(pp + (long)(k + 16 * u) * 64 + c4 * 4);
      t[u] = NT ? __builtin_nontemporal_load(q) : *q;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) s += t[u];
  }
  for (; k < parts; k += 16) {
    const f32x4* q = reinterpret_cast<const f32x4*>(pp + (long)k * 64 + c4 * 4);
    s += NT ? __builtin_nontemporal_load(q) : *q;
  }
  *reinterpret_cast<f32x4*>(red + grp * 64 + c4 * 4) = s;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x < 64) {
    r = red[threadIdx.x];
#pragma unroll
    for (int g = 1; g < 16; ++g) r += red[g * 64 + threadIdx.x];
  }
  return r;
}

// Four wave sums at once: the same six exchange steps per value, interleaved (a wave sum is six dependent cross-lane
// exchanges; the gate's hidden units are independent of each other).  Bit-identical to four calls of wave_sum.
__device__ __forceinline__ void wave_sum4(float (&v)[4]) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = __shfl_xor(v[j], o);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += t[j];
  }
}

// g_lds (optional, 64 floats of LDS outside `red`): the gate is left there too, for a caller that consumes it in the same
// launch (the conv kernels' gate heads: no store -> load round trip through L2 on the serial path).
template <bool NT>
__device__ __forceinline__ void ca_gate_fwd_sample(const float* __restrict__ part, int parts, float inv_hw, int b,
                                                   const float* __restrict__ w1, const float* __restrict__ b1,
                                                   const float* __restrict__ w2, const float* __restrict__ b2, int R,
                                                   const float* __restrict__ mul, float* __restrict__ s_out,
                                                   float* __restrict__ hid_out, float* __restrict__ ca_out,
                                                   float* __restrict__ g_out, float* red, float* g_lds = nullptr) {
  const int c = threadIdx.x & 63;
  // the MLP's operands are requested BEFORE the partial sums (they do not depend on them): one memory round trip on the
  // serial path between two convs instead of two.  R <= 4 (64 channels / reduction 16) takes this path; any other R the loop
  const bool pre = R <= 4 && threadIdx.x < 64;
  float w1r[4] = {0.f, 0.f, 0.f, 0.f}, w2r[4] = {0.f, 0.f, 0.f, 0.f}, b1r[4] = {0.f, 0.f, 0.f, 0.f}, b2c = 0.f, mc = 1.f;
  if (pre) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < R) {
        w1r[j] = w1[j * 64 + c];
        w2r[j] = w2[c * R + j];
        b1r[j] = b1[j];
      }
    b2c = b2[c];
    if (mul) mc = mul[b * 64 + c];
  }
  float s = block_sum_parts_t<NT>(part + (long)b * parts * 64, parts, red);
  if (threadIdx.x >= 64) return;
  s *= inv_hw;
  s_out[b * 64 + c] = s;
  float z, ca;
  if (pre) {
    float h[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = w1r[j] * s;
    wave_sum4(h);
    z = b2c;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < R) {
        const float hj = fmaxf(h[j] + b1r[j], 0.f);
        if (c == 0) hid_out[b * R + j] = hj;
        z += w2r[j] * hj;
      }
    ca = sigmoidf(z);
    const float g = mul ? ca * mc : ca;
    ca_out[b * 64 + c] = ca;
    g_out[b * 64 + c] = g;
    if (g_lds) g_lds[c] = g;
    return;
  }
  z = b2[c];
  for (int j = 0; j < R; ++j) {
    float h = wave_sum(w1[j * 64 + c] * s) + b1[j];
    h = fmaxf(h, 0.f);
    if (c == 0) hid_out[b * R + j] = h;
    z += w2[c * R + j] * h;
  }
  ca = sigmoidf(z);
  ca_out[b * 64 + c] = ca;
  const float g = mul ? ca * mul[b * 64 + c] : ca;
  g_out[b * 64 + c] = g;
  if (g_lds) g_lds[c] = g;
}

// ---- backward, per sample (whole workgroup):  dg = sum of the partials of sum_hw dOut*t;  dca = dg*mul;
//   dz2 = dca*ca*(1-ca); dh = W2^T dz2; dz1 = dh*[hid>0]; ds = W1^T dz1  ->  shift[b][c] = ds*inv_hw (the GAP backward
//   broadcast, consumed as the dgrad / wgrad prologue shift), dmul = dg*ca, and dz2 / dz1 into the workspace: one row of
//   CA_WS_ROW = 80 floats per sample, dz2 in its first 64, dz1 (R <= 16) behind them -- rows, so that a launch over a range of
//   samples (a sample lane) writes a plain slice of the whole batch's workspace.  dz1_out = dz2_out + 64.
#define CA_WS_ROW 80
template <bool NT>
__device__ __forceinline__ void ca_gate_bwd_sample(const float* __restrict__ dgpart, int parts, float inv_hw, int b,
                                                   const float* __restrict__ w1, const float* __restrict__ w2, int R,
                                                   const float* __restrict__ hid, const float* __restrict__ ca_in,
                                                   const float* __restrict__ mul, float* __restrict__ shift,
                                                   float* __restrict__ dmul, float* dz2_out, float* dz1_out, float* red,
                                                   float* shift_lds = nullptr) {
  const int c = threadIdx.x & 63;
  // operands first, partial sums second (see ca_gate_fwd_sample)
  const bool pre = R <= 4 && threadIdx.x < 64;
  float w1r[4] = {0.f, 0.f, 0.f, 0.f}, w2r[4] = {0.f, 0.f, 0.f, 0.f}, hr[4] = {0.f, 0.f, 0.f, 0.f}, cac = 0.f, mc = 1.f;
  if (pre) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < R) {
        w1r[j] = w1[j * 64 + c];
        w2r[j] = w2[c * R + j];
        hr[j] = hid[b * R + j];
      }
    cac = ca_in[b * 64 + c];
    if (mul) mc = mul[b * 64 + c];
  }
  const float dg = block_sum_parts_t<NT>(dgpart + (long)b * parts * 64, parts, red);
  if (threadIdx.x < 64) {
    const float ca = pre ? cac : ca_in[b * 64 + c];
    float dca = dg;
    if (mul) {
      dmul[b * 64 + c] = dg * ca;
      dca = dg * (pre ? mc : mul[b * 64 + c]);
    }
    const float dz2 = dca * ca * (1.f - ca);
    dz2_out[b * CA_WS_ROW + c] = dz2;
    float ds = 0.f;
    if (pre) {
      float dh[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) dh[j] = w2r[j] * dz2;
      wave_sum4(dh);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < R) {
          const float dz1 = hr[j] > 0.f ? dh[j] : 0.f;
          if (c == 0) dz1_out[b * CA_WS_ROW + j] = dz1;
          ds += w1r[j] * dz1;
        }
    } else {
      for (int j = 0; j < R; ++j) {
        const float dh = wave_sum(w2[c * R + j] * dz2);
        const float dz1 = hid[b * R + j] > 0.f ? dh : 0.f;
        if (c == 0) dz1_out[b * CA_WS_ROW + j] = dz1;
        ds += w1[j * 64 + c] * dz1;
      }
    }
    shift[b * 64 + c] = ds * inv_hw;
    if (shift_lds) shift_lds[c] = ds * inv_hw;
  }
}

// ---- backward, parameter gradients: sums over the batch in batch order (whole workgroup; run by whichever workgroup
// finished last -- which one it is does not change the result).  dz2 / dz1 were written by other workgroups.
__device__ __forceinline__ void ca_gate_bwd_params(const float* dz2_out, const float* dz1_out, const float* __restrict__ hid,
                                                   const float* __restrict__ s_in, int R, int B, float* __restrict__ dw1,
                                                   float* __restrict__ db1, float* __restrict__ dw2,
                                                   float* __restrict__ db2) {
  const int n = 64 * R;
  // sum over the batch of a[bb * sa] * b[bb * sb] (b == nullptr: of a alone), in batch order; eight cache-bypassing loads in
  // flight (one at a time this loop is B dependent memory round trips on the serial dgrad chain: 30 us at B = 32)
  auto dot_b = [&](const float* a, int sa, const float* b, int sb) {
    float acc = 0.f;
    int bb = 0;
    for (; bb + 8 <= B; bb += 8) {
      float t[8], u[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        t[k] = __builtin_nontemporal_load(a + (long)(bb + k) * sa);
        u[k] = b ? b[(long)(bb + k) * sb] : 1.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc = b ? acc + t[k] * u[k] : acc + t[k];
    }
    for (; bb < B; ++bb) {
      const float t = __builtin_nontemporal_load(a + (long)bb * sa);
      acc = b ? acc + t * b[(long)bb * sb] : acc + t;
    }
    return acc;
  };
  for (int i = threadIdx.x; i < 2 * n + 64 + R; i += 256) {
    if (i < n) {  // dw2[c][j] = sum_b dz2[b][c] * hid[b][j]
      const int cc = i / R, j = i - cc * R;
      dw2[i] = dot_b(dz2_out + cc, CA_WS_ROW, hid + j, R);
    } else if (i < 2 * n) {  // dw1[j][c] = sum_b dz1[b][j] * s[b][c]
      const int k = i - n, j = k >> 6, cc = k & 63;
      dw1[k] = dot_b(dz1_out + j, CA_WS_ROW, s_in + cc, 64);
    } else if (i < 2 * n + 64) {
      db2[i - 2 * n] = dot_b(dz2_out + (i - 2 * n), CA_WS_ROW, nullptr, 0);
    } else {
      db1[i - 2 * n - 64] = dot_b(dz1_out + (i - 2 * n - 64), CA_WS_ROW, nullptr, 0);
    }
  }
}
