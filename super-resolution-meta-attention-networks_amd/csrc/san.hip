// SAN attention kernels for gfx950: second-order channel attention (covariance pooling + Newton-Schulz matrix
// square root) and the embedded-Gaussian non-local attention of Nonlocal_CA.
// ref: advanced/SAN_blocks.py:104-148 (_embedded_gaussian), :244-302 (SOCA), advanced/mpncov.py:12-112
// (Covpool / Sqrtm autograd Functions, whose hand-written backward formulas are followed here).
//
// None of this is on the FLOP-critical path (SAN spends > 97 % of its work in the same 3x3 convs as RCAN); the
// kernels are written to stream each 64-channel map once, keep all reductions ordered (bitwise reproducible) and
// never materialise the reference's M x M centering matrix or its N x N attention matrix.
#include "sisr_common.h"

extern "C" int sisr_sum_partials(const float* part, int parts, int B, int channels, float scale, float* out,
                                 void* stream);

#define SD 64         // channels of the pooled maps (SAN: n_feats = 64)
#define SLD 68        // LDS row stride of a 64x64 matrix (16-B aligned rows, A-operand reads 4 banks apart)
#define SMAT (SD * SLD)

// ------------------------------------------------------------------------------------------ covariance pooling
// cov[b] = (1/M) sum_p (x_p - mean)(x_p)^T     (= X I^ X^T with I^ = I/M - 11^T/M^2, mpncov.py:24-30)
// x: [B][M][64] (channels-last map), mean: [B][64].  Stage 1: one workgroup per (pixel slab, sample) accumulates a
// 64x64 partial, a thread owning a 4x4 tile; stage 2 (sisr_sum_partials) adds the slabs in order and scales.
#define COV_PIX 64  // pixels staged per LDS round

static inline int covpool_parts(long hw) {
  long p = (hw + 511) / 512;
  return (int)(p < 1 ? 1 : (p > 64 ? 64 : p));
}

__global__ __launch_bounds__(256) void covpool_partial_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                              float* __restrict__ part, long hw, int parts) {
  __shared__ __attribute__((aligned(16))) float xs[COV_PIX * SLD];
  const int b = blockIdx.y, s = blockIdx.x;
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const long per = ((hw + parts - 1) / parts + COV_PIX - 1) / COV_PIX * COV_PIX;
  const long p0 = (long)s * per, p1 = p0 + per < hw ? p0 + per : hw;
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + (long)b * SD + ty * 4);
  float acc[4][4] = {};
  const float* xb = x + (long)b * hw * SD;
  for (long p = p0; p < p1; p += COV_PIX) {
    __syncthreads();
    for (int i = threadIdx.x; i < COV_PIX * 16; i += 256) {  // 64 pixels x 16 float4, coalesced
      const int pp = i >> 4, c4 = i & 15;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (p + pp < p1) v = *reinterpret_cast<const f32x4*>(xb + (p + pp) * SD + c4 * 4);
      *reinterpret_cast<f32x4*>(xs + pp * SLD + c4 * 4) = v;
    }
    __syncthreads();
    const int n = (int)(p1 - p < COV_PIX ? p1 - p : COV_PIX);
    for (int pp = 0; pp < n; ++pp) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(xs + pp * SLD + ty * 4) - mu;
      const f32x4 v = *reinterpret_cast<const f32x4*>(xs + pp * SLD + tx * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * v[j];
    }
  }
  float* o = part + ((long)b * parts + s) * (SD * SD);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    *reinterpret_cast<f32x4*>(o + (ty * 4 + i) * SD + tx * 4) = (f32x4){acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
}

extern "C" size_t sisr_covpool_workspace_bytes(int B, long hw) {
  return (B > 0 && hw > 0) ? (size_t)B * covpool_parts(hw) * SD * SD * sizeof(float) : 0;
}

extern "C" int sisr_covpool_fwd(const float* x, const float* mean, float* cov, float* workspace, int B, long hw,
                                int channels, void* stream) {
  if (!x || !mean || !cov || !workspace || B <= 0 || hw <= 0) return SISR_ERR_ARG;
  if (channels != SD) return SISR_ERR_UNSUPPORTED;
  if (!sisr_aligned16(x) || !sisr_aligned16(mean) || !sisr_aligned16(workspace)) return SISR_ERR_ALIGN;
  const int parts = covpool_parts(hw);
  hipLaunchKernelGGL(covpool_partial_kernel, dim3(parts, B), dim3(256), 0, (hipStream_t)stream, x, mean, workspace, hw,
                     parts);
  int rc = sisr_check_launch();
  if (rc != SISR_OK) return rc;
  return sisr_sum_partials(workspace, parts, B, SD * SD, 1.0f / (float)hw, cov, stream);
}

// ------------------------------------------------------------------------------- 64x64 matrix algebra in LDS
// One workgroup (256 threads) per sample; a thread owns the 4x4 tile (rows 4*ty.., cols 4*tx..) of every matrix.
// mm: C = alpha * A B + beta * C, C distinct from A and B.  Elementwise steps touch only the thread's own tile,
// so the barrier at the top of mm is the only ordering needed between steps.
struct Tile {
  float v[4][4];
};

__device__ __forceinline__ void mm64(float* __restrict__ C, const float* __restrict__ A, const float* __restrict__ Bm,
                                     float alpha, float beta, int ty, int tx) {
  __syncthreads();
  float acc[4][4] = {};
  for (int k = 0; k < SD; k += 4) {
    f32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const f32x4*>(A + (ty * 4 + i) * SLD + k);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) b[kk] = *reinterpret_cast<const f32x4*>(Bm + (k + kk) * SLD + tx * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i][kk] * b[kk][j];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x4* c = reinterpret_cast<f32x4*>(C + (ty * 4 + i) * SLD + tx * 4);
    f32x4 r = {alpha * acc[i][0], alpha * acc[i][1], alpha * acc[i][2], alpha * acc[i][3]};
    if (beta != 0.f) r += beta * *c;
    *c = r;
  }
}

// M <- s * (3I - M) on the own tile
__device__ __forceinline__ void three_i_minus(float* M, float s, int ty, int tx) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* e = M + (ty * 4 + i) * SLD + tx * 4 + j;
      *e = s * (((ty * 4 + i) == (tx * 4 + j) ? 3.f : 0.f) - *e);
    }
}

__device__ __forceinline__ void tile_load(float* M, const float* __restrict__ g, float s, int ty, int tx) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    *reinterpret_cast<f32x4*>(M + (ty * 4 + i) * SLD + tx * 4) =
        s * *reinterpret_cast<const f32x4*>(g + (ty * 4 + i) * SD + tx * 4);
}

__device__ __forceinline__ void tile_store(float* __restrict__ g, const float* M, float s, int ty, int tx) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    *reinterpret_cast<f32x4*>(g + (ty * 4 + i) * SD + tx * 4) =
        s * *reinterpret_cast<const f32x4*>(M + (ty * 4 + i) * SLD + tx * 4);
}

__device__ __forceinline__ void tile_copy(float* D, const float* S, float s, int ty, int tx) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    *reinterpret_cast<f32x4*>(D + (ty * 4 + i) * SLD + tx * 4) =
        s * *reinterpret_cast<const f32x4*>(S + (ty * 4 + i) * SLD + tx * 4);
}

// ordered block sum of one value per thread (256 threads); every thread gets the result
__device__ __forceinline__ float block_sum(float v, float* red) {
  __syncthreads();
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}

// Saved state per sample: [4 floats: trace, 0, 0, 0] [Y_0 .. Y_{n-2}] [Z_0 .. Z_{n-2}] [last], matrices dense 64x64.
static inline size_t sqrtm_saved_floats(int iters) { return 4 + (size_t)(2 * (iters - 1) + 1) * SD * SD; }

// ref: advanced/mpncov.py:51-77 (forward), SAN_blocks.py:293-294 (column means of the result).
__global__ __launch_bounds__(256) void sqrtm_fwd_kernel(const float* __restrict__ cov, float* __restrict__ saved,
                                                        float* __restrict__ pooled, int iters, size_t per_sample) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *Y = lds, *Z = lds + SMAT, *T = lds + 2 * SMAT, *Yn = lds + 3 * SMAT, *Zn = lds + 4 * SMAT;
  float* red = lds + 5 * SMAT;
  const int b = blockIdx.x, ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const float* cb = cov + (long)b * SD * SD;
  float* sv = saved + (size_t)b * per_sample;
  const float tr = block_sum(threadIdx.x < SD ? cb[threadIdx.x * SD + threadIdx.x] : 0.f, red);
  const int k = iters - 1;  // number of stored (Y, Z) pairs
  float* svY = sv + 4;
  float* svZ = svY + (size_t)k * SD * SD;
  float* svL = svZ + (size_t)k * SD * SD;
  if (threadIdx.x == 0) *reinterpret_cast<f32x4*>(sv) = (f32x4){tr, 0.f, 0.f, 0.f};
  tile_load(T, cb, 1.f / tr, ty, tx);  // A = cov / trace
  tile_copy(Z, T, 1.f, ty, tx);
  three_i_minus(Z, 0.5f, ty, tx);       // Z_0 = ZY = (3I - A)/2
  mm64(Y, T, Z, 1.f, 0.f, ty, tx);      // Y_0 = A ZY
  tile_store(svY, Y, 1.f, ty, tx);
  tile_store(svZ, Z, 1.f, ty, tx);
  for (int i = 1; i < k; ++i) {
    mm64(T, Z, Y, 1.f, 0.f, ty, tx);
    three_i_minus(T, 0.5f, ty, tx);     // ZY = (3I - Z Y)/2
    mm64(Yn, Y, T, 1.f, 0.f, ty, tx);
    mm64(Zn, T, Z, 1.f, 0.f, ty, tx);
    float* t = Y; Y = Yn; Yn = t;
    t = Z; Z = Zn; Zn = t;
    tile_store(svY + (size_t)i * SD * SD, Y, 1.f, ty, tx);
    tile_store(svZ + (size_t)i * SD * SD, Z, 1.f, ty, tx);
  }
  mm64(T, Z, Y, 1.f, 0.f, ty, tx);
  three_i_minus(T, 1.f, ty, tx);
  mm64(Yn, Y, T, 0.5f, 0.f, ty, tx);    // last = Y (3I - Z Y) / 2
  tile_store(svL, Yn, 1.f, ty, tx);
  __syncthreads();
  if (threadIdx.x < SD) {               // pooled[j] = mean_i (sqrt(trace) * last[i][j])
    float s = 0.f;
    for (int i = 0; i < SD; ++i) s += Yn[i * SLD + threadIdx.x];
    pooled[(long)b * SD + threadIdx.x] = s * sqrtf(tr) * (1.f / SD);
  }
}

// ref: advanced/mpncov.py:78-112.  Input: dL/dpooled [B][64]; output: G + G^T with G = dL/dcov (the symmetrised
// form Covpool.backward consumes, mpncov.py:44).
__global__ __launch_bounds__(256) void sqrtm_bwd_kernel(const float* __restrict__ cov, const float* __restrict__ saved,
                                                        const float* __restrict__ dpooled, float* __restrict__ dsym,
                                                        int iters, size_t per_sample) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *BY = lds, *BZ = lds + SMAT, *E = lds + 2 * SMAT, *D1 = lds + 3 * SMAT, *D2 = lds + 4 * SMAT,
        *M1 = lds + 5 * SMAT, *M2 = lds + 6 * SMAT;
  float* red = lds + 7 * SMAT;
  const int b = blockIdx.x, ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const float* sv = saved + (size_t)b * per_sample;
  const float* cb = cov + (long)b * SD * SD;
  const int k = iters - 1;
  const float* svY = sv + 4;
  const float* svZ = svY + (size_t)k * SD * SD;
  const float* svL = svZ + (size_t)k * SD * SD;
  const float tr = sv[0], rt = sqrtf(tr);
  // gp = g * sqrt(trace), g[i][j] = dpooled[j] / 64;  aux = sum(g o last) / (2 sqrt(trace))
  const f32x4 gj = *reinterpret_cast<const f32x4*>(dpooled + (long)b * SD + tx * 4) * (1.f / SD);
  float part = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x4 l = *reinterpret_cast<const f32x4*>(svL + (ty * 4 + i) * SD + tx * 4);
    part += gj[0] * l[0] + gj[1] * l[1] + gj[2] * l[2] + gj[3] * l[3];
    *reinterpret_cast<f32x4*>(M1 + (ty * 4 + i) * SLD + tx * 4) = gj * rt;
  }
  const float aux = block_sum(part, red) / (2.f * rt);
  tile_load(BY, svY + (size_t)(k - 1) * SD * SD, 1.f, ty, tx);
  tile_load(BZ, svZ + (size_t)(k - 1) * SD * SD, 1.f, ty, tx);
  mm64(E, BY, BZ, 1.f, 0.f, ty, tx);
  three_i_minus(E, 1.f, ty, tx);
  mm64(D1, M1, E, 0.5f, 0.f, ty, tx);     // dy = (gp (3I - Y Z) - Z Y gp) / 2
  mm64(E, BZ, BY, 1.f, 0.f, ty, tx);
  mm64(D1, E, M1, -0.5f, 1.f, ty, tx);
  mm64(E, BY, M1, 1.f, 0.f, ty, tx);
  mm64(D2, E, BY, -0.5f, 0.f, ty, tx);    // dz = -Y gp Y / 2
  for (int i = k - 2; i >= 0; --i) {
    __syncthreads();  // all reads of BY / BZ by the previous products are done
    tile_load(BY, svY + (size_t)i * SD * SD, 1.f, ty, tx);
    tile_load(BZ, svZ + (size_t)i * SD * SD, 1.f, ty, tx);
    mm64(E, BY, BZ, 1.f, 0.f, ty, tx);
    three_i_minus(E, 1.f, ty, tx);          // yz = 3I - Y Z
    mm64(M1, D1, E, 0.5f, 0.f, ty, tx);     // dy' = (dy yz - Z dz Z - zy dy) / 2
    mm64(M2, E, D2, 0.5f, 0.f, ty, tx);     // dz' = (yz dz - Y dy Y - dz zy) / 2
    mm64(E, BZ, BY, 1.f, 0.f, ty, tx);      // zy
    mm64(M1, E, D1, -0.5f, 1.f, ty, tx);
    mm64(M2, D2, E, -0.5f, 1.f, ty, tx);
    mm64(E, BZ, D2, 1.f, 0.f, ty, tx);
    mm64(M1, E, BZ, -0.5f, 1.f, ty, tx);
    mm64(E, BY, D1, 1.f, 0.f, ty, tx);
    mm64(M2, E, BY, -0.5f, 1.f, ty, tx);
    float* t = D1; D1 = M1; M1 = t;
    t = D2; D2 = M2; M2 = t;
  }
  __syncthreads();
  tile_load(BY, cb, 1.f / tr, ty, tx);      // A
  tile_copy(E, BY, 1.f, ty, tx);
  three_i_minus(E, 1.f, ty, tx);
  mm64(M1, D1, E, 0.5f, 0.f, ty, tx);       // dn = (dy (3I - A) - dz - A dy) / 2
  mm64(M1, BY, D1, -0.5f, 1.f, ty, tx);
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x4* m = reinterpret_cast<f32x4*>(M1 + (ty * 4 + i) * SLD + tx * 4);
    const f32x4 r = *m - 0.5f * *reinterpret_cast<const f32x4*>(D2 + (ty * 4 + i) * SLD + tx * 4);
    *m = r;
    const f32x4 c = *reinterpret_cast<const f32x4*>(cb + (ty * 4 + i) * SD + tx * 4);
    dot += r[0] * c[0] + r[1] * c[1] + r[2] * c[2] + r[3] * c[3];
  }
  const float gaux = block_sum(dot, red);   // (barriers inside also publish M1)
  const float diag = aux - gaux / (tr * tr);
  float* o = dsym + (long)b * SD * SD;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = ty * 4 + i, c = tx * 4 + j;
      o[r * SD + c] = (M1[r * SLD + c] + M1[c * SLD + r]) / tr + (r == c ? 2.f * diag : 0.f);
    }
}

extern "C" size_t sisr_sqrtm_saved_bytes(int B, int dim, int iters) {
  return (B > 0 && dim == SD && iters >= 2) ? (size_t)B * sqrtm_saved_floats(iters) * sizeof(float) : 0;
}

extern "C" int sisr_sqrtm_fwd(const float* cov, float* saved, float* pooled, int B, int dim, int iters, void* stream) {
  if (!cov || !saved || !pooled || B <= 0) return SISR_ERR_ARG;
  if (dim != SD || iters < 2 || iters > 16) return SISR_ERR_UNSUPPORTED;
  if (!sisr_aligned16(cov) || !sisr_aligned16(saved)) return SISR_ERR_ALIGN;
  const size_t lb = (5 * SMAT + 256) * sizeof(float);
  SISR_ALLOW_LDS(sqrtm_fwd_kernel, lb);
  hipLaunchKernelGGL(sqrtm_fwd_kernel, dim3(B), dim3(256), lb, (hipStream_t)stream, cov, saved, pooled, iters,
                     sqrtm_saved_floats(iters));
  return sisr_check_launch();
}

extern "C" int sisr_sqrtm_bwd(const float* cov, const float* saved, const float* dpooled, float* dcov_sym, int B, int dim,
                              int iters, void* stream) {
  if (!cov || !saved || !dpooled || !dcov_sym || B <= 0) return SISR_ERR_ARG;
  if (dim != SD || iters < 2 || iters > 16) return SISR_ERR_UNSUPPORTED;
  if (!sisr_aligned16(cov) || !sisr_aligned16(saved) || !sisr_aligned16(dpooled)) return SISR_ERR_ALIGN;
  const size_t lb = (7 * SMAT + 256) * sizeof(float);
  SISR_ALLOW_LDS(sqrtm_bwd_kernel, lb);
  hipLaunchKernelGGL(sqrtm_bwd_kernel, dim3(B), dim3(256), lb, (hipStream_t)stream, cov, saved, dpooled, dcov_sym, iters,
                     sqrtm_saved_floats(iters));
  return sisr_check_launch();
}

// ------------------------------------------------------------------------------------ SOCA input gradient
// dx[p][c] = dy[p][c] * gate[c] + (1/M) sum_c' S[c][c'] (x[p][c'] - mean[c'])      S = G + G^T (symmetric)
// ref: `y_cov * x` (SAN_blocks.py:302) + Covpool.backward (mpncov.py:35-47: (G + G^T) X I^).
// Workgroup = 64 pixels per round x 4 waves; wave w produces channels 16w..16w+15 of its lane's pixel, S rows are
// wave-uniform LDS broadcasts, the pixel tile sits in LDS at stride 65 (lane-conflict-free column walks).
#define SB_ROUNDS 8

__global__ __launch_bounds__(256) void soca_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ gate,
                                                             const float* __restrict__ x, const float* __restrict__ mean,
                                                             const float* __restrict__ dsym, float* __restrict__ dx,
                                                             long hw) {
  __shared__ __attribute__((aligned(16))) float S[SD * SD];
  __shared__ float xs[64 * 65];
  __shared__ float shift[SD];
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float inv = 1.f / (float)hw;
  for (int i = threadIdx.x; i < SD * SD / 4; i += 256)
    *reinterpret_cast<f32x4*>(S + i * 4) = inv * *reinterpret_cast<const f32x4*>(dsym + (long)b * SD * SD + i * 4);
  __syncthreads();
  if (threadIdx.x < SD) {  // shift[c] = -(S mean)[c]
    float s = 0.f;
    for (int k = 0; k < SD; ++k) s += S[k * SD + threadIdx.x] * mean[(long)b * SD + k];
    shift[threadIdx.x] = -s;
  }
  f32x4 gt[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) gt[q] = *reinterpret_cast<const f32x4*>(gate + (long)b * SD + w * 16 + q * 4);
  const long base = (long)blockIdx.x * (64 * SB_ROUNDS);
  for (int r = 0; r < SB_ROUNDS; ++r) {
    const long p0 = base + r * 64;
    if (p0 >= hw) break;  // uniform
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
      const int pp = i >> 4, c4 = i & 15;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (p0 + pp < hw) v = *reinterpret_cast<const f32x4*>(x + ((long)b * hw + p0 + pp) * SD + c4 * 4);
      float* d = xs + pp * 65 + c4 * 4;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = *reinterpret_cast<const f32x4*>(shift + w * 16 + q * 4);
    for (int k = 0; k < SD; ++k) {
      const float xv = xs[lane * 65 + k];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] += xv * *reinterpret_cast<const f32x4*>(S + k * SD + w * 16 + q * 4);
    }
    const long p = p0 + lane;
    if (p < hw) {
      const long o = ((long)b * hw + p) * SD + w * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(dx + o + q * 4) = *reinterpret_cast<const f32x4*>(dy + o + q * 4) * gt[q] + acc[q];
    }
  }
}

extern "C" int sisr_soca_bwd_apply(const float* dy, const float* gate, const float* x, const float* mean,
                                   const float* dcov_sym, float* dx, int B, long hw, int channels, void* stream) {
  if (!dy || !gate || !x || !mean || !dcov_sym || !dx || B <= 0 || hw <= 0) return SISR_ERR_ARG;
  if (channels != SD) return SISR_ERR_UNSUPPORTED;
  if (!sisr_aligned16(dy) || !sisr_aligned16(gate) || !sisr_aligned16(x) || !sisr_aligned16(dcov_sym) ||
      !sisr_aligned16(dx))
    return SISR_ERR_ALIGN;
  const unsigned gx = (unsigned)((hw + 64 * SB_ROUNDS - 1) / (64 * SB_ROUNDS));
  hipLaunchKernelGGL(soca_bwd_apply_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, dy, gate, x, mean, dcov_sym,
                     dx, hw);
  return sisr_check_launch();
}

// ------------------------------------------------------------------------------------- non-local attention
// y_i = sum_j softmax_j(theta_i . phi_j) g_j     theta: [nb][nq][8], phi / g: [nb][nk][8]  (8 = n_feats / 8)
// ref: advanced/SAN_blocks.py:126-141 (f = theta^T phi; softmax over keys; y = f g).  Streaming softmax: a thread
// owns a query, keys pass through LDS in chunks (wave-uniform broadcasts), nothing of size nq x nk is stored;
// the log-sum-exp per query is kept for the backward pass, which recomputes the probabilities.
#define NLD 8
#define NL_KC 512  // keys (fwd, bwd_q) / queries (bwd_k) per LDS chunk

__device__ __forceinline__ float dot8(const float* a, f32x4 b0, f32x4 b1) {
  return a[0] * b0[0] + a[1] * b0[1] + a[2] * b0[2] + a[3] * b0[3] + a[4] * b1[0] + a[5] * b1[1] + a[6] * b1[2] +
         a[7] * b1[3];
}

__global__ __launch_bounds__(256) void nl_attn_fwd_kernel(const float* __restrict__ theta, const float* __restrict__ phi,
                                                          const float* __restrict__ g, float* __restrict__ y,
                                                          float* __restrict__ lse, int nq, int nk) {
  __shared__ __attribute__((aligned(16))) float ks[NL_KC * NLD];
  __shared__ __attribute__((aligned(16))) float vs[NL_KC * NLD];
  const int b = blockIdx.y;
  const int qi = blockIdx.x * 256 + threadIdx.x;
  const bool live = qi < nq;
  float q[NLD], acc[NLD] = {};
  {
    const float* tp = theta + ((long)b * nq + (live ? qi : 0)) * NLD;
    const f32x4 a = *reinterpret_cast<const f32x4*>(tp), c = *reinterpret_cast<const f32x4*>(tp + 4);
    q[0] = a[0]; q[1] = a[1]; q[2] = a[2]; q[3] = a[3]; q[4] = c[0]; q[5] = c[1]; q[6] = c[2]; q[7] = c[3];
  }
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < nk; k0 += NL_KC) {
    const int cnt = nk - k0 < NL_KC ? nk - k0 : NL_KC;
    __syncthreads();
    for (int i = threadIdx.x; i < cnt * 2; i += 256) {
      *reinterpret_cast<f32x4*>(ks + i * 4) = *reinterpret_cast<const f32x4*>(phi + ((long)b * nk + k0) * NLD + i * 4);
      *reinterpret_cast<f32x4*>(vs + i * 4) = *reinterpret_cast<const f32x4*>(g + ((long)b * nk + k0) * NLD + i * 4);
    }
    __syncthreads();
    for (int j = 0; j < cnt; j += 4) {
      float s[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int jj = j + t < cnt ? j + t : cnt - 1;
        const float d = dot8(q, *reinterpret_cast<const f32x4*>(ks + jj * NLD),
                             *reinterpret_cast<const f32x4*>(ks + jj * NLD + 4));
        s[t] = j + t < cnt ? d : -INFINITY;
      }
      const float mn = fmaxf(fmaxf(fmaxf(m, s[0]), fmaxf(s[1], s[2])), s[3]);
      const float sc = expf(m - mn);
      l *= sc;
#pragma unroll
      for (int d = 0; d < NLD; ++d) acc[d] *= sc;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int jj = j + t < cnt ? j + t : cnt - 1;
        const float e = expf(s[t] - mn);
        l += e;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(vs + jj * NLD), v1 = *reinterpret_cast<const f32x4*>(vs + jj * NLD + 4);
        acc[0] += e * v0[0]; acc[1] += e * v0[1]; acc[2] += e * v0[2]; acc[3] += e * v0[3];
        acc[4] += e * v1[0]; acc[5] += e * v1[1]; acc[6] += e * v1[2]; acc[7] += e * v1[3];
      }
      m = mn;
    }
  }
  if (live) {
    const float r = 1.f / l;
    float* o = y + ((long)b * nq + qi) * NLD;
    *reinterpret_cast<f32x4*>(o) = (f32x4){acc[0] * r, acc[1] * r, acc[2] * r, acc[3] * r};
    *reinterpret_cast<f32x4*>(o + 4) = (f32x4){acc[4] * r, acc[5] * r, acc[6] * r, acc[7] * r};
    lse[(long)b * nq + qi] = m + logf(l);
  }
}

// dtheta_i = sum_j ds_ij phi_j,  ds_ij = p_ij (dy_i . g_j - D_i),  D_i = dy_i . y_i,  p_ij = exp(theta_i . phi_j - lse_i)
__global__ __launch_bounds__(256) void nl_attn_bwd_q_kernel(const float* __restrict__ theta, const float* __restrict__ phi,
                                                            const float* __restrict__ g, const float* __restrict__ y,
                                                            const float* __restrict__ lse, const float* __restrict__ dy,
                                                            float* __restrict__ dtheta, float* __restrict__ dsum, int nq,
                                                            int nk) {
  __shared__ __attribute__((aligned(16))) float ks[NL_KC * NLD];
  __shared__ __attribute__((aligned(16))) float vs[NL_KC * NLD];
  const int b = blockIdx.y;
  const int qi = blockIdx.x * 256 + threadIdx.x;
  const bool live = qi < nq;
  const long row = (long)b * nq + (live ? qi : 0);
  float q[NLD], go[NLD], dq[NLD] = {};
  float D = 0.f;
#pragma unroll
  for (int d = 0; d < NLD; ++d) {
    q[d] = theta[row * NLD + d];
    go[d] = dy[row * NLD + d];
    D += go[d] * y[row * NLD + d];
  }
  const float L = lse[row];
  for (int k0 = 0; k0 < nk; k0 += NL_KC) {
    const int cnt = nk - k0 < NL_KC ? nk - k0 : NL_KC;
    __syncthreads();
    for (int i = threadIdx.x; i < cnt * 2; i += 256) {
      *reinterpret_cast<f32x4*>(ks + i * 4) = *reinterpret_cast<const f32x4*>(phi + ((long)b * nk + k0) * NLD + i * 4);
      *reinterpret_cast<f32x4*>(vs + i * 4) = *reinterpret_cast<const f32x4*>(g + ((long)b * nk + k0) * NLD + i * 4);
    }
    __syncthreads();
    for (int j = 0; j < cnt; ++j) {
      const f32x4 k0v = *reinterpret_cast<const f32x4*>(ks + j * NLD), k1v = *reinterpret_cast<const f32x4*>(ks + j * NLD + 4);
      const float p = expf(dot8(q, k0v, k1v) - L);
      const float ds = p * (dot8(go, *reinterpret_cast<const f32x4*>(vs + j * NLD),
                                 *reinterpret_cast<const f32x4*>(vs + j * NLD + 4)) - D);
      dq[0] += ds * k0v[0]; dq[1] += ds * k0v[1]; dq[2] += ds * k0v[2]; dq[3] += ds * k0v[3];
      dq[4] += ds * k1v[0]; dq[5] += ds * k1v[1]; dq[6] += ds * k1v[2]; dq[7] += ds * k1v[3];
    }
  }
  if (live) {
    float* o = dtheta + row * NLD;
    *reinterpret_cast<f32x4*>(o) = (f32x4){dq[0], dq[1], dq[2], dq[3]};
    *reinterpret_cast<f32x4*>(o + 4) = (f32x4){dq[4], dq[5], dq[6], dq[7]};
    dsum[row] = D;
  }
}

// dg_j = sum_i p_ij dy_i,  dphi_j = sum_i ds_ij theta_i   (a thread owns a key, queries stream through LDS in order)
__global__ __launch_bounds__(256) void nl_attn_bwd_k_kernel(const float* __restrict__ theta, const float* __restrict__ phi,
                                                            const float* __restrict__ g, const float* __restrict__ lse,
                                                            const float* __restrict__ dy, const float* __restrict__ dsum,
                                                            float* __restrict__ dphi, float* __restrict__ dg, int nq,
                                                            int nk) {
  __shared__ __attribute__((aligned(16))) float qs[NL_KC * NLD];
  __shared__ __attribute__((aligned(16))) float os[NL_KC * NLD];
  __shared__ float ls[NL_KC], dsm[NL_KC];
  const int b = blockIdx.y;
  const int ki = blockIdx.x * 256 + threadIdx.x;
  const bool live = ki < nk;
  const long row = (long)b * nk + (live ? ki : 0);
  float kk[NLD], vv[NLD], dk[NLD] = {}, dv[NLD] = {};
#pragma unroll
  for (int d = 0; d < NLD; ++d) {
    kk[d] = phi[row * NLD + d];
    vv[d] = g[row * NLD + d];
  }
  for (int q0 = 0; q0 < nq; q0 += NL_KC) {
    const int cnt = nq - q0 < NL_KC ? nq - q0 : NL_KC;
    __syncthreads();
    for (int i = threadIdx.x; i < cnt * 2; i += 256) {
      *reinterpret_cast<f32x4*>(qs + i * 4) = *reinterpret_cast<const f32x4*>(theta + ((long)b * nq + q0) * NLD + i * 4);
      *reinterpret_cast<f32x4*>(os + i * 4) = *reinterpret_cast<const f32x4*>(dy + ((long)b * nq + q0) * NLD + i * 4);
    }
    for (int i = threadIdx.x; i < cnt; i += 256) {
      ls[i] = lse[(long)b * nq + q0 + i];
      dsm[i] = dsum[(long)b * nq + q0 + i];
    }
    __syncthreads();
    for (int i = 0; i < cnt; ++i) {
      const f32x4 q0v = *reinterpret_cast<const f32x4*>(qs + i * NLD), q1v = *reinterpret_cast<const f32x4*>(qs + i * NLD + 4);
      const f32x4 o0 = *reinterpret_cast<const f32x4*>(os + i * NLD), o1 = *reinterpret_cast<const f32x4*>(os + i * NLD + 4);
      const float p = expf(dot8(kk, q0v, q1v) - ls[i]);
      const float ds = p * (dot8(vv, o0, o1) - dsm[i]);
      dv[0] += p * o0[0]; dv[1] += p * o0[1]; dv[2] += p * o0[2]; dv[3] += p * o0[3];
      dv[4] += p * o1[0]; dv[5] += p * o1[1]; dv[6] += p * o1[2]; dv[7] += p * o1[3];
      dk[0] += ds * q0v[0]; dk[1] += ds * q0v[1]; dk[2] += ds * q0v[2]; dk[3] += ds * q0v[3];
      dk[4] += ds * q1v[0]; dk[5] += ds * q1v[1]; dk[6] += ds * q1v[2]; dk[7] += ds * q1v[3];
    }
  }
  if (live) {
    *reinterpret_cast<f32x4*>(dphi + row * NLD) = (f32x4){dk[0], dk[1], dk[2], dk[3]};
    *reinterpret_cast<f32x4*>(dphi + row * NLD + 4) = (f32x4){dk[4], dk[5], dk[6], dk[7]};
    *reinterpret_cast<f32x4*>(dg + row * NLD) = (f32x4){dv[0], dv[1], dv[2], dv[3]};
    *reinterpret_cast<f32x4*>(dg + row * NLD + 4) = (f32x4){dv[4], dv[5], dv[6], dv[7]};
  }
}

static inline bool nl_args_ok(int nb, int nq, int nk) { return nb > 0 && nb <= 65535 && nq > 0 && nk > 0; }

extern "C" int sisr_nl_attn_fwd(const float* theta, const float* phi, const float* g, float* y, float* lse, int nb, int nq,
                                int nk, int dim, void* stream) {
  if (!theta || !phi || !g || !y || !lse || !nl_args_ok(nb, nq, nk)) return SISR_ERR_ARG;
  if (dim != NLD) return SISR_ERR_UNSUPPORTED;
  if (!sisr_aligned16(theta) || !sisr_aligned16(phi) || !sisr_aligned16(g) || !sisr_aligned16(y)) return SISR_ERR_ALIGN;
  hipLaunchKernelGGL(nl_attn_fwd_kernel, dim3((nq + 255) / 256, nb), dim3(256), 0, (hipStream_t)stream, theta, phi, g, y,
                     lse, nq, nk);
  return sisr_check_launch();
}

extern "C" int sisr_nl_attn_bwd(const float* theta, const float* phi, const float* g, const float* y, const float* lse,
                                const float* dy, float* dtheta, float* dphi, float* dg, float* dsum, int nb, int nq,
                                int nk, int dim, void* stream) {
  if (!theta || !phi || !g || !y || !lse || !dy || !dtheta || !dphi || !dg || !dsum || !nl_args_ok(nb, nq, nk))
    return SISR_ERR_ARG;
  if (dim != NLD) return SISR_ERR_UNSUPPORTED;
  if (!sisr_aligned16(theta) || !sisr_aligned16(phi) || !sisr_aligned16(g) || !sisr_aligned16(dy) ||
      !sisr_aligned16(dtheta) || !sisr_aligned16(dphi) || !sisr_aligned16(dg))
    return SISR_ERR_ALIGN;
  hipLaunchKernelGGL(nl_attn_bwd_q_kernel, dim3((nq + 255) / 256, nb), dim3(256), 0, (hipStream_t)stream, theta, phi, g, y,
                     lse, dy, dtheta, dsum, nq, nk);
  int rc = sisr_check_launch();
  if (rc != SISR_OK) return rc;
  hipLaunchKernelGGL(nl_attn_bwd_k_kernel, dim3((nk + 255) / 256, nb), dim3(256), 0, (hipStream_t)stream, theta, phi, g,
                     lse, dy, dsum, dphi, dg, nq, nk);
  return sisr_check_launch();
}
