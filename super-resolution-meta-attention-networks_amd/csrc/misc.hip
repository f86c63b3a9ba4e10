// L1 loss (forward value + unit-scale gradient) and a flat fused Adam step, fp32.
//
//   L1   (ref: SISR/models/__init__.py:268 nn.L1Loss(), :472 criterion(out, y)):
//          loss = mean |a - b|;   dloss/da = sign(a - b) / N      (sign(0) = 0, as torch)
//   Adam (ref: SISR/models/__init__.py:299-308 optim.Adam(lr, betas), no weight decay / amsgrad):
//          m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//          p -= lr / (1-b1^t) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
// Both are single-pass HBM-bound streams; the loss reduction is two-stage and ordered.
#include "sisr_common.h"
#include <string.h>

#define L1_BLOCKS 512

__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ grad, float inv_n, long n,
                                                         float* __restrict__ part) {
  __shared__ float red[256];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float d = a[i] - b[i];
    s += fabsf(d);
    if (grad) grad[i] = d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void l1_final_kernel(const float* __restrict__ part, int nparts, float inv_n,
                                                       float* __restrict__ loss) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = red[0] * inv_n;
}

extern "C" size_t sisr_l1_loss_workspace_bytes() { return L1_BLOCKS * sizeof(float); }

extern "C" int sisr_l1_loss(const float* a, const float* b, long n, float* loss, float* grad, float* workspace,
                            void* stream) {
  if (!a || !b || !loss || !workspace || n <= 0) return SISR_ERR_ARG;
  long blocks = (n + 255) / 256;
  if (blocks > L1_BLOCKS) blocks = L1_BLOCKS;
  const float inv_n = 1.0f / (float)n;
  hipLaunchKernelGGL(l1_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, grad, inv_n, n,
                     workspace);
  int rc = sisr_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, workspace, (int)blocks, inv_n, loss);
  return sisr_check_launch();
}

// torch.optim.Adam (no amsgrad, no weight decay) over one flat fp32 range, in torch's operation order:
//   exp_avg.lerp_(g, 1 - b1);  exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2);
//   denom = exp_avg_sq.sqrt() / sqrt(1 - b2^t) + eps;  p.addcdiv_(exp_avg, denom, value = -lr / (1 - b1^t))
// step_size = lr / (1 - b1^t), bc2_sqrt = sqrt(1 - b2^t) and the weights omb1 = 1 - b1, omb2 = 1 - b2 are computed on
// the host in double and then rounded to fp32, as torch does (1.f - 0.999f is 4.7e-5 off the fp32 value of 0.001).
__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long n4, long n,
                                                        float b2, float omb1, float omb2, float eps, float step_size,
                                                        float bc2_sqrt, float gscale) {
#pragma clang fp contract(off)
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 gg = reinterpret_cast<const f32x4*>(g)[i] * gscale;
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
    mm = mm + (gg - mm) * omb1;
    vv = vv * b2 + gg * gg * omb2;
#pragma unroll
    for (int e = 0; e < 4; ++e) pp[e] = pp[e] - step_size * (mm[e] / (sqrtf(vv[e]) / bc2_sqrt + eps));
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    reinterpret_cast<f32x4*>(p)[i] = pp;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // tail
    const long i = (n4 << 2) + threadIdx.x;
    const float gg = g[i] * gscale;
    const float mm = m[i] + (gg - m[i]) * omb1;
    const float vv = v[i] * b2 + gg * gg * omb2;
    p[i] = p[i] - step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
    m[i] = mm;
    v[i] = vv;
  }
}

extern "C" int sisr_adam_flat(float* p, const float* g, float* m, float* v, long n, float beta2, float one_minus_beta1,
                              float one_minus_beta2, float eps, float step_size, float bc2_sqrt, float grad_scale,
                              void* stream) {
  if (!p || !g || !m || !v || n <= 0) return SISR_ERR_ARG;
  if (!sisr_aligned16(p) || !sisr_aligned16(g) || !sisr_aligned16(m) || !sisr_aligned16(v)) return SISR_ERR_ALIGN;
  const long n4 = n >> 2;
  long blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, n, beta2,
                     one_minus_beta1, one_minus_beta2, eps, step_size, bc2_sqrt, grad_scale);
  return sisr_check_launch();
}

// ---------------------------------------------------------------- progress words a captured step publishes to the host
// A training step replayed from a hipGraph fires no autograd hooks, so a data-parallel reducer cannot learn from the host
// side when a gradient bucket is complete.  sisr_signal_host is a one-thread kernel placed in the captured stream right
// after a bucket's last gradient kernel: it adds 1 to a word in host-coherent pinned memory (system-scope release), which
// the host polls before it issues that bucket's all-reduce on its own stream while the rest of the replay keeps running
// (parallel.GradReducer; ref: the reference's DataParallel gathers gradients only after the whole backward,
// SISR/models/__init__.py:344-347, 481-489).  Kernels ahead of the signal in the capture have completed, and their writes
// are device-visible, when the word changes.
__global__ void signal_host_kernel(unsigned* flag) {
  __threadfence_system();
  __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One lane that sleeps and re-reads the progress word until it has reached `value` (wrap-around safe), bounded: after
// ~2^26 polls (about a minute) it gives up and raises *timed_out, so the grid always drains.  Launched on the reducer's
// stream in front of a bucket's all-reduce: the stream then waits for the replay's signal node without the host (which
// would block) and without a command-processor wait packet (hipStreamWaitValue32 slowed the replay's own launches).
__global__ void wait_flag_kernel(const unsigned* flag, unsigned value, unsigned* timed_out) {
  for (unsigned i = 0; i < (1u << 26); ++i) {
    const unsigned v = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((int)(v - value) >= 0) return;
    __builtin_amdgcn_s_sleep(127);
  }
  if (timed_out) __hip_atomic_store(timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" void* sisr_host_flags_alloc(int n) {
  if (n <= 0) return nullptr;
  void* p = nullptr;
  if (hipHostMalloc(&p, (size_t)n * sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return nullptr;
  memset(p, 0, (size_t)n * sizeof(unsigned));
  return p;
}

extern "C" void sisr_host_flags_free(void* flags) {
  if (flags) (void)hipHostFree(flags);
}

// Device-side wait: `stream` proceeds once *flag - value >= 0 (32-bit, wrap-around safe while the two stay within 2^31).
// With it the host enqueues a bucket's all-reduce right after graph.replay() without blocking: the reducer stream itself
// waits for the replay's signal node.
extern "C" int sisr_stream_wait_flag(void* flag, unsigned value, void* stream) {
  if (!flag) return SISR_ERR_ARG;
  const hipError_t e = hipStreamWaitValue32((hipStream_t)stream, flag, value, hipStreamWaitValueGte, 0xffffffffu);
  return e == hipSuccess ? SISR_OK : SISR_ERR_LAUNCH - (int)e * 16;
}

// the same wait as a one-lane polling kernel (bounded; *timed_out, a word of the same allocation or nullptr, is set to 1 if
// the value never arrives)
extern "C" int sisr_stream_spin_flag(void* flag, unsigned value, void* timed_out, void* stream) {
  if (!flag) return SISR_ERR_ARG;
  hipLaunchKernelGGL(wait_flag_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, static_cast<const unsigned*>(flag), value,
                     static_cast<unsigned*>(timed_out));
  return sisr_check_launch();
}

extern "C" int sisr_signal_host(void* flag, void* stream) {
  if (!flag) return SISR_ERR_ARG;
  hipLaunchKernelGGL(signal_host_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, static_cast<unsigned*>(flag));
  return sisr_check_launch();
}

// ---------------------------------------------------------------- training tiles cut on the device
// dst[b][c][i][j] = aug_b[c][top + i][left + j], aug = transpose?(vflip?(hflip?(src_b)))  -- the reference's
// random_flip_rotate followed by random_matched_crop (ref: sr_tools/image_manipulation.py:233-257, applied in that
// order by data_handler.py:500-513) as ONE gather over a device-resident copy of the dataset.  The random draws
// stay on the host (Python `random`, the reference's call order); params[b] = {H, W, top, left, hflip, vflip,
// transpose, 0} with H, W the SOURCE image size.  Planar fp32 in, NCHW fp32 out.
struct CropRec {
  int H, W, top, left, hflip, vflip, rot, pad;
};

__global__ __launch_bounds__(256) void crop_augment_kernel(const float* const* __restrict__ src,
                                                           const CropRec* __restrict__ prm, float* __restrict__ dst,
                                                           int C, int crop) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= crop * crop) return;
  const CropRec r = prm[b];
  const int i = idx / crop, j = idx - i * crop;
  const int y = r.top + i, x = r.left + j;          // coordinates in the augmented image
  const int y2 = r.rot ? x : y, x2 = r.rot ? y : x;  // undo the transpose
  const int sy = r.vflip ? r.H - 1 - y2 : y2, sx = r.hflip ? r.W - 1 - x2 : x2;
  dst[(((long)b * C + c) * crop + i) * crop + j] = src[b][((long)c * r.H + sy) * r.W + sx];
}

extern "C" int sisr_crop_augment(const float* const* src, const int* params, float* dst, int B, int C, int crop,
                                 void* stream) {
  if (!src || !params || !dst || B <= 0 || C <= 0 || crop <= 0 || B > 65535 || C > 65535) return SISR_ERR_ARG;
  hipLaunchKernelGGL(crop_augment_kernel, dim3((crop * crop + 255) / 256, C, B), dim3(256), 0, (hipStream_t)stream, src,
                     reinterpret_cast<const CropRec*>(params), dst, C, crop);
  return sisr_check_launch();
}

// ---------------------------------------------------------------- channel padding / RGB pixel-shuffle (SRMD widening)
// ref: advanced/architectures.py:380-425 SRMD: conv(3+M -> nc) ... conv(nc -> 3 r^2) + PixelShuffle(r).  The MFMA conv
// kernels work on 64-channel chunks, so the (3+M)-channel NCHW input is laid out once as a zero-padded NHWC map, the
// head / tail weights are zero-padded copies, and the tail's 64-channel NHWC result is shuffled into the NCHW RGB image.

// NCHW (B, C, H, W) -> NHWC (B, H, W, Cp), channels >= C zero.  One thread per float4 of the output.
__global__ __launch_bounds__(256) void nchw_to_nhwc_pad_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                               long hw, int Cp, long total4) {
  const int c4n = Cp >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    const long pix = i / c4n;
    const int c0 = (int)(i - pix * c4n) * 4;
    const long b = pix / hw, p = pix - b * hw;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c0 + e < C) v[e] = x[((long)b * C + c0 + e) * hw + p];
    reinterpret_cast<f32x4*>(y)[i] = v;
  }
}

// OIHW weight (co, ci, 3, 3) <-> zero-padded (cop, cip, 3, 3): dir 0 pads (dst = padded), dir 1 crops (dst = real)
__global__ __launch_bounds__(256) void pad_oihw_kernel(const float* __restrict__ src, float* __restrict__ dst, int co, int ci,
                                                       int cop, int cip, int taps, int dir) {
  const long total = dir ? (long)co * ci * taps : (long)cop * cip * taps;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int t = (int)(i % taps);
    const long r = i / taps;
    if (dir) {
      const int o = (int)(r / ci), c = (int)(r - (long)o * ci);
      dst[i] = src[((long)o * cip + c) * taps + t];
    } else {
      const int o = (int)(r / cip), c = (int)(r - (long)o * cip);
      dst[i] = (o < co && c < ci) ? src[((long)o * ci + c) * taps + t] : 0.f;
    }
  }
}

// PixelShuffle(r) of the first C*r*r channels of an NHWC (B, H, W, Cp) map into NCHW (B, C, rH, rW):
// out[b][c][r h + i][r w + j] = y[b][h][w][c r^2 + i r + j]; dir 1 is the adjoint (gradient): padded channels <- 0.
__global__ __launch_bounds__(256) void shuffle_rgb_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int r,
                                                          int H, int W, int Cp, long total, int dir) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    if (!dir) {  // i indexes the NCHW output
      const int ow = (int)(i % ((long)W * r));
      long t = i / ((long)W * r);
      const int oh = (int)(t % ((long)H * r));
      t /= (long)H * r;
      const int c = (int)(t % C);
      const long b = t / C;
      const int h = oh / r, ii = oh - h * r, w = ow / r, jj = ow - w * r;
      dst[i] = src[(((long)b * H + h) * W + w) * Cp + c * r * r + ii * r + jj];
    } else {  // i indexes the NHWC gradient
      const int ch = (int)(i % Cp);
      const long pix = i / Cp;
      float v = 0.f;
      if (ch < C * r * r) {
        const int w = (int)(pix % W);
        const long t = pix / W;
        const int h = (int)(t % H);
        const long b = t / H;
        const int c = ch / (r * r), q = ch - c * r * r, ii = q / r, jj = q - ii * r;
        v = src[(((long)b * C + c) * ((long)H * r) + (long)h * r + ii) * ((long)W * r) + (long)w * r + jj];
      }
      dst[i] = v;
    }
  }
}

static unsigned misc_blocks(long n) {
  long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 65535 ? 65535 : b));
}

extern "C" int sisr_nchw_to_nhwc_pad(const float* x, float* y, int B, int C, int H, int W, int Cp, void* stream) {
  if (!x || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cp < C || (Cp & 3)) return SISR_ERR_ARG;
  if (!sisr_aligned16(y)) return SISR_ERR_ALIGN;
  const long total4 = (long)B * H * W * (Cp >> 2);
  hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel, dim3(misc_blocks(total4)), dim3(256), 0, (hipStream_t)stream, x, y, C,
                     (long)H * W, Cp, total4);
  return sisr_check_launch();
}

extern "C" int sisr_pad_oihw(const float* src, float* dst, int cout, int cin, int cout_padded, int cin_padded, int taps,
                             int crop, void* stream) {
  if (!src || !dst || cout <= 0 || cin <= 0 || cout_padded < cout || cin_padded < cin || taps <= 0) return SISR_ERR_ARG;
  const long total = crop ? (long)cout * cin * taps : (long)cout_padded * cin_padded * taps;
  hipLaunchKernelGGL(pad_oihw_kernel, dim3(misc_blocks(total)), dim3(256), 0, (hipStream_t)stream, src, dst, cout, cin,
                     cout_padded, cin_padded, taps, crop);
  return sisr_check_launch();
}

// fp32 map -> bf16 map (RNE), eight elements per thread (include/sisr_hip.h: bf16 storage)
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long n8) {
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const f32x4 a = reinterpret_cast<const f32x4*>(src)[2 * i], b = reinterpret_cast<const f32x4*>(src)[2 * i + 1];
    bf16x8_t r;
    r[0] = (__bf16)a[0]; r[1] = (__bf16)a[1]; r[2] = (__bf16)a[2]; r[3] = (__bf16)a[3];
    r[4] = (__bf16)b[0]; r[5] = (__bf16)b[1]; r[6] = (__bf16)b[2]; r[7] = (__bf16)b[3];
    reinterpret_cast<u32x4_t*>(dst)[i] = __builtin_bit_cast(u32x4_t, r);
  }
}

extern "C" int sisr_f32_to_bf16(const float* src, void* dst, long n, void* stream) {
  if (!src || !dst || n <= 0 || (n & 7)) return SISR_ERR_ARG;
  if (!sisr_aligned16(src) || !sisr_aligned16(dst)) return SISR_ERR_ALIGN;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(misc_blocks(n / 8)), dim3(256), 0, (hipStream_t)stream, src,
                     static_cast<unsigned short*>(dst), n / 8);
  return sisr_check_launch();
}

extern "C" int sisr_shuffle_rgb(const float* src, float* dst, int B, int C, int r, int H, int W, int Cp, int adjoint,
                                void* stream) {
  if (!src || !dst || B <= 0 || C <= 0 || r <= 0 || H <= 0 || W <= 0 || Cp < C * r * r) return SISR_ERR_ARG;
  const long total = adjoint ? (long)B * H * W * Cp : (long)B * C * H * r * W * r;
  hipLaunchKernelGGL(shuffle_rgb_kernel, dim3(misc_blocks(total)), dim3(256), 0, (hipStream_t)stream, src, dst, C, r, H, W,
                     Cp, total, adjoint);
  return sisr_check_launch();
}

// ------------------------------------------------------------------ map stacks (HAN: ref advanced/architectures.py:357-362)
// The reference concatenates the 11 intermediate maps along channels; here they sit in one [B][N][hw][64] stack that LAM and
// the 704 -> 64 conv read as chunks.  unstack == 0: map [B][hw][64] -> slot k of the stack; else the reverse (gradients).
__global__ __launch_bounds__(256) void stack_maps_kernel(const float* __restrict__ src, float* __restrict__ dst, long hw16,
                                                         int N, int k, long total, int unstack) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / hw16, r = i - b * hw16;
    const long s = (b * N + k) * hw16 + r;
    if (!unstack) reinterpret_cast<f32x4*>(dst)[s] = reinterpret_cast<const f32x4*>(src)[i];
    else reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(src)[s];
  }
}

extern "C" int sisr_stack_maps(const float* src, float* dst, int B, long hw, int N, int k, int unstack, void* stream) {
  if (!src || !dst || B <= 0 || hw <= 0 || N <= 0 || k < 0 || k >= N) return SISR_ERR_ARG;
  if (!sisr_aligned16(src) || !sisr_aligned16(dst)) return SISR_ERR_ALIGN;
  const long total = (long)B * hw * 16;
  hipLaunchKernelGGL(stack_maps_kernel, dim3(misc_blocks(total)), dim3(256), 0, (hipStream_t)stream, src, dst, hw * 16, N, k, total,
                     unstack);
  return sisr_check_launch();
}

// ------------------------------------------------------------------ PixelShuffle on channels-last maps (wide upsamplers)
// ref: advanced/common.py:20-45 Upsampler = conv(C -> r^2 C) + nn.PixelShuffle(r).  For C = 64 the shuffle is the conv's store
// address map (View, sisr_common.h); wider maps (EDSR-256) come here: in [B][H][W][C r^2] -> out [B][rH][rW][C],
// out[b][r h + dy][r w + dx][c] = in[b][h][w][c r^2 + dy r + dx]; adjoint != 0: the inverse gather (the gradient).
__global__ __launch_bounds__(256) void pixel_shuffle_cl_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W,
                                                               int C, int r, long total, int adjoint) {
  const int rr = r * r;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    // i indexes the shuffled tensor [B][rH][rW][C]
    const int c = (int)(i % C);
    long t = i / C;
    const int ow = (int)(t % ((long)r * W));
    t /= (long)r * W;
    const int oh = (int)(t % ((long)r * H));
    const long b = t / ((long)r * H);
    const int h = oh / r, dy = oh - h * r, w = ow / r, dx = ow - w * r;
    const long j = (((b * H + h) * W + w) * C + c) * rr + dy * r + dx;
    if (!adjoint) dst[i] = src[j];
    else dst[j] = src[i];
  }
}

extern "C" int sisr_pixel_shuffle_cl(const float* src, float* dst, int B, int H, int W, int C, int r, int adjoint, void* stream) {
  if (!src || !dst || B <= 0 || H <= 0 || W <= 0 || C <= 0 || r < 1) return SISR_ERR_ARG;
  const long total = (long)B * H * W * C * r * r;
  hipLaunchKernelGGL(pixel_shuffle_cl_kernel, dim3(misc_blocks(total)), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, C, r, total,
                     adjoint);
  return sisr_check_launch();
}
