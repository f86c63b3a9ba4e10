// On-the-fly LR synthesis on the device: the reference's per-image degradation
//   blurred = BatchBlur(l = 21)(hr, kernel)                 ref: sr_tools/gaussian_utils.py:346-368 (reflection pad + depthwise conv)
//   pil     = ToPILImage(blurred)  = uint8(blurred * 255)   ref: gaussian_utils.py:52-53 (`pic.mul(255).byte()`: truncation)
//   lr      = pil.resize((W/s, H/s), PIL.Image.BICUBIC)     ref: sr_tools/image_manipulation.py:32-53 (downsample)
//   tensor  = ToTensor(lr) = float(lr) / 255                ref: data_handler.py lr_transform
// as three kernels over planar images: blur + quantise, PIL's horizontal resample pass, PIL's vertical pass (+ /255).
// The resample passes are PIL's 8-bit fixed-point arithmetic (libImaging/Resample.c: 22 fractional bits, rounding
// constant 1 << 21, clip to [0, 255] after EACH pass) with the coefficient tables computed on the host exactly as
// precompute_coeffs / normalize_coeffs_8bpc do, so the LR image is bit-identical to PIL's given the same uint8 input.
// The blur is an fp32 sum of l*l products in row-major tap order; the reference's MKLDNN convolution sums in another
// order, so a pixel whose 255 x value lies within ~1e-5 of an integer can truncate to the neighbouring byte.
#include "sisr_common.h"

#define BT 32                 // output tile edge
#define BL_MAX 21             // largest blur kernel edge (reference default and maximum used: 21)

__device__ __forceinline__ int reflect_index(int i, int n) {  // nn.ReflectionPad2d: -1 -> 1, n -> n-2
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

// x: [C][H][W] fp32, k: [l][l] fp32 (cross-correlation, as F.conv2d), y: [C][H][W] uint8 = (uint8)(conv * 255)
__global__ __launch_bounds__(256) void blur_quant_kernel(const float* __restrict__ x, const float* __restrict__ k,
                                                         unsigned char* __restrict__ y, float* __restrict__ yf, int H, int W,
                                                         int l) {
  __shared__ float tile[(BT + BL_MAX - 1) * (BT + BL_MAX - 1)];
  __shared__ float kk[BL_MAX * BL_MAX];
  const int c = blockIdx.z, h0 = blockIdx.y * BT, w0 = blockIdx.x * BT;
  const int pl = l / 2, pr = l - 1 - pl;  // odd l: l/2 both sides; even l: (l/2, l/2 - 1)  (ref :350-353)
  const int TS = BT + l - 1;
  const float* xc = x + (long)c * H * W;
  for (int i = threadIdx.x; i < l * l; i += 256) kk[i] = k[i];
  for (int i = threadIdx.x; i < TS * TS; i += 256) {
    const int r = i / TS, q = i - r * TS;
    tile[i] = xc[(long)reflect_index(h0 - pl + r, H) * W + reflect_index(w0 - pl + q, W)];
  }
  (void)pr;
  __syncthreads();
  for (int o = threadIdx.x; o < BT * BT; o += 256) {
    const int r = o / BT, q = o - r * BT;
    if (h0 + r >= H || w0 + q >= W) continue;
    float acc = 0.f;
    for (int i = 0; i < l; ++i)
      for (int j = 0; j < l; ++j) acc = __builtin_fmaf(tile[(r + i) * TS + q + j], kk[i * l + j], acc);
    const long at = ((long)c * H + h0 + r) * W + w0 + q;
    if (yf) yf[at] = acc;
    if (y) y[at] = (unsigned char)(int)(acc * 255.0f);
  }
}

// Gaussian noise on the blurred image, then ToPILImage's quantisation (ref: gaussian_utils.py:306-312 b_GaussianNoising,
// :409-411): y = byte(255 * clamp(noise * sigma + blur, 0, 1)).  The noise field itself is drawn on the host from numpy's
// global stream (the reference's np.random.normal call); product and sum are rounded separately, as torch's mul / add are.
__global__ __launch_bounds__(256) void noise_quant_kernel(const float* __restrict__ blur, const float* __restrict__ noise,
                                                          float sigma, unsigned char* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v = __fadd_rn(__fmul_rn(noise[i], sigma), blur[i]);
    v = fminf(fmaxf(v, 0.f), 1.f);
    y[i] = (unsigned char)(int)(v * 255.0f);
  }
}

// One PIL resample pass over a planar uint8 image: out[c][y][x] = clip8((2^21 + sum_k in[...] * coef[x][k]) >> 22).
// horizontal: in [C][H][Win] -> out [C][H][Wout], taps run along x; vertical: in [C][Hin][W] -> out [C][Hout][W].
// bounds[o] = (first input index, tap count); coef [out][ksize] int32.
template <bool VERT, bool TO_FLOAT>
__global__ __launch_bounds__(256) void pil_resample_kernel(const unsigned char* __restrict__ in, void* __restrict__ outp,
                                                           const int* __restrict__ bounds, const int* __restrict__ coef,
                                                           int ksize, int C, int Hin, int Win, int Hout, int Wout) {
  const long total = (long)C * Hout * Wout;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int xo = (int)(i % Wout);
    const long t = i / Wout;
    const int yo = (int)(t % Hout), c = (int)(t / Hout);
    const int o = VERT ? yo : xo;
    const int lo = bounds[2 * o], n = bounds[2 * o + 1];
    const int* kp = coef + (long)o * ksize;
    int ss = 1 << 21;
    const unsigned char* src = in + (long)c * Hin * Win;
    if (VERT) {
      for (int k = 0; k < n; ++k) ss += (int)src[(long)(lo + k) * Win + xo] * kp[k];
    } else {
      for (int k = 0; k < n; ++k) ss += (int)src[(long)yo * Win + lo + k] * kp[k];
    }
    int v = ss >> 22;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    if (TO_FLOAT) static_cast<float*>(outp)[i] = (float)v / 255.0f;  // ToTensor: .float().div(255)
    else static_cast<unsigned char*>(outp)[i] = (unsigned char)v;
  }
}

static unsigned deg_blocks(long n) {
  long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 65535 ? 65535 : b));
}

// y_u8 and / or y_f32 (nullable, not both null): the quantised image PIL receives / the unquantised blur (tests)
extern "C" int sisr_blur_quant(const float* x, const float* kernel, unsigned char* y_u8, float* y_f32, int C, int H, int W,
                               int l, void* stream) {
  if (!x || !kernel || (!y_u8 && !y_f32) || C <= 0 || H <= 0 || W <= 0 || l < 1) return SISR_ERR_ARG;
  if (l > BL_MAX || C > 65535) return SISR_ERR_UNSUPPORTED;
  if (l / 2 >= H || l / 2 >= W) return SISR_ERR_UNSUPPORTED;  // reflection padding needs pad < size (torch raises too)
  const dim3 grid((W + BT - 1) / BT, (H + BT - 1) / BT, C);
  if (grid.y > 65535) return SISR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(blur_quant_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, kernel, y_u8, y_f32, H, W, l);
  return sisr_check_launch();
}

extern "C" int sisr_noise_quant(const float* blur, const float* noise, float sigma, unsigned char* y_u8, long n,
                                void* stream) {
  if (!blur || !noise || !y_u8 || n <= 0) return SISR_ERR_ARG;
  hipLaunchKernelGGL(noise_quant_kernel, dim3(deg_blocks(n)), dim3(256), 0, (hipStream_t)stream, blur, noise, sigma, y_u8, n);
  return sisr_check_launch();
}

// vertical = 0: [C][H][Win] -> [C][H][Wout] (uint8);  vertical = 1: [C][Hin][W] -> [C][Hout][W], uint8 or, with
// to_float, fp32 / 255.  bounds / coef device arrays from the host table (pil_bicubic_table in degrade.py).
extern "C" int sisr_pil_resample(const unsigned char* in, void* out, const int* bounds, const int* coef, int ksize, int C,
                                 int Hin, int Win, int Hout, int Wout, int vertical, int to_float, void* stream) {
  if (!in || !out || !bounds || !coef || ksize <= 0 || C <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0)
    return SISR_ERR_ARG;
  if (vertical ? Win != Wout : Hin != Hout) return SISR_ERR_ARG;
  const long total = (long)C * Hout * Wout;
  const dim3 g(deg_blocks(total));
  hipStream_t st = (hipStream_t)stream;
  if (vertical) {
    if (to_float) hipLaunchKernelGGL((pil_resample_kernel<true, true>), g, dim3(256), 0, st, in, out, bounds, coef, ksize, C, Hin, Win, Hout, Wout);
    else hipLaunchKernelGGL((pil_resample_kernel<true, false>), g, dim3(256), 0, st, in, out, bounds, coef, ksize, C, Hin, Win, Hout, Wout);
  } else {
    if (to_float) return SISR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((pil_resample_kernel<false, false>), g, dim3(256), 0, st, in, out, bounds, coef, ksize, C, Hin, Win, Hout, Wout);
  }
  return sisr_check_launch();
}
