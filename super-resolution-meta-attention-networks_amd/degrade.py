"""On-the-fly LR synthesis (`online_degradations`): random Gaussian blur + bicubic down-sampling of
the HR image on the device, with the blur-kernel PCA code as the sample's metadata.

ref: Code/sr_tools/gaussian_utils.py:218-303 (random kernels), :193-198 (PCA), :332-343 (PCAEncoder), :346-368
     (BatchBlur), :371-424 (SRMDPreprocessing); Code/sr_tools/data_handler.py:222-238 (degrader set-up: PCA basis from
     30 000 random kernels), :446-456 (per image: degrader -> ToPILImage -> downsample -> metadata);
     Code/sr_tools/image_manipulation.py:32-53 (downsample = centre crop to a multiple of the scale + PIL BICUBIC).

Split of labour.  Everything random or tiny stays on the host, in the reference's call order and on the same numpy
global stream (np.random.random draws per kernel; torch.svd for the PCA basis; the 1 x 441 by 441 x 10 code product).
Pixels stay on the device: csrc/degrade.hip blurs the planar fp32 HR image with reflection padding and quantises it
as ToPILImage does (`mul(255).byte()`), then runs PIL's two 8-bit fixed-point resample passes with coefficient tables
computed here exactly as libImaging/Resample.c computes them -- the LR image is bit-identical to PIL's for the same
uint8 input.  There is no CPU path: the tensors must live on a HIP device.
"""
import math

import numpy as np
import torch

from . import hip

PRECISION_BITS = 32 - 8 - 2  # libImaging/Resample.c


# ----------------------------------------------------------------------------- random blur kernels (host, np.random)
def isotropic_gaussian_kernel(l, sigma):
    ax = np.arange(-l // 2 + 1., l // 2 + 1.)
    xx, yy = np.meshgrid(ax, ax)
    kernel = np.exp(-(xx ** 2 + yy ** 2) / (2. * sigma ** 2))
    return kernel / np.sum(kernel)


def cal_sigma(sig_x, sig_y, radians):
    D = np.array([[sig_x ** 2, 0], [0, sig_y ** 2]])
    U = np.array([[np.cos(radians), -np.sin(radians)], [np.sin(radians), 1 * np.cos(radians)]])
    return np.dot(U, np.dot(D, U.T))


def anisotropic_gaussian_kernel(l, sigma_matrix):
    ax = np.arange(-l // 2 + 1., l // 2 + 1.)
    xx, yy = np.meshgrid(ax, ax)
    xy = np.hstack((xx.reshape((l * l, 1)), yy.reshape(l * l, 1))).reshape(l, l, 2)
    inverse_sigma = np.linalg.inv(sigma_matrix)
    kernel = np.exp(-0.5 * np.sum(np.dot(xy, inverse_sigma) * xy, 2))
    return kernel / np.sum(kernel)


def random_gaussian_kernel(l=21, sig_min=0.2, sig_max=4.0, rate_iso=1.0, scaling=3):
    """One kernel; consumes np.random exactly as gaussian_utils.random_gaussian_kernel (:278-282)."""
    if np.random.random() < rate_iso:
        x = np.random.random() * (sig_max - sig_min) + sig_min
        return isotropic_gaussian_kernel(l, x)
    pi = np.random.random() * math.pi * 2 - math.pi
    x = np.random.random() * (sig_max - sig_min) + sig_min
    y = np.clip(np.random.random() * scaling * x, sig_min, sig_max)
    return anisotropic_gaussian_kernel(l, cal_sigma(x, y, pi))


def random_batch_kernel(batch, l=21, sig_min=0.2, sig_max=4.0, rate_iso=1.0, scaling=3):
    out = np.zeros((batch, l, l))
    for i in range(batch):
        out[i] = random_gaussian_kernel(l=l, sig_min=sig_min, sig_max=sig_max, rate_iso=rate_iso, scaling=scaling)
    return out


def pca_matrix(batch=30000, k=10):
    """The kernel-code basis the dataset builds at construction (data_handler.py:228-231): `batch` random default
    kernels from the np.random stream, centred, torch.svd of the transposed data, first k left singular vectors."""
    data = random_batch_kernel(batch=batch).reshape((batch, -1))
    X = torch.from_numpy(data)
    X = X - torch.mean(X, 0).expand_as(X)
    U, S, V = torch.svd(torch.t(X))
    return U[:, :k].float()


def encode_kernel(kernel_f32, pca):
    """PCAEncoder (:332-343): (l, l) float32 kernel -> (k,) code."""
    kt = torch.as_tensor(kernel_f32, dtype=torch.float32)
    return torch.bmm(kt.reshape(1, 1, -1), pca.expand((1,) + tuple(pca.shape))).view(-1)


# ----------------------------------------------------------------------------- PIL's bicubic coefficient tables (host)
def _bicubic_filter(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_bicubic_table(in_size, out_size):
    """libImaging/Resample.c precompute_coeffs + normalize_coeffs_8bpc for a full-extent box:
    -> (bounds int32 [out][2] = (first tap, taps), coef int32 [out][ksize], ksize)."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    coef = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            coef[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, coef, ksize


_tables = {}


def _device_table(in_size, out_size, device):
    key = (in_size, out_size, device.index)
    t = _tables.get(key)
    if t is None:
        b, c, ks = pil_bicubic_table(in_size, out_size)
        t = (torch.from_numpy(b).to(device), torch.from_numpy(c).to(device), ks)
        _tables[key] = t
    return t


# ----------------------------------------------------------------------------- device pipeline
def blur_quant(hr, kernel, want_float=False, want_u8=True):
    """(C, H, W) fp32 on the device, (l, l) kernel -> uint8 (C, H, W) = byte(255 * BatchBlur(hr)); want_float: also
    the unquantised blur (want_u8 False: only that)."""
    if not hr.is_cuda:
        raise RuntimeError("degrade.blur_quant: the image must be on a HIP device (no CPU path)")
    C, H, W = hr.shape
    k = torch.as_tensor(kernel, dtype=torch.float32).to(hr.device).contiguous()
    l = k.shape[-1]
    y = torch.empty((C, H, W), device=hr.device, dtype=torch.uint8) if want_u8 else None
    yf = torch.empty((C, H, W), device=hr.device, dtype=torch.float32) if want_float else None
    hip.check(hip.lib().sisr_blur_quant(hip.ptr(hr.contiguous()), hip.ptr(k), y.data_ptr() if y is not None else None,
                                        hip.ptr(yf), C, H, W, l, hip.stream()), "sisr_blur_quant")
    return (y, yf) if want_float else y


def random_batch_noise(batch, high, rate_cln=1.0):
    """Noise level per sample; consumes np.random exactly as gaussian_utils.random_batch_noise (:299-304)."""
    noise_level = np.random.uniform(size=(batch, 1)) * high
    noise_mask = np.random.uniform(size=(batch, 1))
    noise_mask[noise_mask < rate_cln] = 0
    noise_mask[noise_mask >= rate_cln] = 1
    return noise_level * noise_mask


def noise_quant(blur, sigma):
    """(C, H, W) fp32 blurred image on the device -> uint8 byte(255 * clamp(N(0,1) * sigma + blur, 0, 1)).  The N(0,1) field
    comes from numpy's global stream on the host, as in the reference (b_GaussianNoising, :306-312), and is uploaded."""
    C, H, W = blur.shape
    field = torch.FloatTensor(np.random.normal(loc=0.0, scale=1.0, size=(1, C, H, W))).to(blur.device)
    y = torch.empty((C, H, W), device=blur.device, dtype=torch.uint8)
    hip.check(hip.lib().sisr_noise_quant(hip.ptr(blur), hip.ptr(field), float(sigma), y.data_ptr(), blur.numel(), hip.stream()),
              "sisr_noise_quant")
    return y


def pil_bicubic_downsample(u8, scale, to_float=True):
    """PIL `resize((W // s, H // s), BICUBIC)` of a planar uint8 (C, H, W) device image whose sides are multiples of s
    -> (C, H / s, W / s): fp32 / 255 (ToTensor) or uint8."""
    C, H, W = u8.shape
    if H % scale or W % scale:
        raise ValueError("centre-crop the image to a multiple of the scale first (image_manipulation.downsample)")
    h, w = H // scale, W // scale
    dev = u8.device
    bh, ch, ksh = _device_table(W, w, dev)
    bv, cv, ksv = _device_table(H, h, dev)
    tmp = torch.empty((C, H, w), device=dev, dtype=torch.uint8)
    L = hip.lib()
    hip.check(L.sisr_pil_resample(u8.contiguous().data_ptr(), tmp.data_ptr(), bh.data_ptr(), ch.data_ptr(), ksh, C, H, W, H, w,
                                  0, 0, hip.stream()), "sisr_pil_resample(h)")
    out = torch.empty((C, h, w), device=dev, dtype=torch.float32 if to_float else torch.uint8)
    hip.check(L.sisr_pil_resample(tmp.data_ptr(), out.data_ptr(), bv.data_ptr(), cv.data_ptr(), ksv, C, H, w, h, w, 1,
                                  int(to_float), hip.stream()), "sisr_pil_resample(v)")
    return out


def center_crop_box(height, width, scale):
    """image_manipulation.downsample's crop: (top, left, r_height, r_width), rounding as center_crop does."""
    rh, rw = (height // scale) * scale, (width // scale) * scale
    return int(round((height - rh) / 2.)), int(round((width - rw) / 2.)), rh, rw


class OnlineDegrader:
    """SRMDPreprocessing(pca, random=True[, noise]) + ToPILImage + downsample for one image at a time, as
    SuperResImages.__getitem__ applies it (data_handler.py:446-456); returns device tensors.  With `noise` (ref
    gaussian_utils.py:371-424; the reference's own default for SRMDPreprocessing, off in the dataset's default set-up):
    a noise level per image (random_batch_noise: uniform * noise_high, zero with probability rate_cln), Gaussian noise of
    that level on the BLURRED full-size image, clamped to [0, 1], and 10 * level appended to the kernel code."""

    def __init__(self, scale=4, pca=None, kernel=21, sig_min=0.2, sig_max=4.0, rate_iso=1.0, scaling=3, noise=False,
                 random=True, sig=2.6, para_input=10, rate_cln=0.2, noise_high=0.08, **unused):
        self.noise, self.rate_cln, self.noise_high = bool(noise), rate_cln, noise_high
        self.scale, self.l, self.random, self.sig = int(scale), int(kernel), bool(random), 2.6 if sig is None else sig
        self.sig_min, self.sig_max, self.rate_iso, self.scaling = sig_min, sig_max, rate_iso, scaling
        self.pca = pca if pca is not None else pca_matrix()
        self.para_in = para_input

    def draw_kernel(self):
        if self.random:
            k = random_gaussian_kernel(l=self.l, sig_min=self.sig_min, sig_max=self.sig_max, rate_iso=self.rate_iso,
                                       scaling=self.scaling)
        else:
            k = isotropic_gaussian_kernel(self.l, self.sig)
        return torch.FloatTensor(k)  # the reference converts the float64 kernel to float32 here (:303)

    def __call__(self, hr):
        """hr: (3, H, W) fp32 in [0, 1] on the device -> (lr (3, h, w) fp32 device, code (k,) float32 CPU tensor,
        kernel (l, l) float32 CPU tensor, (top, left, rh, rw) the HR centre crop that matches lr)."""
        kernel = self.draw_kernel()
        code = encode_kernel(kernel, self.pca)
        if self.noise:
            level = torch.FloatTensor(random_batch_noise(1, self.noise_high, self.rate_cln))  # (1, 1) float32, as the reference
            _, blurred = blur_quant(hr, kernel, want_float=True, want_u8=False)
            u8 = noise_quant(blurred, float(level[0, 0]))
            code = torch.cat([code, (level * 10).view(-1)])
        else:
            u8 = blur_quant(hr, kernel)
        top, left, rh, rw = center_crop_box(hr.shape[1], hr.shape[2], self.scale)
        if (rh, rw) != (hr.shape[1], hr.shape[2]):
            u8 = u8[:, top:top + rh, left:left + rw].contiguous()
        return pil_bicubic_downsample(u8, self.scale), code, kernel, (top, left, rh, rw)
