"""TEST INFRASTRUCTURE -- CPU restatement of the reference's on-the-fly degradation.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path never does.

ref: Code/sr_tools/gaussian_utils.py:346-368 (BatchBlur), :52-53 (to_pil_image: mul(255).byte()),
     Code/sr_tools/image_manipulation.py:32-53 (downsample: centre crop + PIL BICUBIC resize),
     Code/sr_tools/data_handler.py:446-456 (the order they are applied in).
The bicubic resize is done by Pillow itself -- the third-party library the reference calls (present on the GPU box
too); `resample_pass_numpy` restates libImaging/Resample.c's 8-bit pass so that the product's host-side coefficient
tables can be checked against Pillow without a GPU.  Pinned by tests/golden/d_degrade.npz (the reference's own
SRMDPreprocessing + downsample outputs, tools/make_fixtures_degrade.py).
"""
import numpy as np
import torch
import torch.nn.functional as F


def batch_blur(x, kernel):
    """BatchBlur.forward for one (l, l) kernel shared by the batch: reflection pad, per-channel correlation."""
    B, C, H, W = x.shape
    l = kernel.shape[-1]
    pad = (l // 2, l // 2, l // 2, l // 2) if l % 2 == 1 else (l // 2, l // 2 - 1, l // 2, l // 2 - 1)
    p = F.pad(x, pad, mode="reflect")
    Hp, Wp = p.shape[-2:]
    return F.conv2d(p.view(C * B, 1, Hp, Wp), kernel.contiguous().view(1, 1, l, l), padding=0).view(B, C, H, W)


def to_u8(blurred):
    """ToPILImage's quantisation of a float CHW tensor: mul(255).byte() -> HWC uint8."""
    return np.transpose(blurred.mul(255).byte().numpy(), (1, 2, 0))


def center_crop_u8(img, scale):
    H, W = img.shape[:2]
    rh, rw = (H // scale) * scale, (W // scale) * scale
    top, left = int(round((H - rh) / 2.)), int(round((W - rw) / 2.))
    return img[top:top + rh, left:left + rw]


def pil_downsample(img_u8_hwc, scale):
    """image_manipulation.downsample: centre crop to a multiple of the scale, PIL BICUBIC resize -> HWC uint8."""
    from PIL import Image
    im = Image.fromarray(np.ascontiguousarray(center_crop_u8(img_u8_hwc, scale)), mode="RGB")
    return np.asarray(im.resize((im.width // scale, im.height // scale), resample=Image.BICUBIC))


def degrade(hr_chw, kernel, scale):
    """HR float CHW in [0, 1] + (l, l) float32 kernel -> LR float CHW, as data_handler.py:446-456 + lr_transform."""
    blurred = batch_blur(hr_chw[None], torch.as_tensor(kernel, dtype=torch.float32))[0]
    lr = pil_downsample(to_u8(blurred), scale)
    return torch.from_numpy(np.ascontiguousarray(lr.transpose(2, 0, 1))).float().div(255)


def resample_pass_numpy(img_u8, bounds, coef, vertical):
    """One 8-bit pass of libImaging/Resample.c (ImagingResampleHorizontal_8bpc / Vertical_8bpc) on an (H, W, C)
    image with host tables (bounds [out][2], coef [out][ksize]): int32 accumulate from 1 << 21, >> 22, clip."""
    a = img_u8.astype(np.int64)
    if vertical:
        a = a.transpose(1, 0, 2)
    out = np.zeros((a.shape[0], len(bounds), a.shape[2]), np.int64)
    for o, (lo, n) in enumerate(bounds):
        ss = np.full((a.shape[0], a.shape[2]), 1 << 21, np.int64)
        for k in range(n):
            ss += a[:, lo + k, :] * int(coef[o, k])
        out[:, o, :] = np.clip(ss >> 22, 0, 255)
    out = out.astype(np.uint8)
    return out.transpose(1, 0, 2) if vertical else out
