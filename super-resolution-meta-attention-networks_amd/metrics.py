"""PSNR on the BT.601 'jpg' luma, the reference's parity metric (host-side numpy).

ref: Code/sr_tools/metrics.py:6-17 (psnr), Code/sr_tools/image_manipulation.py:65-89 (rgb_to_ycbcr 'jpg'),
     Code/SISR/models/__init__.py:158-169 (clip to [0,1] before conversion).
"""
import numpy as np


def psnr(img1, img2, max_value=255.0):
    mse = np.mean((np.array(img1, dtype=np.float32) - np.array(img2, dtype=np.float32)) ** 2)
    if mse == 0:
        return 100
    return 20 * np.log10(max_value / (np.sqrt(mse)))


def rgb_to_ycbcr_jpg(img, max_val=1):
    """C,H,W RGB -> (Y, Cb, Cr), full-range BT.601 without luma offset."""
    bias_c = 128. * (max_val / 255)
    y = 0.299 * img[0] + 0.587 * img[1] + 0.114 * img[2]
    cb = bias_c + (-0.168736 * img[0] - 0.331264 * img[1] + 0.5 * img[2])
    cr = bias_c + (0.5 * img[0] - 0.418688 * img[1] - 0.081312 * img[2])
    return np.array([y, cb, cr])


def standard_image_formatting(im, min_value=0, max_value=1):
    return np.clip(np.copy(im), min_value, max_value)


def batch_rgb_to_ycbcr(batch):
    """(N,3,H,W) in [0,1] (clipped first) -> (N,3,H,W) YCbCr."""
    out = standard_image_formatting(np.asarray(batch))
    for i in range(out.shape[0]):
        out[i] = rgb_to_ycbcr_jpg(out[i])
    return out


def y_psnr(sr, hr, max_value=1):
    """PSNR between the Y channels of a clipped SR image and its HR reference (both C,H,W RGB in [0,1])."""
    a = rgb_to_ycbcr_jpg(standard_image_formatting(np.asarray(sr)))[0]
    b = rgb_to_ycbcr_jpg(standard_image_formatting(np.asarray(hr)))[0]
    return psnr(a, b, max_value=max_value)
