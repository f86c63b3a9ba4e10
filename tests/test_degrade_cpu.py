"""Host side of the on-the-fly degradation against Pillow and the reference's own outputs (CPU)."""
import os

import numpy as np
import pytest
import torch

import sisr_amd
from conftest import GOLDEN
from oracle import degrade_oracle as DO

D = sisr_amd.degrade


@pytest.mark.parametrize("hw,scale", [((64, 96), 4), ((50, 34), 2), ((93, 61), 3), ((256, 256), 4), ((37, 45), 4)])
def test_host_coefficient_tables_reproduce_pillow_bit_for_bit(hw, scale):
    """pil_bicubic_table (precompute_coeffs + normalize_coeffs_8bpc) driven through a numpy restatement of the two 8-bit
    passes equals Image.resize(BICUBIC) exactly -- the tables are what the HIP kernel consumes."""
    rng = np.random.RandomState(hw[0] * 7 + scale)
    img = DO.center_crop_u8(rng.randint(0, 256, size=hw + (3,)).astype(np.uint8), scale)
    H, W = img.shape[:2]
    bh, ch, _ = D.pil_bicubic_table(W, W // scale)
    bv, cv, _ = D.pil_bicubic_table(H, H // scale)
    mine = DO.resample_pass_numpy(DO.resample_pass_numpy(img, bh, ch, False), bv, cv, True)
    np.testing.assert_array_equal(mine, DO.pil_downsample(img, scale))


def test_kernels_codes_and_lr_images_match_the_reference():
    """Fixture d_degrade.npz: the reference's dataset set-up (np.random.seed(8); PCA basis from random kernels) and its
    SRMDPreprocessing + ToPILImage + downsample on two Set5 HR images.  The host mirror reproduces the kernel draws and
    codes from the same seed; the oracle reproduces the LR images from the stored kernels."""
    z = np.load(os.path.join(GOLDEN, "d_degrade.npz"))
    np.random.seed(int(z["seed"]))
    pca = D.pca_matrix(batch=int(z["pca_batch"]))
    np.testing.assert_allclose(pca.numpy(), z["pca"], rtol=0, atol=1e-6)
    deg = D.OnlineDegrader(scale=4, pca=pca)
    from PIL import Image
    for i, name in enumerate(str(z["names"]).split(",")):
        k = deg.draw_kernel()
        np.testing.assert_array_equal(k.numpy(), z[f"kernel{i}"])
        np.testing.assert_allclose(D.encode_kernel(k, pca).numpy(), z[f"code{i}"], rtol=0, atol=1e-7)
        hr = np.asarray(Image.open(os.path.join(GOLDEN, "set5", "hr", name)).convert("RGB"))
        x = torch.from_numpy(hr.transpose(2, 0, 1).copy()).float().div(255)
        lr = DO.degrade(x, k, 4)
        np.testing.assert_array_equal((lr * 255).round().byte().numpy(), z[f"lr{i}"].transpose(2, 0, 1))
    # anisotropic draws (rate_iso = 0), the same stream
    np.random.seed(11)
    ka = D.random_batch_kernel(3, rate_iso=0.0)
    np.testing.assert_allclose(ka, z["aniso"], rtol=0, atol=0)
