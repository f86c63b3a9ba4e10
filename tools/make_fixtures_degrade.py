#!/usr/bin/env python3
"""Golden vectors of the on-the-fly degradation, produced by RUNNING THE REFERENCE (build container).

    python tools/make_fixtures_degrade.py   ->  tests/golden/d_degrade.npz

As SuperResImages does with online_degradations (data_handler.py:228-238, :446-456): np.random.seed(8); PCA basis from
random kernels (2 000 here instead of 30 000 to keep the run short -- `pca_batch` is stored); SRMDPreprocessing(random,
no noise); per image degrader(ToTensor(hr)) -> to_pil_image -> downsample(scale 4).  Stored: the basis, per image the
drawn kernel, its code and the LR image; plus three anisotropic kernels from a second seed.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as MF  # noqa: E402

from sr_tools import gaussian_utils as G  # noqa: E402
from sr_tools.image_manipulation import downsample  # noqa: E402

NAMES = ["butterfly.png", "woman.png"]
PCA_BATCH = 2000

if __name__ == "__main__":
    from PIL import Image
    np.random.seed(8)
    ker = G.random_batch_kernel(batch=PCA_BATCH, tensor=False)
    pca = G.PCA(ker.reshape((PCA_BATCH, -1)), k=10).float()
    deg = G.SRMDPreprocessing(pca, random=True, kernel=21, rate_iso=1.0, sig_min=0.2, sig_max=4.0, noise=False, cuda=False,
                              noise_high=0.0)
    blob = {"seed": np.array(8), "pca_batch": np.array(PCA_BATCH), "pca": pca.numpy(), "names": np.array(",".join(NAMES))}
    for i, name in enumerate(NAMES):
        hr = np.asarray(Image.open(os.path.join(MF.SET5, "hr", name)).convert("RGB"))
        x = torch.from_numpy(hr.transpose(2, 0, 1).copy()).float().div(255)
        blurred, code, kernel = deg(x)
        pil = G.to_pil_image(blurred.squeeze(0).cpu())
        _, lr = downsample(pil, scale=4, jm=False)
        blob[f"kernel{i}"] = kernel.numpy()[0]
        blob[f"code{i}"] = code.numpy()[0]
        blob[f"lr{i}"] = np.asarray(lr)
        print(name, "lr", np.asarray(lr).shape, "code", code.numpy()[0][:3])
    np.random.seed(11)
    blob["aniso"] = G.random_batch_kernel(batch=3, rate_iso=0.0, tensor=False)
    # noise on (ref gaussian_utils.py:371-424 defaults rate_cln = 0.2, noise_high = 0.08): second seed, 'woman' twice so that
    # both a noisy and (with luck of the draw) a clean access are on record; stored: code (11 values), LR image
    np.random.seed(21)
    degn = G.SRMDPreprocessing(pca, random=True, kernel=21, rate_iso=1.0, sig_min=0.2, sig_max=4.0, noise=True, cuda=False,
                               noise_high=0.08, rate_cln=0.2)
    hr = np.asarray(Image.open(os.path.join(MF.SET5, "hr", "woman.png")).convert("RGB"))
    x = torch.from_numpy(hr.transpose(2, 0, 1).copy()).float().div(255)
    for i in range(3):
        blurred, code, kernel = degn(x)
        _, lr = downsample(G.to_pil_image(blurred.squeeze(0).cpu()), scale=4, jm=False)
        blob[f"n_code{i}"] = code.numpy()[0]
        blob[f"n_lr{i}"] = np.asarray(lr)
        print("noise", i, "code tail", code.numpy()[0][-1])
    np.savez_compressed(os.path.join(MF.OUT, "d_degrade.npz"), **blob)
