"""On-the-fly degradation kernels (csrc/degrade.hip) on a real MI355X (pytest -m gpu)."""
import os

import numpy as np
import pytest
import torch

import sisr_amd
from conftest import GOLDEN
from oracle import degrade_oracle as DO

pytestmark = pytest.mark.gpu
D = sisr_amd.degrade


@pytest.mark.parametrize("hw,scale", [((64, 96), 4), ((50, 34), 2), ((93, 61), 3), ((512, 384), 4), ((37, 45), 4)])
def test_pil_bicubic_downsample_is_bit_identical_to_pillow(hw, scale):
    rng = np.random.RandomState(hw[0] + scale)
    img = DO.center_crop_u8(rng.randint(0, 256, size=hw + (3,)).astype(np.uint8), scale)
    u8 = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).cuda()
    got = D.pil_bicubic_downsample(u8, scale, to_float=False).cpu().numpy().transpose(1, 2, 0)
    want = DO.pil_downsample(img, scale)
    np.testing.assert_array_equal(got, want)
    gotf = D.pil_bicubic_downsample(u8, scale).cpu().numpy().transpose(1, 2, 0)
    np.testing.assert_array_equal(gotf, want.astype(np.float32) / np.float32(255))


@pytest.mark.parametrize("hw,l", [((40, 52), 21), ((33, 47), 21), ((64, 64), 15), ((30, 30), 8)])
def test_blur_matches_batchblur(hw, l):
    g = torch.Generator().manual_seed(l + hw[0])
    x = torch.rand(3, *hw, generator=g)
    k = torch.rand(l, l, generator=g)
    k = k / k.sum()
    want = DO.batch_blur(x[None], k)[0]
    u8, yf = D.blur_quant(x.cuda(), k, want_float=True)
    np.testing.assert_allclose(yf.cpu().numpy(), want.numpy(), rtol=0, atol=2e-6)
    diff = np.abs(u8.cpu().numpy().astype(int) - want.mul(255).byte().numpy().astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3  # truncation ties only (summation order)


def test_online_degrader_reproduces_the_reference_lr_images():
    """Same seed, same Set5 HR images as tools/make_fixtures_degrade.py: kernels and codes equal, LR images equal to the
    reference's except where the fp32 blur lands within rounding of a byte boundary (<= 1 LSB on < 0.2 % of pixels)."""
    z = np.load(os.path.join(GOLDEN, "d_degrade.npz"))
    np.random.seed(int(z["seed"]))
    pca = D.pca_matrix(batch=int(z["pca_batch"]))
    deg = D.OnlineDegrader(scale=4, pca=pca)
    from PIL import Image
    for i, name in enumerate(str(z["names"]).split(",")):
        hr = np.asarray(Image.open(os.path.join(GOLDEN, "set5", "hr", name)).convert("RGB"))
        x = torch.from_numpy(hr.transpose(2, 0, 1).copy()).float().div(255).cuda()
        lr, code, kernel, box = deg(x)
        np.testing.assert_array_equal(kernel.numpy(), z[f"kernel{i}"])
        np.testing.assert_allclose(code.numpy(), z[f"code{i}"], rtol=0, atol=1e-7)
        want = z[f"lr{i}"].transpose(2, 0, 1).astype(int)
        got = (lr.cpu() * 255).round().numpy().astype(int)
        assert got.shape == want.shape and box[2:] == (want.shape[1] * 4, want.shape[2] * 4)
        d = np.abs(got - want)
        assert d.max() <= 1 and (d > 0).mean() < 2e-3, (name, d.max(), (d > 0).mean())


def test_dataset_with_online_degradations_yields_the_reference_batch_schema(monkeypatch):
    """SuperResImages(online_degradations=True) over the Set5 HR images: same np.random stream as a hand-driven degrader
    (PCA basis at construction, one kernel per item), LR = degraded HR, metadata = the kernel code, keys = 10 x
    'blur_kernel', 'blur_kernels' = the 21 x 21 kernel (ref: data_handler.py:222-238, :293-297, :446-456)."""
    real = D.pca_matrix
    monkeypatch.setattr(D, "pca_matrix", lambda batch=2000, k=10: real(batch=2000, k=k))
    hr_dir = os.path.join(GOLDEN, "set5", "hr")
    np.random.seed(5)
    ds = sisr_amd.data.SuperResImages(hr_dir=hr_dir, online_degradations=True, split="all", scale=4)
    items = [ds[i] for i in range(2)]
    np.random.seed(5)
    deg = D.OnlineDegrader(scale=4)
    from PIL import Image
    for i, it in enumerate(items):
        hr = np.asarray(Image.open(os.path.join(hr_dir, ds.base_filenames[i])).convert("RGB"))
        x = torch.from_numpy(hr.transpose(2, 0, 1).copy()).float().div(255)
        lr, code, kernel, (top, left, rh, rw) = deg(x.cuda())
        assert torch.equal(it["lr"], lr.cpu()) and torch.equal(it["hr"], x[:, top:top + rh, left:left + rw])
        np.testing.assert_array_equal(it["metadata"], code.numpy())
        np.testing.assert_array_equal(it["blur_kernels"], kernel.numpy())
        assert it["metadata_keys"] == ["blur_kernel"] * 10 and it["tag"] == it["hr_tag"] == ds.base_filenames[i]
        assert it["hr"].shape[1] == 4 * it["lr"].shape[1] and it["hr"].shape[2] == 4 * it["lr"].shape[2]


def test_dataset_with_online_degradations_and_a_metadata_file(monkeypatch, tmp_path):
    """ref: data_handler.py:264-297, :451-452: with online degradations the metadata file is looked up by HR name, a sample's
    vector is [file metadata | kernel code], and the keys are the file's keys followed by the kernel keys AS ONE NESTED LIST
    (the reference appends the list itself).  Same through the device tile loader, batched as default_collate would."""
    import pandas as pd
    real = D.pca_matrix
    monkeypatch.setattr(D, "pca_matrix", lambda batch=2000, k=10: real(batch=2000, k=k))
    hr_dir = os.path.join(GOLDEN, "set5", "hr")
    names = sorted(os.listdir(hr_dir))
    meta_file = tmp_path / "meta.csv"
    pd.DataFrame({"QPI": [20 + 4 * i for i in range(len(names))]}, index=names).to_csv(meta_file)
    np.random.seed(5)
    ds = sisr_amd.data.SuperResImages(hr_dir=hr_dir, online_degradations=True, split="all", scale=4,
                                      degradation_metadata_file=str(meta_file), random_crop=24)
    assert ds.metadata_keys == ["qpi", ["blur_kernel"] * 10]
    np.random.seed(7)
    it = ds[1]
    np.random.seed(7)
    code = ds.degrader(torch.zeros(3, 64, 64).cuda())[1].numpy()  # the kernel draw only depends on the np.random stream
    want_q = (20 + 4 * names.index(ds.base_filenames[1]) - 20) / 20.0
    assert it["metadata"].shape == (11,) and abs(float(it["metadata"][0]) - want_q) < 1e-12
    np.testing.assert_array_equal(it["metadata"][1:], code)
    loader = sisr_amd.data.DeviceTileLoader([ds], batch_size=2, device=torch.device("cuda:0"))
    batch = next(iter(loader))
    assert tuple(batch["metadata"].shape) == (2, 11) and batch["metadata_keys"][0] == ("qpi", "qpi")
    assert batch["metadata_keys"][1] == [("blur_kernel", "blur_kernel")] * 10
    for row, tag in zip(batch["metadata"], batch["tag"]):
        assert abs(float(row[0]) - 4 * names.index(tag) / 20.0) < 1e-12


def test_online_degrader_with_noise_reproduces_the_reference():
    """ref: gaussian_utils.py:299-312, 371-424 (SRMDPreprocessing(noise=True, noise_high=0.08, rate_cln=0.2)): same seed and
    image as the fixture -- noise levels (one access drew a clean image, two a noisy one), 11-value codes and LR images."""
    z = np.load(os.path.join(GOLDEN, "d_degrade.npz"))
    np.random.seed(int(z["seed"]))
    pca = D.pca_matrix(batch=int(z["pca_batch"]))
    deg = D.OnlineDegrader(scale=4, pca=pca, noise=True, noise_high=0.08, rate_cln=0.2)
    from PIL import Image
    hr = np.asarray(Image.open(os.path.join(GOLDEN, "set5", "hr", "woman.png")).convert("RGB"))
    x = torch.from_numpy(hr.transpose(2, 0, 1).copy()).float().div(255).cuda()
    np.random.seed(21)
    tails = []
    for i in range(3):
        lr, code, kernel, box = deg(x)
        assert code.shape == (11,)
        np.testing.assert_allclose(code.numpy(), z[f"n_code{i}"], rtol=0, atol=1e-6)
        tails.append(float(code[-1]))
        want = z[f"n_lr{i}"].transpose(2, 0, 1).astype(int)
        got = (lr.cpu() * 255).round().numpy().astype(int)
        d = np.abs(got - want)
        assert got.shape == want.shape and d.max() <= 1 and (d > 0).mean() < 3e-3, (i, d.max(), (d > 0).mean())
    assert min(tails) == 0.0 and max(tails) > 0.0  # both branches of the noise mask are on record


def test_device_tile_loader_with_online_degradations_equals_the_dataloader_path(monkeypatch):
    """`device_tiles` + `online_degradations`: HR images resident on the device, every batch degraded and cut there.  Under
    the same seeds (torch: shuffle; numpy: kernels; random: flips / crops) the batches equal those of the reference-shaped
    path, DataLoader(shuffle=True, num_workers=0) over SuperResImages.__getitem__ (ref: data_handler.py:446-456, :500-525)."""
    import random
    from torch.utils.data import DataLoader
    real = D.pca_matrix
    monkeypatch.setattr(D, "pca_matrix", lambda batch=2000, k=10: real(batch=2000, k=k))
    hr_dir = os.path.join(GOLDEN, "set5", "hr")

    def dataset():
        np.random.seed(5)
        return sisr_amd.data.SuperResImages(hr_dir=hr_dir, online_degradations=True, split="all", scale=4, random_crop=24,
                                            random_augments=True)

    def seed():
        torch.manual_seed(3)
        np.random.seed(6)
        random.seed(7)

    host_loader = DataLoader(dataset=dataset(), batch_size=2, shuffle=True, num_workers=0)
    dev_loader = sisr_amd.data.DeviceTileLoader([dataset()], 2, torch.device("cuda:0"))
    seed()
    host = list(host_loader)
    seed()
    dev = list(dev_loader)
    assert len(host) == len(dev) == 3
    for a, b in zip(host, dev):
        assert list(a["tag"]) == list(b["tag"]) and list(a["hr_tag"]) == list(b["hr_tag"])
        assert torch.equal(a["lr"], b["lr"].cpu()) and torch.equal(a["hr"], b["hr"].cpu())
        assert torch.equal(a["metadata"], b["metadata"]) and a["metadata"].dtype == b["metadata"].dtype
        assert torch.equal(a["blur_kernels"], b["blur_kernels"])
        assert [tuple(k) for k in a["metadata_keys"]] == [tuple(k) for k in b["metadata_keys"]]
