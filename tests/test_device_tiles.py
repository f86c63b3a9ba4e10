"""Device-side training tiles (data.DeviceTileSampler + sisr_crop_augment) against the host path that mirrors the
reference's random_flip_rotate -> random_matched_crop (sr_tools/image_manipulation.py:233-257,
data_handler.py:500-513): same `random` stream in, same tiles out, bit for bit."""
import random

import pytest
import torch

import sisr_amd
from test_init_parity import set5

data = sisr_amd.data


def host_tiles(lr_images, hr_images, indices, crop, scale, seed):
    random.seed(seed)
    lrs, hrs = [], []
    for i in indices:
        lr, hr = data.random_flip_rotate(lr_images[i], hr_images[i])
        lr, hr = data.random_matched_crop(lr, hr, crop_size=crop, scale=scale)
        lrs.append(lr)
        hrs.append(hr)
    return torch.stack(lrs), torch.stack(hrs)


def gather_on_host(img, rec, crop):
    """The kernel's index map, evaluated with torch on the CPU."""
    H, W, top, left, hf, vf, rot = rec
    i = torch.arange(crop).view(crop, 1).expand(crop, crop)
    j = torch.arange(crop).view(1, crop).expand(crop, crop)
    y, x = top + i, left + j
    y2, x2 = (x, y) if rot else (y, x)
    sy = (H - 1 - y2) if vf else y2
    sx = (W - 1 - x2) if hf else x2
    return img[:, sy, sx]


def _images():
    lrs, hrs = [], []
    for _, x, y, _ in set5():
        lrs.append(x[0].clone())
        hrs.append(y[0].clone())
    return lrs, hrs


def test_draw_order_and_index_map_match_the_host_path():
    lrs, hrs = _images()
    idx = [0, 3, 1, 4, 2, 2, 0]
    crop, scale = 24, 4
    want_lr, want_hr = host_tiles(lrs, hrs, idx, crop, scale, seed=11)
    s = data.DeviceTileSampler(lrs, hrs, scale=scale, crop=crop, device="cpu")
    random.seed(11)
    for n, i in enumerate(idx):
        top, left, hf, vf, rot = s.draw(i)
        _, h, w = lrs[i].shape
        got_lr = gather_on_host(lrs[i], (h, w, top, left, hf, vf, rot), crop)
        got_hr = gather_on_host(hrs[i], (h * scale, w * scale, top * scale, left * scale, hf, vf, rot), crop * scale)
        assert torch.equal(got_lr, want_lr[n]) and torch.equal(got_hr, want_hr[n]), (n, i)
    assert random.random() == (random.seed(11), [s.draw(i) for i in idx], random.random())[2]  # same stream position


def test_rejects_mismatched_or_small_images():
    lrs, hrs = _images()
    with pytest.raises(ValueError):
        data.DeviceTileSampler(lrs, hrs[:-1], scale=4, crop=16, device="cpu")
    with pytest.raises(ValueError):
        data.DeviceTileSampler(lrs, hrs, scale=4, crop=4096, device="cpu")
    with pytest.raises(ValueError):
        data.DeviceTileSampler(lrs, [h[:, :-4] for h in hrs], scale=4, crop=16, device="cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("augment", [True, False])
def test_device_tiles_equal_host_tiles(augment):
    lrs, hrs = _images()
    idx = [4, 0, 2, 2, 1, 3, 0, 4]
    crop, scale = 32, 4
    s = data.DeviceTileSampler(lrs, hrs, scale=scale, crop=crop, device="cuda:0", augment=augment)
    random.seed(5)
    if augment:
        want_lr, want_hr = host_tiles(lrs, hrs, idx, crop, scale, seed=5)
    else:
        want = [data.random_matched_crop(lrs[i], hrs[i], crop_size=crop, scale=scale) for i in idx]
        want_lr, want_hr = torch.stack([w[0] for w in want]), torch.stack([w[1] for w in want])
    random.seed(5)
    lr, hr = s.sample(idx)
    assert lr.shape == (8, 3, crop, crop) and hr.shape == (8, 3, crop * scale, crop * scale)
    assert torch.equal(lr.cpu(), want_lr) and torch.equal(hr.cpu(), want_hr)


@pytest.mark.gpu
def test_training_step_from_device_tiles():
    """A step fed by the device sampler equals the step fed by the host path (same tiles -> same loss)."""
    lrs, hrs = _images()
    idx = [0, 1, 2, 3]
    losses = []
    for mode in ("host", "device"):
        torch.manual_seed(8)
        h = sisr_amd.available_models["edsr"](device=0, model_save_dir="/tmp", eval_mode=False, scale=4, lr=1e-4)
        if mode == "host":
            x, y = host_tiles(lrs, hrs, idx, 32, 4, seed=9)
        else:
            random.seed(9)
            x, y = data.DeviceTileSampler(lrs, hrs, scale=4, crop=32, device="cuda:0").sample(idx)
        loss, _ = h.run_train(x, y, keep_on_device=True)
        losses.append(float(loss))
    assert losses[0] == losses[1], losses
