"""Pin oracle/sisr_oracle.py to golden vectors produced by the reference itself
(tools/make_fixtures.py).  CPU only."""
import csv
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_json, load_golden
from oracle import sisr_oracle as O

ATOL = 2e-6  # same ATen kernels, different call structure (mean vs adaptive pool, view vs reshape)
RTOL = 2e-5


def _run(name, fn, n_inputs, nonleaf=()):
    a, meta = load_golden(name)
    sd = {k[3:]: torch.from_numpy(v).clone().requires_grad_(True) for k, v in a.items() if k.startswith("sd/")}
    ins = [torch.from_numpy(a[f"in{i}"]).clone().requires_grad_(True) for i in range(n_inputs)]
    out = fn(sd, meta, *ins)
    np.testing.assert_allclose(out.detach().numpy(), a["out"], rtol=RTOL, atol=ATOL, err_msg=name + " out")
    out.backward(torch.from_numpy(a["cot"]))
    for i, t in enumerate(ins):
        g = t.grad if t.grad is not None else torch.zeros_like(t)
        np.testing.assert_allclose(g.numpy(), a[f"gin{i}"], rtol=1e-4, atol=5e-6, err_msg=f"{name} gin{i}")
    pg = {k[3:]: v for k, v in a.items() if k.startswith("pg/")}
    assert set(pg) == set(sd), "state_dict key set differs from the reference's"
    for k, v in pg.items():
        g = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        np.testing.assert_allclose(g.numpy(), v, rtol=2e-4, atol=2e-5, err_msg=f"{name} pg/{k}")


# name -> (callable(sd, meta, *inputs), number of inputs)
def _wrap(key_fn):
    return key_fn


BLOCKS = {
    "g1_conv64": (lambda sd, m, x: O.conv({"c." + k: v for k, v in sd.items()}, "c", x), 1),
    "g1_conv64_odd": (lambda sd, m, x: O.conv({"c." + k: v for k, v in sd.items()}, "c", x), 1),
    "g1_conv_head": (lambda sd, m, x: O.conv({"c." + k: v for k, v in sd.items()}, "c", x), 1),
    "g1_conv_tail": (lambda sd, m, x: O.conv({"c." + k: v for k, v in sd.items()}, "c", x), 1),
    "g1_calayer": (lambda sd, m, x: O.ca_layer({"b." + k: v for k, v in sd.items()}, "b", x), 1),
    "g1_rcab": (lambda sd, m, x: O.rcab({"b." + k: v for k, v in sd.items()}, "b", x), 1),
    "g1_rcab_odd": (lambda sd, m, x: O.rcab({"b." + k: v for k, v in sd.items()}, "b", x), 1),
    "g1_resblock": (lambda sd, m, x: O.res_block({"b." + k: v for k, v in sd.items()}, "b", x, m["res_scale"]), 1),
    "g1_resgroup": (lambda sd, m, x: O.residual_group({"b." + k: v for k, v in sd.items()}, "b", x,
                                                      m["n_resblocks"]), 1),
    "g1_upsampler_x4": (lambda sd, m, x: O.upsampler({"u." + k: v for k, v in sd.items()}, "u", x, 4), 1),
    "g1_upsampler_x3": (lambda sd, m, x: O.upsampler({"u." + k: v for k, v in sd.items()}, "u", x, 3), 1),
    "g1_palayer": (lambda sd, m, x: O.pa_layer({"b." + k: v for k, v in sd.items()}, "b", x), 1),
    "g1_lam": (lambda sd, m, x: O.lam_module({"b." + k: v for k, v in sd.items()}, "b", x), 1),
    "g1_csam": (lambda sd, m, x: O.csam_module({"b." + k: v for k, v in sd.items()}, "b", x), 1),
    "g1_csam_c64": (lambda sd, m, x: O.csam_module({"b." + k: v for k, v in sd.items()}, "b", x), 1),
}
for M in (1, 10, 11, 20):
    for nl in (0, 1):
        BLOCKS[f"g1_paraca_m{M}_nl{nl}"] = (
            lambda sd, m, x, a: O.para_ca_layer({"b." + k: v for k, v in sd.items()}, "b", x, a, m["nonlinearity"]), 2)
for style in ("standard", "modulate", "mini_concat", "max_concat", "softmax", "extended_attention"):
    BLOCKS[f"g1_qca_{style}"] = (
        lambda sd, m, x, a: O.qca_layer({"b." + k: v for k, v in sd.items()}, "b", x, a, m["style"]), 2)
for q in (0, 1):
    for pa in (0, 1):
        BLOCKS[f"g1_qrcab_q{q}_pa{pa}"] = (
            lambda sd, m, x, a: O.qrcab({"b." + k: v for k, v in sd.items()}, "b", x, a, m["style"], m["pa"],
                                        m["q_layer"]), 2)
for nl in (0, 1):
    BLOCKS[f"g1_paramresblock_nl{nl}"] = (
        lambda sd, m, x, a: O.param_res_block({"b." + k: v for k, v in sd.items()}, "b", x, a, m["res_scale"],
                                              m["nonlinearity"]), 2)


@pytest.mark.parametrize("name", sorted(BLOCKS))
def test_g1_block(name):
    fn, n = BLOCKS[name]
    _run(name, fn, n)


def test_para_ca_widths_follow_reference_shapes():
    for M in (1, 10, 11, 20):
        a, _ = load_golden(f"g1_paraca_m{M}_nl0")
        w = O.para_ca_widths(64, M)
        assert a["sd/attribute_integrator.0.weight"].shape[:2] == (w[1], w[0])
        assert a["sd/attribute_integrator.1.weight"].shape[:2] == (w[2], w[1])


NETS = {
    "g2_rcan": lambda sd, m, x: O.rcan(sd, x, m["n_resgroups"], m["n_resblocks"], m["scale"]),
    "g2_edsr": lambda sd, m, x: O.edsr(sd, x, m["num_blocks"], m["scale"], m["res_scale"]),
    "g2_edsr_x3": lambda sd, m, x: O.edsr(sd, x, m["num_blocks"], m["scale"], m["res_scale"]),
    "g2_han": lambda sd, m, x: O.han(sd, x, m["n_resgroups"], m["n_resblocks"], m["scale"]),
}
QNETS = {
    "g2_qrcan_standard": lambda sd, m, x, a: O.qrcan(sd, x, a, m["n_resgroups"], m["n_resblocks"], m["scale"],
                                                     m["style"], False, m["include_q_layer"]),
    "g2_qrcan_modulate": lambda sd, m, x, a: O.qrcan(sd, x, a, m["n_resgroups"], m["n_resblocks"], m["scale"],
                                                     m["style"], False, m["include_q_layer"]),
    "g2_qrcan_selective": lambda sd, m, x, a: O.qrcan(sd, x, a, m["n_resgroups"], m["n_resblocks"], m["scale"],
                                                      m["style"], m["include_pixel_attention"], m["include_q_layer"],
                                                      m["selective_meta_blocks"], m["num_q_layers_inner_residual"]),
    "g2_qedsr": lambda sd, m, x, a: O.qedsr(sd, x, a, m["num_blocks"], m["scale"], m["res_scale"],
                                            m["q_layer_nonlinearity"]),
    "g2_qhan": lambda sd, m, x, a: O.qhan(sd, x, a, m["n_resgroups"], m["n_resblocks"], m["scale"]),
}


@pytest.mark.parametrize("name", sorted(NETS))
def test_g2_net(name):
    _run(name, NETS[name], 1)


@pytest.mark.parametrize("name", sorted(QNETS))
def test_g2_meta_net(name):
    _run(name, QNETS[name], 2)


def test_g6_generate_channels():
    a, _ = load_golden("g6_generate_channels")
    md = a["md"]
    keys = [("qpi",) * 3, ("other",) * 3] + [("blur_kernel",) * 3] * 10
    got = O.generate_channels(3, md, keys, ["blur_kernel"], O.num_metadata(["blur_kernel"]))
    np.testing.assert_array_equal(got.numpy(), a["blur_only"])
    got = O.generate_channels(3, md, keys, ["qpi", "blur_kernel"], O.num_metadata(["qpi", "blur_kernel"]))
    np.testing.assert_array_equal(got.numpy(), a["qpi_and_blur"])
    qpi = O.generate_channels(3, md[:, :1], [("qpi",) * 3], ["qpi"], O.num_metadata(None))
    np.testing.assert_allclose(O.scale_qpi(qpi).numpy(), a["modulate_qpi"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(O.scale_qpi(qpi, clamp=True).numpy(), a["modulate_qpi_clamp"], rtol=1e-6, atol=1e-7)
    with pytest.raises(RuntimeError):
        O.generate_channels(3, None, keys, ["blur_kernel"], 10)


def test_num_metadata_rules():
    assert O.num_metadata(None) == 1
    assert O.num_metadata(["blur_kernel"]) == 10
    assert O.num_metadata(["qpi", "blur_kernel"]) == 11
    assert O.num_metadata(["unmodified_blur_kernel"]) == 441
    assert O.num_metadata(["all"]) == 40


def _set5():
    from PIL import Image
    d = os.path.join(GOLDEN, "set5")
    with open(os.path.join(d, "lr_random_blur", "degradation_metadata.csv")) as f:
        rows = {r["image"]: json.loads(r["blur_kernel"]) for r in csv.DictReader(f)}
    for name in sorted(rows):
        lr = np.asarray(Image.open(os.path.join(d, "lr_random_blur", name)).convert("RGB"))
        hr = np.asarray(Image.open(os.path.join(d, "hr", name)).convert("RGB"))
        yield name, lr, hr, rows[name]


def test_g7_psnr_and_luma():
    from PIL import Image
    ref = golden_json("g7_psnr")
    for name, lr, hr, _ in _set5():
        up = np.asarray(Image.fromarray(lr).resize((hr.shape[1], hr.shape[0]), resample=Image.BICUBIC))
        a = up.transpose(2, 0, 1).astype(np.float32) / 255
        b = hr.transpose(2, 0, 1).astype(np.float32) / 255
        assert abs(O.y_psnr(a, b) - ref[name]["y_psnr_bicubic"]) < 1e-4
        assert abs(O.psnr(a, b, 1) - ref[name]["rgb_psnr_bicubic"]) < 1e-4
        assert abs(float(O.rgb_to_y(b).mean()) - ref[name]["y_mean_hr"]) < 1e-6
        assert O.psnr(b, b, 1) == 100
