"""Feature counts that are not multiples of 64 (ref: attention_manipulators/handlers.py:24-29 forwards any n_feats;
advanced/architectures.py:126-161, 183-225): the HIP path runs such a network as its zero-padded 64-wide twin
(architectures.ChannelPadded) with the reference's parameter shapes, state-dict keys and checkpoints.

What this puts on the HIP path for the first time: EVERY reduced-net fixture the reference generated at n_feats = 16
(G2: rcan, edsr x4 / x3, han, qhan, qedsr, qrcan standard / modulate / selective + pixel attention), the 32-channel
ResidualGroup and the 16-channel Upsampler x3 / x4 block fixtures (G1), and two new ones at n_feats = 48 (W3,
tools/make_fixtures_wide.py --c48).  CPU: the oracle against W3, module trees / seed-8 weights, the padding plans.
"""
import hashlib

import numpy as np
import pytest
import torch
from torch import nn

import sisr_amd
from conftest import load_golden
from oracle import sisr_oracle as O

A, H = sisr_amd.architectures, sisr_amd.han


def _digest(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def _net(name, m):
    if name.startswith("g2_rcan") or name == "g2p_rcan_c48":
        return A.RCAN(n_resblocks=m["n_resblocks"], n_resgroups=m["n_resgroups"], n_feats=m["n_feats"], scale=m["scale"],
                      reduction=m["reduction"])
    if name.startswith("g2_edsr"):
        return A.EDSR(net_features=m["net_features"], num_blocks=m["num_blocks"], scale=m["scale"], res_scale=m["res_scale"])
    if name == "g2_han":
        return H.HAN(n_resgroups=m["n_resgroups"], n_resblocks=m["n_resblocks"], n_feats=m["n_feats"], reduction=m["reduction"],
                     scale=m["scale"])
    if name == "g2_qhan":
        return H.QHAN(n_resgroups=m["n_resgroups"], n_resblocks=m["n_resblocks"], n_feats=m["n_feats"], reduction=m["reduction"],
                      num_metadata=m["num_metadata"], scale=m["scale"])
    if name == "g2_qedsr":
        return A.QEDSR(num_features=m["num_features"], input_para=m["input_para"], num_blocks=m["num_blocks"], scale=m["scale"],
                       res_scale=m["res_scale"], q_layer_nonlinearity=m["q_layer_nonlinearity"])
    return A.QRCAN(**m)


G2 = {"g2_rcan": 1, "g2_edsr": 1, "g2_edsr_x3": 1, "g2_han": 1, "g2_qhan": 2, "g2_qedsr": 2, "g2_qrcan_standard": 2,
      "g2_qrcan_modulate": 2, "g2_qrcan_selective": 2}
W3 = {"g2p_rcan_c48": 1, "g2p_qrcan_c48": 2}
W3_ORACLE = {
    "g2p_rcan_c48": lambda sd, m, x: O.rcan(sd, x, m["n_resgroups"], m["n_resblocks"], m["scale"]),
    "g2p_qrcan_c48": lambda sd, m, x, a: O.qrcan(sd, x, a, m["n_resgroups"], m["n_resblocks"], m["scale"], m["style"], False,
                                                 m["include_q_layer"]),
}


def _check_light(a, grads, name, rtol_norm, atol_head):
    for k in [k[4:] for k in a if k.startswith("pgn/")]:
        g = grads[k]
        g = torch.zeros(1) if g is None else g.detach().double().cpu()
        n_ref = float(a["pgn/" + k])
        assert abs(float(g.norm()) - n_ref) <= rtol_norm * n_ref + 1e-7, (name, k, float(g.norm()), n_ref)
        np.testing.assert_allclose(g.reshape(-1)[:8].float().numpy(), a["pgh/" + k], rtol=50 * rtol_norm, atol=atol_head,
                                   err_msg=f"{name} {k}")


# ------------------------------------------------------------------------------------------------------------- CPU
@pytest.mark.parametrize("name", sorted(W3))
def test_oracle_matches_the_reference_at_48_features(name):
    a, meta = load_golden(name)
    torch.manual_seed(8)
    net = _net(name, meta)
    assert _digest(net.state_dict()) == str(a["sd_sha256"]), "seed-8 initial weights differ from the reference's"
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    ins = [torch.from_numpy(a[f"in{i}"]).clone().requires_grad_(True) for i in range(W3[name])]
    out = W3_ORACLE[name](sd, meta, *ins)
    np.testing.assert_allclose(out.detach().numpy(), a["out"], rtol=2e-5, atol=4e-6, err_msg=name)
    out.backward(torch.from_numpy(a["cot"]))
    _check_light(a, {k: v.grad for k, v in sd.items() if ("pgn/" + k) in a}, name, 2e-5, 2e-5)


def test_padded_networks_keep_the_reference_parameters_and_know_their_twin():
    for name in sorted(G2) + sorted(W3):
        a, meta = load_golden(name)
        torch.manual_seed(8)
        net = _net(name, meta)
        assert net.padded()
        keys = [k[3:] for k in a if k.startswith("sd/")]
        if keys:  # G2 fixtures store the reference's state dict: same key set, same shapes
            sd = net.state_dict()
            assert set(keys) == set(sd) and all(tuple(sd[k].shape) == a["sd/" + k].shape for k in keys), name
        n, P = net._pad_width
        assert P == 64 and n in (16, 48)
        twin_shapes = {k: tuple(v.shape) for k, v in net._twin.named_parameters()}
        for k, p in net.named_parameters():
            steps, shape = net._pad_plans[k]
            assert shape == twin_shapes[k]
            numel = p.numel()
            for outer, nn_, inner, PP in steps:  # every step pads exactly one axis of the running shape
                assert numel == outer * nn_ * inner, (name, k, steps)
                numel = outer * PP * inner
            assert numel == int(np.prod(shape)), (name, k)
        assert "_twin" not in dict(net.named_children()) and not any("_twin" in k for k in net.state_dict())


def test_metadata_concatenating_styles_refuse_unpadded_widths():
    with pytest.raises(NotImplementedError):
        A.QRCAN(n_resblocks=1, n_resgroups=1, n_feats=48, style="max_concat", num_metadata=10)


# ------------------------------------------------------------------------------------------------------------- GPU
def _run(name, net, n_in, rtol=2e-4, atol=3e-5):
    a, meta = load_golden(name)
    net.to("cuda:0")
    ins = [torch.from_numpy(a[f"in{i}"]).to("cuda:0").requires_grad_(True) for i in range(n_in)]
    out = net(*ins)
    np.testing.assert_allclose(out.detach().cpu().numpy(), a["out"], rtol=rtol, atol=atol, err_msg=name + " out")
    out.backward(torch.from_numpy(a["cot"]).to("cuda:0"))
    for i, t in enumerate(ins):
        if t.requires_grad:
            g = t.grad if t.grad is not None else torch.zeros_like(t)
            np.testing.assert_allclose(g.cpu().numpy(), a[f"gin{i}"], rtol=5e-4, atol=5e-5, err_msg=f"{name} gin{i}")
    return a


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(G2))
def test_hip_meets_the_16_feature_reduced_net_fixtures(name):
    """The reference's own G2 vectors (full state dict, output, every parameter gradient in full), directly on the HIP path."""
    a, meta = load_golden(name)
    net = _net(name, meta)
    net.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in a.items() if k.startswith("sd/")}, strict=True)
    a = _run(name, net, G2[name], rtol=5e-4, atol=5e-5)
    for k, p in net.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        want = a["pg/" + k]
        scale = max(1.0, float(np.abs(want).max()))
        np.testing.assert_allclose(g.cpu().numpy(), want, rtol=2e-3, atol=2e-4 * scale, err_msg=f"{name} pg/{k}")
        assert tuple(g.shape) == want.shape


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(W3))
def test_hip_meets_the_48_feature_fixtures(name):
    a, meta = load_golden(name)
    torch.manual_seed(8)
    net = _net(name, meta)
    assert _digest(net.state_dict()) == str(a["sd_sha256"])
    a = _run(name, net, W3[name])
    _check_light(a, {k: p.grad for k, p in net.named_parameters()}, name, 2e-4, 1e-4)


class _PaddedBlock(A.ChannelPadded, nn.Module):
    """A feature-map block (input and output maps of n channels) run through the same padding machinery as the networks."""

    def __init__(self, build, n, out_mult=1):
        super().__init__()
        self.m = build(n)
        self.n, self.out_mult = n, out_mult
        self._init_padding(n, lambda P: _PaddedBlock(build, P, out_mult))

    def forward(self, x):
        if not self.padded():
            return self.m(x)
        B, n, Hh, Ww = x.shape
        P = self._pad_width[1]
        xp = sisr_amd.ops.pad_param(x.contiguous(), [(B, n, Hh * Ww, P)], (B, P, Hh, Ww))
        return self._run_padded(xp)[:, :n]


@pytest.mark.gpu
def test_hip_meets_the_narrow_block_fixtures():
    """g1_resgroup (32 channels) and g1_upsampler_x3 / x4 (16 channels): the remaining reference block vectors."""
    a, meta = load_golden("g1_resgroup")
    blk = _PaddedBlock(lambda n: A.ResidualGroup(A.default_conv, n, 3, 16, act=nn.ReLU(True), res_scale=1.0,
                                                 n_resblocks=meta["n_resblocks"]), 32)
    blk.m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in a.items() if k.startswith("sd/")}, strict=True)
    _run("g1_resgroup", blk, 1)
    for k, p in blk.m.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), a["pg/" + k], rtol=1e-3, atol=1e-4 * max(1.0, float(np.abs(a["pg/" + k]).max())),
                                   err_msg=f"g1_resgroup pg/{k}")
    for scale in (3, 4):
        name = f"g1_upsampler_x{scale}"
        a, meta = load_golden(name)

        class Up(_PaddedBlock):
            def _init_padding(self, n_feats, build_twin):  # a bare Upsampler's keys carry no 'tail.0.': say which axes are [n][r^2]
                super()._init_padding(n_feats, build_twin, nk=lambda k, d: d == 0)

        up = Up(lambda n: A.Upsampler(A.default_conv, scale, n), 16)
        up.m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in a.items() if k.startswith("sd/")}, strict=True)
        _run(name, up, 1)
        for k, p in up.m.named_parameters():
            np.testing.assert_allclose(p.grad.cpu().numpy(), a["pg/" + k], rtol=1e-3,
                                       atol=1e-4 * max(1.0, float(np.abs(a["pg/" + k]).max())), err_msg=f"{name} pg/{k}")


@pytest.mark.gpu
def test_qrcan_handler_trains_and_checkpoints_at_48_features(tmp_path):
    torch.manual_seed(8)
    kw = dict(scale=2, lr=1e-3, n_feats=48, n_resgroups=2, n_resblocks=2, metadata=["blur_kernel"], style="standard",
              include_q_layer=True)
    h = sisr_amd.available_models["qrcan"](device=0, model_save_dir=str(tmp_path), eval_mode=False, **kw)
    g = torch.Generator().manual_seed(3)
    x, y = torch.rand(2, 3, 20, 24, generator=g), torch.rand(2, 3, 40, 48, generator=g)
    md = torch.rand(2, 10, generator=g, dtype=torch.float64) * 0.4
    keys = [("blur_kernel",) * 2] * 10
    l0 = float(h.run_train(x, y, metadata=md, metadata_keys=keys)[0])
    for _ in range(12):
        l1 = float(h.run_train(x, y, metadata=md, metadata_keys=keys)[0])
    assert np.isfinite(l1) and l1 < l0
    assert h.net.state_dict()["body.0.body.0.body.0.weight"].shape == (48, 48, 3, 3)  # the reference's shapes
    h.save_model("train_model", 1)
    h2 = sisr_amd.available_models["qrcan"](device=0, model_save_dir=str(tmp_path), eval_mode=True, **kw)
    h2.load_model("train_model", 1)
    o1, _, _ = h.run_eval(x, metadata=md, metadata_keys=keys)
    o2, _, _ = h2.run_eval(x, metadata=md, metadata_keys=keys)
    assert torch.equal(o1, o2)
