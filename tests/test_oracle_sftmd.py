"""SFTMD: the oracle and the host-side mirror against the reference's own vectors (CPU).

Fixtures: tools/make_fixtures_sftmd.py ran the reference (SFTMD_variants/architectures.py:110-176 through SFTMDHandler,
SFTMD_variants/handlers.py:6-60): f1 reduced net (weights = seed-8 init, pinned by digest), f2 full-depth init digest +
Set5 run_eval, f3 five run_train steps."""
import numpy as np
import pytest
import torch

import sisr_amd
from conftest import golden_json, load_golden
from oracle import sisr_oracle as O
from test_init_parity import digest, set5

PARAMS = {"metadata": ["blur_kernel"], "num_blocks": 16, "num_features": 64, "in_nc": 3}


def build(eval_mode=True, **extra):
    torch.manual_seed(8)
    return sisr_amd.available_models["sftmd"](device=torch.device("cpu"), model_save_dir="/tmp", eval_mode=eval_mode,
                                              scale=4, **PARAMS, **extra)


def reduced_net():
    torch.manual_seed(8)
    return sisr_amd.sftmd.SFTMD(in_nc=3, num_features=64, num_blocks=2, scale=4, input_para=10)


def test_f1_reduced_net_init_output_and_gradients():
    a, meta = load_golden("f1_sftmd_reduced")
    net = reduced_net()
    assert digest(net.state_dict()) == str(a["sd_sha256"]), "seed-8 init differs from the reference's"
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    out = O.sftmd(sd, torch.from_numpy(a["in0"]), torch.from_numpy(a["in1"]), num_blocks=meta["num_blocks"], scale=meta["scale"])
    np.testing.assert_allclose(out.detach().numpy(), a["out"], rtol=1e-5, atol=1e-6)
    out.backward(torch.from_numpy(a["cot"]))
    for k, v in sd.items():
        assert abs(float(v.grad.double().norm()) / float(a["pgn/" + k]) - 1) < 1e-4, k
        np.testing.assert_allclose(v.grad.reshape(-1)[:32].numpy(), a["pg32/" + k], rtol=1e-3,
                                   atol=2e-5 * float(a["pgn/" + k]) / np.sqrt(v.numel()) + 1e-9, err_msg=k)


def test_f2_init_keys_and_digest_match_the_reference():
    ref = golden_json("f_sftmd")["full_depth"]
    h = build()
    sd = h.net.state_dict()
    assert list(sd) == ref["keys"] and len(sd) == ref["n_tensors"]
    assert int(sum(p.numel() for p in h.net.parameters())) == ref["n_params"]
    assert digest(sd) == ref["sha256"]
    assert h.model_name == "sftmd" and h.channel_concat is False and h.colorspace == "augmented_rgb" and h.num_metadata == 10


def test_f2_oracle_set5_forward_matches_the_reference():
    ref = golden_json("f_sftmd")["full_depth"]["images"]
    crops = np.load(f"{sisr_amd.__path__[0]}/../tests/golden/f2_sftmd_crops.npz")
    h = build()
    sd = h.net.state_dict()
    with torch.no_grad():
        for im, x, y, md in set5():
            if im not in ref:
                continue
            maps = h.generate_channels(x, md, [("blur_kernel",)] * 10)
            assert maps.shape == (1, 10, x.shape[2], x.shape[3])
            out = O.sftmd(sd, x, maps)[0].numpy()
            assert abs(O.y_psnr(out, y[0].numpy()) - ref[im]["y_psnr"]) < 1e-3, im
            assert abs(float(np.abs(out - y[0].numpy()).mean()) - ref[im]["l1"]) < 1e-6
            hh, ww = out.shape[1:]
            np.testing.assert_allclose(out[:, hh // 2 - 16:hh // 2 + 16, ww // 2 - 16:ww // 2 + 16], crops[im],
                                       rtol=1e-4, atol=1e-5)


def test_f3_oracle_trajectory_matches_the_reference():
    ref = golden_json("f_sftmd")["train_steps"]
    h = build(eval_mode=False)
    tr = O.Trainer("sftmd", h.net.state_dict(), lr=1e-4, scheduler=ref["scheduler"], scheduler_params=ref["scheduler_params"])
    g = torch.Generator().manual_seed(77)
    for step in ref["steps"]:
        x, y = torch.rand(2, 3, 16, 16, generator=g), torch.rand(2, 3, 64, 64, generator=g)
        md = torch.rand(2, 10, generator=g, dtype=torch.float64) * 0.4
        loss, out, gn = tr.step(x, y, O.sft_channels(x, md))
        assert abs(loss - step["loss"]) < 2e-6 and abs(gn / step["grad_norm"] - 1) < 2e-4
        assert abs(float(out.mean()) - step["out_mean"]) < 1e-5 and abs(tr.lr - step["lr_after"]) < 1e-12
    assert abs(float(sum(v.double().sum() for v in tr.sd.values())) - ref["final_param_sum"]) < 1e-3


VARIANTS = {  # tools/make_fixtures_sftmd.py VARIANTS: SFTMD kwargs, metadata as vectors (q_injection) instead of maps
    "concat": (dict(SFT_type="concat", input_para=10), False),
    "weak1": (dict(SFT_type="weak", input_para=1), False),
    "none_q": (dict(SFT_type="none", q_injection=True, q_layers=2, input_para=10), True),
    "maskpara_q3": (dict(mask_para=True, q_injection=True, q_layers=3, input_para=10), True),
    "repeats3": (dict(repeats=3, input_para=10), False),
    "concat_input": (dict(input_para=10, in_nc=13), False),  # concat_strategy: maps also concatenated to the RGB input
}


def variant_net(name):
    kw, vector = VARIANTS[name]
    torch.manual_seed(8)
    return sisr_amd.sftmd.SFTMD(num_features=64, num_blocks=2, scale=2, **{"in_nc": 3, **kw}), kw, vector


def variant_inputs(a, name, vector):
    x, md = torch.from_numpy(a[f"{name}/in0"]), torch.from_numpy(a[f"{name}/in1"])
    return x, (md if vector else md.expand(-1, -1, x.shape[2], x.shape[3]).contiguous())


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_f4_non_default_options_oracle_and_init(name):
    """SFT_type 'concat' / 'weak' / 'none', mask_para, repeats, q_injection (ref: SFTMD_variants/architectures.py:8-22, :25-56,
    :80-107, :130-136): same seed-8 weights as the reference, and the oracle reproduces its output and gradients."""
    a = np.load(f"{sisr_amd.__path__[0]}/../tests/golden/f4_sftmd_variants.npz")
    net, kw, vector = variant_net(name)
    assert digest(net.state_dict()) == str(a[f"{name}/sd_sha256"]), "seed-8 init differs from the reference's"
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    x, md = variant_inputs(a, name, vector)
    cfg = {k: v for k, v in kw.items() if k not in ("input_para", "in_nc")}
    cfg["sft_type"] = cfg.pop("SFT_type", "standard")
    out = O.sftmd(sd, x, md, num_blocks=2, scale=2, **cfg)
    np.testing.assert_allclose(out.detach().numpy(), a[f"{name}/out"], rtol=1e-5, atol=1e-6)
    out.backward(torch.from_numpy(a[f"{name}/cot"]))
    keys = [k[len(name) + 5:] for k in a.files if k.startswith(name + "/pgn/")]
    assert keys and all(sd[k].grad is not None for k in keys)
    for k in keys:
        gn = float(a[f"{name}/pgn/{k}"])
        assert abs(float(sd[k].grad.double().norm()) - gn) <= 1e-4 * gn + 1e-12, k
