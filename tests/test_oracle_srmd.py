"""SRMD: the oracle and the host-side mirror against the reference's own vectors (CPU).

Fixtures: tools/make_fixtures_srmd.py ran the reference (advanced/architectures.py:380-425 through SRMDHandler,
advanced/handlers.py:132-158) and stored m1 (reduced net: output + parameter gradients), m2 (full-depth seed-8 init
digest, Set5 run_eval) and m3 (five run_train steps)."""
import numpy as np
import torch

import sisr_amd
from conftest import golden_json, load_golden
from oracle import sisr_oracle as O
from test_init_parity import digest, set5

PARAMS = {"metadata": ["blur_kernel"], "nc": 128, "nb": 12}


def build(eval_mode=True, **extra):
    torch.manual_seed(8)
    return sisr_amd.available_models["srmd"](device=torch.device("cpu"), model_save_dir="/tmp", eval_mode=eval_mode,
                                             scale=4, **PARAMS, **extra)


def test_m1_reduced_net_output_and_gradients():
    a, meta = load_golden("m1_srmd_reduced")
    sd = {k[3:]: torch.from_numpy(v).requires_grad_(True) for k, v in a.items() if k.startswith("sd/")}
    out = O.srmd(sd, torch.from_numpy(a["in0"]), nb=meta["nb"], scale=meta["scale"])
    np.testing.assert_allclose(out.detach().numpy(), a["out"], rtol=1e-5, atol=1e-6)
    out.backward(torch.from_numpy(a["cot"]))
    for k, v in sd.items():
        np.testing.assert_allclose(v.grad.numpy(), a["pg/" + k], rtol=1e-4, atol=1e-5 * np.abs(a["pg/" + k]).max(), err_msg=k)


def test_m2_init_keys_and_digest_match_the_reference():
    ref = golden_json("m_srmd")["full_depth"]
    h = build()
    sd = h.net.state_dict()
    assert list(sd) == ref["keys"] and len(sd) == ref["n_tensors"]
    assert int(sum(p.numel() for p in h.net.parameters())) == ref["n_params"]
    assert digest(sd) == ref["sha256"], "initial weights differ from the reference's for seed 8"
    assert h.model_name == "srmd" and h.channel_concat is True and h.legacy_load is False
    assert h.colorspace == "augmented_rgb" and h.im_input == "unmodified" and h.num_metadata == 10


def test_m2_oracle_set5_forward_matches_the_reference():
    ref = golden_json("m_srmd")["full_depth"]["images"]
    crops = np.load(f"{sisr_amd.__path__[0]}/../tests/golden/m2_srmd_crops.npz")
    h = build()
    sd = h.net.state_dict()
    with torch.no_grad():
        for im, x, y, md in set5():
            maps = h.generate_sft_channels(x, md, [("blur_kernel",)] * 10)
            assert maps.shape == (1, 10, x.shape[2], x.shape[3])
            np.testing.assert_array_equal(maps.numpy(), O.sft_channels(x, md).numpy())
            out = O.srmd(sd, torch.cat((x, maps), 1))[0].numpy()
            assert abs(O.y_psnr(out, y[0].numpy()) - ref[im]["y_psnr"]) < 1e-3, im
            assert abs(float(np.abs(out - y[0].numpy()).mean()) - ref[im]["l1"]) < 1e-6
            hh, ww = out.shape[1:]
            np.testing.assert_allclose(out[:, hh // 2 - 16:hh // 2 + 16, ww // 2 - 16:ww // 2 + 16], crops[im],
                                       rtol=1e-4, atol=1e-5)


def test_m3_oracle_trajectory_matches_the_reference():
    ref = golden_json("m_srmd")["train_steps"]
    h = build(eval_mode=False)
    tr = O.Trainer("srmd", h.net.state_dict(), lr=1e-4, scheduler=ref["scheduler"], scheduler_params=ref["scheduler_params"])
    g = torch.Generator().manual_seed(77)
    for step in ref["steps"]:
        x, y = torch.rand(2, 3, 16, 16, generator=g), torch.rand(2, 3, 64, 64, generator=g)
        md = torch.rand(2, 10, generator=g, dtype=torch.float64) * 0.4
        assert abs(tr.lr - step["lr_before"]) < 1e-12
        loss, out, gn = tr.step(torch.cat((x, O.sft_channels(x, md)), 1), y)
        assert abs(loss - step["loss"]) < 2e-6 and abs(gn / step["grad_norm"] - 1) < 1e-4
        assert abs(float(out.mean()) - step["out_mean"]) < 1e-5 and abs(tr.lr - step["lr_after"]) < 1e-12
    assert abs(float(sum(v.double().sum() for v in tr.sd.values())) - ref["final_param_sum"]) < 1e-3


# ---- the non-default act_mode / upsample_mode variants (ref advanced/architectures.py:385-411; fixtures m4)
M4 = ["m4_srmd_BL_upconv", "m4_srmd_L", "m4_srmd_BR", "m4_srmd_R_upconv", "m4_srmd_IL_convtranspose", "m4_srmd_IR_convtranspose"]


def _m4_digest(sd):
    return digest({k: v for k, v in sd.items() if "running_" not in k and "num_batches" not in k})


def _check_light(a, grads, rtol_norm, atol_rel, rtol_head):
    # (a conv bias in front of a batch norm has an exactly zero gradient: what the fixture holds there is the reference's
    # rounding noise, ~1e-6 of the other gradients -- hence the floor relative to the largest gradient norm)
    floor = 1e-5 * max(float(a[k]) for k in a if k.startswith("pgn/"))
    for k in [k[4:] for k in a if k.startswith("pgn/")]:
        g = grads[k].detach().double().cpu()
        n_ref = float(a["pgn/" + k])
        assert abs(float(g.norm()) - n_ref) <= rtol_norm * n_ref + floor, (k, float(g.norm()), n_ref)
        head = a["pgh/" + k]
        np.testing.assert_allclose(g.reshape(-1)[:head.size].float().numpy(), head, rtol=rtol_head,
                                   atol=atol_rel * float(np.abs(head).max()) + floor, err_msg=k)


import pytest  # noqa: E402


@pytest.mark.parametrize("name", M4)
def test_m4_variants_module_tree_and_oracle(name):
    a, meta = load_golden(name)
    torch.manual_seed(8)
    net = sisr_amd.srmd.SRMD(**meta)
    assert _m4_digest(net.state_dict()) == str(a["sd_sha256"]), "keys / seed-8 weights differ from the reference's"
    sd = {}
    for k, v in net.state_dict().items():
        v = v.detach().clone()
        sd[k] = v.requires_grad_(True) if v.dtype == torch.float32 and "running_" not in k else v
    cfg = dict(nb=meta["nb"], scale=meta["scale"], act_mode=meta["act_mode"], upsample_mode=meta["upsample_mode"])
    out = O.srmd(sd, torch.from_numpy(a["in0"]), training=True, **cfg)
    np.testing.assert_allclose(out.detach().numpy(), a["out"], rtol=1e-5, atol=2e-6)
    out.backward(torch.from_numpy(a["cot"]))
    _check_light(a, {k: v.grad for k, v in sd.items() if v.requires_grad}, 2e-5, 1e-5, 1e-3)
    for k in [k[4:] for k in a if k.startswith("buf/")]:
        np.testing.assert_allclose(sd[k].numpy(), a["buf/" + k], rtol=1e-6, atol=1e-7, err_msg=k)
    with torch.no_grad():
        np.testing.assert_allclose(O.srmd(sd, torch.from_numpy(a["in0"]), training=False, **cfg).numpy(), a["out_eval"],
                                   rtol=1e-5, atol=2e-6)


def test_srmd_rejects_what_the_reference_rejects():
    with pytest.raises(NotImplementedError):  # ref architectures.py:404-405
        sisr_amd.srmd.SRMD(upsample_mode="bilinear")
    with pytest.raises(AssertionError):  # ref architectures.py:395
        sisr_amd.srmd.SRMD(act_mode="B")
