"""SRMD on the HIP kernels vs the reference's own vectors (pytest -m gpu; fixtures: tools/make_fixtures_srmd.py)."""
import numpy as np
import pytest
import torch

import sisr_amd
from conftest import golden_json, load_golden
from test_init_parity import set5

pytestmark = pytest.mark.gpu
PARAMS = {"metadata": ["blur_kernel"], "nc": 128, "nb": 12}


def build(eval_mode=True, **extra):
    torch.manual_seed(8)
    return sisr_amd.available_models["srmd"](device=0, model_save_dir="/tmp", eval_mode=eval_mode, scale=4, **PARAMS, **extra)


def test_m1_reduced_net_output_and_gradients():
    """nc = 64, nb = 4 on a 13-channel 9 x 21 input: zero-padded head (13 -> 64 in) and tail (48 -> 64 out) weights, the
    conv chain node, the RGB shuffle gather -- output and every parameter gradient vs the reference."""
    a, meta = load_golden("m1_srmd_reduced")
    net = sisr_amd.srmd.SRMD(in_nc=meta["in_nc"], nc=meta["nc"], nb=meta["nb"], scale=meta["scale"])
    net.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in a.items() if k.startswith("sd/")}, strict=True)
    net.to("cuda:0")
    out = net(torch.from_numpy(a["in0"]).cuda())
    np.testing.assert_allclose(out.detach().cpu().numpy(), a["out"], rtol=2e-4, atol=2e-5)
    out.backward(torch.from_numpy(a["cot"]).cuda())
    for k, p in net.named_parameters():
        want = a["pg/" + k]
        err = np.linalg.norm(p.grad.cpu().numpy().ravel() - want.ravel()) / (np.linalg.norm(want.ravel()) + 1e-30)
        assert err < 5e-5, (k, err)


def test_m2_set5_forward_psnr_parity_with_reference():
    ref = golden_json("m_srmd")["full_depth"]["images"]
    crops = np.load(f"{sisr_amd.__path__[0]}/../tests/golden/m2_srmd_crops.npz")
    h = build()
    for im, x, y, md in set5():
        out, loss, _ = h.run_eval(x, y, request_loss=True, metadata=md, metadata_keys=[("blur_kernel",)] * 10)
        o = out[0].numpy()
        assert abs(sisr_amd.metrics.y_psnr(o, y[0].numpy()) - ref[im]["y_psnr"]) < 1e-3, im
        assert abs(float(loss) - ref[im]["l1"]) < 1e-5
        hh, ww = o.shape[1:]
        np.testing.assert_allclose(o[:, hh // 2 - 16:hh // 2 + 16, ww // 2 - 16:ww // 2 + 16], crops[im], rtol=1e-3, atol=1e-4)


def test_m3_run_train_trajectory_matches_reference():
    ref = golden_json("m_srmd")["train_steps"]
    h = build(eval_mode=False, lr=1e-4, scheduler=ref["scheduler"], scheduler_params=ref["scheduler_params"])
    g = torch.Generator().manual_seed(77)
    for step in ref["steps"]:
        x, y = torch.rand(2, 3, 16, 16, generator=g), torch.rand(2, 3, 64, 64, generator=g)
        md = torch.rand(2, 10, generator=g, dtype=torch.float64) * 0.4
        assert abs(h.get_learning_rate() - step["lr_before"]) < 1e-12
        loss, out = h.run_train(x, y, metadata=md, metadata_keys=[("blur_kernel", "blur_kernel")] * 10)
        gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in h.net.parameters())))
        assert abs(float(loss) - step["loss"]) < 5e-6 and abs(gn / step["grad_norm"] - 1) < 1e-4
        assert abs(float(out.mean()) - step["out_mean"]) < 5e-5 and abs(h.get_learning_rate() - step["lr_after"]) < 1e-12
    psum = float(sum(v.double().sum() for v in h.net.state_dict().values()))
    assert abs(psum - ref["final_param_sum"]) < 5e-2
