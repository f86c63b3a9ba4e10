"""bf16 matrix-core mode (ops.set_precision('bf16')) on a real MI355X (pytest -m gpu).

The reference has no reduced-precision mode, so there is no reference vector for it: PARITY UNPINNED against the
reference.  What is pinned: (1) the kernels against the oracle's restatement of the mode (operands of every
64-channel-multiple conv contraction rounded to bf16, exact products, fp32 accumulation -- oracle.CONV_PRECISION =
'bf16'), to fp32 summation-order tolerance; (2) the distance from the fp32 result, reported and bounded loosely.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import sisr_amd
from oracle import sisr_oracle as O
from test_hip_gpu import DEV, close, rnd
from test_init_parity import set5

pytestmark = pytest.mark.gpu
A = sisr_amd.architectures
ops = sisr_amd.ops
hip = sisr_amd.hip


@pytest.fixture(autouse=True)
def bf16_mode():
    ops.set_precision("bf16")
    O.CONV_PRECISION = "bf16"
    yield
    ops.set_precision("fp32")
    O.CONV_PRECISION = "fp32"


def r16(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("B,H,W", [(2, 16, 16), (1, 13, 9), (1, 57, 86), (2, 4, 32), (1, 128, 128)])
def test_conv64_bf16_fwd_bwd(B, H, W):
    m = A.default_conv(64, 64, 3)
    x = rnd(B, 64, H, W, seed=1)
    cot = rnd(B, 64, H, W, seed=2)
    sd = {"c." + k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    xo = x.clone().requires_grad_(True)
    ref = O.conv(sd, "c", xo)
    ref.backward(cot)
    m.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    out = ops.conv3x3(xd, m.weight, m.bias)
    close(out, ref, 2e-5, 2e-6, "bf16 conv out")
    out.backward(cot.to(DEV))
    close(xd.grad, xo.grad, 2e-5, 2e-6, "bf16 conv dx")
    close(m.weight.grad, sd["c.weight"].grad, 1e-4, 1e-5, "bf16 conv dw")
    close(m.bias.grad, sd["c.bias"].grad, 1e-4, 1e-5, "bf16 conv db")
    # and it really is the bf16 contraction, not fp32: the fp32 result differs at the 1e-3 level
    with torch.no_grad():
        exact = F.conv2d(x, m.weight.cpu(), m.bias.cpu(), padding=1)
    assert float((out.detach().cpu() - exact).abs().max()) > 1e-4


def test_multichunk_and_shuffle_bf16():
    """128 -> 64 (two input chunks) and 64 -> 256 + PixelShuffle(2) through the bf16 kernels."""
    for cin, cout, shuffle in ((128, 64, 1), (64, 256, 2)):
        m = A.default_conv(cin, cout, 3)
        x = rnd(2, cin, 9, 37, seed=3)
        sd = {"c." + k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
        xo = x.clone().requires_grad_(True)
        ref = O.conv(sd, "c", xo)
        if shuffle > 1:
            ref = F.pixel_shuffle(ref, shuffle)
        cot = rnd(*ref.shape, seed=4)
        ref.backward(cot)
        m.to(DEV)
        xd = x.to(DEV).requires_grad_(True)
        out = ops.conv3x3(xd, m.weight, m.bias, shuffle=shuffle)
        close(out, ref, 2e-5, 2e-6, f"{cin}->{cout} out")
        out.backward(cot.to(DEV))
        close(xd.grad, xo.grad, 3e-5, 3e-6, f"{cin}->{cout} dx")
        close(m.weight.grad, sd["c.weight"].grad, 1e-4, 1e-5, f"{cin}->{cout} dw")
        close(m.bias.grad, sd["c.bias"].grad, 1e-4, 1e-5, f"{cin}->{cout} db")


def _net_vs_oracle(net, name, cfg, x, md=None):
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    ref = O.forward(name, sd, x.clone(), md, **cfg)
    cot = rnd(*ref.shape, seed=31)
    ref.backward(cot)
    net.to(DEV)
    out = net(x.to(DEV), md.to(DEV)) if md is not None else net(x.to(DEV))
    close(out, ref, 5e-4, 5e-5, name + " out")
    out.backward(cot.to(DEV))
    for k, p in net.named_parameters():
        close(p.grad, sd[k].grad, 3e-3, 3e-4, f"{name} grad {k}")
    return out.detach().cpu()


def test_rcan_reduced_bf16_vs_oracle_restatement():
    torch.manual_seed(8)
    net = A.RCAN(n_resblocks=2, n_resgroups=2, n_feats=64, scale=4)
    x = rnd(2, 3, 20, 36, seed=30, scale=0.5)
    out = _net_vs_oracle(net, "rcan", dict(n_resgroups=2, n_resblocks=2, scale=4), x)
    O.CONV_PRECISION = "fp32"
    with torch.no_grad():
        exact = O.forward("rcan", {k: v.cpu() for k, v in net.state_dict().items()}, x, None, n_resgroups=2,
                          n_resblocks=2, scale=4)
    rel = float((out - exact).abs().max() / exact.abs().max())
    print("reduced RCAN bf16 vs fp32: max rel diff", rel)
    assert 1e-5 < rel < 2e-2


def test_qrcan_and_han_reduced_bf16_vs_oracle_restatement():
    torch.manual_seed(8)
    net = A.QRCAN(n_resblocks=2, n_resgroups=2, n_feats=64, scale=4, style="standard", num_metadata=10,
                  include_q_layer=True)
    _net_vs_oracle(net, "qrcan", dict(n_resgroups=2, n_resblocks=2, scale=4, style="standard", include_q_layer=True),
                   rnd(2, 3, 12, 34, seed=33, scale=0.5), rnd(2, 10, 1, 1, seed=34, scale=0.3))
    torch.manual_seed(8)
    net = sisr_amd.han.HAN(n_resgroups=10, n_resblocks=1, n_feats=64, scale=4)
    with torch.no_grad():
        net.la.gamma.fill_(0.37)
        net.csa.gamma.fill_(0.37)
    _net_vs_oracle(net, "han", dict(n_resgroups=10, n_resblocks=1, scale=4), rnd(1, 3, 16, 20, seed=35, scale=0.5))


def test_han_full_depth_bf16_set5_psnr_close_to_fp32():
    """BASELINE config 'HAN x4 bf16': full-depth HAN on Set5 in bf16 mode vs the fp32 HIP path (same weights)."""
    torch.manual_seed(8)
    h = sisr_amd.handlers.available_models["han"](device=0, model_save_dir="/tmp", eval_mode=True, scale=4)
    rows = []
    for im, x, y, md in set5():
        ops.set_precision("bf16")
        o16, _, _ = h.run_eval(x, y, request_loss=False)
        ops.set_precision("fp32")
        o32, _, _ = h.run_eval(x, y, request_loss=False)
        p16 = sisr_amd.metrics.y_psnr(o16[0].numpy(), y[0].numpy())
        p32 = sisr_amd.metrics.y_psnr(o32[0].numpy(), y[0].numpy())
        rows.append((im, p16, p32, float((o16 - o32).abs().max())))
    print("HAN bf16 vs fp32 Set5 (image, psnr16, psnr32, max abs diff):", rows)
    for im, p16, p32, d in rows:
        assert abs(p16 - p32) < 0.05, rows
        assert d > 0, "bf16 mode produced the fp32 result bit for bit: the switch did nothing"


def test_training_step_bf16_runs_and_tracks_fp32():
    """Ten RCAN training steps in each mode from the same init / data: losses stay within 2 % of each other."""
    losses = {}
    for mode in ("fp32", "bf16"):
        ops.set_precision(mode)
        torch.manual_seed(8)
        h = sisr_amd.handlers.available_models["rcan"](device=0, model_save_dir="/tmp", eval_mode=False, scale=4, lr=1e-4)
        g = torch.Generator().manual_seed(9)
        ls = []
        for _ in range(10):
            x, y = torch.rand(4, 3, 32, 32, generator=g), torch.rand(4, 3, 128, 128, generator=g)
            loss, _ = h.run_train(x, y)
            ls.append(float(loss))
        losses[mode] = ls
    print("loss trajectories:", losses)
    for a, b in zip(losses["fp32"], losses["bf16"]):
        assert np.isfinite(b) and abs(a - b) < 0.02 * abs(a) + 1e-3, losses
