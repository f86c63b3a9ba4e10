"""bf16 STORAGE of the maps a residual group keeps (ops.set_storage('act') on top of ops.set_precision('bf16'); BASELINE
config 5) on a real MI355X (pytest -m gpu).

PARITY UNPINNED against the reference (it has no reduced-precision mode).  What is pinned:
  (1) every bf16-storage launch against the ALREADY TESTED bf16-operand kernels, bit for bit: a bf16-stored input read as it
      is equals that kernel on the exactly-upcast fp32 copy, and a bf16-stored output equals the round-to-nearest-even of that
      kernel's fp32 output (same accumulators, same order) -- conv forms of the forward and backward passes, the GATE
      prologue's stored skip, mask / dot operands, the weight gradient;
  (2) whole reduced nets against the oracle's restatement of the storage rounding (oracle.MAP_STORAGE = 'bf16'), statistically
      (no further from it than it is from fp32 -- the argument of tests/test_bf16_gpu.py);
  (3) the distance from fp32: Set5 PSNR of the full-depth HAN and ten training steps.
"""
import numpy as np
import pytest
import torch

import sisr_amd
from oracle import sisr_oracle as O
from test_bf16_gpu import _net_vs_oracle, rms
from test_hip_gpu import DEV, rnd
from test_init_parity import set5

pytestmark = pytest.mark.gpu
A = sisr_amd.architectures
ops = sisr_amd.ops
hip = sisr_amd.hip
CL = torch.channels_last


@pytest.fixture(autouse=True, params=["act", "all"])
def bf16_storage_mode(request):
    """act: the groups' activations are bf16 maps; all: the gradient maps their backward hands from launch to launch too"""
    ops.set_precision("bf16")
    ops.set_storage(request.param)
    O.CONV_PRECISION, O.MAP_STORAGE, O.GRAD_STORAGE = "bf16", "bf16", ("bf16" if request.param == "all" else "fp32")
    yield request.param
    ops.set_precision("fp32")
    ops.set_storage("0")
    O.CONV_PRECISION, O.MAP_STORAGE, O.GRAD_STORAGE = "fp32", "fp32", "fp32"


def _maps(B, H, W, n, seed):
    """n random maps as (bf16 channels-last on the device, the exact fp32 upcast of the same values)"""
    out = []
    for i in range(n):
        t = rnd(B, 64, H, W, seed=seed + i).to(DEV).contiguous(memory_format=CL)
        t16 = t.to(torch.bfloat16).contiguous(memory_format=CL)
        out.append((t16, t16.to(torch.float32).contiguous(memory_format=CL)))
    return out


def _f32(B, H, W):
    return torch.full((B, 64, H, W), float("nan"), device=DEV).contiguous(memory_format=CL)


def _b16(B, H, W):
    return torch.full((B, 64, H, W), float("nan"), device=DEV, dtype=torch.bfloat16).contiguous(memory_format=CL)


@pytest.mark.parametrize("B,H,W", [(1, 13, 9), (2, 16, 40), (1, 57, 86), (9, 128, 128)])
def test_every_bf16_storage_conv_form_equals_the_bf16_operand_kernel(B, H, W):
    (x16, x32), (s16, s32), (m16, m32), (d16, d32) = _maps(B, H, W, 4, seed=200)
    res = rnd(B, 64, H, W, seed=210).to(DEV).contiguous(memory_format=CL)
    w, b = rnd(64, 64, 3, 3, seed=211, scale=0.05).to(DEV), rnd(64, seed=212).to(DEV)
    g = (rnd(B, 64, seed=213).abs() + 0.25).to(DEV)
    sc, sh = (rnd(B, 64, seed=214).abs() + 0.5).to(DEV), rnd(B, 64, seed=215).to(DEV)
    v = hip.view_plain(H, W, 64)
    pk = ops.pack_weight(w, "fwd")
    parts = ops.gap_parts(H, W)

    def ref(xin, **kw):  # the tested bf16-operand kernel on fp32 maps (select 0 / 1 are bit-identical: test_bf16_gpu.py)
        y = _f32(B, H, W)
        gap = torch.full((B, parts, 64), float("nan"), device=DEV) if kw.pop("gap", False) else None
        go = _f32(B, H, W) if kw.pop("gate_out", False) else None
        ops.conv_c64(xin, v, pk, kw.pop("bias", None), (1, 64), y, v, B, H, W, 64, 64, gap=gap, gate_out=go, **kw)
        return y, gap, go

    def new(xin, storage, out16, **kw):
        y = _b16(B, H, W) if out16 else _f32(B, H, W)
        gap = torch.full((B, parts, 64), float("nan"), device=DEV) if kw.pop("gap", False) else None
        go = _b16(B, H, W) if kw.pop("gate_out", False) else None
        ops.conv_c64s(xin, pk, kw.pop("bias", None), y, B, H, W, storage, gap=gap, gate_out=go, **kw)
        return y, gap, go

    def same(got, want, what):
        (y, gap, go), (y0, gap0, go0) = got, want
        assert not torch.isnan(y.float()).any(), what
        want_y = y0.to(torch.bfloat16) if y.dtype == torch.bfloat16 else y0  # .to(bfloat16) rounds to nearest even
        assert torch.equal(y, want_y), f"{what}: output"
        if gap is not None:
            assert torch.equal(gap, gap0), f"{what}: partial sums"
        if go is not None:
            assert torch.equal(go, go0.to(torch.bfloat16)), f"{what}: stored gated skip"

    # forward of the group node
    same(new(x16, 3, True, bias=b, relu=True), ref(x32, bias=b, relu=True), "conv + ReLU, bf16 in / out")
    same(new(x16, 3, True, bias=b, gap=True), ref(x32, bias=b, gap=True), "conv + GAP sums, bf16 in / out")
    same(new(x16, 3, True, bias=b, relu=True, in_scale=g, gate_add=s16, gate_out=True),
         ref(x32, bias=b, relu=True, in_scale=g, gate_add=s32, gate_out=True), "GATE prologue, bf16 in / out")
    same(new(x16, 1, False, bias=b, in_scale=g, gate_add=s16, gate_out=True, res=res),
         ref(x32, bias=b, in_scale=g, gate_add=s32, gate_out=True, res=res), "group tail: GATE + fp32 residual -> fp32")
    # backward of the group node: fp32 gradient maps, bf16 saved activations as mask / dot operands
    dy = res
    same(new(dy, 4, False, gap=True, dot=d16), ref(dy, gap=True, dot=d32), "first backward conv: DOT")
    same(new(dy, 4, False, gap=True, dot=d16, res=x32), ref(dy, gap=True, dot=d32, res=x32), "dgrad + residual, DOT")
    same(new(dy, 4, False, mask=m16, in_scale=sc, in_shift=sh), ref(dy, mask=m32, in_scale=sc, in_shift=sh), "dgrad, mask + affine")
    # ... and with bf16 gradient maps as well (storage 'all')
    same(new(dy, 6, True, gap=True, dot=d16), ref(dy, gap=True, dot=d32), "first backward conv: fp32 dOut -> bf16 dU, DOT")
    same(new(x16, 15, True, gap=True, dot=d16, res=s16), ref(x32, gap=True, dot=d32, res=s32), "dgrad + bf16 residual, DOT, bf16 maps")
    same(new(x16, 7, True, mask=m16, in_scale=sc, in_shift=sh), ref(x32, mask=m32, in_scale=sc, in_shift=sh), "dgrad, mask + affine, bf16 maps")
    same(new(x16, 11, True, res=s16), ref(x32, res=s32), "dgrad + bf16 residual, bf16 out")
    same(new(x16, 9, False, res=s16), ref(x32, res=s32), "block 0: dgrad + bf16 residual -> the group's fp32 dX")


@pytest.mark.parametrize("B,H,W", [(1, 13, 9), (2, 40, 48), (3, 128, 128)])
def test_weight_gradient_reads_a_bf16_map_as_it_is(B, H, W):
    (x16, x32), = _maps(B, H, W, 1, seed=230)
    dy = rnd(B, 64, H, W, seed=231).to(DEV).contiguous(memory_format=CL)
    sc, sh = (rnd(B, 64, seed=232).abs() + 0.5).to(DEV), rnd(B, 64, seed=233).to(DEV)
    v = hip.view_plain(H, W, 64)
    for kw in (dict(), dict(dy_scale=sc, dy_shift=sh)):
        out = []
        dy16 = dy.to(torch.bfloat16).contiguous(memory_format=CL)
        dy16up = dy16.to(torch.float32).contiguous(memory_format=CL)
        for xin, dyin, storage in ((x32, dy, 0), (x16, dy, 1), (x32, dy16up, 0), (x16, dy16, 3)):
            dw, db = torch.full((64, 64, 3, 3), float("nan"), device=DEV), torch.full((64,), float("nan"), device=DEV)
            ops.wgrad_c64(xin, v, dyin, v, dw, db, B, H, W, 64, 64, storage=storage, **kw)
            out.append((dw, db))
        assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]), kw.keys()  # bf16 x
        assert torch.equal(out[2][0], out[3][0]) and torch.equal(out[2][1], out[3][1]), kw.keys()  # bf16 x and dY
        assert not torch.isnan(out[1][0]).any() and not torch.isnan(out[3][0]).any()


def test_storage_entry_points_refuse_what_they_do_not_build():
    B, H, W = 1, 8, 32
    (x16, x32), = _maps(B, H, W, 1, seed=240)
    pk = ops.pack_weight(rnd(64, 64, 3, 3, seed=241, scale=0.05).to(DEV), "fwd")
    with pytest.raises(RuntimeError):  # a form / storage pair the group node never launches
        ops.conv_c64s(x16, pk, None, _f32(B, H, W), B, H, W, 1)
    ops.set_precision("fp32")
    with pytest.raises(RuntimeError):  # bf16-stored maps exist in the bf16 operand mode only
        ops.wgrad_c64(x16, hip.view_plain(H, W, 64), x32, hip.view_plain(H, W, 64), torch.empty(64, 64, 3, 3, device=DEV),
                      torch.empty(64, device=DEV), B, H, W, 64, 64, storage=1)
    ops.set_precision("bf16")


def test_group_node_keeps_bf16_maps_and_meets_the_storage_restatement():
    torch.manual_seed(8)
    net = A.RCAN(n_resblocks=2, n_resgroups=2, n_feats=64, scale=4)
    o = _net_vs_oracle(net, "rcan", dict(n_resgroups=2, n_resblocks=2, scale=4), rnd(2, 3, 20, 36, seed=30, scale=0.5))
    # ... and the storage switch really changes what is computed (the same net with fp32 maps gives other bits)
    ops.set_storage("0")
    with torch.no_grad():
        o32 = net(rnd(2, 3, 20, 36, seed=30, scale=0.5).to(DEV)).cpu()
    ops.set_storage("act")
    assert float((o - o32).abs().max()) > 0


def test_qrcan_and_han_reduced_with_bf16_storage():
    torch.manual_seed(8)
    net = A.QRCAN(n_resblocks=2, n_resgroups=2, n_feats=64, scale=4, style="standard", num_metadata=10, include_q_layer=True)
    _net_vs_oracle(net, "qrcan", dict(n_resgroups=2, n_resblocks=2, scale=4, style="standard", include_q_layer=True),
                   rnd(2, 3, 12, 34, seed=33, scale=0.5), rnd(2, 10, 1, 1, seed=34, scale=0.3))
    torch.manual_seed(8)
    net = sisr_amd.han.HAN(n_resgroups=10, n_resblocks=1, n_feats=64, scale=4)
    with torch.no_grad():
        net.la.gamma.fill_(0.37)
        net.csa.gamma.fill_(0.37)
    _net_vs_oracle(net, "han", dict(n_resgroups=10, n_resblocks=1, scale=4), rnd(1, 3, 16, 20, seed=35, scale=0.5))


def test_saved_maps_are_bf16():
    """The autograd node really holds bf16 tensors for its backward (half the bytes), gradients come out fp32."""
    torch.manual_seed(8)
    net = A.RCAN(n_resblocks=2, n_resgroups=1, n_feats=64, scale=2).to(DEV)
    saved = []
    with torch.autograd.graph.saved_tensors_hooks(lambda t: saved.append((t.dtype, tuple(t.shape))) or t, lambda t: t):
        out = net(rnd(2, 3, 16, 32, seed=36, scale=0.5).to(DEV))
    maps = [d for d, s in saved if len(s) == 4 and s[1] == 64 and s[2:] == (16, 32)]
    assert maps.count(torch.bfloat16) >= 2 * 3 + 1, maps  # per block: input, t1, t2; the last skip
    out.sum().backward()
    assert all(p.grad.dtype == torch.float32 and bool(torch.isfinite(p.grad).all()) for p in net.parameters())


def test_han_full_depth_set5_psnr_with_bf16_storage():
    torch.manual_seed(8)
    h = sisr_amd.handlers.available_models["han"](device=0, model_save_dir="/tmp", eval_mode=True, scale=4)
    rows = []
    for im, x, y, md in set5():
        o16, _, _ = h.run_eval(x, y, request_loss=False)
        ops.set_precision("fp32")
        o32, _, _ = h.run_eval(x, y, request_loss=False)
        ops.set_precision("bf16")
        rows.append((im, sisr_amd.metrics.y_psnr(o16[0].numpy(), y[0].numpy()), sisr_amd.metrics.y_psnr(o32[0].numpy(), y[0].numpy()),
                     float((o16 - o32).abs().max())))
    print("HAN bf16 operands + bf16 storage vs fp32, Set5 (image, psnr16, psnr32, max abs diff):", rows)
    for im, p16, p32, d in rows:
        assert abs(p16 - p32) < 0.05 and d > 0, rows


def test_training_steps_with_bf16_storage_track_fp32():
    losses = {}
    for mode in ("fp32", "bf16"):
        ops.set_precision(mode)
        torch.manual_seed(8)
        h = sisr_amd.handlers.available_models["rcan"](device=0, model_save_dir="/tmp", eval_mode=False, scale=4, lr=1e-4)
        g = torch.Generator().manual_seed(9)
        ls = []
        for _ in range(10):
            x, y = torch.rand(4, 3, 32, 32, generator=g), torch.rand(4, 3, 128, 128, generator=g)
            ls.append(float(h.run_train(x, y)[0]))
        losses[mode] = ls
    ops.set_precision("bf16")
    print("loss trajectories (bf16 = operands + storage):", losses)
    for a, b in zip(losses["fp32"], losses["bf16"]):
        assert np.isfinite(b) and abs(a - b) < 0.02 * abs(a) + 1e-3, losses
