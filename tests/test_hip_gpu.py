"""HIP path vs the oracle / the reference's golden vectors, on a real MI355X (pytest -m gpu).

Tolerances (fp32 everywhere): the MFMA conv sums K = 576 products in a different order than the
CPU reference, so outputs agree to ~1e-5 relative; gradients that reduce over all pixels to ~1e-4.
north_star's bar is 1e-3 dB PSNR on full images; the whole-net tests below assert that too.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import sisr_amd
from conftest import golden_json, load_golden
from oracle import sisr_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["fp32", "bf16x3"])
def arith(request):
    """Arithmetic of the 64-channel convs for the golden-fixture tests: the exact-fp32 MFMA kernels (the headline) and fp32
    through the bf16 matrix cores (operands split exactly into three bf16 numbers, six products, fp32 accumulate: DESIGN.md
    section 7b).  Both must meet the SAME reference fixtures with the SAME tolerances."""
    sisr_amd.ops.set_precision(request.param)
    yield request.param
    sisr_amd.ops.set_precision("fp32")


ARITH = pytest.mark.usefixtures("arith")
A = sisr_amd.architectures
ops = sisr_amd.ops
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def close(got, want, rtol, atol, msg=""):
    got = got.detach().float().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    want = want.detach().float().cpu().numpy() if torch.is_tensor(want) else np.asarray(want)
    scale = max(1.0, float(np.abs(want).max()))
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol * scale, err_msg=msg)


def rnd(*shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ----------------------------------------------------------------------------- conv kernels vs ATen on CPU
@pytest.mark.parametrize("B,H,W", [(2, 16, 16), (1, 13, 9), (1, 57, 86), (2, 4, 32), (1, 5, 33), (1, 128, 128)])
def test_conv64_fwd_bwd(B, H, W):
    x = rnd(B, 64, H, W, seed=1).requires_grad_(True)
    w = (rnd(64, 64, 3, 3, seed=2) * 0.05).requires_grad_(True)
    b = (rnd(64, seed=3) * 0.1).requires_grad_(True)
    cot = rnd(B, 64, H, W, seed=4)
    ref = F.conv2d(x, w, b, padding=1)
    ref.backward(cot)
    xg = x.detach().to(DEV).requires_grad_(True)
    wg = w.detach().to(DEV).requires_grad_(True)
    bg = b.detach().to(DEV).requires_grad_(True)
    out = ops.conv3x3(xg, wg, bg)
    out.backward(cot.to(DEV))
    close(out, ref, 1e-4, 2e-5, "out")
    close(xg.grad, x.grad, 1e-4, 2e-5, "dx")
    close(wg.grad, w.grad, 2e-4, 2e-5, "dw")
    close(bg.grad, b.grad, 2e-4, 2e-5, "db")


@pytest.mark.parametrize("B,H,W", [(2, 16, 16), (1, 13, 9), (1, 57, 86), (2, 4, 32), (1, 5, 33), (1, 128, 128), (9, 128, 128)])
def test_conv_tile_heights_are_bit_identical(B, H, W):
    """The fp32 conv's kernel forms -- 4-row tiles (variant 5), 2-row tiles (6), and the persistent two-workgroups-per-CU form
    that prefetches its next tile (7; the automatic choice from two tiles per workgroup on) -- must give the same bits for
    outputs and GAP partials, for every epilogue they serve (the 9 x 128 x 128 case walks 2 - 3 tiles per persistent
    workgroup, ragged last round included)."""
    if ops.PRECISION != "fp32":
        pytest.skip("tile-height selection is an argument of the fp32 kernel family only")
    hip = sisr_amd.hip
    cl = torch.channels_last
    x = rnd(B, 64, H, W, seed=50).to(DEV).contiguous(memory_format=cl)
    res = rnd(B, 64, H, W, seed=51).to(DEV).contiguous(memory_format=cl)
    mask = rnd(B, 64, H, W, seed=52).to(DEV).contiguous(memory_format=cl)
    w = rnd(64, 64, 3, 3, seed=53, scale=0.05).to(DEV)
    b = rnd(64, seed=54).to(DEV)
    sc, sh = (rnd(B, 64, seed=55).abs() + 0.5).to(DEV), rnd(B, 64, seed=56).to(DEV)
    v = hip.view_plain(H, W, 64)
    pk = ops.pack_weight(w, "fwd")
    cases = [dict(bias=b, relu=True, gap=True), dict(res=res, alpha=0.3), dict(mask=mask),
             dict(mask=mask, in_scale=sc, in_shift=sh)]
    outs = {}
    for variant in (5, 6, 7):  # per-call kernel selection (the library keeps no state)
        for i, kw in enumerate(cases):
            kw = dict(kw)
            y = torch.zeros(B, 64, H, W, device=DEV).contiguous(memory_format=cl)
            gap = torch.full((B, ops.gap_parts(H, W), 64), float("nan"), device=DEV) if kw.pop("gap", False) else None
            ops.conv_c64(x, v, pk, kw.pop("bias", None), (1, 64), y, v, B, H, W, 64, 64, gap=gap, select=variant, **kw)
            outs[(variant, i)] = (y, gap)
    for i in range(len(cases)):
        y5, g5 = outs[(5, i)]
        for other in (6, 7):
            y6, g6 = outs[(other, i)]
            assert torch.equal(y5, y6), f"case {i}: outputs differ between kernel forms 5 and {other}"
            if g5 is not None:
                assert torch.equal(g5, g6), f"case {i}: GAP partials differ between kernel forms 5 and {other}"
    want = F.relu(F.conv2d(x.cpu(), w.cpu(), b.cpu(), padding=1))
    close(outs[(5, 0)][0], want, 2e-5, 2e-6, "4-row tile vs ATen")


@pytest.mark.parametrize("B,H,W", [(1, 13, 9), (2, 16, 40), (1, 57, 86), (1, 128, 128), (10, 128, 128)])
def test_conv_gate_prologue_and_dot_epilogue(B, H, W):
    """The neighbour-fusion hooks of the 64->64 conv: GATE (input = t*g + skip, also written out) and DOT (GAP
    slots hold sum(v * dot)), on both tile heights (B = 10 at 128x128 uses the 4-row tile)."""
    hip = sisr_amd.hip
    cl = torch.channels_last
    dev4 = lambda t: t.to(DEV).contiguous(memory_format=cl)  # noqa: E731
    t, skip, res, dot = (rnd(B, 64, H, W, seed=60 + i) for i in range(4))
    g = rnd(B, 64, seed=64).abs() + 0.25
    w, b = rnd(64, 64, 3, 3, seed=65, scale=0.05), rnd(64, seed=66)
    td, skd, resd, dotd, gd, wd, bd = dev4(t), dev4(skip), dev4(res), dev4(dot), g.to(DEV), w.to(DEV), b.to(DEV)
    v = hip.view_plain(H, W, 64)
    pk = ops.pack_weight(wd, "fwd")
    u = t * g.view(B, 64, 1, 1) + skip
    # GATE + ReLU (conv1 of the next block) and GATE + residual (group tail)
    for kw, want in ((dict(relu=True), F.relu(F.conv2d(u, w, b, padding=1))),
                     (dict(res=resd), F.conv2d(u, w, b, padding=1) + res)):
        y = torch.full((B, 64, H, W), float("nan"), device=DEV).contiguous(memory_format=cl)
        uo = torch.full((B, 64, H, W), float("nan"), device=DEV).contiguous(memory_format=cl)
        ops.conv_c64(td, v, pk, bd, (1, 64), y, v, B, H, W, 64, 64, in_scale=gd, gate_add=skd, gate_out=uo, **kw)
        close(uo, u, 1e-6, 1e-6, "gate_out")
        close(y, want, 2e-5, 2e-6, "gated conv")
    # DOT with and without residual
    for kw, want in ((dict(), F.conv2d(t, w, None, padding=1)), (dict(res=resd), F.conv2d(t, w, None, padding=1) + res)):
        y = torch.empty(B, 64, H, W, device=DEV).contiguous(memory_format=cl)
        gap = torch.full((B, ops.gap_parts(H, W), 64), float("nan"), device=DEV)
        ops.conv_c64(td, v, pk, None, (1, 64), y, v, B, H, W, 64, 64, gap=gap, dot=dotd, **kw)
        close(y, want, 2e-5, 2e-6, "dot conv output")
        close(gap.sum(dim=1), (want * dot).sum(dim=(2, 3)), 2e-4, 2e-5, "dot partials")


@pytest.mark.parametrize("meta", [False, True, "qedsr"])
def test_fused_group_node_matches_per_block_nodes(meta):
    """ops._GatedGroup (one autograd node per residual group, gate passes folded into the neighbouring convs) against
    the per-block nodes it replaces: same outputs and gradients to fp32 reduction-order tolerance.  'qedsr': the node without
    channel attention (gate = the meta-attention vector, res_scale in conv2's epilogue) for the whole QEDSR body."""
    torch.manual_seed(8)
    if meta == "qedsr":
        net = A.QEDSR(num_features=64, input_para=10, num_blocks=5, scale=2, res_scale=0.1).to(DEV)
    elif meta:
        net = A.QRCAN(n_resblocks=3, n_resgroups=2, n_feats=64, scale=2, style="standard", num_metadata=10,
                      include_q_layer=True, num_q_layers_inner_residual=2).to(DEV)
    else:
        net = A.RCAN(n_resblocks=3, n_resgroups=2, n_feats=64, scale=2).to(DEV)
    x = rnd(2, 3, 21, 30, seed=70, scale=0.5).to(DEV)
    md = rnd(2, 10, 1, 1, seed=71, scale=0.3).to(DEV)
    cot = None
    res = {}
    try:
        for fused in (True, False):
            ops.FUSED_GROUPS = fused
            net.zero_grad(set_to_none=True)
            out = net(x, md) if meta else net(x)
            if cot is None:
                cot = rnd(*out.shape, seed=72).to(DEV)
            out.backward(cot)
            res[fused] = (out.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters()})
    finally:
        ops.FUSED_GROUPS = True
    # the gated skip is formed with the same two roundings (product, then sum) in both paths, so the forward pass -- and
    # with it every ReLU mask -- is bit-identical; the gradients then differ only by how the gate sums are partitioned
    assert torch.equal(res[True][0], res[False][0])
    for k in res[True][1]:
        a, b = res[True][1][k].double(), res[False][1][k].double()
        assert float((a - b).norm()) <= 2e-5 * float(b.norm()) + 1e-7, (k, float((a - b).norm()), float(b.norm()))


@pytest.mark.parametrize("meta", [False, True])
def test_channel_attention_tails_are_bit_identical_to_the_gate_launches(meta):
    """The gate (and its backward) computed by the last-arriving workgroup of the conv launch that writes its partial sums
    (ops.CA_TAIL) against the stand-alone gate launches: same code path (ca_gate.h), same summation order -> every output
    and every gradient equal to the bit, on a batch whose samples finish in any order (B = 3, odd sizes)."""
    torch.manual_seed(8)
    if meta:
        net = A.QRCAN(n_resblocks=3, n_resgroups=2, n_feats=64, scale=2, style="standard", num_metadata=10,
                      include_q_layer=True).to(DEV)
    else:
        net = A.RCAN(n_resblocks=3, n_resgroups=2, n_feats=64, scale=2).to(DEV)
    x = rnd(3, 3, 37, 70, seed=80, scale=0.5).to(DEV)
    md = rnd(3, 10, 1, 1, seed=81, scale=0.3).to(DEV)
    cot, res, prev = None, {}, ops.CA_TAIL
    try:
        for mode in ("1", "0", "1"):  # twice with tails: the counters must come back to zero
            ops.CA_TAIL = mode
            net.zero_grad(set_to_none=True)
            out = net(x, md) if meta else net(x)
            if cot is None:
                cot = rnd(*out.shape, seed=82).to(DEV)
            out.backward(cot)
            cur = (out.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters()})
            if mode in res:
                assert torch.equal(res[mode][0], cur[0]) and all(torch.equal(res[mode][1][k], cur[1][k]) for k in cur[1])
            res[mode] = cur
    finally:
        ops.CA_TAIL = prev
    assert torch.equal(res["1"][0], res["0"][0])
    for k in res["1"][1]:
        assert torch.equal(res["1"][1][k], res["0"][1][k]), k


@pytest.mark.parametrize("meta", [False, True])
@pytest.mark.parametrize("shape", [(3, 37, 70), (2, 128, 128)])  # both within ops.GATE_HEADS_MAX_PIXELS
def test_gate_heads_are_bit_identical_to_the_gate_launches(meta, shape):
    """Small launches: the channel-attention gate and its backward are computed by the conv that CONSUMES them (gate heads,
    ops.GATE_HEADS; every workgroup for its own sample) and the gates' parameter gradients by one launch per group -- against
    the stand-alone gate launches: same device functions (ca_gate.h), same summation order -> equal to the bit.  Both tile
    heights (the second shape has enough tiles for the 4-row kernel)."""
    torch.manual_seed(8)
    if meta:
        net = A.QRCAN(n_resblocks=3, n_resgroups=2, n_feats=64, scale=2, style="standard", num_metadata=10,
                      include_q_layer=True).to(DEV)
    else:
        net = A.RCAN(n_resblocks=3, n_resgroups=2, n_feats=64, scale=2).to(DEV)
    B, H, W = shape
    x = rnd(B, 3, H, W, seed=80, scale=0.5).to(DEV)
    md = rnd(B, 10, 1, 1, seed=81, scale=0.3).to(DEV)
    cot, res, prev = None, {}, ops.GATE_HEADS
    assert ops.WgradQueue.wanted(B, H, W) or ops.PRECISION != "fp32"
    try:
        for mode in (True, False):
            ops.GATE_HEADS = mode
            net.zero_grad(set_to_none=True)
            out = net(x, md) if meta else net(x)
            if cot is None:
                cot = rnd(*out.shape, seed=82).to(DEV)
            out.backward(cot)
            res[mode] = (out.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters()})
    finally:
        ops.GATE_HEADS = prev
    assert torch.equal(res[True][0], res[False][0])
    for k in res[True][1]:
        assert torch.equal(res[True][1][k], res[False][1][k]), k


@pytest.mark.parametrize("meta", [False, True])
@pytest.mark.parametrize("shape,lanes,heads", [((4, 37, 70), 2, True), ((4, 128, 128), 2, True), ((4, 40, 48), 4, True),
                                               ((6, 32, 40), 2, False), ((6, 32, 40), 3, True)])
def test_sample_lanes_are_bit_identical_to_one_chain(meta, shape, lanes, heads):
    """Small launches of a residual group run as LANES chains of B / LANES samples on parallel streams (ops._lane_cuts; inside
    a hipGraph capture by default, forced here in eager mode): every launch of a lane is the same kernel on a contiguous
    slice of the batch and the weight gradients stay whole-batch launches, so outputs and every gradient equal the single
    chain's to the bit -- with gate heads and with stand-alone gate launches, five blocks (two weight-gradient segments)."""
    torch.manual_seed(8)
    if meta:
        net = A.QRCAN(n_resblocks=5, n_resgroups=2, n_feats=64, scale=2, style="standard", num_metadata=10,
                      include_q_layer=True).to(DEV)
    else:
        net = A.RCAN(n_resblocks=5, n_resgroups=2, n_feats=64, scale=2).to(DEV)
    B, H, W = shape
    x = rnd(B, 3, H, W, seed=83, scale=0.5).to(DEV)
    md = rnd(B, 10, 1, 1, seed=84, scale=0.3).to(DEV)
    cot, res = None, {}
    prev = (ops.LANES, ops.LANES_EAGER, ops.GATE_HEADS)
    if ops.PRECISION != "fp32":
        pytest.skip("sample lanes are an fp32-path feature")
    try:
        ops.GATE_HEADS = heads
        for mode in (lanes, 1):
            ops.LANES, ops.LANES_EAGER = mode, True
            assert len(ops._lane_cuts(B, H, W)) == mode
            net.zero_grad(set_to_none=True)
            out = net(x, md) if meta else net(x)
            if cot is None:
                cot = rnd(*out.shape, seed=85).to(DEV)
            out.backward(cot)
            torch.cuda.synchronize()
            res[mode] = (out.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters()})
    finally:
        ops.LANES, ops.LANES_EAGER, ops.GATE_HEADS = prev
    assert torch.equal(res[lanes][0], res[1][0])
    for k in res[lanes][1]:
        assert torch.equal(res[lanes][1][k], res[1][1][k]), k


def test_conv_residual_alpha_and_multichunk():
    # 128 -> 192 channels, y = conv*alpha + res  (multi-chunk K loop and multiple output chunks)
    B, H, W = 1, 9, 35
    x = rnd(B, 128, H, W, seed=5).requires_grad_(True)
    w = (rnd(192, 128, 3, 3, seed=6) * 0.03).requires_grad_(True)
    b = (rnd(192, seed=7) * 0.1).requires_grad_(True)
    r = rnd(B, 192, H, W, seed=8).requires_grad_(True)
    cot = rnd(B, 192, H, W, seed=9)
    ref = F.conv2d(x, w, b, padding=1) * 0.3 + r
    ref.backward(cot)
    g = [t.detach().to(DEV).requires_grad_(True) for t in (x, w, b, r)]
    out = ops.conv3x3(g[0], g[1], g[2], residual=g[3], alpha=0.3)
    out.backward(cot.to(DEV))
    close(out, ref, 1e-4, 2e-5)
    for got, want, nm in zip(g, (x, w, b, r), "xwbr"):
        close(got.grad, want.grad, 2e-4, 2e-5, "d" + nm)


@pytest.mark.parametrize("r", [2, 3])
def test_conv_pixelshuffle_fused(r):
    B, H, W = 2, 7, 10
    x = rnd(B, 64, H, W, seed=10).requires_grad_(True)
    w = (rnd(64 * r * r, 64, 3, 3, seed=11) * 0.05).requires_grad_(True)
    b = (rnd(64 * r * r, seed=12) * 0.1).requires_grad_(True)
    cot = rnd(B, 64, H * r, W * r, seed=13)
    ref = F.pixel_shuffle(F.conv2d(x, w, b, padding=1), r)
    ref.backward(cot)
    g = [t.detach().to(DEV).requires_grad_(True) for t in (x, w, b)]
    out = ops.conv3x3(g[0], g[1], g[2], shuffle=r)
    assert out.shape == ref.shape
    out.backward(cot.to(DEV))
    close(out, ref, 1e-4, 2e-5)
    for got, want, nm in zip(g, (x, w, b), "xwb"):
        close(got.grad, want.grad, 2e-4, 2e-5, "d" + nm)


@pytest.mark.parametrize("shape", [(2, 11, 19), (1, 1, 1), (1, 2, 31), (3, 64, 95), (1, 7, 61), (2, 200, 333)])
@pytest.mark.parametrize("cin,cout", [(3, 64), (64, 3)])
def test_rgb_side_convs(cin, cout, shape):
    """Head / tail convs and their gradients; 64 -> 3 (forward, and the head's input gradient with flipped role-swapped
    weights) runs on the matrix cores in tiles of 3 rows x 30 columns (csrc/conv_rgb_out.h): sizes off both tile edges."""
    B, H, W = shape
    x = rnd(B, cin, H, W, seed=14).requires_grad_(True)
    w = (rnd(cout, cin, 3, 3, seed=15) * 0.1).requires_grad_(True)
    b = (rnd(cout, seed=16) * 0.1).requires_grad_(True)
    cot = rnd(B, cout, H, W, seed=17)
    ref = F.conv2d(x, w, b, padding=1)
    ref.backward(cot)
    g = [t.detach().to(DEV).requires_grad_(True) for t in (x, w, b)]
    out = ops.conv3x3(g[0], g[1], g[2])
    out.backward(cot.to(DEV))
    close(out, ref, 1e-4, 1e-5)
    for got, want, nm in zip(g, (x, w, b), "xwb"):
        close(got.grad, want.grad, 2e-4, 2e-5, "d" + nm)


def test_l1_loss():
    a = rnd(2, 3, 40, 24, seed=18).requires_grad_(True)
    b = rnd(2, 3, 40, 24, seed=19)
    b.view(-1)[:7] = a.detach().view(-1)[:7]  # exact ties -> sign 0
    ref = F.l1_loss(a, b)
    ref.backward()
    ag = a.detach().to(DEV).requires_grad_(True)
    out = ops.l1_loss(ag, b.to(DEV))
    (out * 1.0).backward()
    close(out, ref, 1e-6, 1e-7)
    close(ag.grad, a.grad, 0, 1e-9)


# ----------------------------------------------------------------------------- G1: blocks vs the reference's own vectors
def run_block(name, module, n_inputs, call=None, rtol=2e-4, atol=3e-5):
    a, meta = load_golden(name)
    sd = {k[3:]: torch.from_numpy(v) for k, v in a.items() if k.startswith("sd/")}
    module.load_state_dict(sd, strict=True)
    module.to(DEV)
    ins = [torch.from_numpy(a[f"in{i}"]).to(DEV).requires_grad_(True) for i in range(n_inputs)]
    out = call(module, ins) if call else module(*ins)
    close(out, a["out"], rtol, atol, name + " out")
    out.backward(torch.from_numpy(a["cot"]).to(DEV))
    for i, t in enumerate(ins):
        g = t.grad if t.grad is not None else torch.zeros_like(t)
        close(g, a[f"gin{i}"], 5e-4, 5e-5, f"{name} gin{i}")
    for k, p in module.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        close(g, a["pg/" + k], 1e-3, 1e-4, f"{name} pg/{k}")


@ARITH
def test_g1_conv64():
    for name in ("g1_conv64", "g1_conv64_odd"):
        m = A.default_conv(64, 64, 3)
        run_block(name, m, 1, call=lambda mod, i: ops.conv3x3(i[0], mod.weight, mod.bias))


def test_g1_conv_head_tail():
    run_block("g1_conv_head", A.default_conv(3, 64, 3), 1, call=lambda mod, i: ops.conv3x3(i[0], mod.weight, mod.bias))
    run_block("g1_conv_tail", A.default_conv(64, 3, 3), 1, call=lambda mod, i: ops.conv3x3(i[0], mod.weight, mod.bias))


def test_g1_calayer():
    run_block("g1_calayer", A.CALayer(64, 16), 1)


@ARITH
@pytest.mark.parametrize("name", ["g1_rcab", "g1_rcab_odd"])
def test_g1_rcab(name):
    run_block(name, A.RCAB(A.default_conv, 64, 3, 16), 1)


@ARITH
def test_g1_resblock():
    run_block("g1_resblock", A.ResBlock(A.default_conv, 64, 3, res_scale=0.1), 1)


@pytest.mark.parametrize("M", [1, 10, 11, 20])
@pytest.mark.parametrize("nl", [0, 1])
def test_g1_paraca(M, nl):
    run_block(f"g1_paraca_m{M}_nl{nl}", A.ParaCALayer(64, M, nonlinearity=bool(nl)), 2)


@pytest.mark.parametrize("style", ["standard", "modulate", "mini_concat", "max_concat", "softmax", "extended_attention"])
def test_g1_qca(style):
    run_block(f"g1_qca_{style}", A.QCALayer(64, style, reduction=16, num_metadata=10), 2)


@ARITH
@pytest.mark.parametrize("q", [0, 1])
@pytest.mark.parametrize("pa", [0, 1])
def test_g1_qrcab(q, pa):
    m = A.QRCAB(A.default_conv, 64, 3, 16, style="standard", pa=bool(pa), q_layer=bool(q), num_metadata=10)
    run_block(f"g1_qrcab_q{q}_pa{pa}", m, 2, call=lambda mod, i: mod((i[0], i[1]))[0])


def test_g1_palayer():
    run_block("g1_palayer", A.PALayer(64), 1)


@ARITH
@pytest.mark.parametrize("nl", [0, 1])
def test_g1_paramresblock(nl):
    m = A.ParamResBlock(A.default_conv, 64, 10, 3, res_scale=0.1, q_layer_nonlinearity=bool(nl))
    run_block(f"g1_paramresblock_nl{nl}", m, 2, call=lambda mod, i: mod((i[0], i[1]))[0])


def test_g1_han_modules():
    H = sisr_amd.han
    run_block("g1_lam", H.LAM_Module(16), 1, rtol=5e-4, atol=5e-5)
    run_block("g1_csam_c64", H.CSAM_Module(64), 1, rtol=5e-4, atol=5e-5)


def test_lam_full_size_vs_oracle():
    """11 maps of 64x32x32: the production map count, gamma != 0 so the attention branch is live."""
    H = sisr_amd.han
    m = H.LAM_Module(64)
    with torch.no_grad():
        m.gamma.fill_(0.5)
    x = rnd(2, 11, 64, 32, 32, seed=40, scale=0.05)
    xo = x.clone().requires_grad_(True)
    sd = {"la.gamma": m.gamma.detach().clone().requires_grad_(True)}
    ref = O.lam_module(sd, "la", xo)
    cot = rnd(*ref.shape, seed=41)
    ref.backward(cot)
    m.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    out = m(xg)
    out.backward(cot.to(DEV))
    close(out, ref, 5e-4, 5e-5)
    close(xg.grad, xo.grad, 2e-3, 2e-4)
    close(m.gamma.grad, sd["la.gamma"].grad, 2e-3, 2e-4)


# ----------------------------------------------------------------------------- reduced-depth nets (n_feats = 64) vs the oracle
def net_vs_oracle(net, name, cfg, x, md=None, rtol=5e-4, atol=5e-5):
    torch.manual_seed(8)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    xo = x.clone().requires_grad_(False)
    ref = O.forward(name, sd, xo, md, **cfg)
    cot = rnd(*ref.shape, seed=31)
    ref.backward(cot)
    net.to(DEV)
    out = net(x.to(DEV), md.to(DEV)) if md is not None else net(x.to(DEV))
    assert out.shape == ref.shape and out.is_contiguous()
    close(out, ref, rtol, atol, name + " out")
    out.backward(cot.to(DEV))
    for k, p in net.named_parameters():
        close(p.grad, sd[k].grad, 2e-3, 2e-4, f"{name} grad {k}")


@ARITH
def test_rcan_reduced_vs_oracle():
    torch.manual_seed(8)
    net = A.RCAN(n_resblocks=2, n_resgroups=2, n_feats=64, scale=4)
    net_vs_oracle(net, "rcan", dict(n_resgroups=2, n_resblocks=2, scale=4), rnd(2, 3, 20, 36, seed=30, scale=0.5))


@ARITH
@pytest.mark.parametrize("scale", [2, 3, 4])
def test_edsr_reduced_vs_oracle(scale):
    torch.manual_seed(8)
    net = A.EDSR(net_features=64, num_blocks=2, scale=scale, res_scale=0.1)
    net_vs_oracle(net, "edsr", dict(num_blocks=2, scale=scale, res_scale=0.1), rnd(1, 3, 17, 23, seed=32, scale=0.5))


@pytest.mark.parametrize("shape,njobs", [((4, 128, 128), 8), ((2, 37, 45), 5), ((1, 16, 32), 1), ((3, 9, 70), 3), ((2, 64, 64), 41)])
def test_batched_weight_gradients_equal_single_launches(shape, njobs):
    """ops.WgradQueue (eight 64 -> 64 weight gradients of one geometry per launch; 41 jobs = six launches, csrc/wgrad3x3_mfma.hip
    wgrad3x3_c64_batch_kernel): every job equals its own single launch to summation-order rounding (the K-split differs with
    the job count, and single launches of few tiles run the quadrant form of the kernel); jobs with and without the
    dY * scale + shift rebuild and with / without a bias."""
    B, H, W = shape
    dev = torch.device(DEV)
    cl = torch.channels_last
    hip = sisr_amd.hip
    v = hip.view_plain(H, W, 64)
    jobs, want = [], []
    for k in range(njobs):
        x = rnd(B, 64, H, W, seed=300 + k).to(dev).contiguous(memory_format=cl)
        dy = rnd(B, 64, H, W, seed=400 + k).to(dev).contiguous(memory_format=cl)
        sc = (torch.rand(B, 64, generator=torch.Generator().manual_seed(500 + k)) + 0.5).to(dev) if k % 2 else None
        sh = rnd(B, 64, seed=600 + k, scale=0.1).to(dev) if k % 2 else None
        has_b = k % 3 != 2
        dw1, db1 = torch.empty(64, 64, 3, 3, device=dev), (torch.empty(64, device=dev) if has_b else None)
        ops.wgrad_c64(x, v, dy, v, dw1, db1, B, H, W, 64, 64, dy_scale=sc, dy_shift=sh)
        want.append((dw1, db1))
        jobs.append((x, dy, sc, sh, torch.full((64, 64, 3, 3), float("nan"), device=dev),
                     torch.full((64,), float("nan"), device=dev) if has_b else None))
    q = ops.WgradQueue(B, H, W, dev)
    for x, dy, sc, sh, dw, db in jobs:
        q.add(x, dy, dw, db, dy_scale=sc, dy_shift=sh)
    q.flush()
    for (x, dy, sc, sh, dw, db), (dw1, db1) in zip(jobs, want):
        close(dw, dw1, 2e-5, 2e-6, "dw")
        if db is not None:
            close(db, db1, 2e-5, 2e-6, "db")


@pytest.mark.parametrize("L,M,nl", [(1, 10, 1), (3, 10, 1), (3, 20, 0), (4, 1, 1), (4, 20, 1)])
def test_paraca_other_depths(L, M, nl):
    """ParaCALayer with 1 / 3 / 4 FC layers (ref: attention_manipulators/q_layer.py:12-31; the published configs use 2) on the
    generic gate-MLP kernel vs the same nn.Sequential run by torch on the CPU (the reference's own formulation)."""
    torch.manual_seed(30 + L)
    mod = A.ParaCALayer(64, M, nonlinearity=bool(nl), num_layers=L)
    x, md = rnd(2, 64, 9, 11, seed=31), torch.rand(2, M, 1, 1, generator=torch.Generator().manual_seed(32))
    cot = rnd(2, 64, 9, 11, seed=33)
    xr = x.clone().requires_grad_(True)
    want = xr * mod.attribute_integrator(md)
    want.backward(cot)
    ref_g = {k: p.grad.clone() for k, p in mod.named_parameters()}
    mod.zero_grad()
    mod.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    out = mod(xg, md.to(DEV))
    close(out, want, 1e-5, 1e-6)
    out.backward(cot.to(DEV))
    close(xg.grad, xr.grad, 1e-5, 1e-6)
    for k, p in mod.named_parameters():
        close(p.grad, ref_g[k], 2e-4, 1e-5 * float(ref_g[k].abs().max()) + 1e-9, k)


@pytest.mark.parametrize("C,r", [(256, 2), (128, 3), (64, 2)])
def test_pixel_shuffle_gather_and_its_adjoint(C, r):
    """ops.pixel_shuffle (wide upsamplers, ref: advanced/common.py:20-45 nn.PixelShuffle) == torch's, bit for bit."""
    x = rnd(2, C * r * r, 5, 7, seed=90)
    want = torch.nn.functional.pixel_shuffle(x, r)
    xg = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = ops.pixel_shuffle(xg, r)
    assert torch.equal(out.cpu(), want)
    cot = rnd(*want.shape, seed=91)
    out.backward(cot.to(DEV))
    assert torch.equal(xg.grad.cpu(), torch.nn.functional.pixel_unshuffle(cot, r))


def test_stack_maps_and_its_backward():
    """ops.stack_maps (HAN's torch.cat of intermediate maps, ref: advanced/architectures.py:357-362) == torch.stack."""
    maps = [rnd(2, 64, 6, 9, seed=92 + k).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            for k in range(3)]
    stack = ops.stack_maps(maps)
    assert torch.equal(stack, torch.stack([m.permute(0, 2, 3, 1) for m in maps], dim=1))
    cot = rnd(*stack.shape, seed=99).to(DEV)
    stack.backward(cot)
    for k, m in enumerate(maps):
        assert torch.equal(m.grad, cot[:, k].permute(0, 3, 1, 2))


def test_edsr_paper_width_vs_oracle():
    """EDSR at the paper's width (n_feats = 256): multi-chunk fused ResBlock, 256 -> 1024
    upsampler convs, 256 -> 3 tail."""
    torch.manual_seed(8)
    net = A.EDSR(net_features=256, num_blocks=2, scale=4, res_scale=0.1)
    net_vs_oracle(net, "edsr", dict(num_blocks=2, scale=4, res_scale=0.1), rnd(1, 3, 18, 35, seed=37, scale=0.5))


F64_CASES = {
    "rcan": (lambda: A.RCAN(n_resblocks=3, n_resgroups=2, n_feats=64, scale=2), dict(n_resgroups=2, n_resblocks=3, scale=2)),
    "qrcan": (lambda: A.QRCAN(n_resblocks=3, n_resgroups=2, n_feats=64, scale=2, style="standard", num_metadata=10,
                              include_q_layer=True, num_q_layers_inner_residual=2),
              dict(n_resgroups=2, n_resblocks=3, scale=2, style="standard", include_q_layer=True,
                   num_q_layers_inner_residual=2)),
    "edsr": (lambda: A.EDSR(net_features=64, num_blocks=4, scale=2, res_scale=0.1), dict(num_blocks=4, scale=2, res_scale=0.1)),
    "qedsr": (lambda: A.QEDSR(num_features=64, num_blocks=4, scale=2, res_scale=0.1, input_para=10),
              dict(num_blocks=4, scale=2, res_scale=0.1, q_layer_nonlinearity=False)),
    "han": (lambda: sisr_amd.han.HAN(n_resgroups=10, n_resblocks=1, n_feats=64, scale=2),
            dict(n_resgroups=10, n_resblocks=1, scale=2)),
    "san": (lambda: sisr_amd.san.SAN(n_resgroups=2, n_resblocks=2, n_feats=64, reduction=16, scale=2),
            dict(n_resgroups=2, n_resblocks=2, scale=2)),
}


@ARITH
@pytest.mark.parametrize("kind", sorted(F64_CASES))
def test_reduced_net_gradients_against_float64_oracle(kind):
    """Tighter than the elementwise fp32 comparisons: every parameter gradient of a reduced net within 5e-5 of a FLOAT64
    evaluation of the oracle, by norm.  (Deterministic kernels and fixed seeds: no ReLU-mask element sits within fp32
    noise of zero for these inputs; a single flipped element would show as ~3e-3.  This is the test that exposed the
    fma-contracted gated skip.)"""
    torch.manual_seed(8)
    mk, cfg = F64_CASES[kind]
    net = mk()
    with torch.no_grad():  # wake the zero-initialised attention branches
        if kind == "han":
            net.la.gamma.fill_(0.37)
            net.csa.gamma.fill_(0.37)
        if kind == "san":
            g = torch.Generator().manual_seed(777)
            named = dict(net.named_parameters())
            for k in ("non_local.non_local.W.weight", "non_local.non_local.W.bias", "gamma"):
                named[k].copy_(torch.randn(named[k].shape, generator=g) * 0.2)
    meta = kind in O.META_NETS
    x, md = rnd(2, 3, 21, 30, seed=70, scale=0.5), rnd(2, 10, 1, 1, seed=71, scale=0.3)
    sd = {k: v.detach().double().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    ref = O.forward(kind, sd, x.double(), md.double() if meta else None, **cfg)
    cot = rnd(*ref.shape, seed=72)
    ref.backward(cot.double())
    net.to(DEV)
    out = net(x.to(DEV), md.to(DEV)) if meta else net(x.to(DEV))
    out.backward(cot.to(DEV))
    assert float((out.detach().double().cpu() - ref.detach()).norm()) < 3e-6 * float(ref.detach().norm())
    for k, p in net.named_parameters():
        if p.grad is None:
            assert sd[k].grad is None, k
            continue
        want = sd[k].grad
        err = float((p.grad.double().cpu() - want).norm())
        # floor: gradients that are zero by symmetry (the non-local phi bias shifts every score of a query equally)
        assert err <= 5e-5 * float(want.norm()) + 1e-5, (k, err, float(want.norm()))


@ARITH
def test_qrcan_reduced_vs_oracle():
    torch.manual_seed(8)
    net = A.QRCAN(n_resblocks=3, n_resgroups=2, n_feats=64, scale=4, style="standard", num_metadata=10,
                  include_q_layer=True, selective_meta_blocks=[True, False], num_q_layers_inner_residual=2)
    cfg = dict(n_resgroups=2, n_resblocks=3, scale=4, style="standard", include_q_layer=True,
               selective_meta_blocks=[True, False], num_q_layers_inner_residual=2)
    net_vs_oracle(net, "qrcan", cfg, rnd(2, 3, 12, 34, seed=33, scale=0.5), rnd(2, 10, 1, 1, seed=34, scale=0.3))


def _live_gammas(net):
    with torch.no_grad():  # zero-initialised gammas would switch both attention branches off
        net.la.gamma.fill_(0.37)
        net.csa.gamma.fill_(0.37)


@ARITH
def test_han_reduced_vs_oracle():
    torch.manual_seed(8)
    net = sisr_amd.han.HAN(n_resgroups=10, n_resblocks=1, n_feats=64, scale=4)
    _live_gammas(net)
    net_vs_oracle(net, "han", dict(n_resgroups=10, n_resblocks=1, scale=4), rnd(1, 3, 12, 20, seed=37, scale=0.5))


@ARITH
def test_qhan_reduced_vs_oracle():
    torch.manual_seed(8)
    net = sisr_amd.han.QHAN(n_resgroups=10, n_resblocks=1, n_feats=64, num_metadata=10, scale=4)
    _live_gammas(net)
    net_vs_oracle(net, "qhan", dict(n_resgroups=10, n_resblocks=1, scale=4), rnd(2, 3, 12, 20, seed=38, scale=0.5),
                  rnd(2, 10, 1, 1, seed=39, scale=0.3))


def test_qrcan_pixel_attention_and_modulate_vs_oracle():
    torch.manual_seed(8)
    net = A.QRCAN(n_resblocks=2, n_resgroups=2, n_feats=64, scale=2, style="standard", num_metadata=10,
                  include_q_layer=True, include_pixel_attention=True)
    cfg = dict(n_resgroups=2, n_resblocks=2, scale=2, style="standard", include_q_layer=True,
               include_pixel_attention=True)
    net_vs_oracle(net, "qrcan", cfg, rnd(2, 3, 9, 21, seed=43, scale=0.5), rnd(2, 10, 1, 1, seed=44, scale=0.3))
    torch.manual_seed(8)
    net = A.QRCAN(n_resblocks=2, n_resgroups=1, n_feats=64, scale=2, style="modulate", num_metadata=1)
    cfg = dict(n_resgroups=1, n_resblocks=2, scale=2, style="modulate")
    net_vs_oracle(net, "qrcan", cfg, rnd(2, 3, 9, 21, seed=45, scale=0.5), rnd(2, 64, 1, 1, seed=46, scale=0.3).abs())


@ARITH
def test_qedsr_reduced_vs_oracle():
    torch.manual_seed(8)
    net = A.QEDSR(num_features=64, num_blocks=2, scale=4, res_scale=0.1, input_para=10)
    net_vs_oracle(net, "qedsr", dict(num_blocks=2, scale=4, res_scale=0.1, q_layer_nonlinearity=False),
                  rnd(2, 3, 12, 34, seed=35, scale=0.5), rnd(2, 10, 1, 1, seed=36, scale=0.3))


# ----------------------------------------------------------------------------- full depth vs the reference (G3 / G4)
from test_init_parity import PARAMS, set5  # noqa: E402


def build_gpu(name, eval_mode=True, **extra):
    torch.manual_seed(8)
    return sisr_amd.handlers.available_models[name](device=0, model_save_dir="/tmp", eval_mode=eval_mode, scale=4,
                                                    **PARAMS[name], **extra)


@pytest.mark.parametrize("name", [n for n in ("edsr", "rcan", "qedsr", "qrcan", "han", "qhan")
                                  if n in sisr_amd.available_models])
@ARITH
def test_set5_forward_psnr_parity_with_reference(name):
    """Seed-8 full-depth net on every Set5 LR image: Y-PSNR within 1e-3 dB of the reference's CPU output."""
    ref = golden_json("g3_full_depth")[name]["images"]
    crops = np.load(f"{sisr_amd.__path__[0]}/../tests/golden/g3_{name}_crops.npz")
    h = build_gpu(name)
    for im, x, y, md in set5():
        kw = dict(metadata=md, metadata_keys=[("blur_kernel",)] * 10) if "metadata" in PARAMS[name] else {}
        out, loss, _ = h.run_eval(x, y, request_loss=True, **kw)
        o = out[0].numpy()
        assert abs(sisr_amd.metrics.y_psnr(o, y[0].numpy()) - ref[im]["y_psnr"]) < 1e-3, im
        assert abs(float(loss) - ref[im]["l1"]) < 1e-5
        hh, ww = o.shape[1:]
        close(o[:, hh // 2 - 16:hh // 2 + 16, ww // 2 - 16:ww // 2 + 16], crops[im], 1e-3, 1e-4, im)


@ARITH
@pytest.mark.parametrize("name", [n for n in ("edsr", "qedsr", "rcan", "qrcan") if n in sisr_amd.available_models])
def test_run_train_trajectory_matches_reference(name):
    """5 handler.run_train steps (L1 + backward + Adam + per-batch cosine restarts) vs the reference's."""
    ref = golden_json("g4_train_steps")[name]
    h = build_gpu(name, eval_mode=False, lr=1e-4, grad_clip=ref["grad_clip"], scheduler=ref["scheduler"],
                  scheduler_params=ref["scheduler_params"])
    g = torch.Generator().manual_seed(77)
    # Flat per-step tolerances -- 5e-6 loss, 1e-4 gradient norm, 5e-5 output mean, the same at every step (measured on
    # MI355X once the gated skip was made fma-free: <= 9e-7 / 1.9e-5 / 1.1e-5 over all four models and five steps) --
    # plus, per model and step, what the REFERENCE drifts from ITSELF between torch thread counts (fixtures
    # g4_train_steps_t{1,3}.json, generated by tools/make_fixtures.py g4t at 1 and 3 threads against the 8-thread G4):
    # EDSR / QEDSR / QRCAN reproduce themselves to 1e-5, full-depth RCAN at this tile size is chaotic -- its own step-4
    # loss / gradient norm / output mean move by 2.9e-5 / 3.4e-4 / 2.7e-4 -- and a trajectory cannot be pinned tighter
    # than the reference pins itself.
    alt = [golden_json(f"g4_train_steps_t{t}")[name]["steps"] for t in (1, 3)]

    def ref_drift(i, key, rel=False):
        base = ref["steps"][i][key]
        return max(abs(a[i][key] / base - 1) if rel else abs(a[i][key] - base) for a in alt)

    rows = []
    for i, step in enumerate(ref["steps"]):
        x = torch.rand(2, 3, 16, 16, generator=g)
        y = torch.rand(2, 3, 64, 64, generator=g)
        md = torch.rand(2, 10, generator=g, dtype=torch.float64) * 0.4
        kw = dict(metadata=md, metadata_keys=[("blur_kernel", "blur_kernel")] * 10) if "metadata" in PARAMS[name] else {}
        assert abs(h.get_learning_rate() - step["lr_before"]) < 1e-12
        loss, out = h.run_train(x, y, **kw)
        gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in h.net.parameters())))
        assert abs(h.get_learning_rate() - step["lr_after"]) < 1e-12
        rows.append((i, float(loss) - step["loss"], gn / step["grad_norm"] - 1, float(out.mean()) - step["out_mean"]))
    print(name, "trajectory (step, dloss, rel dgradnorm, dmean):", rows)
    try:
        import json
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"trajectory_{name}.json"), "w") as f:
            json.dump({"rows": rows, "ref_drift": [(ref_drift(i, "loss"), ref_drift(i, "grad_norm", True),
                                                    ref_drift(i, "out_mean")) for i in range(len(rows))]}, f)
    except OSError:
        pass
    for i, dl, dg, dm in rows:
        assert abs(dl) < 5e-6 + ref_drift(i, "loss"), rows
        assert abs(dg) < 1e-4 + ref_drift(i, "grad_norm", rel=True), rows
        assert abs(dm) < 5e-5 + ref_drift(i, "out_mean"), rows
    psum = float(sum(v.double().sum() for v in h.net.state_dict().values()))
    assert abs(psum - ref["final_param_sum"]) < 5e-2


def test_native_library_is_loaded():
    """The tests above ran on libsisr_hip.so, not on a fallback."""
    with open("/proc/self/maps") as f:
        assert "libsisr_hip.so" in f.read()


def _run_dp_worker(tmp_path, model, mode, steps, ranks, alt=0, port=29731, bucket_mb=None, overlap="auto"):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / f"{model}_{mode}_{ranks}_{alt}.pt")
    worker = os.path.join(root, "tests", "_dp_worker.py")
    env = dict(os.environ, SISR_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    if bucket_mb is not None:
        env["SISR_DP_BUCKET_MB"] = str(bucket_mb)
    env["SISR_GRAPH_OVERLAP"] = overlap  # auto = all buckets at the join after a replay; "1" = released by signal nodes
    args = [worker, out, model, mode, str(steps), str(alt)]
    if ranks == 1:
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
               "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    return torch.load(out, weights_only=True)


def _assert_same_grads(one, two, tol, what):
    assert len(one["grads"]) == len(two["grads"])
    for step, (g1, g2) in enumerate(zip(one["grads"], two["grads"])):
        assert g1.keys() == g2.keys()
        assert abs(one["loss"][step] - two["loss"][step]) < 1e-6, (what, step, one["loss"], two["loss"])
        worst = max(((g2[n] - g1[n]).double().norm().item() / (g1[n].double().norm().item() + 1e-30), n) for n in g1)
        assert worst[0] < tol, (what, step, worst)
    # consecutive steps see different batches: a reducer / graph handing out a stale gradient would repeat itself
    n0 = next(iter(two["grads"][0]))
    assert two["grads"][0][n0].shape != two["grads"][2][n0].shape or (two["grads"][0][n0] - two["grads"][2][n0]).abs().max() > 0


@pytest.mark.parametrize("model,mode,alt", [("qrcan", "eager", 0), ("qrcan", "graph", 0), ("rcan", "eager_noside", 0),
                                            ("edsr", "graph", 1)])
def test_two_ranks_on_one_gpu_equal_one_process_gradients(tmp_path, model, mode, alt):
    """SURVEY 8e equivalence, on HIP tensors: two processes on cuda:0 (gloo transport), each with its contiguous half of
    a 4-tile global batch, through GradReducer -- weight-gradient kernels writing straight into the all-reduce buckets
    (ops.GRAD_SINK) on the side stream, bucket hooks, join -- give every parameter the gradient a single process computes
    on the whole batch (mean of equal shards' mean losses is the global mean; only the summation order over samples
    differs: <= 2e-6 of each gradient's norm), for four consecutive steps with different batches.  'graph' replays
    forward+backward from a hipGraph with the reducer joined after it; alt = 1 alternates two batch shapes."""
    port = 29731 + sum(map(ord, model + mode)) % 40
    one = _run_dp_worker(tmp_path, model, "eager", 4, 1, alt)
    two = _run_dp_worker(tmp_path, model, mode, 4, 2, alt, port)
    _assert_same_grads(one, two, 2e-6, (model, mode, alt))


@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_two_ranks_many_buckets_equal_one_process_gradients(tmp_path, mode):
    """The same equivalence with 0.25 MB buckets (>= 4 of them on the reduced QRCAN): in eager mode the hooks fire mid-backward
    and bucket k is all-reduced on the reducer stream while the side-stream weight-gradient kernels are still writing
    bucket k + 1 of the same arena; in graph mode the replay's signal nodes release the buckets one by one
    (GradReducer.launch_signalled) while the rest of the captured backward runs.  Same 2e-6 bound, four different batches."""
    one = _run_dp_worker(tmp_path, "qrcan", "eager", 4, 1, 0)
    two = _run_dp_worker(tmp_path, "qrcan", mode, 4, 2, 0, 29777 + (mode == "graph"), bucket_mb=0.25, overlap="1")
    assert two["buckets"] >= 4, two["buckets"]
    if mode == "graph":
        assert two["signalled"] and min(two["signalled"]) >= 4, two["signalled"]  # the overlapped path, not the join
    _assert_same_grads(one, two, 2e-6, ("qrcan", mode, "many buckets"))


def test_hip_graph_two_batch_shapes_alternating(tmp_path):
    """One process, use_graph, batch shapes A B A B: every replay's gradients are the eager ones (the second capture
    re-points p.grad; the first graph's replay must bind it back to the tensors its kernels write)."""
    eager = _run_dp_worker(tmp_path, "qrcan", "eager", 4, 1, 1)
    graph = _run_dp_worker(tmp_path, "qrcan", "graph", 4, 1, 1)
    _assert_same_grads(eager, graph, 1e-6, "graph, alternating shapes")


def test_rccl_backend_single_rank_grad_reducer():
    """The nccl (= RCCL) transport of GradReducer on the one GPU this box has: a world of one rank still runs
    init_process_group('nccl'), the bucketed asynchronous all_reduce on the side stream and the join, and must
    reproduce the plain single-process losses bit for bit (sum over one rank, times 1/1)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, json, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {root!r})\n"
        "import sisr_amd\n"
        "use_dp = sys.argv[1] == '1'\n"
        "torch.cuda.set_device(0)\n"
        "if use_dp:\n"
        "    dist.init_process_group(backend='nccl', rank=0, world_size=1)\n"
        "torch.manual_seed(8)\n"
        "h = sisr_amd.available_models['edsr'](device=0, model_save_dir='/tmp', eval_mode=False, scale=4, lr=1e-4)\n"
        "if use_dp:\n"
        "    h.set_multi_gpu()\n"
        "g = torch.Generator().manual_seed(3)\n"
        "out = []\n"
        "for _ in range(3):\n"
        "    x, y = torch.rand(2, 3, 24, 24, generator=g), torch.rand(2, 3, 96, 96, generator=g)\n"
        "    loss, _ = h.run_train(x, y)\n"
        "    out.append(float(loss))\n"
        "print(json.dumps(out))\n"
        "if use_dp:\n"
        "    dist.destroy_process_group()\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29741")
    res = {}
    for flag in ("0", "1"):
        out = subprocess.run([sys.executable, "-c", code, flag], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        res[flag] = [ln for ln in out.stdout.splitlines() if ln.startswith("[")][-1]
    assert res["0"] == res["1"], res


@ARITH
def test_set5_training_psnr_parity_at_equal_steps():
    """north_star: 'PSNR on Set5 within 0.02 dB of the reference at equal steps'.  EDSR-baseline (16 blocks,
    full depth) trained for 40 Adam steps on seeded Set5 crops, once on the HIP kernels and once by the CPU
    oracle (itself pinned to the reference), same init / batches / schedule; then full-image Set5 Y-PSNR."""
    import random
    torch.manual_seed(8)
    h = build_gpu("edsr", eval_mode=False, lr=1e-4, scheduler="cosine_annealing_warm_restarts",
                  scheduler_params={"t_mult": 1, "restart_period": 25, "lr_min": 1e-7})
    tr = O.Trainer("edsr", {k: v.detach().cpu() for k, v in h.net.state_dict().items()}, lr=1e-4,
                   scheduler="cosine_annealing_warm_restarts",
                   scheduler_params={"t_mult": 1, "restart_period": 25, "lr_min": 1e-7}, num_blocks=16, scale=4,
                   res_scale=0.1)
    images = [(x[0], y[0]) for _, x, y, _ in set5()]
    rng = random.Random(3)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    for step in range(40):
        xs, ys = [], []
        for _ in range(4):
            x, y = images[rng.randrange(5)]
            i, j = rng.randrange(x.shape[1] - 31), rng.randrange(x.shape[2] - 31)
            xs.append(x[:, i:i + 32, j:j + 32])
            ys.append(y[:, 4 * i:4 * i + 128, 4 * j:4 * j + 128])
        xb, yb = torch.stack(xs), torch.stack(ys)
        loss_hip, _ = h.run_train(xb, yb)
        loss_ref, _, _ = tr.step(xb, yb)
        assert abs(float(loss_hip) - loss_ref) < 2e-3, (step, float(loss_hip), loss_ref)
    diffs = []
    with torch.no_grad():
        for x, y in images:
            out, _, _ = h.run_eval(x[None])
            ref = O.edsr(tr.sd, x[None], num_blocks=16, scale=4, res_scale=0.1)
            p_hip = sisr_amd.metrics.y_psnr(out[0].numpy(), y.numpy())
            p_ref = O.y_psnr(ref[0].numpy(), y.numpy())
            diffs.append(p_hip - p_ref)
    print("Set5 Y-PSNR (HIP - oracle) after 40 steps:", diffs)
    assert max(abs(d) for d in diffs) < 0.02, diffs


@ARITH
@pytest.mark.parametrize("groups,blocks", [(2, 3), (10, 20)])
def test_set5_training_psnr_parity_meta_rcan(groups, blocks):
    """The same for north_star's target family -- RCAN + meta-attention (QRCAN, style 'standard', q-layers on), at a
    reduced depth (2 groups x 3 blocks) and at FULL depth (10 x 20, the BASELINE config): 20 Adam steps through the
    group-level fused node with the blur-kernel metadata of the Set5 example data, the CPU oracle taking the same steps,
    then Set5 Y-PSNR within 0.02 dB."""
    import random
    torch.manual_seed(8)
    kw = dict(metadata=["blur_kernel"], style="standard", include_q_layer=True, n_resgroups=groups, n_resblocks=blocks)
    h = sisr_amd.handlers.available_models["qrcan"](device=0, model_save_dir="/tmp", eval_mode=False, scale=4, lr=1e-4, **kw)
    cfg = dict(n_resgroups=groups, n_resblocks=blocks, scale=4, style="standard", include_q_layer=True)
    tr = O.Trainer("qrcan", {k: v.detach().cpu() for k, v in h.net.state_dict().items()}, lr=1e-4, **cfg)
    data = [(x[0], y[0], md[0]) for _, x, y, md in set5()]
    rng = random.Random(4)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    keys = [("blur_kernel",) * 4] * 10
    for step in range(20):
        xs, ys, ms = [], [], []
        for _ in range(4):
            x, y, m = data[rng.randrange(5)]
            i, j = rng.randrange(x.shape[1] - 31), rng.randrange(x.shape[2] - 31)
            xs.append(x[:, i:i + 32, j:j + 32])
            ys.append(y[:, 4 * i:4 * i + 128, 4 * j:4 * j + 128])
            ms.append(m)
        xb, yb, mb = torch.stack(xs), torch.stack(ys), torch.stack(ms)
        loss_hip, _ = h.run_train(xb, yb, metadata=mb, metadata_keys=keys)
        loss_ref, _, _ = tr.step(xb, yb, mb.float().view(4, 10, 1, 1))
        assert abs(float(loss_hip) - loss_ref) < 2e-3, (step, float(loss_hip), loss_ref)
    diffs = []
    with torch.no_grad():
        for x, y, m in data:
            out, _, _ = h.run_eval(x[None], metadata=m[None], metadata_keys=[("blur_kernel",)] * 10)
            ref = O.qrcan(tr.sd, x[None], m.float().view(1, 10, 1, 1), groups, blocks, 4, "standard", False, True)
            diffs.append(sisr_amd.metrics.y_psnr(out[0].numpy(), y.numpy()) - O.y_psnr(ref[0].numpy(), y.numpy()))
    print(f"Set5 Y-PSNR (HIP - oracle), meta-RCAN {groups}x{blocks} after 20 steps:", diffs)
    assert max(abs(d) for d in diffs) < 0.02, diffs


# ----------------------------------------------------------------------------- size-independent properties, edge cases
def test_adjoint_identities_at_bench_size():
    """<conv(x), dy> = <x, dgrad(dy)> = <w, wgrad(x, dy)> + <b, sum dy> on a full 8x64x128x128 map: the three
    kernels are mutually consistent at production size (fp64 accumulation of the inner products)."""
    g = torch.Generator(device="cpu").manual_seed(50)
    x = torch.randn(8, 64, 128, 128, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(DEV).requires_grad_(True)
    b = torch.randn(64, generator=g).to(DEV).requires_grad_(True)
    dy = torch.randn(8, 64, 128, 128, generator=g).to(DEV)
    y = ops.conv3x3(x, w, b)
    y.backward(dy)
    lhs = (y.double() * dy.double()).sum().item()
    via_x = (x.detach().double() * x.grad.double()).sum().item() + (b.detach().double() * b.grad.double()).sum().item()
    via_w = (w.detach().double() * w.grad.double()).sum().item() + (b.detach().double() * b.grad.double()).sum().item()
    assert abs(lhs - via_x) < 1e-6 * abs(lhs) + 1e-2 and abs(lhs - via_w) < 1e-6 * abs(lhs) + 1e-2, (lhs, via_x, via_w)
    # linearity in the input
    x2 = torch.randn(8, 64, 128, 128, generator=g).to(DEV)
    with torch.no_grad():
        close(ops.conv3x3(x.detach() + 2 * x2, w, None), ops.conv3x3(x.detach(), w, None) + 2 * ops.conv3x3(x2, w, None),
              1e-4, 2e-5)


def test_backward_is_bitwise_reproducible():
    """No atomics anywhere: two identical training steps give bit-identical gradients (side stream included)."""
    grads = []
    for _ in range(2):
        torch.manual_seed(8)
        net = A.RCAN(n_resblocks=2, n_resgroups=2, n_feats=64, scale=4).to(DEV)
        x = rnd(2, 3, 40, 72, seed=60).to(DEV)
        y = rnd(2, 3, 160, 288, seed=61).to(DEV)
        ops.l1_loss(net(x), y).backward()
        torch.cuda.synchronize()
        grads.append(torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu())
    assert torch.equal(grads[0], grads[1])


def test_large_odd_image_eval_vs_oracle():
    torch.manual_seed(8)
    net = A.RCAN(n_resblocks=1, n_resgroups=1, n_feats=64, scale=2)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    x = rnd(1, 3, 203, 311, seed=62, scale=0.5)
    with torch.no_grad():
        ref = O.rcan(sd, x, n_resgroups=1, n_resblocks=1, scale=2)
        out = net.to(DEV)(x.to(DEV))
    assert out.shape == (1, 3, 406, 622)
    close(out, ref, 2e-4, 2e-5)


def test_invalid_inputs_fail_loudly():
    w = torch.zeros(64, 64, 3, 3, device=DEV)
    with pytest.raises(RuntimeError, match="bad argument"):
        ops.conv3x3(torch.zeros(0, 64, 8, 8, device=DEV), w)          # empty batch
    with pytest.raises(NotImplementedError, match="multiples of 64"):
        ops.conv3x3(torch.zeros(1, 48, 8, 8, device=DEV), torch.zeros(48, 48, 3, 3, device=DEV))
    with pytest.raises(RuntimeError, match="fp32"):
        ops.conv3x3(torch.zeros(1, 64, 8, 8, device=DEV, dtype=torch.bfloat16), w)
    with pytest.raises(RuntimeError, match="shape mismatch"):
        ops.l1_loss(torch.zeros(1, 3, 8, 8, device=DEV), torch.zeros(1, 3, 8, 9, device=DEV))
    # 1x1 spatial extent and a single pixel column are legal
    x = rnd(1, 64, 1, 1, seed=63)
    close(ops.conv3x3(x.to(DEV), w + 0.01), F.conv2d(x, w.cpu() + 0.01, padding=1), 1e-5, 1e-6)
    x = rnd(2, 64, 37, 1, seed=64)
    close(ops.conv3x3(x.to(DEV), w + 0.01), F.conv2d(x, w.cpu() + 0.01, padding=1), 1e-5, 1e-6)


def test_hip_graph_replay_matches_eager():
    """BaseModel.use_graph: forward+loss+backward replayed from a hipGraph gives the eager trajectory."""
    losses = {}
    for mode in (False, True):
        torch.manual_seed(8)
        h = sisr_amd.handlers.EDSRHandler(device=0, model_save_dir="/tmp", eval_mode=False, num_blocks=3, lr=1e-4)
        h.use_graph = mode
        g = torch.Generator().manual_seed(5)
        ls = []
        for _ in range(4):
            x, y = torch.rand(2, 3, 24, 40, generator=g), torch.rand(2, 3, 96, 160, generator=g)
            loss, out = h.train_step(x, y)
            ls.append(float(loss))
        losses[mode] = (ls, float(sum(p.double().sum() for p in h.net.parameters())))
    assert np.allclose(losses[True][0], losses[False][0], rtol=0, atol=1e-6), losses
    assert abs(losses[True][1] - losses[False][1]) < 1e-5


# ----------------------------------------------------------------------------- deferred weight-gradient queue: guards
def _two_conv_grads(shared, deferred, fail=False):
    g = torch.Generator(device="cpu").manual_seed(70)
    x = torch.randn(2, 64, 24, 40, generator=g).to(DEV)
    w1 = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(DEV).requires_grad_(True)
    w2 = w1 if shared else (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(DEV).requires_grad_(True)
    b1 = torch.randn(64, generator=g).to(DEV).requires_grad_(True)
    b2 = torch.randn(64, generator=g).to(DEV).requires_grad_(True)
    dy = torch.randn(2, 64, 24, 40, generator=g).to(DEV)
    y = ops.conv3x3(ops.conv3x3(x, w1, b1), w2, b2)
    try:
        with ops.deferred_wgrads(enabled=deferred):
            y.backward(dy)
            if fail:
                raise KeyError("the step dies after backward, inside the block")
    except KeyError:
        assert fail
    torch.cuda.synchronize()
    return [t.grad.detach().cpu().clone() for t in ((w1, b1, b2) if shared else (w1, w2, b1, b2))]


def test_deferred_wgrad_block_flushes_on_exceptional_exit():
    """A deferred_wgrads() block that exits on an exception still launches what it queued: the gradients autograd adopted
    hold their values (not uninitialised memory) when the block has closed."""
    assert ops.WgradQueue.wanted(2, 24, 40)
    plain = _two_conv_grads(False, False)
    normal = _two_conv_grads(False, True)
    died = _two_conv_grads(False, True, fail=True)
    for a, b, c in zip(plain, normal, died):
        assert torch.equal(b, c)  # the same batched launches either way: bit-identical
        # single launches of this size run the quadrant form of the kernel, the batch the full-tile form: another summation order
        assert torch.allclose(a, c, rtol=2e-5, atol=2e-5 * float(a.abs().max()))


def test_deferred_wgrad_queue_takes_a_weight_once_per_pass():
    """One 64 -> 64 weight used by two convs of the same backward pass: the second weight gradient is not queued (autograd
    adds the two as soon as the second arrives) and the first is flushed before it; result = the undeferred one."""
    plain = _two_conv_grads(True, False)
    queued = _two_conv_grads(True, True)
    for a, b in zip(plain, queued):
        assert torch.allclose(a, b, rtol=2e-5, atol=2e-5 * float(a.abs().max()))  # (kernel forms differ: see above)
    ref_w = plain[0].double()
    assert torch.isfinite(ref_w).all() and ref_w.abs().max() > 0
