"""CPU oracle for the SISR residual-conv + channel/meta-attention hot path.

TEST INFRASTRUCTURE ONLY.  This file is a functional, plain-PyTorch (fp32, CPU)
restatement of the reference's forward passes for the path SURVEY.md §8 names.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
it; the product package never does and fails loudly without its HIP library.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here
against golden vectors produced by running the *imported reference itself* in
the build container (tools/make_fixtures.py -> tests/golden/*.npz).

Style: stateless functions over a flat ``{reference state_dict key: tensor}``
mapping, so a reference checkpoint (``state['network']``) drives it directly and
autograd on those tensors yields the reference's parameter gradients.  The
arithmetic semantics relied upon are PyTorch's (the reference's only numeric
dependency, requirements.txt:1): zero-padded cross-correlation, mean pooling,
PixelShuffle channel order, Softmax(dim=-1), L1Loss(mean), Adam.

All ``ref:`` citations are relative to /root/reference/Code/.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- primitives
# Arithmetic of the 64-channel-multiple 3x3 convolutions.  "fp32" is the reference's.  "bf16" restates what the
# product's bf16 matrix-core mode computes: both operands of the forward, input-gradient and
# weight-gradient contractions rounded to bf16 (round-to-nearest-even), exact products, fp32 accumulation, fp32
# storage; bias gradient from the unrounded output gradient.  The reference has no such mode: vectors for it are
# "parity unpinned" against the reference and pinned only against this restatement.
CONV_PRECISION = "fp32"
# Storage of the maps a residual group keeps, in the product's bf16 mode ("bf16 storage"): "fp32", or
# "bf16" = the group's input copy, t1 = ReLU(conv1), t2 = conv2 and the gated skips u_k are rounded to bf16 (nearest even)
# where they are written -- what sits in HBM and what every later launch reads -- while the channel-attention pooling takes
# the unrounded conv2 output (the kernel sums its fp32 accumulators), gradient maps stay fp32 and pass the rounding unchanged.
# Restated for the blocks the product runs as ONE group node (RCAB stacks, QRCAB 'standard' without pixel attention).
MAP_STORAGE = "fp32"


def _r16(t):
    return t.to(torch.bfloat16).to(torch.float32)


class _StoreBf16(torch.autograd.Function):
    """A map written to HBM as bf16: the value is rounded, the gradient passes (the backward kernels rebuild gradients in fp32)."""

    @staticmethod
    def forward(ctx, x):
        return _r16(x)

    @staticmethod
    def backward(ctx, g):
        return g


def _st(x):
    return _StoreBf16.apply(x) if (MAP_STORAGE == "bf16" and CONV_PRECISION == "bf16") else x


# ... and of the gradient maps a group's backward pass hands from launch to launch ("bf16": the gradient at every gated skip
# u_k (k >= 1) and at every conv1 output is rounded to bf16 once, after the skip's gradient has been added / the ReLU mask
# applied in fp32; the group's input gradient is not).  One difference to the product is left in this restatement: the
# product takes the gate gradient's sums from the fp32 value BEFORE that rounding, autograd below from the rounded map.
GRAD_STORAGE = "fp32"


class _StoreGrad16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _r16(g)


def _gr(x):
    return _StoreGrad16.apply(x) if (GRAD_STORAGE == "bf16" and MAP_STORAGE == "bf16" and CONV_PRECISION == "bf16") else x


class _Bf16Conv(torch.autograd.Function):
    """alpha * (conv(r16(x), r16(w)) + b).  A residual scale (EDSR's res_scale) is part of the operator because the
    product applies it to the fp32 accumulator, i.e. AFTER the bf16 rounding of the incoming gradient in backward;
    scaling outside would round alpha * dy instead of dy."""

    @staticmethod
    def forward(ctx, x, w, b, alpha):
        xr, wr = _r16(x), _r16(w)
        ctx.save_for_backward(xr, wr)
        ctx.has_bias, ctx.alpha = b is not None, alpha
        return F.conv2d(xr, wr, b, padding=w.shape[-1] // 2) * alpha

    @staticmethod
    def backward(ctx, dy):
        xr, wr = ctx.saved_tensors
        dyr = _r16(dy)
        pad = wr.shape[-1] // 2
        dx = torch.nn.grad.conv2d_input(xr.shape, wr, dyr, padding=pad) * ctx.alpha
        dw = torch.nn.grad.conv2d_weight(xr, wr.shape, dyr, padding=pad) * ctx.alpha
        return dx, dw, (dy.sum(dim=(0, 2, 3)) * ctx.alpha if ctx.has_bias else None), None


def conv(sd, key, x, alpha=1.0):
    """ref: SISR/models/advanced/common.py:5-8 (default_conv): k x k, pad k//2, bias.  alpha: output scale the
    caller applies right after (ResBlock's res_scale); in fp32 it is just that multiplication."""
    w = sd[key + ".weight"]
    if CONV_PRECISION == "bf16" and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0:
        return _Bf16Conv.apply(x, w, sd.get(key + ".bias"), alpha)
    y = F.conv2d(x, w, sd.get(key + ".bias"), padding=w.shape[-1] // 2)
    return y if alpha == 1.0 else y * alpha


def _fc(sd, key, v):
    """1x1 conv on a (B,C,1,1) vector == dense layer; kept as conv2d to share ATen path."""
    return F.conv2d(v, sd[key + ".weight"], sd[key + ".bias"])


def gap(x):
    """ref: advanced/architectures.py:21 nn.AdaptiveAvgPool2d(1)."""
    return F.adaptive_avg_pool2d(x, 1)


def upsampler(sd, key, x, scale):
    """ref: advanced/common.py:20-45: [conv C->4C, PixelShuffle(2)] x log2(scale) or conv C->9C + PS(3).
    No activation, no BN on every in-scope model (act=False)."""
    if scale & (scale - 1) == 0:
        for i in range(int(math.log2(scale))):
            x = F.pixel_shuffle(conv(sd, f"{key}.{2 * i}", x), 2)
        return x
    if scale == 3:
        return F.pixel_shuffle(conv(sd, f"{key}.0", x), 3)
    raise NotImplementedError(scale)


# ----------------------------------------------------------------------------- attention gates
def ca_gate(sd, key, x):
    """sigmoid(W2 relu(W1 gap(x))) -> (B,C,1,1).  ref: advanced/architectures.py:23-31."""
    y = F.relu(_fc(sd, key + ".conv_du.0", gap(x)))
    return torch.sigmoid(_fc(sd, key + ".conv_du.2", y))


def ca_layer(sd, key, x):
    """ref: advanced/architectures.py:13-32 CALayer.forward."""
    return x * ca_gate(sd, key, x)


def para_ca_gate(sd, key, attributes, nonlinearity, num_layers=2):
    """Meta-attention gate.  ref: attention_manipulators/q_layer.py:20-43.
    Layer list is [conv, (relu,)]* + sigmoid, so conv indices depend on ``nonlinearity``."""
    y = attributes
    idx = 0
    multiplier = num_layers
    for _ in range(num_layers):
        y = _fc(sd, f"{key}.attribute_integrator.{idx}", y)
        idx += 1
        if nonlinearity and multiplier != 1:
            y = F.relu(y)
            idx += 1
        multiplier -= 1
    return torch.sigmoid(y)


def para_ca_layer(sd, key, x, attributes, nonlinearity, num_layers=2):
    """ref: q_layer.py:39-43."""
    return x * para_ca_gate(sd, key, attributes, nonlinearity, num_layers)


def para_ca_widths(channels, num_metadata, num_layers=2):
    """FC widths [M, h1, ..., C].  ref: q_layer.py:22-31 (M>15 uses the offset formula)."""
    widths = [num_metadata]
    mult = num_layers
    for _ in range(num_layers):
        if num_metadata > 15:
            widths.append((channels - num_metadata) // mult + num_metadata)
        else:
            widths.append(channels // mult)
        mult -= 1
    return widths


def qca_layer(sd, key, x, attributes, style):
    """Six-style combined channel/meta attention.  ref: attention_manipulators/architectures.py:105-127."""
    y = gap(x)
    if style == "standard":
        y = torch.sigmoid(_fc(sd, key + ".conv_du.2", F.relu(_fc(sd, key + ".conv_du.0", y))))
    elif style == "modulate":
        y = torch.sigmoid(_fc(sd, key + ".conv_du.2", F.relu(_fc(sd, key + ".conv_du.0", y)))) * attributes
    elif style in ("max_concat", "softmax"):
        y = torch.cat((y, attributes), dim=1)
        y = torch.sigmoid(_fc(sd, key + ".conv_du.2", F.relu(_fc(sd, key + ".conv_du.0", y))))
        if style == "softmax":
            y = torch.softmax(y, dim=1)
    elif style == "mini_concat":
        y = _fc(sd, key + ".pre_concat", y)
        y = torch.cat((y, attributes), dim=1)
        # conv_du = [ReLU, conv, Sigmoid]: the ReLU also hits the concatenated metadata
        y = torch.sigmoid(_fc(sd, key + ".conv_du.1", F.relu(y)))
    elif style == "extended_attention":
        for i in range(3):
            y = F.relu(_fc(sd, f"{key}.feature_convs.{i}.0", torch.cat((y, attributes), dim=1)))
        y = torch.sigmoid(_fc(sd, key + ".final_conv.0", y))
    else:
        raise NotImplementedError(style)
    return x * y


def pa_layer(sd, key, x):
    """Per-pixel attention 64->8->1.  ref: attention_manipulators/architectures.py:13-26."""
    y = F.relu(F.conv2d(x, sd[key + ".pa.0.weight"], sd[key + ".pa.0.bias"]))
    y = torch.sigmoid(F.conv2d(y, sd[key + ".pa.2.weight"], sd[key + ".pa.2.bias"]))
    return x * y


# ----------------------------------------------------------------------------- blocks
def res_block(sd, key, x, res_scale):
    """ref: advanced/common.py:68-72: x + res_scale * conv(relu(conv(x)))."""
    return conv(sd, key + ".body.2", F.relu(conv(sd, key + ".body.0", x)), alpha=res_scale) + x


def rcab(sd, key, x):
    """ref: advanced/architectures.py:68-71.  res_scale is stored but never applied."""
    r = conv(sd, key + ".body.2", F.relu(conv(sd, key + ".body.0", x)))
    return ca_layer(sd, key + ".body.3", r) + x


def _stored_group(x, n_resblocks, block_key, tail_key, sd, gate_fn):
    """A group of channel-attention blocks with its kept maps in bf16 (MAP_STORAGE): gate_fn(k, t2_unrounded) -> (B,C,1,1)."""
    u = _st(x)
    for i in range(n_resblocks):
        key = block_key(i)
        t1 = _st(F.relu(_gr(conv(sd, key + ".body.0", u))))
        t2 = conv(sd, key + ".body.2", t1)
        u = _gr(_st(_st(t2) * gate_fn(i, t2) + u))
    return conv(sd, tail_key, u) + x


def residual_group(sd, key, x, n_resblocks):
    """ref: advanced/architectures.py:107-110."""
    if MAP_STORAGE == "bf16" and CONV_PRECISION == "bf16" and x.shape[1] == 64:
        return _stored_group(x, n_resblocks, lambda i: f"{key}.body.{i}", f"{key}.body.{n_resblocks}", sd,
                             lambda i, t2: ca_gate(sd, f"{key}.body.{i}.body.3", t2))
    r = x
    for i in range(n_resblocks):
        r = rcab(sd, f"{key}.body.{i}", r)
    return conv(sd, f"{key}.body.{n_resblocks}", r) + x


def qrcab(sd, key, x, md, style, pa, q_layer):
    """ref: attention_manipulators/architectures.py:172-180."""
    r = conv(sd, key + ".body.2", F.relu(conv(sd, key + ".body.0", x)))
    r = qca_layer(sd, key + ".final_body", r, md, style)
    if pa:
        r = pa_layer(sd, key + ".pa_node", r)
    if q_layer:
        r = para_ca_layer(sd, key + ".q_node", r, md, nonlinearity=True)
    return r + x


def q_residual_group(sd, key, x, md, n_resblocks, style, pa, q_layer, num_q_layers):
    """ref: attention_manipulators/architectures.py:215-233."""
    if MAP_STORAGE == "bf16" and CONV_PRECISION == "bf16" and style == "standard" and not pa and x.shape[1] == 64:
        def gate(i, t2):
            q_in = q_layer if (num_q_layers is None or i < num_q_layers) else False
            g = ca_gate(sd, f"{key}.body.{i}.final_body", t2)
            return g * para_ca_gate(sd, f"{key}.body.{i}.q_node", md, nonlinearity=True) if q_in else g
        return _stored_group(x, n_resblocks, lambda i: f"{key}.body.{i}", key + ".final_body", sd, gate)
    r = x
    for i in range(n_resblocks):
        q_in = q_layer if (num_q_layers is None or i < num_q_layers) else False
        r = qrcab(sd, f"{key}.body.{i}", r, md, style, pa, q_in)
    return conv(sd, key + ".final_body", r) + x


def param_res_block(sd, key, x, md, res_scale, q_layer_nonlinearity):
    """ref: attention_manipulators/architectures.py:348-356."""
    r = conv(sd, key + ".body.2", F.relu(conv(sd, key + ".body.0", x)), alpha=res_scale)
    return para_ca_layer(sd, key + ".attention_layer", r, md, q_layer_nonlinearity) + x


def lam_module(sd, key, x):
    """Layer attention.  x: (B,N,C,H,W) -> (B,N*C,H,W).  ref: advanced/HAN_blocks.py:16-37."""
    b, n, c, h, w = x.shape
    q = x.reshape(b, n, -1)
    energy = torch.bmm(q, q.transpose(1, 2))
    energy = energy.max(dim=-1, keepdim=True)[0] - energy
    att = torch.softmax(energy, dim=-1)
    out = torch.bmm(att, q).reshape(b, n, c, h, w)
    out = sd[key + ".gamma"] * out + x
    return out.reshape(b, n * c, h, w)


def csam_module(sd, key, x):
    """Channel-spatial attention.  ref: advanced/HAN_blocks.py:51-76 (Conv3d 1->1, k3, pad1)."""
    att = torch.sigmoid(F.conv3d(x.unsqueeze(1), sd[key + ".conv.weight"], sd[key + ".conv.bias"], padding=1))
    att = (sd[key + ".gamma"] * att).reshape(x.shape)
    return x * att + x


# ----------------------------------------------------------------------------- whole nets
def rcan(sd, x, n_resgroups=10, n_resblocks=20, scale=4):
    """ref: advanced/architectures.py:156-161."""
    h = conv(sd, "head.0", x)
    r = h
    for g in range(n_resgroups):
        r = residual_group(sd, f"body.{g}", r, n_resblocks)
    r = conv(sd, f"body.{n_resgroups}", r) + h
    return conv(sd, "tail.1", upsampler(sd, "tail.0", r, scale))


def edsr(sd, x, num_blocks=16, scale=4, res_scale=0.1):
    """ref: advanced/architectures.py:219-225."""
    h = conv(sd, "head.0", x)
    r = h
    for i in range(num_blocks):
        r = res_block(sd, f"body.{i}", r, res_scale)
    r = conv(sd, f"body.{num_blocks}", r) + h
    return conv(sd, "tail.1", upsampler(sd, "tail.0", r, scale))


def _han_tail(sd, h, layers, scale):
    """Shared HAN/QHAN epilogue.  ``layers`` oldest-first; the reference stacks newest-first
    (advanced/architectures.py:359-362) and feeds the *last* map to CSAM."""
    res1 = torch.stack(layers[::-1], dim=1)
    out2 = conv(sd, "last_conv", lam_module(sd, "la", res1))
    out1 = csam_module(sd, "csa", layers[-1])
    r = conv(sd, "last", torch.cat([out1, out2], 1)) + h
    return conv(sd, "tail.1", upsampler(sd, "tail.0", r, scale))


def han(sd, x, n_resgroups=10, n_resblocks=20, scale=4):
    """ref: advanced/architectures.py:352-377."""
    h = conv(sd, "head.0", x)
    r = h
    layers = []
    for g in range(n_resgroups):
        r = residual_group(sd, f"body.{g}", r, n_resblocks)
        layers.append(r)
    r = conv(sd, f"body.{n_resgroups}", r)
    layers.append(r)
    return _han_tail(sd, h, layers, scale)


def qrcan(sd, x, md, n_resgroups=10, n_resblocks=20, scale=4, style="modulate", include_pixel_attention=False,
          include_q_layer=False, selective_meta_blocks=None, num_q_layers_inner_residual=None):
    """ref: attention_manipulators/architectures.py:285-316."""
    h = conv(sd, "head.0", x)
    r = h
    for g in range(n_resgroups):
        q = include_q_layer if (selective_meta_blocks is None or selective_meta_blocks[g]) else False
        r = q_residual_group(sd, f"body.{g}", r, md, n_resblocks, style, include_pixel_attention, q,
                             num_q_layers_inner_residual)
    r = conv(sd, "final_body", r) + h
    return conv(sd, "tail.1", upsampler(sd, "tail.0", r, scale))


def qedsr(sd, x, md, num_blocks=16, scale=4, res_scale=0.1, q_layer_nonlinearity=False):
    """ref: attention_manipulators/architectures.py:392-399 (head is a bare conv: key 'head')."""
    h = conv(sd, "head", x)
    r = h
    for i in range(num_blocks):
        r = param_res_block(sd, f"body.{i}", r, md, res_scale, q_layer_nonlinearity)
    r = conv(sd, "final_body", r) + h
    return conv(sd, "tail.1", upsampler(sd, "tail.0", r, scale))


def qhan(sd, x, md, n_resgroups=10, n_resblocks=20, scale=4, num_q_layers_inner_residual=None):
    """ref: attention_manipulators/architectures.py:512-540 (style 'standard', q-layer on, no PA)."""
    h = conv(sd, "head.0", x)
    r = h
    layers = []
    for g in range(n_resgroups):
        r = q_residual_group(sd, f"body.{g}", r, md, n_resblocks, "standard", False, True,
                             num_q_layers_inner_residual)
        layers.append(r)
    r = conv(sd, f"body.{n_resgroups}", r)
    layers.append(r)
    return _han_tail(sd, h, layers, scale)


# ----------------------------------------------------------------------------- SAN: second-order + non-local attention
class _CovPool(torch.autograd.Function):
    """Covariance pooling  X I^ X^T  with I^ = I/M - 11^T/M^2  (ref: advanced/mpncov.py:12-47).
    The reference materialises the M x M matrix I^; X I^ is simply (X - rowmean(X)) / M, which is what is
    computed here (identical algebra, no M^2 memory).  Backward is the reference's: (G + G^T) X I^."""

    @staticmethod
    def forward(ctx, x):
        b, c, h, w = x.shape
        rows = x.reshape(b, c, h * w)
        centred = (rows - rows.mean(dim=2, keepdim=True)) / (h * w)
        ctx.save_for_backward(centred)
        ctx.shape = x.shape
        return centred.bmm(rows.transpose(1, 2))

    @staticmethod
    def backward(ctx, g):
        (centred,) = ctx.saved_tensors
        return (g + g.transpose(1, 2)).bmm(centred).reshape(ctx.shape)


class _SqrtmNS(torch.autograd.Function):
    """Matrix square root by coupled Newton-Schulz iteration with trace pre-normalisation and sqrt(trace)
    post-compensation; hand-derived backward (ref: advanced/mpncov.py:49-112).  iterN >= 2 only (SOCA uses 5)."""

    @staticmethod
    def forward(ctx, a, n_iter):
        b, d, _ = a.shape
        eye3 = 3.0 * torch.eye(d, dtype=a.dtype).expand(b, d, d)
        tr = a.diagonal(dim1=1, dim2=2).sum(dim=1)
        an = a / tr.view(b, 1, 1)
        zy = 0.5 * (eye3 - an)
        ys, zs = [an.bmm(zy)], [zy]
        for _ in range(1, n_iter - 1):
            zy = 0.5 * (eye3 - zs[-1].bmm(ys[-1]))
            ys.append(ys[-1].bmm(zy))
            zs.append(zy.bmm(zs[-1]))
        last = 0.5 * ys[-1].bmm(eye3 - zs[-1].bmm(ys[-1]))
        ctx.save_for_backward(a, an, last, tr, torch.stack(ys, 1), torch.stack(zs, 1))
        return last * tr.sqrt().view(b, 1, 1)

    @staticmethod
    def backward(ctx, g):
        a, an, last, tr, ys, zs = ctx.saved_tensors
        b, d, _ = a.shape
        eye3 = 3.0 * torch.eye(d, dtype=a.dtype).expand(b, d, d)
        rt = tr.sqrt()
        gp = g * rt.view(b, 1, 1)                                   # through the post-compensation
        aux = (g * last).sum(dim=(1, 2)) / (2.0 * rt)               # d/d(trace) of sqrt(trace)
        k = ys.shape[1] - 1
        yk, zk = ys[:, k], zs[:, k]
        dy = 0.5 * (gp.bmm(eye3 - yk.bmm(zk)) - zk.bmm(yk).bmm(gp))
        dz = -0.5 * yk.bmm(gp).bmm(yk)
        for i in range(k - 1, -1, -1):
            yi, zi = ys[:, i], zs[:, i]
            yz = eye3 - yi.bmm(zi)
            zyi = zi.bmm(yi)
            dy, dz = (0.5 * (dy.bmm(yz) - zi.bmm(dz).bmm(zi) - zyi.bmm(dy)),
                      0.5 * (yz.bmm(dz) - yi.bmm(dy).bmm(yi) - dz.bmm(zyi)))
        dn = 0.5 * (dy.bmm(eye3 - an) - dz - an.bmm(dy))
        ga = dn / tr.view(b, 1, 1)
        diag = aux - (dn * a).sum(dim=(1, 2)) / (tr * tr)           # trace normalisation + post-compensation
        return ga + diag.view(b, 1, 1) * torch.eye(d, dtype=a.dtype), None


def cov_sqrt(x, n_iter=5):
    """(B,C,H,W) -> (B,C,C): sqrtm(covpool(x)).  ref: advanced/SAN_blocks.py:290-291."""
    return _SqrtmNS.apply(_CovPool.apply(x), n_iter)


def _soca_window(x, limit=1000):
    """Centre crop applied before pooling when a side exceeds 1000 (ref: advanced/SAN_blocks.py:264-280); the
    branch structure (strict comparisons, Python slice semantics for a negative start) is kept as is."""
    h, w = x.shape[2:]
    if h < limit and w < limit:
        return x
    if h < limit and w > limit:
        w0 = (w - limit) // 2
        return x[:, :, :, w0:w0 + limit]
    if w < limit and h > limit:
        h0 = (h - limit) // 2
        return x[:, :, h0:h0 + limit, :]
    h0, w0 = (h - limit) // 2, (w - limit) // 2
    return x[:, :, h0:h0 + limit, w0:w0 + limit]


def soca(sd, key, x):
    """Second-order channel attention.  ref: advanced/SAN_blocks.py:244-302: gate = conv_du(mean over rows of
    sqrtm(cov(x))), output gate * x."""
    b, c = x.shape[:2]
    pooled = cov_sqrt(_soca_window(x)).mean(dim=1).view(b, c, 1, 1)
    gate = torch.sigmoid(_fc(sd, key + ".conv_du.2", F.relu(_fc(sd, key + ".conv_du.0", pooled))))
    return gate * x


def nonlocal_block(sd, key, x):
    """Embedded-Gaussian non-local block (ref: advanced/SAN_blocks.py:104-148):
    z = W(softmax(theta(x)^T phi(x)) g(x)) + x with 1x1 projections to C/8 channels.
    In the 2-D block the constructor's local `sub_sample` is rebound to the nn.Upsample class (:40), which is
    truthy, so phi and g are ALWAYS followed by MaxPool2d(2) (:88-93, state_dict keys `phi.0`, `g.0`) although
    Nonlocal_CA passes sub_sample=False: queries are all positions, keys/values the 2x2-max-pooled ones."""
    b, _, h, w = x.shape
    ci = sd[key + ".g.0.weight"].shape[0]
    g = F.max_pool2d(F.conv2d(x, sd[key + ".g.0.weight"], sd[key + ".g.0.bias"]), 2).reshape(b, ci, -1).transpose(1, 2)
    th = F.conv2d(x, sd[key + ".theta.weight"], sd[key + ".theta.bias"]).reshape(b, ci, h * w).transpose(1, 2)
    ph = F.max_pool2d(F.conv2d(x, sd[key + ".phi.0.weight"], sd[key + ".phi.0.bias"]), 2).reshape(b, ci, -1)
    att = torch.softmax(th.bmm(ph), dim=-1)
    y = att.bmm(g).transpose(1, 2).reshape(b, ci, h, w)
    return F.conv2d(y, sd[key + ".W.weight"], sd[key + ".W.bias"]) + x


def nonlocal_ca(sd, key, x):
    """Non-local attention applied independently to the four quadrants split at (H//2, W//2).
    ref: advanced/SAN_blocks.py:314-336.  The module's `soca` child is never called."""
    h1, w1 = x.shape[2] // 2, x.shape[3] // 2
    rows = []
    for hs in (slice(0, h1), slice(h1, None)):
        rows.append(torch.cat([nonlocal_block(sd, key + ".non_local", x[:, :, hs, ws])
                               for ws in (slice(0, w1), slice(w1, None))], dim=3))
    return torch.cat(rows, dim=2)


def rb(sd, key, x):
    """ref: advanced/SAN_blocks.py:359-363: conv(relu(conv(x))) + x."""
    return conv(sd, key + ".conv_first.2", F.relu(conv(sd, key + ".conv_first.0", x))) + x


def lsrag(sd, key, x, n_resblocks):
    """ref: advanced/SAN_blocks.py:394-412: RB^n -> SOCA -> conv_last, + x (the group's gamma is unused)."""
    r = x
    for i in range(n_resblocks):
        r = rb(sd, f"{key}.rcab.{i}", r)
    return conv(sd, key + ".conv_last", soca(sd, key + ".soca", r)) + x


def qrb(sd, key, x, md):
    """ref: attention_manipulators/qsan_blocks.py:29-34: meta gate (ReLU variant) between conv pair and skip."""
    r = conv(sd, key + ".conv_first.2", F.relu(conv(sd, key + ".conv_first.0", x)))
    return para_ca_layer(sd, key + ".q_layer", r, md, True) + x


def qlsrag(sd, key, x, md, n_resblocks):
    """ref: attention_manipulators/qsan_blocks.py:66-85."""
    r = x
    for i in range(n_resblocks):
        r = qrb(sd, f"{key}.rcab.{i}", r, md)
    return conv(sd, key + ".conv_last", soca(sd, key + ".soca", r)) + x


def san(sd, x, n_resgroups=20, n_resblocks=10, scale=4):
    """ref: advanced/architectures.py:291-312.  One Nonlocal_CA is applied twice (shared weights); every group
    adds gamma * (the first non-local output); SAN.conv_last is never used."""
    h = conv(sd, "head.0", x)
    shared = nonlocal_ca(sd, "non_local", h)
    r = shared
    for g in range(n_resgroups):
        r = lsrag(sd, f"RG.{g}", r, n_resblocks) + sd["gamma"] * shared
    r = nonlocal_ca(sd, "non_local", r) + h
    return conv(sd, "tail.1", upsampler(sd, "tail.0", r, scale))


def qsan(sd, x, md, n_resgroups=20, n_resblocks=10, scale=4):
    """ref: attention_manipulators/architectures.py:449-467."""
    h = conv(sd, "head.0", x)
    shared = nonlocal_ca(sd, "non_local", h)
    r = shared
    for g in range(n_resgroups):
        r = qlsrag(sd, f"RG.{g}", r, md, n_resblocks) + sd["gamma"] * shared
    r = nonlocal_ca(sd, "non_local", r) + h
    return conv(sd, "tail.1", upsampler(sd, "tail.0", r, scale))


def chop_forward(run, x, scale, max_pixels=160000, shave=10):
    """Overlapping 4-way tiling used by the SAN / QSAN handlers at evaluation time
    (ref: advanced/handlers.py:80-118, attention_manipulators/handlers.py:99-137): corners of size
    (h//2 + shave, w//2 + shave), recursion while a corner holds >= max_pixels, seams cut at the half points."""
    b, c, h, w = x.shape
    hh, wh = h // 2, w // 2
    hs, ws = hh + shave, wh + shave
    corners = [x[:, :, :hs, :ws], x[:, :, :hs, w - ws:], x[:, :, h - hs:, :ws], x[:, :, h - hs:, w - ws:]]
    if hs * ws < max_pixels:
        sr = [run(t) for t in corners]
    else:
        sr = [chop_forward(run, t, scale, max_pixels, shave) for t in corners]
    H, W, hh, wh, hs, ws = scale * h, scale * w, scale * hh, scale * wh, scale * hs, scale * ws
    out = x.new_empty(b, sr[0].shape[1], H, W)
    out[:, :, :hh, :wh] = sr[0][:, :, :hh, :wh]
    out[:, :, :hh, wh:] = sr[1][:, :, :hh, ws - W + wh:ws]
    out[:, :, hh:, :wh] = sr[2][:, :, hs - H + hh:hs, :wh]
    out[:, :, hh:, wh:] = sr[3][:, :, hs - H + hh:hs, ws - W + wh:ws]
    return out


def srmd(sd, x, nb=12, scale=4, act_mode="R", upsample_mode="pixelshuffle", training=True):
    """ref: advanced/architectures.py:380-425 with advanced/SRMD_blocks.py:33-142: x is the (3+M)-channel input (RGB + metadata
    maps).  The model is a flat Sequential with one module per character of the mode strings: head 'C' + act_mode[-1], body
    ('C' + act_mode) x (nb - 2), tail 'C' + str(scale) (conv + PixelShuffle), Upsample(nearest) + conv ('upconv') or one ConvTranspose2d ('convtranspose').  'B' is
    BatchNorm2d(momentum=0.9, eps=1e-4): batch statistics when `training` (running statistics in `sd` updated in place)."""
    tail = {"pixelshuffle": "C" + str(scale), "upconv": {2: "UC", 3: "uC", 4: "vC"}[scale], "convtranspose": "T"}[upsample_mode]
    y = x
    for i, t in enumerate("C" + act_mode[-1] + ("C" + act_mode) * (nb - 2) + tail):
        key = f"model.{i}"
        if t == "C":
            y = conv(sd, key, y)
        elif t == "B":
            y = F.batch_norm(y, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"], sd[key + ".bias"],
                             training=training, momentum=0.9, eps=1e-4)
        elif t == "I":  # nn.InstanceNorm2d(affine=True): per-sample statistics in train() and eval() alike
            y = F.instance_norm(y, weight=sd[key + ".weight"], bias=sd[key + ".bias"], eps=1e-5)
        elif t == "T":  # ConvTranspose2d(kernel = stride = scale, padding 0)
            y = F.conv_transpose2d(y, sd[key + ".weight"], sd[key + ".bias"], stride=scale)
        elif t in "Rr":
            y = F.relu(y)
        elif t in "Ll":
            y = F.leaky_relu(y, 0.2)
        elif t in "234":
            y = F.pixel_shuffle(y, int(t))
        elif t in "Uuv":
            y = F.interpolate(y, scale_factor={"U": 2, "u": 3, "v": 4}[t], mode="nearest")
        else:
            raise NotImplementedError(t)
    return y


def sft_channels(x, metadata):
    """(B, M) metadata -> (B, M, H, W) maps.  ref: attention_manipulators/__init__.py:53-80."""
    B, _, H, W = x.shape
    return metadata.reshape(B, -1, 1, 1).to(torch.float32).expand(B, metadata.shape[1], H, W)


NETS = {"srmd": srmd, "rcan": rcan, "edsr": edsr, "han": han, "san": san, "qrcan": qrcan, "qedsr": qedsr, "qhan": qhan, "qsan": qsan}
META_NETS = ("qrcan", "qedsr", "qhan", "qsan")


def forward(name, sd, x, metadata=None, **cfg):
    """Dispatch by registry name (ref: SISR/models/__init__.py:26-30 lower-cased handler names)."""
    if name in META_NETS:
        return NETS[name](sd, x, metadata, **cfg)
    return NETS[name](sd, x, **cfg)


# ----------------------------------------------------------------------------- host-side metadata logic
def num_metadata(metadata_list):
    """ref: attention_manipulators/__init__.py:13-24."""
    if metadata_list is None:
        return 1
    n = len(metadata_list)
    if "all" in metadata_list:
        n += 39
    if "blur_kernel" in metadata_list:
        n += 9
    elif "unmodified_blur_kernel" in metadata_list:
        n += 440
    return n


def generate_channels(batch_size, metadata, keys, metadata_list, n_meta):
    """(B,M) collated metadata (+ key tuples) -> fp32 (B,n_meta,1,1).  ref: attention_manipulators/__init__.py:30-51.
    ``keys`` is default_collate's list[M] of B-tuples; the mask is built from sample 0's key."""
    if metadata is None:
        raise RuntimeError("Metadata needs to be specified for this network to run properly.")
    md = torch.as_tensor(np.asarray(metadata))
    if "all" in metadata_list:
        mask = torch.ones(n_meta, dtype=torch.bool)
    else:
        mask = torch.tensor([k[0] in metadata_list for k in keys], dtype=torch.bool)
    out = torch.ones(batch_size, n_meta)
    for i in range(batch_size):
        row = md[i] if len(keys) == 1 else md[i][mask]
        out[i] = out[i] * row  # float64 row promoted then stored into the fp32 buffer
    return out[:, :, None, None]


def scale_qpi(qpi, n_feats=64, min_mu=-0.2, max_mu=0.8, sig=0.2, clamp=False):
    """'modulate' style: scalar -> n_feats-bin Gaussian.  ref: attention_manipulators/handlers.py:38-54."""
    mu = qpi * (max_mu - min_mu) + min_mu
    base = np.linspace(0, 1, n_feats)
    rows = []
    for i in range(mu.shape[0]):
        m = mu[i].squeeze().numpy()
        rows.append(torch.from_numpy((1 / (np.sqrt(2 * np.pi) * sig)) * np.exp(-np.power(base - m, 2.0) /
                                                                              (2 * np.power(sig, 2.0)))).float())
    full = torch.stack(rows)
    if clamp:
        full = torch.clamp(full, 0, 1)
    return full[:, :, None, None]


# ----------------------------------------------------------------------------- metric
def rgb_to_y(img):
    """BT.601 'jpg' luma, no offset.  ref: sr_tools/image_manipulation.py:65-70."""
    return 0.299 * img[0] + 0.587 * img[1] + 0.114 * img[2]


def psnr(a, b, max_value=1.0):
    """ref: sr_tools/metrics.py:6-17 (float32 mse, 100 when identical)."""
    mse = np.mean((np.array(a, dtype=np.float32) - np.array(b, dtype=np.float32)) ** 2)
    if mse == 0:
        return 100
    return 20 * np.log10(max_value / np.sqrt(mse))


def y_psnr(sr, hr):
    """Eval protocol: clip to [0,1], Y of each, PSNR(max=1).  ref: SISR/models/__init__.py:158-169,
    training_handler.py:179-224 / standard_eval.py:262-294."""
    sr = np.clip(np.asarray(sr), 0, 1)
    return psnr(rgb_to_y(sr), rgb_to_y(np.asarray(hr)), max_value=1)


# ----------------------------------------------------------------------------- train step
class Trainer:
    """Functional mirror of BaseModel.run_train/standard_update (ref: SISR/models/__init__.py:466-489,
    :299-335): L1Loss(mean) -> zero_grad -> backward -> [clip_grad_norm_] -> Adam.step -> scheduler.step
    (scheduler stepped per *batch*)."""

    def __init__(self, name, state_dict, lr=1e-4, scheduler=None, scheduler_params=None, grad_clip=None,
                 betas=(0.9, 0.999), **cfg):
        self.name, self.cfg = name, cfg
        # parameters are optimised; batch-norm buffers (SPARNet) ride along and are updated in place by the forward
        buf = lambda k: "running_" in k or "num_batches_tracked" in k  # noqa: E731
        self.sd = {k: (v.detach().clone() if buf(k) else v.detach().clone().float().requires_grad_(True))
                   for k, v in state_dict.items()}
        self.opt = torch.optim.Adam([v for v in self.sd.values() if v.requires_grad], lr=lr, betas=betas)
        self.sched = None
        if scheduler == "cosine_annealing_warm_restarts":
            self.sched = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(
                self.opt, T_0=scheduler_params["restart_period"], T_mult=scheduler_params["t_mult"],
                eta_min=scheduler_params["lr_min"])
        elif scheduler == "multi_step_lr":
            self.sched = torch.optim.lr_scheduler.MultiStepLR(
                self.opt, milestones=scheduler_params["milestones"], gamma=scheduler_params["gamma"])
        elif scheduler == "step_lr":
            self.sched = torch.optim.lr_scheduler.StepLR(
                self.opt, step_size=scheduler_params["step_size"], gamma=scheduler_params["gamma"])
        elif scheduler is not None:
            raise RuntimeError("%s scheduler not implemented" % scheduler)
        self.grad_clip = grad_clip or None

    def step(self, x, y, metadata=None):
        out = forward(self.name, self.sd, x, metadata, **self.cfg)
        loss = F.l1_loss(out, y)
        self.opt.zero_grad()
        loss.backward()
        if self.grad_clip is not None:
            torch.nn.utils.clip_grad_norm_([v for v in self.sd.values() if v.requires_grad], self.grad_clip)
        gnorm = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in self.sd.values()
                               if p.grad is not None)).item()  # post-clip; SAN holds parameters its forward never uses
        self.opt.step()
        if self.sched is not None:
            self.sched.step()
        return loss.item(), out.detach(), gnorm

    @property
    def lr(self):
        return self.opt.param_groups[0]["lr"]


# ----------------------------------------------------------------------------- SFTMD (second half)
def _sft_standard(sd, key, feat, para, mask_para=False, repeats=None):
    """StandardSft.  ref: SFTMD_variants/architectures.py:25-56."""
    if repeats is not None:
        para = para.repeat(1, repeats, 1, 1)
    cat = feat if mask_para else torch.cat((feat, para), dim=1)
    mul = torch.sigmoid(conv(sd, key + ".mul_conv2", F.leaky_relu(conv(sd, key + ".mul_conv1", cat), 0.2)))
    add = conv(sd, key + ".add_conv2", F.leaky_relu(conv(sd, key + ".add_conv1", cat), 0.2))
    return feat * mul + add


def _sft_layer(sd, key, feat, para, sft_type, mask_para, repeats):
    """SFT_Layer.  ref: SFTMD_variants/architectures.py:8-22 (ConcatSft, WeakSft), :59-77."""
    if sft_type == "standard":
        return _sft_standard(sd, key + ".sft_module", feat, para, mask_para, repeats)
    if sft_type == "concat":
        return conv(sd, key + ".sft_module.conv", torch.cat((feat, para), dim=1))
    if sft_type == "weak":
        return feat * para
    if sft_type == "none":
        return feat
    raise ValueError(sft_type)


def sftmd(sd, x, para_maps, num_blocks=16, scale=4, sft_type="standard", mask_para=False, repeats=None, q_injection=False,
          q_layers=2):
    """ref: SFTMD_variants/architectures.py:110-176; para_maps (B, M, H, W), or (B, M, 1, 1) vectors with q_injection
    (SFTMD_variants/handlers.py:19-22, :35-41)."""
    lr = lambda t: F.leaky_relu(t, 0.2)  # noqa: E731
    sft = lambda key, t: _sft_layer(sd, key, t, para_maps, sft_type, mask_para, repeats)  # noqa: E731
    inject = (lambda key, t: para_ca_layer(sd, key, t, para_maps, True, q_layers)) if q_injection else (lambda key, t: t)
    bef = conv(sd, "conv3", lr(conv(sd, "conv2", lr(conv(sd, "conv1", x)))))
    fea = bef
    for i in range(num_blocks):
        k = f"SFT-residual{i + 1}"
        f1 = inject(k + ".q_1", F.relu(sft(k + ".sft1", fea)))
        f2 = inject(k + ".q_2", F.relu(sft(k + ".sft2", conv(sd, k + ".conv1", f1))))
        fea = fea + conv(sd, k + ".conv2", f2)
    fin = inject("final_injection", sft("sft", fea + bef))
    up = conv(sd, "conv_mid", fin)
    if scale == 4:
        up = lr(F.pixel_shuffle(conv(sd, "upscale.0", up), 2))
        up = lr(F.pixel_shuffle(conv(sd, "upscale.3", up), 2))
    else:
        up = lr(F.pixel_shuffle(conv(sd, "upscale.0", up), scale))
    return torch.clamp(conv(sd, "conv_output", up), min=0.0, max=1.0)


NETS["sftmd"] = sftmd
META_NETS = META_NETS + ("sftmd",)


# ----------------------------------------------------------------------------- SPARNet / QSPARNet
# ref: SISR/models/SPARNet/blocks.py:10-243, SPARNet/architectures.py:7-155.  Batch norm: `training` selects batch
# statistics (biased variance; the running statistics in `sd` are updated in place with momentum 0.1 and the unbiased
# variance, as nn.BatchNorm2d does) or the running statistics.
def _sp_norm(sd, key, x, norm_type, training):
    """ref: blocks.py:10-36 NormLayer (key = the NormLayer's prefix; its module is `.norm`)"""
    if norm_type == "bn":
        return _sp_bn(sd, key + ".norm", x, training)
    if norm_type == "in":
        return F.instance_norm(x, weight=sd[key + ".norm.weight"], bias=sd[key + ".norm.bias"], eps=1e-5)
    if norm_type == "gn":
        return F.group_norm(x, 32, sd[key + ".norm.weight"], sd[key + ".norm.bias"], 1e-5)
    if norm_type == "pixel":
        return F.normalize(x, p=2, dim=1)
    if norm_type == "none":
        return x
    raise ValueError(norm_type)


def _sp_act(sd, key, x, relu_type):
    """ref: blocks.py:39-66 ReluLayer (key = the ReluLayer's prefix; PReLU's slopes are `.func.weight`)"""
    if relu_type == "relu":
        return F.relu(x)
    if relu_type == "leakyrelu":
        return F.leaky_relu(x, 0.2)
    if relu_type == "prelu":
        return F.prelu(x, sd[key + ".func.weight"])
    if relu_type == "selu":
        return F.selu(x)
    if relu_type == "none":
        return x
    raise ValueError(relu_type)


def _sp_conv_layer(sd, key, x, scale="none", norm="none", relu="none", training=True):
    """ref: blocks.py:69-103 ConvLayer.forward: [nearest x2] -> ReflectionPad2d(1) -> Conv2d(3x3, stride) -> norm -> relu."""
    if scale == "up":
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    x = F.pad(x, (1, 1, 1, 1), mode="reflect")
    x = F.conv2d(x, sd[key + ".conv2d.weight"], sd.get(key + ".conv2d.bias"), stride=2 if scale == "down" else 1)
    return _sp_act(sd, key + ".relu", _sp_norm(sd, key + ".norm", x, norm, training), relu)


def _sp_bn(sd, key, x, training):
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"], sd[key + ".bias"],
                        training=training, momentum=0.1, eps=1e-5)


def _sp_hourglass(sd, key, x, depth, norm, relu, training):
    """ref: blocks.py:177-243 (the attention map has one channel, 'spar', or one per feature, 'spar3d': the product broadcasts)"""
    if depth == 0:
        return x

    def level(lv, t):
        up1 = _sp_conv_layer(sd, f"{key}.b1_{lv}", t, "none", norm, relu, training)
        low1 = _sp_conv_layer(sd, f"{key}.b2_{lv}", t, "down", norm, relu, training)
        low2 = level(lv - 1, low1) if lv > 1 else _sp_conv_layer(sd, f"{key}.b2_plus_{lv}", low1, "none", norm, relu, training)
        up2 = _sp_conv_layer(sd, f"{key}.b3_{lv}", low2, "up", norm, relu, training)
        if up1.shape[2:] != up2.shape[2:]:
            up2 = F.interpolate(up2, up1.shape[2:])
        return up1 + up2

    att = torch.sigmoid(_sp_conv_layer(sd, key + ".out_block.0", level(depth, x)))
    return x * att


def _sp_block(sd, key, x, scale, depth, norm, relu, training, md=None):
    """ref: blocks.py:106-174 ResidualBlock.forward (metadata_attention when its parameters are present)"""
    identity = _sp_conv_layer(sd, key + ".shortcut_func", x, scale) if (key + ".shortcut_func.conv2d.weight") in sd else x
    out = _sp_act(sd, key + ".preact_func.1", _sp_norm(sd, key + ".preact_func.0", x, norm, training), relu)
    s1, s2 = {"down": ("none", "down"), "up": ("up", "none"), "none": ("none", "none")}[scale]
    out = _sp_conv_layer(sd, key + ".conv1", out, s1, norm, relu, training)
    out = _sp_conv_layer(sd, key + ".conv2", out, s2, norm, "none", training)
    out = identity + _sp_hourglass(sd, key + ".att_func", out, depth, norm, relu, training)
    if md is not None and (key + ".metadata_attention.attribute_integrator.0.weight") in sd:
        out = para_ca_layer(sd, key + ".metadata_attention", out, md, True)
    return out


def sparnet(sd, x, md=None, in_size=128, out_size=128, min_feat_size=16, res_depth=10, bottleneck_size=4, slope=0.2,
            training=False, norm_type="bn", relu_type=None):
    """ref: architectures.py:7-76 (SPARNet) / :79-155 (QSPARNet: md = (B, M, 1, 1) metadata).  Channel counts come from the
    state dict; the layer plan (which blocks scale, the hourglass depths) from the size arguments as in the constructors.
    relu_type None: LeakyReLU(0.2) / ReLU / none from `slope` (0.2 / 0 / 1), the round-3 signature."""
    if relu_type is None:
        relu_type = {0.2: "leakyrelu", 0.0: "relu", 1.0: "none"}[float(slope)]
    down_steps = int(np.log2(in_size // min_feat_size))
    up_steps = int(np.log2(out_size // min_feat_size))
    hg = int(np.log2(64 / bottleneck_size))
    out = _sp_conv_layer(sd, "encoder.0", x)
    for i in range(down_steps):
        out = _sp_block(sd, f"encoder.{i + 1}", out, "down", hg, norm_type, relu_type, training, md)
        hg -= 1
    hg += 1
    for i in range(res_depth + 3 - down_steps):
        out = _sp_block(sd, f"res_layers.{i}", out, "none", hg, norm_type, relu_type, training, md)
    for i in range(up_steps):
        hg += 1
        out = _sp_block(sd, f"decoder.{i}", out, "up", hg, norm_type, relu_type, training, md)
    return _sp_conv_layer(sd, "out_conv", out)


def qsparnet(sd, x, md, **cfg):
    return sparnet(sd, x, md, **cfg)


NETS["sparnet"] = sparnet
NETS["qsparnet"] = qsparnet
META_NETS = META_NETS + ("qsparnet",)
