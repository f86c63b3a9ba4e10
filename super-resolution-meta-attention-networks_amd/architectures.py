"""Drop-in nn.Module classes for the in-scope SISR networks, executing on the HIP kernels.

Each class keeps the reference's constructor signature, sub-module names, parameter shapes
(OIHW fp32) and construction ORDER, so that (a) ``state_dict()`` keys / checkpoints are
interchangeable with the reference and (b) the same seed yields the same initial weights
(tests/test_init_parity.py pins both against the reference's own output).  The nn.Conv2d
children are parameter holders only: ``forward`` never calls them, it hands their tensors to
the fused operators in ops.py.  Feature maps flow as channels_last (NHWC) tensors end to end;
only the 3-channel input/output are NCHW.

ref files: Code/SISR/models/advanced/{common,architectures}.py,
           Code/SISR/models/attention_manipulators/{q_layer,architectures}.py
"""
import math

import torch
from torch import nn

from . import ops


def default_conv(in_channels, out_channels, kernel_size, bias=True):
    """ref: advanced/common.py:5-8"""
    return nn.Conv2d(in_channels, out_channels, kernel_size, padding=(kernel_size // 2), bias=bias)


def _conv(holder, x, **kw):
    return ops.conv3x3(x, holder.weight, holder.bias, **kw)


def _ca_params(seq):
    """(w1, b1, w2, b2) of a conv_du = [Conv1x1, ReLU, Conv1x1, Sigmoid] stack."""
    return seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias


def conv_weights(net):
    """[(weight, shuffle)] of every 3x3 conv the MFMA kernels run (channel counts multiples of 64), for ops.pack_all:
    shuffle = r for the Upsampler convs whose PixelShuffle is fused into the store, 1 otherwise."""
    out, seen = [], set()
    for m in net.modules():
        if isinstance(m, Upsampler):
            mods = list(m)
            for i in range(0, len(mods), 2):
                r, w = mods[i + 1].upscale_factor, mods[i].weight
                if w.shape[0] == 64 * r * r and w.shape[1] % 64 == 0 and id(w) not in seen:
                    out.append((w, r))
                    seen.add(id(w))
    any_width = getattr(net, "pack_padded_convs", False)  # SPARNet: maps zero-padded to 64-multiples, weights packed likewise
    for m in net.modules():
        if isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3) and m.groups == 1 and id(m.weight) not in seen:
            w = m.weight
            if (w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0) or any_width:
                out.append((w, 1))
                seen.add(id(w))
    return out


def _meta_sig(lay):
    a, b = (lay.attribute_integrator[i] for i in lay.fc_index)
    return (tuple(a.weight.shape), tuple(b.weight.shape), lay.nonlinearity, a.bias is not None, b.bias is not None)


def meta_gates_batched(layers):
    """True when meta_gates() runs these ParaCALayers as ONE launch (and their parameter gradients therefore all arrive from
    one launch at the very end of a backward pass): more than one layer, all two-layer stacks of one shape with biases."""
    if len(layers) < 2 or any(lay.num_layers != 2 for lay in layers):
        return False
    first = _meta_sig(layers[0])
    return first[3] and first[4] and all(_meta_sig(lay) == first for lay in layers)


def late_parameters(net):
    """The parameters whose gradients a backward pass of `net` delivers last, in one launch: those of the meta-attention
    layers the network hands to meta_gates() in one batch (a network says which through meta_gate_layers()).  Layers applied
    block by block (QSPARNet's per-block metadata_attention, the metadata-mixing QCALayer styles' FC stacks) deliver their
    gradients in the middle of backward and are NOT late."""
    layers = net.meta_gate_layers() if hasattr(net, "meta_gate_layers") else []
    if not meta_gates_batched(layers):
        return []
    return [p for lay in layers for p in lay.parameters()]


def meta_gates(layers, attributes):
    """Gates of a list of ParaCALayers: one batched launch when they are uniform (the normal case: every q-layer of a
    network has the same metadata / hidden / channel sizes), else one launch per layer."""
    if not layers:
        return []
    if meta_gates_batched(layers) and not attributes.requires_grad:
        params = []
        for lay in layers:
            a, b = (lay.attribute_integrator[i] for i in lay.fc_index)
            params.append((a.weight, a.bias, b.weight, b.bias))
        if all(t.is_contiguous() for lay in params for t in lay):
            return list(ops.meta_gate_many(attributes, params, layers[0].nonlinearity))
    return [lay.gate(attributes) for lay in layers]


# ----------------------------------------------------------------------------- plain blocks
class Upsampler(nn.Sequential):
    """ref: advanced/common.py:20-45.  conv(C -> r^2 C) + PixelShuffle(r); for n_feat = 64 the shuffle is fused
    into the conv's store (an address map) and the nn.PixelShuffle children only keep the module indices; wider
    nets (EDSR 256) run the conv to a plain channels-last map and shuffle it with one gather kernel (ops.pixel_shuffle)."""

    def __init__(self, conv, scale, n_feat, bn=False, act=False, bias=True):
        if bn or act:
            raise NotImplementedError("Upsampler with BatchNorm/activation is not used by any in-scope model")
        m = []
        if (scale & (scale - 1)) == 0:
            for _ in range(int(math.log(scale, 2))):
                m.append(conv(n_feat, 4 * n_feat, 3, bias))
                m.append(nn.PixelShuffle(2))
        elif scale == 3:
            m.append(conv(n_feat, 9 * n_feat, 3, bias))
            m.append(nn.PixelShuffle(3))
        else:
            raise NotImplementedError
        super().__init__(*m)

    def forward(self, x):
        mods = list(self)
        for i in range(0, len(mods), 2):
            r = mods[i + 1].upscale_factor
            if mods[i].weight.shape[0] == 64 * r * r:
                x = _conv(mods[i], x, shuffle=r)
            else:
                x = ops.pixel_shuffle(_conv(mods[i], x), r)
        return x


class ResBlock(nn.Module):
    """ref: advanced/common.py:48-72: x + res_scale * conv(relu(conv(x)))"""

    def __init__(self, conv, n_feats, kernel_size, bias=True, bn=False, act=None, res_scale=1.0):
        super().__init__()
        if bn:
            raise NotImplementedError("BatchNorm ResBlock is not used by any in-scope model")
        self.body = nn.Sequential(conv(n_feats, n_feats, kernel_size, bias=bias), nn.ReLU(True),
                                  conv(n_feats, n_feats, kernel_size, bias=bias))
        self.res_scale = res_scale

    def forward(self, x):
        b = self.body
        return ops.res_block(x, b[0].weight, b[0].bias, b[2].weight, b[2].bias, res_scale=self.res_scale)


class CALayer(nn.Module):
    """ref: advanced/architectures.py:13-32"""

    def __init__(self, channel, reduction=16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.conv_du = nn.Sequential(nn.Conv2d(channel, channel // reduction, 1, padding=0, bias=True),
                                     nn.ReLU(inplace=True),
                                     nn.Conv2d(channel // reduction, channel, 1, padding=0, bias=True), nn.Sigmoid())

    def forward(self, x):
        return ops.ca_layer(x, *_ca_params(self.conv_du))


class RCAB(nn.Module):
    """ref: advanced/architectures.py:48-71 (res_scale stored, never applied)"""

    def __init__(self, conv, n_feat, kernel_size, reduction, bias=True, bn=False, act=None, res_scale=1):
        super().__init__()
        if bn:
            raise NotImplementedError("BatchNorm RCAB is not used by any in-scope model")
        self.body = nn.Sequential(conv(n_feat, n_feat, kernel_size, bias=bias), nn.ReLU(True),
                                  conv(n_feat, n_feat, kernel_size, bias=bias), CALayer(n_feat, reduction))
        self.res_scale = res_scale

    def forward(self, x):
        b = self.body
        return ops.res_block(x, b[0].weight, b[0].bias, b[2].weight, b[2].bias, ca=_ca_params(b[3].conv_du))


class ResidualGroup(nn.Module):
    """ref: advanced/architectures.py:94-110"""

    def __init__(self, conv, n_feat, kernel_size, reduction, act, res_scale, n_resblocks):
        super().__init__()
        body = [RCAB(conv, n_feat, kernel_size, reduction, bias=True, bn=False, act=act, res_scale=res_scale)
                for _ in range(n_resblocks)]
        body.append(conv(n_feat, n_feat, kernel_size))
        self.body = nn.Sequential(*body)

    def forward(self, x):
        mods = list(self.body)
        if ops.fused_groups_enabled() and x.shape[1] == 64:
            blocks = [(b.body[0].weight, b.body[0].bias, b.body[2].weight, b.body[2].bias,
                       _ca_params(b.body[3].conv_du), None) for b in mods[:-1]]
            return ops.gated_group(x, blocks, mods[-1].weight, mods[-1].bias)
        res = x
        for blk in mods[:-1]:
            res = blk(res)
        return _conv(mods[-1], res, residual=x)


# ----------------------------------------------------------------------------- feature counts that are not multiples of 64
class ChannelPadded:
    """Mixin of the residual-conv networks (RCAN / QRCAN / EDSR / QEDSR / HAN / QHAN) for a feature count n that is not a
    multiple of 64 (the reference takes any width: ref attention_manipulators/handlers.py:24-29, advanced/architectures.py:126-161).

    The kernels work on 64-channel chunks, so such a network RUNS as its twin of width P = the next multiple of 64 with
    every parameter zero-padded -- exactly the same function: padded conv outputs are 0 (zero weights and bias), padded
    inputs meet zero weights, a padded hidden unit of a gate is ReLU(0) = 0, and a padded gate value (sigmoid(0)) only ever
    multiplies a zero map.  Parameters, state_dict() keys and shapes, the optimiser and checkpoints stay the reference's: the
    padding is a differentiable op per parameter (ops.pad_param, one launch each way) applied every forward, the twin is
    a parameter-less skeleton on the meta device called through torch.func.functional_call, and autograd hands the gradients
    back cropped.  Costs one small launch per parameter and pass -- these widths appear in no published configuration.

    Layout rules of the padded axes (a dimension of size k * n is padded per n-run, never as a suffix of the whole):
      * [n][k]  the Upsampler convs' output channels and biases (PixelShuffle reads channel c * r^2 + i * r + j);
      * [k][n]  channel concatenations on the input side (HAN's 11 n and 2 n convs);
      * anything else that differs from the twin (gate hidden sizes such as n // 16, n // 2) is padded as a suffix."""

    _twin = None

    def _init_padding(self, n_feats, build_twin, nk=None):
        """Call at the end of __init__: build_twin(P) -> the same architecture at width P.  nk(key, dim) -> True where a
        k * n axis is laid out [n][k] (default: dimension 0 of the Upsampler's parameters, 'tail.0.*')."""
        if n_feats % 64 == 0:
            return
        if nk is None:
            nk = lambda k, d: d == 0 and ".tail.0." in "." + k  # noqa: E731
        P = (n_feats + 63) // 64 * 64
        with torch.device("meta"):
            twin = build_twin(P)
        want = {k: tuple(v.shape) for k, v in twin.named_parameters()}
        have = {k: tuple(v.shape) for k, v in self.named_parameters()}
        if list(want) != list(have):
            raise RuntimeError("channel padding: the twin network's parameter names differ")
        plans = {}
        for k, real in have.items():
            cur, steps = list(real), []
            for d in range(len(real)):
                r, t = real[d], want[k][d]
                if r == t:
                    continue
                before = 1
                for q in cur[:d]:
                    before *= q
                after = 1
                for q in cur[d + 1:]:
                    after *= q
                if r % n_feats == 0 and t == r // n_feats * P:
                    kk = r // n_feats
                    if kk > 1 and nk(k, d):                           # Upsampler: [n][r^2]
                        steps.append((before, n_feats, kk * after, P))
                    else:                                             # plain axis, or a [k][n] concatenation
                        steps.append((before * kk, n_feats, after, P))
                elif t > r:
                    steps.append((before, r, after, t))
                else:
                    raise RuntimeError(f"channel padding: {k} {real} -> {want[k]}")
                cur[d] = t
            plans[k] = (steps, want[k])
        object.__setattr__(self, "_twin", twin)  # not a registered sub-module: no parameters, no state-dict entries
        self._pad_plans, self._pad_width = plans, (n_feats, P)

    def padded(self):
        return self._twin is not None

    def _run_padded(self, *args):
        shadow = {k: ops.pad_param(p, *self._pad_plans[k]) for k, p in self.named_parameters()}
        return torch.func.functional_call(self._twin, shadow, args)


def _pad_channels(t, n, P):
    """(B, n, 1, 1) per-channel metadata (QRCAN 'modulate': one value per feature channel) -> (B, P, 1, 1), zeros behind."""
    B = t.shape[0]
    return ops.pad_param(t.reshape(B, n), [(B, n, 1, P)], (B, P, 1, 1))


def _check_rgb(x, what):
    if not x.is_cuda:
        raise RuntimeError(f"{what}: this network only runs on a HIP device (no CPU fallback); got a CPU tensor")


class RCAN(ChannelPadded, nn.Module):
    """ref: advanced/architectures.py:126-161"""

    def __init__(self, n_resblocks=20, n_resgroups=10, n_feats=64, in_feats=3, out_feats=3, scale=4, reduction=16,
                 res_scale=1.0):
        super().__init__()
        act = nn.ReLU(True)
        head = [default_conv(in_feats, n_feats, 3)]
        body = [ResidualGroup(default_conv, n_feats, 3, reduction, act=act, res_scale=res_scale,
                              n_resblocks=n_resblocks) for _ in range(n_resgroups)]
        body.append(default_conv(n_feats, n_feats, 3))
        tail = [Upsampler(default_conv, scale, n_feats, act=False), default_conv(n_feats, out_feats, 3)]
        self.head = nn.Sequential(*head)
        self.body = nn.Sequential(*body)
        self.tail = nn.Sequential(*tail)
        self._init_padding(n_feats, lambda P: RCAN(n_resblocks, n_resgroups, P, in_feats, out_feats, scale, reduction, res_scale))

    def forward(self, x):
        _check_rgb(x, "RCAN")
        if self.padded():
            return self._run_padded(x)
        x = _conv(self.head[0], x)
        res = x
        mods = list(self.body)
        for g in mods[:-1]:
            res = g(res)
        res = _conv(mods[-1], res, residual=x)
        return _conv(self.tail[1], self.tail[0](res))


class EDSR(ChannelPadded, nn.Module):
    """ref: advanced/architectures.py:183-225"""

    def __init__(self, in_features=3, out_features=3, net_features=64, num_blocks=16, scale=4, res_scale=0.1):
        super().__init__()
        act = nn.ReLU(True)
        head = [default_conv(in_features, net_features, 3)]
        body = [ResBlock(default_conv, net_features, 3, act=act, res_scale=res_scale) for _ in range(num_blocks)]
        body.append(default_conv(net_features, net_features, 3))
        tail = [Upsampler(default_conv, scale, net_features), default_conv(net_features, out_features, 3)]
        self.head = nn.Sequential(*head)
        self.body = nn.Sequential(*body)
        self.tail = nn.Sequential(*tail)
        self._init_padding(net_features, lambda P: EDSR(in_features, out_features, P, num_blocks, scale, res_scale))

    def forward(self, x):
        _check_rgb(x, "EDSR")
        if self.padded():
            return self._run_padded(x)
        x = _conv(self.head[0], x)
        res = x
        mods = list(self.body)
        for blk in mods[:-1]:
            res = blk(res)
        res = _conv(mods[-1], res, residual=x)
        return _conv(self.tail[1], self.tail[0](res))


# ----------------------------------------------------------------------------- meta-attention blocks
class ParaCALayer(nn.Module):
    """Meta-attention.  ref: attention_manipulators/q_layer.py:4-43"""

    def __init__(self, network_channels, num_metadata, nonlinearity=False, num_layers=2):
        super().__init__()
        layers, multiplier, inputs = [], num_layers, [num_metadata]
        self.fc_index = []
        for i in range(num_layers):
            if num_metadata > 15:
                inputs.append((network_channels - num_metadata) // multiplier + num_metadata)
            else:
                inputs.append(network_channels // multiplier)
            self.fc_index.append(len(layers))
            layers.append(nn.Conv2d(inputs[i], inputs[i + 1], 1, padding=0, bias=True))
            if nonlinearity and multiplier != 1:
                layers.append(nn.ReLU(inplace=True))
            multiplier -= 1
        layers.append(nn.Sigmoid())
        self.attribute_integrator = nn.Sequential(*layers)
        self.nonlinearity = bool(nonlinearity)
        self.num_layers = num_layers
        if not 1 <= num_layers <= 4:
            raise NotImplementedError("meta-attention gate kernels: 1 to 4 FC layers (the reference's default is 2)")

    def gate(self, attributes):
        """(B,M,1,1) -> (B,C) sigmoid gate (the only part that depends on parameters).  Two layers (the reference's
        default and every published config): the dedicated meta-gate kernel; 1, 3 or 4: the generic gate MLP."""
        fcs = [self.attribute_integrator[i] for i in self.fc_index]
        if self.num_layers == 2:
            a, b = fcs
            return ops.meta_gate(attributes, a.weight, a.bias, b.weight, b.bias, self.nonlinearity)
        n = len(fcs)
        spec = ([(0, 0, 2 if k == n - 1 else int(self.nonlinearity)) for k in range(n)], 0)
        params = []
        for c in fcs:
            params += [c.weight, c.bias]
        B = attributes.shape[0]
        return ops._GateMlp.apply(attributes.detach(), attributes.detach(), None, spec, *params).reshape(B, -1)

    def forward(self, x, attributes):
        return ops.gate_mul(x, self.gate(attributes))


class PALayer(nn.Module):
    """ref: attention_manipulators/architectures.py:13-26 (per-pixel 64 -> 8 -> 1 sigmoid gate)."""

    def __init__(self, channel):
        super().__init__()
        self.pa = nn.Sequential(nn.Conv2d(channel, channel // 8, 1, padding=0, bias=True), nn.ReLU(inplace=True),
                                nn.Conv2d(channel // 8, 1, 1, padding=0, bias=True), nn.Sigmoid())

    def forward(self, x):
        return ops.pa_layer(x, self.pa[0].weight, self.pa[0].bias, self.pa[2].weight, self.pa[2].bias)


class QCALayer(nn.Module):
    """ref: attention_manipulators/architectures.py:34-127.  'standard' (the paper's configuration) runs the
    fused CA kernels; the five metadata-mixing styles run the HIP global-average-pool, their whole FC stack as one
    generic gate-MLP launch (ops.qca_gate) and the HIP gate multiply.  The nn.Conv2d / Softmax children only hold
    parameters and keep the reference's state-dict keys."""

    def __init__(self, channel, style, reduction=16, num_metadata=1):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        if reduction < 16:
            raise RuntimeError('Using an extreme channel attention reduction value')
        channel_in = channel if style in ('modulate', 'mini_concat', 'standard') else channel + num_metadata
        cr = channel // reduction
        if style in ('modulate', 'max_concat', 'softmax', 'standard'):
            self.conv_du = nn.Sequential(nn.Conv2d(channel_in, cr, 1, padding=0, bias=True), nn.ReLU(inplace=True),
                                         nn.Conv2d(cr, channel, 1, padding=0, bias=True), nn.Sigmoid())
        elif style == 'mini_concat':
            self.pre_concat = nn.Conv2d(channel_in, cr, 1, padding=0, bias=True)
            self.conv_du = nn.Sequential(nn.ReLU(inplace=True),
                                         nn.Conv2d(cr + num_metadata, channel, 1, padding=0, bias=True), nn.Sigmoid())
        elif style == 'extended_attention':
            fractions = [(channel_in, channel // 2), (channel // 2 + num_metadata, channel // 4),
                         (channel // 4 + num_metadata, cr)]
            self.feature_convs = nn.ModuleList()
            for inp, outp in fractions:
                self.feature_convs.append(nn.Sequential(nn.Conv2d(inp, outp, 1, padding=0, bias=True),
                                                        nn.ReLU(inplace=True)))
            self.final_conv = nn.Sequential(nn.Conv2d(cr, channel, 1, padding=0, bias=True), nn.Sigmoid())
        if style == 'softmax':
            self.softmax = nn.Softmax(dim=1)
        self.style = style

    def gate_convs(self):
        s = self.style
        if s in ('modulate', 'max_concat', 'softmax'):
            return [self.conv_du[0], self.conv_du[2]]
        if s == 'mini_concat':
            return [self.pre_concat, self.conv_du[1]]
        if s == 'extended_attention':
            return [sec[0] for sec in self.feature_convs] + [self.final_conv[0]]
        raise NotImplementedError(s)

    def gate_from_pool(self, y, attributes, mul=None):
        """pooled vector (B,C,1,1) -> gate (B,C,1,1) [x mul]: the style's whole FC stack in one HIP launch."""
        return ops.qca_gate(y, attributes, self.style, self.gate_convs(), mul)

    def forward(self, x, attributes):
        if self.style == 'standard':
            return ops.ca_layer(x, *_ca_params(self.conv_du))
        return ops.gate_mul(x, self.gate_from_pool(ops.global_avg_pool(x), attributes))


class QRCAB(nn.Module):
    """ref: attention_manipulators/architectures.py:145-180"""

    def __init__(self, conv, n_feat, kernel_size, reduction, style='modulate', pa=False, q_layer=False, bias=True,
                 bn=False, act=None, res_scale=1, num_metadata=1):
        super().__init__()
        if bn:
            raise NotImplementedError("BatchNorm QRCAB is not used by any in-scope model")
        body = [conv(n_feat, n_feat, kernel_size, bias=bias), nn.ReLU(True), conv(n_feat, n_feat, kernel_size, bias=bias)]
        self.final_body = QCALayer(channel=n_feat, reduction=reduction, style=style, num_metadata=num_metadata)
        self.pa = pa
        self.q_layer = q_layer
        if pa:
            self.pa_node = PALayer(channel=n_feat)
        if q_layer:
            self.q_node = ParaCALayer(network_channels=n_feat, num_metadata=num_metadata, nonlinearity=True)
        self.body = nn.Sequential(*body)
        self.res_scale = res_scale

    def forward(self, x, m=None):
        """m: this block's meta gate when the network computed all of them up front (meta_gates)."""
        feat, md = x
        b = self.body
        if self.q_layer and m is None:
            m = self.q_node.gate(md)
        if self.pa:
            # per-pixel gate between the channel and meta gates: the per-(b,c) fusion does not apply
            t = ops.res_block_convs(feat, b[0].weight, b[0].bias, b[2].weight, b[2].bias)
            t = self.pa_node(self.final_body(t, md))
            if self.q_layer:
                return ops.gate_mul(t, m, feat), md
            return ops.add_residual(t, feat), md
        if self.final_body.style == 'standard':
            y = ops.res_block(feat, b[0].weight, b[0].bias, b[2].weight, b[2].bias,
                              ca=_ca_params(self.final_body.conv_du), m=m)
            return y, md
        # metadata-mixing CA styles: conv pair fused, gates composed
        t = ops.res_block_convs(feat, b[0].weight, b[0].bias, b[2].weight, b[2].bias)
        g = self.final_body.gate_from_pool(ops.global_avg_pool(t), md, m if self.q_layer else None)
        return ops.gate_mul(t, g, feat), md


class QResidualGroup(nn.Module):
    """ref: attention_manipulators/architectures.py:208-233"""

    def __init__(self, conv, n_feat, kernel_size, reduction, act, res_scale, n_resblocks, style, num_metadata, pa,
                 q_layer, num_q_layers):
        super().__init__()
        body = []
        for index in range(n_resblocks):
            q_in = q_layer if (num_q_layers is None or index < num_q_layers) else False
            body.append(QRCAB(conv, n_feat, kernel_size, reduction, bias=True, bn=False, act=act, res_scale=res_scale,
                              style=style, pa=pa, q_layer=q_in, num_metadata=num_metadata))
        self.final_body = conv(n_feat, n_feat, kernel_size)
        self.body = nn.Sequential(*body)

    def forward(self, x, gates=None):
        """gates: {id(block): meta gate} computed by the network for all its q-layers at once, or None."""
        feat, md = x
        ms = [(gates.get(id(b)) if gates else None) for b in self.body]
        if (ops.fused_groups_enabled() and feat.shape[1] == 64
                and all(not b.pa and b.final_body.style == 'standard' for b in self.body)):
            blocks = [(b.body[0].weight, b.body[0].bias, b.body[2].weight, b.body[2].bias,
                       _ca_params(b.final_body.conv_du),
                       (m if m is not None else b.q_node.gate(md)) if b.q_layer else None) for b, m in zip(self.body, ms)]
            return ops.gated_group(feat, blocks, self.final_body.weight, self.final_body.bias), md
        res = feat
        for blk, m in zip(self.body, ms):
            res, _ = blk((res, md), m)
        return _conv(self.final_body, res, residual=feat), md


class QRCAN(ChannelPadded, nn.Module):
    """ref: attention_manipulators/architectures.py:246-316"""

    def __init__(self, n_resblocks=20, n_resgroups=10, n_feats=64, in_feats=3, out_feats=3, scale=4, reduction=16,
                 res_scale=1.0, style='modulate', num_metadata=1, include_pixel_attention=False,
                 selective_meta_blocks=None, num_q_layers_inner_residual=None, include_q_layer=False, **kwargs):
        super().__init__()
        act = nn.ReLU(True)
        self.style = style
        head = [default_conv(in_feats, n_feats, 3)]
        body = []
        for index in range(n_resgroups):
            include_q = include_q_layer if (selective_meta_blocks is None or selective_meta_blocks[index]) else False
            body.append(QResidualGroup(default_conv, n_feats, 3, reduction, style=style, num_metadata=num_metadata,
                                       pa=include_pixel_attention, q_layer=include_q, act=act, res_scale=res_scale,
                                       n_resblocks=n_resblocks, num_q_layers=num_q_layers_inner_residual))
        self.final_body = default_conv(n_feats, n_feats, 3)
        tail = [Upsampler(default_conv, scale, n_feats, act=False), default_conv(n_feats, out_feats, 3)]
        self.head = nn.Sequential(*head)
        self.body = nn.Sequential(*body)
        self.tail = nn.Sequential(*tail)
        if n_feats % 64 and style not in ('standard', 'modulate'):
            raise NotImplementedError(f"QRCAN style '{style}' concatenates metadata inside its FC stack; with n_feats not a multiple "
                                      f"of 64 only 'standard' and 'modulate' are built (channel padding, ChannelPadded)")
        self._init_padding(n_feats, lambda P: QRCAN(
            n_resblocks, n_resgroups, P, in_feats, out_feats, scale, reduction, res_scale, style, num_metadata,
            include_pixel_attention, selective_meta_blocks, num_q_layers_inner_residual, include_q_layer))

    def forward(self, x, metadata):
        _check_rgb(x, "QRCAN")
        if self.padded():
            n, P = self._pad_width
            if self.style == 'modulate':  # one metadata value per feature channel (handlers.QRCANHandler.scale_qpi)
                metadata = _pad_channels(metadata, n, P)
            return self._run_padded(x, metadata)
        x = _conv(self.head[0], x)
        res = x
        qblocks = [b for g in self.body for b in g.body if b.q_layer]
        gates = dict(zip(map(id, qblocks), meta_gates([b.q_node for b in qblocks], metadata)))
        for g in self.body:
            res, _ = g((res, metadata), gates)
        res = _conv(self.final_body, res, residual=x)
        return _conv(self.tail[1], self.tail[0](res))

    def meta_gate_layers(self):
        """The layers forward() hands to meta_gates() in one batch (late_parameters)."""
        return [b.q_node for g in self.body for b in g.body if b.q_layer]


class ParamResBlock(nn.Module):
    """ref: attention_manipulators/architectures.py:332-356"""

    def __init__(self, conv, n_feats, n_params, kernel_size, act=None, bias=True, res_scale=1.0,
                 q_layer_nonlinearity=False):
        super().__init__()
        self.body = nn.Sequential(conv(n_feats, n_feats, kernel_size, bias=bias), nn.ReLU(True),
                                  conv(n_feats, n_feats, kernel_size, bias=bias))
        self.attention_layer = ParaCALayer(n_feats, n_params, nonlinearity=q_layer_nonlinearity)
        self.res_scale = res_scale

    def forward(self, x, m=None):
        feat, md = x
        b = self.body
        y = ops.res_block(feat, b[0].weight, b[0].bias, b[2].weight, b[2].bias,
                          m=m if m is not None else self.attention_layer.gate(md), res_scale=self.res_scale)
        return y, md


class QEDSR(ChannelPadded, nn.Module):
    """ref: attention_manipulators/architectures.py:359-399"""

    def __init__(self, in_features=3, out_features=3, num_features=64, input_para=1, num_blocks=16, scale=4,
                 res_scale=0.1, q_layer_nonlinearity=False, **kwargs):
        super().__init__()
        self.head = default_conv(in_features, num_features, 3)
        body = [ParamResBlock(default_conv, num_features, input_para, 3, res_scale=res_scale,
                              q_layer_nonlinearity=q_layer_nonlinearity) for _ in range(num_blocks)]
        self.final_body = default_conv(num_features, num_features, 3)
        tail = [Upsampler(default_conv, scale, num_features), default_conv(num_features, out_features, 3)]
        self.body = nn.Sequential(*body)
        self.tail = nn.Sequential(*tail)
        self._init_padding(num_features, lambda P: QEDSR(in_features, out_features, P, input_para, num_blocks, scale, res_scale,
                                                         q_layer_nonlinearity))

    def forward(self, x, metadata):
        _check_rgb(x, "QEDSR")
        if self.padded():
            return self._run_padded(x, metadata)
        x = _conv(self.head, x)
        gates = meta_gates([b.attention_layer for b in self.body], metadata)
        if ops.fused_groups_enabled() and x.shape[1] == 64 and len(self.body) > 0:
            # the whole body -- blocks, final conv, long skip -- as ONE node: the gated skips are built by the next conv's staging
            # and the meta gates' gradients taken by the previous backward conv's epilogue (ops._GatedGroup without channel
            # attention): no gate-multiply pass forward, no gradient-reduction pass backward
            blocks = [(b.body[0].weight, b.body[0].bias, b.body[2].weight, b.body[2].bias, None, m)
                      for b, m in zip(self.body, gates)]
            res = ops.gated_group(x, blocks, self.final_body.weight, self.final_body.bias, alpha=self.body[0].res_scale)
        else:
            res = x
            for blk, m in zip(self.body, gates):
                res, _ = blk((res, metadata), m)
            res = _conv(self.final_body, res, residual=x)
        return _conv(self.tail[1], self.tail[0](res))

    def meta_gate_layers(self):
        """The layers forward() hands to meta_gates() in one batch (late_parameters)."""
        return [b.attention_layer for b in self.body]
