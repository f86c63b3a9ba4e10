"""SPARNet / QSPARNet: the face-SR residual network with hourglass spatial attention (SURVEY.md 8f-4 "then SPARNet").

ref: Code/SISR/models/SPARNet/architectures.py:7-155 (SPARNet, QSPARNet), SPARNet/blocks.py:10-243 (NormLayer, ReluLayer,
     ConvLayer, ResidualBlock, HourGlassBlock), SPARNet/handlers.py:6-36 (SPARNetHandler, QSPARNetHandler).

Every conv of the network is a reference ConvLayer: [nearest x2] -> ReflectionPad2d(1) -> Conv2d(3x3, stride 1 | 2) ->
[BatchNorm2d] -> [LeakyReLU(0.2)], at 32 / 64 / 128 channels (and 3 at the RGB ends, 1 for the attention map) on maps of
128^2 down to 4^2 pixels.  On the gfx950 kernels (ops.refl_conv / batch_norm_act / spar_combine, csrc/sparnet.hip):

  * feature maps are channels-last with the channel count zero-padded to a multiple of 64 (32 -> 64, 3 -> 64, 1 -> 64): the
    padded channels stay exactly zero through conv (zero-padded weights), batch norm (written as zero) and attention;
  * a ConvLayer's conv is ONE autograd node: a gather builds the upsampled + reflection-padded map, the zero-padded MFMA conv
    runs over it, a second gather takes the interior (every second pixel for the stride-2 convs); its backward runs the MFMA
    input-gradient and weight-gradient kernels on the same padded geometry between the adjoint gathers;
  * BatchNorm2d + LeakyReLU is one op (two-pass batch statistics, running statistics updated as torch does; eval mode uses
    the running statistics);
  * the hourglass output (64 -> 1 conv, sigmoid, broadcast product) and the block's residual sum are one kernel each way;
  * QSPARNet's per-block meta-attention is the ParaCALayer of the other Q-models (gate from the metadata vector, HIP gate
    multiply), its gate zero-padded with the map.

The nn.Module tree keeps the reference's names (encoder.1.conv1.conv2d.weight, res_layers.3.att_func.b2_plus_1.norm.norm
.running_var, ...), shapes and construction order, so checkpoints interchange and the same seed gives the same initial
weights.  Options: norm_type 'bn' | 'in' | 'gn' | 'pixel' | 'none' ('layer' raises a TypeError, as it does in the reference, whose
callers never pass the shape nn.LayerNorm needs), relu_type 'leakyrelu' | 'relu' | 'prelu' | 'selu' | 'none', att_name 'spar' |
'spar3d'.  The reference's defaults (batch norm + LeakyReLU, one attention channel) run on the fused kernels; the other
options are one plain HIP pass per norm / activation and direction (csrc/sparnet.hip, "non-default ConvLayer options";
fixtures P5).  Map sizes must halve exactly through the hourglasses (powers of two times the bottleneck, as the reference's
defaults are) -- the reference's fallback interpolation for odd sizes is not built.
"""
import numpy as np
import torch
from torch import nn

from . import architectures as A
from . import ops
from .handlers import BaseModel, QModel, L1Loss


class NormLayer(nn.Module):
    """ref: blocks.py:10-36 (parameter / buffer holder: `norm` is the nn.BatchNorm2d, or None for 'none')"""

    def __init__(self, channels, normalize_shape=None, norm_type='bn'):
        super().__init__()
        norm_type = norm_type.lower()
        self.kind, self.channels = norm_type, channels
        if norm_type == 'bn':
            self.norm = nn.BatchNorm2d(channels)
        elif norm_type == 'in':
            self.norm = nn.InstanceNorm2d(channels, affine=True)
        elif norm_type == 'gn':
            self.norm = nn.GroupNorm(32, channels, affine=True)  # (raises for channel counts that 32 does not divide, as the reference)
        elif norm_type in ('pixel', 'none'):
            self.norm = None
        elif norm_type == 'layer':
            # ref blocks.py:27-28 builds nn.LayerNorm(normalize_shape), and every caller leaves normalize_shape at None: the
            # reference itself raises a TypeError for this option
            raise TypeError("norm_type 'layer': the reference's ConvLayer / ResidualBlock never pass normalize_shape, so "
                            "nn.LayerNorm(None) fails there too")
        else:
            raise AssertionError('Norm type {} not support.'.format(norm_type))


class ReluLayer(nn.Module):
    """ref: blocks.py:39-66 (`slope`: the activation as a LeakyReLU slope; 1 = none)"""

    def __init__(self, channels, relu_type='relu'):
        super().__init__()
        relu_type = relu_type.lower()
        self.kind = relu_type
        if relu_type == 'relu':
            self.func, self.slope = nn.ReLU(True), 0.0
        elif relu_type == 'leakyrelu':
            self.func, self.slope = nn.LeakyReLU(0.2, inplace=True), 0.2
        elif relu_type == 'prelu':
            self.func, self.slope = nn.PReLU(channels), None
        elif relu_type == 'selu':
            self.func, self.slope = nn.SELU(True), None
        elif relu_type == 'none':
            self.func, self.slope = None, 1.0
        else:
            raise AssertionError('Relu type {} not support.'.format(relu_type))


def _norm_act(x, norm, relu):
    """norm then activation.  Batch norm with (Leaky)ReLU / no activation -- the reference's defaults -- is ONE kernel; every other
    combination is the norm's op followed by the activation's."""
    if norm.kind == 'bn' and relu.slope is not None:
        return ops.batch_norm_act(x, norm.norm, slope=relu.slope)
    if norm.kind == 'bn':
        x = ops.batch_norm_act(x, norm.norm, slope=1.0)
    elif norm.kind == 'in':
        x = ops.group_norm(x, norm.norm.weight, norm.norm.bias, 1, norm.norm.eps)
    elif norm.kind == 'gn':
        x = ops.group_norm(x, norm.norm.weight, norm.norm.bias, norm.channels // 32, norm.norm.eps)
    elif norm.kind == 'pixel':
        x = ops.pixel_norm(x)
    if relu.kind == 'prelu':
        return ops.prelu(x, relu.func.weight)
    if relu.kind == 'selu':
        return ops.selu(x)
    if relu.kind in ('relu', 'leakyrelu'):
        return ops.leaky_relu(x, relu.slope)
    return x


class ConvLayer(nn.Module):
    """ref: blocks.py:69-103.  Takes and returns channels-last maps zero-padded to 64-multiples (or (map, metadata) tuples,
    as the reference's Sequential containers pass them)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, scale='none', norm_type='none', relu_type='none',
                 use_pad=True):
        super().__init__()
        if kernel_size != 3 or not use_pad:
            raise NotImplementedError("SPARNet ConvLayer on the gfx950 kernels: 3x3 with reflection padding")
        self.use_pad, self.scale = use_pad, scale
        bias = norm_type in ('pixel', 'none')
        self.reflection_pad = nn.ReflectionPad2d(kernel_size // 2)
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, 2 if scale == 'down' else 1, bias=bias)
        self.relu = ReluLayer(out_channels, relu_type)
        self.norm = NormLayer(out_channels, norm_type=norm_type)

    def forward(self, x):
        data = x[0] if type(x) == tuple else x
        out = ops.refl_conv(data, self.conv2d.weight, self.conv2d.bias, up=2 if self.scale == 'up' else 1,
                            stride=2 if self.scale == 'down' else 1)
        out = _norm_act(out, self.norm, self.relu)
        return (out, x[1]) if type(x) == tuple else out


class HourGlassBlock(nn.Module):
    """ref: blocks.py:177-243.  forward(x, identity) returns identity + x * attention (the caller's residual sum is part of
    the same kernel); the last attention map is kept in `att_map` as the reference does."""

    def __init__(self, depth, c_in, c_out, c_mid=64, norm_type='bn', relu_type='prelu'):
        super().__init__()
        self.depth, self.c_in, self.c_mid, self.c_out = depth, c_in, c_mid, c_out
        self.kwargs = {'norm_type': norm_type, 'relu_type': relu_type}
        if c_out not in (1, c_in):
            raise NotImplementedError("hourglass attention: one attention channel ('spar') or one per feature ('spar3d')")
        if self.depth:
            self._generate_network(self.depth)
            self.out_block = nn.Sequential(ConvLayer(self.c_mid, self.c_out, norm_type='none', relu_type='none'), nn.Sigmoid())

    def _generate_network(self, level):
        c1, c2 = (self.c_in, self.c_mid) if level == self.depth else (self.c_mid, self.c_mid)
        self.add_module('b1_' + str(level), ConvLayer(c1, c2, **self.kwargs))
        self.add_module('b2_' + str(level), ConvLayer(c1, c2, scale='down', **self.kwargs))
        if level > 1:
            self._generate_network(level - 1)
        else:
            self.add_module('b2_plus_' + str(level), ConvLayer(self.c_mid, self.c_mid, **self.kwargs))
        self.add_module('b3_' + str(level), ConvLayer(self.c_mid, self.c_mid, scale='up', **self.kwargs))

    def _forward(self, level, in_x):
        up1 = self._modules['b1_' + str(level)](in_x)
        low1 = self._modules['b2_' + str(level)](in_x)
        low2 = self._forward(level - 1, low1) if level > 1 else self._modules['b2_plus_' + str(level)](low1)
        up2 = self._modules['b3_' + str(level)](low2)
        if up1.shape[2:] != up2.shape[2:]:
            raise NotImplementedError("hourglass on a map whose size does not halve exactly (%s vs %s): the reference's "
                                      "interpolation fallback is not built" % (tuple(up1.shape[2:]), tuple(up2.shape[2:])))
        return ops.add_residual(up1, up2)

    def forward(self, x, identity=None):
        if self.depth == 0:
            return x if identity is None else ops.add_residual(identity, x)
        logits = self.out_block[0](self._forward(self.depth, x))
        out = ops.spar_combine(x, logits, identity) if self.c_out == 1 else ops.spar_combine3d(x, logits, identity)
        self.att_map = None  # (the reference keeps the map for visualisation; it lives in the kernel's saved state here)
        return out


class ResidualBlock(nn.Module):
    """ref: blocks.py:106-174"""

    def __init__(self, c_in, c_out, relu_type='prelu', norm_type='bn', scale='none', hg_depth=2, att_name='spar',
                 include_metadata=None):
        super().__init__()
        self.c_in, self.c_out, self.norm_type, self.relu_type, self.hg_depth = c_in, c_out, norm_type, relu_type, hg_depth
        kwargs = {'norm_type': norm_type, 'relu_type': relu_type}
        self.shortcut_func = None if (scale == 'none' and c_in == c_out) else ConvLayer(c_in, c_out, 3, scale)
        self.preact_func = nn.Sequential(NormLayer(c_in, norm_type=self.norm_type), ReluLayer(c_in, self.relu_type))
        scales = {'down': ['none', 'down'], 'up': ['up', 'none'], 'none': ['none', 'none']}[scale]
        self.conv1 = ConvLayer(c_in, c_out, 3, scales[0], **kwargs)
        self.conv2 = ConvLayer(c_out, c_out, 3, scales[1], norm_type=norm_type, relu_type='none')
        if att_name.lower() == 'spar':
            c_attn = 1
        elif att_name.lower() == 'spar3d':
            c_attn = c_out
        else:
            raise Exception("Attention type {} not implemented".format(att_name))
        self.att_func = HourGlassBlock(self.hg_depth, c_out, c_attn, **kwargs)
        self.include_metadata = include_metadata is not None
        if self.include_metadata:
            self.metadata_attention = A.ParaCALayer(network_channels=self.c_out, num_metadata=include_metadata, nonlinearity=True)

    def forward(self, x):
        data = x[0] if type(x) == tuple else x
        identity = data if self.shortcut_func is None else self.shortcut_func(data)
        out = _norm_act(data, self.preact_func[0], self.preact_func[1])
        out = self.conv2(self.conv1(out))
        out = self.att_func(out, identity)
        if type(x) == tuple:
            if self.include_metadata:
                g = self.metadata_attention.gate(x[1])
                pad = out.shape[1] - g.shape[1]
                out = ops.gate_mul(out, nn.functional.pad(g, (0, pad)) if pad else g)
            return out, x[1]
        return out


def _build(net, min_ch, max_ch, in_size, out_size, min_feat_size, res_depth, relu_type, norm_type, att_name, bottleneck_size,
           metadata_count, metadata_encoder_only):
    """The layer plan shared by SPARNet and QSPARNet (ref: architectures.py:17-76, :84-155)."""
    nrargs = {'norm_type': norm_type, 'relu_type': relu_type}

    def ch_clip(x):
        return max(min_ch, min(x, max_ch))

    def md(encoder):
        if metadata_count is None:
            return {}
        return {'include_metadata': metadata_count if (encoder or not metadata_encoder_only) else None}

    down_steps = int(np.log2(in_size // min_feat_size))
    up_steps = int(np.log2(out_size // min_feat_size))
    n_ch = ch_clip(max_ch // int(np.log2(in_size // min_feat_size) + 1))
    if max_ch > 256 or min_ch < 1:
        raise NotImplementedError("SPARNet on the gfx950 kernels: at most 256 channels")
    encoder = [ConvLayer(3, n_ch, 3, 1)]
    hg_depth = int(np.log2(64 / bottleneck_size))
    for _ in range(down_steps):
        cin, cout = ch_clip(n_ch), ch_clip(n_ch * 2)
        encoder.append(ResidualBlock(cin, cout, scale='down', hg_depth=hg_depth, att_name=att_name, **md(True), **nrargs))
        n_ch, hg_depth = n_ch * 2, hg_depth - 1
    hg_depth = hg_depth + 1
    net.encoder = nn.Sequential(*encoder)
    res_layers = []
    for _ in range(res_depth + 3 - down_steps):
        channels = ch_clip(n_ch)
        res_layers.append(ResidualBlock(channels, channels, hg_depth=hg_depth, att_name=att_name, **md(False), **nrargs))
    net.res_layers = nn.Sequential(*res_layers)
    decoder = []
    for _ in range(up_steps):
        hg_depth = hg_depth + 1
        cin, cout = ch_clip(n_ch), ch_clip(n_ch // 2)
        decoder.append(ResidualBlock(cin, cout, scale='up', hg_depth=hg_depth, att_name=att_name, **md(False), **nrargs))
        n_ch = n_ch // 2
    net.decoder = nn.Sequential(*decoder)
    net.out_conv = ConvLayer(ch_clip(n_ch), 3, 3, 1)


def _rgb_in(img):
    if not img.is_cuda:
        raise RuntimeError("SPARNet: this network only runs on a HIP device (no CPU fallback); got a CPU tensor")
    return ops.nchw_to_nhwc_pad(img)


class _count_batches:
    """nn.BatchNorm2d.forward adds one to num_batches_tracked per training forward: the network's ~190 counters in one
    multi-tensor add instead of ~190 one-element kernels.  While the network's forward runs, every batch norm that was counted
    here carries a mark that tells ops.batch_norm_act to leave its counter alone; the marks come off when the forward returns,
    so a sub-module called on its own later counts for itself again.  A batch norm counts iff ITS OWN .training is set (a
    frozen batch norm inside a training network does not)."""

    def __init__(self, net):
        self.bns = [m for m in net.modules() if isinstance(m, nn.BatchNorm2d) and m.training and m.num_batches_tracked is not None]

    def __enter__(self):
        for m in self.bns:
            m._sisr_counted_by_net = True
        if self.bns:
            torch._foreach_add_([m.num_batches_tracked for m in self.bns], 1)
        return self

    def __exit__(self, *exc):
        for m in self.bns:
            m._sisr_counted_by_net = False
        return False


class SPARNet(nn.Module):
    """ref: architectures.py:7-76"""
    pack_padded_convs = True  # architectures.conv_weights: every 3x3 weight rides in the step's one packing launch, zero-padded

    def __init__(self, min_ch=32, max_ch=128, in_size=128, out_size=128, min_feat_size=16, res_depth=10,
                 relu_type='leakyrelu', norm_type='bn', att_name='spar', bottleneck_size=4, **kwargs):
        super().__init__()
        _build(self, min_ch, max_ch, in_size, out_size, min_feat_size, res_depth, relu_type, norm_type, att_name,
               bottleneck_size, None, False)

    def forward(self, input_img):
        with _count_batches(self):
            out = self.encoder(_rgb_in(input_img))
            out = self.res_layers(out)
            out = self.decoder(out)
            return ops.shuffle_rgb(self.out_conv(out), 3, 1)


class QSPARNet(nn.Module):
    """ref: architectures.py:79-155"""
    pack_padded_convs = True

    def __init__(self, min_ch=32, max_ch=128, in_size=128, out_size=128, min_feat_size=16, res_depth=10,
                 relu_type='leakyrelu', norm_type='bn', att_name='spar', bottleneck_size=4, metadata_count=None,
                 metadata_encoder_only=False, **kwargs):
        super().__init__()
        _build(self, min_ch, max_ch, in_size, out_size, min_feat_size, res_depth, relu_type, norm_type, att_name,
               bottleneck_size, metadata_count, metadata_encoder_only)

    def forward(self, input_img, metadata):
        with _count_batches(self):
            out, _ = self.encoder((_rgb_in(input_img), metadata))
            out, _ = self.res_layers((out, metadata))
            out, _ = self.decoder((out, metadata))
            return ops.shuffle_rgb(self.out_conv(out), 3, 1)


class SPARNetHandler(BaseModel):
    """ref: SPARNet/handlers.py:6-20"""

    def __init__(self, device, model_save_dir, eval_mode=False, lr=1e-4, scale=4, hr_data_loc=None, scheduler=None,
                 scheduler_params=None, perceptual=None, **kwargs):
        super().__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode, hr_data_loc=hr_data_loc, **kwargs)
        self.net = SPARNet(**kwargs)
        self.colorspace = 'rgb'
        self.im_input = 'interp'
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device)
        self.model_name = 'sparnet'
        self.criterion = L1Loss()
        self.scale = scale


class QSPARNetHandler(QModel):
    """ref: SPARNet/handlers.py:23-36"""

    def __init__(self, device, model_save_dir, eval_mode=False, lr=1e-4, scale=4, hr_data_loc=None, scheduler=None,
                 scheduler_params=None, perceptual=None, **kwargs):
        super().__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode, hr_data_loc=hr_data_loc, **kwargs)
        self.net = QSPARNet(metadata_count=self.num_metadata, **kwargs)
        self.colorspace = 'rgb'
        self.im_input = 'interp'
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device)
        self.model_name = 'qsparnet'
        self.criterion = L1Loss()
        self.scale = scale
