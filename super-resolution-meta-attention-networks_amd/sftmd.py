"""SFTMD: spatial-feature-transform network conditioned on metadata maps (second half).

ref: Code/SISR/models/SFTMD_variants/architectures.py:25-176 (StandardSft, SFT_Layer, SFT_Residual_Block, SFTMD),
     Code/SISR/models/SFTMD_variants/handlers.py:6-60 (SFTMDHandler),
     Code/SISR/models/attention_manipulators/__init__.py:53-80 (generate_sft_channels).

Covered: SFT_type 'standard' (the sample config's default), 'concat', 'weak' (1 or 64 maps) and 'none'; mask_para; repeats;
q_injection (meta-attention gates after the SFT layers -- with SFT layers that take no maps, as the reference's handler then
supplies metadata vectors); scale 2 / 3 / 4; the handler's concat_strategy (metadata maps concatenated to the RGB input: the
head conv takes 3 + M <= 64 channels) and da_injection (metadata as vectors).  How the default configuration maps onto the
gfx950 kernels (ops.sft_* / ops.sftmd_forward):

  * An SFT layer's four convs run as two MFMA convs over a 128-channel map (chunk 0 = the 64 features, chunk 1 = the
    metadata maps zero-padded to 64): A = [mul_conv1 | add_conv1] merged along the outputs (128 -> 64, LeakyReLU 0.2 in
    the epilogue), B = block-diagonal [mul_conv2, add_conv2] (64 -> 128); the merged weights are composed per call from
    the four parameters (`sisr_sft_compose`) and their gradients split back; the structural zeros are skipped by the
    kernels (`select` 8 / 9, `active_units`).  Then one combine kernel: out = [relu](x * sigmoid(B[:64]) + B[64:]).
  * LeakyReLU is a conv epilogue (and a slope in the mask epilogue of the input-gradient conv); the 9x9 64 -> 3 output
    conv and its two gradients run on the matrix cores (csrc/conv_rgb_out.h, csrc/sft.hip), the clamp is its own kernel.

The nn.Module tree keeps the reference's names ('SFT-residual1.sft1.sft_module.mul_conv1.weight', ...), shapes and
construction order, so checkpoints interchange and the same seed gives the same initial weights.
"""
import torch
from torch import nn

from . import architectures as A
from . import ops
from .handlers import QModel
from .srmd import SRMDHandler


class ConcatSft(nn.Module):
    """ref: SFTMD_variants/architectures.py:8-14: conv3x3(cat(features, maps)) (parameter holder; ops.sft_apply executes it)."""

    def __init__(self, nf=64, para=1, **kwargs):
        super().__init__()
        self.conv = nn.Conv2d(para + nf, nf, kernel_size=3, stride=1, padding=1)


class WeakSft(nn.Module):
    """ref: :17-22: features * maps (maps: one channel, or as many as the features)."""


class StandardSft(nn.Module):
    """ref: SFTMD_variants/architectures.py:25-56 (parameter holder; ops.sft_layer executes it)."""

    def __init__(self, nf=64, para=1, mask_para=False, repeats=None, **kwargs):
        super().__init__()
        self.mask_para, self.repeats = bool(mask_para), repeats
        if mask_para:
            para = 0
        if repeats is not None:
            para = para * repeats
        if para > 64:
            raise NotImplementedError("SFT layers on the gfx950 kernels: at most 64 metadata channels (after `repeats`)")
        self.mul_conv1 = nn.Conv2d(para + nf, 32, kernel_size=3, stride=1, padding=1)
        self.mul_leaky = nn.LeakyReLU(0.2)
        self.mul_conv2 = nn.Conv2d(32, nf, kernel_size=3, stride=1, padding=1)
        self.add_conv1 = nn.Conv2d(para + nf, 32, kernel_size=3, stride=1, padding=1)
        self.add_leaky = nn.LeakyReLU(0.2)
        self.add_conv2 = nn.Conv2d(32, nf, kernel_size=3, stride=1, padding=1)

    def params(self):
        return (self.mul_conv1.weight, self.mul_conv1.bias, self.add_conv1.weight, self.add_conv1.bias,
                self.mul_conv2.weight, self.mul_conv2.bias, self.add_conv2.weight, self.add_conv2.bias)


class SFT_Layer(nn.Module):
    """ref: :59-77"""

    def __init__(self, sft_type='standard', **kwargs):
        super().__init__()
        self.kind, self.maps = sft_type, kwargs.get('para', 1)
        if sft_type == 'weak':
            self.sft_module = WeakSft()
        elif sft_type == 'concat':
            self.sft_module = ConcatSft(**kwargs)
        elif sft_type == 'standard':
            self.sft_module = StandardSft(**kwargs)
        elif sft_type == 'none':
            self.sft_module = None
        else:
            raise NotImplementedError("SFT type %r (the reference has 'standard', 'concat', 'weak', 'none')" % sft_type)


class SFT_Residual_Block(nn.Module):
    """ref: :80-107"""

    def __init__(self, nf=64, para=1, SFT_type='standard', mask_para=False, repeats=None, q_injection=False, q_layers=2,
                 split='22'):
        super().__init__()
        self.sft1 = SFT_Layer(nf=nf, para=para, mask_para=mask_para, repeats=repeats, sft_type=SFT_type, split=split)
        self.sft2 = SFT_Layer(nf=nf, para=para, mask_para=mask_para, repeats=repeats, sft_type=SFT_type, split=split)
        self.conv1 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.conv2 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.q_injection = bool(q_injection)
        if q_injection:
            self.q_1 = A.ParaCALayer(network_channels=nf, num_metadata=para, nonlinearity=True, num_layers=q_layers)
            self.q_2 = A.ParaCALayer(network_channels=nf, num_metadata=para, nonlinearity=True, num_layers=q_layers)


class SFTMD(nn.Module):
    """ref: :110-176"""

    def __init__(self, in_nc=3, out_nc=3, num_features=64, num_blocks=16, scale=4, input_para=1, split='22',
                 SFT_type='standard', mask_para=False, repeats=None, q_injection=False, q_layers=2, **kwargs):
        super().__init__()
        if num_features != 64 or not 3 <= in_nc <= 64 or out_nc != 3:
            raise NotImplementedError("SFTMD on the gfx950 kernels: 64 features, RGB out, RGB in (or, with the handler's "
                                      "concat_strategy, RGB + metadata maps: at most 64 input channels)")
        if input_para > 64:
            raise NotImplementedError("SFTMD: at most 64 metadata channels")
        uses_maps = SFT_type in ('concat', 'weak') or (SFT_type == 'standard' and not mask_para)
        if q_injection and uses_maps:
            # the reference's handler then feeds (B, M, 1, 1) vectors to layers that concatenate / multiply H x W maps
            raise NotImplementedError("SFTMD q_injection goes with SFT layers that take no metadata maps "
                                      "(SFT_type 'none', or mask_para)")
        if SFT_type == 'weak' and input_para not in (1, 64):
            raise NotImplementedError("SFT type 'weak' multiplies the features by the maps: 1 or 64 metadata channels")
        self.min, self.max = 0.0, 1.0
        self.para, self.num_blocks, self.scale = input_para, num_blocks, scale
        self.repeats, self.uses_maps = repeats, uses_maps
        self.conv1 = nn.Conv2d(in_nc, num_features, 3, stride=1, padding=1)
        self.relu_conv1 = nn.LeakyReLU(0.2)
        self.conv2 = nn.Conv2d(num_features, num_features, 3, stride=1, padding=1)
        self.relu_conv2 = nn.LeakyReLU(0.2)
        self.conv3 = nn.Conv2d(num_features, num_features, 3, stride=1, padding=1)
        for i in range(num_blocks):
            self.add_module('SFT-residual' + str(i + 1),
                            SFT_Residual_Block(nf=num_features, para=input_para, SFT_type=SFT_type, split=split,
                                               q_injection=q_injection, q_layers=q_layers, mask_para=mask_para,
                                               repeats=repeats))
        self.sft = SFT_Layer(nf=num_features, para=input_para, mask_para=mask_para, repeats=repeats, split=split,
                             sft_type=SFT_type)
        self.q_injection = bool(q_injection)
        if q_injection:
            self.final_injection = A.ParaCALayer(network_channels=num_features, num_metadata=input_para, nonlinearity=True,
                                                 num_layers=q_layers)
        self.conv_mid = nn.Conv2d(num_features, num_features, 3, 1, 1, bias=True)
        if scale == 4:
            self.upscale = nn.Sequential(nn.Conv2d(num_features, num_features * scale, 3, 1, 1, bias=True),
                                         nn.PixelShuffle(scale // 2), nn.LeakyReLU(0.2, inplace=True),
                                         nn.Conv2d(num_features, num_features * scale, 3, 1, 1, bias=True),
                                         nn.PixelShuffle(scale // 2), nn.LeakyReLU(0.2, inplace=True))
        else:
            self.upscale = nn.Sequential(nn.Conv2d(num_features, num_features * scale ** 2, 3, 1, 1, bias=True),
                                         nn.PixelShuffle(scale), nn.LeakyReLU(0.2, inplace=True))
        self.conv_output = nn.Conv2d(num_features, out_nc, kernel_size=9, stride=1, padding=4, bias=True)

    def blocks(self):
        return [getattr(self, 'SFT-residual' + str(i + 1)) for i in range(self.num_blocks)]

    def forward(self, x, metadata):
        if not x.is_cuda:
            raise RuntimeError("SFTMD: this network only runs on a HIP device (no CPU fallback); got a CPU tensor")
        return ops.sftmd_forward(self, x, metadata)


class SFTMDHandler(QModel):
    """ref: SFTMD_variants/handlers.py:6-60"""

    def __init__(self, device, eval_mode=False, lr=1e-4, scheduler=None, concat_strategy=False, scheduler_params=None,
                 perceptual=None, q_injection=False, da_injection=False, in_nc=3, optimizer_params=None, **kwargs):
        super().__init__(device=device, eval_mode=eval_mode, **kwargs)
        if concat_strategy:  # ref :12-14: the metadata maps are concatenated to the RGB input (QModel.channel_concat_logic)
            self.channel_concat = True
            in_nc = self.num_metadata + in_nc
        # ref :16-17: da_injection is handed to the network, whose constructor swallows it (architectures.py:108-110 **kwargs):
        # its only effect is the metadata FORMAT below
        self.net = SFTMD(input_para=self.num_metadata, q_injection=q_injection, in_nc=in_nc, **kwargs)
        self.vector_metadata = bool(q_injection or da_injection)  # ref: handlers.py:19-22
        self.colorspace = 'augmented_rgb'
        self.im_input = 'unmodified'
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device, optimizer_params=optimizer_params)
        self.model_name = 'sftmd'

    generate_sft_channels = SRMDHandler.generate_sft_channels

    def generate_channels(self, x, metadata, metadata_keys, vector_override=False):
        if self.vector_metadata or vector_override:
            return super().generate_channels(x, metadata, metadata_keys)
        return self.generate_sft_channels(x, metadata, metadata_keys)

    def legacy_switch(self, state_dict):
        from collections import OrderedDict
        new = super().legacy_switch(state_dict)
        out = OrderedDict()
        for k, v in new.items():
            if 'sft_branch' in k:
                continue
            if 'sft_module' in k:
                out[k] = v
            elif 'sft1' in k or 'sft2' in k:
                out[k.replace('sft1', 'sft1.sft_module').replace('sft2', 'sft2.sft_module')] = v
            elif k[:4] == 'sft.':
                out[k.replace('sft.', 'sft.sft_module.')] = v
            else:
                out[k] = v
        return out
