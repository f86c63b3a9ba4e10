"""SFTMD: spatial-feature-transform network conditioned on metadata maps (second half).

ref: Code/SISR/models/SFTMD_variants/architectures.py:25-176 (StandardSft, SFT_Layer, SFT_Residual_Block, SFTMD),
     Code/SISR/models/SFTMD_variants/handlers.py:6-60 (SFTMDHandler),
     Code/SISR/models/attention_manipulators/__init__.py:53-80 (generate_sft_channels).

Reference configuration covered: SFT_type 'standard', no q / da injection, mask_para False, repeats None, scale 2 / 3 / 4
(the sample config's defaults).  How it maps onto the gfx950 kernels (ops.sft_*):

  * Feature maps that feed an SFT layer live in 128-channel channels-last tensors: chunk 0 = the 64 features, chunk 1 =
    the metadata maps (M <= 64 channels, zero padded), so `cat(features, para_maps)` is never materialised -- the conv
    that produces the features writes chunk 0 in place (pixel stride 128), chunk 1 is filled by the combine kernel.
  * An SFT layer's four convs run as two MFMA convs: A = [mul_conv1 | add_conv1] merged along the outputs
    (128 -> 64, LeakyReLU 0.2 in the epilogue), B = block-diagonal [mul_conv2, add_conv2] (64 -> 128); the merged /
    block-diagonal weights are composed per step from the four parameters (`sisr_sft_compose`) and their gradients
    split back.  Then one combine kernel: out = [relu](x * sigmoid(B[:64]) + B[64:]).
  * LeakyReLU is a conv epilogue (and a slope in the ReLU-mask epilogue of the input-gradient conv); the 9x9 64 -> 3 output
    conv, its two gradients and the clamp are their own kernels (csrc/sft.hip).

The nn.Module tree keeps the reference's names ('SFT-residual1.sft1.sft_module.mul_conv1.weight', ...), shapes and
construction order, so checkpoints interchange and the same seed gives the same initial weights.
"""
import torch
from torch import nn

from . import ops
from .handlers import QModel
from .srmd import SRMDHandler


class StandardSft(nn.Module):
    """ref: SFTMD_variants/architectures.py:25-56 (parameter holder; ops.sft_layer executes it)."""

    def __init__(self, nf=64, para=1, mask_para=False, repeats=None, **kwargs):
        super().__init__()
        if mask_para or repeats is not None:
            raise NotImplementedError("SFT layers with mask_para / repeats are not built (reference defaults: off)")
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
        if sft_type != 'standard':
            raise NotImplementedError("SFT type %r is not built (the reference default and sample config use 'standard')" % sft_type)
        self.sft_module = StandardSft(**kwargs)


class SFT_Residual_Block(nn.Module):
    """ref: :80-107"""

    def __init__(self, nf=64, para=1, SFT_type='standard', mask_para=False, repeats=None, q_injection=False, q_layers=2,
                 split='22'):
        super().__init__()
        if q_injection:
            raise NotImplementedError("SFTMD q-injection is not built (reference default: off)")
        self.sft1 = SFT_Layer(nf=nf, para=para, mask_para=mask_para, repeats=repeats, sft_type=SFT_type, split=split)
        self.sft2 = SFT_Layer(nf=nf, para=para, mask_para=mask_para, repeats=repeats, sft_type=SFT_type, split=split)
        self.conv1 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.conv2 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.q_injection = False


class SFTMD(nn.Module):
    """ref: :110-176"""

    def __init__(self, in_nc=3, out_nc=3, num_features=64, num_blocks=16, scale=4, input_para=1, split='22',
                 SFT_type='standard', mask_para=False, repeats=None, q_injection=False, q_layers=2, **kwargs):
        super().__init__()
        if num_features != 64 or in_nc != 3 or out_nc != 3:
            raise NotImplementedError("SFTMD on the gfx950 kernels: 64 features, RGB in / out (the reference's configuration)")
        if input_para > 64:
            raise NotImplementedError("SFTMD: at most 64 metadata channels")
        if q_injection:
            raise NotImplementedError("SFTMD q-injection is not built (reference default: off)")
        self.min, self.max = 0.0, 1.0
        self.para, self.num_blocks, self.scale = input_para, num_blocks, scale
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
        self.q_injection = False
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
        if concat_strategy or q_injection or da_injection:
            raise NotImplementedError("SFTMD: concat_strategy / q_injection / da_injection are not built (defaults: off)")
        self.net = SFTMD(input_para=self.num_metadata, q_injection=q_injection, da_injection=da_injection, in_nc=in_nc,
                         **kwargs)
        self.vector_metadata = False
        self.colorspace = 'augmented_rgb'
        self.im_input = 'unmodified'
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device, optimizer_params=optimizer_params)
        self.model_name = 'sftmd'

    generate_sft_channels = SRMDHandler.generate_sft_channels

    def generate_channels(self, x, metadata, metadata_keys, vector_override=False):
        if vector_override:
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
