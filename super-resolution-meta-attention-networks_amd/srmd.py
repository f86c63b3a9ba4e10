"""SRMD: the metadata-map network on the HIP conv kernels.

ref: Code/SISR/models/advanced/architectures.py:380-425 (SRMD), advanced/SRMD_blocks.py:9-126 (sequential / conv /
     upsample_pixelshuffle), advanced/handlers.py:132-158 (SRMDHandler),
     attention_manipulators/__init__.py:53-80 (generate_sft_channels).

The reference concatenates the blur-kernel code, stretched to H x W maps, to the RGB input and runs a plain stack
conv(3+M -> nc) ReLU [conv(nc -> nc) ReLU] x (nb-2) conv(nc -> 3 r^2) PixelShuffle(r).  Here the (3+M)-channel NCHW
input is laid out once as a zero-padded channels-last map (HIP), the whole conv stack is one autograd node on the MFMA
3x3 kernels (ops.conv_chain: zero-padded head / tail weights, ReLU masks applied by the input-gradient epilogues) and
the tail's result is shuffled into the NCHW image by a HIP gather.  `model` keeps the reference's flat nn.Sequential
indices, so state-dict keys (model.0.weight, model.2.weight, ...) and the seed-8 initial weights are the reference's.
"""
import numpy as np
import torch
from torch import nn

from . import ops
from .handlers import QModel


class SRMD(nn.Module):
    def __init__(self, in_nc=18, out_nc=3, nc=128, nb=12, scale=4, act_mode='R', upsample_mode='pixelshuffle', **kwargs):
        super().__init__()
        assert 'R' in act_mode or 'L' in act_mode, 'Examples of activation function: R, L, BR, BL, IR, IL'
        if act_mode != 'R':
            raise NotImplementedError("SRMD on the HIP kernels implements the reference's default act_mode='R' "
                                      "(no BatchNorm / LeakyReLU variants)")
        if upsample_mode != 'pixelshuffle':
            raise NotImplementedError("upsample mode [%s] is not built (the reference default is 'pixelshuffle')" % upsample_mode)
        if nc % 64:
            raise NotImplementedError("SRMD: nc must be a multiple of 64 for the gfx950 conv kernels (reference default 128)")
        layers = [nn.Conv2d(in_nc, nc, 3, 1, 1, bias=True), nn.ReLU(inplace=True)]
        for _ in range(nb - 2):
            layers += [nn.Conv2d(nc, nc, 3, 1, 1, bias=True), nn.ReLU(inplace=True)]
        layers += [nn.Conv2d(nc, out_nc * scale ** 2, 3, 1, 1, bias=True), nn.PixelShuffle(upscale_factor=scale)]
        self.model = nn.Sequential(*layers)
        self.scale, self.out_nc = scale, out_nc

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("SRMD: this network only runs on a HIP device (no CPU fallback); got a CPU tensor")
        mods = list(self.model)
        convs = [m for m in mods if isinstance(m, nn.Conv2d)]
        feat = ops.nchw_to_nhwc_pad(x)
        chain = [(c.weight, c.bias, i < len(convs) - 1) for i, c in enumerate(convs)]
        return ops.shuffle_rgb(ops.conv_chain(feat, chain), self.out_nc, self.scale)


class SRMDHandler(QModel):
    """ref: advanced/handlers.py:132-158"""

    def __init__(self, device, model_save_dir, eval_mode=False, lr=1e-4, scheduler=None, scheduler_params=None,
                 in_features=3, perceptual=None, **kwargs):
        super().__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode, **kwargs)
        self.net = SRMD(in_nc=in_features + self.num_metadata, **kwargs)
        self.colorspace = 'augmented_rgb'
        self.im_input = 'unmodified'
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device)
        self.model_name = 'srmd'
        self.channel_concat = True
        self.legacy_load = False

    def generate_sft_channels(self, x, metadata, metadata_keys):
        """(B, M) metadata -> (B, num_metadata, H, W) maps on the host (ref: attention_manipulators/__init__.py:53-80)."""
        if metadata is None:
            raise RuntimeError('Metadata needs to be specified for this network to run properly.')
        md = torch.as_tensor(np.asarray(metadata)) if not torch.is_tensor(metadata) else metadata
        H, W = x.shape[2], x.shape[3]
        mask = [key[0] in self.metadata for key in metadata_keys]
        maps = torch.ones(x.size(0), self.num_metadata, H, W)
        for index in range(x.size(0)):
            info = md[index] if len(metadata_keys) == 1 else md[index][mask]
            info = torch.as_tensor(info)
            if self.num_metadata == 1:
                maps[index] = maps[index] * info
            else:
                maps[index] = info.to(torch.float32).reshape(-1, 1, 1).expand(self.num_metadata, H, W)
        return maps

    def run_train(self, x, y, metadata=None, metadata_keys=None, *args, **kwargs):
        extra = self.generate_sft_channels(x, metadata, metadata_keys)
        return super().run_train(x, y, extra_channels=extra, **kwargs)

    def train_step(self, x, y, metadata=None, extra_channels=None, metadata_keys=None, **kwargs):
        if extra_channels is None:
            extra_channels = self.generate_sft_channels(x, metadata, metadata_keys)
        if x.shape[1] != self.net.model[0].weight.shape[1]:  # not yet concatenated (direct train_step callers)
            x = torch.cat((x, extra_channels.to(x.device)), 1)
        return super(QModel, self).train_step(x, y, **kwargs)

    def run_eval(self, x, y=None, metadata=None, metadata_keys=None, request_loss=False, *args, **kwargs):
        extra = self.generate_sft_channels(x, metadata, metadata_keys)
        return super().run_eval(x, y, extra_channels=extra, request_loss=request_loss, **kwargs)

    def run_model(self, x, *args, **kwargs):
        return self.net.forward(x)
