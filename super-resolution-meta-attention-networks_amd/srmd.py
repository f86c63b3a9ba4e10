"""SRMD: the metadata-map network on the HIP conv kernels.

ref: Code/SISR/models/advanced/architectures.py:380-425 (SRMD), advanced/SRMD_blocks.py:9-126 (sequential / conv /
     upsample_pixelshuffle), advanced/handlers.py:132-158 (SRMDHandler),
     attention_manipulators/__init__.py:53-80 (generate_sft_channels).

The reference concatenates the blur-kernel code, stretched to H x W maps, to the RGB input and runs a plain stack
conv(3+M -> nc) ReLU [conv(nc -> nc) ReLU] x (nb-2) conv(nc -> 3 r^2) PixelShuffle(r) (act_mode 'R'; 'L', 'BR', 'BL' put a
LeakyReLU / BatchNorm2d(momentum 0.9, eps 1e-4) there, upsample_mode 'upconv' ends in nearest upsample + conv).  Here the (3+M)-channel NCHW
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


def _layers(in_channels, out_channels, mode, negative_slope=0.2):
    """ref: advanced/SRMD_blocks.py:33-68 `conv`: one module per character of `mode`, in order (so the flat Sequential indices,
    hence the state-dict keys, are the reference's)."""
    out = []
    for t in mode:
        if t == 'C':
            out.append(nn.Conv2d(in_channels, out_channels, 3, 1, 1, bias=True))
        elif t == 'B':
            out.append(nn.BatchNorm2d(out_channels, momentum=0.9, eps=1e-04, affine=True))
        elif t in 'Rr':
            out.append(nn.ReLU(inplace=t == 'R'))
        elif t in 'Ll':
            out.append(nn.LeakyReLU(negative_slope=negative_slope, inplace=t == 'L'))
        elif t in '234':
            out.append(nn.PixelShuffle(upscale_factor=int(t)))
        elif t in 'Uuv':
            out.append(nn.Upsample(scale_factor={'U': 2, 'u': 3, 'v': 4}[t], mode='nearest'))
        elif t == 'I':
            out.append(nn.InstanceNorm2d(out_channels, affine=True))
        else:
            raise NotImplementedError('Undefined type: ' + t)
    return out


class SRMD(nn.Module):
    """ref: advanced/architectures.py:380-425.  act_mode 'R' (the reference default), 'L', 'BR', 'BL', 'IR', 'IL' (and the
    non-inplace 'r' / 'l' spellings); upsample_mode 'pixelshuffle' (default), 'upconv' (nearest upsample + conv) and
    'convtranspose' (ConvTranspose2d with kernel = stride = scale)."""

    def __init__(self, in_nc=18, out_nc=3, nc=128, nb=12, scale=4, act_mode='R', upsample_mode='pixelshuffle', **kwargs):
        super().__init__()
        assert 'R' in act_mode or 'L' in act_mode, 'Examples of activation function: R, L, BR, BL, IR, IL'
        if upsample_mode not in ('pixelshuffle', 'upconv', 'convtranspose'):
            raise NotImplementedError('upsample mode [{:s}] is not found'.format(upsample_mode))
        if nc % 64 or nc > 256:
            raise NotImplementedError("SRMD: nc must be a multiple of 64 (at most 256) for the gfx950 kernels (reference default 128)")
        layers = _layers(in_nc, nc, 'C' + act_mode[-1])
        for _ in range(nb - 2):
            layers += _layers(nc, nc, 'C' + act_mode)
        if upsample_mode == 'pixelshuffle':  # ref SRMD_blocks.py:123-126: conv(nc -> out_nc r^2) + PixelShuffle(r)
            layers += _layers(nc, out_nc * scale ** 2, 'C' + str(scale))
        elif upsample_mode == 'upconv':  # ref SRMD_blocks.py:132-142: Upsample(nearest, r) + conv(nc -> out_nc)
            layers += _layers(nc, out_nc, {2: 'UC', 3: 'uC', 4: 'vC'}[scale])
        else:  # ref SRMD_blocks.py:148-154: ConvTranspose2d(nc, out_nc, kernel_size = stride = r, padding 0)
            layers.append(nn.ConvTranspose2d(nc, out_nc, kernel_size=scale, stride=scale, padding=0, bias=True))
        self.model = nn.Sequential(*layers)
        self.scale, self.out_nc = scale, out_nc

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("SRMD: this network only runs on a HIP device (no CPU fallback); got a CPU tensor")
        feat = ops.nchw_to_nhwc_pad(x)
        mods, chain, shuffle = list(self.model), [], 1

        def flush(t):
            nonlocal chain
            if chain:
                t = ops.conv_chain(t, [tuple(c) for c in chain])
                chain = []
            return t

        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Conv2d):
                chain.append([m.weight, m.bias, 0])
            elif isinstance(m, (nn.ReLU, nn.LeakyReLU)):
                if isinstance(m, nn.LeakyReLU) and abs(m.negative_slope - 0.2) > 1e-12:
                    raise NotImplementedError("LeakyReLU slope 0.2 only")
                chain[-1][2] = 2 if isinstance(m, nn.LeakyReLU) else 1
            elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d)):  # conv -> norm -> activation: one op (activation folded in)
                feat = flush(feat)
                act = mods[i + 1] if i + 1 < len(mods) else None
                slope = 0.2 if isinstance(act, nn.LeakyReLU) else (0.0 if isinstance(act, nn.ReLU) else 1.0)
                feat = (ops.batch_norm_act if isinstance(m, nn.BatchNorm2d) else ops.instance_norm_act)(feat, m, slope=slope)
                if slope != 1.0:
                    i += 1
            elif isinstance(m, nn.ConvTranspose2d):
                # kernel = stride = r, no padding: every LR pixel maps its nc features to r x r output pixels of out_nc channels
                # -- a 1 x 1 conv to out_nc r^2 channels in PixelShuffle order.  It runs as the centre tap of a 3 x 3 conv on the
                # MFMA kernel (the eight zero taps are wasted arithmetic on one small layer; gradients reach m.weight through the
                # index arithmetic below), followed by the same shuffle gather as the default tail.
                r, co = m.stride[0], m.out_channels
                w1 = m.weight.permute(1, 2, 3, 0).reshape(co * r * r, m.in_channels)  # [(o, i, j), c] = W[c, o, i, j]
                w3 = torch.zeros(co * r * r, m.in_channels, 3, 3, device=w1.device, dtype=w1.dtype)
                w3[:, :, 1, 1] = w1
                chain.append([w3, m.bias.repeat_interleave(r * r) if m.bias is not None else None, 0])
                shuffle = r
            elif isinstance(m, nn.Upsample):
                feat = ops.nearest_up(flush(feat), int(m.scale_factor))
            elif isinstance(m, nn.PixelShuffle):
                shuffle = m.upscale_factor
            i += 1
        return ops.shuffle_rgb(flush(feat), self.out_nc, shuffle)


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
