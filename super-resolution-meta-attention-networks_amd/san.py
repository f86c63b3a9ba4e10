"""SAN / QSAN: second-order attention networks on the HIP kernels.

ref: Code/SISR/models/advanced/SAN_blocks.py (NONLocalBlock2D, SOCA, Nonlocal_CA, RB, LSRAG),
     Code/SISR/models/advanced/mpncov.py (covariance pooling, Newton-Schulz square root),
     Code/SISR/models/advanced/architectures.py:244-312 (SAN),
     Code/SISR/models/attention_manipulators/qsan_blocks.py (QRB, QLSRAG),
     Code/SISR/models/attention_manipulators/architectures.py:402-467 (QSAN),
     handlers + forward_chop: advanced/handlers.py:58-128, attention_manipulators/handlers.py:79-148.

Work split: the RB / QRB residual blocks and every 3x3 conv run on the same fused MFMA operators as RCAN
(> 97 % of the FLOPs); SOCA and the non-local attention have their own kernels (csrc/san.hip).  The projections
around the attention are hand-written too (csrc/nonlocal.hip, ops.nonlocal_block): theta | phi | g as one
[B*H*W, 64] x [64, 24] GEMM on the fp32 matrix cores with both of its gradients, the 2x2 max-pool of phi / g fused
into the gather that lays out the attention rows per quadrant, and the 8 -> 64 output projection + skip as a
streaming kernel.  No library GEMM or torch pooling is left on this path.

Quirks of the reference that are kept because checkpoints and results depend on them:
  * NONLocalBlock2D rebinds its `sub_sample` argument to the nn.Upsample class (SAN_blocks.py:40), so phi and g
    are ALWAYS followed by MaxPool2d(2) (keys `phi.0.*`, `g.0.*`) even though Nonlocal_CA passes sub_sample=False;
  * Nonlocal_CA.soca, SAN.conv_last and LSRAG.gamma are constructed (they consume init RNG and appear in
    checkpoints) but never used in forward;
  * one Nonlocal_CA instance is applied twice per forward (shared weights).
"""
import time

import torch
from torch import nn

from . import architectures as A
from . import ops
from .handlers import BaseModel, QModel


class SOCA(nn.Module):
    """ref: advanced/SAN_blocks.py:244-302"""

    def __init__(self, channel, reduction=8):
        super().__init__()
        self.max_pool = nn.MaxPool2d(kernel_size=2)  # constructed by the reference, never applied
        self.conv_du = nn.Sequential(nn.Conv2d(channel, channel // reduction, 1, padding=0, bias=True),
                                     nn.ReLU(inplace=True),
                                     nn.Conv2d(channel // reduction, channel, 1, padding=0, bias=True), nn.Sigmoid())

    def forward(self, x):
        return ops.soca(x, *A._ca_params(self.conv_du))


class NONLocalBlock2D(nn.Module):
    """Embedded-Gaussian non-local block as the reference actually builds it for dimension 2
    (ref: advanced/SAN_blocks.py:11-148, :235-242): 1x1 projections theta / phi / g to `inter_channels`,
    phi and g max-pooled 2x2, softmax(theta^T phi) g, zero-initialised 1x1 output projection W, + x."""

    def __init__(self, in_channels, inter_channels=None, mode='embedded_gaussian', sub_sample=True, bn_layer=True):
        super().__init__()
        if mode != 'embedded_gaussian' or bn_layer:
            raise NotImplementedError("only the configuration SAN uses (embedded_gaussian, no BatchNorm) is built")
        self.in_channels = in_channels
        self.inter_channels = inter_channels if inter_channels is not None else max(in_channels // 2, 1)
        self.sub_sample = sub_sample
        self.g = nn.Sequential(nn.Conv2d(in_channels, self.inter_channels, 1), nn.MaxPool2d(kernel_size=2))
        self.W = nn.Conv2d(self.inter_channels, in_channels, 1)
        nn.init.constant_(self.W.weight, 0)
        nn.init.constant_(self.W.bias, 0)
        self.theta = nn.Conv2d(in_channels, self.inter_channels, 1)
        self.phi = nn.Sequential(nn.Conv2d(in_channels, self.inter_channels, 1), nn.MaxPool2d(kernel_size=2))

    def forward(self, x):
        """Whole map as one attention domain (Nonlocal_CA applies the block per quadrant instead)."""
        return ops.nonlocal_block(x, self, quadrants=False)


class Nonlocal_CA(nn.Module):
    """ref: advanced/SAN_blocks.py:305-336: the non-local block applied to the four quadrants independently."""

    def __init__(self, in_feat=64, inter_feat=32, reduction=8, sub_sample=False, bn_layer=True):
        super().__init__()
        self.soca = SOCA(in_feat, reduction=reduction)  # never used in forward (reference quirk), holds parameters
        self.non_local = NONLocalBlock2D(in_channels=in_feat, inter_channels=inter_feat, sub_sample=sub_sample,
                                         bn_layer=bn_layer)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        return ops.nonlocal_block(x, self.non_local, quadrants=True)


class RB(nn.Module):
    """ref: advanced/SAN_blocks.py:339-363: conv(relu(conv(x))) + x"""

    def __init__(self, conv, n_feat, kernel_size, reduction, bias=True, bn=False, act=None, res_scale=1, dilation=2):
        super().__init__()
        self.gamma1 = 1.0
        self.conv_first = nn.Sequential(conv(n_feat, n_feat, kernel_size, bias=bias), nn.ReLU(True),
                                        conv(n_feat, n_feat, kernel_size, bias=bias))
        self.res_scale = res_scale

    def forward(self, x):
        c = self.conv_first
        return ops.res_block(x, c[0].weight, c[0].bias, c[2].weight, c[2].bias)


class LSRAG(nn.Module):
    """ref: advanced/SAN_blocks.py:366-412: RB^n -> SOCA -> conv, + x"""

    def __init__(self, conv, n_feat, kernel_size, reduction, act, res_scale, n_resblocks):
        super().__init__()
        self.rcab = nn.ModuleList([RB(conv, n_feat, kernel_size, reduction, bias=True, bn=False, act=None, res_scale=1)
                                   for _ in range(n_resblocks)])
        self.soca = SOCA(n_feat, reduction=reduction)
        self.conv_last = conv(n_feat, n_feat, kernel_size)
        self.n_resblocks = n_resblocks
        self.gamma = nn.Parameter(torch.zeros(1))  # unused by the reference's forward

    def forward(self, x):
        r = x
        for blk in self.rcab:
            r = blk(r)
        return A._conv(self.conv_last, self.soca(r), residual=x)


class QRB(nn.Module):
    """ref: attention_manipulators/qsan_blocks.py:9-34: conv pair -> meta-attention gate (ReLU variant) -> + x"""

    def __init__(self, conv, n_feat, kernel_size, reduction, bias=True, bn=False, act=None, res_scale=1, dilation=2,
                 num_metadata=0):
        super().__init__()
        self.gamma1 = 1.0
        self.conv_first = nn.Sequential(conv(n_feat, n_feat, kernel_size, bias=bias), nn.ReLU(True),
                                        conv(n_feat, n_feat, kernel_size, bias=bias))
        self.res_scale = res_scale
        self.q_layer = A.ParaCALayer(n_feat, num_metadata, nonlinearity=True, num_layers=2)

    def forward(self, x):
        feat, md = x
        c = self.conv_first
        return ops.res_block(feat, c[0].weight, c[0].bias, c[2].weight, c[2].bias, m=self.q_layer.gate(md))


class QLSRAG(nn.Module):
    """ref: attention_manipulators/qsan_blocks.py:37-85"""

    def __init__(self, conv, n_feat, kernel_size, reduction, act, res_scale, n_resblocks, num_metadata=0):
        super().__init__()
        self.rcab = nn.ModuleList([QRB(conv, n_feat, kernel_size, reduction, bias=True, bn=False, act=None, res_scale=1,
                                       num_metadata=num_metadata) for _ in range(n_resblocks)])
        self.soca = SOCA(n_feat, reduction=reduction)
        self.conv_last = conv(n_feat, n_feat, kernel_size)
        self.n_resblocks = n_resblocks
        self.gamma = nn.Parameter(torch.zeros(1))  # unused by the reference's forward

    def forward(self, x):
        feat, md = x
        r = feat
        for blk in self.rcab:
            r = blk((r, md))
        return A._conv(self.conv_last, self.soca(r), residual=feat), md


class SAN(nn.Module):
    """ref: advanced/architectures.py:244-312"""

    def __init__(self, n_resgroups=20, n_resblocks=10, n_feats=64, reduction=16, scale=4, rgb_range=255, n_colors=3,
                 res_scale=1, conv=A.default_conv):
        super().__init__()
        head = [conv(n_colors, n_feats, 3)]  # created first: same RNG consumption order as the reference
        self.gamma = nn.Parameter(torch.zeros(1))
        self.n_resgroups = n_resgroups
        self.RG = nn.ModuleList([LSRAG(conv, n_feats, 3, reduction, act=None, res_scale=res_scale,
                                       n_resblocks=n_resblocks) for _ in range(n_resgroups)])
        self.conv_last = conv(n_feats, n_feats, 3)  # unused by the reference's forward
        tail = [A.Upsampler(conv, scale, n_feats, act=False), conv(n_feats, n_colors, 3)]
        self.non_local = Nonlocal_CA(in_feat=n_feats, inter_feat=n_feats // 8, reduction=8, sub_sample=False,
                                     bn_layer=False)
        self.head = nn.Sequential(*head)
        self.tail = nn.Sequential(*tail)

    def _groups(self, xx, metadata=None):
        shared = xx
        for grp in self.RG:
            out = grp(xx) if metadata is None else grp((xx, metadata))[0]
            xx = ops.scale_add(out, shared, self.gamma)
        return xx

    def forward(self, x):
        A._check_rgb(x, "SAN")
        x = A._conv(self.head[0], x)
        xx = self._groups(self.non_local(x))
        res = ops.add_residual(self.non_local(xx), x)
        return A._conv(self.tail[1], self.tail[0](res))


class QSAN(SAN):
    """ref: attention_manipulators/architectures.py:402-467"""

    def __init__(self, n_resgroups=20, n_resblocks=10, n_feats=64, reduction=16, scale=4, rgb_range=255, n_colors=3,
                 res_scale=1, conv=A.default_conv, input_para=1, **kwargs):
        nn.Module.__init__(self)
        head = [conv(n_colors, n_feats, 3)]
        self.gamma = nn.Parameter(torch.zeros(1))
        self.n_resgroups = n_resgroups
        self.RG = nn.ModuleList([QLSRAG(conv, n_feats, 3, reduction, num_metadata=input_para, act=None,
                                        res_scale=res_scale, n_resblocks=n_resblocks) for _ in range(n_resgroups)])
        self.conv_last = conv(n_feats, n_feats, 3)
        tail = [A.Upsampler(conv, scale, n_feats, act=False), conv(n_feats, n_colors, 3)]
        self.non_local = Nonlocal_CA(in_feat=n_feats, inter_feat=n_feats // 8, reduction=8, sub_sample=False,
                                     bn_layer=False)
        self.head = nn.Sequential(*head)
        self.tail = nn.Sequential(*tail)

    def forward(self, x, metadata):
        A._check_rgb(x, "QSAN")
        x = A._conv(self.head[0], x)
        xx = self._groups(self.non_local(x), metadata)
        res = ops.add_residual(self.non_local(xx), x)
        return A._conv(self.tail[1], self.tail[0](res))


# ----------------------------------------------------------------------------- handlers
def forward_chop(run, x, scale, max_pixels, shave=10):
    """Overlapping 4-way tiling of large evaluation inputs (ref: advanced/handlers.py:80-118): corner tiles of
    (h//2 + shave) x (w//2 + shave), recursion while a tile holds >= max_pixels positions, outputs stitched at the
    half points.  `run` maps an LR tile to its SR tile on the same device."""
    b, c, h, w = x.shape
    hh, wh = h // 2, w // 2
    hs, ws = hh + shave, wh + shave
    tiles = [x[:, :, 0:hs, 0:ws], x[:, :, 0:hs, w - ws:w], x[:, :, h - hs:h, 0:ws], x[:, :, h - hs:h, w - ws:w]]
    if hs * ws < max_pixels:
        sr = [run(t.contiguous()) for t in tiles]
    else:
        sr = [forward_chop(run, t, scale, max_pixels, shave) for t in tiles]
    H, W, hh, wh, hs, ws = scale * h, scale * w, scale * hh, scale * wh, scale * hs, scale * ws
    out = sr[0].new_empty((b, sr[0].shape[1], H, W))
    out[:, :, 0:hh, 0:wh] = sr[0][:, :, 0:hh, 0:wh]
    out[:, :, 0:hh, wh:W] = sr[1][:, :, 0:hh, ws - W + wh:ws]
    out[:, :, hh:H, 0:wh] = sr[2][:, :, hs - H + hh:hs, 0:wh]
    out[:, :, hh:H, wh:W] = sr[3][:, :, hs - H + hh:hs, ws - W + wh:ws]
    return out


class _ChoppedEval:
    """run_eval of the SAN handlers: always through forward_chop (ref: advanced/handlers.py:120-129)."""

    def _chopped_eval(self, x, y, request_loss, timing, keep_on_device, run):
        self.net.eval()
        tic = toc = None
        with torch.no_grad():
            x = x.to(device=self.device)
            if timing:
                torch.cuda.synchronize()
                tic = time.perf_counter()
            out = forward_chop(run, x, self.scale, self.max_combined_im_size)
            if timing:
                torch.cuda.synchronize()
                toc = time.perf_counter()
            loss = None
            if request_loss and y is not None:
                loss = self.criterion(out, y.to(device=self.device)).detach().cpu().numpy()
        out = out.detach() if keep_on_device else out.detach().cpu()
        return out, loss, toc - tic if timing else None


class SANHandler(_ChoppedEval, BaseModel):
    """ref: advanced/handlers.py:58-129.  Architecture parameters are locked like the reference's."""

    def __init__(self, device, model_save_dir, eval_mode=False, lr=1e-4, scale=4, perceptual=None,
                 max_combined_im_size=160000, scheduler=None, scheduler_params=None, **kwargs):
        super().__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode, **kwargs)
        self.net = SAN(scale=scale)
        self.scale = scale
        self.colorspace = 'rgb'
        self.im_input = 'unmodified'
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device)
        self.max_combined_im_size = max_combined_im_size
        self.model_name = 'san'

    def run_eval(self, x, y=None, request_loss=False, metadata=None, metadata_keys=None, timing=False,
                 keep_on_device=False, *args, **kwargs):
        return self._chopped_eval(x, y, request_loss, timing, keep_on_device, lambda t: self.net.forward(t))


class QSANHandler(_ChoppedEval, QModel):
    """ref: attention_manipulators/handlers.py:79-148"""

    def __init__(self, device, model_save_dir, eval_mode=False, lr=1e-4, scale=4, perceptual=None,
                 max_combined_im_size=160000, scheduler=None, scheduler_params=None, **kwargs):
        super().__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode, **kwargs)
        self.net = QSAN(scale=scale, input_para=self.num_metadata)
        self.scale = scale
        self.colorspace = 'rgb'
        self.im_input = 'unmodified'
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device)
        self.max_combined_im_size = max_combined_im_size
        self.model_name = 'qsan'

    def run_eval(self, x, y=None, request_loss=False, metadata=None, metadata_keys=None, timing=False,
                 keep_on_device=False, extra_channels=None, *args, **kwargs):
        if extra_channels is None:
            extra_channels = self.generate_channels(x, metadata, metadata_keys)
        extra_channels = extra_channels.to(self.device)
        return self._chopped_eval(x, y, request_loss, timing, keep_on_device,
                                  lambda t: self.net.forward(t, metadata=extra_channels))
