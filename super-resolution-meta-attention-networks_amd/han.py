"""HAN / QHAN: holistic (layer + channel-spatial) attention networks on the HIP kernels.

ref: Code/SISR/models/advanced/HAN_blocks.py (LAM_Module, CSAM_Module),
     Code/SISR/models/advanced/architectures.py:314-377 (HAN),
     Code/SISR/models/attention_manipulators/architectures.py:470-540 (QHAN),
     handlers: advanced/handlers.py:42-55, attention_manipulators/handlers.py:156-171.

The reference collects the 11 intermediate maps newest-first with torch.cat and concatenates CSAM / LAM
branches along channels.  Here the maps are copied once into a [B][N][H][W][64] stack (ops.stack_maps, a
HIP gather of channels-last maps) and every consumer -- the LAM kernels and the 704->64 / 128->64 convolutions -- reads
the stack as 64-channel chunks, so no channel concatenation is materialised.
"""
import torch
from torch import nn

from . import architectures as A
from . import ops
from .handlers import BaseModel, QModel


class LAM_Module(nn.Module):
    """ref: advanced/HAN_blocks.py:7-37.  Input (B,N,C,H,W) logical; returns (B,N*C,H,W) logical."""

    def __init__(self, in_dim):
        super().__init__()
        self.chanel_in = in_dim
        self.gamma = nn.Parameter(torch.zeros(1))
        self.softmax = nn.Softmax(dim=-1)

    def forward_stack(self, stack):
        return ops.lam(stack, self.gamma)

    def forward(self, x):
        B, N, C, H, W = x.shape
        out = ops.lam(x.contiguous().reshape(B, N, C * H * W, 1, 1), self.gamma)  # layout-agnostic
        return out.reshape(B, N * C, H, W)


class CSAM_Module(nn.Module):
    """ref: advanced/HAN_blocks.py:40-76"""

    def __init__(self, in_dim):
        super().__init__()
        self.chanel_in = in_dim
        self.conv = nn.Conv3d(1, 1, 3, 1, 1)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        return ops.csam(x, self.conv.weight, self.conv.bias, self.gamma)


def _han_tail(net, x_head, maps):
    """Shared HAN/QHAN epilogue; ``maps`` oldest-first (group outputs + post-body conv)."""
    stack = ops.stack_maps(maps[::-1])                    # newest first (ref :359-362)
    out2 = ops.conv3x3_stack(net.la.forward_stack(stack), net.last_conv.weight, net.last_conv.bias)
    out1 = net.csa(maps[-1])
    pair = ops.stack_maps([out1, out2])                   # torch.cat([out1, out2], 1) as two chunks
    res = ops.conv3x3_stack(pair, net.last.weight, net.last.bias, residual=x_head)  # + x fused into the conv's store
    return A._conv(net.tail[1], net.tail[0](res))


class HAN(A.ChannelPadded, nn.Module):
    """ref: advanced/architectures.py:314-377"""

    def __init__(self, n_resgroups=10, n_resblocks=20, n_feats=64, reduction=16, scale=4, n_colors=3, res_scale=1.0,
                 conv=A.default_conv):
        super().__init__()
        act = nn.ReLU(True)
        head = [conv(n_colors, n_feats, 3)]
        body = [A.ResidualGroup(conv, n_feats, 3, reduction, act=act, res_scale=res_scale, n_resblocks=n_resblocks)
                for _ in range(n_resgroups)]
        body.append(conv(n_feats, n_feats, 3))
        tail = [A.Upsampler(conv, scale, n_feats, act=False), conv(n_feats, n_colors, 3)]
        self.head = nn.Sequential(*head)
        self.body = nn.Sequential(*body)
        self.csa = CSAM_Module(n_feats)
        self.la = LAM_Module(n_feats)
        self.last_conv = nn.Conv2d(n_feats * 11, n_feats, 3, 1, 1)
        self.last = nn.Conv2d(n_feats * 2, n_feats, 3, 1, 1)
        self.tail = nn.Sequential(*tail)
        self._init_padding(n_feats, lambda P: HAN(n_resgroups, n_resblocks, P, reduction, scale, n_colors, res_scale, conv))

    def forward(self, x):
        A._check_rgb(x, "HAN")
        if self.padded():
            return self._run_padded(x)
        x = A._conv(self.head[0], x)
        res, maps = x, []
        mods = list(self.body)
        for g in mods[:-1]:
            res = g(res)
            maps.append(res)
        res = A._conv(mods[-1], res)
        maps.append(res)
        return _han_tail(self, x, maps)


class QHAN(A.ChannelPadded, nn.Module):
    """ref: attention_manipulators/architectures.py:470-540"""

    def __init__(self, n_resgroups=10, n_resblocks=20, n_feats=64, reduction=16, num_metadata=0, scale=4, n_colors=3,
                 res_scale=1.0, conv=A.default_conv, num_q_layers_inner_residual=None):
        super().__init__()
        act = nn.ReLU(True)
        head = [conv(n_colors, n_feats, 3)]
        body = [A.QResidualGroup(conv, n_feats, 3, reduction, act=act, res_scale=res_scale, style='standard',
                                 num_metadata=num_metadata, pa=False, q_layer=True, n_resblocks=n_resblocks,
                                 num_q_layers=num_q_layers_inner_residual) for _ in range(n_resgroups)]
        body.append(conv(n_feats, n_feats, 3))
        tail = [A.Upsampler(conv, scale, n_feats, act=False), conv(n_feats, n_colors, 3)]
        self.head = nn.Sequential(*head)
        self.body = nn.Sequential(*body)
        self.csa = CSAM_Module(n_feats)
        self.la = LAM_Module(n_feats)
        self.last_conv = nn.Conv2d(n_feats * 11, n_feats, 3, 1, 1)
        self.last = nn.Conv2d(n_feats * 2, n_feats, 3, 1, 1)
        self.tail = nn.Sequential(*tail)
        self._init_padding(n_feats, lambda P: QHAN(n_resgroups, n_resblocks, P, reduction, num_metadata, scale, n_colors,
                                                   res_scale, conv, num_q_layers_inner_residual))

    def forward(self, x, metadata):
        A._check_rgb(x, "QHAN")
        if self.padded():
            return self._run_padded(x, metadata)
        x = A._conv(self.head[0], x)
        res, maps = x, []
        mods = list(self.body)
        qblocks = [b for g in mods[:-1] for b in g.body if b.q_layer]
        gates = dict(zip(map(id, qblocks), A.meta_gates([b.q_node for b in qblocks], metadata)))
        for g in mods[:-1]:
            res, _ = g((res, metadata), gates)
            maps.append(res)
        res = A._conv(mods[-1], res)
        maps.append(res)
        return _han_tail(self, x, maps)

    def meta_gate_layers(self):
        """The layers forward() hands to meta_gates() in one batch (architectures.late_parameters)."""
        return [b.q_node for g in list(self.body)[:-1] for b in g.body if b.q_layer]


class HANHandler(BaseModel):
    def __init__(self, device, model_save_dir, eval_mode=False, lr=1e-4, scale=4, perceptual=None, scheduler=None,
                 scheduler_params=None, **kwargs):
        super().__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode, **kwargs)
        self.net = HAN(scale=scale)
        self.colorspace = 'rgb'
        self.im_input = 'unmodified'
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device)
        self.model_name = 'han'


class QHANHandler(QModel):
    def __init__(self, device, model_save_dir, eval_mode=False, lr=1e-4, scale=4, perceptual=None, scheduler=None,
                 scheduler_params=None, **kwargs):
        super().__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode, **kwargs)
        self.net = QHAN(scale=scale, num_metadata=self.num_metadata)
        self.colorspace = 'rgb'
        self.im_input = 'unmodified'
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device)
        self.model_name = 'qhan'
