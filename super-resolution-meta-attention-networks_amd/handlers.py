"""Model-handler / operator API of the reference, kept intact over the HIP networks.

Mirrors (names, arguments, return values, error behaviour):
  registry + ModelInterface   ref: Code/SISR/models/__init__.py:20-254
  BaseModel                   ref: Code/SISR/models/__init__.py:257-575
  QModel                      ref: Code/SISR/models/attention_manipulators/__init__.py:6-118
  EDSR/RCAN/HAN handlers      ref: Code/SISR/models/advanced/handlers.py:7-55
  QRCAN/QEDSR/QHAN handlers   ref: Code/SISR/models/attention_manipulators/handlers.py:7-76,156-171
so TrainingHandler / EvalHub style callers (train_batch, net_run_and_process, save, ...) work unchanged
and checkpoints ({'network','optimizer','model_name','model_epoch'[,'scheduler_G']}) interchange with the
reference.  The only intentional difference: gpu='multi' means one process per GPU + RCCL gradient
all-reduce (parallel.py) instead of single-process nn.DataParallel (ref :344-347).
"""
import glob
import math
import os
import time
from collections import OrderedDict

import numpy as np
import torch
from torch import nn, optim

from . import architectures as A
from . import metrics, ops


class L1Loss(nn.Module):
    """nn.L1Loss() stand-in running the HIP loss kernel (ref: models/__init__.py:268)."""

    def forward(self, out, target):
        return ops.l1_loss(out, target)


def create_dir_if_empty(*directories):
    for d in directories:
        os.makedirs(d, exist_ok=True)  # exist_ok: every rank of a data-parallel run builds the interface


class BaseModel(nn.Module):
    def __init__(self, device, model_save_dir, eval_mode, grad_clip=None, **kwargs):
        super().__init__()
        self.criterion = L1Loss()
        self.device = torch.device('cpu') if device == 'cpu' else device
        self.optimizer = None
        self.net = None
        self.face_finder = False
        self.model_name = None
        self.im_input = None
        self.colorspace = None
        self.grad_clip = None if grad_clip == 0 else grad_clip
        self.model_save_dir = model_save_dir
        self.eval_mode = eval_mode
        self.curr_epoch = 0
        self.state = {}
        self.learning_rate_scheduler = None
        self.legacy_load = True
        self.reducer = None  # parallel.GradReducer when running data-parallel
        # hipGraph replay of forward+loss+backward for fixed-shape training batches (launch-bound regime: at
        # <= 4 tiles per GPU the ~3700 launches of an RCAN step cost more host time than GPU time)
        self.use_graph = os.environ.get("SISR_GRAPH", "0") == "1"
        self._graphs = {}

    # -- optimiser / scheduler (ref :292-335)
    def define_optimizer(self, lr=1e-4, optimizer_params=None):
        params = [p for p in self.net.parameters() if p.requires_grad]
        betas = (optimizer_params['beta_1'], optimizer_params['beta_2']) if optimizer_params is not None else (0.9, 0.999)
        if params and all(p.is_cuda for p in params) and os.environ.get("SISR_FLAT_ADAM", "1") != "0":
            from .optim import FlatAdam  # one-launch Adam over a flat arena, torch.optim.Adam's schema (optim.py)
            # the meta-attention layers a network runs as ONE launch deliver their gradients at the very end of backward
            # (architectures.meta_gates); layers applied block by block (QSPARNet, the metadata-mixing styles) do not
            self.optimizer = FlatAdam(params, lr=lr, betas=betas, late=A.late_parameters(self.net))
        else:  # CPU handlers exist for construction / checkpoint plumbing only and are never stepped
            self.optimizer = optim.Adam(params, lr=lr, betas=betas)

    def define_scheduler(self, scheduler, scheduler_params):
        if scheduler == 'cosine_annealing_warm_restarts':
            self.learning_rate_scheduler = optim.lr_scheduler.CosineAnnealingWarmRestarts(
                self.optimizer, T_mult=scheduler_params['t_mult'], T_0=scheduler_params['restart_period'],
                eta_min=scheduler_params['lr_min'])
        elif scheduler == 'multi_step_lr':
            self.learning_rate_scheduler = optim.lr_scheduler.MultiStepLR(
                self.optimizer, milestones=scheduler_params['milestones'], gamma=scheduler_params['gamma'])
        elif scheduler == 'custom_dasr':
            def dasr(epoch):
                if epoch < 60:
                    return 1e-3
                if epoch < 225:
                    return 1e-4
                return 1e-4 * math.pow(0.5, (epoch - 100) // 125)
            self.learning_rate_scheduler = optim.lr_scheduler.LambdaLR(self.optimizer, lr_lambda=dasr)
        elif scheduler == 'step_lr':
            self.learning_rate_scheduler = optim.lr_scheduler.StepLR(
                self.optimizer, step_size=scheduler_params['step_size'], gamma=scheduler_params['gamma'])
        else:
            raise RuntimeError('%s scheduler not implemented' % scheduler)

    def activate_device(self):
        self.net.to(self.device)

    def training_setup(self, lr, scheduler, scheduler_params, perceptual, device, optimizer_params=None):
        if not self.eval_mode:
            self.define_optimizer(lr=lr, optimizer_params=optimizer_params)
            if scheduler is not None:
                self.define_scheduler(scheduler=scheduler, scheduler_params=scheduler_params)
        if perceptual is not None and self.eval_mode is False:
            raise NotImplementedError('perceptual (VGG) loss is outside the HIP hot path; every sample config '
                                      'of the reference uses perceptual=None')

    def set_multi_gpu(self, device_ids=None, bucket_mb=None):
        """One process per GPU: gradients are averaged over the torch.distributed (RCCL) world.  bucket_mb: size of the
        all-reduce buckets (default 8, or SISR_DP_BUCKET_MB)."""
        from .parallel import GradReducer
        self.remove_multi_gpu()  # graphs captured with another reducer signal ITS progress words and write into its buckets
        self._graphs = {}
        self.reducer = GradReducer(self.net, bucket_mb=bucket_mb, arena=getattr(self.optimizer, 'grad_views', None),
                                   arena_flat=getattr(self.optimizer, 'flat_g', None),
                                   arena_offsets=getattr(self.optimizer, 'offsets', None),
                                   arena_order=getattr(self.optimizer, 'arena_order', None))

    def remove_multi_gpu(self):
        """Back to a single-process handler: the captured graphs go first (their signal nodes hold raw pointers to the reducer's
        pinned progress words), then the reducer's hooks, words and gradient sinks."""
        if self.reducer is None:
            return
        if self._graphs:
            torch.cuda.synchronize()
            self._graphs = {}
        red, self.reducer = self.reducer, None
        red.remove()
        if getattr(self.optimizer, 'grad_views', None) is not None:  # the optimiser's arena stays the gradients' home
            for p, view in self.optimizer.grad_views.items():
                ops.GRAD_SINK[p.data_ptr()] = view

    # -- checkpoints (ref :349-464)
    def save_model(self, model_save_name, model_idx, extract_state_only=False):
        self.state['network'] = self.net.state_dict()
        self.state['optimizer'] = self.optimizer.state_dict()
        self.state['model_name'] = self.model_name
        self.state['model_epoch'] = self.curr_epoch
        if self.learning_rate_scheduler is not None:
            self.state['scheduler_G'] = self.learning_rate_scheduler.state_dict()
        if extract_state_only:
            return self.state
        torch.save(self.state, f=os.path.join(self.model_save_dir, "{}_{}".format(model_save_name, str(model_idx))))

    @staticmethod
    def legacy_switch(state_dict):
        out = OrderedDict()
        for k, v in state_dict.items():
            if k[:13] == 'model.module.':
                out[k[13:]] = v
            elif k[:6] == 'model.':
                out[k[6:]] = v
            else:
                out[k] = v
        return out

    def load_model(self, model_save_name, model_idx, legacy=False, load_override=None, preloaded_state=None):
        loc = self.device if self.device == torch.device('cpu') else "cuda:%d" % self.device
        folder = self.model_save_dir if load_override is None else load_override
        load_file = os.path.join(folder, "{}_{}".format(model_save_name, str(model_idx)))
        if preloaded_state is None:
            # checkpoints are plain tensor/number dicts: the safe loader is enough
            state = torch.load(f=load_file, map_location=loc, weights_only=True)
        else:
            state = preloaded_state
        net_state = self.legacy_switch(state['network']) if legacy else state['network']
        self.net.load_state_dict(state_dict=net_state)
        if not self.eval_mode:
            self.optimizer.load_state_dict(state['optimizer'])
            if self.learning_rate_scheduler is not None:
                self.learning_rate_scheduler.load_state_dict(state['scheduler_G'])
        self.set_epoch(state['model_epoch'])
        if state['model_name'] == 'qpircan':
            state['model_name'] = 'qrcan'
        print('Loaded model uses the following architecture:', state['model_name'])
        return state

    # -- train / eval steps (ref :466-533)
    def train_step(self, x, y, tag=None, loss_scale=1.0, **kwargs):
        """run_train without its host round trips: returns (loss, out) as device tensors and never
        synchronises, so consecutive steps queue back to back on the stream.
        loss_scale: weight of this rank's mean loss in the data-parallel average (parallel.shard_batch sets it for
        ragged batches); 0 with an empty shard = take part in the gradient exchange and the update with zero gradients."""
        if self.eval_mode:
            raise RuntimeError('Model initialized in eval mode, training not possible.')
        self.net.train()
        x, y = x.to(device=self.device), y.to(device=self.device)
        if x.shape[0] == 0:
            self.optimizer.zero_grad()
            self._finish_update()
            nan = torch.full((), float('nan'), device=x.device)
            return nan, x.new_zeros((0,) + tuple(y.shape[1:]))
        if self.use_graph and x.is_cuda and loss_scale == 1.0:
            return self._graphed_step(x, y, kwargs)
        try:
            ops.pack_all(self.net, A.conv_weights)  # every conv weight repacked by one launch
            out = self.run_model(x, image_names=tag, **kwargs)
            loss = self.criterion(out, y)
            scale = loss_scale * self._dp_scale()
            self.standard_update(loss if scale == 1.0 else loss * scale)
        finally:
            ops.invalidate_packs()  # the optimiser moved the weights
        return loss.detach(), out.detach()

    def _dp_scale(self):
        """1 / world with a reducer: backward then produces this rank's share of the AVERAGE gradient, the all-reduce sums the
        shares, and nothing is left to scale between the exchange and the optimiser (GradReducer.reduce(prescaled=True))."""
        return 1.0 / self.reducer.world if self.reducer is not None else 1.0

    def _graphed_step(self, x, y, kwargs):
        """Forward + loss + backward captured once per batch shape into a hipGraph and replayed; the reducer,
        gradient clipping, Adam and the scheduler stay eager (they are a handful of launches).

        A replay writes gradients into the tensors that were the parameters' .grad at capture time, so every
        entry keeps those tensors and re-binds p.grad to them after each replay: the data-parallel join (which
        hands out bucket views as .grad) and a capture for another batch shape both re-point p.grad in between.
        With a GradReducer the captured weight-gradient kernels write straight into its buckets (ops.GRAD_SINK).  A
        replay fires no hooks; instead the capture places a signal node behind every bucket's last gradient kernel and the
        host issues each all-reduce as soon as the replay reports the bucket complete (GradReducer.launch_signalled), so the
        exchange overlaps the rest of the replayed backward as it does in eager mode.  SISR_GRAPH_OVERLAP=0 (or a CPU
        reducer): all buckets are all-reduced at the join."""
        extra = kwargs.get('extra_channels')
        key = (tuple(x.shape), tuple(y.shape), None if extra is None else tuple(extra.shape))
        entry = self._graphs.get(key)
        red = self.reducer
        signal = red is not None and red.can_signal()
        if red is not None:
            red.hooks_enabled = False
        try:
            if entry is None:
                sx, sy = x.clone(), y.clone()
                se = None if extra is None else extra.to(self.device).clone()
                kw = dict(kwargs)
                if se is not None:
                    kw['extra_channels'] = se
                # The warm-up pass below is a real forward: a network with batch norms (SPARNet) would advance its running
                # statistics and step counters once more than the eager path does for this batch.  Buffers are put back after it.
                saved_buffers = [(b, b.detach().clone()) for b in self.net.buffers()]
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):  # one eager pass off the default stream: allocator + lazy-init warm-up
                    self.optimizer.zero_grad(set_to_none=True)
                    ops.pack_all(self.net, A.conv_weights)
                    with ops.deferred_wgrads():
                        (self.criterion(self.run_model(sx, **kw), sy) * self._dp_scale()).backward()
                torch.cuda.current_stream().wait_stream(side)
                self.optimizer.zero_grad(set_to_none=True)
                with torch.no_grad():
                    for b, keep in saved_buffers:
                        b.copy_(keep)
                graph = torch.cuda.CUDAGraph()
                dump = os.environ.get("SISR_GRAPH_DUMP")  # diagnostic: write the captured graph as a DOT file
                if dump:
                    graph.enable_debug_mode()
                order = []
                with torch.cuda.graph(graph):
                    ops.pack_all(self.net, A.conv_weights)  # first node of the graph: the replay repacks
                    out = self.run_model(sx, **kw)
                    loss = self.criterion(out, sy)
                    if signal:
                        red.begin_capture()
                    try:
                        # with signal nodes a gradient must be launched before its parameter's hook runs: nothing is deferred
                        # past the autograd node that produces it (the group nodes flush their own batches before they return)
                        with ops.deferred_wgrads(enabled=not signal):
                            (loss if red is None or red.world == 1 else loss * self._dp_scale()).backward()
                    finally:
                        if signal:
                            order = red.end_capture()
                if dump:
                    graph.debug_dump(dump)
                grads = [(p, p.grad) for p in self.net.parameters()]
                entry = (graph, sx, sy, se, loss, out, grads, order)
                self._graphs[key] = entry
            graph, sx, sy, se, loss, out, grads, order = entry
            sx.copy_(x)
            sy.copy_(y)
            if se is not None:
                se.copy_(extra)
            graph.replay()
            for p, g in grads:  # this replay's gradients live in this entry's capture-time tensors
                p.grad = g
            if signal and order:
                red.launch_signalled(order)
        finally:
            ops.invalidate_packs()
            if red is not None:
                red.hooks_enabled = True
        if red is not None:
            red.reduce(prescaled=True)
        if self.grad_clip is not None:
            nn.utils.clip_grad_norm_(self.net.parameters(), self.grad_clip)
        self.optimizer.step()
        if self.learning_rate_scheduler is not None:
            self.learning_rate_scheduler.step()
        return loss.detach().clone(), out.detach()

    def run_train(self, x, y, tag=None, mask=None, keep_on_device=False, *args, **kwargs):
        loss, out = self.train_step(x, y, tag=tag, **kwargs)
        if keep_on_device:
            return loss.cpu().numpy(), out
        return loss.cpu().numpy(), out.cpu()

    def standard_update(self, loss):
        self.optimizer.zero_grad()
        with ops.deferred_wgrads(enabled=self.reducer is None):  # with a reducer its hooks read gradients mid-backward
            loss.backward()
        self._finish_update()

    def _finish_update(self):
        if self.reducer is not None:
            self.reducer.reduce(prescaled=True)
        if self.grad_clip is not None:
            nn.utils.clip_grad_norm_(self.net.parameters(), self.grad_clip)
        self.optimizer.step()
        if self.learning_rate_scheduler is not None:
            self.learning_rate_scheduler.step()

    def run_eval(self, x, y=None, request_loss=False, tag=None, timing=False, keep_on_device=False, *args, **kwargs):
        self.net.eval()
        tic = toc = None
        with torch.no_grad():
            x = x.to(device=self.device)
            if timing:
                torch.cuda.synchronize()  # the reference times without a device sync (ref :508-512); we do sync
                tic = time.perf_counter()
            try:
                if x.is_cuda:
                    ops.pack_all(self.net, A.conv_weights)
                out = self.run_model(x, image_names=tag, **kwargs)
            finally:
                ops.invalidate_packs()
            if timing:
                torch.cuda.synchronize()
                toc = time.perf_counter()
            if request_loss and y is not None:
                loss = self.criterion(out, y.to(device=self.device)).detach().cpu().numpy()
            else:
                loss = None
        if keep_on_device:
            return out.detach(), loss, toc - tic if timing else None
        return out.detach().cpu(), loss, toc - tic if timing else None

    def run_model(self, x, *args, **kwargs):
        return self.net.forward(x)

    def print_parameters(self, verbose=False):
        total = 0
        for name, value in self.named_parameters():
            if verbose:
                print(name, value.shape)
            total += np.prod(value.shape)
        return total

    def epoch_end_calls(self):
        pass

    def set_epoch(self, epoch):
        self.curr_epoch = epoch

    def get_learning_rate(self):
        return self.optimizer.param_groups[0]['lr']

    def extra_diagnostics(self):
        pass

    def pre_training_model_load(self):
        pass


class QModel(BaseModel):
    """Metadata plumbing for the meta-attention networks (host side, stays Python)."""

    def __init__(self, metadata=None, **kwargs):
        self.style = None
        self.channel_concat = False
        if metadata is not None:
            self.num_metadata = len(metadata)
            if 'all' in metadata:
                self.num_metadata += 39
            if 'blur_kernel' in metadata:
                self.num_metadata += 9
            elif 'unmodified_blur_kernel' in metadata:
                self.num_metadata += 440
            self.metadata = metadata
        else:
            self.metadata = ['qpi']
            self.num_metadata = 1
        super().__init__(**kwargs)

    def generate_channels(self, x, metadata, keys):
        """collated (B,M) metadata + list[M] of B-tuples of keys -> fp32 (B,num_metadata,1,1) on the host."""
        if metadata is None:
            raise RuntimeError('Metadata needs to be specified for this network to run properly.')
        if 'all' in self.metadata:
            mask = [True] * self.num_metadata
        else:
            mask = [key[0] in self.metadata for key in keys]
        md = torch.as_tensor(np.asarray(metadata)) if not torch.is_tensor(metadata) else metadata
        rows = []
        for index in range(x.size(0)):
            row = md[index] if len(keys) == 1 else md[index][mask]
            rows.append((torch.ones(self.num_metadata) * row).to(torch.float32))
        extra = torch.stack(rows)[:, :, None, None]
        if self.style == 'modulate':
            extra = self.scale_qpi(extra)
        return extra

    def channel_concat_logic(self, x, extra_channels, metadata, metadata_keys):
        if extra_channels is None:
            extra_channels = self.generate_channels(x, metadata, metadata_keys)
            if not self.channel_concat and self.device != extra_channels.device:
                extra_channels = extra_channels.to(self.device)
        input_data = torch.cat((x, extra_channels), 1) if self.channel_concat else x
        return input_data, extra_channels

    def run_train(self, x, y, metadata=None, extra_channels=None, metadata_keys=None, *args, **kwargs):
        input_data, extra_channels = self.channel_concat_logic(x, extra_channels, metadata, metadata_keys)
        return super().run_train(input_data, y, extra_channels=extra_channels, **kwargs)

    def train_step(self, x, y, metadata=None, extra_channels=None, metadata_keys=None, **kwargs):
        if extra_channels is None and metadata is None:
            raise RuntimeError('Metadata needs to be specified for this network to run properly.')
        if extra_channels is None:  # (callers that pass ready-made channels have also prepared the input: run_train above)
            x, extra_channels = self.channel_concat_logic(x, None, metadata, metadata_keys)
        return super().train_step(x, y, extra_channels=extra_channels, **kwargs)

    def run_eval(self, x, y=None, request_loss=False, metadata=None, metadata_keys=None, extra_channels=None, *args,
                 **kwargs):
        input_data, extra_channels = self.channel_concat_logic(x, extra_channels, metadata, metadata_keys)
        return super().run_eval(input_data, y, request_loss=request_loss, extra_channels=extra_channels, **kwargs)

    def run_model(self, x, extra_channels=None, *args, **kwargs):
        return self.net.forward(x, metadata=extra_channels)


# ----------------------------------------------------------------------------- handlers
class EDSRHandler(BaseModel):
    def __init__(self, device, model_save_dir, eval_mode=False, lr=1e-4, scale=4, in_features=3, hr_data_loc=None,
                 scheduler=None, scheduler_params=None, perceptual=None, num_features=64, num_blocks=16,
                 res_scale=0.1, **kwargs):
        super().__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode, hr_data_loc=hr_data_loc,
                         **kwargs)
        self.net = A.EDSR(scale=scale, in_features=in_features, net_features=num_features, num_blocks=num_blocks,
                          res_scale=res_scale)
        self.colorspace = 'rgb'
        self.im_input = 'unmodified'
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device)
        self.model_name = 'edsr'


class RCANHandler(BaseModel):
    def __init__(self, device, model_save_dir, eval_mode=False, lr=1e-4, scale=4, in_features=3, perceptual=None,
                 scheduler=None, scheduler_params=None, **kwargs):
        super().__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode, **kwargs)
        self.net = A.RCAN(scale=scale, in_feats=in_features)
        self.colorspace = 'rgb'
        self.im_input = 'unmodified'
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device)
        self.model_name = 'rcan'


class QRCANHandler(QModel):
    def __init__(self, device, model_save_dir, eval_mode=False, lr=1e-4, scale=4, in_features=3, scheduler=None,
                 scheduler_params=None, style='modulate', perceptual=None, clamp=False, min_mu=-0.2, max_mu=0.8,
                 n_feats=64, **kwargs):
        super().__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode, **kwargs)
        self.net = A.QRCAN(scale=scale, in_feats=in_features, num_metadata=self.num_metadata, n_feats=n_feats,
                           style=style, **kwargs)
        self.colorspace = 'augmented_rgb'
        self.im_input = 'unmodified'
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device)
        self.model_name = 'qrcan'
        self.min_mu = min_mu
        self.max_mu = max_mu
        self.base_scaler = np.linspace(0, 1, n_feats)
        self.clamp = clamp
        self.style = style

    @staticmethod
    def gaussian(x, mu, sig=0.2):
        return torch.from_numpy((1 / (np.sqrt(2 * np.pi) * sig)) *
                                np.exp(-np.power(x - mu, 2.) / (2 * np.power(sig, 2.)))).type(torch.float32)

    def scale_qpi(self, qpi):
        scaled = (qpi * (self.max_mu - self.min_mu)) + self.min_mu
        full = torch.stack([self.gaussian(self.base_scaler, scaled[i].squeeze().numpy())
                            for i in range(scaled.size(0))])
        if self.clamp:
            full = torch.clamp(full, 0, 1)
        return full.unsqueeze(2).unsqueeze(3)


class QEDSRHandler(QModel):
    def __init__(self, device, model_save_dir, eval_mode=False, lr=1e-4, scale=4, in_features=3, num_blocks=16,
                 num_features=64, res_scale=0.1, scheduler=None, scheduler_params=None, perceptual=None, **kwargs):
        super().__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode, **kwargs)
        self.net = A.QEDSR(scale=scale, in_features=in_features, num_features=num_features, num_blocks=num_blocks,
                           res_scale=res_scale, input_para=self.num_metadata, **kwargs)
        self.colorspace = 'augmented_rgb'
        self.im_input = 'unmodified'
        self.activate_device()
        self.model_name = 'qedsr'
        self.criterion = L1Loss()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device)


HANDLERS = [EDSRHandler, RCANHandler, QRCANHandler, QEDSRHandler]
try:  # HAN / QHAN are registered once their attention kernels are built (han.py)
    from .han import HANHandler, QHANHandler
    HANDLERS += [HANHandler, QHANHandler]
except ImportError:
    pass
try:  # SAN / QSAN (san.py)
    from .san import SANHandler, QSANHandler
    HANDLERS += [SANHandler, QSANHandler]
except ImportError:
    pass
try:  # SRMD (srmd.py)
    from .srmd import SRMDHandler
    HANDLERS += [SRMDHandler]
except ImportError:
    pass
try:  # SFTMD (sftmd.py)
    from .sftmd import SFTMDHandler
    HANDLERS += [SFTMDHandler]
except ImportError:
    pass
try:  # SPARNet / QSPARNet (sparnet.py)
    from .sparnet import SPARNetHandler, QSPARNetHandler
    HANDLERS += [SPARNetHandler, QSPARNetHandler]
except ImportError:
    pass
# registry key = class name minus 'Handler', lower-cased (ref: models/__init__.py:26-30)
available_models = {h.__name__.split('Handler')[0].lower(): h for h in HANDLERS}


# ----------------------------------------------------------------------------- client façade
class ModelInterface:
    """ref: Code/SISR/models/__init__.py:33-254"""

    def __init__(self, model_loc, experiment, gpu='off', sp_gpu=0, mode='eval', new_params=None, load_epoch=None,
                 scale=None, save_subdir=None, new_branch=False):
        """Same contract as the reference's constructor: experiment folder layout (<model_loc>/<experiment>/{result_outputs,
        saved_models}[/<save_subdir>]), parameters from `new_params` (new model) or the folder's config.toml (load_epoch given,
        a number or 'best' / 'last'), the same refusals with the same messages."""
        self.experiment, self.mode = experiment, mode
        resuming = load_epoch is not None
        self._lay_out(model_loc, experiment, save_subdir, create=mode == 'train')
        if mode == 'train' and new_params is None and not resuming:
            raise RuntimeError('Need to specify model parameters to train a new model.')
        if mode == 'eval' and not resuming:
            raise RuntimeError('Need to specify which model epoch to load.')
        self.metadata = self._stored_parameters() if resuming else new_params
        self.name = {'qpircan': 'qrcan'}.get(self.metadata['name'], self.metadata['name'])  # (a legacy spelling)
        if scale not in (None, self.metadata['internal_params']['scale']):
            raise Exception('The model loaded has been trained for a different scale, '
                            'and cannot produce the requested images.')
        self.device = sp_gpu if (gpu != 'off' and torch.cuda.is_available()) else torch.device('cpu')
        self.model = self.define_model(name=self.name, model_save_dir=self.saved_models, device=self.device,
                                       eval_mode=mode == 'eval', **self.metadata['internal_params'])
        self.model_epoch = self._epoch_number(load_epoch) if resuming else 0
        if resuming:
            self.model.load_model(model_save_name='train_model', model_idx=self.model_epoch, legacy=self.model.legacy_load,
                                  load_override=os.path.dirname(self.saved_models) if new_branch else None)
        else:
            self.model.pre_training_model_load()
        self.full_name = '%s_%d' % (experiment, self.model_epoch)
        if gpu == 'multi':
            self.model.set_multi_gpu()
        self.configuration = {'input': self.model.im_input, 'colorspace': self.model.colorspace}
        self.print_overview()

    def _lay_out(self, model_loc, experiment, save_subdir, create):
        sub = (save_subdir,) if save_subdir is not None else ()
        self.base_folder = os.path.abspath(os.path.join(model_loc, experiment))
        self.logs = os.path.join(self.base_folder, 'result_outputs', *sub)
        self.saved_models = os.path.join(self.base_folder, 'saved_models', *sub)
        if create:
            create_dir_if_empty(self.base_folder, self.logs, self.saved_models)

    def _stored_parameters(self):
        if not glob.glob(os.path.join(self.base_folder, '*.toml')):
            raise RuntimeError('No config.toml in %s - model structure unknown.' % self.base_folder)
        import tomli
        with open(os.path.join(self.base_folder, 'config.toml'), 'rb') as f:
            return tomli.load(f)['model']

    def _epoch_number(self, load_epoch):
        if load_epoch not in ('best', 'last'):
            return load_epoch
        import pandas as pd
        psnr = pd.read_csv(os.path.join(self.logs, 'summary.csv'))['val-PSNR']
        return psnr.idxmax() if load_epoch == 'best' else len(psnr) - 1

    def train_batch(self, lr, hr, **kwargs):
        return self.model.run_train(x=lr, y=hr, **kwargs)

    def set_epoch(self, epoch):
        self.model_epoch = epoch
        self.model.set_epoch(epoch)

    def net_run_and_process(self, lr=None, hr=None, **kwargs):
        if 'rgb' not in self.configuration['colorspace']:
            raise NotImplementedError('Y-channel-only models (SRCNN/VDSR) are out of scope of the HIP path')
        out_rgb, loss, timing = self.model.run_eval(x=lr, y=hr, **kwargs)
        out_ycbcr = self.colorspace_convert(out_rgb, colorspace='rgb')
        out_rgb = self._standard_image_formatting(out_rgb.numpy())
        return out_rgb, out_ycbcr, loss, timing

    @staticmethod
    def colorspace_convert(image, colorspace='rgb'):
        if colorspace != 'rgb':
            raise NotImplementedError(colorspace)
        return metrics.batch_rgb_to_ycbcr(image.numpy())

    @staticmethod
    def _standard_image_formatting(im, min_value=0, max_value=1):
        return metrics.standard_image_formatting(im, min_value, max_value)

    def save(self, name='train_model', override=False, dry_run=False):
        save_path = os.path.join(self.saved_models, "{}_{}".format(name, str(self.model_epoch)))
        if os.path.isfile(save_path) and not override:
            raise RuntimeError('Saving this model will result in overwriting existing data!  '
                               'Change model location or enable override.')
        if not dry_run:
            self.model.save_model(model_save_name=name, model_idx=self.model_epoch)
        else:
            print('Training cleared to run.')

    def save_metadata(self):
        import pandas as pd
        pd.DataFrame.from_dict({'model_parameters': [self.model.print_parameters()]}).to_csv(
            os.path.join(self.base_folder, 'extra_metadata.csv'), index=False)

    def print_overview(self):
        if self.mode == 'eval':
            pmode, epoch, message = 'eval', self.model_epoch, 'currently evaluating'
        else:
            pmode, message = 'train', 'will start training from'
            epoch = self.model_epoch if self.model_epoch == 0 else self.model_epoch + 1
        print('----------------------------')
        print('Handler for experiment %s initialized successfully.' % self.experiment)
        print('System loaded in %s mode - %s architecture provided.' % (pmode, self.name))
        print('Model has %d trainable parameters.' % self.model.print_parameters())
        device = self.model.device if str(self.model.device) == 'cpu' else 'GPU ' + str(self.model.device)
        print("Using %s as the model's primary device, and %s epoch %d of the model." % (device, message, epoch))
        self.model.extra_diagnostics()
        print('----------------------------')

    @staticmethod
    def define_model(name, **kwargs):
        return available_models[name](**kwargs)

    def epoch_end_calls(self):
        self.model.epoch_end_calls()

    def get_learning_rate(self):
        return self.model.get_learning_rate()
