"""Data parallelism: one process per GPU, RCCL gradient all-reduce over xGMI.

Replaces the reference's single-process ``nn.DataParallel`` (ref: Code/SISR/models/__init__.py:344-347,
triggered by gpu='multi' at :121-122).  Tiles of a minibatch are independent (no BatchNorm on the path;
GAP/LAM/CSAM reduce within a sample), so the only exchange is one gradient average per step:

  * every rank holds a full replica and runs forward/backward on its shard of the global batch;
  * parameters are grouped into ~8 MB buckets in REVERSE registration order (≈ the order backward
    produces their gradients); as soon as the last gradient of a bucket is accumulated a
    post-accumulate-grad hook flattens the bucket and issues ``all_reduce`` on a side HIP stream, so the
    collective overlaps the rest of backward (62 MB fp32 for RCAN ≈ 0.1-0.7 ms on 7 xGMI links versus
    tens of ms of backward);
  * ``reduce()`` (called by ``standard_update`` between backward and the optimiser step) joins the side
    stream and hands the parameters views of the buckets as their ``.grad`` (no copy back).  The 1 / world_size
    factor is applied to the LOSS by the handlers (``reduce(prescaled=True)``; exact for power-of-two worlds), so the
    sum IS the average; a stand-alone caller gets one in-place scaling launch per bucket instead.  L1 'mean' over
    equal shards makes mean-of-means exact.

Works unchanged on CPU tensors with the gloo backend (used by the world_size-2 tests).
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise torch.distributed from torchrun's environment; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("SISR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":  # "nccl" is RCCL on ROCm
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_batch(batch, rank, world):
    """Contiguous rank slice of a collated batch dict (ref schema: sr_tools/data_handler.py:516-525).
    'metadata_keys' is default_collate's list[M] of B-tuples, so it is sliced per entry.

    A batch that does not divide by the world size (the reference keeps the ragged last batch of an epoch:
    drop_last_training_batch defaults to False, and its DataParallel scatters unevenly) is cut unevenly: the first
    n % world ranks take one sample more.  The slice then carries 'loss_scale' = n_r * world / n, the factor that
    turns the mean of the ranks' mean losses -- what averaging the all-reduced gradients computes -- back into the
    mean over the n samples.  A rank left without a sample gets 'loss_scale' = 0 and empty tensors."""
    n = None
    for k, v in batch.items():
        if torch.is_tensor(v) and v.dim() > 0:
            n = v.shape[0]
            break
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    sl = slice(start, start + base + (1 if rank < extra else 0))
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == n:
            out[k] = v[sl]
        elif k == "metadata_keys" and isinstance(v, (list, tuple)):
            out[k] = [tuple(e[sl]) if isinstance(e, (list, tuple)) and len(e) == n else e for e in v]
        elif isinstance(v, (list, tuple)) and len(v) == n:
            out[k] = list(v[sl])
        else:
            out[k] = v
    if extra:
        out["loss_scale"] = (sl.stop - sl.start) * world / n
    return out


class GradReducer:
    def __init__(self, net, bucket_mb=None, process_group=None, overlap=True, arena=None, arena_flat=None,
                 arena_offsets=None, arena_order=None):
        """arena / arena_flat / arena_offsets: optim.FlatAdam's gradient views, flat gradient buffer and parameter
        offsets.  With them a bucket IS a slice of that buffer (buckets are runs of consecutive parameters), so the
        gradients are all-reduced where the optimiser reads them and nothing is copied back."""
        if not dist.is_initialized():
            raise RuntimeError("gpu='multi' runs one process per GPU: launch with "
                               "`python -m torch.distributed.run --nproc-per-node N ...` "
                               "(see parallel.init_distributed)")
        if bucket_mb is None:  # ~8 MB ~ one residual group of a 64-wide net; SISR_DP_BUCKET_MB overrides (tests, tuning)
            bucket_mb = float(os.environ.get("SISR_DP_BUCKET_MB", 8.0))
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.params = [p for p in net.parameters() if p.requires_grad]
        self.cuda = self.params[0].is_cuda
        cap = int(bucket_mb * (1 << 20) / 4)
        self.buckets, cur, size = [], [], 0
        # buckets walk the optimiser's arena backwards (= reverse registration order, except that parameters whose gradients
        # arrive last -- optim.FlatAdam's `late` -- sit at its end and so share the first bucket(s) instead of holding
        # every bucket open until the end of backward)
        walk = self.params
        if arena_order is not None and arena is not None and all(p in arena for p in self.params):
            keep = {id(p) for p in self.params}
            walk = [p for p in arena_order if id(p) in keep]
        for p in reversed(walk):
            cur.append(p)
            size += p.numel()
            if size >= cap:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        dev = self.params[0].device
        self.bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for p in b:
                self.bucket_of[p] = bi
        # views of the buckets in parameter shapes; conv weights advertise theirs to the weight-gradient kernels
        self.views = {}
        if arena is not None and all(p in arena for p in self.params):
            self.flat = []
            for b in self.buckets:  # reversed registration order: b[-1] has the lowest offset
                lo = min(arena_offsets[p] for p in b)
                hi = max(arena_offsets[p] + (p.numel() + 3) // 4 * 4 for p in b)
                if hi - lo != sum((p.numel() + 3) // 4 * 4 for p in b):
                    raise RuntimeError("GradReducer: bucket is not a contiguous run of the optimiser's gradient arena")
                self.flat.append(arena_flat[lo:hi])
            self.views = {p: arena[p] for p in self.params}
        else:
            self.flat = [torch.zeros(sum(p.numel() for p in b), device=dev) for b in self.buckets]
            for b, flat in zip(self.buckets, self.flat):
                for p, piece in zip(b, flat.split([p.numel() for p in b])):
                    self.views[p] = piece.view_as(p)
        if self.cuda:
            from . import ops
            for p, view in self.views.items():
                if p.is_contiguous():
                    ops.GRAD_SINK[p.data_ptr()] = view
        self.pending = [len(b) for b in self.buckets]
        self.works = [None] * len(self.buckets)
        self.launched = [False] * len(self.buckets)
        self.stream = torch.cuda.Stream(device=dev) if self.cuda else None
        self.overlap = overlap  # False: no hooks, everything is reduced at the join
        # hooks_enabled is switched off by BaseModel._graphed_step while it captures / replays a hipGraph: a captured
        # backward must not launch collectives, and a replay fires no hooks, so every bucket is reduced at the join
        self.hooks_enabled = True
        # hipGraph replays: a replay fires no hooks, so while a step is CAPTURED the hooks put a one-thread signal kernel
        # behind each bucket's last gradient kernel instead (progress words in host-coherent memory, csrc/misc.hip);
        # after graph.replay() the host polls the words and issues every bucket's all-reduce as soon as its word moves,
        # on the reducer stream, while the rest of the replay is still running (launch_signalled).
        self.capturing = False
        self.signal_order = []
        self.flags = None
        self.flag_seen = [0] * len(self.buckets)
        if self.cuda:
            from . import hip
            self.flags = hip.lib().sisr_host_flags_alloc(len(self.buckets) + 1)  # + one "a wait timed out" word
        self.handles = []
        if overlap:
            for p in self.params:
                self.handles.append(p.register_post_accumulate_grad_hook(self._on_grad))
        # replicas must start identical (the reference's DataParallel re-broadcasts every step)
        for p in net.parameters():
            dist.broadcast(p.data, src=0, group=process_group)
        for b in net.buffers():
            dist.broadcast(b.data, src=0, group=process_group)

    def _on_grad(self, p):
        if not self.hooks_enabled:
            return
        bi = self.bucket_of[p]
        self.pending[bi] -= 1
        if self.pending[bi] == 0:
            if self.capturing:  # becomes a node of the graph being captured, right behind this bucket's last gradient
                # on the CAPTURE stream, by handle: a hook runs in the autograd engine's worker thread, whose current stream
                # is not the capturing one (a launch on it would become a parallel branch of the graph, joined at the end).
                # The engine runs one node at a time, so every kernel of the nodes before this hook has been issued already.
                from . import hip
                hip.check(hip.lib().sisr_signal_host(self.flags + 4 * bi, self.capture_stream), "sisr_signal_host")
                self.signal_order.append(bi)
                self.pending[bi] = len(self.buckets[bi])
            else:
                self._launch(bi)

    # -- hipGraph replays (BaseModel._graphed_step)
    @staticmethod
    def graph_overlap_mode():
        """SISR_GRAPH_OVERLAP: auto (default) = 0: all buckets are all-reduced at the join after the replay -- the exchange is
        <= 1 ms of a 53 ms step at 4 tiles per GPU, the signal nodes + wait kernels cost 1.3 ms where there is nothing to hide,
        and the signalled path has never met a multi-rank RCCL world (no node was available to this build), so the robust
        path is the default until it has been measured there | 1 = spin | poll | wait."""
        mode = os.environ.get("SISR_GRAPH_OVERLAP", "auto")
        return "spin" if mode == "1" else mode

    def can_signal(self):
        mode = self.graph_overlap_mode()
        if mode == "auto":
            mode = "0"
        from . import ops
        # SISR_GRAPH_FORK=1 captures the weight gradients as parallel branches: a signal node on the capture stream would not
        # be ordered behind them, so that combination joins all buckets after the replay
        return self.flags is not None and self.overlap and mode != "0" and not ops.GRAPH_FORK

    def begin_capture(self):
        """Call on the capturing stream, right before the backward pass that is being captured."""
        self.capture_stream = torch.cuda.current_stream().cuda_stream
        self.capturing, self.signal_order, self.hooks_enabled = True, [], True
        self.pending = [len(b) for b in self.buckets]

    def end_capture(self):
        """-> the buckets that got a signal node, in the order the captured backward completes them."""
        order, self.capturing, self.signal_order, self.hooks_enabled = self.signal_order, False, [], False
        self.pending = [len(b) for b in self.buckets]
        return order

    def launch_signalled(self, order, timeout_s=60.0):
        """Called right after graph.replay(): every bucket of `order` is all-reduced as soon as the replay reports it complete.

        spin (the default where overlap is on): a one-lane polling kernel on the reducer stream holds each bucket's all-reduce
        back until the replay's signal node has fired; the host enqueues everything at once and keeps running ahead of the
        device, as in eager mode.  Measured on one MI355X (QRCAN, 4 tiles, one-rank RCCL world, ms per step): join 59.1,
        spin 60.4, wait (hipStreamWaitValue32 on the reducer stream: the pending command-processor wait slows the replay's own
        launches) 67.1, poll (host polls the words and blocks meanwhile; the last SISR_GRAPH_OVERLAP_TAIL = 2 buckets are
        left to the stream-ordered join so that the host's remaining work still fits under the device's) 64.2."""
        from . import hip
        mode = self.graph_overlap_mode()
        if mode == "auto":
            mode = "spin"
        tail = max(0, int(os.environ.get("SISR_GRAPH_OVERLAP_TAIL", 2))) if mode == "poll" else 0
        polled = order[:max(0, len(order) - tail)]
        for bi in order:
            self.flag_seen[bi] = (self.flag_seen[bi] + 1) & 0x7fffffff
            if self.flag_seen[bi] == 0:
                raise RuntimeError("GradReducer: progress word wrapped after 2^31 replays; re-create the reducer")
        for bi in polled:
            if mode == "wait":
                hip.check(hip.lib().sisr_stream_wait_flag(self.flags + 4 * bi, self.flag_seen[bi], self.stream.cuda_stream),
                          "sisr_stream_wait_flag")
            elif mode != "poll":  # spin
                hip.check(hip.lib().sisr_stream_spin_flag(self.flags + 4 * bi, self.flag_seen[bi],
                                                          self.flags + 4 * len(self.buckets), self.stream.cuda_stream),
                          "sisr_stream_spin_flag")
            else:
                import ctypes
                import time
                word, t0, spins = ctypes.c_uint32.from_address(self.flags + 4 * bi), None, 0
                while word.value < self.flag_seen[bi]:
                    spins += 1
                    if spins % 2000 == 0:
                        if t0 is None:
                            t0 = time.perf_counter()
                        elif time.perf_counter() - t0 > timeout_s:
                            raise RuntimeError(f"GradReducer: the replayed step never signalled gradient bucket {bi}")
            self._launch(bi, gradients_complete=True)
        for bi in order[len(polled):]:
            self._launch(bi)

    def _launch(self, bi, gradients_complete=False):
        bucket, flat = self.buckets[bi], self.flat[bi]
        # Gradients the kernels of THIS step wrote straight into the bucket (ops.GRAD_SINK) need no copy.  The test is
        # "p.grad aliases the bucket", which is only sound because p.grad always names this step's gradient when we
        # get here: eager steps start from zero_grad() (grads None, or accumulated in place into the view), and a
        # hipGraph replay rebinds p.grad to the tensors its kernels write (BaseModel._graphed_step).
        todo = [p for p in bucket if p.grad is None or p.grad.data_ptr() != self.views[p].data_ptr()]
        dst = [self.views[p] for p in todo]
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in todo]
        if self.cuda:
            if not gradients_complete:  # (a signalled bucket: its kernels have finished; waiting for the stream = the whole replay)
                self.stream.wait_stream(torch.cuda.current_stream())
                from . import ops
                wside = ops.side_stream(flat.device, create=False)  # weight gradients are produced on this stream
                if wside is not None:
                    self.stream.wait_stream(wside)
            with torch.cuda.stream(self.stream):
                if todo:
                    torch._foreach_copy_(dst, grads)
                self.works[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            if todo:
                torch._foreach_copy_(dst, grads)
            self.works[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.launched[bi] = True

    def reduce(self, prescaled=False):
        """Join: every bucket reduced (launching those whose hooks did not fire), grads <- sum / world.
        prescaled: the caller has already divided by the world size (the handlers scale the LOSS by 1 / world before backward,
        so the summed gradients are the average and the per-bucket scaling launches disappear from the step's tail)."""
        for bi in range(len(self.buckets)):
            if not self.launched[bi]:
                self._launch(bi)
        if self.flags is not None:
            import ctypes
            if ctypes.c_uint32.from_address(self.flags + 4 * len(self.buckets)).value:
                raise RuntimeError("GradReducer: a replayed step never signalled one of its gradient buckets (the reducer "
                                   "stream's wait gave up after about a minute)")
        inv = 1.0 / self.world
        for bi, bucket in enumerate(self.buckets):
            self.works[bi].wait()
            if self.cuda:
                torch.cuda.current_stream().wait_stream(self.stream)
            if not prescaled:
                self.flat[bi].mul_(inv)  # one kernel per bucket; the averaged gradients are handed out as views of it
            for p in bucket:
                if p.grad is not self.views[p]:
                    p.grad = self.views[p]
            self.pending[bi] = len(bucket)
            self.launched[bi] = False
            self.works[bi] = None

    def remove(self):
        """Detach from the network.  Captured hipGraphs whose signal nodes write this reducer's progress words must be destroyed
        FIRST (BaseModel.remove_multi_gpu does both in that order): the words are freed here."""
        for h in self.handles:
            h.remove()
        self.handles = []
        if self.flags is not None:
            from . import hip
            torch.cuda.synchronize()
            hip.lib().sisr_host_flags_free(self.flags)
            self.flags = None
        if self.cuda:
            from . import ops
            for p in self.views:
                ops.GRAD_SINK.pop(p.data_ptr(), None)
