"""Autograd operators over the HIP kernels (host side stays Python, as north_star asks).

Layout contract: feature maps are logical (B, C, H, W) tensors with channels_last strides, i.e.
physically NHWC, C a multiple of 64; RGB tensors (C == 3) are plain contiguous NCHW.  Every
operator returns tensors in that contract, so a network never converts layouts internally.

Operators
  conv3x3        default_conv forward/backward incl. fused residual / scalar scale / PixelShuffle store
                 (ref: advanced/common.py:5-8, :20-45)
  res_block      conv-ReLU-conv (+gate) + skip as ONE autograd node: RCAB, QRCAB('standard'), ResBlock,
                 ParamResBlock (ref: advanced/architectures.py:48-71, advanced/common.py:48-72,
                 attention_manipulators/architectures.py:145-180, :332-356)
  meta_gate      ParaCALayer FC stack -> per-(b,c) gate (ref: attention_manipulators/q_layer.py:4-43)
  ca_layer / gate_mul   stand-alone CALayer / x*gate (ref: advanced/architectures.py:13-32)
  l1_loss        nn.L1Loss(mean) (ref: SISR/models/__init__.py:268)
  gated_group    a whole ResidualGroup / QResidualGroup as ONE node (GATE / DOT conv hooks; its weight gradients and gate
                 parameter gradients go out in batches while launches are small: WgradQueue)
  qca_gate       the metadata-mixing QCALayer styles' FC stacks (ref: attention_manipulators/architectures.py:105-127)
  conv_chain / nchw_to_nhwc_pad / shuffle_rgb       SRMD (ref: advanced/architectures.py:380-425)
  sft_layer / sft_apply / sftmd_forward             SFTMD (ref: SFTMD_variants/architectures.py:8-176)
  lam / csam / conv3x3_stack / stack_maps           HAN (ref: advanced/HAN_blocks.py, advanced/architectures.py:314-377)
  soca / nonlocal_block / nonlocal_attention        SAN (ref: advanced/SAN_blocks.py, advanced/mpncov.py)
  pixel_shuffle  nn.PixelShuffle on channels-last maps wider than 64 channels (64-wide: fused into the conv's store)
"""
import os

import torch
from torch.autograd import Function

from . import hip

CL = torch.channels_last


# ----------------------------------------------------------------------------- side stream for weight gradients
# In a block's backward the weight-gradient kernels (wgrad2, wgrad1) do not feed the data-gradient chain
# (dgrad2 -> dgrad1 -> previous block), so they are issued on a second HIP stream: the ramp-up and tail of
# each ~170 us MFMA launch is then filled by workgroups of the other queue.  The main stream re-joins the side
# stream once, at the end of the backward pass (autograd engine callback), before anything reads the grads.
# (Measured +5 % at 16 tiles per GPU when introduced; neutral at 32 with the current kernels -- see DESIGN.md §3.)
WGRAD_SIDE_STREAM = os.environ.get("SISR_WGRAD_SIDE_STREAM", "1") != "0"
# weight gradients as parallel branches of a captured backward: measured SLOWER than one serial stream at 4 tiles per
# GPU (QRCAN, one-rank RCCL world: 48.0 vs 51.5 patches/s -- every cross-branch edge costs a barrier packet), so off
GRAPH_FORK = os.environ.get("SISR_GRAPH_FORK", "0") != "0"
IN_BACKWARD = False  # set while a conv operator's backward runs (bench.py times forward launches only)
_side_streams = {}
_join_pending = set()


def side_stream(device, create=True):
    s = _side_streams.get(device.index)
    if s is None and create:
        s = torch.cuda.Stream(device=device)
        _side_streams[device.index] = s
    return s


def _join_side(device):
    _join_pending.discard(device.index)
    s = _side_streams.get(device.index)
    if s is not None:
        torch.cuda.current_stream(device).wait_stream(s)


def _on_side(device, after_event, fn, tensors):
    """Run fn() on the side stream once `after_event` (recorded on the main stream) has passed; keep the
    caching allocator from recycling `tensors` before the side stream is done with them."""
    side = side_stream(device)
    with torch.cuda.stream(side):
        side.wait_event(after_event)
        fn()
        for t in tensors:
            if t is not None:
                t.record_stream(side)
    if device.index not in _join_pending:
        _join_pending.add(device.index)
        torch.autograd.Variable._execution_engine.queue_callback(lambda: _join_side(device))


_side_groups = {}  # device index -> [(fn, tensors)] waiting for their group's fork


def _on_side_grouped(device, fn, tensors, group):
    """_on_side for many small launches: every cross-stream edge costs a barrier packet (a parallel branch per launch was
    measured slower than no branch at all), so the launches wait and go to the side stream `group` at a time behind ONE
    event; what is left goes out, and the side stream is joined, when the backward pass ends."""
    lst = _side_groups.get(device.index)
    if lst is None:
        lst = _side_groups[device.index] = []
        torch.autograd.Variable._execution_engine.queue_callback(lambda: _flush_side_group(device, final=True))
    lst.append((fn, tensors))
    if len(lst) >= group:
        _flush_side_group(device)


def _flush_side_group(device, final=False):
    lst = _side_groups.get(device.index)
    if final:
        _side_groups.pop(device.index, None)
    elif lst is not None:
        _side_groups[device.index] = []
    if lst:
        ev = torch.cuda.Event()
        ev.record()

        def run():
            for fn, _ in lst:
                fn()
        _on_side(device, ev, run, [t for _, ts in lst for t in ts])
    if final:
        _join_side(device)


def _cl(x):
    return x if x.is_contiguous(memory_format=CL) else x.contiguous(memory_format=CL)


def _empty_cl(B, C, H, W, device):
    return torch.empty((B, C, H, W), device=device, dtype=torch.float32, memory_format=CL)


# Data-parallel runs register, per conv weight, the slice of the all-reduce bucket its gradient belongs to
# (parallel.GradReducer): the weight-gradient kernels then write straight into the bucket and the reducer has
# nothing to copy.  Keyed by the parameter's data pointer; empty in single-process runs.
GRAD_SINK = {}


def _grad_buf(w):
    sink = GRAD_SINK.get(w.data_ptr())
    # only when this backward creates the gradient: with a pre-existing .grad (zero_grad(set_to_none=False),
    # micro-batch accumulation) autograd adds the result to it, and .grad may itself be the bucket view
    if sink is not None and sink.shape == w.shape and w.grad is None:
        return sink.view(sink.shape)  # a fresh alias: autograd adopts a gradient it holds the only reference to
    return torch.empty_like(w)


def _grad_buf_or(p, n, device):
    """_grad_buf for an optional small parameter (bias): a plain [n] buffer when there is none."""
    return _grad_buf(p) if p is not None else torch.empty(n, device=device)


def _side_ok(*weights, pixels=None, fork=False):
    """pixels (B * H * W of the launches): inside a deferred_wgrads() block small weight gradients are queued for batched
    launches on the main stream instead (a queue flushed from the side stream would read operands the allocator only tracks
    on the main stream).
    Weight gradients may be produced on the side stream (which re-joins the main stream only at the end of the
    backward pass) iff autograd will merely ADOPT them: with an existing .grad AccumulateGrad would run
    `p.grad += dw` on the main stream before the side-stream kernel has written dw.  The same holds for a weight that is not a
    leaf (one computed from parameters, e.g. SRMD's transposed-conv tail): its gradient is consumed by further backward nodes
    on the main stream at once."""
    if not (WGRAD_SIDE_STREAM and torch.is_grad_enabled() is False and all(w.is_leaf and w.grad is None for w in weights)):
        return False
    if _DEFERRED is not None and pixels is not None and PRECISION == "fp32" and pixels <= BATCH_WGRAD_MAX_PIXELS:
        return False
    # under hipGraph capture the fork / join would become graph edges (SISR_GRAPH_FORK=1); default: one stream.  fork=True: a
    # caller whose launches leave most of the chip idle (SPARNet's small maps) asks for the branch under capture as well
    return GRAPH_FORK or fork or not torch.cuda.is_current_stream_capturing()


_gate_ws_cache = {}


def _gate_ws(B, device):
    """Scratch of sisr_ca_gate_bwd ([B][80] floats), one per (device, batch): a named, persistent tensor -- a
    temporary built inside the call expression is released before the launch and the very next allocation (the
    lazily created gate counter, once) can be handed the same block."""
    key = (device.index, B)
    t = _gate_ws_cache.get(key)
    if t is None:
        t = torch.empty((B, 80), device=device, dtype=torch.float32)
        _gate_ws_cache[key] = t
    return t


def _vec(B, C, device):
    return torch.empty((B, C), device=device, dtype=torch.float32)


# ----------------------------------------------------------------------------- raw launches (no autograd)
# Arithmetic of the 64-channel-chunk convolutions (forward, input gradient, weight gradient):
#   "fp32"  v_mfma_f32_32x32x2_f32, exact fp32 -- the reference's arithmetic, the default
#   "bf16"  v_mfma_f32_32x32x16_bf16: both operands rounded to bf16 (RNE) on their way into LDS, fp32 accumulate,
#           fp32 feature maps / gradients / optimiser state in HBM (BASELINE config "HAN x4 bf16 ... MFMA")
#   "bf16x3" forward and input-gradient convs with every fp32 operand split exactly into three bf16 numbers and the six
#           significant products on the bf16 matrix cores (fp32-class error, 6/16 of the fp32 MFMA's cycles), weight
#           gradients likewise.  Opt-in, never the headline
# Process-wide; packed weights are rebuilt every step, so switching between steps is safe.
PRECISION = os.environ.get("SISR_PRECISION", "fp32")
X3_WGRAD = os.environ.get("SISR_X3_WGRAD", "1") != "0"  # bf16x3 mode: weight gradients split too (0: exact fp32 kernel)
# Channel-attention gate (and its backward) computed by the last-arriving workgroup of the conv launch that produces its
# partial sums, instead of by a launch of its own on the serial chain ("1" always, "auto" = launches of at most
# CA_TAIL_MAX_BLOCKS workgroups, "0" never).  Bit-identical, but MEASURED SLOWER on MI355X (QRCAN B = 4: 43.7 vs 58.2
# patches/s; RCAN B = 32: 50.6 vs 75.0): the device-scope release every workgroup needs before it is counted
# (__threadfence = L2 write-back on a part whose 8 XCDs have private L2s) flushes the conv's freshly written output
# lines and stalls the wave, which costs far more than the 6-14 us launch it saves.  Default off; kept as the measured
# negative (profiles/r02_ca_tail.json).
CA_TAIL = os.environ.get("SISR_CA_TAIL", "0")
CA_TAIL_MAX_BLOCKS = 1024
FUSED_GROUPS = os.environ.get("SISR_FUSED_GROUPS", "1") != "0"  # group-level autograd node for CA block stacks


def set_precision(name):
    global PRECISION
    if name not in ("fp32", "bf16", "bf16x3"):
        raise ValueError(f"precision must be 'fp32', 'bf16' or 'bf16x3', got {name!r}")
    PRECISION = name


def _wptr(packed):
    return hip.ptr_bf16(packed) if packed.dtype == torch.bfloat16 else hip.ptr(packed)


# Step-level packing: BaseModel.train_step / run_eval call pack_all(net) once, which repacks EVERY conv weight of the
# network with one launch into persistent buffers and publishes them here; pack_pair / pack_weight then find their
# weight (by object identity) and launch nothing.  The entries are dropped as soon as the optimiser has moved the
# weights (invalidate_packs), so a stale packing can never be used; code that calls a network outside the handlers
# simply packs per conv as before.
_STEP_PACKS = {}


class _PackPlan:
    def __init__(self, weights, device):
        import numpy as np
        self.precision = PRECISION
        self.items = [(w, int(r)) for w, r in weights]
        self.ptrs = [w.data_ptr() for w, _ in self.items]
        planes = 3 if PRECISION == "bf16x3" else 1
        # a weight whose channel counts are not multiples of 64 (SPARNet) is packed as its zero-padded twin, same launch
        padded = [(_pad64(w.shape[0]), _pad64(w.shape[1])) for w, _ in self.items]
        numel = [co * ci * 9 for co, ci in padded]
        total = planes * sum(numel)
        dt = torch.float32 if PRECISION == "fp32" else torch.bfloat16
        self.fwd = torch.empty(total, device=device, dtype=dt)
        self.dgrad = torch.empty(total, device=device, dtype=dt)
        esz = self.fwd.element_size()
        jobs = np.zeros(len(self.items), dtype=[("w", "<u8"), ("pf", "<u8"), ("pd", "<u8"), ("cout", "<i4"),
                                                ("cin", "<i4"), ("r", "<i4"), ("first", "<i4"), ("co_real", "<i4"),
                                                ("ci_real", "<i4")])
        assert jobs.dtype.itemsize == hip.lib().sisr_pack_job_bytes()
        off = blocks = 0
        self.slices = []
        for k, (w, r) in enumerate(self.items):
            n = planes * numel[k]
            real = (w.shape[0], w.shape[1]) if padded[k] != (w.shape[0], w.shape[1]) else (0, 0)
            jobs[k] = (w.data_ptr(), self.fwd.data_ptr() + off * esz, self.dgrad.data_ptr() + off * esz, padded[k][0],
                       padded[k][1], r, blocks, real[0], real[1])
            self.slices.append((self.fwd[off:off + n], self.dgrad[off:off + n]))
            off += n
            blocks += (numel[k] + 255) // 256
        self.blocks = blocks
        self.jobs = torch.from_numpy(jobs.view(np.uint8)).to(device)

    def valid(self):
        return self.precision == PRECISION and all(w.data_ptr() == p for (w, _), p in zip(self.items, self.ptrs))

    def run(self):
        hip.check(hip.lib().sisr_pack_conv3x3_many(self.jobs.data_ptr(), len(self.items), self.blocks,
                                                   {"fp32": 0, "bf16": 1, "bf16x3": 2}[PRECISION], hip.stream()),
                  "sisr_pack_conv3x3_many")
        for (w, r), (pf, pd) in zip(self.items, self.slices):
            _STEP_PACKS[id(w)] = (w, r, PRECISION, pf, pd)


_SFT_STEP = {}  # id(mul_conv1.weight) -> (that weight, WA, bA, WB, bB): merged SFT weights composed for this step


class _SftComposePlan:
    """Merged weights of every StandardSft of a network, composed in one launch (sisr_sft_compose_many) into persistent
    buffers; their MFMA packings then ride along in the network's one-launch weight packing."""

    def __init__(self, mods, device):
        import numpy as np
        self.mods = mods
        self.merged = [(torch.empty((64, 128, 3, 3), device=device), torch.empty(64, device=device),
                        torch.empty((128, 64, 3, 3), device=device), torch.empty(128, device=device)) for _ in mods]
        rec = np.zeros(len(mods), dtype=[("p", "<u8", (12,)), ("M", "<i4"), ("split", "<i4")])
        assert rec.dtype.itemsize == hip.lib().sisr_sft_compose_record_bytes()
        self.ptrs = []
        for k, m in enumerate(mods):
            ps = [t.data_ptr() for t in m.params()] + [t.data_ptr() for t in self.merged[k]]
            rec[k] = (ps, m.mul_conv1.weight.shape[1] - 64, 0)
            self.ptrs.append(ps[:8])
        self.table = torch.from_numpy(rec.view(np.uint8)).to(device)

    def valid(self):
        return all(all(t.data_ptr() == q and t.is_contiguous() for t, q in zip(m.params(), ps))
                   for m, ps in zip(self.mods, self.ptrs))

    def run(self):
        hip.check(hip.lib().sisr_sft_compose_many(self.table.data_ptr(), len(self.mods), hip.stream()), "sisr_sft_compose_many")
        for m, mg in zip(self.mods, self.merged):
            w = m.mul_conv1.weight
            _SFT_STEP[id(w)] = (w, *mg)

    def weights(self):
        out = []
        for WA, _, WB, _ in self.merged:
            out += [(WA, 1), (WB, 1)]
        return out


def _sft_plan(net):
    mods = [m for m in net.modules() if type(m).__name__ == "StandardSft" and m.mul_conv1.weight.is_cuda]
    if not mods or PRECISION != "fp32" or not all(t.is_contiguous() for m in mods for t in m.params()):
        return None
    plan = getattr(net, "_sisr_sft_plan", None)
    if plan is None or not plan.valid():
        plan = _SftComposePlan(mods, mods[0].mul_conv1.weight.device)
        net._sisr_sft_plan = plan
        net._sisr_pack_plan = None  # the merged buffers are new tensors: rebuild the packing plan around them
    return plan


def pack_all(net, weights_fn):
    """Repack every 64-multiple 3x3 conv weight of `net` in one launch; weights_fn(net) -> [(weight, shuffle)].  SFT layers:
    their merged weights are composed first (one launch for all of them) and packed in the same launch as the rest."""
    sft = _sft_plan(net)
    if sft is not None:
        sft.run()
    plan = getattr(net, "_sisr_pack_plan", None)
    if plan is None or not plan.valid():
        ws = [(w, r) for w, r in weights_fn(net) if w.is_cuda and w.is_contiguous()]
        if sft is not None:
            ws += sft.weights()
        if not ws:
            return
        plan = _PackPlan(ws, ws[0][0].device)
        net._sisr_pack_plan = plan
    plan.run()


def invalidate_packs():
    _STEP_PACKS.clear()
    _SFT_STEP.clear()


def _step_pack(w, shuffle):
    e = _STEP_PACKS.get(id(w))
    if e is not None and e[0] is w and e[1] == shuffle and e[2] == PRECISION:
        return e[3], e[4]
    return None


def pack_weight(w, mode, shuffle=1):
    """OIHW fp32 weight -> B-fragment order of the MFMA conv ('fwd') or of its input gradient ('dgrad')."""
    hit = _step_pack(w, shuffle)
    if hit is not None:
        return hit[0] if mode == "fwd" else hit[1]
    if PRECISION != "fp32":
        pf, pd = pack_pair(w, shuffle)
        return pf if mode == "fwd" else pd
    cout, cin = w.shape[0], w.shape[1]
    packed = torch.empty(cout * cin * 9, device=w.device, dtype=torch.float32)
    rr = shuffle * shuffle
    if mode == "fwd":
        on, oq = (rr, 1) if shuffle > 1 else (1, 64)
        rc = hip.lib().sisr_pack_conv3x3(hip.ptr(w), hip.ptr(packed), cout, cin, cin * 9, 9, 0, on, oq, 1, 64,
                                         hip.stream())
    else:
        in_, iq = (rr, 1) if shuffle > 1 else (1, 64)
        rc = hip.lib().sisr_pack_conv3x3(hip.ptr(w), hip.ptr(packed), cin, cout, 9, cin * 9, 1, 1, 64, in_, iq,
                                         hip.stream())
    hip.check(rc, "sisr_pack_conv3x3")
    return packed


def pack_pair(w, shuffle=1):
    """(forward packing, input-gradient packing) of one weight, one launch."""
    hit = _step_pack(w, shuffle)
    if hit is not None:
        return hit
    cout, cin = w.shape[0], w.shape[1]
    if PRECISION == "bf16x3":
        buf = torch.empty(2, 3 * cout * cin * 9, device=w.device, dtype=torch.bfloat16)
        hip.check(hip.lib().sisr_pack_conv3x3_x3_both(hip.ptr(w), hip.ptr_bf16(buf[0]), hip.ptr_bf16(buf[1]), cout, cin,
                                                      shuffle, hip.stream()), "sisr_pack_conv3x3_x3_both")
        return buf[0], buf[1]
    if PRECISION == "bf16":
        buf = torch.empty(2, cout * cin * 9, device=w.device, dtype=torch.bfloat16)
        hip.check(hip.lib().sisr_pack_conv3x3_bf16_both(hip.ptr(w), hip.ptr_bf16(buf[0]), hip.ptr_bf16(buf[1]), cout,
                                                        cin, shuffle, hip.stream()), "sisr_pack_conv3x3_bf16_both")
        return buf[0], buf[1]
    buf = torch.empty(2, cout * cin * 9, device=w.device, dtype=torch.float32)
    hip.check(hip.lib().sisr_pack_conv3x3_both(hip.ptr(w), hip.ptr(buf[0]), hip.ptr(buf[1]), cout, cin, shuffle,
                                               hip.stream()), "sisr_pack_conv3x3_both")
    return buf[0], buf[1]


def conv_c64(x, xview, packed, bias, bias_nq, y, yview, B, H, W, cin, cout, res=None, mask=None, in_scale=None,
             in_shift=None, out_scale=None, alpha=1.0, relu=False, gap=None, gate_add=None, gate_out=None, dot=None,
             select=0, ca_tail=None):
    L = hip.lib()
    # the packing decides: a weight packed under one mode runs under it (three bf16 planes = the bf16x3 split)
    if packed.dtype != torch.bfloat16:
        import ctypes
        rc = L.sisr_conv3x3_c64(hip.ptr(x), xview, _wptr(packed), hip.ptr(bias), bias_nq[0], bias_nq[1], hip.ptr(y), yview,
                                hip.ptr(res), hip.ptr(mask), hip.ptr(in_scale), hip.ptr(in_shift), hip.ptr(out_scale),
                                float(alpha), int(relu), hip.ptr(gap), hip.ptr(gate_add), hip.ptr(gate_out), hip.ptr(dot), B,
                                H, W, cin, cout, ctypes.addressof(ca_tail) if ca_tail is not None else None, int(select),
                                hip.stream())
        hip.check(rc, "sisr_conv3x3_c64")
        return
    if ca_tail is not None:
        raise RuntimeError("channel-attention tails exist on the fp32 conv kernels only")
    if packed.numel() == 3 * cin * cout * 9:
        fn, name = L.sisr_conv3x3_c64_x3, "sisr_conv3x3_c64_x3"
    else:
        fn, name = L.sisr_conv3x3_c64_bf16, "sisr_conv3x3_c64_bf16"
    rc = fn(hip.ptr(x), xview, _wptr(packed), hip.ptr(bias), bias_nq[0], bias_nq[1], hip.ptr(y), yview, hip.ptr(res),
            hip.ptr(mask), hip.ptr(in_scale), hip.ptr(in_shift), hip.ptr(out_scale), float(alpha), int(relu),
            hip.ptr(gap), hip.ptr(gate_add), hip.ptr(gate_out), hip.ptr(dot), B, H, W, cin, cout, int(select), hip.stream())
    hip.check(rc, name)


# bf16 STORAGE (on top of PRECISION == "bf16", BASELINE config 5): the maps a residual group keeps for itself and for its
# backward pass -- t1 = ReLU(conv1), t2 = conv2, the gated skips u_k -- live in HBM as bf16 (rounded once, where they are
# written; read back without conversion as MFMA operands), which halves what the group's HBM-bound kernels move for them.
# Group inputs / outputs, gradient maps, gates, partial sums, weights' master copies and the optimiser stay fp32.
#   "0"    fp32 maps (the default)      "act"  the group's activations as above
#   "all"  ... and the gradient maps the group's backward pass hands from launch to launch (dU_k, dt1): rounded once where
#          they are written, after the ReLU mask / the skip's gradient have been applied in fp32; the partial sums of the gate
#          gradients and the bias gradients are taken from fp32 values, the group's input gradient leaves it in fp32
BF16_STORAGE = os.environ.get("SISR_BF16_STORAGE", "0")


def set_storage(name):
    """Storage format of the maps a residual group keeps, in the bf16 operand mode: "0" (fp32), "act" (bf16 activations) or
    "all" (bf16 activations and gradient maps)."""
    global BF16_STORAGE
    if name not in ("0", "act", "all"):
        raise ValueError(f"storage must be '0', 'act' or 'all', got {name!r}")
    BF16_STORAGE = name


def _empty_cl16(B, C, H, W, device):
    return torch.empty((B, C, H, W), device=device, dtype=torch.bfloat16, memory_format=CL)


def to_bf16_map(x):
    """fp32 channels-last map -> bf16 channels-last map (round to nearest even), one launch."""
    y = _empty_cl16(*x.shape, x.device)
    hip.check(hip.lib().sisr_f32_to_bf16(hip.ptr(x), hip.ptr_any(y), x.numel(), hip.stream()), "sisr_f32_to_bf16")
    return y


def conv_c64s(x, packed, bias, y, B, H, W, storage, res=None, mask=None, in_scale=None, in_shift=None, alpha=1.0, relu=False,
              gap=None, gate_add=None, gate_out=None, dot=None):
    """64 -> 64 conv of the bf16 operand mode with bf16-stored maps (include/sisr_hip.h: sisr_conv3x3_c64_bf16s storage bits)."""
    if packed.dtype != torch.bfloat16 or packed.numel() != 64 * 64 * 9:
        raise RuntimeError("bf16 storage needs a weight packed in the bf16 operand mode (64 -> 64)")
    v = hip.view_plain(H, W, 64)
    rc = hip.lib().sisr_conv3x3_c64_bf16s(hip.ptr_any(x), v, _wptr(packed), hip.ptr(bias), 1, 64, hip.ptr_any(y), v,
                                          hip.ptr_any(res), hip.ptr_any(mask), hip.ptr(in_scale), hip.ptr(in_shift), float(alpha),
                                          int(relu), hip.ptr(gap), hip.ptr_any(gate_add), hip.ptr_any(gate_out), hip.ptr_any(dot),
                                          B, H, W, int(storage), hip.stream())
    hip.check(rc, "sisr_conv3x3_c64_bf16s")


_DEFERRED = None  # {(B, H, W, device index): WgradQueue} while a deferred_wgrads() block is open
_DEFERRED_STREAM = None  # the stream the block was opened on: launches issued from another stream are not queued


class deferred_wgrads:
    """While open (around ONE backward pass), plain 64 -> 64 weight gradients of small launches are queued and go out eight at
    a time (WgradQueue), the rest when the block closes.  Only sound when nothing reads a parameter gradient before the block
    closes and every gradient is merely ADOPTED by autograd (grads None on entry): the handlers open it when no reducer hook
    can fire (single process, or hipGraph capture / replay where the buckets are reduced at the join).  The queued launches
    write into the tensors autograd has adopted as .grad (weight AND bias of the conv: both are assumed trainable together).

    Two guards keep a queued gradient from ever being read unwritten:
      * the block flushes on EVERY exit, exceptional ones included: a queue entry keeps the storages of its output buffers
        alive (not the tensors -- see WgradQueue.add), so the late launches are memory-safe even if the backward pass died
        before autograd adopted them, and every .grad that was adopted holds its gradient when the block has closed;
      * a weight is queued at most once per block: a second weight gradient of the same parameter (a conv weight used twice
        in one backward) first flushes every queue -- autograd sums the two results as soon as the second arrives, on this
        stream -- and is launched directly."""

    def __init__(self, enabled=True):
        self.enabled = enabled and BATCH_WGRAD

    def __enter__(self):
        global _DEFERRED, _DEFERRED_STREAM, _DEFERRED_OWNERS
        if self.enabled:
            _DEFERRED, _DEFERRED_STREAM = {}, torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else None
            _DEFERRED_OWNERS = set()
        return self

    def __exit__(self, *exc):
        global _DEFERRED, _DEFERRED_OWNERS
        queues, _DEFERRED, _DEFERRED_OWNERS = _DEFERRED, None, None
        if queues:
            for q in queues.values():
                try:
                    q.flush()
                except Exception:
                    if exc[0] is None:
                        raise  # on an exceptional exit the first error is the one to report
        return False


_DEFERRED_OWNERS = None  # ids of the weights with a queued gradient in the open deferred_wgrads() block


def _flush_deferred():
    if _DEFERRED:
        for q in _DEFERRED.values():
            q.flush()


def wgrad_c64(x, xview, dy, dyview, dw, db, B, H, W, cin, cout, alpha=1.0, dy_scale=None, dy_shift=None, shuffle=1,
              active_units=0, owner=None, storage=0):
    """active_units (fp32 kernel): bit mask of the 32 x 32-channel blocks of the gradient to compute (0 = all).
    owner: the weight Parameter -- with it, inside a deferred_wgrads() block, an eligible launch is queued instead."""
    if (_DEFERRED is not None and hip.stream() == _DEFERRED_STREAM and owner is not None and owner.requires_grad and
            owner.grad is None and cin == 64 and cout == 64 and shuffle == 1 and
            alpha == 1.0 and not active_units and WgradQueue.wanted(B, H, W) and xview is hip.view_plain(H, W, 64) and
            dyview is hip.view_plain(H, W, 64)):
        if id(owner) in _DEFERRED_OWNERS:
            _flush_deferred()  # second gradient of one weight in this pass: the first must be written before autograd adds them
        else:
            _DEFERRED_OWNERS.add(id(owner))
            key = (B, H, W, x.device.index)
            q = _DEFERRED.get(key)
            if q is None:
                q = _DEFERRED[key] = WgradQueue(B, H, W, x.device)
            q.add(x, dy, dw, db, dy_scale=dy_scale, dy_shift=dy_shift, hold_outputs=False)
            return
    L = hip.lib()
    fp32 = False
    if PRECISION == "bf16x3" and X3_WGRAD:
        size_fn, fn, name = L.sisr_wgrad3x3_c64_x3_workspace_bytes, L.sisr_wgrad3x3_c64_x3, "sisr_wgrad3x3_c64_x3"
    elif PRECISION == "bf16":
        size_fn, fn, name = L.sisr_wgrad3x3_c64_bf16_workspace_bytes, L.sisr_wgrad3x3_c64_bf16, "sisr_wgrad3x3_c64_bf16"
    else:
        size_fn, fn, name, fp32 = L.sisr_wgrad3x3_c64_workspace_bytes, L.sisr_wgrad3x3_c64, "sisr_wgrad3x3_c64", True
    nbytes = size_fn(B, H, W, cin, cout)
    ws = hip.workspace(x.device, nbytes)
    rr = shuffle * shuffle
    on, oq = (rr, 1) if shuffle > 1 else (1, 64)
    if storage:  # bf16-stored x (1) or x and dY (3): the bf16 operand mode's kernel reading 2-byte elements
        if PRECISION != "bf16":
            raise RuntimeError("bf16-stored maps exist in the bf16 operand mode only")
        fn, name = L.sisr_wgrad3x3_c64_bf16s, "sisr_wgrad3x3_c64_bf16s"
        args = (hip.ptr_any(x), xview, hip.ptr_any(dy), dyview, hip.ptr(dy_scale), hip.ptr(dy_shift), float(alpha), hip.ptr(dw),
                cin * 9, 9, 0, on, oq, 1, 64, hip.ptr(db), on, oq, hip.ptr(ws), nbytes, B, H, W, cin, cout)
        hip.check(fn(*args, int(storage), hip.stream()), name)
        return
    args = (hip.ptr(x), xview, hip.ptr(dy), dyview, hip.ptr(dy_scale), hip.ptr(dy_shift), float(alpha), hip.ptr(dw),
            cin * 9, 9, 0, on, oq, 1, 64, hip.ptr(db), on, oq, hip.ptr(ws), nbytes, B, H, W, cin, cout)
    rc = fn(*args, int(active_units), hip.stream()) if fp32 else fn(*args, hip.stream())
    hip.check(rc, name)


# Weight gradients of one geometry are launched eight at a time while a launch is small (<= BATCH_WGRAD_MAX_PIXELS pixels per
# map: up to 8 tiles of 128 x 128 per GPU -- BASELINE config 4 has 4): see csrc/wgrad3x3_mfma.hip wgrad3x3_c64_batch_kernel.
BATCH_WGRAD = os.environ.get("SISR_BATCH_WGRAD", "1") != "0"
# ... and in the same regime the channel-attention gate (and its backward) is computed by the conv that consumes it (gate
# HEADS, csrc/conv3x3_mfma.hip template HEAD) instead of a 6 us launch of its own between two convs of the serial chain.
GATE_HEADS = os.environ.get("SISR_GATE_HEADS", "1") != "0"
GATE_HEADS_MAX_PIXELS = int(os.environ.get("SISR_GATE_HEADS_MAX_PIXELS", 4 * 128 * 128))  # measured: +2.3 % at 4 tiles, -1 % at 8
# jobs per launch: the library's maximum, 8 (full per-job argument records by value in the launch arguments).  A variant with
# slim records and up to 48 jobs per launch (a whole group's 41 in one) was tried: the argument struct copied per job went to
# scratch memory and the step got 11 % slower; eight also keeps what the jobs read (2 x 8 maps of 16.8 MB) recent.
BATCH_WGRAD_JOBS = int(os.environ.get("SISR_BATCH_WGRAD_JOBS", 8))
BATCH_WGRAD_MAX_PIXELS = int(os.environ.get("SISR_BATCH_WGRAD_MAX_PIXELS", 8 * 128 * 128))


class WgradQueue:
    """64 -> 64 weight gradients of one (B, H, W) queued by a backward pass and launched in batches (flush() before the
    gradients leave the autograd node).  Holds references to every operand until its launch has been issued."""

    def __init__(self, B, H, W, device):
        self.geo, self.device, self.jobs = (B, H, W), device, []
        self.max = min(BATCH_WGRAD_JOBS, hip.lib().sisr_wgrad3x3_c64_batch_max())

    @staticmethod
    def wanted(B, H, W):
        return BATCH_WGRAD and PRECISION == "fp32" and B * H * W <= BATCH_WGRAD_MAX_PIXELS

    def add(self, x, dy, dw, db, dy_scale=None, dy_shift=None, hold_outputs=True):
        """hold_outputs=False: keep only the ADDRESSES of dw / db, and their STORAGES.  A queue that outlives the autograd node
        must not hold references to the gradient tensors the node returns: AccumulateGrad adopts a gradient only if it holds
        the sole reference to the tensor and clones it otherwise -- the clone would be taken before the launch has written it.
        A storage reference does not count there, and it keeps the memory from being recycled before the flush even when
        the backward pass dies before the gradient is adopted."""
        if hold_outputs:
            self.jobs.append((x, dy, dy_scale, dy_shift, dw, db, None))
        else:
            keep = (dw.untyped_storage(), db.untyped_storage() if db is not None else None)
            self.jobs.append((x, dy, dy_scale, dy_shift, hip.ptr(dw), hip.ptr(db), keep))
        if len(self.jobs) == self.max:
            self.flush()

    def flush(self):
        if not self.jobs:
            return
        B, H, W = self.geo
        L, n = hip.lib(), len(self.jobs)
        arr = (hip.WgradJob * n)()
        for k, (x, dy, sc, sh, dw, db, _) in enumerate(self.jobs):
            arr[k].x, arr[k].dy, arr[k].dy_scale, arr[k].dy_shift = hip.ptr(x), hip.ptr(dy), hip.ptr(sc), hip.ptr(sh)
            arr[k].dw, arr[k].dbias = (dw, db) if isinstance(dw, int) else (hip.ptr(dw), hip.ptr(db))
        nbytes = L.sisr_wgrad3x3_c64_batch_workspace_bytes(n, B, H, W)
        ws = hip.workspace(self.device, nbytes)
        v = hip.view_plain(H, W, 64)
        import ctypes
        hip.check(L.sisr_wgrad3x3_c64_batch(ctypes.addressof(arr), n, v, v, hip.ptr(ws), nbytes, B, H, W, hip.stream()),
                  "sisr_wgrad3x3_c64_batch")
        self.jobs = []


class WgradGeoQueue:
    """Weight gradients of SPARNet's ConvLayer convs (any geometry) queued by a backward pass inside a deferred_wgrads() block
    and launched eight per launch, similar sizes together, when the block closes (sisr_wgrad3x3_c64_geo_batch).  Holds the
    operands, and the STORAGES of the outputs (see WgradQueue.add), until the launches have been issued."""
    MAX_JOBS = 256  # flush earlier than the end of the pass beyond this (bounds what the queue keeps alive)

    def __init__(self, device):
        self.device, self.jobs = device, []

    def add(self, x, dy, dw, db, ints, units, work):
        keep = (dw.untyped_storage(), db.untyped_storage() if db is not None else None)
        self.jobs.append((work, x, dy, hip.ptr(dw), hip.ptr(db), ints, units, keep))
        if len(self.jobs) >= self.MAX_JOBS:
            self.flush()

    def flush(self):
        if not self.jobs:
            return
        import ctypes
        L = hip.lib()
        if L.sisr_wgrad_geo_job_bytes() != ctypes.sizeof(hip.WgradGeoJob):
            raise RuntimeError("hip.WgradGeoJob does not match sisr_wgrad_geo_job of the loaded library")
        jobs = sorted(self.jobs, key=lambda j: j[0])  # launches of eight similar sizes
        arr = (hip.WgradGeoJob * len(jobs))()
        for k, (_, x, dy, dw, db, ints, units, _keep) in enumerate(jobs):
            a = arr[k]
            a.x, a.dy, a.dw, a.dbias = hip.ptr(x), hip.ptr(dy), dw, db
            a.B, a.H, a.W, a.cin, a.cout, a.up, a.co_real, a.ci_real = ints
            a.active_units = units
        nbytes = L.sisr_wgrad3x3_c64_geo_batch_workspace_bytes(ctypes.addressof(arr), len(jobs))
        ws = hip.workspace(self.device, nbytes)
        hip.check(L.sisr_wgrad3x3_c64_geo_batch(ctypes.addressof(arr), len(jobs), hip.ptr(ws), nbytes, hip.stream()),
                  "sisr_wgrad3x3_c64_geo_batch")
        self.jobs = []


def gap_parts(H, W):
    return hip.lib().sisr_conv3x3_c64_gap_parts(H, W)


def _use_ca_tail(B, H, W):
    if PRECISION != "fp32" or CA_TAIL == "0":
        return False
    return CA_TAIL == "1" or B * ((H + 3) // 4) * ((W + 31) // 32) <= CA_TAIL_MAX_BLOCKS


def _tail_fwd(B, H, W, R, caw1c, cab1, caw2c, cab2, mm, sv, hid, ca, g, dev):
    t = hip.CaTail()
    t.backward, t.hidden, t.inv_hw = 0, R, 1.0 / (H * W)
    t.w1, t.b1, t.w2, t.b2, t.mul = hip.ptr(caw1c), hip.ptr_c(cab1), hip.ptr(caw2c), hip.ptr_c(cab2), hip.ptr(mm)
    t.s_out, t.hid_out, t.ca_out, t.g_out = hip.ptr(sv), hip.ptr(hid), hip.ptr(ca), hip.ptr(g)
    t.counter = hip.tail_counter(dev, B)
    return t


def _tail_bwd(B, H, W, R, caw1c, caw2c, s, hid, ca, mm, shift, dmv, dcaw1, dcab1, dcaw2, dcab2, dev):
    t = hip.CaTail()
    t.backward, t.hidden, t.inv_hw = 1, R, 1.0 / (H * W)
    t.w1, t.w2, t.mul = hip.ptr(caw1c), hip.ptr(caw2c), hip.ptr(mm)
    t.s, t.hid, t.ca = hip.ptr(s), hip.ptr(hid), hip.ptr(ca)
    t.shift, t.dmul = hip.ptr(shift), hip.ptr(dmv)
    t.dw1, t.db1, t.dw2, t.db2 = hip.ptr(dcaw1), hip.ptr(dcab1), hip.ptr(dcaw2), hip.ptr(dcab2)
    t.workspace, t.counter = hip.ptr(_gate_ws(B, dev)), hip.tail_counter(dev, B)
    return t


# ----------------------------------------------------------------------------- conv3x3
class _Conv3x3(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, alpha, shuffle):
        cout, cin = weight.shape[0], weight.shape[1]
        B, _, H, W = x.shape
        dev = x.device
        w = weight.contiguous()
        ctx.alpha, ctx.shuffle, ctx.geom = alpha, shuffle, (B, H, W, cin, cout)
        ctx.has_res = residual is not None
        if cin % 64 == 0 and cout % 64 == 0:
            x = _cl(x)
            if ctx.needs_input_grad[0]:
                packed, ctx.packed_dgrad = pack_pair(w, shuffle)
            else:
                packed, ctx.packed_dgrad = pack_weight(w, "fwd", shuffle), None
            if shuffle > 1:
                if cout != 64 * shuffle * shuffle:
                    raise NotImplementedError("fused PixelShuffle needs Cout == 64*r*r")
                y = _empty_cl(B, 64, H * shuffle, W * shuffle, dev)
                yview, bnq = hip.view_shuffle(H, W, shuffle), (shuffle * shuffle, 1)
            else:
                y = _empty_cl(B, cout, H, W, dev)
                yview, bnq = hip.view_plain(H, W, cout), (1, 64)
            res = _cl(residual) if residual is not None else None
            conv_c64(x, hip.view_plain(H, W, cin), packed, bias, bnq, y, yview, B, H, W, cin, cout, res=res,
                     alpha=alpha)
            ctx.kind = "c64"
        elif cin == 3 and cout % 64 == 0:
            if shuffle > 1 or residual is not None or alpha != 1.0:
                raise NotImplementedError("3->64k conv has no fused epilogue")
            x = x.contiguous()
            y = _empty_cl(B, cout, H, W, dev)
            rc = hip.lib().sisr_conv3x3_cin3(hip.ptr(x), hip.ptr(w), 27, 9, 0, hip.ptr(bias), hip.ptr(y),
                                             hip.view_plain(H, W, cout), B, H, W, cout, hip.stream())
            hip.check(rc, "sisr_conv3x3_cin3")
            ctx.kind = "cin3"
        elif cout == 3 and cin % 64 == 0:
            if shuffle > 1 or residual is not None or alpha != 1.0:
                raise NotImplementedError("64k->3 conv has no fused epilogue")
            x = _cl(x)
            y = torch.empty((B, 3, H, W), device=dev, dtype=torch.float32)
            rc = hip.lib().sisr_conv3x3_cout3(hip.ptr(x), hip.view_plain(H, W, cin), hip.ptr(w), cin * 9, 9, 0,
                                              hip.ptr(bias), hip.ptr(y), B, H, W, cin, hip.stream())
            hip.check(rc, "sisr_conv3x3_cout3")
            ctx.kind = "cout3"
        else:
            raise NotImplementedError(
                f"conv3x3 {cin}->{cout}: the gfx950 kernels cover channel counts that are multiples of 64 and the "
                f"3-channel RGB ends (the reference's in-scope configs all use n_feats = 64 or 256)")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.bias = bias  # the leaf itself: its gradient goes straight into the optimiser's arena when it has a slot there
        return y

    @staticmethod
    def backward(ctx, dy):
        global IN_BACKWARD
        IN_BACKWARD = True
        try:
            x, w = ctx.saved_tensors
            B, H, W, cin, cout = ctx.geom
            dev = x.device
            L = hip.lib()
            need_x, need_w, need_b, need_r = ctx.needs_input_grad[:4]
            dx = dw = db = dres = None
            if ctx.kind == "c64":
                dy = _cl(dy)
                r = ctx.shuffle
                dyview = hip.view_shuffle(H, W, r) if r > 1 else hip.view_plain(H, W, cout)
                if need_x:
                    packed = ctx.packed_dgrad if ctx.packed_dgrad is not None else pack_weight(w, "dgrad", r)
                    dx = _empty_cl(B, cin, H, W, dev)
                    conv_c64(dy, dyview, packed, None, (1, 64), dx, hip.view_plain(H, W, cin), B, H, W, cout, cin,
                             alpha=ctx.alpha)
                if need_w or need_b:
                    dw = _grad_buf(w)
                    db = _grad_buf(ctx.bias) if ctx.has_bias else None
                    wgrad_c64(x, hip.view_plain(H, W, cin), dy, dyview, dw, db, B, H, W, cin, cout, alpha=ctx.alpha,
                              shuffle=r, owner=w)
                if ctx.has_res and need_r:
                    dres = dy
            elif ctx.kind == "cin3":
                dy = _cl(dy)
                if need_x:
                    dx = torch.empty((B, 3, H, W), device=dev, dtype=torch.float32)
                    rc = L.sisr_conv3x3_cout3(hip.ptr(dy), hip.view_plain(H, W, cout), hip.ptr(w), 9, 27, 1, None,
                                              hip.ptr(dx), B, H, W, cout, hip.stream())
                    hip.check(rc, "sisr_conv3x3_cout3(dgrad)")
                if need_w or need_b:
                    dw = _grad_buf(w)
                    db = _grad_buf(ctx.bias) if ctx.has_bias else None
                    nbytes = L.sisr_corr3x3_c3_workspace_bytes(B, H, W, cout)
                    ws = hip.workspace(dev, nbytes)
                    rc = L.sisr_corr3x3_c3(hip.ptr(x), hip.ptr(dy), hip.view_plain(H, W, cout), 1.0, hip.ptr(dw), 27, 9, 0,
                                           0, hip.ptr(db), hip.ptr(ws), nbytes, B, H, W, cout, hip.stream())
                    hip.check(rc, "sisr_corr3x3_c3(head)")
            else:  # cout3
                dy = dy.contiguous()
                if need_x:
                    dx = _empty_cl(B, cin, H, W, dev)
                    rc = L.sisr_conv3x3_cin3(hip.ptr(dy), hip.ptr(w), 9, cin * 9, 1, None, hip.ptr(dx),
                                             hip.view_plain(H, W, cin), B, H, W, cin, hip.stream())
                    hip.check(rc, "sisr_conv3x3_cin3(dgrad)")
                if need_w or need_b:
                    dw = _grad_buf(w)
                    db = _grad_buf(ctx.bias) if ctx.has_bias else None
                    nbytes = L.sisr_corr3x3_c3_workspace_bytes(B, H, W, cin)
                    ws = hip.workspace(dev, nbytes)
                    rc = L.sisr_corr3x3_c3(hip.ptr(dy), hip.ptr(x), hip.view_plain(H, W, cin), 1.0, hip.ptr(dw), cin * 9, 9,
                                           1, 1, hip.ptr(db), hip.ptr(ws), nbytes, B, H, W, cin, hip.stream())
                    hip.check(rc, "sisr_corr3x3_c3(tail)")
            return dx, dw, db, dres, None, None
        finally:
            IN_BACKWARD = False


def conv3x3(x, weight, bias=None, residual=None, alpha=1.0, shuffle=1):
    """y = conv3x3(x)*alpha + residual, optionally stored through PixelShuffle(shuffle)."""
    return _Conv3x3.apply(x, weight, bias, residual, float(alpha), int(shuffle))


# ----------------------------------------------------------------------------- meta gate (ParaCALayer FC stack)
class _MetaGate(Function):
    @staticmethod
    def forward(ctx, md, v1, c1, v2, c2, relu):
        B, M = md.shape[0], md.shape[1]
        md2 = md.reshape(B, M).contiguous()
        Hd, C = v1.shape[0], v2.shape[0]
        v1c, v2c = v1.reshape(Hd, M).contiguous(), v2.reshape(C, Hd).contiguous()
        hid, m = _vec(B, Hd, md.device), _vec(B, C, md.device)
        rc = hip.lib().sisr_meta_gate_fwd(hip.ptr(md2), B, M, Hd, C, hip.ptr(v1c), hip.ptr_c(c1),
                                          hip.ptr(v2c), hip.ptr_c(c2), int(relu), hip.ptr(hid), hip.ptr(m),
                                          hip.stream())
        hip.check(rc, "sisr_meta_gate_fwd")
        ctx.save_for_backward(md2, v1c, v2c, hid, m)
        ctx.relu, ctx.shapes = relu, (md.shape, v1.shape, v2.shape)
        return m

    @staticmethod
    def backward(ctx, dm):
        md2, v1c, v2c, hid, m = ctx.saved_tensors
        B, M = md2.shape
        Hd, C = v1c.shape[0], v2c.shape[0]
        dev = md2.device
        s_md, s_v1, s_v2 = ctx.shapes
        # allocated in the parameters' own shapes: a reshaped (view) gradient would be cloned by AccumulateGrad
        dv1, dc1 = torch.empty(s_v1, device=dev), torch.empty(Hd, device=dev)
        dv2, dc2 = torch.empty(s_v2, device=dev), torch.empty(C, device=dev)
        dmd = torch.empty_like(md2) if ctx.needs_input_grad[0] else None
        dmc, ws = dm.contiguous(), _vec(B, Hd + C, dev)  # named: temporaries of a call expression may share a block
        rc = hip.lib().sisr_meta_gate_bwd(hip.ptr(dmc), hip.ptr(m), hip.ptr(hid), hip.ptr(md2), B, M, Hd, C,
                                          hip.ptr(v1c), hip.ptr(v2c), int(ctx.relu), hip.ptr(dv1), hip.ptr(dc1),
                                          hip.ptr(dv2), hip.ptr(dc2), hip.ptr(dmd), hip.ptr(ws), hip.stream())
        hip.check(rc, "sisr_meta_gate_bwd")
        return (dmd.reshape(s_md) if dmd is not None else None, dv1, dc1, dv2, dc2, None)


def meta_gate(md, v1, c1, v2, c2, relu):
    """(B,M,1,1) metadata -> (B,C) sigmoid gate."""
    return _MetaGate.apply(md, v1, c1, v2, c2, bool(relu))


_meta_tables = {}


def _meta_table(params, device):
    """Device table [4][L] of the layers' (v1, c1, v2, c2) addresses, rebuilt when a parameter moved."""
    ptrs = tuple(t.data_ptr() for t in params)
    key = (device.index, len(ptrs))
    hit = _meta_tables.get(key)
    if hit is None or hit[0] != ptrs:
        L = len(ptrs) // 4
        tab = torch.tensor([[ptrs[4 * l + k] for l in range(L)] for k in range(4)], dtype=torch.int64).to(device)
        hit = (ptrs, tab)
        _meta_tables[key] = hit
    return hit[1]


class _MetaGateMany(Function):
    """The gates of L ParaCALayers (same M, hidden size, channel count, nonlinearity) in one launch each way:
    m[l] = sigmoid(V2_l act(V1_l md + c1_l) + c2_l)  ->  (L, B, C).  They depend on the metadata and the layers'
    own weights only, so a network computes all of them before its first block."""

    @staticmethod
    def forward(ctx, md, relu, *params):
        L = len(params) // 4
        B, M = md.shape[0], md.shape[1]
        if md.requires_grad:
            raise NotImplementedError("batched meta gates do not return a metadata gradient")
        for t in params:
            if not t.is_contiguous():
                raise NotImplementedError("batched meta gates need contiguous layer parameters")
        md2 = md.reshape(B, M).contiguous()
        Hd, C = params[0].shape[0], params[2].shape[0]
        dev = md.device
        tab = _meta_table(params, dev)
        hid = torch.empty((L, B, Hd), device=dev, dtype=torch.float32)
        m = torch.empty((L, B, C), device=dev, dtype=torch.float32)
        es = tab.element_size() * L
        base = tab.data_ptr()
        hip.check(hip.lib().sisr_meta_gate_many_fwd(hip.ptr(md2), B, M, Hd, C, L, base, base + es, base + 2 * es,
                                                    base + 3 * es, int(relu), hip.ptr(hid), hip.ptr(m), hip.stream()),
                  "sisr_meta_gate_many_fwd")
        ctx.save_for_backward(md2, hid, m, tab)
        ctx.cfg = (relu, L, B, M, Hd, C, [tuple(t.shape) for t in params[:4]])
        ctx.params = params  # the leaves themselves: their gradient sinks are looked up in backward
        return m

    @staticmethod
    def backward(ctx, dm):
        md2, hid, m, tab = ctx.saved_tensors
        relu, L, B, M, Hd, C, shapes = ctx.cfg
        dev = md2.device
        dmc = dm.contiguous()
        ws = torch.empty(L * B * (Hd + C), device=dev)
        es = tab.element_size() * L
        base = tab.data_ptr()
        sinks = [GRAD_SINK.get(t.data_ptr()) for t in ctx.params]
        if all(sk is not None and sk.shape == t.shape and t.grad is None for sk, t in zip(sinks, ctx.params)):
            # every gradient goes straight into the optimiser's arena (no gather before the update, no per-layer copy)
            gt = _meta_grad_table(sinks, dev)
            ges = gt.element_size() * L
            gb = gt.data_ptr()
            hip.check(hip.lib().sisr_meta_gate_many_bwd_scatter(hip.ptr(dmc), hip.ptr(m), hip.ptr(hid), hip.ptr(md2), B, M, Hd, C,
                                                                L, base, base + 2 * es, int(relu), gb, gb + ges, gb + 2 * ges,
                                                                gb + 3 * ges, hip.ptr(ws), hip.stream()),
                      "sisr_meta_gate_many_bwd_scatter")
            return (None, None, *[sk.view(sk.shape) for sk in sinks])  # fresh aliases: autograd adopts what it alone holds
        dv1 = torch.empty((L,) + shapes[0], device=dev)
        dc1 = torch.empty((L,) + shapes[1], device=dev)
        dv2 = torch.empty((L,) + shapes[2], device=dev)
        dc2 = torch.empty((L,) + shapes[3], device=dev)
        hip.check(hip.lib().sisr_meta_gate_many_bwd(hip.ptr(dmc), hip.ptr(m), hip.ptr(hid), hip.ptr(md2), B, M, Hd, C, L,
                                                    base, base + 2 * es, int(relu), hip.ptr(dv1), hip.ptr(dc1),
                                                    hip.ptr(dv2), hip.ptr(dc2), hip.ptr(ws), hip.stream()),
                  "sisr_meta_gate_many_bwd")
        grads = []
        for l in range(L):  # row views: each is the only reference to its tensor object, so autograd adopts it
            grads += [dv1[l], dc1[l], dv2[l], dc2[l]]
        return (None, None, *grads)


_meta_grad_tables = {}


def _meta_grad_table(sinks, device):
    """Device table [4][L] of the layers' gradient-sink addresses (dv1, dc1, dv2, dc2), rebuilt when a sink moved."""
    ptrs = tuple(t.data_ptr() for t in sinks)
    key = (device.index, len(ptrs))
    hit = _meta_grad_tables.get(key)
    if hit is None or hit[0] != ptrs:
        L = len(ptrs) // 4
        tab = torch.tensor([[ptrs[4 * l + k] for l in range(L)] for k in range(4)], dtype=torch.int64).to(device)
        hit = (ptrs, tab)
        _meta_grad_tables[key] = hit
    return hit[1]


def meta_gate_many(md, layers, relu):
    """layers: [(v1, c1, v2, c2)] of L uniform ParaCALayers -> tuple of L (B, C) gates (views of one tensor)."""
    flat = [t for lay in layers for t in lay]
    return _MetaGateMany.apply(md, bool(relu), *flat).unbind(0)


class _GateMlp(Function):
    """The FC stack of a metadata-mixing QCALayer style on the pooled vector (csrc/attention.hip gate_mlp_*):
    (pool (B,C,1,1), metadata (B,M,1,1), optional meta gate (B,C)) -> gate (B,C,1,1)."""

    @staticmethod
    def forward(ctx, pool, md, mul, spec, *params):
        import ctypes
        layers, final_mode = spec  # layers: [(cat, relu_in, act)], params: w0, b0, w1, b1, ...
        B, C0 = pool.shape[0], pool.shape[1]
        M = md.shape[1]
        dev = pool.device
        L = len(layers)
        d = hip.GateMlpDesc()
        ws = [params[2 * k].reshape(params[2 * k].shape[0], -1).contiguous() for k in range(L)]
        bs = [params[2 * k + 1].contiguous() if params[2 * k + 1] is not None else None for k in range(L)]
        prev = C0
        for k, (cat, relu_in, act) in enumerate(layers):
            nout, inw = ws[k].shape
            if inw != prev + (M if cat else 0):
                raise RuntimeError(f"gate MLP layer {k}: weight expects {inw} inputs, got {prev + (M if cat else 0)}")
            d.w[k], d.b[k] = hip.ptr(ws[k]), hip.ptr(bs[k])
            d.nin[k], d.nout[k], d.cat[k], d.relu_in[k], d.act[k] = prev, nout, int(cat), int(relu_in), int(act)
            prev = nout
        d.L, d.M, d.C, d.final_mode = L, M, prev, final_mode
        p2, md2 = pool.reshape(B, C0).contiguous(), md.reshape(B, M).contiguous()
        mul2 = mul.reshape(B, prev).contiguous() if mul is not None else None
        aw = C0 + sum(w.shape[0] for w in ws)
        acts = torch.empty((B, aw), device=dev, dtype=torch.float32)
        yfin, y = _vec(B, prev, dev), _vec(B, prev, dev)
        hip.check(hip.lib().sisr_gate_mlp_fwd(hip.ptr(p2), hip.ptr(md2), hip.ptr(mul2), B, ctypes.addressof(d),
                                              hip.ptr(acts), hip.ptr(yfin), hip.ptr(y), hip.stream()), "sisr_gate_mlp_fwd")
        ctx.save_for_backward(md2, mul2, acts, yfin, *ws, *[b for b in bs if b is not None])
        ctx.cfg = (layers, final_mode, B, C0, M, prev, [tuple(params[2 * k].shape) for k in range(L)],
                   [b is not None for b in bs], tuple(pool.shape), tuple(md.shape),
                   tuple(mul.shape) if mul is not None else None)
        return y.reshape(B, prev, 1, 1)

    @staticmethod
    def backward(ctx, dy):
        import ctypes
        layers, final_mode, B, C0, M, C, wshapes, has_b, s_pool, s_md, s_mul = ctx.cfg
        sv = list(ctx.saved_tensors)
        md2, mul2, acts, yfin = sv[:4]
        L = len(layers)
        ws = sv[4:4 + L]
        bs_present = sv[4 + L:]
        dev = acts.device
        d = hip.GateMlpDesc()
        prev, bi = C0, 0
        for k, (cat, relu_in, act) in enumerate(layers):
            d.w[k] = hip.ptr(ws[k])
            d.b[k] = hip.ptr(bs_present[bi]) if has_b[k] else None
            bi += int(has_b[k])
            d.nin[k], d.nout[k], d.cat[k], d.relu_in[k], d.act[k] = prev, ws[k].shape[0], int(cat), int(relu_in), int(act)
            prev = ws[k].shape[0]
        d.L, d.M, d.C, d.final_mode = L, M, C, final_mode
        dyc = dy.reshape(B, C).contiguous()
        zw = sum(w.shape[0] for w in ws)
        wsp = torch.empty((B, zw), device=dev, dtype=torch.float32)
        dpool = _vec(B, C0, dev)
        dmd = _vec(B, M, dev) if ctx.needs_input_grad[1] else None
        dmul = _vec(B, C, dev) if (mul2 is not None and ctx.needs_input_grad[2]) else None
        dws = [torch.empty(wshapes[k], device=dev, dtype=torch.float32) for k in range(L)]
        dbs = [torch.empty(ws[k].shape[0], device=dev, dtype=torch.float32) if has_b[k] else None for k in range(L)]
        PA = ctypes.c_void_p * hip.GM_MAXL
        dwa = PA(*[hip.ptr(t) for t in dws])
        dba = PA(*[hip.ptr(t) for t in dbs])
        hip.check(hip.lib().sisr_gate_mlp_bwd(hip.ptr(dyc), hip.ptr(md2), hip.ptr(mul2), B, ctypes.addressof(d),
                                              hip.ptr(acts), hip.ptr(yfin), hip.ptr(wsp), hip.ptr(dpool), hip.ptr(dmd),
                                              hip.ptr(dmul), dwa, dba, hip.stream()), "sisr_gate_mlp_bwd")
        grads = []
        for k in range(L):
            grads += [dws[k], dbs[k]]
        return (dpool.reshape(s_pool), dmd.reshape(s_md) if dmd is not None else None,
                dmul.reshape(s_mul) if dmul is not None else None, None, *grads)


# (cat metadata, ReLU on the concatenated input, activation 0 none / 1 relu / 2 sigmoid) per layer, final mode
QCA_STYLES = {
    "modulate": ([(0, 0, 1), (0, 0, 2)], 2),
    "max_concat": ([(1, 0, 1), (0, 0, 2)], 0),
    "softmax": ([(1, 0, 1), (0, 0, 2)], 1),
    "mini_concat": ([(0, 0, 0), (1, 1, 2)], 0),
    "extended_attention": ([(1, 0, 1), (1, 0, 1), (1, 0, 1), (0, 0, 2)], 0),
}


def qca_gate(pool, md, style, convs, mul=None):
    """convs: the style's nn.Conv2d 1x1 layers in order; mul: optional (B,C) meta-attention gate folded in."""
    params = []
    for c in convs:
        params += [c.weight, c.bias]
    return _GateMlp.apply(pool, md, mul, QCA_STYLES[style], *params)


# ----------------------------------------------------------------------------- fused residual block
class _ResBlock(Function):
    """y = x + gate * res_scale * conv2(relu(conv1(x)));  gate = CA(GAP(.)) [* m] | m | 1."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, caw1, cab1, caw2, cab2, m, res_scale):
        B, C, H, W = x.shape
        has_ca, has_m = caw1 is not None, m is not None
        if C % 64 or tuple(w1.shape) != (C, C, 3, 3) or tuple(w2.shape) != (C, C, 3, 3):
            raise NotImplementedError("fused residual block needs n_feats to be a multiple of 64")
        if C != 64 and (has_ca or has_m):
            raise NotImplementedError("the gated residual blocks (RCAB / QRCAB / ParamResBlock) are specialised for "
                                      "n_feats = 64; the plain ResBlock runs at any multiple of 64 (EDSR 256)")
        _join_pending.clear()  # a backward pass that died mid-way must not leave the join flag set
        dev = x.device
        x = _cl(x)
        w1, w2 = w1.contiguous(), w2.contiguous()
        v = hip.view_plain(H, W, C)
        t1 = _empty_cl(B, C, H, W, dev)
        if any(ctx.needs_input_grad):
            p1, ctx.pd1 = pack_pair(w1)
            p2, ctx.pd2 = pack_pair(w2)
        else:
            p1, p2 = pack_weight(w1, "fwd"), pack_weight(w2, "fwd")
        conv_c64(x, v, p1, b1, (1, 64), t1, v, B, H, W, C, C, relu=True)
        y = _empty_cl(B, C, H, W, dev)
        saved_vecs = []
        if not has_ca and not has_m:  # ResBlock: everything fuses into conv2's epilogue
            conv_c64(t1, v, p2, b2, (1, 64), y, v, B, H, W, C, C, res=x, alpha=res_scale)
            t2 = None
        else:
            t2 = _empty_cl(B, 64, H, W, dev)
            if has_ca:
                parts = gap_parts(H, W)
                gap = torch.empty((B, parts, 64), device=dev, dtype=torch.float32)
                conv_c64(t1, v, p2, b2, (1, 64), t2, v, B, H, W, 64, 64, alpha=res_scale, gap=gap)
                R = caw1.shape[0]
                caw1c, caw2c = caw1.reshape(R, 64).contiguous(), caw2.reshape(64, R).contiguous()
                s, hid, ca, g = _vec(B, 64, dev), _vec(B, R, dev), _vec(B, 64, dev), _vec(B, 64, dev)
                mm = m.contiguous() if has_m else None
                rc = hip.lib().sisr_ca_gate_fwd(hip.ptr(gap), parts, B, 1.0 / (H * W), hip.ptr(caw1c),
                                                hip.ptr_c(cab1), hip.ptr(caw2c), hip.ptr_c(cab2),
                                                64, R, hip.ptr(mm), hip.ptr(s), hip.ptr(hid), hip.ptr(ca), hip.ptr(g),
                                                hip.stream())
                hip.check(rc, "sisr_ca_gate_fwd")
                saved_vecs = [caw1c, caw2c, s, hid, ca, g] + ([mm] if has_m else [])
            else:
                conv_c64(t1, v, p2, b2, (1, 64), t2, v, B, H, W, 64, 64, alpha=res_scale)
                g = m.contiguous()
                saved_vecs = [g]
            rc = hip.lib().sisr_gate_residual_fwd(hip.ptr(t2), hip.ptr(g), None, hip.ptr(x), hip.ptr(y), B, H * W, 64,
                                                  hip.stream())
            hip.check(rc, "sisr_gate_residual_fwd")
        ctx.save_for_backward(x, w1, w2, t1, *([t2] if t2 is not None else []), *saved_vecs)
        ctx.cfg = (has_ca, has_m, float(res_scale), (B, H, W), tuple(caw1.shape) if has_ca else None,
                   tuple(caw2.shape) if has_ca else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        global IN_BACKWARD
        IN_BACKWARD = True
        try:
            has_ca, has_m, rs, (B, H, W), s_caw1, s_caw2 = ctx.cfg
            sv = list(ctx.saved_tensors)
            x, w1, w2, t1 = sv[:4]
            C = x.shape[1]  # 64 for the gated modes; any multiple of 64 for the plain ResBlock
            dev = x.device
            L = hip.lib()
            dy = _cl(dy)
            v = hip.view_plain(H, W, C)
            hw = H * W
            dcaw1 = dcab1 = dcaw2 = dcab2 = dm = None
            scale = shift = None
            if has_ca or has_m:
                t2 = sv[4]
                parts = L.sisr_gate_dg_parts(hw)
                dgp = torch.empty((B, parts, 64), device=dev, dtype=torch.float32)
                hip.check(L.sisr_gate_dg_partial(hip.ptr(dy), hip.ptr(t2), hip.ptr(dgp), B, hw, 64, hip.stream()),
                          "sisr_gate_dg_partial")
                if has_ca:
                    caw1c, caw2c, s, hid, ca, g = sv[5:11]
                    mm = sv[11] if has_m else None
                    R = caw1c.shape[0]
                    shift = _vec(B, 64, dev)
                    dmv = _vec(B, 64, dev) if has_m else None
                    dcaw1, dcab1 = torch.empty(s_caw1, device=dev), torch.empty(R, device=dev)
                    dcaw2, dcab2 = torch.empty(s_caw2, device=dev), torch.empty(64, device=dev)
                    rc = L.sisr_ca_gate_bwd(hip.ptr(dgp), parts, B, 1.0 / hw, hip.ptr(caw1c), hip.ptr(caw2c), 64, R,
                                            hip.ptr(s), hip.ptr(hid), hip.ptr(ca), hip.ptr(mm), hip.ptr(shift),
                                            hip.ptr(dmv), hip.ptr(dcaw1), hip.ptr(dcab1), hip.ptr(dcaw2), hip.ptr(dcab2),
                                            hip.ptr(_gate_ws(B, dev)), hip.gate_counter(dev), hip.stream())
                    hip.check(rc, "sisr_ca_gate_bwd")
                    dm, scale = dmv, g
                else:
                    g = sv[5]
                    dm = _vec(B, 64, dev)
                    hip.check(L.sisr_sum_partials(hip.ptr(dgp), parts, B, 64, 1.0, hip.ptr(dm), hip.stream()),
                              "sisr_sum_partials")
                    scale = g
            # conv2 backward: dt2 = dy*scale + shift is rebuilt on load, never stored
            dw2, db2 = _grad_buf(w2), torch.empty(C, device=dev)
            dw1, db1 = _grad_buf(w1), torch.empty(C, device=dev)
            # plain first-order backward only; not under hipGraph capture (record_stream + private pools)
            side = _side_ok(w1, w2, pixels=B * H * W)

            def wgrad2():
                wgrad_c64(t1, v, dy, v, dw2, db2, B, H, W, C, C, alpha=rs, dy_scale=scale, dy_shift=shift, owner=w2)

            if side:
                ev = torch.cuda.Event()
                ev.record()
                _on_side(dev, ev, wgrad2, (t1, dy, scale, shift, dw2, db2))
            dt1 = _empty_cl(B, C, H, W, dev)
            conv_c64(dy, v, ctx.pd2, None, (1, 64), dt1, v, B, H, W, C, C, mask=t1, in_scale=scale, in_shift=shift,
                     alpha=rs)
            if not side:
                wgrad2()

            def wgrad1():
                wgrad_c64(x, v, dt1, v, dw1, db1, B, H, W, C, C, owner=w1)

            # conv1 backward (+ skip connection gradient)
            if side:
                ev = torch.cuda.Event()
                ev.record()
                _on_side(dev, ev, wgrad1, (x, dt1, dw1, db1))
            dx = None
            if ctx.needs_input_grad[0]:
                dx = _empty_cl(B, C, H, W, dev)
                conv_c64(dt1, v, ctx.pd1, None, (1, 64), dx, v, B, H, W, C, C, res=dy)
            if not side:
                wgrad1()
            return dx, dw1, db1, dw2, db2, dcaw1, dcab1, dcaw2, dcab2, (dm if has_m else None), None
        finally:
            IN_BACKWARD = False


# Sample lanes.  The tiles of a minibatch never meet inside a residual group (no batch statistics on the path), so the chain of
# B-tile launches of a group can run as LANES chains of B / LANES tiles on LANES streams -- parallel branches of the captured
# step.  At 4 tiles per GPU a single chain fills the chip with ONE round of workgroups that stage, multiply and store in
# lock-step (43 - 51 us per launch in the step against 31 us of MFMA time); two chains drift out of phase and one's K loops
# cover the other's staging and stores (tools/lane_probe.py: 38.7 -> 36.7 us per 4-tile link, mask form 40.5 -> 37.6).
# Used where launches are small AND the step is replayed from a hipGraph (eager launches at these sizes are host-bound and
# lanes double their number); the weight gradients stay whole-batch launches: the lanes meet before every batch of eight.
LANES = int(os.environ.get("SISR_LANES", 1))  # measured: no gain inside the replayed step, so one chain by default
LANES_EAGER = os.environ.get("SISR_LANES_EAGER", "0") != "0"  # tests / probes: lanes outside a capture too
LANES_MAX_PIXELS = int(os.environ.get("SISR_LANES_MAX_PIXELS", 8 * 128 * 128))
LANES_INTERLEAVE = os.environ.get("SISR_LANES_INTERLEAVE", "1") != "0"  # issue the lanes' launches block by block, alternating
_lane_streams = {}


def _lane_cuts(B, H, W):
    """[(b0, b1)] sample ranges of the lanes of a group launch chain ([(0, B)]: one chain on the calling stream)."""
    if (LANES < 2 or B % LANES or PRECISION != "fp32" or not WgradQueue.wanted(B, H, W) or B * H * W > LANES_MAX_PIXELS
            or _use_ca_tail(B, H, W) or not (LANES_EAGER or torch.cuda.is_current_stream_capturing())):
        return [(0, B)]
    return [(B * k // LANES, B * (k + 1) // LANES) for k in range(LANES)]


class _Lanes:
    """Lane 0 is the calling stream; lanes 1.. are side streams forked from it (fork()) and joined back into it (join()).
    Every buffer a lane touches is allocated on the calling stream BEFORE the fork and released after the join, so the caching
    allocator's stream-ordered reuse stays sound."""

    def __init__(self, device, cuts):
        self.cuts, self.main = cuts, torch.cuda.current_stream(device)
        pool = _lane_streams.setdefault(device.index, [])
        while len(pool) < len(cuts) - 1:
            pool.append(torch.cuda.Stream(device=device))
        self.side = pool[:len(cuts) - 1]

    def fork(self):
        for s in self.side:
            s.wait_stream(self.main)

    def join(self):
        for s in self.side:
            self.main.wait_stream(s)

    def run(self, fn):
        """fn(b0, b1) once per lane, each on its stream.  A generator function is advanced round-robin -- lane 0 up to its
        first yield, lane 1 up to its first yield, ... -- so that the lanes' launches are issued (and captured) interleaved,
        block by block, instead of one whole chain after the other."""
        gens = []
        for k, (b0, b1) in enumerate(self.cuts):
            if k == 0:
                g = fn(b0, b1)
            else:
                with torch.cuda.stream(self.side[k - 1]):
                    g = fn(b0, b1)
            if hasattr(g, "__next__"):
                gens.append((k, g))
        while gens:
            alive = []
            for k, g in gens:
                try:
                    if k == 0:
                        next(g)
                    else:
                        with torch.cuda.stream(self.side[k - 1]):
                            next(g)
                    alive.append((k, g))
                except StopIteration:
                    pass
            gens = alive


class _GatedGroup(Function):
    """A whole residual group of channel-attention blocks as ONE autograd node:

        out = x + conv_t(u_n),   u_k = u_{k-1} + g_k * conv2_k(relu(conv1_k(u_{k-1}))),   u_0 = x,
        g_k = sigmoid(CA_k(mean_hw(conv2_k(..)))) [* m_k]          (ref: advanced/architectures.py:68-71, :107-110;
                                                                    attention_manipulators/architectures.py:172-180, :229-233)

    Same kernels and arithmetic as the per-block node, but the two passes a block cannot fuse on its own move into
    its neighbours: the gated skip  u_k = t2_k * g_k + u_{k-1}  is built (and written once) by the NEXT conv's
    halo staging, and the gate gradient  sum(dU_k * t2_k)  is taken by the PREVIOUS backward conv's epilogue while
    it produces dU_k.  Per block that removes two HBM-bound launches (3 + 2 map passes with the MFMA units idle).

    Both passes are written as a chain over a sample range [b0, b1) of buffers allocated for the whole batch: one chain over
    all samples on the calling stream, or (small launches inside a hipGraph capture, _lane_cuts) one chain per sample lane on
    parallel streams -- every launch of a lane is the same kernel on a contiguous slice of the batch, so results do not depend
    on the number of lanes (bit-identical)."""

    PER = 9  # w1, b1, w2, b2, caw1, cab1, caw2, cab2, m

    @staticmethod
    def forward(ctx, x, n, alpha, *args):
        B, C, H, W = x.shape
        if C != 64:
            raise NotImplementedError("fused residual group is specialised for n_feats = 64")
        # Blocks WITHOUT channel attention (QEDSR's ParamResBlock, ref attention_manipulators/architectures.py:348-356:
        # x + m * (res_scale * conv2(relu(conv1 x)))): the gate is the meta-attention vector alone, known before the first
        # launch -- no pooling, no gate launch; `alpha` = res_scale rides in conv2's epilogue.  All blocks of a group alike.
        has_ca = args[4] is not None
        if not has_ca and args[8] is None:
            raise NotImplementedError("fused residual group: a block needs a channel-attention gate or a meta gate")
        alpha = float(alpha)
        _join_pending.clear()
        dev, L = x.device, hip.lib()
        x = _cl(x)
        v = hip.view_plain(H, W, 64)
        need = any(ctx.needs_input_grad)
        parts = gap_parts(H, W)
        tails = has_ca and _use_ca_tail(B, H, W)
        heads = has_ca and GATE_HEADS and not tails and WgradQueue.wanted(B, H, W) and B * H * W <= GATE_HEADS_MAX_PIXELS
        PER = _GatedGroup.PER
        lanes = _Lanes(dev, _lane_cuts(B, H, W))
        # bf16 operand mode with bf16 storage: the maps this group keeps (t1, t2, the gated skips) are bf16 in HBM
        st16 = PRECISION == "bf16" and BF16_STORAGE in ("act", "all") and not tails and not heads
        new_map = _empty_cl16 if st16 else _empty_cl
        x_in = to_bf16_map(x) if st16 else x  # block 0's input and first skip, in the group's storage format

        def conv(xx, pk, bias, yy, Bl, storage, **kw):
            if st16:
                conv_c64s(xx, pk, bias, yy, Bl, H, W, storage, **kw)
            else:
                conv_c64(xx, v, pk, bias, (1, 64), yy, v, Bl, H, W, 64, 64, **kw)

        def pack(w):
            return pack_pair(w) if need else (pack_weight(w, "fwd"), None)

        # every buffer of the group, on the calling stream
        blks, tensors, packs, meta, small = [], [], [], [], []
        if not has_ca and alpha != 1.0:  # alpha * m of every block in two launches
            gs_all = torch.stack([args[k * PER + 8] for k in range(n)]).mul_(alpha)
        for k in range(n):
            w1, b1, w2, b2, caw1, cab1, caw2, cab2, m = args[k * PER:(k + 1) * PER]
            w1, w2 = w1.contiguous(), w2.contiguous()
            p1, pd1 = pack(w1)
            p2, pd2 = pack(w2)
            if has_ca:
                R = caw1.shape[0]
                d = dict(p1=p1, p2=p2, b1=b1, b2=b2, R=R, t1=new_map(B, 64, H, W, dev), t2=new_map(B, 64, H, W, dev),
                         u=new_map(B, 64, H, W, dev) if k > 0 else x_in,
                         gap=torch.empty((B, parts, 64), device=dev, dtype=torch.float32),
                         caw1c=caw1.reshape(R, 64).contiguous(), caw2c=caw2.reshape(64, R).contiguous(),
                         cb1=cab1.contiguous(), cb2=cab2.contiguous(), mm=m.contiguous() if m is not None else None,
                         sv=_vec(B, 64, dev), hid=_vec(B, R, dev), ca=_vec(B, 64, dev), g=_vec(B, 64, dev))
                blk = [d["u"], w1, w2, d["t1"], d["t2"], d["caw1c"], d["caw2c"], d["sv"], d["hid"], d["ca"], d["g"]]
                if d["mm"] is not None:
                    blk.append(d["mm"])
                meta.append((len(blk), d["mm"] is not None, tuple(caw1.shape), tuple(caw2.shape)))
            else:
                mm = m.contiguous()
                # gs = alpha * m: the scale of the gradient entering conv2 (its input gradient and its weight gradient both
                # rebuild dt2 = dU * gs while staging, so both run with alpha = 1 and the weight gradient stays batchable)
                d = dict(p1=p1, p2=p2, b1=b1, b2=b2, R=0, t1=new_map(B, 64, H, W, dev), t2=new_map(B, 64, H, W, dev),
                         u=new_map(B, 64, H, W, dev) if k > 0 else x_in, gap=None, mm=mm, g=mm,
                         gs=mm if alpha == 1.0 else gs_all[k], sv=None, hid=None, ca=None)
                blk = [d["u"], w1, w2, d["t1"], d["t2"], d["g"], d["gs"]]
                meta.append((len(blk), True, None, None))
            blks.append(d)
            small.append((b1, b2, caw1, cab1, caw2, cab2))
            tensors += blk
            packs.append((pd1, pd2))
        wt, bt = args[n * PER:n * PER + 2]
        wt = wt.contiguous()
        pt, pdt = pack(wt)
        un, out = new_map(B, 64, H, W, dev), _empty_cl(B, 64, H, W, dev)
        keep = []  # argument records the launches of a lane point into
        interleave = len(lanes.cuts) > 1 and LANES_INTERLEAVE

        def chain(b0, b1):
            """The group's forward launches on samples [b0, b1) (a generator: with several lanes it yields after every block)."""
            sl, Bl = slice(b0, b1), b1 - b0
            pend = None  # (t2, g, gate head) of the block whose gated skip the next conv builds
            for k in range(n):
                d = blks[k]
                t1, t2, g = d["t1"][sl], d["t2"][sl], d["g"][sl]
                gap, sv, hid, ca = (d[key][sl] if has_ca else None for key in ("gap", "sv", "hid", "ca"))
                mm = d["mm"][sl] if d["mm"] is not None else None
                if pend is None:
                    conv(x_in[sl], d["p1"], d["b1"], t1, Bl, 3, relu=True)
                elif st16:
                    conv(pend[0], d["p1"], d["b1"], t1, Bl, 3, relu=True, in_scale=pend[1], gate_add=blks[k - 1]["u"][sl],
                         gate_out=d["u"][sl])
                else:
                    conv_c64(pend[0], v, d["p1"], d["b1"], (1, 64), t1, v, Bl, H, W, 64, 64, relu=True, in_scale=pend[1],
                             gate_add=blks[k - 1]["u"][sl], gate_out=d["u"][sl], ca_tail=pend[2])
                hd = None
                if not has_ca:  # the gate is the meta-attention vector: nothing to pool, nothing to launch
                    conv(t1, d["p2"], d["b2"], t2, Bl, 3, alpha=alpha)
                elif tails:  # the gate is computed by conv2's last-arriving workgroup per sample
                    conv_c64(t1, v, d["p2"], d["b2"], (1, 64), t2, v, Bl, H, W, 64, 64, gap=gap,
                             ca_tail=_tail_fwd(Bl, H, W, d["R"], d["caw1c"], d["cb1"], d["caw2c"], d["cb2"], mm, sv, hid, ca, g, dev))
                elif heads:  # the gate is computed by the conv that consumes it (the next block's conv1, or the group's tail conv)
                    conv_c64(t1, v, d["p2"], d["b2"], (1, 64), t2, v, Bl, H, W, 64, 64, gap=gap)
                    hd = hip.CaTail()
                    hd.backward, hd.hidden, hd.inv_hw, hd.head, hd.head_parts = 0, d["R"], 1.0 / (H * W), 1, parts
                    hd.w1, hd.b1, hd.w2, hd.b2, hd.mul = (hip.ptr(d["caw1c"]), hip.ptr(d["cb1"]), hip.ptr(d["caw2c"]),
                                                          hip.ptr(d["cb2"]), hip.ptr(mm))
                    hd.s_out, hd.hid_out, hd.ca_out, hd.g_out = hip.ptr(sv), hip.ptr(hid), hip.ptr(ca), hip.ptr(g)
                    hd.head_part = hip.ptr(gap)
                    keep.append(hd)
                else:
                    conv(t1, d["p2"], d["b2"], t2, Bl, 3, gap=gap)
                    hip.check(L.sisr_ca_gate_fwd(hip.ptr(gap), parts, Bl, 1.0 / (H * W), hip.ptr(d["caw1c"]), hip.ptr(d["cb1"]),
                                                 hip.ptr(d["caw2c"]), hip.ptr(d["cb2"]), 64, d["R"], hip.ptr(mm), hip.ptr(sv),
                                                 hip.ptr(hid), hip.ptr(ca), hip.ptr(g), hip.stream()), "sisr_ca_gate_fwd")
                pend = (t2, g, hd)
                if interleave:
                    yield
            if st16:  # bf16 t2 and skips in, the group's output (+ its fp32 input as the residual) in fp32
                conv(pend[0], pt, bt, out[sl], Bl, 1, in_scale=pend[1], gate_add=blks[n - 1]["u"][sl], gate_out=un[sl], res=x[sl])
            else:
                conv_c64(pend[0], v, pt, bt, (1, 64), out[sl], v, Bl, H, W, 64, 64, in_scale=pend[1],
                         gate_add=blks[n - 1]["u"][sl], gate_out=un[sl], res=x[sl], ca_tail=pend[2])

        lanes.fork()
        if interleave:
            lanes.run(chain)
        else:
            lanes.run(lambda b0, b1: [None for _ in chain(b0, b1)] and None)
        lanes.join()
        ctx.save_for_backward(*tensors, un, wt)
        ctx.cfg = (n, (B, H, W), meta, parts)
        ctx.packs, ctx.pdt = packs, pdt
        ctx.small, ctx.bt = small, bt  # the small parameters: their gradients go straight into the optimiser's arena too
        ctx.st16, ctx.grad16 = st16, st16 and BF16_STORAGE == "all"
        ctx.has_ca = has_ca
        return out

    @staticmethod
    def backward(ctx, dout):
        global IN_BACKWARD
        IN_BACKWARD = True
        try:
            n, (B, H, W), meta, parts = ctx.cfg
            sv_all = list(ctx.saved_tensors)
            un, wt = sv_all[-2], sv_all[-1]
            dev, L = un.device, hip.lib()
            v = hip.view_plain(H, W, 64)
            hw = H * W
            dout = _cl(dout)

            def run(fn, keep):
                if side:
                    ev = torch.cuda.Event()
                    ev.record()
                    _on_side(dev, ev, fn, keep)
                else:
                    fn()

            blocks, pos = [], 0
            for cnt, has_m, s1, s2 in meta:
                blocks.append((sv_all[pos:pos + cnt], has_m, s1, s2))
                pos += cnt
            # st16: the saved activations (un, every block's input, t1, t2) are bf16 maps; g16: so are the gradient maps the
            # launches of this pass hand to each other (the group's own input / output gradients stay fp32)
            st16, g16 = ctx.st16, ctx.grad16
            has_ca = ctx.has_ca
            wst = 1 if st16 else 0
            grad_map = _empty_cl16 if g16 else _empty_cl

            def conv(xx, pk, yy, Bl, **kw):
                """a backward conv of the group under st16: bf16 mask / dot operands; storage bits from the maps' own dtypes"""
                st = ((1 if xx.dtype == torch.bfloat16 else 0) | (2 if yy.dtype == torch.bfloat16 else 0) |
                      (4 if (kw.get("mask") is not None or kw.get("dot") is not None) else 0) |
                      (8 if (kw.get("res") is not None and kw["res"].dtype == torch.bfloat16) else 0))
                if st:
                    conv_c64s(xx, pk, None, yy, Bl, H, W, st, **kw)
                else:
                    conv_c64(xx, v, pk, None, (1, 64), yy, v, Bl, H, W, 64, 64, **kw)

            side = _side_ok(wt, *(t for blk in blocks for t in blk[0][1:3]))
            # small launches: the group's 2n + 1 weight gradients go out eight to a launch (WgradQueue) instead of one by one
            queue = WgradQueue(B, H, W, dev) if WgradQueue.wanted(B, H, W) else None
            tails = has_ca and _use_ca_tail(B, H, W)
            bheads = has_ca and not tails and queue is not None and GATE_HEADS and B * H * W <= GATE_HEADS_MAX_PIXELS
            lanes = _Lanes(dev, _lane_cuts(B, H, W) if queue is not None else [(0, B)])
            nl = len(lanes.cuts)
            gate_jobs, keep = [], []
            # tail conv: weight gradient from (u_n, dout); dU_n = convT(dout), with sum(dU_n * t2_n) on the side
            dwt, dbt = _grad_buf(wt), _grad_buf_or(ctx.bt, 64, dev)
            if queue is not None:
                queue.add(un, dout, dwt, dbt)
            else:
                run(lambda: wgrad_c64(un, v, dout, v, dwt, dbt, B, H, W, 64, 64, storage=wst), (un, dout, dwt, dbt))

            def gate_bwd_out(k):
                """Outputs of block k's gate backward (allocated before the conv launch whose tail / head fills them)."""
                tens, has_m, s_caw1, s_caw2 = blocks[k]
                _, _, caw1, cab1, caw2, cab2 = ctx.small[k]
                if not has_ca:  # the meta gate's gradient is the sum of the DOT partials; no pooling gradient, no CA parameters
                    return dict(shift=None, dmv=_vec(B, 64, dev), dcaw1=None, dcab1=None, dcaw2=None, dcab2=None, dzw=None,
                                dgp=torch.empty((B, parts, 64), device=dev, dtype=torch.float32))
                return dict(shift=_vec(B, 64, dev), dmv=_vec(B, 64, dev) if has_m else None,
                            dcaw1=_grad_buf(caw1), dcab1=_grad_buf(cab1), dcaw2=_grad_buf(caw2), dcab2=_grad_buf(cab2),
                            dgp=torch.empty((B, parts, 64), device=dev, dtype=torch.float32),
                            dzw=torch.empty((B, 80), device=dev) if queue is not None and not tails else None)

            def tail_for(k, o):
                if not tails:
                    return None
                tens, has_m = blocks[k][0], blocks[k][1]
                caw1c, caw2c, s, hid, ca = tens[5:10]
                return _tail_bwd(B, H, W, caw1c.shape[0], caw1c, caw2c, s, hid, ca, tens[11] if has_m else None, o["shift"],
                                 o["dmv"], o["dcaw1"], o["dcab1"], o["dcaw2"], o["dcab2"], dev)

            def block_bufs(k):
                """Maps block k's backward writes: dt1 (gradient at ReLU(conv1)) and dprev (gradient at the block's input), plus
                the gate-backward outputs of block k - 1, which block k's last conv feeds."""
                return dict(dt1=grad_map(B, 64, H, W, dev), dprev=(grad_map if k > 0 else _empty_cl)(B, 64, H, W, dev),
                            go=gate_bwd_out(k - 1) if k > 0 else None)

            def first_conv(b0, b1, dy, go):
                """dU_n = convT_tail(dout), with the partial sums of sum(dU_n * t2_n) for block n - 1's gate backward."""
                sl = slice(b0, b1)
                if st16:
                    conv(dout[sl], ctx.pdt, dy[sl], b1 - b0, gap=go["dgp"][sl], dot=blocks[-1][0][4][sl])
                else:
                    conv_c64(dout[sl], v, ctx.pdt, None, (1, 64), dy[sl], v, b1 - b0, H, W, 64, 64, gap=go["dgp"][sl],
                             dot=blocks[-1][0][4][sl], ca_tail=tail_for(n - 1, go))

            def block_chain(k, b0, b1, dy, go, bufs, mid1=None, mid2=None):
                """Block k's part of the input-gradient chain on samples [b0, b1): gate backward (as a launch, a head of the next
                conv, or the previous conv's tail), dgrad through conv2 (ReLU mask, gated gradient rebuilt while staging), dgrad
                through conv1 (+ the skip's gradient; partial sums for block k - 1's gate backward).  mid1 / mid2: called after
                the gate backward / after the first conv (where a single chain issues the block's weight gradients)."""
                sl, Bl = slice(b0, b1), b1 - b0
                tens, has_m, s_caw1, s_caw2 = blocks[k]
                pd1, pd2 = ctx.packs[k]
                bhead = None
                if not has_ca:
                    xk, w1, w2, t1, t2, g, gs = tens
                    hip.check(L.sisr_sum_partials(hip.ptr(go["dgp"][sl]), parts, Bl, 64, 1.0, hip.ptr(go["dmv"][sl]), hip.stream()),
                              "sisr_sum_partials")
                    g, shift = gs, None  # dt2 = dU * (alpha * m)
                else:
                    xk, w1, w2, t1, t2, caw1c, caw2c, s, hid, ca, g = tens[:11]
                    mm = tens[11][sl] if has_m else None
                    R = caw1c.shape[0]
                    shift, dmv = go["shift"][sl], go["dmv"][sl] if has_m else None
                if not has_ca:
                    pass
                elif bheads:  # the per-sample part of the gate backward is computed by the conv that consumes `shift` (gate head)
                    bhead = hip.CaTail()
                    bhead.backward, bhead.hidden, bhead.inv_hw, bhead.head, bhead.head_parts = 1, R, 1.0 / hw, 1, parts
                    bhead.w1, bhead.w2, bhead.hid, bhead.ca, bhead.mul = (hip.ptr(caw1c), hip.ptr(caw2c), hip.ptr(hid[sl]),
                                                                          hip.ptr(ca[sl]), hip.ptr(mm))
                    bhead.shift, bhead.dmul, bhead.workspace, bhead.head_part = (hip.ptr(shift), hip.ptr(dmv), hip.ptr(go["dzw"][sl]),
                                                                                  hip.ptr(go["dgp"][sl]))
                    keep.append(bhead)
                elif not tails and queue is not None:
                    # small launches: the chain kernel only produces what the next conv waits for; the gates' parameter
                    # gradients of the whole group are one launch at the end (their dz2 / dz1 wait in per-gate workspaces)
                    hip.check(L.sisr_ca_gate_bwd(hip.ptr(go["dgp"][sl]), parts, Bl, 1.0 / hw, hip.ptr(caw1c), hip.ptr(caw2c), 64, R,
                                                 hip.ptr(s[sl]), hip.ptr(hid[sl]), hip.ptr(ca[sl]), hip.ptr(mm), hip.ptr(shift),
                                                 hip.ptr(dmv), None, None, None, None, hip.ptr(go["dzw"][sl]), None, hip.stream()),
                              "sisr_ca_gate_bwd")
                elif not tails:  # (one chain over the whole batch: queue is None implies a single lane)
                    hip.check(L.sisr_ca_gate_bwd(hip.ptr(go["dgp"]), parts, B, 1.0 / hw, hip.ptr(caw1c), hip.ptr(caw2c), 64, R,
                                                 hip.ptr(s), hip.ptr(hid), hip.ptr(ca), hip.ptr(mm), hip.ptr(shift),
                                                 hip.ptr(dmv), hip.ptr(go["dcaw1"]), hip.ptr(go["dcab1"]), hip.ptr(go["dcaw2"]),
                                                 hip.ptr(go["dcab2"]), hip.ptr(_gate_ws(B, dev)), hip.gate_counter(dev), hip.stream()),
                              "sisr_ca_gate_bwd")
                if mid1 is not None:
                    mid1()
                dt1, dprev = bufs["dt1"], bufs["dprev"]
                if st16:
                    conv(dy[sl], pd2, dt1[sl], Bl, mask=t1[sl], in_scale=g[sl], in_shift=shift)
                else:
                    conv_c64(dy[sl], v, pd2, None, (1, 64), dt1[sl], v, Bl, H, W, 64, 64, mask=t1[sl], in_scale=g[sl],
                             in_shift=shift, ca_tail=bhead)
                if mid2 is not None:
                    mid2()
                if k > 0 and st16:
                    gn = bufs["go"]
                    conv(dt1[sl], pd1, dprev[sl], Bl, res=dy[sl], gap=gn["dgp"][sl], dot=blocks[k - 1][0][4][sl])
                elif k > 0:
                    gn = bufs["go"]
                    conv_c64(dt1[sl], v, pd1, None, (1, 64), dprev[sl], v, Bl, H, W, 64, 64, res=dy[sl], gap=gn["dgp"][sl],
                             dot=blocks[k - 1][0][4][sl], ca_tail=tail_for(k - 1, gn))
                elif g16:
                    conv(dt1[sl], pd1, dprev[sl], Bl, res=dy[sl])  # block 0: bf16 in, bf16 residual, the group's fp32 dX out
                else:
                    conv_c64(dt1[sl], v, pd1, None, (1, 64), dprev[sl], v, Bl, H, W, 64, 64, res=dy[sl])

            grads = [None] * (n * _GatedGroup.PER)

            def wgrad2(k, dy, go):
                """Weight gradient of block k's second conv, from (t1, the gated gradient dy * g + shift rebuilt while staging)."""
                tens = blocks[k][0]
                w2, t1, g = tens[2], tens[3], tens[10] if has_ca else tens[6]
                dw2, db2 = _grad_buf(w2), _grad_buf_or(ctx.small[k][1], 64, dev)
                shift = go["shift"]
                if queue is not None:
                    queue.add(t1, dy, dw2, db2, dy_scale=g, dy_shift=shift)
                else:
                    run(lambda: wgrad_c64(t1, v, dy, v, dw2, db2, B, H, W, 64, 64, dy_scale=g, dy_shift=shift,
                                          storage=3 if g16 else wst), (t1, dy, g, shift, dw2, db2))
                grads[k * _GatedGroup.PER + 2:k * _GatedGroup.PER + 4] = [dw2, db2]

            def wgrad1(k, go, bufs):
                """Weight gradient of block k's first conv, from (the block's input, dt1); the block's small gradients."""
                tens, has_m = blocks[k][0], blocks[k][1]
                xk, w1 = tens[0], tens[1]
                caw1c, s, hid = (tens[5], tens[7], tens[8]) if has_ca else (None, None, None)
                dw1, db1 = _grad_buf(w1), _grad_buf_or(ctx.small[k][0], 64, dev)
                dt1 = bufs["dt1"]
                if queue is not None:
                    if not tails and has_ca:
                        gate_jobs.append((go["dzw"], hid, s, go["dcaw1"], go["dcab1"], go["dcaw2"], go["dcab2"], caw1c.shape[0]))
                    queue.add(xk, dt1, dw1, db1)
                else:
                    run(lambda: wgrad_c64(xk, v, dt1, v, dw1, db1, B, H, W, 64, 64, storage=3 if g16 else wst),
                        (xk, dt1, dw1, db1))
                P = _GatedGroup.PER
                grads[k * P:k * P + 2] = [dw1, db1]
                grads[k * P + 4:k * P + 9] = [go["dcaw1"], go["dcab1"], go["dcaw2"], go["dcab2"], go["dmv"] if has_m else None]

            dy = grad_map(B, 64, H, W, dev)
            go = gate_bwd_out(n - 1)
            if nl == 1:
                # one chain on the calling stream: buffers come and go block by block, weight gradients follow their operands
                first_conv(0, B, dy, go)
                for k in range(n - 1, -1, -1):
                    bufs = block_bufs(k)
                    if queue is None:  # side stream: wgrad2 beside the conv2 input gradient, wgrad1 beside conv1's
                        block_chain(k, 0, B, dy, go, bufs, mid1=lambda: wgrad2(k, dy, go), mid2=lambda: wgrad1(k, go, bufs))
                    else:  # queued: after the first conv (with a gate head it is that launch which fills `shift`)
                        block_chain(k, 0, B, dy, go, bufs, mid2=lambda: (wgrad2(k, dy, go), wgrad1(k, go, bufs)))
                    dy, go = bufs["dprev"], bufs["go"]
            else:
                # sample lanes: the chains of a SEGMENT of blocks run side by side; then the lanes meet, the segment's weight
                # gradients (whole-batch launches, eight to a launch) run, and the lanes part again for the next segment
                seg = max(1, int(os.environ.get("SISR_LANES_SEGMENT", 4)))
                state = {"dy": dy, "go": go}
                k_hi = n - 1
                first = True
                while k_hi >= 0:
                    ks = list(range(k_hi, max(-1, k_hi - seg), -1))
                    allb = {k: block_bufs(k) for k in ks}  # before the fork, on the calling stream

                    def segment(b0, b1, first=first, ks=ks, allb=allb, dy0=state["dy"], go0=state["go"]):
                        dyl, gol = dy0, go0
                        if first:
                            first_conv(b0, b1, dyl, gol)
                        for k in ks:
                            block_chain(k, b0, b1, dyl, gol, allb[k])
                            dyl, gol = allb[k]["dprev"], allb[k]["go"]
                            if LANES_INTERLEAVE:
                                yield

                    lanes.fork()
                    if LANES_INTERLEAVE:
                        lanes.run(segment)
                    else:
                        lanes.run(lambda b0, b1: [None for _ in segment(b0, b1)] and None)
                    lanes.join()
                    dyl, gol = state["dy"], state["go"]
                    for k in ks:
                        wgrad2(k, dyl, gol)
                        wgrad1(k, gol, allb[k])
                        dyl, gol = allb[k]["dprev"], allb[k]["go"]
                    state["dy"], state["go"] = dyl, gol
                    first = False
                    k_hi -= seg
                dy = state["dy"]
            dx = _affine(dy, None, None, dout, B, H, W, 64) if ctx.needs_input_grad[0] else None
            if queue is not None:
                queue.flush()  # before the gradients leave the node (reducer hooks may read them right after)
                import ctypes
                cap = L.sisr_ca_gate_bwd_params_batch_max()
                for i0 in range(0, len(gate_jobs), cap):
                    chunk = gate_jobs[i0:i0 + cap]
                    arr = (hip.CaParamJob * len(chunk))()
                    for k, (dzw, hid, s, a1, c1, a2, c2, _) in enumerate(chunk):
                        arr[k].dz, arr[k].hid, arr[k].s = hip.ptr(dzw), hip.ptr(hid), hip.ptr(s)
                        arr[k].dw1, arr[k].db1, arr[k].dw2, arr[k].db2 = hip.ptr(a1), hip.ptr(c1), hip.ptr(a2), hip.ptr(c2)
                    hip.check(L.sisr_ca_gate_bwd_params_batch(ctypes.addressof(arr), len(chunk), B, chunk[0][7], hip.stream()),
                              "sisr_ca_gate_bwd_params_batch")
            return (dx, None, None, *grads, dwt, dbt)
        finally:
            IN_BACKWARD = False


def gated_group(x, blocks, tail_w, tail_b, alpha=1.0):
    """blocks: list of (w1, b1, w2, b2, (caw1, cab1, caw2, cab2) or None, m or None); alpha: scale of every block's second conv
    (res_scale); see _GatedGroup."""
    flat = []
    for w1, b1, w2, b2, ca, m in blocks:
        flat += [w1, b1, w2, b2, *(ca if ca is not None else (None,) * 4), m]
    return _GatedGroup.apply(x, len(blocks), float(alpha), *flat, tail_w, tail_b)


def fused_groups_enabled():
    """Group-level node (GATE / DOT conv variants) on or off; SISR_FUSED_GROUPS=0 keeps the per-block nodes."""
    return FUSED_GROUPS


def res_block(x, w1, b1, w2, b2, ca=None, m=None, res_scale=1.0):
    """ca = (w1,b1,w2,b2) of the CA squeeze/excite 1x1 convs or None; m = (B,64) meta gate or None."""
    if x.shape[1] != 64 and (ca is not None or m is not None):
        return _wide_gated_block(x, w1, b1, w2, b2, ca, m, float(res_scale))
    caw1, cab1, caw2, cab2 = ca if ca is not None else (None, None, None, None)
    return _ResBlock.apply(x, w1, b1, w2, b2, caw1, cab1, caw2, cab2, m, float(res_scale))


class _ConvReluConv(Function):
    """t2 = alpha * conv2(relu(conv1(x))) without gate/skip (the metadata-mixing QCALayer styles; gated blocks wider than 64
    channels); any multiple of 64 channels."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, alpha):
        B, C, H, W = x.shape
        if C % 64 or tuple(w1.shape) != (C, C, 3, 3) or tuple(w2.shape) != (C, C, 3, 3):
            raise NotImplementedError("fused conv pair needs n_feats to be a multiple of 64")
        x = _cl(x)
        w1, w2 = w1.contiguous(), w2.contiguous()
        v = hip.view_plain(H, W, C)
        t1, t2 = _empty_cl(B, C, H, W, x.device), _empty_cl(B, C, H, W, x.device)
        conv_c64(x, v, pack_weight(w1, "fwd"), b1, (1, 64), t1, v, B, H, W, C, C, relu=True)
        conv_c64(t1, v, pack_weight(w2, "fwd"), b2, (1, 64), t2, v, B, H, W, C, C, alpha=alpha)
        ctx.save_for_backward(x, w1, w2, t1)
        ctx.alpha, ctx.b1, ctx.b2 = alpha, b1, b2
        return t2

    @staticmethod
    def backward(ctx, dt2):
        global IN_BACKWARD
        IN_BACKWARD = True
        try:
            x, w1, w2, t1 = ctx.saved_tensors
            B, C, H, W = x.shape
            dev = x.device
            dt2 = _cl(dt2)
            v = hip.view_plain(H, W, C)
            dt1 = _empty_cl(B, C, H, W, dev)
            conv_c64(dt2, v, pack_weight(w2, "dgrad"), None, (1, 64), dt1, v, B, H, W, C, C, mask=t1, alpha=ctx.alpha)
            dw2, db2 = _grad_buf(w2), _grad_buf_or(ctx.b2, C, dev)
            wgrad_c64(t1, v, dt2, v, dw2, db2, B, H, W, C, C, alpha=ctx.alpha, owner=w2)
            dx = None
            if ctx.needs_input_grad[0]:
                dx = _empty_cl(B, C, H, W, dev)
                conv_c64(dt1, v, pack_weight(w1, "dgrad"), None, (1, 64), dx, v, B, H, W, C, C)
            dw1, db1 = _grad_buf(w1), _grad_buf_or(ctx.b1, C, dev)
            wgrad_c64(x, v, dt1, v, dw1, db1, B, H, W, C, C, owner=w1)
            return dx, dw1, db1, dw2, db2, None
        finally:
            IN_BACKWARD = False


def res_block_convs(x, w1, b1, w2, b2, alpha=1.0):
    return _ConvReluConv.apply(x, w1, b1, w2, b2, float(alpha))


def _wide_gated_block(x, w1, b1, w2, b2, ca, m, res_scale):
    """RCAB / QRCAB('standard') / ParamResBlock at n_feats = 128, 192, 256, ... (ref: attention_manipulators/handlers.py:24-29
    forwards n_feats to QRCAN): the same arithmetic from the modular operators -- conv pair on the multi-chunk MFMA kernels,
    ordered pixel sums, the squeeze / excite MLP on the generic gate kernel (one workgroup per sample), gate * t2 + x --
    instead of the 64-channel fused node."""
    B, C = x.shape[0], x.shape[1]
    t2 = res_block_convs(x, w1, b1, w2, b2, alpha=res_scale)
    if ca is not None:
        caw1, cab1, caw2, cab2 = ca
        pool = global_avg_pool(t2)
        md0 = x.new_zeros((B, 1, 1, 1))  # the plain squeeze / excite stack concatenates no metadata
        g = _GateMlp.apply(pool, md0, m, ([(0, 0, 1), (0, 0, 2)], 0), caw1, cab1, caw2, cab2)
    else:
        g = m
    return gate_mul(t2, g.reshape(B, C, 1, 1), x)


# ----------------------------------------------------------------------------- plain conv(+ReLU) chains (SRMD)
def _pad64(n):
    return (n + 63) // 64 * 64


def _pad_oihw(w, cop, cip):
    """(co, ci, kh, kw) -> zero-padded (cop, cip, kh, kw) copy (the weight itself when nothing is to pad)."""
    co, ci = w.shape[0], w.shape[1]
    if co == cop and ci == cip:
        return w.contiguous()
    taps = w.shape[2] * w.shape[3] if w.dim() == 4 else 1
    out = torch.empty((cop, cip) + tuple(w.shape[2:]), device=w.device, dtype=torch.float32)
    hip.check(hip.lib().sisr_pad_oihw(hip.ptr(w.contiguous()), hip.ptr(out), co, ci, cop, cip, taps, 0, hip.stream()),
              "sisr_pad_oihw")
    return out


def _crop_oihw(wp, shape):
    """Inverse of _pad_oihw for gradients."""
    if tuple(wp.shape) == tuple(shape):
        return wp
    co, ci = shape[0], shape[1] if len(shape) > 1 else 1
    taps = shape[2] * shape[3] if len(shape) == 4 else 1
    out = torch.empty(shape, device=wp.device, dtype=torch.float32)
    hip.check(hip.lib().sisr_pad_oihw(hip.ptr(wp), hip.ptr(out), co, ci, wp.shape[0], wp.shape[1] if wp.dim() > 1 else 1,
                                      taps, 1, hip.stream()), "sisr_pad_oihw(crop)")
    return out


class _PadAxis(Function):
    """Zero-pad ONE axis of a contiguous tensor seen as [outer][n][inner] to [outer][P][inner] (one sisr_pad_oihw launch);
    backward crops.  The building block of pad_param."""

    @staticmethod
    def forward(ctx, t, outer, n, inner, P):
        ctx.in_shape = tuple(t.shape)
        t = t.contiguous()
        ctx.geo = (outer, n, inner, P)
        out = torch.empty(outer * P * inner, device=t.device, dtype=torch.float32)
        hip.check(hip.lib().sisr_pad_oihw(hip.ptr(t), hip.ptr(out), outer, n, outer, P, inner, 0, hip.stream()), "sisr_pad_oihw")
        return out

    @staticmethod
    def backward(ctx, g):
        outer, n, inner, P = ctx.geo
        g = g.contiguous()
        out = torch.empty(outer * n * inner, device=g.device, dtype=torch.float32)
        hip.check(hip.lib().sisr_pad_oihw(hip.ptr(g), hip.ptr(out), outer, n, outer, P, inner, 1, hip.stream()),
                  "sisr_pad_oihw(crop)")
        return out.reshape(ctx.in_shape), None, None, None, None


def pad_param(t, steps, shape):
    """Differentiable zero-padding of a parameter for a network whose feature count is not a multiple of 64 (architectures.
    ChannelPadded): steps = [(outer, n, inner, P)] applied in order, the result viewed as `shape`.  Gradients come back cropped."""
    for outer, n, inner, P in steps:
        t = _PadAxis.apply(t, outer, n, inner, P)
    return t.reshape(shape)


class _ConvChain(Function):
    """y_k = act_k(conv3x3_k(y_{k-1}) + b_k), act = ReLU or identity, as ONE autograd node (ref: the C / CR stacks of
    advanced/SRMD_blocks.py:33-126 behind advanced/architectures.py:380-425 SRMD).  Input: channels-last map whose channel
    count is a multiple of 64 (zero-padded); weights keep their OIHW shapes and are zero-padded to 64-multiples per step;
    the output has the last layer's padded channel count.  Backward: the ReLU mask of layer k-1 is applied by the epilogue
    of layer k's input-gradient conv, so no gradient map is touched by an elementwise pass."""

    @staticmethod
    def forward(ctx, x, relus, *params):
        n = len(relus)
        B, C0, H, W = x.shape
        if C0 % 64:
            raise NotImplementedError("conv chain input must be zero-padded to a multiple of 64 channels")
        dev = x.device
        x = _cl(x)
        _join_pending.clear()
        cur, cin_p = x, C0
        maps, packs, geo = [], [], []
        for k in range(n):
            w, b = params[2 * k], params[2 * k + 1]
            co, ci = w.shape[0], w.shape[1]
            cop = _pad64(co)
            if ci > cin_p:
                raise RuntimeError(f"conv chain layer {k}: weight takes {ci} channels, the map has {cin_p}")
            wp = _pad_oihw(w, cop, cin_p)
            bp = _pad_oihw(b.reshape(co, 1), cop, 1).reshape(cop) if b is not None else None
            pf, pd = pack_pair(wp)
            y = _empty_cl(B, cop, H, W, dev)
            conv_c64(cur, hip.view_plain(H, W, cin_p), pf, bp, (1, 64), y, hip.view_plain(H, W, cop), B, H, W, cin_p, cop,
                     relu=int(relus[k]))  # 0 linear, 1 ReLU, 2 LeakyReLU(0.2)
            maps.append(cur)
            packs.append(pd)
            geo.append((cin_p, cop, tuple(w.shape), b is not None))
            cur, cin_p = y, cop
        if relus[-1]:  # an activated last layer: its mask is applied to the incoming gradient by one elementwise pass
            maps.append(cur)
        ctx.save_for_backward(*maps, *[params[2 * k] for k in range(n)])
        ctx.cfg = (n, tuple(relus), (B, H, W), geo)
        ctx.packs = packs
        return cur

    @staticmethod
    def backward(ctx, dy):
        global IN_BACKWARD
        IN_BACKWARD = True
        try:
            n, relus, (B, H, W), geo = ctx.cfg
            sv = list(ctx.saved_tensors)
            nm = n + (1 if relus[-1] else 0)
            maps, ws = sv[:nm], sv[nm:]
            dev = dy.device
            g = _cl(dy)
            if relus[-1]:  # g <- dy * act'(last output): ReLU' (op 6) or LeakyReLU' (op 3) over the 64-channel blocks
                gm = torch.empty_like(g)
                _map64(maps[n], 64, g, 64, gm, 64, g.numel() // 64, 3 if relus[-1] == 2 else 6)
                g = gm
            side = _side_ok(*ws)
            grads = [None] * (2 * n)
            for k in range(n - 1, -1, -1):
                cin_p, cop, wshape, has_b = geo[k]
                xin = maps[k]
                padded = (cop, cin_p) != (wshape[0], wshape[1])
                dwp = torch.empty((cop, cin_p, 3, 3), device=dev) if padded else _grad_buf(ws[k])
                dbp = torch.empty(cop, device=dev) if has_b else None

                def wg(xin=xin, g=g, dwp=dwp, dbp=dbp, cin_p=cin_p, cop=cop):
                    wgrad_c64(xin, hip.view_plain(H, W, cin_p), g, hip.view_plain(H, W, cop), dwp, dbp, B, H, W, cin_p, cop)

                if side:
                    ev = torch.cuda.Event()
                    ev.record()
                    _on_side(dev, ev, wg, (xin, g, dwp, dbp))
                else:
                    wg()
                if k > 0 or ctx.needs_input_grad[0]:
                    gin = _empty_cl(B, cin_p, H, W, dev)
                    conv_c64(g, hip.view_plain(H, W, cop), ctx.packs[k], None, (1, 64), gin, hip.view_plain(H, W, cin_p), B, H,
                             W, cop, cin_p, mask=(xin if (k > 0 and relus[k - 1]) else None),
                             relu=(LEAKY_MASK if (k > 0 and relus[k - 1] == 2) else 0))
                else:
                    gin = None
                if padded:  # the crop reads what the (possibly side-stream) weight gradient wrote
                    if side:
                        def crop(dwp=dwp, dbp=dbp, k=k, wshape=wshape, has_b=has_b):
                            grads[2 * k] = _crop_oihw(dwp, wshape)
                            grads[2 * k + 1] = _crop_oihw(dbp.reshape(-1, 1), (wshape[0], 1)).reshape(wshape[0]) if has_b else None
                        with torch.cuda.stream(side_stream(dev)):
                            crop()
                            for t in (grads[2 * k], grads[2 * k + 1]):
                                if t is not None:
                                    t.record_stream(torch.cuda.current_stream(dev))
                    else:
                        grads[2 * k] = _crop_oihw(dwp, wshape)
                        grads[2 * k + 1] = _crop_oihw(dbp.reshape(-1, 1), (wshape[0], 1)).reshape(wshape[0]) if has_b else None
                else:
                    grads[2 * k], grads[2 * k + 1] = dwp, dbp
                g = gin
            return (g if ctx.needs_input_grad[0] else None, None, *grads)
        finally:
            IN_BACKWARD = False


def conv_chain(x, layers):
    """layers: [(weight, bias, act)] with act 0 / False linear, 1 / True ReLU, 2 LeakyReLU(0.2) -> channels-last map with the
    last layer's (64-padded) channel count."""
    flat = []
    for w, b, _ in layers:
        flat += [w, b]
    return _ConvChain.apply(x, tuple(int(r) for _, _, r in layers), *flat)


def nchw_to_nhwc_pad(x, cp=None):
    """(B, C, H, W) contiguous NCHW -> channels-last (B, cp, H, W) map, channels >= C zero (no gradient: network input)."""
    if x.requires_grad:
        raise NotImplementedError("nchw_to_nhwc_pad is the network-input layout step and returns no gradient")
    B, C, H, W = x.shape
    cp = _pad64(C) if cp is None else cp
    y = _empty_cl(B, cp, H, W, x.device)
    hip.check(hip.lib().sisr_nchw_to_nhwc_pad(hip.ptr(x.contiguous()), hip.ptr(y), B, C, H, W, cp, hip.stream()),
              "sisr_nchw_to_nhwc_pad")
    return y


# ----------------------------------------------------------------------------- SPARNet pieces (csrc/sparnet.hip)
def _padded_packs(weight, cop, cip, need_dgrad):
    """(forward packing, input-gradient packing | None) of `weight` zero-padded to (cop, cip, 3, 3): the step-level packing
    when the handler has run pack_all (the padding is part of that one launch), else pad + pack here."""
    hit = _step_pack(weight, 1)
    if hit is not None and hit[0].numel() == cop * cip * 9:
        return hit[0], (hit[1] if need_dgrad else None)
    wp = _pad_oihw(weight, cop, cip)
    if need_dgrad:
        return pack_pair(wp)
    return pack_weight(wp, "fwd"), None


# SPARNet's stride-1 ConvLayer convs with reflection / nearest upsampling inside the MFMA kernels' staging (0: the gather ->
# conv -> gather composition of round 3, kept for the stride-2 convs and as the A/B reference; results are bit-identical)
REFL_GEO = os.environ.get("SISR_REFL_GEO", "1") != "0"
REFL_BATCH = os.environ.get("SISR_REFL_BATCH", "1") != "0"  # their small weight gradients eight per launch (WgradGeoQueue)
# ... or on the side stream (captured: a parallel branch).  Measured slower on MI355X, per launch (459 vs 500 images/s) and
# forked sixteen at a time (467): off
REFL_SIDE = os.environ.get("SISR_REFL_SIDE", "0") != "0"
REFL_SIDE_GROUP = int(os.environ.get("SISR_REFL_SIDE_GROUP", 16))  # ... forked this many at a time


class _ReflConv(Function):
    """The conv of one reference ConvLayer as one autograd node (ref: SPARNet/blocks.py:69-103):
    [nearest x2] -> ReflectionPad2d(1) -> Conv2d(3x3, stride 1 | 2, no padding).  x: channels-last map zero-padded to a
    multiple of 64 channels; weight / bias keep their reference shapes (zero-padded per call); y: (B, pad64(cout), Ho, Wo).
    The padded, upsampled map is materialised by one gather; the zero-padded MFMA conv over it equals the reference's conv on
    its interior, which a second gather takes (every stride-th pixel).  Backward: the adjoint gathers around the MFMA
    input-gradient / weight-gradient kernels on the padded geometry."""

    @staticmethod
    def forward(ctx, x, weight, bias, up, stride):
        B, Cp, H, W = x.shape
        co, ci = weight.shape[0], weight.shape[1]
        if Cp % 64 or ci > Cp or tuple(weight.shape[2:]) != (3, 3):
            raise NotImplementedError(f"reflection-padded conv: 3x3 weights on a 64-multiple map (got {tuple(weight.shape)} on {Cp})")
        if H < 2 or W < 2:
            raise NotImplementedError("ReflectionPad2d(1) needs at least 2 x 2 pixels")
        dev, L = x.device, hip.lib()
        x = _cl(x)
        cop = _pad64(co)
        ctx.geo = REFL_GEO and stride == 1 and up in (1, 2) and PRECISION == "fp32"
        if ctx.geo:
            # reflection / nearest upsampling as address arithmetic of the conv's own staging: no padded copy, no ring of
            # throw-away outputs, no crop (csrc/conv3x3_mfma.hip GEO); channels >= ci of a 64-channel input are zero padding
            # and their octets of the K loop are skipped
            Hv, Wv = up * H, up * W
            pf, ctx.pd = _padded_packs(weight, cop, Cp, ctx.needs_input_grad[0])
            bp = _pad_oihw(bias.reshape(co, 1), cop, 1).reshape(cop) if bias is not None else None
            y = _empty_cl(B, cop, Hv, Wv, dev)
            hip.check(L.sisr_conv3x3_c64_geo(hip.ptr(x), hip.view_plain(H, W, Cp), _wptr(pf), hip.ptr(bp), hip.ptr(y),
                                             hip.view_plain(Hv, Wv, cop), None, B, Hv, Wv, Cp, cop, 1, up - 1,
                                             ci if Cp == 64 else 0, hip.stream()), "sisr_conv3x3_c64_geo")
            ctx.save_for_backward(x, weight)
            ctx.bias = bias
            ctx.geom = (B, H, W, Cp, cop, Hv, Wv, up, stride)
            return y
        Hp, Wp = up * H + 2, up * W + 2
        ctx.s2 = REFL_GEO and stride == 2 and up == 1 and PRECISION == "fp32"
        if ctx.s2:
            # the stride-2 conv computed at its output pixels (a quarter of the stride-1 arithmetic, no gathers).  The backward
            # pass still runs on the padded geometry: it rebuilds the padded map there instead of keeping it
            pf, ctx.pd = _padded_packs(weight, cop, Cp, ctx.needs_input_grad[0])
            bp = _pad_oihw(bias.reshape(co, 1), cop, 1).reshape(cop) if bias is not None else None
            Ho, Wo = (H + 1) // 2, (W + 1) // 2
            y = _empty_cl(B, cop, Ho, Wo, dev)
            hip.check(L.sisr_conv3x3_c64_geo(hip.ptr(x), hip.view_plain(H, W, Cp), _wptr(pf), hip.ptr(bp), hip.ptr(y),
                                             hip.view_plain(Ho, Wo, cop), None, B, H, W, Cp, cop, 3, 0,
                                             ci if Cp == 64 else 0, hip.stream()), "sisr_conv3x3_c64_geo(stride 2)")
            ctx.save_for_backward(x, weight)
            ctx.bias = bias
            ctx.geom = (B, H, W, Cp, cop, Hp, Wp, up, stride)
            return y
        xp = _empty_cl(B, Cp, Hp, Wp, dev)
        hip.check(L.sisr_pad_reflect_up(hip.ptr(x), hip.ptr(xp), B, H, W, Cp, up, 0, hip.stream()), "sisr_pad_reflect_up")
        wp = _pad_oihw(weight, cop, Cp)
        bp = _pad_oihw(bias.reshape(co, 1), cop, 1).reshape(cop) if bias is not None else None
        if ctx.needs_input_grad[0]:
            pf, ctx.pd = pack_pair(wp)
        else:
            pf, ctx.pd = pack_weight(wp, "fwd"), None
        yf = _empty_cl(B, cop, Hp, Wp, dev)
        conv_c64(xp, hip.view_plain(Hp, Wp, Cp), pf, bp, (1, 64), yf, hip.view_plain(Hp, Wp, cop), B, Hp, Wp, Cp, cop)
        Ho, Wo = (Hp - 3) // stride + 1, (Wp - 3) // stride + 1
        y = _empty_cl(B, cop, Ho, Wo, dev)
        hip.check(L.sisr_crop_stride(hip.ptr(yf), hip.ptr(y), B, Hp, Wp, cop, stride, 0, hip.stream()), "sisr_crop_stride")
        ctx.save_for_backward(xp, weight)
        ctx.bias = bias
        ctx.geom = (B, H, W, Cp, cop, Hp, Wp, up, stride)
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, weight = ctx.saved_tensors
        B, H, W, Cp, cop, Hp, Wp, up, stride = ctx.geom
        dev, L = dy.device, hip.lib()
        dy = _cl(dy)
        if ctx.geo or ctx.s2:
            x, (co, ci) = xp, tuple(weight.shape[:2])
            if ctx.geo:    # stride 1: saved geometry holds the conv's (= dy's) grid
                Hv, Wv, Hd, Wd, tmode, wup = Hp, Wp, Hp, Wp, 2, up - 1
            else:          # stride 2: the grid is the input's; dy has every second pixel of it and is read zero-stuffed
                Hv, Wv, Hd, Wd, tmode, wup = H, W, (H + 1) // 2, (W + 1) // 2, 4, 2
            dx = dw = db = None
            if ctx.needs_input_grad[0]:
                # gradient of the padded map: the transposed conv, two pixels larger than the grid, straight from dy (no embedding
                # into a zero map); then the reflection / upsampling folded back
                dxp = _empty_cl(B, Cp, Hv + 2, Wv + 2, dev)
                hip.check(L.sisr_conv3x3_c64_geo(hip.ptr(dy), hip.view_plain(Hd, Wd, cop), _wptr(ctx.pd), None, hip.ptr(dxp),
                                                 hip.view_plain(Hv + 2, Wv + 2, Cp), None, B, Hv + 2, Wv + 2, cop, Cp, tmode, 0,
                                                 co if cop == 64 else 0, hip.stream()), "sisr_conv3x3_c64_geo(transposed)")
                dx = _empty_cl(B, Cp, H, W, dev)
                hip.check(L.sisr_pad_reflect_up(hip.ptr(dxp), hip.ptr(dx), B, H, W, Cp, up, 1, hip.stream()), "sisr_pad_reflect_up(adjoint)")
            has_b = ctx.bias is not None
            if ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2]):
                dw = _grad_buf(weight)
                db = _grad_buf(ctx.bias) if has_b else None
                # 32 x 32-channel blocks of the gradient that hold real channels (the others: zero padding of the maps)
                units, n_in, n_out = 0, Cp // 64, cop // 64
                for cc in range(n_in):
                    for cq in range(n_out):
                        for cih in range(2):
                            for coh in range(2):
                                if cc * 64 + cih * 32 < ci and cq * 64 + coh * 32 < co:
                                    units |= 1 << ((cc * n_out + cq) * 4 + cih * 2 + coh)
                if units == (1 << (n_in * n_out * 4)) - 1 or n_in * n_out * 4 > 64:
                    units = 0
                # small launches (fewer 8 x 32-pixel tiles than half the CUs) inside a deferred_wgrads() block: queued, eight per
                # launch when the block closes
                tiles = B * ((Hv + 7) // 8) * ((Wv + 31) // 32) * n_in * n_out
                owners = (weight, ctx.bias) if has_b else (weight,)
                if (REFL_BATCH and _DEFERRED is not None and hip.stream() == _DEFERRED_STREAM and tiles < 128 and
                        all(o.requires_grad and o.grad is None for o in owners)):
                    if id(weight) in _DEFERRED_OWNERS:
                        _flush_deferred()  # second gradient of one weight in this pass: the first must be written first
                    else:
                        _DEFERRED_OWNERS.add(id(weight))
                        q = _DEFERRED.get(("geo", dev.index))
                        if q is None:
                            q = _DEFERRED[("geo", dev.index)] = WgradGeoQueue(dev)
                        q.add(x, dy, dw, db, (B, Hv, Wv, Cp, cop, wup, co, ci), units, tiles)
                        return dx, dw, db, None, None
                nbytes = L.sisr_wgrad3x3_c64_workspace_bytes(B, Hv, Wv, Cp, cop)

                def wgrad():  # (the workspace is per stream: taken where the launch is issued)
                    ws = hip.workspace(dev, nbytes)
                    hip.check(L.sisr_wgrad3x3_c64_geo(hip.ptr(x), hip.view_plain(H, W, Cp), hip.ptr(dy), hip.view_plain(Hd, Wd, cop),
                                                      hip.ptr(dw), co, ci, hip.ptr(db), hip.ptr(ws), nbytes, B, Hv, Wv, Cp, cop,
                                                      wup, units, hip.stream()), "sisr_wgrad3x3_c64_geo")

                # The maps of this network are far smaller than the chip (most launches: 16 - 80 workgroups on 256 CUs): the
                # weight gradients, which nothing on the input-gradient chain waits for, fill the idle CUs from a second stream
                # (a parallel branch of the captured step), joined once at the end of the backward pass.
                if REFL_SIDE and _side_ok(*((weight, ctx.bias) if has_b else (weight,)), fork=True):
                    _on_side_grouped(dev, wgrad, (x, dy, dw, db), REFL_SIDE_GROUP)
                else:
                    wgrad()
            return dx, dw, db, None, None
        dyf = _empty_cl(B, cop, Hp, Wp, dev)
        hip.check(L.sisr_crop_stride(hip.ptr(dy), hip.ptr(dyf), B, Hp, Wp, cop, stride, 1, hip.stream()), "sisr_crop_stride(embed)")
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dxp = _empty_cl(B, Cp, Hp, Wp, dev)
            conv_c64(dyf, hip.view_plain(Hp, Wp, cop), ctx.pd, None, (1, 64), dxp, hip.view_plain(Hp, Wp, Cp), B, Hp, Wp, cop, Cp)
            dx = _empty_cl(B, Cp, H, W, dev)
            hip.check(L.sisr_pad_reflect_up(hip.ptr(dxp), hip.ptr(dx), B, H, W, Cp, up, 1, hip.stream()), "sisr_pad_reflect_up(adjoint)")
        if ctx.needs_input_grad[1] or (ctx.bias is not None and ctx.needs_input_grad[2]):
            padded = (cop, Cp) != tuple(weight.shape[:2])
            dwp = torch.empty((cop, Cp, 3, 3), device=dev) if padded else _grad_buf(weight)
            has_b = ctx.bias is not None
            dbp = (torch.empty(cop, device=dev) if padded else _grad_buf(ctx.bias)) if has_b else None
            wgrad_c64(xp, hip.view_plain(Hp, Wp, Cp), dyf, hip.view_plain(Hp, Wp, cop), dwp, dbp, B, Hp, Wp, Cp, cop)
            dw = _crop_oihw(dwp, tuple(weight.shape))
            if has_b:
                co = weight.shape[0]
                db = _crop_oihw(dbp.reshape(-1, 1), (co, 1)).reshape(co)
        return dx, dw, db, None, None


def refl_conv(x, weight, bias=None, up=1, stride=1):
    return _ReflConv.apply(x, weight, bias, int(up), int(stride))


class _BatchNormAct(Function):
    """LeakyReLU(slope)(BatchNorm2d(x)) on a channels-last map whose channels >= C_real are zero padding (ref: NormLayer +
    ReluLayer, SPARNet/blocks.py:10-66; slope 1 = no activation).  training: batch statistics, running statistics updated in
    place as torch does; else the running statistics (forward only)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, slope):
        B, C, H, W = x.shape
        Cr = gamma.shape[0]
        if C % 64 or Cr > C or C > 256:
            raise NotImplementedError("batch norm kernels: a 64-multiple map of at most 256 channels")
        dev, L = x.device, hip.lib()
        x = _cl(x)
        y = torch.empty_like(x)
        npix = B * H * W
        nbytes = L.sisr_bn_workspace_bytes(npix, C)
        ws = hip.workspace(dev, nbytes)
        mean, invstd = (torch.empty(C, device=dev), torch.empty(C, device=dev)) if training else (None, None)
        g, b = gamma.contiguous(), beta.contiguous()
        hip.check(L.sisr_bn_act_fwd(hip.ptr(x), hip.ptr(y), hip.ptr(g), hip.ptr(b), hip.ptr(running_mean),
                                    hip.ptr(running_var), hip.ptr(mean), hip.ptr(invstd), npix, C, Cr, int(training),
                                    float(momentum), float(eps), float(slope), hip.ptr(ws), nbytes, hip.stream()),
                  "sisr_bn_act_fwd")
        ctx.training, ctx.slope, ctx.geom = training, slope, (npix, C, Cr)
        if training:
            ctx.save_for_backward(x, g, b, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise NotImplementedError("batch norm backward in eval mode (running statistics) is not built: the reference "
                                      "trains in train() mode and evaluates under no_grad")
        x, g, b, mean, invstd = ctx.saved_tensors
        npix, C, Cr = ctx.geom
        dev, L = dy.device, hip.lib()
        dy = _cl(dy)
        dx = torch.empty_like(x)
        dg, db = torch.empty(Cr, device=dev), torch.empty(Cr, device=dev)
        nbytes = L.sisr_bn_workspace_bytes(npix, C)
        ws = hip.workspace(dev, nbytes)
        hip.check(L.sisr_bn_act_bwd(hip.ptr(x), hip.ptr(dy), hip.ptr(g), hip.ptr(b), hip.ptr(mean), hip.ptr(invstd),
                                    hip.ptr(dx), hip.ptr(dg), hip.ptr(db), npix, C, Cr, float(ctx.slope), hip.ptr(ws), nbytes,
                                    hip.stream()), "sisr_bn_act_bwd")
        return dx, dg, db, None, None, None, None, None, None


def batch_norm_act(x, bn, slope=1.0):
    """bn: an nn.BatchNorm2d (parameter / buffer holder); slope: LeakyReLU slope folded in (1 = none)."""
    training = bn.training or bn.running_mean is None
    if bn.training and bn.num_batches_tracked is not None and not getattr(bn, "_sisr_counted_by_net", False):
        bn.num_batches_tracked += 1  # as nn.BatchNorm2d.forward does (momentum is a number here: the average is exponential)
    if bn.momentum is None:
        raise NotImplementedError("BatchNorm2d(momentum=None) (cumulative average) is not built")
    return _BatchNormAct.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bool(training), float(bn.momentum),
                               float(bn.eps), float(slope))


class _NearestUp(Function):
    """nn.Upsample(scale_factor=up, mode='nearest') on a channels-last map (ref: advanced/SRMD_blocks.py:58-63)."""

    @staticmethod
    def forward(ctx, x, up):
        B, C, H, W = x.shape
        x = _cl(x)
        y = _empty_cl(B, C, H * up, W * up, x.device)
        hip.check(hip.lib().sisr_nearest_up(hip.ptr(x), hip.ptr(y), B, H, W, C, up, 0, hip.stream()), "sisr_nearest_up")
        ctx.geom = (B, C, H, W, up)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W, up = ctx.geom
        dy = _cl(dy)
        dx = _empty_cl(B, C, H, W, dy.device)
        hip.check(hip.lib().sisr_nearest_up(hip.ptr(dy), hip.ptr(dx), B, H, W, C, up, 1, hip.stream()), "sisr_nearest_up(adjoint)")
        return dx, None


def nearest_up(x, up):
    return _NearestUp.apply(x, int(up))


def instance_norm_act(x, norm, slope=1.0):
    """LeakyReLU(slope)(InstanceNorm2d(x)) (ref: advanced/SRMD_blocks.py:44-45 'I': nn.InstanceNorm2d(affine=True), statistics per
    sample and channel over H x W, in train() and eval() alike -- it tracks no running statistics).  Instance norm of a batch =
    batch norm of each sample on its own: the batch-norm kernels run once per sample (the affine parameters' gradients add up
    over the samples in autograd), and the normalised samples are concatenated."""
    if getattr(norm, "track_running_stats", False):
        raise NotImplementedError("InstanceNorm2d(track_running_stats=True) is not built")
    eps = float(norm.eps)
    outs = [_BatchNormAct.apply(x[b:b + 1], norm.weight, norm.bias, None, None, True, 0.0, eps, float(slope))
            for b in range(x.shape[0])]
    return outs[0] if len(outs) == 1 else torch.cat(outs, 0)


class _SparCombine(Function):
    """out = identity + x * sigmoid(logits[:, 0])  (ref: HourGlassBlock.forward, SPARNet/blocks.py:236-243, and the residual sum
    of ResidualBlock.forward :166).  logits: the 64 -> 1 attention conv's zero-padded 64-channel result."""

    @staticmethod
    def forward(ctx, x, logits, identity):
        B, C, H, W = x.shape
        dev, L = x.device, hip.lib()
        x, logits = _cl(x), _cl(logits)
        idn = _cl(identity) if identity is not None else None
        y = torch.empty_like(x)
        att = torch.empty(B * H * W, device=dev)
        hip.check(L.sisr_spar_combine_fwd(hip.ptr(x), hip.ptr(logits), hip.ptr(idn), hip.ptr(y), hip.ptr(att), B * H * W, C,
                                          logits.shape[1], hip.stream()), "sisr_spar_combine_fwd")
        ctx.save_for_backward(x, att)
        ctx.cl, ctx.has_idn = logits.shape[1], identity is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, att = ctx.saved_tensors
        B, C, H, W = x.shape
        dev, L = dy.device, hip.lib()
        dy = _cl(dy)
        dx = torch.empty_like(x)
        dl = _empty_cl(B, ctx.cl, H, W, dev)
        hip.check(L.sisr_spar_combine_bwd(hip.ptr(dy), hip.ptr(x), hip.ptr(att), hip.ptr(dx), hip.ptr(dl), B * H * W, C, ctx.cl,
                                          hip.stream()), "sisr_spar_combine_bwd")
        return dx, dl, (dy if ctx.has_idn else None)


def spar_combine(x, logits, identity=None):
    return _SparCombine.apply(x, logits, identity)


# ---- SPARNet's non-default ConvLayer options (csrc/sparnet.hip, "non-default ConvLayer options")
class _GroupNorm(Function):
    """InstanceNorm2d(affine) (cg = 1) / GroupNorm(32, C) (cg = C / 32) on a channels-last map whose channels >= C_real are zero
    padding (ref: SPARNet/blocks.py:21-24)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, cg, eps):
        B, C, H, W = x.shape
        Cr = gamma.shape[0]
        if Cr > C or Cr % cg:
            raise NotImplementedError(f"group statistics over groups of {cg} channels of {Cr}")
        dev, L = x.device, hip.lib()
        x = _cl(x)
        y = torch.empty_like(x)
        mean, inv = torch.empty(B * (Cr // cg), device=dev), torch.empty(B * (Cr // cg), device=dev)
        g, b = gamma.contiguous(), beta.contiguous()
        hip.check(L.sisr_group_norm_fwd(hip.ptr(x), hip.ptr(y), hip.ptr(g), hip.ptr(b), hip.ptr(mean), hip.ptr(inv), B, H * W, C, Cr,
                                        cg, float(eps), hip.stream()), "sisr_group_norm_fwd")
        ctx.save_for_backward(x, g, mean, inv)
        ctx.geom = (B, C, H, W, Cr, cg)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, mean, inv = ctx.saved_tensors
        B, C, H, W, Cr, cg = ctx.geom
        dev, L = dy.device, hip.lib()
        dy = _cl(dy)
        dx = torch.empty_like(x)
        dgb, dbb = torch.empty((B, Cr), device=dev), torch.empty((B, Cr), device=dev)
        hip.check(L.sisr_group_norm_bwd(hip.ptr(x), hip.ptr(dy), hip.ptr(g), hip.ptr(mean), hip.ptr(inv), hip.ptr(dx), hip.ptr(dgb),
                                        hip.ptr(dbb), B, H * W, C, Cr, cg, hip.stream()), "sisr_group_norm_bwd")
        dg, db = torch.empty(Cr, device=dev), torch.empty(Cr, device=dev)
        for part, out in ((dgb, dg), (dbb, db)):  # the samples' sums added in batch order
            hip.check(L.sisr_sum_partials(hip.ptr(part), B, 1, Cr, 1.0, hip.ptr(out), hip.stream()), "sisr_sum_partials")
        return dx, dg, db, None, None


def group_norm(x, gamma, beta, cg, eps=1e-5):
    return _GroupNorm.apply(x, gamma, beta, int(cg), float(eps))


class _PixelNorm(Function):
    """F.normalize(x, p = 2, dim = 1) on a channels-last map (ref: SPARNet/blocks.py:25-26)."""

    @staticmethod
    def forward(ctx, x):
        B, C, H, W = x.shape
        x = _cl(x)
        y = torch.empty_like(x)
        hip.check(hip.lib().sisr_pixel_norm(hip.ptr(x), None, hip.ptr(y), B * H * W, C, 0, hip.stream()), "sisr_pixel_norm")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        B, C, H, W = x.shape
        dy = _cl(dy)
        dx = torch.empty_like(x)
        hip.check(hip.lib().sisr_pixel_norm(hip.ptr(x), hip.ptr(dy), hip.ptr(dx), B * H * W, C, 1, hip.stream()), "sisr_pixel_norm(bwd)")
        return dx


def pixel_norm(x):
    return _PixelNorm.apply(x)


class _Act(Function):
    """PReLU(C) (mode 0, slope: the module's weight) / SELU (mode 1) on a channels-last map (ref: SPARNet/blocks.py:55-58)."""

    @staticmethod
    def forward(ctx, x, slope, mode):
        B, C, H, W = x.shape
        x = _cl(x)
        y = torch.empty_like(x)
        a = slope.contiguous() if slope is not None else None
        if a is not None and a.shape[0] > C:
            raise NotImplementedError("PReLU slope count above the map's channels")
        if a is not None and a.shape[0] == 1 and C != 1:
            a = a.expand(C).contiguous()  # nn.PReLU(1): one slope for every channel
        hip.check(hip.lib().sisr_act(hip.ptr(x), None, hip.ptr(a), hip.ptr(y), None, B * H * W, C, a.shape[0] if a is not None else C,
                                     mode, 0, hip.stream()), "sisr_act")
        ctx.save_for_backward(x, a)
        ctx.mode, ctx.n_slope = mode, (slope.shape[0] if slope is not None else 0)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, a = ctx.saved_tensors
        B, C, H, W = x.shape
        dev, L = dy.device, hip.lib()
        dy = _cl(dy)
        dx = torch.empty_like(x)
        dyx = torch.empty_like(x) if ctx.mode == 0 else None
        hip.check(L.sisr_act(hip.ptr(x), hip.ptr(dy), hip.ptr(a), hip.ptr(dx), hip.ptr(dyx), B * H * W, C,
                             a.shape[0] if a is not None else C, ctx.mode, 1, hip.stream()), "sisr_act(bwd)")
        da = None
        if ctx.mode == 0 and ctx.needs_input_grad[1]:
            part, parts = _pixel_sums(dyx, None, B, H, W, C)  # [B][parts][C] ordered sums of dy * min(x, 0)
            per = torch.empty(C, device=dev)
            hip.check(L.sisr_sum_partials(hip.ptr(part), B * parts, 1, C, 1.0, hip.ptr(per), hip.stream()), "sisr_sum_partials")
            da = per[:a.shape[0]] if ctx.n_slope != 1 else per[:a.shape[0]].sum().reshape(1)
        return dx, da, None


def prelu(x, slope):
    return _Act.apply(x, slope, 0)


def selu(x):
    return _Act.apply(x, None, 1)


_const_slopes = {}


def leaky_relu(x, slope):
    """LeakyReLU(slope) (0: ReLU) as the PReLU kernel with a constant slope vector -- the activation behind a norm other than
    batch norm (behind batch norm it is part of the batch-norm kernel)."""
    key = (x.device.index, x.shape[1], float(slope))
    a = _const_slopes.get(key)
    if a is None:
        a = _const_slopes[key] = torch.full((x.shape[1],), float(slope), device=x.device)
    return _Act.apply(x, a, 0)


class _Spar3d(Function):
    """identity + x * sigmoid(logits), one logit per element (ref: SPARNet/blocks.py:147-151 att_name 'spar3d', :241-243)."""

    @staticmethod
    def forward(ctx, x, logits, identity):
        x, logits = _cl(x), _cl(logits)
        idn = _cl(identity) if identity is not None else None
        y = torch.empty_like(x)
        hip.check(hip.lib().sisr_spar3d(hip.ptr(x), hip.ptr(logits), hip.ptr(idn), hip.ptr(y), None, x.numel(), 0, hip.stream()),
                  "sisr_spar3d")
        ctx.save_for_backward(x, logits)
        ctx.has_idn = identity is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, logits = ctx.saved_tensors
        dy = _cl(dy)
        dx, dl = torch.empty_like(x), torch.empty_like(x)
        hip.check(hip.lib().sisr_spar3d(hip.ptr(x), hip.ptr(logits), hip.ptr(dy), hip.ptr(dx), hip.ptr(dl), x.numel(), 1, hip.stream()),
                  "sisr_spar3d(bwd)")
        return dx, dl, (dy if ctx.has_idn else None)


def spar_combine3d(x, logits, identity=None):
    return _Spar3d.apply(x, logits, identity)


class _ShuffleRGB(Function):
    """PixelShuffle(r) of the first C r^2 channels of a channels-last map into an NCHW (B, C, rH, rW) image."""

    @staticmethod
    def forward(ctx, y, C, r):
        B, Cp, H, W = y.shape
        y = _cl(y)
        out = torch.empty((B, C, H * r, W * r), device=y.device, dtype=torch.float32)
        hip.check(hip.lib().sisr_shuffle_rgb(hip.ptr(y), hip.ptr(out), B, C, r, H, W, Cp, 0, hip.stream()), "sisr_shuffle_rgb")
        ctx.geo = (B, C, r, H, W, Cp)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, C, r, H, W, Cp = ctx.geo
        dy = _empty_cl(B, Cp, H, W, dout.device)
        hip.check(hip.lib().sisr_shuffle_rgb(hip.ptr(dout.contiguous()), hip.ptr(dy), B, C, r, H, W, Cp, 1, hip.stream()),
                  "sisr_shuffle_rgb(adjoint)")
        return dy, None, None


def shuffle_rgb(y, channels, r):
    return _ShuffleRGB.apply(y, int(channels), int(r))


# ----------------------------------------------------------------------------- SFTMD (csrc/sft.hip)
LEAKY, LEAKY_MASK = 2, 4  # `relu` codes of sisr_conv3x3_c64: LeakyReLU(0.2) epilogue / the mask is its derivative
SPARSE_BLOCK_DIAGONAL, SPARSE_SECOND_CHUNK, SPARSE_HALVES = 8, 9, 10  # `select` codes: structural zeros of the merged SFT weights are skipped


def _fp32_only(what):
    if PRECISION != "fp32":
        raise NotImplementedError(f"{what} runs on the fp32 kernels only (SISR_PRECISION={PRECISION})")


def _sft_compose(params, merged, M, split=0):
    """params: the eight parameter (or gradient) tensors of a StandardSft; merged: (WA, bA, WB, bB)."""
    hip.check(hip.lib().sisr_sft_compose(*[hip.ptr_c(t) for t in params], *[hip.ptr(t) for t in merged], int(M), int(split),
                                         hip.stream()), "sisr_sft_compose")


def _map64(a, a_stride, b, b_stride, out, out_stride, npix, op, a_off=0, out_off=0):
    """op 0 copy / 1 a + b / 2 LeakyReLU(a) / 3 b * LeakyReLU'(a) on 64-channel maps with pixel strides (floats)."""
    hip.check(hip.lib().sisr_map64(hip.ptr(a) + 4 * a_off, a_stride, hip.ptr(b), b_stride, hip.ptr(out) + 4 * out_off,
                                   out_stride, npix, op, hip.stream()), "sisr_map64")


class _SftLayer(Function):
    """out = [relu](x * sigmoid(mul(cat)) + add(cat)), cat = (x, metadata maps); mul / add = conv3x3 -> LeakyReLU(0.2) ->
    conv3x3 with 32 hidden channels each (ref: SFTMD_variants/architectures.py:25-56 StandardSft; the F.relu of
    SFT_Residual_Block.forward :100-102 folded in).  Two MFMA convs instead of four: A = [mul_conv1 | add_conv1] merged
    along the outputs (128 -> 64, LeakyReLU epilogue), B = block-diagonal [mul_conv2, add_conv2] (64 -> 128); the merged
    weights are composed per call from the four parameters and their gradients split back.  md: channels-last
    [B, 64, H, W], the M metadata maps zero-padded (no gradient); torch.cat((x, maps)) is never materialised: conv A and its
    weight gradient read the pair through a view whose second chunk points at md (hip.view_pair)."""

    @staticmethod
    def forward(ctx, x, md, relu, M, mw1, mb1, aw1, ab1, mw2, mb2, aw2, ab2):
        _fp32_only("SFT layer")
        B, C, H, W = x.shape
        if C != 64 or md.shape != (B, 64, H, W) or tuple(mw1.shape) != (32, 64 + M, 3, 3) or tuple(mw2.shape) != (64, 32, 3, 3):
            raise NotImplementedError("SFT layer: 64 features, 32 hidden channels, at most 64 metadata maps")
        dev, npix = x.device, B * H * W
        x, md = _cl(x), _cl(md)
        # cat(x, maps) is never built: the conv reads chunk 0 from x and chunk 1 from the (shared) metadata map through the
        # view's chunk offset -- two allocations, one logical 128-channel input
        diff = md.data_ptr() - x.data_ptr()
        if diff % 16:
            raise RuntimeError("SFT layer: feature and metadata maps must be 16-byte aligned relative to each other")
        vcat = hip.view_pair(H, W, diff // 4)
        hit = _SFT_STEP.get(id(mw1))
        if hit is not None and hit[0] is mw1:  # composed (and packed) once for the whole step by pack_all
            WA, bA, WB, bB = hit[1:]
        else:
            WA = torch.empty((64, 128, 3, 3), device=dev)
            bA = torch.empty(64, device=dev)
            WB = torch.empty((128, 64, 3, 3), device=dev)
            bB = torch.empty(128, device=dev)
            _sft_compose((mw1, mb1, aw1, ab1, mw2, mb2, aw2, ab2), (WA, bA, WB, bB), M)
        pfA, pdA = pack_pair(WA)
        pfB, pdB = pack_pair(WB)
        t = _empty_cl(B, 64, H, W, dev)
        conv_c64(x, vcat, pfA, bA, (1, 64), t, hip.view_plain(H, W, 64), B, H, W, 128, 64, relu=LEAKY,
                 select=SPARSE_SECOND_CHUNK if M <= 16 else 0)
        y2 = _empty_cl(B, 128, H, W, dev)
        conv_c64(t, hip.view_plain(H, W, 64), pfB, bB, (1, 64), y2, hip.view_plain(H, W, 128), B, H, W, 64, 128,
                 select=SPARSE_BLOCK_DIAGONAL)
        out = _empty_cl(B, 64, H, W, dev)
        hip.check(hip.lib().sisr_sft_combine_fwd(hip.ptr(x), 64, hip.ptr(y2), None, hip.ptr(out), 64, npix, int(relu),
                                                 hip.stream()), "sisr_sft_combine_fwd")
        ctx.save_for_backward(x, md, t, y2)
        ctx.packs = (pdA, pdB)
        ctx.cfg = (B, H, W, int(relu), M)
        return out

    @staticmethod
    def backward(ctx, dout):
        global IN_BACKWARD
        IN_BACKWARD = True
        try:
            x, md, t, y2 = ctx.saved_tensors
            pdA, pdB = ctx.packs
            B, H, W, relu, M = ctx.cfg
            vcat = hip.view_pair(H, W, (md.data_ptr() - x.data_ptr()) // 4)
            dev, npix = dout.device, B * H * W
            dout = _cl(dout)
            v64, v128 = hip.view_plain(H, W, 64), hip.view_plain(H, W, 128)
            dx0 = _empty_cl(B, 64, H, W, dev)
            dy2 = _empty_cl(B, 128, H, W, dev)
            hip.check(hip.lib().sisr_sft_combine_bwd(hip.ptr(dout), 64, hip.ptr(x), 64, hip.ptr(y2), hip.ptr(dx0),
                                                     hip.ptr(dy2), npix, relu, hip.stream()), "sisr_sft_combine_bwd")
            dWB = torch.empty((128, 64, 3, 3), device=dev)
            dbB = torch.empty(128, device=dev)
            wgrad_c64(t, v64, dy2, v128, dWB, dbB, B, H, W, 64, 128, active_units=0xC3)  # the two diagonal 64 x 32 blocks
            dt = _empty_cl(B, 64, H, W, dev)
            conv_c64(dy2, v128, pdB, None, (1, 64), dt, v64, B, H, W, 128, 64, mask=t, relu=LEAKY_MASK, select=SPARSE_HALVES)
            dWA = torch.empty((64, 128, 3, 3), device=dev)
            dbA = torch.empty(64, device=dev)
            # input channels >= 96 are padding when M <= 32: the last ci half of the second chunk is never read back
            wgrad_c64(x, vcat, dt, v64, dWA, dbA, B, H, W, 128, 64, active_units=0x3F if M <= 32 else 0)
            dx = None
            if ctx.needs_input_grad[0]:
                # input gradient of A for the 64 feature channels only (output chunk 0 of the packing) + the direct term
                dx = _empty_cl(B, 64, H, W, dev)
                conv_c64(dt, v64, pdA, None, (1, 64), dx, v64, B, H, W, 64, 64, res=dx0)
            g = [torch.empty(s, device=dev) for s in ((32, 64 + M, 3, 3), (32,), (32, 64 + M, 3, 3), (32,), (64, 32, 3, 3),
                                                      (64,), (64, 32, 3, 3), (64,))]
            _sft_compose(g, (dWA, dbA, dWB, dbB), M, split=1)
            return (dx, None, None, None, *g)
        finally:
            IN_BACKWARD = False


def sft_layer(x, md, module, relu):
    """module: sftmd.StandardSft (parameter holder)."""
    M = module.mul_conv1.weight.shape[1] - 64
    return _SftLayer.apply(x, md, bool(relu), int(M), *module.params())


class _Map64(Function):
    """Elementwise pieces of the non-default SFT types on channels-last 64-channel maps (csrc/sft.hip map64):
    mode 'relu': relu(x); 'mul': x * md (WeakSft, 64 maps); 'mul0': x * md[:, 0:1] (WeakSft, one map).  md has no gradient."""

    @staticmethod
    def forward(ctx, x, md, mode):
        B, C, H, W = x.shape
        if C != 64:
            raise NotImplementedError("64-channel maps")
        x = _cl(x)
        out = _empty_cl(B, 64, H, W, x.device)
        npix = B * H * W
        if mode == "relu":
            _map64(x, 64, None, 0, out, 64, npix, 5)
            ctx.save_for_backward(out)
        else:
            md = _cl(md)
            _map64(x, 64, md, 64, out, 64, npix, 4 if mode == "mul" else 7)
            ctx.save_for_backward(md)
        ctx.mode = mode
        return out

    @staticmethod
    def backward(ctx, dy):
        (t,) = ctx.saved_tensors
        B, C, H, W = dy.shape
        dy = _cl(dy)
        dx = _empty_cl(B, 64, H, W, dy.device)
        if ctx.mode == "relu":
            _map64(t, 64, dy, 64, dx, 64, B * H * W, 6)
        else:
            _map64(dy, 64, t, 64, dx, 64, B * H * W, 4 if ctx.mode == "mul" else 7)
        return dx, None, None


class _ConcatSft(Function):
    """conv3x3(cat(x, maps)) (ref: SFTMD_variants/architectures.py:8-14) as one 128 -> 64 MFMA conv over the (features | maps)
    map, weight zero-padded per call; only the feature half of the input gradient is computed."""

    @staticmethod
    def forward(ctx, x, md, w, b):
        _fp32_only("SFT layer")
        B, C, H, W = x.shape
        M = w.shape[1] - 64
        if C != 64 or tuple(w.shape) != (64, 64 + M, 3, 3) or not 0 <= M <= 64:
            raise NotImplementedError("ConcatSft: 64 features, at most 64 metadata maps")
        dev = x.device
        x, md = _cl(x), _cl(md)
        diff = md.data_ptr() - x.data_ptr()
        if diff % 16:
            raise RuntimeError("ConcatSft: feature and metadata maps must be 16-byte aligned relative to each other")
        pf, pd = pack_pair(_pad_oihw(w, 64, 128))
        y = _empty_cl(B, 64, H, W, dev)
        conv_c64(x, hip.view_pair(H, W, diff // 4), pf, b, (1, 64), y, hip.view_plain(H, W, 64), B, H, W, 128, 64)
        ctx.save_for_backward(x, md)
        ctx.pd, ctx.cfg = pd, (B, H, W, tuple(w.shape), b is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        global IN_BACKWARD
        IN_BACKWARD = True
        try:
            x, md = ctx.saved_tensors
            B, H, W, wshape, has_b = ctx.cfg
            dev = dy.device
            dy = _cl(dy)
            v64 = hip.view_plain(H, W, 64)
            dwp = torch.empty((64, 128, 3, 3), device=dev)
            db = torch.empty(64, device=dev) if has_b else None
            wgrad_c64(x, hip.view_pair(H, W, (md.data_ptr() - x.data_ptr()) // 4), dy, v64, dwp, db, B, H, W, 128, 64)
            dx = None
            if ctx.needs_input_grad[0]:
                dx = _empty_cl(B, 64, H, W, dev)
                conv_c64(dy, v64, ctx.pd, None, (1, 64), dx, v64, B, H, W, 64, 64)  # output chunk 0 of the packing = d features
            return dx, None, _crop_oihw(dwp, wshape), db
        finally:
            IN_BACKWARD = False


def sft_apply(x, md, layer, relu):
    """One SFT_Layer (sftmd.SFT_Layer: 'standard' / 'concat' / 'weak' / 'none', ref: SFTMD_variants/architectures.py:59-77)
    followed by the block's F.relu when `relu`."""
    kind = layer.kind
    if kind == "standard":
        return sft_layer(x, md, layer.sft_module, relu)
    if kind == "concat":
        y = _ConcatSft.apply(x, md, layer.sft_module.conv.weight, layer.sft_module.conv.bias)
    elif kind == "weak":
        y = _Map64.apply(x, md, "mul0" if layer.maps == 1 else "mul")
    else:
        y = x
    return _Map64.apply(y, None, "relu") if relu else y


class _SftmdHead(Function):
    """fea_bef = conv3(leaky(conv2(leaky(conv1(x))))) (ref: SFTMD_variants/architectures.py:162): x NCHW RGB."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3):
        _fp32_only("SFTMD")
        B, C, H, W = x.shape
        if (C > 64 or tuple(w1.shape) != (64, C, 3, 3) or tuple(w2.shape) != (64, 64, 3, 3) or
                tuple(w3.shape) != (64, 64, 3, 3)):
            raise NotImplementedError("SFTMD head: (RGB [+ metadata maps, at most 64 channels]) -> 64 -> 64 -> 64")
        dev, npix = x.device, B * H * W
        x = x.contiguous()
        v64 = hip.view_plain(H, W, 64)
        y1 = _empty_cl(B, 64, H, W, dev)
        if C == 3:
            hip.check(hip.lib().sisr_conv3x3_cin3(hip.ptr(x), hip.ptr(w1.contiguous()), 27, 9, 0, hip.ptr(b1), hip.ptr(y1), v64,
                                                  B, H, W, 64, hip.stream()), "sisr_conv3x3_cin3")
            _map64(y1, 64, None, 0, y1, 64, npix, 2)
        else:
            # concat_strategy (ref: SFTMD_variants/handlers.py:12-14, attention_manipulators/__init__.py:97-98): the metadata
            # maps ride in the input, conv1 is (3 + M) -> 64.  The NCHW input is laid out once as a zero-padded channels-last
            # 64-channel map and conv1 runs on the MFMA kernel with zero-padded weights (exact zeros: same sums)
            x = nchw_to_nhwc_pad(x.detach(), 64)
            conv_c64(x, v64, pack_weight(_pad_oihw(w1, 64, 64), "fwd"), b1, (1, 64), y1, v64, B, H, W, 64, 64, relu=LEAKY)
        pf2, pd2 = pack_pair(w2.contiguous())
        pf3, pd3 = pack_pair(w3.contiguous())
        y2 = _empty_cl(B, 64, H, W, dev)
        conv_c64(y1, v64, pf2, b2, (1, 64), y2, v64, B, H, W, 64, 64, relu=LEAKY)
        y3 = _empty_cl(B, 64, H, W, dev)
        conv_c64(y2, v64, pf3, b3, (1, 64), y3, v64, B, H, W, 64, 64)
        ctx.rgb = C == 3
        ctx.save_for_backward(x, y1, y2, w1, w2, w3)
        ctx.packs = (pd2, pd3)
        return y3

    @staticmethod
    def backward(ctx, d3):
        global IN_BACKWARD
        IN_BACKWARD = True
        try:
            x, y1, y2, w1, w2, w3 = ctx.saved_tensors
            pd2, pd3 = ctx.packs
            B, _, H, W = x.shape
            dev = x.device
            L = hip.lib()
            d3 = _cl(d3)
            v64 = hip.view_plain(H, W, 64)
            dw3, db3 = _grad_buf(w3), torch.empty(64, device=dev)
            wgrad_c64(y2, v64, d3, v64, dw3, db3, B, H, W, 64, 64, owner=w3)
            d2 = _empty_cl(B, 64, H, W, dev)
            conv_c64(d3, v64, pd3, None, (1, 64), d2, v64, B, H, W, 64, 64, mask=y2, relu=LEAKY_MASK)
            dw2, db2 = _grad_buf(w2), torch.empty(64, device=dev)
            wgrad_c64(y1, v64, d2, v64, dw2, db2, B, H, W, 64, 64, owner=w2)
            d1 = _empty_cl(B, 64, H, W, dev)
            conv_c64(d2, v64, pd2, None, (1, 64), d1, v64, B, H, W, 64, 64, mask=y1, relu=LEAKY_MASK)
            db1 = torch.empty(64, device=dev)
            if ctx.rgb:
                dw1 = torch.empty_like(w1)
                nbytes = L.sisr_corr3x3_c3_workspace_bytes(B, H, W, 64)
                ws = hip.workspace(dev, nbytes)
                hip.check(L.sisr_corr3x3_c3(hip.ptr(x), hip.ptr(d1), v64, 1.0, hip.ptr(dw1), 27, 9, 0, 0, hip.ptr(db1),
                                            hip.ptr(ws), nbytes, B, H, W, 64, hip.stream()), "sisr_corr3x3_c3(head)")
            else:  # x is the padded 64-channel input map: full 64 x 64 weight gradient, cropped to the (3 + M) real inputs
                dw1p = torch.empty((64, 64, 3, 3), device=dev)
                wgrad_c64(x, v64, d1, v64, dw1p, db1, B, H, W, 64, 64)
                dw1 = _crop_oihw(dw1p, tuple(w1.shape))
            return None, dw1, db1, dw2, db2, dw3, db3
        finally:
            IN_BACKWARD = False


class _SftmdTail(Function):
    """clamp(conv_output(upscale(conv_mid(fea)))) (ref: SFTMD_variants/architectures.py:137-176): conv_mid, then one
    (x2, x3: PixelShuffle(scale)) or two (x4: PixelShuffle(2) twice) conv -> shuffle -> LeakyReLU(0.2) stages with the
    shuffle as the conv's output view and the activation as its epilogue, the 9x9 64 -> 3 conv and the clamp to [0, 1]."""

    @staticmethod
    def forward(ctx, fea, n_up, *params):
        _fp32_only("SFTMD")
        B, C, H, W = fea.shape
        dev = fea.device
        wm, bm = params[0], params[1]
        ups = [(params[2 + 2 * k], params[3 + 2 * k]) for k in range(n_up)]
        wo, bo = params[2 + 2 * n_up], params[3 + 2 * n_up]
        if C != 64 or tuple(wo.shape) != (3, 64, 9, 9):
            raise NotImplementedError("SFTMD tail: 64 features, 9x9 64 -> 3 output conv")
        fea = _cl(fea)
        v64 = hip.view_plain(H, W, 64)
        pfm, pdm = pack_pair(wm.contiguous())
        m = _empty_cl(B, 64, H, W, dev)
        conv_c64(fea, v64, pfm, bm, (1, 64), m, v64, B, H, W, 64, 64)
        maps, geo, packs = [fea, m], [], [pdm]
        cur, h, w = m, H, W
        for wu, bu in ups:
            rr = wu.shape[0] // 64
            r = int(round(rr ** 0.5))
            if r * r != rr or wu.shape[1] != 64 or r < 2:
                raise NotImplementedError("SFTMD upscale conv: 64 -> 64 r^2 feeding PixelShuffle(r)")
            pf, pd = pack_pair(wu.contiguous(), r)
            y = _empty_cl(B, 64, h * r, w * r, dev)
            conv_c64(cur, hip.view_plain(h, w, 64), pf, bu, (rr, 1), y, hip.view_shuffle(h, w, r), B, h, w, 64, 64 * rr,
                     relu=LEAKY)
            geo.append((h, w, r))
            packs.append(pd)
            maps.append(y)
            cur, h, w = y, h * r, w * r
        pre = torch.empty((B, 3, h, w), device=dev)
        hip.check(hip.lib().sisr_conv9_fwd(hip.ptr(cur), hip.ptr(wo.contiguous()), hip.ptr(bo), hip.ptr(pre), B, h, w,
                                           hip.stream()), "sisr_conv9_fwd")
        out = torch.empty_like(pre)
        hip.check(hip.lib().sisr_clamp01(hip.ptr(pre), None, hip.ptr(out), pre.numel(), 0, hip.stream()), "sisr_clamp01")
        ctx.save_for_backward(pre, *maps, wm, *[u[0] for u in ups], wo)
        ctx.cfg = (B, H, W, n_up, geo)
        ctx.packs = packs
        return out

    @staticmethod
    def backward(ctx, dout):
        global IN_BACKWARD
        IN_BACKWARD = True
        try:
            B, H, W, n_up, geo = ctx.cfg
            sv = list(ctx.saved_tensors)
            pre, maps, ws_ = sv[0], sv[1:3 + n_up], sv[3 + n_up:]
            wm, wus, wo = ws_[0], ws_[1:1 + n_up], ws_[1 + n_up]
            dev = pre.device
            L = hip.lib()
            h, w = pre.shape[2], pre.shape[3]
            dpre = torch.empty_like(pre)
            hip.check(L.sisr_clamp01(hip.ptr(pre), hip.ptr(dout.contiguous()), hip.ptr(dpre), pre.numel(), 1, hip.stream()),
                      "sisr_clamp01(backward)")
            top = maps[-1]  # the activated map the 9x9 conv read
            dwo, dbo = torch.empty_like(wo), torch.empty(3, device=dev)
            nbytes = L.sisr_conv9_wgrad_workspace_bytes(B, h, w)
            ws = hip.workspace(dev, nbytes)
            hip.check(L.sisr_conv9_wgrad(hip.ptr(top), hip.ptr(dpre), hip.ptr(dwo), hip.ptr(dbo), hip.ptr(ws), nbytes, B, h, w,
                                         hip.stream()), "sisr_conv9_wgrad")
            g = _empty_cl(B, 64, h, w, dev)
            hip.check(L.sisr_conv9_dgrad(hip.ptr(dpre), hip.ptr(wo.contiguous()), hip.ptr(top) if n_up else None, hip.ptr(g), B,
                                         h, w, hip.stream()), "sisr_conv9_dgrad")
            grads = [None] * (4 + 2 * n_up)
            grads[2 + 2 * n_up], grads[3 + 2 * n_up] = dwo, dbo
            for k in range(n_up - 1, -1, -1):
                hk, wk, r = geo[k]
                rr = r * r
                xin = maps[1 + k]  # m for k == 0, else the previous stage's activated output
                dyv = hip.view_shuffle(hk, wk, r)
                vin = hip.view_plain(hk, wk, 64)
                dwu, dbu = _grad_buf(wus[k]), torch.empty(64 * rr, device=dev)
                wgrad_c64(xin, vin, g, dyv, dwu, dbu, B, hk, wk, 64, 64 * rr, shuffle=r)
                gin = _empty_cl(B, 64, hk, wk, dev)
                if k > 0:  # the map below is a LeakyReLU output
                    conv_c64(g, dyv, ctx.packs[1 + k], None, (1, 64), gin, vin, B, hk, wk, 64 * rr, 64, mask=xin,
                             relu=LEAKY_MASK)
                else:
                    conv_c64(g, dyv, ctx.packs[1 + k], None, (1, 64), gin, vin, B, hk, wk, 64 * rr, 64)
                grads[2 + 2 * k], grads[3 + 2 * k] = dwu, dbu
                g = gin
            v64 = hip.view_plain(H, W, 64)
            dwm, dbm = _grad_buf(wm), torch.empty(64, device=dev)
            wgrad_c64(maps[0], v64, g, v64, dwm, dbm, B, H, W, 64, 64, owner=wm)
            grads[0], grads[1] = dwm, dbm
            dfea = None
            if ctx.needs_input_grad[0]:
                dfea = _empty_cl(B, 64, H, W, dev)
                conv_c64(g, v64, ctx.packs[0], None, (1, 64), dfea, v64, B, H, W, 64, 64)
            return (dfea, None, *grads)
        finally:
            IN_BACKWARD = False


def sftmd_forward(net, x, metadata):
    """The whole SFTMD network (sftmd.SFTMD holds the parameters; ref: SFTMD_variants/architectures.py:161-176).  metadata:
    (B, M, H, W) maps -- or, with q_injection, the (B, M, 1, 1) vectors the reference's handler then supplies (:19-22)."""
    _fp32_only("SFTMD")
    B, _, H, W = x.shape
    if metadata.device != x.device:
        # concat_strategy: the handler concatenates the maps to the input on the host and hands the network the host copy
        # (ref: attention_manipulators/__init__.py:94-98 moves them only when they are NOT concatenated)
        metadata = metadata.to(x.device)
    if net.uses_maps:
        if metadata.dim() != 4 or metadata.shape[0] != B or metadata.shape[1] != net.para or tuple(metadata.shape[2:]) != (H, W):
            raise RuntimeError(f"SFTMD: metadata maps must be (B, {net.para}, H, W); got {tuple(metadata.shape)}")
        maps = metadata.detach().float()
        if net.repeats is not None:
            maps = maps.repeat(1, net.repeats, 1, 1)  # input formatting (ref: StandardSft.forward :46-47)
        md = nchw_to_nhwc_pad(maps, 64)
    else:  # the SFT layers never look at the maps: a zero map stands in for the (all-zero-weight) metadata chunk
        md = _empty_cl(B, 64, H, W, x.device).zero_()
    q = net.q_injection
    if q and (metadata.dim() != 4 or tuple(metadata.shape[1:]) != (net.para, 1, 1)):
        raise RuntimeError(f"SFTMD with q_injection: metadata must be (B, {net.para}, 1, 1) vectors; got {tuple(metadata.shape)}")
    fea_bef = _SftmdHead.apply(x, net.conv1.weight, net.conv1.bias, net.conv2.weight, net.conv2.bias, net.conv3.weight,
                               net.conv3.bias)
    fea = fea_bef
    for blk in net.blocks():
        f1 = sft_apply(fea, md, blk.sft1, True)
        if q:
            f1 = gate_mul(f1, blk.q_1.gate(metadata))
        f2 = sft_apply(conv3x3(f1, blk.conv1.weight, blk.conv1.bias), md, blk.sft2, True)
        if q:
            f2 = gate_mul(f2, blk.q_2.gate(metadata))
        fea = conv3x3(f2, blk.conv2.weight, blk.conv2.bias, residual=fea)
    fea_fin = sft_apply(add_residual(fea, fea_bef), md, net.sft, False)
    if q:
        fea_fin = gate_mul(fea_fin, net.final_injection.gate(metadata))
    convs = [m for m in net.upscale if isinstance(m, torch.nn.Conv2d)]
    params = [net.conv_mid.weight, net.conv_mid.bias]
    for c in convs:
        params += [c.weight, c.bias]
    params += [net.conv_output.weight, net.conv_output.bias]
    return _SftmdTail.apply(fea_fin, len(convs), *params)


# ----------------------------------------------------------------------------- stand-alone gates
def _pixel_sums(t, other, B, H, W, C=64):
    """[B][parts][C] ordered partial sums of t*other (other None: of t); C a multiple of 64."""
    L = hip.lib()
    parts = L.sisr_gate_dg_parts(H * W)
    part = torch.empty((B, parts, C), device=t.device, dtype=torch.float32)
    hip.check(L.sisr_gate_dg_partial(hip.ptr(t), hip.ptr(other), hip.ptr(part), B, H * W, C, hip.stream()),
              "sisr_gate_dg_partial")
    return part, parts


def _affine(t, g, shift, x, B, H, W, C):
    y = _empty_cl(B, C, H, W, t.device)
    hip.check(hip.lib().sisr_gate_residual_fwd(hip.ptr(t), hip.ptr(g), hip.ptr(shift), hip.ptr(x), hip.ptr(y), B,
                                               H * W, C, hip.stream()), "sisr_gate_residual_fwd")
    return y


class _CALayer(Function):
    """y = x * sigmoid(W2 relu(W1 mean_hw(x) + b1) + b2)   (ref: advanced/architectures.py:29-32)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        B, C, H, W = x.shape
        if C != 64:
            raise NotImplementedError("channel attention kernels are specialised for 64 channels")
        x = _cl(x)
        dev = x.device
        R = w1.shape[0]
        w1c, w2c = w1.reshape(R, 64).contiguous(), w2.reshape(64, R).contiguous()
        part, parts = _pixel_sums(x, None, B, H, W)
        s, hid, ca, g = _vec(B, 64, dev), _vec(B, R, dev), _vec(B, 64, dev), _vec(B, 64, dev)
        hip.check(hip.lib().sisr_ca_gate_fwd(hip.ptr(part), parts, B, 1.0 / (H * W), hip.ptr(w1c),
                                             hip.ptr_c(b1), hip.ptr(w2c), hip.ptr_c(b2), 64, R,
                                             None, hip.ptr(s), hip.ptr(hid), hip.ptr(ca), hip.ptr(g), hip.stream()),
                  "sisr_ca_gate_fwd")
        ctx.save_for_backward(x, w1c, w2c, s, hid, ca)
        ctx.shapes = (w1.shape, w2.shape)
        return _affine(x, g, None, None, B, H, W, 64)

    @staticmethod
    def backward(ctx, dy):
        x, w1c, w2c, s, hid, ca = ctx.saved_tensors
        B, C, H, W = x.shape
        dev = x.device
        R = w1c.shape[0]
        dy = _cl(dy)
        dgp, parts = _pixel_sums(dy, x, B, H, W)
        shift = _vec(B, 64, dev)
        dw1, db1 = torch.empty_like(w1c), torch.empty(R, device=dev)
        dw2, db2 = torch.empty_like(w2c), torch.empty(64, device=dev)
        hip.check(hip.lib().sisr_ca_gate_bwd(hip.ptr(dgp), parts, B, 1.0 / (H * W), hip.ptr(w1c), hip.ptr(w2c), 64, R,
                                             hip.ptr(s), hip.ptr(hid), hip.ptr(ca), None, hip.ptr(shift), None,
                                             hip.ptr(dw1), hip.ptr(db1), hip.ptr(dw2), hip.ptr(db2),
                                             hip.ptr(_gate_ws(B, dev)), hip.gate_counter(dev), hip.stream()),
                  "sisr_ca_gate_bwd")
        dx = _affine(dy, ca, shift, None, B, H, W, 64)
        return dx, dw1.reshape(ctx.shapes[0]), db1, dw2.reshape(ctx.shapes[1]), db2


def ca_layer(x, w1, b1, w2, b2):
    if x.shape[1] != 64:  # wider maps: pooled sums + the generic gate MLP + gate multiply (see _wide_gated_block)
        B, C = x.shape[0], x.shape[1]
        g = _GateMlp.apply(global_avg_pool(x), x.new_zeros((B, 1, 1, 1)), None, ([(0, 0, 1), (0, 0, 2)], 0), w1, b1, w2, b2)
        return gate_mul(x, g.reshape(B, C, 1, 1))
    return _CALayer.apply(x, w1, b1, w2, b2)


class _GateMul(Function):
    """y = t * g[b,c] (+ x)."""

    @staticmethod
    def forward(ctx, t, g, x):
        B, C, H, W = t.shape
        t = _cl(t)
        g2 = g.reshape(B, C).contiguous()
        y = _affine(t, g2, None, _cl(x) if x is not None else None, B, H, W, C)
        ctx.save_for_backward(t, g2)
        ctx.gshape, ctx.has_x = g.shape, x is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        t, g2 = ctx.saved_tensors
        B, C, H, W = t.shape
        if C % 64:
            raise NotImplementedError("gate backward needs a multiple of 64 channels")
        dy = _cl(dy)
        dt = _affine(dy, g2, None, None, B, H, W, C)
        dgp, parts = _pixel_sums(dy, t, B, H, W, C)
        dg = _vec(B, C, t.device)
        hip.check(hip.lib().sisr_sum_partials(hip.ptr(dgp), parts, B, C, 1.0, hip.ptr(dg), hip.stream()),
                  "sisr_sum_partials")
        return dt, dg.reshape(ctx.gshape), (dy if ctx.has_x else None)


def gate_mul(t, g, x=None):
    return _GateMul.apply(t, g, x)


class _GlobalAvgPool(Function):
    """(B,64,H,W) -> (B,64,1,1) mean over pixels (ordered two-stage sum)."""

    @staticmethod
    def forward(ctx, t):
        B, C, H, W = t.shape
        if C % 64:
            raise NotImplementedError("GAP kernel needs a multiple of 64 channels")
        t = _cl(t)
        part, parts = _pixel_sums(t, None, B, H, W, C)
        s = _vec(B, C, t.device)
        hip.check(hip.lib().sisr_sum_partials(hip.ptr(part), parts, B, C, 1.0 / (H * W), hip.ptr(s), hip.stream()),
                  "sisr_sum_partials")
        ctx.geom = (B, C, H, W)
        return s.reshape(B, C, 1, 1)

    @staticmethod
    def backward(ctx, ds):
        B, C, H, W = ctx.geom
        dt = (ds.reshape(B, C, 1, 1) / (H * W)).expand(B, C, H, W).contiguous(memory_format=CL)
        return dt


def global_avg_pool(t):
    return _GlobalAvgPool.apply(t)


class _PALayer(Function):
    """Per-pixel attention gate (ref: attention_manipulators/architectures.py:13-26)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        B, C, H, W = x.shape
        if C != 64 or w1.shape[0] != 8:
            raise NotImplementedError("pixel attention kernel is specialised for 64 channels / 8 hidden units")
        x = _cl(x)
        w1c, w2c = w1.reshape(8, 64).contiguous(), w2.reshape(8).contiguous()
        b1c, b2c = b1.contiguous(), b2.reshape(1).contiguous()
        y = _empty_cl(B, C, H, W, x.device)
        hip.check(hip.lib().sisr_pa_fwd(hip.ptr(x), hip.ptr(w1c), hip.ptr(b1c), hip.ptr(w2c), hip.ptr(b2c), hip.ptr(y),
                                        B * H * W, 64, 8, hip.stream()), "sisr_pa_fwd")
        ctx.save_for_backward(x, w1c, b1c, w2c, b2c)
        ctx.shapes = (w1.shape, w2.shape, b2.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1c, b1c, w2c, b2c = ctx.saved_tensors
        B, C, H, W = x.shape
        L = hip.lib()
        dy = _cl(dy)
        npix = B * H * W
        ws = hip.workspace(x.device, L.sisr_pa_bwd_workspace_bytes(npix))
        dx = _empty_cl(B, C, H, W, x.device)
        dw1, db1 = torch.empty_like(w1c), torch.empty_like(b1c)
        dw2, db2 = torch.empty_like(w2c), torch.empty_like(b2c)
        hip.check(L.sisr_pa_bwd(hip.ptr(x), hip.ptr(w1c), hip.ptr(b1c), hip.ptr(w2c), hip.ptr(b2c), hip.ptr(dy),
                                hip.ptr(dx), hip.ptr(dw1), hip.ptr(db1), hip.ptr(dw2), hip.ptr(db2), hip.ptr(ws), npix,
                                64, 8, hip.stream()), "sisr_pa_bwd")
        s1, s2, sb2 = ctx.shapes
        return dx, dw1.reshape(s1), db1, dw2.reshape(s2), db2.reshape(sb2)


def pa_layer(x, w1, b1, w2, b2):
    return _PALayer.apply(x, w1, b1, w2, b2)


class _AddResidual(Function):
    """y = t + x on channels-last maps."""

    @staticmethod
    def forward(ctx, t, x):
        B, C, H, W = t.shape
        return _affine(_cl(t), None, None, _cl(x), B, H, W, C)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add_residual(t, x):
    return _AddResidual.apply(t, x)


# ----------------------------------------------------------------------------- HAN attention modules
class _LAM(Function):
    """Layer attention over a [B][N][H][W][64] stack (ref: advanced/HAN_blocks.py:16-37)."""

    @staticmethod
    def forward(ctx, stack, gamma):
        B, N = stack.shape[0], stack.shape[1]
        chw = stack[0, 0].numel()
        stack = stack.contiguous()
        L = hip.lib()
        nbytes = L.sisr_lam_workspace_bytes(B, N, chw)
        if nbytes == 0:
            raise NotImplementedError(f"LAM kernel: unsupported shape B={B} N={N} chw={chw}")
        ws = hip.workspace(stack.device, nbytes)
        y = torch.empty_like(stack)
        attn = torch.empty((B, N, N), device=stack.device, dtype=torch.float32)
        g = gamma.reshape(1).contiguous()
        hip.check(L.sisr_lam_fwd(hip.ptr(stack), hip.ptr(g), hip.ptr(y), hip.ptr(attn), hip.ptr(ws), B, N, chw,
                                 hip.stream()), "sisr_lam_fwd")
        ctx.save_for_backward(stack, attn, g)
        ctx.gshape = gamma.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        stack, attn, g = ctx.saved_tensors
        B, N = stack.shape[0], stack.shape[1]
        chw = stack[0, 0].numel()
        L = hip.lib()
        nbytes = L.sisr_lam_workspace_bytes(B, N, chw)
        ws = hip.workspace(stack.device, nbytes)
        dy = dy.contiguous()
        dx = torch.empty_like(stack)
        dg = torch.empty(1, device=stack.device, dtype=torch.float32)
        hip.check(L.sisr_lam_bwd(hip.ptr(stack), hip.ptr(attn), hip.ptr(g), hip.ptr(dy), hip.ptr(dx), hip.ptr(dg),
                                 hip.ptr(ws), B, N, chw, hip.stream()), "sisr_lam_bwd")
        return dx, dg.reshape(ctx.gshape)


def lam(stack, gamma):
    return _LAM.apply(stack, gamma)


class _CSAM(Function):
    """x * (1 + gamma * sigmoid(conv3d(x))) on a channels-last 64-channel map (ref: advanced/HAN_blocks.py:58-76)."""

    @staticmethod
    def forward(ctx, x, w, b, gamma):
        B, C, H, W = x.shape
        if C != 64:
            raise NotImplementedError("CSAM kernel is specialised for 64 channels")
        x = _cl(x)
        w27 = w.reshape(27).contiguous()
        bb, g = b.reshape(1).contiguous(), gamma.reshape(1).contiguous()
        y = _empty_cl(B, C, H, W, x.device)
        hip.check(hip.lib().sisr_csam_fwd(hip.ptr(x), hip.ptr(w27), hip.ptr(bb), hip.ptr(g), hip.ptr(y), B, H, W, C,
                                          hip.stream()), "sisr_csam_fwd")
        ctx.save_for_backward(x, w27, bb, g)
        ctx.shapes = (w.shape, b.shape, gamma.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w27, bb, g = ctx.saved_tensors
        B, C, H, W = x.shape
        L = hip.lib()
        dy = _cl(dy)
        nbytes = L.sisr_csam_bwd_workspace_bytes(B, H, W, C)
        ws = hip.workspace(x.device, nbytes)
        dx = _empty_cl(B, C, H, W, x.device)
        dw, db, dg = (torch.empty(n, device=x.device, dtype=torch.float32) for n in (27, 1, 1))
        hip.check(L.sisr_csam_bwd(hip.ptr(x), hip.ptr(w27), hip.ptr(bb), hip.ptr(g), hip.ptr(dy), hip.ptr(dx),
                                  hip.ptr(dw), hip.ptr(db), hip.ptr(dg), hip.ptr(ws), B, H, W, C, hip.stream()),
                  "sisr_csam_bwd")
        sw, sb, sg = ctx.shapes
        return dx, dw.reshape(sw), db.reshape(sb), dg.reshape(sg)


def csam(x, w, b, gamma):
    return _CSAM.apply(x, w, b, gamma)


class _PixelShuffleCL(Function):
    """nn.PixelShuffle(r) of a channels-last map as one gather (upsamplers wider than 64 channels; ref: advanced/common.py:20-45)."""

    @staticmethod
    def forward(ctx, x, r):
        B, Crr, H, W = x.shape
        C = Crr // (r * r)
        if C * r * r != Crr:
            raise RuntimeError(f"PixelShuffle({r}) needs a channel count divisible by {r * r}; got {Crr}")
        y = _empty_cl(B, C, H * r, W * r, x.device)
        hip.check(hip.lib().sisr_pixel_shuffle_cl(hip.ptr(_cl(x)), hip.ptr(y), B, H, W, C, r, 0, hip.stream()),
                  "sisr_pixel_shuffle_cl")
        ctx.geo = (B, C, H, W, r)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W, r = ctx.geo
        dx = _empty_cl(B, C * r * r, H, W, dy.device)
        hip.check(hip.lib().sisr_pixel_shuffle_cl(hip.ptr(_cl(dy)), hip.ptr(dx), B, H, W, C, r, 1, hip.stream()),
                  "sisr_pixel_shuffle_cl(adjoint)")
        return dx, None


def pixel_shuffle(x, r):
    return _PixelShuffleCL.apply(x, int(r))


class _StackMaps(Function):
    """N channels-last (B, 64, H, W) maps -> the [B][N][H][W][64] stack LAM and the stack convs read as chunks (ref:
    advanced/architectures.py:357-362, the torch.cat of HAN's intermediate maps); backward hands every map its slot."""

    @staticmethod
    def forward(ctx, *maps):
        B, C, H, W = maps[0].shape
        if C != 64:
            raise NotImplementedError("map stacks hold 64-channel maps")
        N = len(maps)
        stack = torch.empty((B, N, H, W, 64), device=maps[0].device, dtype=torch.float32)
        for k, m in enumerate(maps):
            hip.check(hip.lib().sisr_stack_maps(hip.ptr(_cl(m)), hip.ptr(stack), B, H * W, N, k, 0, hip.stream()),
                      "sisr_stack_maps")
        return stack

    @staticmethod
    def backward(ctx, dstack):
        B, N, H, W, _ = dstack.shape
        dstack = dstack.contiguous()
        out = []
        for k in range(N):
            if not ctx.needs_input_grad[k]:
                out.append(None)
                continue
            g = _empty_cl(B, 64, H, W, dstack.device)
            hip.check(hip.lib().sisr_stack_maps(hip.ptr(dstack), hip.ptr(g), B, H * W, N, k, 1, hip.stream()),
                      "sisr_stack_maps(unstack)")
            out.append(g)
        return tuple(out)


def stack_maps(maps):
    return _StackMaps.apply(*maps)


class _ConvStack(Function):
    """3x3 conv 64*N -> 64 reading a [B][N][H][W][64] map stack as N channel chunks (HAN last_conv /
    last, ref: advanced/architectures.py:349-350,366-371) -- torch.cat is never materialised."""

    @staticmethod
    def forward(ctx, stack, weight, bias, residual=None):
        B, N, H, W, C = stack.shape
        if C != 64 or weight.shape[0] != 64 or weight.shape[1] != 64 * N:
            raise NotImplementedError("stack conv expects N maps of 64 channels -> 64 channels")
        stack = stack.contiguous()
        w = weight.contiguous()
        if ctx.needs_input_grad[0]:
            packed, ctx.pd = pack_pair(w)
        else:
            packed, ctx.pd = pack_weight(w, "fwd"), None
        y = _empty_cl(B, 64, H, W, stack.device)
        conv_c64(stack, hip.view_maps(H, W, N), packed, bias, (1, 64), y, hip.view_plain(H, W, 64), B, H, W, 64 * N, 64,
                 res=_cl(residual) if residual is not None else None)
        ctx.save_for_backward(stack, w)
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        global IN_BACKWARD
        IN_BACKWARD = True
        try:
            stack, w = ctx.saved_tensors
            B, N, H, W, C = stack.shape
            dev = stack.device
            dy = _cl(dy)
            dstack = None
            if ctx.needs_input_grad[0]:
                dstack = torch.empty_like(stack)
                conv_c64(dy, hip.view_plain(H, W, 64), ctx.pd, None, (1, 64), dstack, hip.view_maps(H, W, N), B, H, W, 64,
                         64 * N)
            dw = torch.empty_like(w)
            db = torch.empty(64, device=dev, dtype=torch.float32) if ctx.has_bias else None
            wgrad_c64(stack, hip.view_maps(H, W, N), dy, hip.view_plain(H, W, 64), dw, db, B, H, W, 64 * N, 64)
            return dstack, dw, db, (dy if ctx.has_res else None)
        finally:
            IN_BACKWARD = False


def conv3x3_stack(stack, weight, bias, residual=None):
    """conv over the stack's N maps as channel chunks (+ residual, added by the conv's epilogue)."""
    return _ConvStack.apply(stack, weight, bias, residual)


# ----------------------------------------------------------------------------- SAN attention modules
SOCA_LIMIT = 1000  # ref: advanced/SAN_blocks.py:265-266 (h1 = w1 = 1000)
SOCA_ITERS = 5     # ref: advanced/SAN_blocks.py:291 SqrtmLayer(cov_mat, 5)


def soca_window(H, W):
    """Slices of the centre crop SOCA pools over when a side exceeds 1000 (ref: advanced/SAN_blocks.py:267-280,
    strict comparisons and Python slice semantics kept); None when the whole map is used."""
    lim = SOCA_LIMIT
    if H < lim and W < lim:
        return None
    if H < lim and W > lim:
        w0 = (W - lim) // 2
        return slice(None), slice(w0, w0 + lim)
    if W < lim and H > lim:
        h0 = (H - lim) // 2
        return slice(h0, h0 + lim), slice(None)
    h0, w0 = (H - lim) // 2, (W - lim) // 2
    return slice(h0, h0 + lim), slice(w0, w0 + lim)


class _SOCA(Function):
    """Second-order channel attention  y = x * sigmoid(W2 relu(W1 v + b1) + b2),  v = column means of
    sqrtm(cov(x))  (ref: advanced/SAN_blocks.py:261-302; advanced/mpncov.py for cov / sqrtm and their
    hand-written backward, which is what the kernels implement)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        B, C, H, W = x.shape
        if C != 64:
            raise NotImplementedError("SOCA kernels are specialised for 64 channels")
        x = _cl(x)
        dev = x.device
        L = hip.lib()
        win = soca_window(H, W)
        xs = x if win is None else x[:, :, win[0], win[1]].contiguous(memory_format=CL)
        hs, ws = xs.shape[2:]
        M = hs * ws
        part, parts = _pixel_sums(xs, None, B, hs, ws)
        mean = _vec(B, 64, dev)
        hip.check(L.sisr_sum_partials(hip.ptr(part), parts, B, 64, 1.0 / M, hip.ptr(mean), hip.stream()),
                  "sisr_sum_partials")
        cov = torch.empty((B, 64, 64), device=dev, dtype=torch.float32)
        wsp = hip.workspace(dev, L.sisr_covpool_workspace_bytes(B, M))
        hip.check(L.sisr_covpool_fwd(hip.ptr(xs), hip.ptr(mean), hip.ptr(cov), hip.ptr(wsp), B, M, 64, hip.stream()),
                  "sisr_covpool_fwd")
        saved = torch.empty(L.sisr_sqrtm_saved_bytes(B, 64, SOCA_ITERS) // 4, device=dev, dtype=torch.float32)
        pooled = _vec(B, 64, dev)
        hip.check(L.sisr_sqrtm_fwd(hip.ptr(cov), hip.ptr(saved), hip.ptr(pooled), B, 64, SOCA_ITERS, hip.stream()),
                  "sisr_sqrtm_fwd")
        R = w1.shape[0]
        w1c, w2c = w1.reshape(R, 64).contiguous(), w2.reshape(64, R).contiguous()
        s, hid, ca, g = _vec(B, 64, dev), _vec(B, R, dev), _vec(B, 64, dev), _vec(B, 64, dev)
        hip.check(L.sisr_ca_gate_fwd(hip.ptr(pooled), 1, B, 1.0, hip.ptr(w1c), hip.ptr_c(b1), hip.ptr(w2c),
                                     hip.ptr_c(b2), 64, R, None, hip.ptr(s), hip.ptr(hid), hip.ptr(ca),
                                     hip.ptr(g), hip.stream()), "sisr_ca_gate_fwd")
        ctx.save_for_backward(x, w1c, w2c, s, hid, ca, mean, cov, saved)
        ctx.shapes, ctx.win = (w1.shape, w2.shape), win
        return _affine(x, g, None, None, B, H, W, 64)

    @staticmethod
    def backward(ctx, dy):
        x, w1c, w2c, s, hid, ca, mean, cov, saved = ctx.saved_tensors
        B, C, H, W = x.shape
        dev = x.device
        L = hip.lib()
        R = w1c.shape[0]
        dy = _cl(dy)
        dgp, parts = _pixel_sums(dy, x, B, H, W)
        dpooled = _vec(B, 64, dev)
        dw1, db1 = torch.empty_like(w1c), torch.empty(R, device=dev)
        dw2, db2 = torch.empty_like(w2c), torch.empty(64, device=dev)
        hip.check(L.sisr_ca_gate_bwd(hip.ptr(dgp), parts, B, 1.0, hip.ptr(w1c), hip.ptr(w2c), 64, R, hip.ptr(s),
                                     hip.ptr(hid), hip.ptr(ca), None, hip.ptr(dpooled), None, hip.ptr(dw1),
                                     hip.ptr(db1), hip.ptr(dw2), hip.ptr(db2), hip.ptr(_gate_ws(B, dev)),
                                     hip.gate_counter(dev), hip.stream()),
                  "sisr_ca_gate_bwd")
        dsym = torch.empty_like(cov)
        hip.check(L.sisr_sqrtm_bwd(hip.ptr(cov), hip.ptr(saved), hip.ptr(dpooled), hip.ptr(dsym), B, 64, SOCA_ITERS,
                                   hip.stream()), "sisr_sqrtm_bwd")
        win = ctx.win
        if win is None:
            dx = _empty_cl(B, 64, H, W, dev)
            hip.check(L.sisr_soca_bwd_apply(hip.ptr(dy), hip.ptr(ca), hip.ptr(x), hip.ptr(mean), hip.ptr(dsym),
                                            hip.ptr(dx), B, H * W, 64, hip.stream()), "sisr_soca_bwd_apply")
        else:  # covariance term only inside the pooled window (maps with a side > 1000: evaluation-size inputs)
            dx = _affine(dy, ca, None, None, B, H, W, 64)
            xs = x[:, :, win[0], win[1]].contiguous(memory_format=CL)
            hs, ws = xs.shape[2:]
            zero = torch.zeros_like(xs)
            dwin = torch.empty_like(xs)
            hip.check(L.sisr_soca_bwd_apply(hip.ptr(zero), hip.ptr(ca), hip.ptr(xs), hip.ptr(mean), hip.ptr(dsym),
                                            hip.ptr(dwin), B, hs * ws, 64, hip.stream()), "sisr_soca_bwd_apply")
            dx[:, :, win[0], win[1]] += dwin
        return dx, dw1.reshape(ctx.shapes[0]), db1, dw2.reshape(ctx.shapes[1]), db2


def soca(x, w1, b1, w2, b2):
    return _SOCA.apply(x, w1, b1, w2, b2)


class _NonLocalAttention(Function):
    """y_i = sum_j softmax_j(theta_i . phi_j) g_j on [nb][n][8] rows (ref: advanced/SAN_blocks.py:126-141)."""

    @staticmethod
    def forward(ctx, theta, phi, g):
        nb, nq, d = theta.shape
        nk = phi.shape[1]
        theta, phi, g = theta.contiguous(), phi.contiguous(), g.contiguous()
        y = torch.empty_like(theta)
        lse = torch.empty((nb, nq), device=theta.device, dtype=torch.float32)
        hip.check(hip.lib().sisr_nl_attn_fwd(hip.ptr(theta), hip.ptr(phi), hip.ptr(g), hip.ptr(y), hip.ptr(lse), nb, nq,
                                             nk, d, hip.stream()), "sisr_nl_attn_fwd")
        ctx.save_for_backward(theta, phi, g, y, lse)
        return y

    @staticmethod
    def backward(ctx, dy):
        theta, phi, g, y, lse = ctx.saved_tensors
        nb, nq, d = theta.shape
        nk = phi.shape[1]
        dy = dy.contiguous()
        dtheta, dphi, dg = torch.empty_like(theta), torch.empty_like(phi), torch.empty_like(g)
        dsum = torch.empty_like(lse)
        hip.check(hip.lib().sisr_nl_attn_bwd(hip.ptr(theta), hip.ptr(phi), hip.ptr(g), hip.ptr(y), hip.ptr(lse),
                                             hip.ptr(dy), hip.ptr(dtheta), hip.ptr(dphi), hip.ptr(dg), hip.ptr(dsum), nb,
                                             nq, nk, d, hip.stream()), "sisr_nl_attn_bwd")
        return dtheta, dphi, dg


def nonlocal_attention(theta, phi, g):
    return _NonLocalAttention.apply(theta, phi, g)


def _nl_domain_groups(B, H, W, quadrants):
    """Attention domains of a map as groups of equal rectangles (sisr_nl_* `domains` arrays): the whole map, or the four
    quadrants (ref: advanced/SAN_blocks.py:316-334: H // 2, W // 2 split) -- one group when they are equal."""
    import ctypes
    mk = lambda *v: (ctypes.c_int * 9)(*v)  # noqa: E731
    if not quadrants:
        return [(mk(B, H, W, 0, 0, H, W, 1, 1), B, H, W)]
    h1, w1 = H // 2, W // 2
    if H % 2 == 0 and W % 2 == 0:
        return [(mk(B, H, W, 0, 0, h1, w1, 2, 2), 4 * B, h1, w1)]
    return [(mk(B, H, W, y0, x0, hq, wq, 1, 1), B, hq, wq)
            for y0, hq in ((0, h1), (h1, H - h1)) for x0, wq in ((0, w1), (w1, W - w1))]


class _NonLocalBlock(Function):
    """z = W(softmax(theta(x)^T phi(x)) g(x)) + x with phi / g max-pooled 2x2, per attention domain (ref:
    advanced/SAN_blocks.py:104-148, :305-336).  Projections 64 -> 24 and their two gradients on the fp32 matrix cores,
    split / pool / scatter and the 8 -> 64 output projection as streaming kernels (csrc/nonlocal.hip), the attention itself
    csrc/san.hip; nothing here is a library call."""

    @staticmethod
    def forward(ctx, x, quadrants, wt, bt, wp, bp, wg, bg, ww, bw):
        B, C, H, W = x.shape
        if C != 64 or wt.shape[0] != 8 or ww.shape[0] != 64:
            raise NotImplementedError("non-local block kernels: 64 channels, 8 embedding channels (SAN's configuration)")
        L, dev, npix = hip.lib(), x.device, B * H * W
        x = _cl(x)
        wt, wp, wg, ww = (t.reshape(t.shape[0], -1).contiguous() for t in (wt, wp, wg, ww))
        proj = torch.empty((npix, 24), device=dev)
        hip.check(L.sisr_nl_project_fwd(hip.ptr(x), hip.ptr(wt), hip.ptr_c(bt), hip.ptr(wp), hip.ptr_c(bp), hip.ptr(wg),
                                        hip.ptr_c(bg), hip.ptr(proj), npix, hip.stream()), "sisr_nl_project_fwd")
        z = _empty_cl(B, 64, H, W, dev)
        saved = []
        for dom, nd, hq, wq in _nl_domain_groups(B, H, W, quadrants):
            if hq < 2 or wq < 2:
                raise RuntimeError("non-local block needs at least 2x2 positions per attention domain (MaxPool2d(2))")
            nq, nk = hq * wq, (hq // 2) * (wq // 2)
            theta = torch.empty((nd, nq, 8), device=dev)
            phi, g = torch.empty((nd, nk, 8), device=dev), torch.empty((nd, nk, 8), device=dev)
            hip.check(L.sisr_nl_split_pool_fwd(hip.ptr(proj), hip.ptr(theta), hip.ptr(phi), hip.ptr(g), dom, hip.stream()),
                      "sisr_nl_split_pool_fwd")
            y = torch.empty_like(theta)
            lse = torch.empty((nd, nq), device=dev)
            hip.check(L.sisr_nl_attn_fwd(hip.ptr(theta), hip.ptr(phi), hip.ptr(g), hip.ptr(y), hip.ptr(lse), nd, nq, nk, 8,
                                         hip.stream()), "sisr_nl_attn_fwd")
            hip.check(L.sisr_nl_output_fwd(hip.ptr(y), hip.ptr(x), hip.ptr(ww), hip.ptr_c(bw), hip.ptr(z), dom, hip.stream()),
                      "sisr_nl_output_fwd")
            saved += [theta, phi, g, y, lse]
        ctx.save_for_backward(x, proj, wt, wp, wg, ww, *saved)
        ctx.cfg = (B, H, W, bool(quadrants))
        return z

    @staticmethod
    def backward(ctx, dz):
        B, H, W, quadrants = ctx.cfg
        x, proj, wt, wp, wg, ww, *saved = ctx.saved_tensors
        L, dev, npix = hip.lib(), x.device, B * H * W
        dz = _cl(dz)
        groups = _nl_domain_groups(B, H, W, quadrants)
        oparts = [L.sisr_nl_output_bwd_parts(dom) for dom, _, _, _ in groups]
        opart = torch.empty((sum(oparts), 64 * 8 + 64), device=dev)
        dproj = torch.empty((npix, 24), device=dev)
        row = 0
        for k, (dom, nd, hq, wq) in enumerate(groups):
            theta, phi, g, y, lse = saved[5 * k:5 * k + 5]
            nq, nk = hq * wq, (hq // 2) * (wq // 2)
            dy = torch.empty_like(y)
            hip.check(L.sisr_nl_output_bwd(hip.ptr(dz), hip.ptr(y), hip.ptr(ww), hip.ptr(dy), hip.ptr(opart[row:]), dom,
                                           hip.stream()), "sisr_nl_output_bwd")
            row += oparts[k]
            dtheta, dphi, dg = torch.empty_like(theta), torch.empty_like(phi), torch.empty_like(g)
            dsum = torch.empty_like(lse)
            hip.check(L.sisr_nl_attn_bwd(hip.ptr(theta), hip.ptr(phi), hip.ptr(g), hip.ptr(y), hip.ptr(lse), hip.ptr(dy),
                                         hip.ptr(dtheta), hip.ptr(dphi), hip.ptr(dg), hip.ptr(dsum), nd, nq, nk, 8,
                                         hip.stream()), "sisr_nl_attn_bwd")
            hip.check(L.sisr_nl_split_pool_bwd(hip.ptr(proj), hip.ptr(dtheta), hip.ptr(dphi), hip.ptr(dg), hip.ptr(dproj), dom,
                                               hip.stream()), "sisr_nl_split_pool_bwd")
        out_g = torch.empty(64 * 8 + 64, device=dev)
        hip.check(L.sisr_sum_partials(hip.ptr(opart), opart.shape[0], 1, 64 * 8 + 64, 1.0, hip.ptr(out_g), hip.stream()),
                  "sisr_sum_partials")
        pparts = L.sisr_nl_project_bwd_parts(npix)
        ppart = torch.empty((pparts, 33 * 64), device=dev)
        dx = _empty_cl(B, 64, H, W, dev)
        hip.check(L.sisr_nl_project_bwd(hip.ptr(x), hip.ptr(dproj), hip.ptr(dz), hip.ptr(wt), hip.ptr(wp), hip.ptr(wg),
                                        hip.ptr(dx), hip.ptr(ppart), npix, hip.stream()), "sisr_nl_project_bwd")
        pg = torch.empty(33 * 64, device=dev)
        hip.check(L.sisr_sum_partials(hip.ptr(ppart), pparts, 1, 33 * 64, 1.0, hip.ptr(pg), hip.stream()), "sisr_sum_partials")
        pg = pg.view(33, 64)
        dw = [pg[8 * i:8 * i + 8].reshape(8, 64, 1, 1) for i in range(3)]
        db = [pg[32, 8 * i:8 * i + 8] for i in range(3)]
        return (dx, None, dw[0], db[0], dw[1], db[1], dw[2], db[2], out_g[:512].view(64, 8, 1, 1), out_g[512:])


def nonlocal_block(x, block, quadrants):
    """block: san.NONLocalBlock2D (parameter holder)."""
    return _NonLocalBlock.apply(x, bool(quadrants), block.theta.weight, block.theta.bias, block.phi[0].weight,
                                block.phi[0].bias, block.g[0].weight, block.g[0].bias, block.W.weight, block.W.bias)


class _ScaleAdd(Function):
    """y = a + gamma * r with a learnable scalar gamma (SAN's share-source skip, ref: advanced/architectures.py:301-302)."""

    @staticmethod
    def forward(ctx, a, r, gamma):
        B, C, H, W = a.shape
        a, r = _cl(a), _cl(r)
        gv = gamma.reshape(1, 1).expand(B, C).contiguous()
        ctx.save_for_backward(r, gv)
        ctx.gshape = gamma.shape
        return _affine(r, gv, None, a, B, H, W, C)

    @staticmethod
    def backward(ctx, dy):
        r, gv = ctx.saved_tensors
        B, C, H, W = r.shape
        dy = _cl(dy)
        dr = _affine(dy, gv, None, None, B, H, W, C)
        if C != 64:
            raise NotImplementedError("scale_add backward is specialised for 64 channels")
        dgp, parts = _pixel_sums(dy, r, B, H, W)  # [B][parts][64] ordered partial sums of dy * r
        L = hip.lib()
        per = _vec(B, C, dy.device)
        hip.check(L.sisr_sum_partials(hip.ptr(dgp), parts, B, C, 1.0, hip.ptr(per), hip.stream()), "sisr_sum_partials")
        dgam = torch.empty(1, device=dy.device, dtype=torch.float32)
        hip.check(L.sisr_sum_partials(hip.ptr(per), B * C, 1, 1, 1.0, hip.ptr(dgam), hip.stream()), "sisr_sum_partials")
        return dy, dr, dgam.reshape(ctx.gshape)


def scale_add(a, r, gamma):
    return _ScaleAdd.apply(a, r, gamma)


# ----------------------------------------------------------------------------- L1 loss
class _L1Loss(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        if a.shape != b.shape:
            raise RuntimeError(f"l1_loss shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
        L = hip.lib()
        loss = torch.empty((), device=a.device, dtype=torch.float32)
        grad = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        ws = hip.workspace(a.device, L.sisr_l1_loss_workspace_bytes())
        hip.check(L.sisr_l1_loss(hip.ptr(a), hip.ptr(b), a.numel(), hip.ptr(loss), hip.ptr(grad), hip.ptr(ws),
                                 hip.stream()), "sisr_l1_loss")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, go):
        (grad,) = ctx.saved_tensors
        return grad * go, None


def l1_loss(a, b):
    return _L1Loss.apply(a, b)
