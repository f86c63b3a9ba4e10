#!/usr/bin/env python3
"""Headline benchmark: LR-patches/s (128x128x3 -> x4) for one training step on N MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py starts its N ranks itself,
                                                            as a child `python -m torch.distributed.run ...`, and relays the line)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step is BaseModel.train_step on the full-depth network: forward + L1 + backward + Adam + scheduler step
(the reference's run_train, Code/SISR/models/__init__.py:466-489, minus its per-step D2H copies), on
synthetic inputs already resident in HBM.  value = all patches of all ranks / max-over-ranks wall
time of exactly K steps.

Default workloads (BASELINE.json configs):
  N = 1   configs[1]: RCAN x4, 32 tiles on the GPU.  The line also carries "meta_rcan": the same measurement for
          RCAN + meta-attention (QRCAN, north_star's target family), taken right after the timed region.
  N > 1   configs[3]: Meta-RCAN (QRCAN) data-parallel, GLOBAL batch 32 -> 32/N tiles per GPU, gradients all-reduced
          over RCCL ("scaling": "strong"); per-GPU batches <= 8 replay forward+backward from a hipGraph.  The line also
          carries "weak_scaling": QRCAN at 32 tiles PER GPU (global batch 32 N), measured after the timed region.
  --batch B (tiles per GPU, weak scaling) or --global-batch G (strong scaling) override either default.

Extra objects on the JSON line:
  roofline      bound "mfma" (fp32 path; SURVEY.md 8d: the step is bound by the fp32 matrix rate, not HBM).  Primary
                achieved / frac = the WHOLE STEP's algorithmic TFLOP/s per GPU against the 157.3 TFLOP/s fp32 MFMA peak.
                "families": each 64->64 conv kernel family of the step (forward plain, forward GATE, dgrad with ReLU mask,
                dgrad with DOT epilogue, weight gradient + slab reduce) timed ALONE with HIP events on its launch stream
                in one extra, untimed step with the weight-gradient side stream off: algorithmic FLOPs per launch /
                mean launch duration.  "kernel" names the family with the largest share of the step.
  cpu_baseline  rank 0, N=1 only: the CPU oracle (a restatement of the reference proven equal to it by the
                golden vectors) doing the same step at batch 1 on the host cores -- a bounded sample.
  config4_point N=1 default run only: BASELINE config 4's per-GPU operating point measured on this one GPU -- QRCAN, 4 tiles,
                through a one-rank RCCL world (GradReducer buckets, all-reduce, join) with forward+backward replayed from a
                hipGraph -- with its own whole-step fraction and per-family launch times.
  han_bf16      N=1 default run only: BASELINE config 5 (HAN x4, bf16 matrix-core operands, 16 tiles), roofline bound "hbm"
                against the bytes of the storage format in use (fp32 maps).
  inference     N=1 default run only: RCAN x4 forward only through the handler's run_eval (32 tiles, output kept on the device).
  meta_edsr     N=1 default run only: BASELINE config 3 (QEDSR = EDSR-baseline + a meta-attention layer per block), 32 tiles.
  b1_point      N=1 default run only: ONE tile per step (the per-GPU batch SURVEY 8 puts configs 2 and 3 at, and the batch
                cpu_baseline is timed at): RCAN and QEDSR, forward+backward replayed from a hipGraph.
  sparnet       N=1 default run only: SURVEY 8 row f4, the default SPARNet plan at 16 images of 128 x 128 per step (hipGraph
                replay), images/s.
  overlapped_grad_exchange   N > 1 default run only: the same workload with the replay's gradient exchange overlapped with its
                backward (opt-in SISR_GRAPH_OVERLAP=1), measured last under a watchdog; config.grad_exchange names the mode
                the headline ran.
Every auxiliary object is measured inside its own try block: an error lands under that object's key as {"error": ...}.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CONV_BODY_FLOP_PER_PIXEL = 2 * 64 * 64 * 9
FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: fp32 matrix == fp32 vector peak
HBM_PEAK_GBS = 8000.0          # same guide: HBM3E ~8 TB/s
# algorithmic HBM bytes per LR patch, fwd + bwd, fp32 maps, layer at a time with the legal fusions
HBM_GB_PER_PATCH = {"rcan": 16.5, "qrcan": 16.5, "edsr": 1.4, "qedsr": 1.4, "han": 17.0, "qhan": 17.0}
# ... with the residual groups' kept activations stored as bf16 (ops.set_storage("act")): per RCAB forward 3 fp32-map units
# instead of 6 (conv1 with the GATE prologue reads t2 and the skip and writes the new skip and t1, conv2 reads t1 and writes
# t2: six half-size maps), backward 9 instead of 11 (the ReLU mask, the DOT operand and both weight gradients' x operand are
# half size; gradient maps stay fp32): 200 blocks x 5 x 4 MiB = 4.2 GB less per patch
HBM_GB_PER_PATCH_ACT16 = {k: v - 4.2 for k, v in HBM_GB_PER_PATCH.items() if k in ("rcan", "qrcan", "han", "qhan")}
# ... and the gradient maps the groups' backward passes hand from launch to launch as well (ops.set_storage("all")): backward 5.5
# units per RCAB (dgrad with mask 1.5, dgrad + residual + DOT 2, the two weight gradients 1 each): 200 x 8.5 x 4 MiB less
HBM_GB_PER_PATCH_ALL16 = {k: v - 7.1 for k, v in HBM_GB_PER_PATCH.items() if k in ("rcan", "qrcan", "han", "qhan")}
STORAGE_TABLE = {"0": HBM_GB_PER_PATCH, "act": HBM_GB_PER_PATCH_ACT16, "all": HBM_GB_PER_PATCH_ALL16}
# algorithmic HBM bytes of ONE 64 -> 64 launch per 128 x 128 sample (fp32 maps of 4 MiB): what each kernel family must move
FAMILY_MAPS = {"conv fwd, plain": 2, "conv fwd, GATE": 4, "dgrad, ReLU mask": 3, "dgrad + residual, DOT": 4,
               "dgrad, plain / residual": 3, "wgrad": 2}
WORKLOADS = {
    # name: (registry name, handler kwargs, fwd+bwd algorithmic TFLOP per LR patch)
    "rcan": ("rcan", {}, 1.565),
    "qrcan": ("qrcan", {"metadata": ["blur_kernel"], "style": "standard", "include_q_layer": True}, 1.565),
    "edsr": ("edsr", {}, 0.195),
    "edsr256": ("edsr", {"num_features": 256, "num_blocks": 32, "res_scale": 0.1}, 4.940),  # the paper's EDSR
    "qedsr": ("qedsr", {"metadata": ["blur_kernel"]}, 0.195),
    "han": ("han", {}, 1.614),
    "qhan": ("qhan", {"metadata": ["blur_kernel"]}, 1.614),
    # SAN: 20 groups x (10 RB x 2 convs + 1) = 420 body convs + head + upsampler/tail; attention FLOPs not counted
    "san": ("san", {}, 1.598),
    "qsan": ("qsan", {"metadata": ["blur_kernel"]}, 1.598),
    # metadata-MAP models: 10 blur-kernel PCA maps at LR size beside the RGB input.
    # SRMD: 2*9*(13*128 + 10*128*128 + 128*48) flop per LR pixel forward; SFTMD: head 151k + 16 blocks * (2 SFT layers
    # [2*(74->32) + 2*(32->64) 3x3 convs = 159k] + 2 convs 73.7k) + SFT + conv_mid + upsampler (295k + 4*295k) + 9x9 output
    # conv at 16 HR pixels (498k) = 9.80 Mflop per LR pixel forward; x3 for forward + both gradients, x 16384 pixels.
    "srmd": ("srmd", {"metadata": ["blur_kernel"], "nc": 128, "nb": 12, "maps": True}, 0.152),
    "sftmd": ("sftmd", {"metadata": ["blur_kernel"], "num_blocks": 16, "num_features": 64, "in_nc": 3, "maps": True}, 0.482),
    # SPARNet (SURVEY.md 8f-4 "then SPARNet"): interp-input model, 128 x 128 in AND out; 14.26 GFLOP forward per image in
    # its 3x3 convs (counted conv by conv over the default plan: 32 / 64 / 128 channels at 128^2 .. 4^2 pixels), x3
    "sparnet": ("sparnet", {"hr_same": True}, 0.0428),
    "qsparnet": ("qsparnet", {"metadata": ["blur_kernel"], "hr_same": True}, 0.0428),
}
GRAPH_MAX_BATCH = 8  # --graph auto: per-GPU batches up to this replay forward+backward from a hipGraph


class KernelTimer:
    """HIP-event timing of every body-shape (width -> width, 3x3) launch, per kernel family.  Events are recorded on
    the stream the kernel is launched on; used for ONE extra step with the weight-gradient side stream off, so every
    kernel runs alone."""

    def __init__(self, ops, width=64):
        self.ops, self.width, self.on = ops, width, False
        self.orig_conv, self.orig_wgrad = ops.conv_c64, ops.wgrad_c64
        self.events = {}

    def _timed(self, family, pixels, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self.events.setdefault(family, []).append((e0, e1, pixels))

    def install(self):
        def conv(x, xview, packed, bias, bnq, y, yview, B, H, W, cin, cout, **kw):
            call = lambda: self.orig_conv(x, xview, packed, bias, bnq, y, yview, B, H, W, cin, cout, **kw)  # noqa: E731
            if not (self.on and cin == self.width and cout == self.width):
                return call()
            if kw.get("gate_add") is not None:
                fam = "conv fwd, GATE prologue (builds and stores the gated skip while staging)"
            elif kw.get("dot") is not None:
                fam = "dgrad + residual, DOT epilogue (gate-gradient partial sums)"
            elif kw.get("mask") is not None:
                fam = "dgrad, ReLU mask (+ per-(b,c) affine prologue)"
            elif self.ops.IN_BACKWARD:
                fam = "dgrad, plain / residual"
            else:
                fam = "conv fwd, plain (bias / ReLU / GAP partials / residual epilogue)"
            self._timed(fam, B * H * W, call)

        def wgrad(x, xview, dy, dyview, dw, db, B, H, W, cin, cout, **kw):
            call = lambda: self.orig_wgrad(x, xview, dy, dyview, dw, db, B, H, W, cin, cout, **kw)  # noqa: E731
            if not (self.on and cin == self.width and cout == self.width):
                return call()
            self._timed("wgrad + slab reduce", B * H * W, call)

        self.ops.conv_c64, self.ops.wgrad_c64 = conv, wgrad

    def remove(self):
        self.ops.conv_c64, self.ops.wgrad_c64 = self.orig_conv, self.orig_wgrad

    def summary(self, peak_tflops):
        out = []
        for fam, evs in self.events.items():
            ms = [a.elapsed_time(b) for a, b, _ in evs]
            flop = [p * CONV_BODY_FLOP_PER_PIXEL * (self.width // 64) ** 2 for _, _, p in evs]
            tf = sum(flop) / (sum(ms) * 1e-3) / 1e12
            out.append({"family": fam, "launches_per_step": len(ms), "avg_launch_us": 1e3 * sum(ms) / len(ms),
                        "flop_per_launch": sum(flop) / len(flop), "achieved": tf, "frac": tf / peak_tflops,
                        "ms_per_step": sum(ms)})
        out.sort(key=lambda d: -d["ms_per_step"])
        return out


def cpu_baseline(workload, seconds_budget=30.0):
    """Oracle (CPU restatement of the reference) timed on the host cores: same step, batch 1."""
    import importlib
    from oracle import sisr_oracle as O
    sisr = importlib.import_module("sisr_amd")
    name, params, _ = WORKLOADS[workload]
    params = dict(params)
    maps = params.pop("maps", False)
    hr = 128 if params.pop("hr_same", False) else 512
    torch.set_num_threads(min(16, os.cpu_count() or 1))  # the GPU box grants 16 host cores per GPU
    torch.manual_seed(8)
    h = sisr.available_models[name](device=torch.device("cpu"), model_save_dir="/tmp", eval_mode=False, scale=4, **params)
    cfg = {"rcan": dict(n_resgroups=10, n_resblocks=20, scale=4),
           "edsr": dict(num_blocks=params.get("num_blocks", 16), scale=4, res_scale=0.1),
           "qrcan": dict(n_resgroups=10, n_resblocks=20, scale=4, style="standard", include_q_layer=True),
           "qedsr": dict(num_blocks=16, scale=4, res_scale=0.1, q_layer_nonlinearity=False),
           "han": dict(n_resgroups=10, n_resblocks=20, scale=4), "qhan": dict(n_resgroups=10, n_resblocks=20, scale=4),
           "san": dict(n_resgroups=20, n_resblocks=10, scale=4), "qsan": dict(n_resgroups=20, n_resblocks=10, scale=4),
           "srmd": {}, "sftmd": {}, "sparnet": dict(training=True), "qsparnet": dict(training=True)}[name]
    tr = O.Trainer(name, h.net.state_dict(), lr=1e-4, **cfg)
    g = torch.Generator().manual_seed(8)
    x, y = torch.rand(1, 3, 128, 128, generator=g), torch.rand(1, 3, hr, hr, generator=g)
    md = (torch.rand(1, 10, 1, 1, generator=g) * 0.4) if name in O.META_NETS else None
    if maps:
        md = O.sft_channels(x, (torch.rand(1, 10, generator=g, dtype=torch.float64) * 0.4))
        if name not in O.META_NETS:  # SRMD: the maps are concatenated to the input (ref: advanced/handlers.py:132-158)
            x, md = torch.cat((x, md), 1), None
    tr.step(x, y, md)  # warm-up
    times = []
    t_all = time.perf_counter()
    while len(times) < 3 and (time.perf_counter() - t_all) < seconds_budget:
        t0 = time.perf_counter()
        tr.step(x, y, md)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    ref = {"rcan": 0.265, "qrcan": 0.266, "edsr": 1.75, "qedsr": 2.44, "han": 0.296, "qhan": 0.246}.get(workload)
    out = {"value": 1.0 / med, "unit": "patches/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{name} x4 full depth, batch 1, 128x128 LR tile, fwd+L1+bwd+Adam, median of {len(times)} "
                     f"timed steps after 1 warm-up (oracle/sisr_oracle.py on torch CPU kernels)"}
    if ref is not None:  # the reference itself, same step, in the build container
        out["reference_anchor"] = {"value": ref, "unit": "patches/s", "cores": 8, "kind": "reference",
                                   "where": "build container (8 vCPU Xeon 2.1 GHz), the reference's own run_train; "
                                            "it cannot travel to the GPU box"}
    return out


def pmc_traffic(kind):
    """tools/traffic_from_pmc.py derive(kind) -- HBM bytes per launch from the committed PMC summaries -- or None without them."""
    import importlib.util
    try:
        spec = importlib.util.spec_from_file_location("traffic_from_pmc", os.path.join(ROOT, "tools", "traffic_from_pmc.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod.derive(kind)
    except (OSError, KeyError, ValueError):
        return None


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py <same
    arguments>` as a CHILD process (never exec: this parent has not touched the GPU and must not), one rank per GPU over RCCL,
    pass rank 0's JSON line through on stdout, and return the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:  # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for ln in child.stdout:  # ranks other than 0 print nothing on stdout; the launcher's own chatter goes to stderr
        if ln.lstrip().startswith("{"):
            sys.stdout.write(ln)
            sys.stdout.flush()
        else:
            sys.stderr.write(ln)
    return child.wait()


def measure(sisr, workload, B, steps, warmup, use_graph, rank, world, local, dev, families=False, dp=False):
    """Build the handler, run warm-up + exactly `steps` timed steps (barrier + sync on both sides, max over ranks)."""
    name, params, tflop_per_patch = WORKLOADS[workload]
    params = dict(params)
    maps = params.pop("maps", False)
    hr = 128 if params.pop("hr_same", False) else 512
    if name not in sisr.available_models:
        raise SystemExit(f"workload {name} is not built yet")
    torch.manual_seed(8)
    h = sisr.available_models[name](device=local, model_save_dir="/tmp", eval_mode=False, scale=4, lr=1e-4,
                                    scheduler="cosine_annealing_warm_restarts",
                                    scheduler_params={"t_mult": 1, "restart_period": 125000, "lr_min": 1e-7}, **params)
    if world > 1 or dp:
        h.set_multi_gpu()
    h.use_graph = use_graph
    g = torch.Generator().manual_seed(8 + rank)
    x = torch.rand(B, 3, 128, 128, generator=g).to(dev)
    y = torch.rand(B, 3, hr, hr, generator=g).to(dev)
    kw = {}
    if "metadata" in params:
        code = torch.rand(B, 10, 1, 1, generator=g) * 0.4
        kw["extra_channels"] = (code.expand(B, 10, 128, 128).contiguous() if maps else code).to(dev)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        h.train_step(x, y, **kw)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = h.train_step(x, y, **kw)
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    res = {"value": world * B * steps / dt, "ms_per_step": 1e3 * dt / steps, "loss": float(loss.item()),
           "tflop_per_patch": tflop_per_patch, "width": params.get("num_features", 64), "name": name, "params": params}
    if families:  # one extra, untimed, eager step with every kernel alone on the stream
        timer = KernelTimer(sisr.ops, width=res["width"])
        timer.install()
        side, sisr.ops.WGRAD_SIDE_STREAM = sisr.ops.WGRAD_SIDE_STREAM, False
        h.use_graph = False
        try:
            timer.on = True
            h.train_step(x, y, **kw)
            timer.on = False
            torch.cuda.synchronize()
        finally:
            sisr.ops.WGRAD_SIDE_STREAM = side
            timer.remove()
        res["timer"] = timer
    if h.reducer is not None:
        res["grad_exchange"] = (("hipGraph replay; buckets released by signal nodes of the replay, all-reduce on the reducer stream beside the "
                                 "rest of the captured backward (SISR_GRAPH_OVERLAP=%s)" % os.environ.get("SISR_GRAPH_OVERLAP", "auto"))
                                if use_graph and h.reducer.can_signal() else
                                "hipGraph replay; all buckets all-reduced at the join after the replay" if use_graph else
                                "eager; bucket all-reduce issued from gradient hooks on the reducer stream, overlapped with backward")
        h.remove_multi_gpu()  # graphs first, then the reducer's hooks, progress words and sinks
    del h, x, y, kw
    torch.cuda.empty_cache()
    return res


def measure_eval(sisr, workload, B, steps, warmup, local, dev):
    """Forward-only throughput through the handler's run_eval (the reference's inference entry point, ref
    SISR/models/__init__.py:491-533), output left on the device: weight packing + forward under no_grad per call."""
    name, params, tflop_per_patch = WORKLOADS[workload]
    params = dict(params)
    params.pop("maps", None)
    params.pop("hr_same", None)
    torch.manual_seed(8)
    h = sisr.available_models[name](device=local, model_save_dir="/tmp", eval_mode=True, scale=4, **params)
    g = torch.Generator().manual_seed(8)
    x = torch.rand(B, 3, 128, 128, generator=g).to(dev)
    for _ in range(warmup):
        h.run_eval(x, keep_on_device=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        h.run_eval(x, keep_on_device=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del h, x
    torch.cuda.empty_cache()
    v = B * steps / dt
    return {"workload": f"{name.upper()} x4 full depth, forward only (run_eval, output kept on the device), {B} tiles of 128x128",
            "value": v, "unit": "patches/s", "ms_per_batch": 1e3 * dt / steps, "steps": steps, "warmup": warmup,
            "algorithmic_tflops": v * tflop_per_patch / 3.0,
            "frac_of_fp32_mfma_peak": v * tflop_per_patch / 3.0 / FP32_MFMA_PEAK_TFLOPS}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="LR patches per GPU per step (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="LR patches per step over ALL GPUs (strong scaling); default 32 when N > 1 (BASELINE config 4)")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS), help="default: rcan at N = 1, qrcan at N > 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the meta_rcan / weak_scaling side measurement")
    ap.add_argument("--graph", nargs="?", const="on", default="auto", choices=["auto", "on", "off"],
                    help=f"replay forward+backward as a hipGraph (auto: per-GPU batch <= {GRAPH_MAX_BATCH})")
    ap.add_argument("--force-dp", action="store_true",
                    help="N = 1 only: run through a one-rank RCCL world (GradReducer buckets, all_reduce, join)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "bf16x3"],
                    help="arithmetic of the 64-channel convs: fp32 MFMA (reference arithmetic, the headline); bf16 MFMA "
                         "operands with fp32 accumulate / storage (BASELINE config 'HAN x4 bf16'); bf16x3 = fp32 operands "
                         "split exactly into three bf16 numbers, six products on the bf16 MFMA (fp32-class error)")
    ap.add_argument("--storage", default="0", choices=["0", "act", "all"],
                    help="with --precision bf16: storage of the maps a residual group keeps: 0 = fp32, act = bf16 activations, "
                         "all = bf16 activations and gradient maps")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started bare (`python bench.py --gpus N`, the way the reference turns multi-GPU on with a flag and no launcher,
        # ref: Code/SISR/models/__init__.py:121-122, net_train.py:20): this process starts the N ranks itself
        raise SystemExit(launch_ranks(args.gpus))

    import importlib
    sisr = importlib.import_module("sisr_amd")
    sisr.ops.set_precision(args.precision)
    if args.storage != "0":
        if args.precision != "bf16":
            raise SystemExit("--storage act / all needs --precision bf16")
        sisr.ops.set_storage(args.storage)
    rank, world, local = sisr.parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus} (or start bench.py bare: it launches its own ranks)")
    if os.environ.get("SISR_BENCH_SHARE_GPU"):  # rehearsal on a 1-GPU box: every rank on cuda:0 (gloo backend)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.force_dp and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)

    explicit = args.workload is not None or args.batch is not None or args.global_batch is not None
    workload = args.workload or ("rcan" if world == 1 else "qrcan")
    if args.batch is not None and args.global_batch is not None:
        raise SystemExit("--batch (per GPU) and --global-batch are exclusive")
    if args.batch is not None:
        B, scaling = args.batch, "weak"
    else:
        G = args.global_batch if args.global_batch is not None else 32
        if G % world:
            raise SystemExit(f"global batch {G} is not divisible by {world} GPUs")
        B = G // world
        scaling = "weak" if (world == 1 and args.global_batch is None) else "strong"
    # (SPARNet: ~3 000 launches of a few microseconds per step at any batch: always launch-bound)
    use_graph = args.graph == "on" or (args.graph == "auto" and (B <= GRAPH_MAX_BATCH or WORKLOADS[workload][1].get("hr_same")))

    main_res = measure(sisr, workload, B, args.steps, args.warmup, use_graph, rank, world, local, dev,
                       families=not args.no_kernel_timing and args.precision == "fp32", dp=args.force_dp)
    secondary = None
    if not explicit and not args.no_secondary and args.precision == "fp32":
        # N = 1: the north_star family beside BASELINE's configs[1]; N > 1: the weak-scaling point beside config 4
        ssteps = max(2, min(args.steps, 5))
        skey = "meta_rcan" if world == 1 else "weak_scaling"
        try:
            s = measure(sisr, "qrcan", 32, ssteps, min(args.warmup, 2), False, rank, world, local, dev)
            secondary = (skey, {
                "workload": "QRCAN (RCAN + meta-attention, style 'standard', q-layers on) x4 full depth, same train step",
                "value": s["value"], "unit": "patches/s", "ms_per_step": s["ms_per_step"], "steps": ssteps,
                "per_gpu_batch": 32, "global_batch": 32 * world, "n_gpus": world,
                "algorithmic_tflops_per_gpu": s["value"] / world * s["tflop_per_patch"],
                "frac_of_fp32_mfma_peak": s["value"] / world * s["tflop_per_patch"] / FP32_MFMA_PEAK_TFLOPS})
        except Exception as e:  # the side measurement must not cost the run its headline line
            secondary = (skey, {"error": repr(e)})

    extras = {}

    def guarded(key, fn):
        """An auxiliary measurement must not cost the run its headline line: an error is recorded under the object's own key."""
        try:
            extras[key] = fn()
        except Exception as e:
            extras[key] = {"error": repr(e)}
            sisr.ops.set_precision("fp32")

    aux = not explicit and not args.no_secondary and args.precision == "fp32" and world == 1

    def point(s, B, what, graph):
        tf = s["value"] * s["tflop_per_patch"]
        return {"workload": what, "value": s["value"], "unit": "patches/s", "ms_per_step": s["ms_per_step"], "per_gpu_batch": B,
                "hip_graph": graph, "algorithmic_tflops": tf, "frac_of_fp32_mfma_peak": tf / FP32_MFMA_PEAK_TFLOPS}

    def config4():
        # BASELINE config 4's per-GPU operating point (global batch 32 over 8 GPUs = 4 tiles each), on this one GPU: the
        # reducer, its buckets and the all-reduce run through a one-rank RCCL world, forward + backward replay from a hipGraph
        # (58 ms steps: 8 warm-up steps -- graph capture, clocks settling after the 400 ms steps before -- and 30 timed ones;
        # with 3 / 10 the same code reads 1.5 - 2 patches/s lower)
        c4_steps, c4_warm = max(2, min(3 * args.steps, 30)), max(min(args.warmup, 3), 8 if args.warmup else 0)
        s4 = measure(sisr, "qrcan", 4, c4_steps, c4_warm, True, rank, world, local, dev,
                     families=not args.no_kernel_timing, dp=True)
        tf4 = s4["value"] * s4["tflop_per_patch"]
        return {"workload": "QRCAN (RCAN + meta-attention) x4 full depth, 4 tiles of 128x128 per GPU (BASELINE config 4: global "
                            "batch 32 on 8 GPUs), one-rank RCCL world, forward+backward replayed from a hipGraph",
                "value": s4["value"], "unit": "patches/s", "ms_per_step": s4["ms_per_step"], "per_gpu_batch": 4,
                "steps": c4_steps, "warmup": c4_warm, "hip_graph": True, "parallelism": "dp1 (one-rank RCCL world: buckets, all_reduce, join)",
                "grad_exchange": s4.get("grad_exchange"), "sample_lanes": sisr.ops.LANES,
                "roofline": {"bound": "mfma", "achieved": tf4, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": tf4 / FP32_MFMA_PEAK_TFLOPS,
                             "families": s4["timer"].summary(FP32_MFMA_PEAK_TFLOPS) if "timer" in s4 else []}}

    def config4_overlapped():
        # the same point with the exchange overlapped with the replayed backward (opt-in: SISR_GRAPH_OVERLAP=1)
        prev = os.environ.get("SISR_GRAPH_OVERLAP")
        os.environ["SISR_GRAPH_OVERLAP"] = "1"
        try:
            s4o = measure(sisr, "qrcan", 4, max(2, min(args.steps, 5)), 2, True, rank, world, local, dev, dp=True)
        finally:
            if prev is None:
                os.environ.pop("SISR_GRAPH_OVERLAP")
            else:
                os.environ["SISR_GRAPH_OVERLAP"] = prev
        return {"what": "SISR_GRAPH_OVERLAP=1: every bucket's all-reduce is released by a signal node of the replay and runs on the "
                        "reducer stream beside the rest of the captured backward (in a one-rank world there is nothing to hide: this "
                        "is the cost of the mechanism)", "value": s4o["value"], "unit": "patches/s", "ms_per_step": s4o["ms_per_step"],
                "grad_exchange": s4o.get("grad_exchange")}

    def han_bf16():
        # BASELINE config 5: HAN x4, bf16 matrix-core operands (opt-in mode), 32 tiles, forward + backward
        # replayed from a hipGraph (the eager step is host-bound in this mode: ~3 300 launches against < 100 ms of GPU work),
        # everything a residual group keeps or hands on stored as bf16; the same step with fp32 maps / bf16 activations beside it
        sisr.ops.set_precision("bf16")
        runs = {}
        try:
            n_t, n_w = max(2, min(args.steps, 8)), max(min(args.warmup, 2), 3 if args.warmup else 0)
            for st in ("0", "act", "all"):
                sisr.ops.set_storage(st)
                runs[st] = measure(sisr, "han", 32, n_t, n_w, True, rank, world, local, dev)
        finally:
            sisr.ops.set_storage("0")
            sisr.ops.set_precision("fp32")
        sh = runs["all"]
        gb_per_patch = HBM_GB_PER_PATCH_ALL16["han"]
        gbs = sh["value"] * gb_per_patch
        t16 = pmc_traffic("bf16_all")  # measured / algorithmic HBM bytes per launch of the bf16-storage kernels (PMC CSVs)
        tfam16 = t16["families_b32"] if t16 is not None else None
        beside = {name: {"value": runs[st]["value"], "unit": "patches/s", "ms_per_step": runs[st]["ms_per_step"],
                         "hbm_gb_per_patch": STORAGE_TABLE[st]["han"],
                         "frac_of_hbm_peak": runs[st]["value"] * STORAGE_TABLE[st]["han"] / HBM_PEAK_GBS}
                  for st, name in (("0", "fp32_map_storage"), ("act", "bf16_activation_storage"))}
        return {"workload": "HAN x4 full depth, 32 tiles of 128x128, train step, forward+backward replayed from a hipGraph "
                            "(BASELINE config 5)",
                "dtype": "bf16 MFMA operands, f32 accumulate; inside the residual groups every kept activation (t1, t2, gated "
                         "skips) and every gradient map handed from launch to launch is stored as bf16 in HBM; group inputs / "
                         "outputs, partial sums, gates, weights' master copies and optimiser state f32",
                "value": sh["value"], "unit": "patches/s", "ms_per_step": sh["ms_per_step"], "per_gpu_batch": 32, "hip_graph": True,
                "final_loss": sh["loss"], **beside,
                "roofline": {"bound": "hbm", "what": "whole step: algorithmic bytes per patch of the storage format in use "
                             "(%.1f GB per patch fwd+bwd) x patches/s" % gb_per_patch, "achieved": gbs, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic_families": tfam16,
                             "note": "halving the bytes did not halve the launch times: with bf16 maps a 64->64 launch moves 67 - 201 MB "
                                     "in 58 - 80 us while its MFMA time (15 us), its LDS fragment reads and its weight stream through "
                                     "the vector L1 each need about the same issue time per tile -- the kernels are no longer HBM-bound "
                                     "(DESIGN.md section 7), so the fraction of the HBM roof FALLS as the format shrinks",
                             "secondary": {"bound": "mfma", "achieved": sh["value"] * sh["tflop_per_patch"], "peak": 2500.0,
                                           "unit": "TFLOP/s", "frac": sh["value"] * sh["tflop_per_patch"] / 2500.0}},
                "parity": "unpinned against the reference (it has no reduced-precision mode); every bf16-storage launch is pinned "
                          "bit for bit to the bf16-operand kernels, whole nets statistically to the oracle's restatement of the "
                          "rounding, Set5 PSNR and training losses to fp32 (tests/test_bf16_storage_gpu.py, tests/test_bf16_gpu.py)"}

    def meta_edsr():
        # BASELINE config 3: Meta-EDSR (QEDSR: EDSR-baseline + a meta-attention layer per block, blur-kernel vector), 32 tiles
        s = measure(sisr, "qedsr", 32, max(2, min(args.steps, 10)), min(args.warmup, 3), False, rank, world, local, dev)
        return point(s, 32, "QEDSR (Meta-EDSR, BASELINE config 3) x4, 16 blocks x 64 features, 32 tiles of 128x128, train step", False)

    def b1_point():
        # the 1-tile-per-GPU operating point SURVEY 8 puts configs 2 and 3 at, and the batch cpu_baseline is timed at
        out = {}
        for wl, what in (("rcan", "RCAN x4 full depth"), ("qedsr", "QEDSR (Meta-EDSR) x4, 16 blocks x 64 features")):
            s = measure(sisr, wl, 1, max(2, min(3 * args.steps, 40)), max(min(args.warmup, 3), 5 if args.warmup else 0), True,
                        rank, world, local, dev)
            out[wl] = point(s, 1, what + ", ONE 128x128 tile per step, forward+backward replayed from a hipGraph", True)
        return out

    def sparnet_point():
        # SURVEY 8 row f4 ("then SPARNet"): the default SPARNet plan, 16 face images of 128 x 128 per step (input and output both at
        # HR size), forward + backward replayed from a hipGraph; images/s, with the fraction of the fp32 matrix peak beside it
        s = measure(sisr, "sparnet", 16, max(2, min(2 * args.steps, 40)), max(min(args.warmup, 3), 5 if args.warmup else 0), True,
                    rank, world, local, dev)
        p = point(s, 16, "SPARNet (default plan: 32 / 64 / 128 features, 10 body blocks, hourglass attention), 16 images of 128x128, "
                         "train step (fwd + L1 + bwd + Adam)", True)
        p["unit"] = "images/s"
        return p

    if aux:
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
            own_group = True
        else:
            own_group = False
        guarded("config4_point", config4)
        if "error" not in extras["config4_point"]:
            guarded("_c4o", config4_overlapped)
            extras["config4_point"]["overlapped_grad_exchange"] = extras.pop("_c4o")
        if own_group:  # the one-rank world existed for config 4 only; later measurements run single-process
            dist.destroy_process_group()
        guarded("han_bf16", han_bf16)
        guarded("meta_edsr", meta_edsr)
        guarded("b1_point", b1_point)
        guarded("sparnet", sparnet_point)
        guarded("inference", lambda: measure_eval(sisr, "rcan", 32, max(2, min(args.steps, 10)), min(args.warmup, 3), local, dev))

        def bf16x3():
            # second line asked for by the round-1 review: the same step with the convs as bf16x3 splits
            sisr.ops.set_precision("bf16x3")
            try:
                s3 = measure(sisr, workload, B, max(2, min(args.steps, 5)), max(args.warmup, 3), False, rank, world, local, dev)
            finally:
                sisr.ops.set_precision("fp32")
            return {"dtype": "f32 via bf16x3 split (six bf16 MFMA products per f32 product), f32 accumulate and storage",
                    "what": "same workload and step as the headline line, opt-in arithmetic: NOT the headline (DESIGN.md section 7b)",
                    "value": s3["value"], "unit": "patches/s", "ms_per_step": s3["ms_per_step"], "final_loss": s3["loss"],
                    "speedup_vs_headline": s3["value"] / main_res["value"],
                    "algorithmic_tflops": s3["value"] * s3["tflop_per_patch"]}
        guarded("bf16x3", bf16x3)

    def emit():
        """Rank 0 prints the ONE JSON line (everything measured so far)."""
        if rank != 0:
            return
        name, params, value = main_res["name"], main_res["params"], main_res["value"]
        tflop_per_patch = main_res["tflop_per_patch"]
        label = name.upper() + (f" ({params['num_features']} features, {params['num_blocks']} blocks)"
                                if "num_features" in params else "")
        line = {
            "metric": "LR-patches/sec (128x128x3, x4) fwd+bwd", "value": value, "unit": "patches/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"],
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16": "bf16 MFMA operands, f32 accumulate and storage",
                      "bf16x3": "f32 via bf16x3 split (six bf16 MFMA products per f32 product), f32 accumulate and storage"}[args.precision],
            "data": "synthetic",
            "config": {"workload": (f"{label} full depth, 128x128 interpolated input -> 128x128 output, train step = "
                                    f"fwd + L1 + bwd + Adam + scheduler" if WORKLOADS[workload][1].get("hr_same") else
                                    f"{label} x4 full depth, 128x128 LR -> 512x512 tiles, train step = "
                                    f"fwd + L1 + bwd + Adam + scheduler"), "per_gpu_batch": B, "global_batch": B * world,
                       "parallelism": f"dp{world}" + (" (one-rank RCCL world)" if args.force_dp and world == 1 else ""),
                       "hip_graph": bool(use_graph), "grad_exchange": main_res.get("grad_exchange"),
                       "sample_lanes": sisr.ops.LANES if use_graph else 1, "final_loss": main_res["loss"],
                       "algorithmic_tflops": value * tflop_per_patch},
        }
        step_tf = value / world * tflop_per_patch
        if args.precision == "fp32":
            fams = main_res["timer"].summary(FP32_MFMA_PEAK_TFLOPS) if "timer" in main_res else []
            traffic, tsrc, tfam = None, None, None
            if main_res["width"] == 64 and B == 32:
                # measured HBM bytes per launch of every family beside what it must move: recomputed here from the committed
                # rocprofv3 --pmc summaries with the one unit rule tools/traffic_from_pmc.py states
                tdoc = pmc_traffic("fp32")
                if tdoc is not None:
                    traffic, tfam = tdoc["32"], tdoc["families_b32"]
                    tsrc = {"csv": tdoc["source"], "unit_rule": tdoc["unit_rule"], "derived_by": tdoc["derived_by"]}
            line["roofline"] = {
                "bound": "mfma", "what": "whole training step: algorithmic TFLOP/s per GPU (patches/s x TFLOP per patch)",
                "achieved": step_tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": step_tf / FP32_MFMA_PEAK_TFLOPS,
                "traffic": traffic, "traffic_source": tsrc, "traffic_families": tfam,
                "kernel": fams[0]["family"] if fams else None, "families": fams,
            }
            if workload in HBM_GB_PER_PATCH:
                gbs = value / world * HBM_GB_PER_PATCH[workload]
                line["roofline"]["secondary"] = {"bound": "hbm", "what": "whole step, algorithmic bytes per patch x "
                                                 "patches/s per GPU", "achieved": gbs, "peak": HBM_PEAK_GBS,
                                                 "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
        elif args.precision == "bf16x3":
            peak = 2500.0 / 6.0  # dense bf16 MFMA peak / six products per fp32 product
            line["roofline"] = {"bound": "mfma", "what": "whole step, fp32-equivalent algorithmic TFLOP/s per GPU against the "
                                "dense bf16 MFMA peak / 6", "achieved": step_tf, "peak": peak, "unit": "TFLOP/s",
                                "frac": step_tf / peak, "traffic": None,
                                "vs_fp32_mfma_peak": step_tf / FP32_MFMA_PEAK_TFLOPS}
        else:
            # bf16 mode: every conv moves two maps per launch and ~15 us of MFMA: HBM-bound
            gbs = value / world * STORAGE_TABLE[args.storage].get(workload, HBM_GB_PER_PATCH.get(workload, 16.5))
            line["roofline"] = {"bound": "hbm", "what": "whole step, algorithmic bytes per patch of the storage format in use x patches/s per GPU",
                                "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                "traffic": None,
                                "secondary": {"bound": "mfma", "achieved": step_tf, "peak": 2500.0, "unit": "TFLOP/s",
                                              "frac": step_tf / 2500.0}}
        if secondary is not None:
            line[secondary[0]] = secondary[1]
        line.update(extras)
        if world == 1 and not args.no_cpu_baseline and args.precision == "fp32":
            line["cpu_baseline"] = cpu_baseline(workload)
        print(json.dumps(line), flush=True)

    if world > 1 and use_graph and not explicit and not args.no_secondary and args.precision == "fp32":
        # The replayed step's exchange overlapped with its backward (signal nodes + wait kernels, SISR_GRAPH_OVERLAP=1) has never
        # met a multi-rank RCCL world: measured LAST, under a watchdog that prints the headline line (with a note) and leaves if
        # the attempt does not come back -- a hang here must not cost the scaling run its line.
        import threading

        def give_up():
            extras["overlapped_grad_exchange"] = {"error": "no result within 180 s; the default (all buckets all-reduced at the join "
                                                           "after the replay) is what the headline line ran"}
            emit()
            sys.stdout.flush()
            os._exit(0)

        dog = threading.Timer(180.0, give_up)
        dog.daemon = True
        dog.start()

        def overlapped():
            prev = os.environ.get("SISR_GRAPH_OVERLAP")
            os.environ["SISR_GRAPH_OVERLAP"] = "1"
            try:
                so = measure(sisr, workload, B, max(2, min(args.steps, 10)), max(2, min(args.warmup, 4)), True, rank, world, local, dev)
            finally:
                if prev is None:
                    os.environ.pop("SISR_GRAPH_OVERLAP")
                else:
                    os.environ["SISR_GRAPH_OVERLAP"] = prev
            return {"what": "same workload with SISR_GRAPH_OVERLAP=1", "value": so["value"], "unit": "patches/s",
                    "ms_per_step": so["ms_per_step"], "grad_exchange": so.get("grad_exchange")}
        guarded("overlapped_grad_exchange", overlapped)
        dog.cancel()

    emit()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
