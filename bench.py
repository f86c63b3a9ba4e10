#!/usr/bin/env python3
"""Headline benchmark: LR-patches/s (128x128x3 -> x4) for one training step on N MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step is BaseModel.train_step on the full-depth network: forward + L1 + backward + Adam + scheduler step
(the reference's run_train, Code/SISR/models/__init__.py:466-489, minus its per-step D2H copies), on
synthetic inputs already resident in HBM.  Per-GPU batch is fixed, so scaling is weak;
value = all patches of all ranks / max-over-ranks wall time of exactly K steps.

Extra objects on the JSON line:
  roofline      the dominant kernel (conv3x3_c64_kernel, 64->64 body shape): algorithmic FLOPs per launch
                divided by its mean launch duration, measured with HIP events around every launch of that
                shape in the forward pass of the last timed step (backward convs overlap the side-stream weight
                gradients, so only forward launches run alone), against the 157.3 TFLOP/s fp32 matrix/vector peak.
  cpu_baseline  rank 0, N=1 only: the CPU oracle (a restatement of the reference proven equal to it by the
                golden vectors) doing the same step at batch 1 on the host cores -- a bounded sample.
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
HBM_GB_PER_PATCH = {"rcan": 16.5, "qrcan": 16.5, "edsr": 1.4, "qedsr": 1.4}
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
}


class ConvTimer:
    """HIP-event timing of every 64->64 body conv launch (recorded on the launch stream)."""

    def __init__(self, ops, width=64):
        self.ops, self.orig, self.events, self.on, self.width = ops, ops.conv_c64, [], False, width

    def install(self):
        def timed(x, xview, packed, bias, bnq, y, yview, B, H, W, cin, cout, **kw):
            # forward launches only: in backward the data-gradient convs share the GPU with the
            # weight-gradient kernels of the side stream, so their individual durations say nothing about the kernel
            # ... and only the plain instantiation: launches that also build the gated skip in their staging
            # (GATE) do a second pass's work that the algorithmic FLOP count does not credit
            if (self.on and cin == self.width and cout == self.width and not self.ops.IN_BACKWARD
                    and kw.get("gate_add") is None):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self.orig(x, xview, packed, bias, bnq, y, yview, B, H, W, cin, cout, **kw)
                e1.record()
                self.events.append((e0, e1, B * H * W))
            else:
                self.orig(x, xview, packed, bias, bnq, y, yview, B, H, W, cin, cout, **kw)
        self.ops.conv_c64 = timed

    def summary(self):
        if not self.events:
            return None
        ms = [a.elapsed_time(b) for a, b, _ in self.events]
        flop = [p * CONV_BODY_FLOP_PER_PIXEL * (self.width // 64) ** 2 for _, _, p in self.events]
        return {"launches": len(ms), "avg_us": 1e3 * sum(ms) / len(ms), "tflops": sum(flop) / (sum(ms) * 1e-3) / 1e12,
                "flop_per_launch": sum(flop) / len(flop)}


def cpu_baseline(workload, seconds_budget=30.0):
    """Oracle (CPU restatement of the reference) timed on the host cores: same step, batch 1."""
    import importlib
    from oracle import sisr_oracle as O
    sisr = importlib.import_module("sisr_amd")
    name, params, _ = WORKLOADS[workload]
    torch.set_num_threads(min(16, os.cpu_count() or 1))  # the GPU box grants 16 host cores per GPU
    torch.manual_seed(8)
    h = sisr.available_models[name](device=torch.device("cpu"), model_save_dir="/tmp", eval_mode=False, scale=4, **params)
    cfg = {"rcan": dict(n_resgroups=10, n_resblocks=20, scale=4),
           "edsr": dict(num_blocks=params.get("num_blocks", 16), scale=4, res_scale=0.1),
           "qrcan": dict(n_resgroups=10, n_resblocks=20, scale=4, style="standard", include_q_layer=True),
           "qedsr": dict(num_blocks=16, scale=4, res_scale=0.1, q_layer_nonlinearity=False),
           "han": dict(n_resgroups=10, n_resblocks=20, scale=4), "qhan": dict(n_resgroups=10, n_resblocks=20, scale=4),
           "san": dict(n_resgroups=20, n_resblocks=10, scale=4), "qsan": dict(n_resgroups=20, n_resblocks=10, scale=4)}[name]
    tr = O.Trainer(name, h.net.state_dict(), lr=1e-4, **cfg)
    g = torch.Generator().manual_seed(8)
    x, y = torch.rand(1, 3, 128, 128, generator=g), torch.rand(1, 3, 512, 512, generator=g)
    md = (torch.rand(1, 10, 1, 1, generator=g) * 0.4) if name in O.META_NETS else None
    tr.step(x, y, md)  # warm-up
    times = []
    t_all = time.perf_counter()
    while len(times) < 3 and (time.perf_counter() - t_all) < seconds_budget:
        t0 = time.perf_counter()
        tr.step(x, y, md)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": 1.0 / med, "unit": "patches/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{name} x4 full depth, batch 1, 128x128 LR tile, fwd+L1+bwd+Adam, median of {len(times)} "
                      f"timed steps after 1 warm-up (oracle/sisr_oracle.py on torch CPU kernels)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="LR patches per GPU per step")
    ap.add_argument("--workload", default="rcan", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay forward+backward as a hipGraph (small batches)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"],
                    help="arithmetic of the 64-channel convs: fp32 MFMA (reference arithmetic, the headline) or bf16 "
                         "MFMA operands with fp32 accumulate / storage (BASELINE config 'HAN x4 bf16')")
    args = ap.parse_args()

    import importlib
    sisr = importlib.import_module("sisr_amd")
    sisr.ops.set_precision(args.precision)
    rank, world, local = sisr.parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if os.environ.get("SISR_BENCH_SHARE_GPU"):  # rehearsal on a 1-GPU box: every rank on cuda:0 (gloo backend)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    name, params, tflop_per_patch = WORKLOADS[args.workload]
    if name not in sisr.available_models:
        raise SystemExit(f"workload {name} is not built yet")

    torch.manual_seed(8)
    h = sisr.available_models[name](device=local, model_save_dir="/tmp", eval_mode=False, scale=4, lr=1e-4,
                                    scheduler="cosine_annealing_warm_restarts",
                                    scheduler_params={"t_mult": 1, "restart_period": 125000, "lr_min": 1e-7}, **params)
    if world > 1:
        h.set_multi_gpu()
    if args.graph:
        h.use_graph = True
        args.no_kernel_timing = True
        if h.reducer is not None:
            h.reducer.remove()
            h.reducer.overlap = False
    B = args.batch
    g = torch.Generator().manual_seed(8 + rank)
    x = torch.rand(B, 3, 128, 128, generator=g).to(dev)
    y = torch.rand(B, 3, 512, 512, generator=g).to(dev)
    kw = {}
    if "metadata" in params:
        kw["extra_channels"] = (torch.rand(B, 10, 1, 1, generator=g) * 0.4).to(dev)

    timer = ConvTimer(sisr.ops, width=params.get("num_features", 64))
    if not args.no_kernel_timing:
        timer.install()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        h.train_step(x, y, **kw)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        timer.on = (i == args.steps - 1)
        loss, _ = h.train_step(x, y, **kw)
    timer.on = False
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    loss_val = float(loss.item())

    if rank == 0:
        label = name.upper() + (f" ({params['num_features']} features, {params['num_blocks']} blocks)"
                                if "num_features" in params else "")
        value = world * B * args.steps / dt
        line = {
            "metric": "LR-patches/sec (128x128x3, x4) fwd+bwd", "value": value, "unit": "patches/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16 MFMA operands, f32 accumulate and storage",
            "data": "synthetic",
            "config": {"workload": f"{label} x4 full depth, 128x128 LR -> 512x512 tiles, train step = "
                                   f"fwd + L1 + bwd + Adam + scheduler", "per_gpu_batch": B, "global_batch": B * world,
                       "parallelism": f"dp{world}", "hip_graph": bool(args.graph), "final_loss": loss_val,
                       "algorithmic_tflops": value * tflop_per_patch},
        }
        ks = timer.summary()
        if ks and args.precision == "bf16":
            # the bf16 conv is HBM-bound: one fp32 map in, one out (weights / bias are L2-resident)
            nbytes = 2 * (ks["flop_per_launch"] / (CONV_BODY_FLOP_PER_PIXEL * (timer.width // 64) ** 2)) * timer.width * 4
            gbs = nbytes / (ks["avg_us"] * 1e-6) / 1e9
            traffic = None
            tj = os.path.join(ROOT, "profiles", "traffic_conv3x3_c64_bf16.json")
            if os.path.exists(tj) and timer.width == 64:
                with open(tj) as f:
                    traffic = json.load(f).get(str(B))
            line["roofline"] = {"bound": "hbm", "kernel": "conv3x3_c64_bf16_kernel (64->64 body conv, forward launches)",
                                "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                "traffic": traffic, "avg_launch_us": ks["avg_us"], "launches_timed": ks["launches"],
                                "bytes_per_launch": nbytes, "mfma_tflops": ks["tflops"]}
        elif ks:
            traffic = None
            tj = os.path.join(ROOT, "profiles", "traffic_conv3x3_c64.json")
            if os.path.exists(tj):
                with open(tj) as f:
                    traffic = json.load(f).get(str(B))
            line["roofline"] = {"bound": "mfma",
                                "kernel": f"conv3x3_c64_v4_kernel<0,0,0,2,0,0> ({timer.width}->{timer.width} body conv, "
                                          f"forward launches without fused prologue)",
                                "achieved": ks["tflops"], "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": ks["tflops"] / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                                "avg_launch_us": ks["avg_us"], "launches_timed": ks["launches"],
                                "flop_per_launch": ks["flop_per_launch"]}
        if "roofline" in line and args.workload in HBM_GB_PER_PATCH:
            # the other roof, for reference (SURVEY.md §8d asks for both): whole-step algorithmic HBM rate
            gbs = value / world * HBM_GB_PER_PATCH[args.workload]
            line["roofline"]["secondary"] = {"bound": "mfma" if args.precision == "bf16" else "hbm",
                                             "what": "whole step, algorithmic bytes per patch x patches/s per GPU",
                                             "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": gbs / HBM_PEAK_GBS} if args.precision == "fp32" else {
                "bound": "mfma", "what": "whole step, algorithmic TFLOP/s per GPU against the dense bf16 MFMA peak",
                "achieved": value / world * tflop_per_patch, "peak": 2500.0, "unit": "TFLOP/s",
                "frac": value / world * tflop_per_patch / 2500.0}
        if world == 1 and not args.no_cpu_baseline and args.precision == "fp32":
            line["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
