#!/usr/bin/env python3
"""Micro-benchmark of the hot kernels through the C ABI (HIP-event timing, one process).

    python tools/kbench.py [--batch 8] [--iters 20] [--only conv,wgrad,...]
Prints one JSON line per kernel: mean launch time and achieved TFLOP/s (or GB/s) on 128x128x64 maps.
Used for interleaved A/B of kernel variants and as the target of rocprofv3 --pmc passes.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sisr_amd  # noqa: E402

ops, hip = sisr_amd.ops, sisr_amd.hip


WARM = [3]


def timeit(fn, iters, warm=None):
    for _ in range(WARM[0] if warm is None else warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--hw", type=int, default=128)
    ap.add_argument("--warm", type=int, default=3, help="untimed launches before each timed loop (clocks settle over tens of ms)")
    ap.add_argument("--only", default="")
    ap.add_argument("--variants", default="4,2", help="conv kernel selections to time: 4 auto, 5 / 6 forced 4-row / 2-row tile, 2 general")
    ap.add_argument("--rounds", type=int, default=0, help="interleaved A/B rounds over --variants (conv only)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "bf16x3"])
    a = ap.parse_args()
    ops.set_precision(a.precision)
    WARM[0] = a.warm
    if a.precision != "fp32":
        a.variants = "4"  # one bf16 conv kernel; report GB/s of the two fp32 maps beside TFLOP/s
    only = set(a.only.split(",")) if a.only else None
    B, H, W = a.batch, a.hw, a.hw
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    cl = torch.channels_last
    x = torch.randn(B, 64, H, W, generator=g).to(dev).contiguous(memory_format=cl)
    dy = torch.randn(B, 64, H, W, generator=g).to(dev).contiguous(memory_format=cl)
    t1 = torch.relu(torch.randn(B, 64, H, W, generator=g)).to(dev).contiguous(memory_format=cl)
    y = torch.empty_like(x)
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev)
    b = torch.randn(64, generator=g).to(dev)
    sc = torch.rand(B, 64, generator=g).to(dev)
    sh = torch.rand(B, 64, generator=g).to(dev)
    v = hip.view_plain(H, W, 64)
    pk = ops.pack_weight(w, "fwd")
    gap = torch.empty(B, ops.gap_parts(H, W), 64, device=dev)
    dw, db = torch.empty_like(w), torch.empty(64, device=dev)
    flop = 2.0 * B * H * W * 64 * 64 * 9
    fm = B * H * W * 64 * 4
    res = {}

    def run(name, fn, work, unit):
        if only and name not in only and name.split("_v")[0] not in only:
            return
        t = timeit(fn, a.iters)
        res[name] = {"us": t * 1e6, unit: work / t / (1e12 if unit == "TFLOP/s" else 1e9)}
        if a.precision == "bf16" and unit == "TFLOP/s":
            res[name]["GB/s (2 fp32 maps)"] = 2 * fm / t / 1e9
        print(json.dumps({"kernel": name, "batch": B, **res[name]}), flush=True)

    variants = [int(t) for t in a.variants.split(",")]
    if a.precision == "bf16" and a.rounds > 0:
        # interleaved A/B of the persistent (1) and per-tile (0) bf16 conv kernels, plain and dgrad-with-mask forms
        SEL = [0]
        forms = {"conv": lambda: ops.conv_c64(x, v, pk, b, (1, 64), y, v, B, H, W, 64, 64, select=SEL[0]),
                 "conv_dgrad2": lambda: ops.conv_c64(dy, v, pk, None, (1, 64), y, v, B, H, W, 64, 64, mask=t1, in_scale=sc,
                                                     in_shift=sh, select=SEL[0]),
                 "conv_res": lambda: ops.conv_c64(dy, v, pk, None, (1, 64), y, v, B, H, W, 64, 64, res=x, select=SEL[0])}
        for name, fn in forms.items():
            times = {0: [], 1: []}
            for r in range(a.rounds):
                for mode in (1, 0):
                    SEL[0] = 0 if mode else 1
                    times[mode].append(timeit(fn, a.iters, warm=1))
            for mode in (1, 0):
                ts = sorted(times[mode])
                print(json.dumps({"kernel": f"{name}_persist{mode}", "batch": B, "median_us": ts[len(ts) // 2] * 1e6,
                                  "min_us": ts[0] * 1e6}), flush=True)
        SEL[0] = 0
        return
    if a.rounds > 0 and (not only or "conv" in only):
        # interleaved A/B (guide rule 24): rounds x variants in one process, median and min per variant
        times = {var: [] for var in variants}
        for _ in range(3):
            ops.conv_c64(x, v, pk, b, (1, 64), y, v, B, H, W, 64, 64)
        for r in range(a.rounds):
            for var in variants:
                times[var].append(timeit(lambda: ops.conv_c64(x, v, pk, b, (1, 64), y, v, B, H, W, 64, 64, select=var), a.iters,
                                         warm=1))
        for var in variants:
            ts = sorted(times[var])
            med, mn = ts[len(ts) // 2], ts[0]
            print(json.dumps({"kernel": f"conv_v{var}", "batch": B, "median_us": med * 1e6, "min_us": mn * 1e6,
                              "median_TFLOP/s": flop / med / 1e12, "best_TFLOP/s": flop / mn / 1e12}), flush=True)
    else:
        for var in variants:
            sel = var if a.precision == "fp32" else 0
            run(f"conv_v{var}", lambda: ops.conv_c64(x, v, pk, b, (1, 64), y, v, B, H, W, 64, 64, select=sel), flop, "TFLOP/s")
            run(f"conv_dgrad2_v{var}", lambda: ops.conv_c64(dy, v, pk, None, (1, 64), y, v, B, H, W, 64, 64, mask=t1,
                                                            in_scale=sc, in_shift=sh, select=sel), flop, "TFLOP/s")
    run("conv_relu_gap", lambda: ops.conv_c64(x, v, pk, b, (1, 64), y, v, B, H, W, 64, 64, relu=True, gap=gap), flop,
        "TFLOP/s")
    run("conv_dgrad2", lambda: ops.conv_c64(dy, v, pk, None, (1, 64), y, v, B, H, W, 64, 64, mask=t1, in_scale=sc,
                                            in_shift=sh), flop, "TFLOP/s")
    run("conv_res", lambda: ops.conv_c64(dy, v, pk, None, (1, 64), y, v, B, H, W, 64, 64, res=x), flop, "TFLOP/s")
    y2 = torch.empty_like(x)
    # the GATE prologue (forward of a gated block: builds and stores the gated skip while staging) and the DOT epilogue
    # (its backward: residual + gate-gradient partial sums), as the fused group node launches them
    run("conv_gate", lambda: ops.conv_c64(dy, v, pk, b, (1, 64), y, v, B, H, W, 64, 64, relu=True, in_scale=sc, gate_add=t1,
                                          gate_out=y2), flop, "TFLOP/s")
    run("conv_dot", lambda: ops.conv_c64(dy, v, pk, None, (1, 64), y, v, B, H, W, 64, 64, res=x, dot=t1, gap=gap), flop,
        "TFLOP/s")
    run("wgrad", lambda: ops.wgrad_c64(x, v, dy, v, dw, db, B, H, W, 64, 64), flop, "TFLOP/s")
    run("wgrad_affine", lambda: ops.wgrad_c64(x, v, dy, v, dw, db, B, H, W, 64, 64, dy_scale=sc, dy_shift=sh), flop,
        "TFLOP/s")
    run("pack", lambda: ops.pack_weight(w, "fwd"), 2 * 36864 * 4, "GB/s")
    L = hip.lib()
    parts = L.sisr_gate_dg_parts(H * W)
    dgp = torch.empty(B, parts, 64, device=dev)
    run("gate_dg_partial", lambda: L.sisr_gate_dg_partial(hip.ptr(dy), hip.ptr(x), hip.ptr(dgp), B, H * W, 64,
                                                          hip.stream()), 2 * fm, "GB/s")
    run("gate_residual", lambda: L.sisr_gate_residual_fwd(hip.ptr(dy), hip.ptr(sc), None, hip.ptr(x), hip.ptr(y), B,
                                                          H * W, 64, hip.stream()), 3 * fm, "GB/s")
    # tail-side kernels on the 4x upsampled map
    Ht, Wt = 4 * H, 4 * W
    Bt = max(1, B // 4)
    xt = torch.randn(Bt, 64, Ht, Wt, generator=g).to(dev).contiguous(memory_format=cl)
    wt = (torch.randn(3, 64, 3, 3, generator=g) * 0.05).to(dev).requires_grad_(True)
    bt = torch.randn(3, generator=g).to(dev).requires_grad_(True)
    xt.requires_grad_(True)
    fmt = Bt * Ht * Wt * 64 * 4
    run("tail_fwd", lambda: ops.conv3x3(xt, wt, bt), fmt, "GB/s")
    if not only or "tail_bwd" in only:
        yt = ops.conv3x3(xt, wt, bt)
        cot = torch.randn_like(yt)
        run("tail_bwd", lambda: torch.autograd.grad(yt, (xt, wt, bt), cot, retain_graph=True), 2 * fmt, "GB/s")
    return res


if __name__ == "__main__":
    main()
