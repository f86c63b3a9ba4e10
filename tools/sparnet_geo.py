#!/usr/bin/env python3
"""Where a SPARNet training step spends its launches, per conv geometry.

    python tools/sparnet_geo.py [BATCH] [qsparnet]

Runs one eager training step of the default SPARNet with every MFMA conv, weight gradient, gather and batch-norm call of
sisr_amd.ops timed alone on the stream (HIP events around the call, a synchronise per call), and prints one JSON line per
(kind, geometry): calls per step, mean microseconds, total milliseconds, GFLOP of the call and the fraction of the fp32
matrix peak it runs at.  Diagnostic; nothing here is on the product path.
"""
import collections
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sisr_amd  # noqa: E402
from sisr_amd import ops  # noqa: E402

PEAK = 157.3e12


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    name = sys.argv[2] if len(sys.argv) > 2 else "sparnet"
    dev = torch.device("cuda:0")
    torch.manual_seed(8)
    params = {"metadata": ["blur_kernel"]} if name == "qsparnet" else {}
    h = sisr_amd.available_models[name](device=0, model_save_dir="/tmp", eval_mode=False, scale=4, lr=1e-4, **params)
    h.use_graph = False
    g = torch.Generator().manual_seed(8)
    x = torch.rand(B, 3, 128, 128, generator=g).to(dev)
    y = torch.rand(B, 3, 128, 128, generator=g).to(dev)
    kw = {}
    if name == "qsparnet":
        kw["extra_channels"] = (torch.rand(B, 10, 1, 1, generator=g) * 0.4).to(dev)
    for _ in range(2):
        h.train_step(x, y, **kw)
    torch.cuda.synchronize()

    acc = collections.defaultdict(list)
    flops = {}

    def timed(kind, key, fl, fn, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        r = fn(*a, **k)
        e1.record()
        torch.cuda.synchronize()
        acc[(kind, key)].append(e0.elapsed_time(e1) * 1e3)
        flops[(kind, key)] = fl
        return r

    conv0, wgrad0 = ops.conv_c64, ops.wgrad_c64

    def conv(x_, xv, pk, bias, bnq, y_, yv, B_, H, W, cin, cout, **k):
        kind = "dgrad" if ops.IN_BACKWARD else "conv"
        return timed(kind, (B_, H, W, cin, cout), 2.0 * B_ * H * W * cin * cout * 9, conv0, x_, xv, pk, bias, bnq, y_, yv, B_, H, W,
                     cin, cout, **k)

    def wgrad(x_, xv, dy, dyv, dw, db, B_, H, W, cin, cout, **k):
        return timed("wgrad", (B_, H, W, cin, cout), 2.0 * B_ * H * W * cin * cout * 9, wgrad0, x_, xv, dy, dyv, dw, db, B_, H, W,
                     cin, cout, **k)

    ops.conv_c64, ops.wgrad_c64 = conv, wgrad
    L = sisr_amd.hip.lib()
    wrapped = {}
    for sym, kind in (("sisr_pad_reflect_up", "pad"), ("sisr_crop_stride", "crop"), ("sisr_bn_act_fwd", "bn_fwd"),
                      ("sisr_bn_act_bwd", "bn_bwd"), ("sisr_spar_combine_fwd", "combine"), ("sisr_spar_combine_bwd", "combine_bwd"),
                      ("sisr_conv3x3_c64_geo", "conv_geo"), ("sisr_wgrad3x3_c64_geo", "wgrad_geo")):
        if not hasattr(L, sym):
            continue
        fn0 = getattr(L, sym)
        wrapped[sym] = fn0

        def make(fn0=fn0, kind=kind, sym=sym):
            def call(*a):
                if kind == "conv_geo":  # B, H, W, cin, cout, mode, up, kreal
                    key = tuple(a[7:15])
                    return timed(kind, key, 2.0 * a[7] * a[8] * a[9] * a[10] * a[11] * 9, fn0, *a)
                if kind == "wgrad_geo":  # co, ci, B, H, W, cin, cout, up, active units
                    key = (a[5], a[6]) + tuple(a[10:17])
                    return timed(kind, key, 2.0 * a[10] * a[11] * a[12] * a[13] * a[14] * 9, fn0, *a)
                ints = tuple(v for v in a[2:8] if isinstance(v, int) and 0 <= v < 100000)
                return timed(kind, ints[:6], 0.0, fn0, *a)
            return call
        setattr(L, sym, make())
    side, ops.WGRAD_SIDE_STREAM = ops.WGRAD_SIDE_STREAM, False
    try:
        h.train_step(x, y, **kw)
    finally:
        ops.conv_c64, ops.wgrad_c64, ops.WGRAD_SIDE_STREAM = conv0, wgrad0, side
        for sym, fn0 in wrapped.items():
            setattr(L, sym, fn0)
    rows = []
    for (kind, key), us in acc.items():
        mean = sum(us) / len(us)
        fl = flops[(kind, key)]
        rows.append({"kind": kind, "geo": list(key), "calls": len(us), "us": round(mean, 1), "ms": round(sum(us) / 1e3, 3),
                     "gflop": round(fl / 1e9, 3), "frac": round(fl / (mean * 1e-6) / PEAK, 3) if fl else None})
    rows.sort(key=lambda r: -r["ms"])
    tot = collections.defaultdict(float)
    for r in rows:
        tot[r["kind"]] += r["ms"]
        print(json.dumps(r))
    print(json.dumps({"totals_ms": {k: round(v, 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}}))


if __name__ == "__main__":
    main()
