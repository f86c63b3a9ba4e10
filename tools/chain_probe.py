#!/usr/bin/env python3
"""What a 64 -> 64 conv costs as one link of a dependent chain (the real step's shape), eager and as hipGraph nodes.

    python tools/chain_probe.py [--batch 4] [--links 200] [--bufs 8]
The kernel loop of tools/kbench.py re-launches one conv on the same two buffers.  A training step is a chain: conv k reads
what conv k-1 wrote, into a buffer nobody touched for a while.  This probe times `links` plain convs y[i+1] = conv(y[i])
cycling over `bufs` maps -- (a) eager on one stream, (b) captured once and replayed -- and prints microseconds per link
beside the same-buffers loop.  Forms: plain, mask (dgrad with ReLU mask + affine), gate-prologue is left to kbench.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sisr_amd  # noqa: E402

ops, hip = sisr_amd.ops, sisr_amd.hip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--links", type=int, default=200)
    ap.add_argument("--bufs", type=int, default=8)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--weights", type=int, default=1, help="cycle over this many different packed weights (a step never reuses one)")
    ap.add_argument("--forms", default="plain,mask", help="plain, mask, alt (the two alternating link by link)")
    a = ap.parse_args()
    B, H, W = a.batch, 128, 128
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    cl = torch.channels_last
    maps = [(torch.randn(B, 64, H, W, generator=g) * 0.1).to(dev).contiguous(memory_format=cl) for _ in range(a.bufs)]
    t1 = torch.relu(torch.randn(B, 64, H, W, generator=g)).to(dev).contiguous(memory_format=cl)
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.02).to(dev)
    b = torch.zeros(64).to(dev)
    sc, sh = torch.rand(B, 64, generator=g).to(dev), torch.rand(B, 64, generator=g).to(dev)
    v = hip.view_plain(H, W, 64)
    pks = [ops.pack_weight(w + 0.001 * k, "fwd") for k in range(a.weights)]
    flop = 2.0 * B * H * W * 64 * 64 * 9

    def link(i, form, same):
        x, y = (maps[0], maps[1]) if same else (maps[i % a.bufs], maps[(i + 1) % a.bufs])
        pk = pks[i % a.weights]
        if form == "plain" or (form == "alt" and i % 2 == 0):
            ops.conv_c64(x, v, pk, b, (1, 64), y, v, B, H, W, 64, 64)
        else:
            ops.conv_c64(x, v, pk, None, (1, 64), y, v, B, H, W, 64, 64, mask=t1, in_scale=sc, in_shift=sh)

    def chain(form, same):
        for i in range(a.links):
            link(i, form, same)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / a.reps / a.links

    for form in a.forms.split(","):
        for same in (False,):
            us_eager = timed(lambda: chain(form, same))
            s = torch.cuda.Stream()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.stream(s):
                chain(form, same)
                torch.cuda.synchronize()
                with torch.cuda.graph(gr, stream=s):
                    chain(form, same)
            torch.cuda.synchronize()
            us_graph = timed(gr.replay)
            print(json.dumps({"batch": B, "form": form, "buffers": "same two" if same else "chain over %d" % a.bufs, "weights": a.weights,
                              "eager_us_per_link": round(us_eager, 2), "graph_us_per_link": round(us_graph, 2),
                              "graph_frac_of_peak": round(flop / us_graph / 1e6 / 157.3, 3)}), flush=True)


if __name__ == "__main__":
    main()
