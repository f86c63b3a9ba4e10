#!/usr/bin/env python3
"""Do two independent half-batch conv chains on two streams hide each other's staging / store phases?

    python tools/lane_probe.py [--batch 4] [--links 200] [--lanes 1,2,4] [--forms plain,mask,alt]
The tiles of a minibatch never meet inside a residual group (no batch statistics on the path), so a chain of B-tile convs can
run as L chains of B/L tiles on L streams.  A single chain at 4 tiles per GPU fills the chip with ONE round of workgroups that
stage, multiply and store in lock-step (43 - 51 us per launch against 31 us of MFMA time); with lanes the chains drift out of
phase and one lane's K loop covers the other's staging.  Captured as ONE hipGraph with a fork at the start and a join at the end
(the replayed step's shape); prints microseconds per B-tile link and the fraction of the fp32 MFMA peak.
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
    ap.add_argument("--lanes", default="1,2,4")
    ap.add_argument("--forms", default="plain,mask,alt")
    ap.add_argument("--eager", action="store_true", help="also time the eager (un-captured) launch loop")
    a = ap.parse_args()
    B, H, W = a.batch, 128, 128
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    cl = torch.channels_last
    maps = [(torch.randn(B, 64, H, W, generator=g) * 0.1).to(dev).contiguous(memory_format=cl) for _ in range(a.bufs)]
    t1 = torch.relu(torch.randn(B, 64, H, W, generator=g)).to(dev).contiguous(memory_format=cl)
    ws = [(torch.randn(64, 64, 3, 3, generator=g) * 0.02).to(dev) for _ in range(4)]
    b = torch.zeros(64).to(dev)
    sc, sh = torch.rand(B, 64, generator=g).to(dev), torch.rand(B, 64, generator=g).to(dev)
    v = hip.view_plain(H, W, 64)
    pks = [ops.pack_weight(w, "fwd") for w in ws]
    flop = 2.0 * B * H * W * 64 * 64 * 9

    def link(i, form, b0, b1):
        x, y = maps[i % a.bufs][b0:b1], maps[(i + 1) % a.bufs][b0:b1]
        pk = pks[i % len(pks)]
        if form == "plain" or (form == "alt" and i % 2 == 0):
            ops.conv_c64(x, v, pk, b, (1, 64), y, v, b1 - b0, H, W, 64, 64)
        else:
            ops.conv_c64(x, v, pk, None, (1, 64), y, v, b1 - b0, H, W, 64, 64, mask=t1[b0:b1], in_scale=sc[b0:b1],
                         in_shift=sh[b0:b1])

    def run(form, lanes, streams):
        main_s = torch.cuda.current_stream()
        cuts = [B * k // lanes for k in range(lanes + 1)]
        if lanes == 1:
            for i in range(a.links):
                link(i, form, 0, B)
            return
        for s in streams[:lanes]:
            s.wait_stream(main_s)
        for i in range(a.links):
            for k in range(lanes):
                with torch.cuda.stream(streams[k]):
                    link(i, form, cuts[k], cuts[k + 1])
        for s in streams[:lanes]:
            main_s.wait_stream(s)

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

    streams = [torch.cuda.Stream() for _ in range(8)]
    for form in a.forms.split(","):
        for lanes in [int(t) for t in a.lanes.split(",")]:
            if B % lanes:
                continue
            s = torch.cuda.Stream()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.stream(s):
                run(form, lanes, streams)
                torch.cuda.synchronize()
                with torch.cuda.graph(gr, stream=s):
                    run(form, lanes, streams)
            torch.cuda.synchronize()
            us_graph = timed(gr.replay)
            rec = {"batch": B, "form": form, "lanes": lanes, "graph_us_per_link": round(us_graph, 2),
                   "graph_frac_of_peak": round(flop / us_graph / 1e6 / 157.3, 3)}
            if a.eager:
                rec["eager_us_per_link"] = round(timed(lambda: run(form, lanes, streams)), 2)
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
