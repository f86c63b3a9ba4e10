#!/usr/bin/env python3
"""Would running a conv's input-gradient and weight-gradient kernels CONCURRENTLY pay at small batch?  Times N pairs
(dgrad-with-mask conv + wgrad) back to back on one stream against the same pairs with the wgrad on a second stream.
python tools/pair_probe.py [batch] [pairs]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sisr_amd  # noqa: E402

ops, hip = sisr_amd.ops, sisr_amd.hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
H = W = 128
dev = torch.device("cuda:0")
cl = torch.channels_last
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 64, H, W, generator=g).to(dev).contiguous(memory_format=cl)
dy = torch.randn(B, 64, H, W, generator=g).to(dev).contiguous(memory_format=cl)
t1 = torch.relu(torch.randn(B, 64, H, W, generator=g)).to(dev).contiguous(memory_format=cl)
w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev)
pk = ops.pack_weight(w, "dgrad")
v = hip.view_plain(H, W, 64)
ys = [torch.empty_like(x) for _ in range(2)]
dws = [torch.empty_like(w) for _ in range(2)]
dbs = [torch.empty(64, device=dev) for _ in range(2)]
side = torch.cuda.Stream()


def conv(i):
    ops.conv_c64(dy, v, pk, None, (1, 64), ys[i & 1], v, B, H, W, 64, 64, mask=t1)


def wgrad(i):
    ops.wgrad_c64(t1, v, dy, v, dws[i & 1], dbs[i & 1], B, H, W, 64, 64)


def run(concurrent):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(N):
        if concurrent:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                wgrad(i)
            conv(i)
            torch.cuda.current_stream().wait_stream(side)  # the next pair depends on both (as a real backward chain does)
        else:
            wgrad(i)
            conv(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1e3


for _ in range(2):
    run(False), run(True)
# captured as graphs too: eager launches at this size are host-bound
res = {"batch": B, "pairs": N, "eager_serial_us": run(False), "eager_concurrent_us": run(True)}
for name, conc in (("graph_serial_us", False), ("graph_concurrent_us", True)):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(20):
            if conc:
                ev = torch.cuda.Event()
                ev.record()
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    wgrad(i)
                conv(i)
                torch.cuda.current_stream().wait_stream(side)
            else:
                wgrad(i)
                conv(i)
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    res[name] = e0.elapsed_time(e1) / 200 * 1e3
print(json.dumps(res))
