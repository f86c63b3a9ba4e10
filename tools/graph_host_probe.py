#!/usr/bin/env python3
"""Where the host time of a hipGraph-replayed training step goes (is graph.replay() asynchronous? what do the eager
pieces around it cost?).  python tools/graph_host_probe.py [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sisr_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(8)
h = sisr_amd.available_models["qrcan"](device=0, model_save_dir="/tmp", eval_mode=False, scale=4, lr=1e-4,
                                       metadata=["blur_kernel"], style="standard", include_q_layer=True)
h.use_graph = True
x, y = torch.rand(B, 3, 128, 128).cuda(), torch.rand(B, 3, 512, 512).cuda()
md = (torch.rand(B, 10, 1, 1) * 0.4).cuda()
for _ in range(3):
    h.train_step(x, y, extra_channels=md)
torch.cuda.synchronize()
entry = next(iter(h._graphs.values()))
graph = entry[0]
for it in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    graph.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for p, g in entry[6]:
        p.grad = g
    t3 = time.perf_counter()
    h.optimizer.step()
    t4 = time.perf_counter()
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    print(f"replay call {1e3 * (t1 - t0):.2f} ms (returns), graph done after {1e3 * (t2 - t0):.2f} ms; rebind "
          f"{1e3 * (t3 - t2):.2f} ms; optimizer.step host {1e3 * (t4 - t3):.2f} ms, done after {1e3 * (t5 - t3):.2f} ms", flush=True)
t0 = time.perf_counter()
for _ in range(5):
    h.train_step(x, y, extra_channels=md)
torch.cuda.synchronize()
print(f"train_step loop: {1e3 * (time.perf_counter() - t0) / 5:.2f} ms / step")
