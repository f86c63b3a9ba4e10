#!/usr/bin/env python3
"""Diagnostic: where do the device-to-device copies of a train step come from?  One eager QRCAN step under torch.profiler
with Python stacks; aten::copy_ / aten::clone / aten::contiguous calls grouped by the innermost frame inside this repo."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import sisr_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
name = sys.argv[2] if len(sys.argv) > 2 else "qrcan"
kw = {"metadata": ["blur_kernel"], "style": "standard", "include_q_layer": True} if name.startswith("q") else {}
torch.manual_seed(8)
h = sisr_amd.available_models[name](device=0, model_save_dir="/tmp", eval_mode=False, scale=4, lr=1e-4, **kw)
x, y = torch.rand(B, 3, 128, 128).cuda(), torch.rand(B, 3, 512, 512).cuda()
extra = {"extra_channels": (torch.rand(B, 10, 1, 1) * 0.4).cuda()} if name.startswith("q") else {}
if len(sys.argv) > 3 and sys.argv[3] == "dp":  # one-rank RCCL world, as bench.py --force-dp
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    h.set_multi_gpu()
graph = len(sys.argv) > 4 and sys.argv[4] == "graph"  # profile the CAPTURE pass: its aten ops are the replay's nodes
h.use_graph = graph
if not graph:
    for _ in range(2):
        h.train_step(x, y, **extra)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    h.train_step(x, y, **extra)
    torch.cuda.synchronize()
by = collections.Counter()
kern = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        kern[ev.name[:70]] += 1
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::add_",
                   "aten::add", "aten::mul", "aten::stack", "aten::cat"):
        chain, q = [], ev.cpu_parent
        while q is not None and len(chain) < 4:
            chain.append(q.name[:60])
            q = q.cpu_parent
        by[(ev.name, tuple(str(s) for s in ev.input_shapes)[:2], " <- ".join(chain))] += 1
for (n, shp, s), c in by.most_common(40):
    print(c, n, shp, s)
print("---- device activities")
for n, c in kern.most_common(60):
    print(c, n)
