#!/usr/bin/env python3
"""Diagnostic: host time of the pieces of a graph-replayed data-parallel step (one-rank RCCL world, QRCAN 4 tiles)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import sisr_amd  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29547")
dev = torch.device("cuda:0")
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
torch.manual_seed(8)
h = sisr_amd.available_models["qrcan"](device=0, model_save_dir="/tmp", eval_mode=False, scale=4, lr=1e-4,
                                       metadata=["blur_kernel"], style="standard", include_q_layer=True)
h.set_multi_gpu()
h.use_graph = True
B = 4
x, y = torch.rand(B, 3, 128, 128).cuda(), torch.rand(B, 3, 512, 512).cuda()
extra = {"extra_channels": (torch.rand(B, 10, 1, 1) * 0.4).cuda()}
red = h.reducer
T = {}


def timed(obj, name):
    fn = getattr(obj, name)

    def wrap(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        T.setdefault(name, []).append(time.perf_counter() - t0)
        return r
    setattr(obj, name, wrap)


timed(red, "launch_signalled")
timed(red, "reduce")
timed(red, "_launch")
timed(h.optimizer, "step")
for _ in range(3):
    h.train_step(x, y, **extra)
torch.cuda.synchronize()
T.clear()
t0 = time.perf_counter()
for _ in range(5):
    ts = time.perf_counter()
    h.train_step(x, y, **extra)
    T.setdefault("train_step (host)", []).append(time.perf_counter() - ts)
torch.cuda.synchronize()
print("ms per step", (time.perf_counter() - t0) / 5 * 1e3, "buckets", len(red.buckets))
for k, v in T.items():
    print(f"{k:22s} calls/step {len(v) / 5:5.1f}  total ms/step {sum(v) / 5 * 1e3:8.3f}")
dist.destroy_process_group()
