#!/usr/bin/env python3
"""Cost of the data-parallel machinery itself: RCAN train steps on one GPU with and without a single-rank RCCL
world (GradReducer hooks, bucket all-reduce, join).  python tools/dp_overhead.py [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import sisr_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29791")
dist.init_process_group(backend="nccl", rank=0, world_size=1)
for use_dp in (False, True):
    torch.manual_seed(8)
    h = sisr_amd.available_models["rcan"](device=0, model_save_dir="/tmp", eval_mode=False, scale=4, lr=1e-4)
    if use_dp:
        h.set_multi_gpu()
    x, y = torch.rand(B, 3, 128, 128).cuda(), torch.rand(B, 3, 512, 512).cuda()
    for _ in range(2):
        h.train_step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        loss, _ = h.train_step(x, y)
    torch.cuda.synchronize()
    print("with reducer" if use_dp else "plain       ", f"{(time.perf_counter() - t0) / 5 * 1e3:.1f} ms/step  loss {float(loss):.6f}")
    if use_dp:
        h.reducer.remove()
dist.destroy_process_group()
