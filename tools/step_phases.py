#!/usr/bin/env python3
"""Wall-clock phases of one training step (forward / loss / backward / optimiser), synchronised between phases."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sisr_amd  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "edsr"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
params = {"qedsr": {"metadata": ["blur_kernel"]}, "qrcan": {"metadata": ["blur_kernel"], "style": "standard",
                                                           "include_q_layer": True}}.get(name, {})
torch.manual_seed(8)
h = sisr_amd.available_models[name](device=0, model_save_dir="/tmp", eval_mode=False, scale=4, lr=1e-4, **params)
x = torch.rand(B, 3, 128, 128, device="cuda")
y = torch.rand(B, 3, 512, 512, device="cuda")
md = (torch.rand(B, 10, 1, 1, device="cuda") * 0.4) if params else None


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


for it in range(4):
    h.net.train()
    t0 = sync()
    out = h.net(x, md) if md is not None else h.net(x)
    t1 = sync()
    loss = h.criterion(out, y)
    t2 = sync()
    h.optimizer.zero_grad()
    loss.backward()
    t3 = sync()
    h.optimizer.step()
    t4 = sync()
    print(f"{name} B={B} it={it}: fwd {1e3 * (t1 - t0):.1f} ms, loss {1e3 * (t2 - t1):.1f}, bwd {1e3 * (t3 - t2):.1f}, "
          f"opt {1e3 * (t4 - t3):.1f}, alloc {torch.cuda.memory_allocated() / 2**30:.1f} GiB, "
          f"reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB", flush=True)
