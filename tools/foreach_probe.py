#!/usr/bin/env python3
"""Diagnostic: what do FlatAdam.step / GradReducer copy per step, eager vs hipGraph replay?"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sisr_amd  # noqa: E402

B = 4
torch.manual_seed(8)
h = sisr_amd.available_models["qrcan"](device=0, model_save_dir="/tmp", eval_mode=False, scale=4, lr=1e-4,
                                       metadata=["blur_kernel"], style="standard", include_q_layer=True)
x, y = torch.rand(B, 3, 128, 128).cuda(), torch.rand(B, 3, 512, 512).cuda()
extra = {"extra_channels": (torch.rand(B, 10, 1, 1) * 0.4).cuda()}
orig = torch._foreach_copy_
log = []


def spy(dst, src, *a, **k):
    log.append((len(dst), sum(t.numel() for t in dst), collections.Counter(tuple(t.shape) for t in dst).most_common(4)))
    return orig(dst, src, *a, **k)


torch._foreach_copy_ = spy
names = {p: n for n, p in h.net.named_parameters()}
for mode in (False, True):
    h.use_graph = mode
    for step in range(3):
        log.clear()
        h.train_step(x, y, **extra)
        torch.cuda.synchronize()
        if step == 2:
            print("graph" if mode else "eager", log)
            bad = [names[p] for p in h.net.parameters() if p.grad is not None and p.grad.data_ptr() != h.optimizer.grad_views[p].data_ptr()]
            print("  grads outside the arena:", len(bad), bad[:6])
