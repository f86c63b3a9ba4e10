#!/usr/bin/env python3
"""Rank program of the data-parallel equivalence tests (not a test module itself).

    python tests/_dp_worker.py OUT.pt MODEL MODE STEPS ALT        (1 process: the reference gradient)
    python -m torch.distributed.run --nproc-per-node 2 ... tests/_dp_worker.py OUT.pt MODEL MODE STEPS ALT

Every rank sits on cuda:0 (transport SISR_DIST_BACKEND, gloo on a 1-GPU box).  A seeded GLOBAL batch per step is cut
into contiguous rank slices (parallel.shard_batch); `BaseModel.train_step` runs with lr = 0, so parameters stay
put and step k's gradient depends on step k's batch only -- a stale gradient (replayed graph writing elsewhere, bucket
not refreshed) shows up at k >= 1.  Rank 0 stores {step: {name: grad}} + losses (+ the reducer's bucket count and, for graph
mode, how many buckets each captured graph signals; SISR_DP_BUCKET_MB in the environment sets the bucket size).  MODE: eager | graph | eager_noside
(eager with SISR_WGRAD_SIDE_STREAM=0); ALT = 1 alternates two batch shapes from step to step.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out_path, model, mode, steps, alt = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
if mode == "eager_noside":
    os.environ["SISR_WGRAD_SIDE_STREAM"] = "0"

import torch  # noqa: E402

import sisr_amd  # noqa: E402

rank, world, _ = sisr_amd.parallel.init_distributed()
torch.cuda.set_device(0)
CFG = {
    "qrcan": dict(metadata=["blur_kernel"], style="standard", include_q_layer=True, n_resgroups=2, n_resblocks=2),
    "rcan": dict(),
    "edsr": dict(num_blocks=2),
}
torch.manual_seed(8)
if model == "rcan":  # the handler builds the 10 x 20 net; a reduced one goes in by hand
    h = sisr_amd.available_models["rcan"].__new__(sisr_amd.available_models["rcan"])
    sisr_amd.handlers.BaseModel.__init__(h, device=0, model_save_dir="/tmp", eval_mode=False)
    h.net = sisr_amd.architectures.RCAN(scale=4, n_resgroups=2, n_resblocks=2)
    h.activate_device()
    h.training_setup(0.0, None, None, None, 0)
    h.model_name = "rcan"
else:
    h = sisr_amd.available_models[model](device=0, model_save_dir="/tmp", eval_mode=False, scale=4, lr=0.0, **CFG[model])
if world > 1:
    h.set_multi_gpu()
h.use_graph = mode.startswith("graph")

g = torch.Generator().manual_seed(21)
record = {"loss": [], "grads": []}
for step in range(steps):
    GB = 4
    hw = 24 if (alt and step % 2) else 16
    batch = {"lr": torch.rand(GB, 3, hw, hw + 8, generator=g), "hr": torch.rand(GB, 3, 4 * hw, 4 * hw + 32, generator=g)}
    if model.startswith("q"):
        batch["metadata"] = torch.rand(GB, 10, generator=g, dtype=torch.float64) * 0.4
        batch["metadata_keys"] = [("blur_kernel",) * GB] * 10
    if world > 1:
        batch = sisr_amd.parallel.shard_batch(batch, rank, world)
    kw = {k: v for k, v in batch.items() if k.startswith("metadata")}
    loss, _ = h.train_step(batch["lr"], batch["hr"], **kw)
    if world > 1:
        import torch.distributed as dist
        lt = loss.detach().clone().cpu()
        dist.all_reduce(lt)  # mean of the equal shards' mean losses == global mean
        loss = lt / world
    record["loss"].append(float(loss))
    record["grads"].append({n: p.grad.detach().cpu().clone() for n, p in h.net.named_parameters() if p.grad is not None})
torch.cuda.synchronize()
if h.reducer is not None:
    record["buckets"] = len(h.reducer.buckets)
    record["signalled"] = [len(e[7]) for e in h._graphs.values()]  # buckets with a signal node, per captured graph
if rank == 0:
    torch.save(record, out_path)
if world > 1:
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()
