"""Data-parallel path on CPU: world_size 2, gloo backend.

GradReducer must (a) broadcast rank 0's replica, (b) average gradients bucket by bucket -- both through the
backward hooks (overlap) and through the join alone -- so that 2 ranks on half-batches reproduce the
single-process gradient of the concatenated batch (L1 'mean' over equal shards => mean of means is exact).
"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class TinyNet(torch.nn.Module):
    """Stock-torch stand-in with several parameter sizes so more than one bucket is formed."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Conv2d(3, 16, 3, padding=1)
        self.b = torch.nn.Conv2d(16, 16, 3, padding=1)
        self.c = torch.nn.Conv2d(16, 3, 3, padding=1)
        self.unused = torch.nn.Parameter(torch.zeros(5))  # never receives a gradient

    def forward(self, x):
        return self.c(F.relu(self.b(F.relu(self.a(x)))))


def _worker(rank, world, port, overlap, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sisr_amd
    r, w, _ = sisr_amd.parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)  # different initial weights per rank: the reducer must broadcast rank 0's
    net = TinyNet()
    red = sisr_amd.parallel.GradReducer(net, bucket_mb=0.004, overlap=overlap)
    assert len(red.buckets) > 1
    g = torch.Generator().manual_seed(5)
    x = torch.rand(4, 3, 8, 8, generator=g)
    y = torch.rand(4, 3, 8, 8, generator=g)
    batch = {"lr": x, "hr": y, "tag": ["a", "b", "c", "d"], "metadata": torch.arange(8.).reshape(4, 2),
             "metadata_keys": [("k0",) * 4, ("k1",) * 4]}
    shard = sisr_amd.parallel.shard_batch(batch, rank, world)
    assert shard["lr"].shape[0] == 2 and shard["tag"] == ["a", "b", "c", "d"][2 * rank:2 * rank + 2]
    assert shard["metadata_keys"] == [("k0", "k0"), ("k1", "k1")]
    for _ in range(2):  # two steps: bucket bookkeeping must reset
        net.zero_grad()
        F.l1_loss(net(shard["lr"]), shard["hr"]).backward()
        red.reduce()
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in net.parameters()])
    w0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    out_q.put((rank, flat, w0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_two_rank_gradients_match_single_process(overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 200) + (1 if overlap else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, flat, w0 = q.get(timeout=120)
        res[rank] = (flat, w0)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(res[0][1], res[1][1]), "replicas were not synchronised to rank 0"
    assert torch.allclose(res[0][0], res[1][0], atol=0, rtol=0), "ranks ended with different gradients"
    # single-process reference on the full batch with rank 0's weights
    torch.manual_seed(100)
    net = TinyNet()
    g = torch.Generator().manual_seed(5)
    x = torch.rand(4, 3, 8, 8, generator=g)
    y = torch.rand(4, 3, 8, 8, generator=g)
    F.l1_loss(net(x), y).backward()
    ref = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in net.parameters()])
    assert torch.allclose(res[0][0], ref, rtol=1e-5, atol=1e-7)


def _ragged_worker(rank, world, port, n, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sisr_amd
    sisr_amd.parallel.init_distributed(backend="gloo")
    torch.manual_seed(100)
    net = TinyNet()
    red = sisr_amd.parallel.GradReducer(net, bucket_mb=0.004)
    g = torch.Generator().manual_seed(6)
    x, y = torch.rand(n, 3, 8, 8, generator=g), torch.rand(n, 3, 8, 8, generator=g)
    shard = sisr_amd.parallel.shard_batch({"lr": x, "hr": y, "tag": [str(i) for i in range(n)]}, rank, world)
    net.zero_grad()
    if shard["lr"].shape[0]:
        (F.l1_loss(net(shard["lr"]), shard["hr"]) * shard.get("loss_scale", 1.0)).backward()
    red.reduce()
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in net.parameters()])
    out_q.put((rank, flat, shard["tag"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [3, 1])
def test_ragged_batch_is_cut_unevenly_and_weighted(n):
    """The reference keeps an epoch's ragged last batch (drop_last False) and its DataParallel scatters it unevenly.
    shard_batch gives the first n % world ranks one sample more and a 'loss_scale' = n_r * world / n; with it the averaged
    gradient is the single-process gradient of the n samples -- also when a rank gets no sample at all (n = 1)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29820 + (os.getpid() % 150) + n
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, flat, tags = q.get(timeout=120)
        res[rank] = (flat, tags)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] + res[1][1] == [str(i) for i in range(n)] and len(res[0][1]) == (n + 1) // 2
    torch.manual_seed(100)
    net = TinyNet()
    g = torch.Generator().manual_seed(6)
    x, y = torch.rand(n, 3, 8, 8, generator=g), torch.rand(n, 3, 8, 8, generator=g)
    F.l1_loss(net(x), y).backward()
    ref = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in net.parameters()])
    assert torch.allclose(res[0][0], ref, rtol=1e-5, atol=1e-7) and torch.equal(res[0][0], res[1][0])


def test_reducer_needs_process_group():
    sys.path.insert(0, ROOT)
    import sisr_amd
    with pytest.raises(RuntimeError, match="one process per GPU"):
        sisr_amd.parallel.GradReducer(TinyNet())
