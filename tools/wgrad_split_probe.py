"""One B = 32 weight gradient as one launch vs as eight 4-sample jobs of a batched launch (+ the sum of the eight results)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import sisr_amd  # noqa: E402
from sisr_amd import hip, ops  # noqa: E402

B, H, W = 32, 128, 128
dev = torch.device("cuda:0")
cl = torch.channels_last
g = torch.Generator().manual_seed(1)
xs = [torch.randn(B, 64, H, W, generator=g).to(dev).contiguous(memory_format=cl) for _ in range(3)]
dys = [torch.randn(B, 64, H, W, generator=g).to(dev).contiguous(memory_format=cl) for _ in range(3)]
v = hip.view_plain(H, W, 64)
dw, db = torch.empty(64, 64, 3, 3, device=dev), torch.empty(64, device=dev)


def single(i):
    ops.wgrad_c64(xs[i % 3], v, dys[i % 3], v, dw, db, B, H, W, 64, 64)


parts = [(torch.empty(64, 64, 3, 3, device=dev), torch.empty(64, device=dev)) for _ in range(8)]
acc = torch.empty(8, 64 * 64 * 9 + 64, device=dev)


def split(i):
    q = ops.WgradQueue(4, H, W, dev)
    for k in range(8):
        q.add(xs[i % 3][4 * k:4 * k + 4], dys[i % 3][4 * k:4 * k + 4], parts[k][0], parts[k][1])
    q.flush()


def timeit(fn, n=20):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rnd in range(3):
    print("single %.1f us   eight 4-sample jobs %.1f us" % (timeit(single), timeit(split)))
single(0)
split(0)
tot = sum(p[0] for p in parts)
print("max rel diff of the sums", float((tot - dw).abs().max() / dw.abs().max()))
