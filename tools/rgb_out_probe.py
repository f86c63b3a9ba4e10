"""The 64 -> 3 tail conv alone (B x 512 x 512 x 64 -> B x 3 x 512 x 512), for rocprofv3 --pmc passes: python tools/rgb_out_probe.py [B]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import sisr_amd  # noqa: E402
from sisr_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.randn(B, 64, 512, 512, device="cuda").contiguous(memory_format=torch.channels_last)
w = torch.randn(3, 64, 3, 3, device="cuda") * 0.05
b = torch.zeros(3, device="cuda")
for _ in range(3):
    ops.conv3x3(x, w, b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.conv3x3(x, w, b)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
print(f"tail conv B={B}: {us:.1f} us, {B * 512 * 512 * 256 / us / 1e6:.2f} TB/s of input")
