#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase durations of the general conv3x3_c64 kernel (stamped build, select 16).
Needs the diagnostic library: `bash csrc/build.sh diag` and SISR_HIP_LIB=.../libsisr_hip_diag.so."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sisr_amd  # noqa: E402

ops, hip = sisr_amd.ops, sisr_amd.hip
B, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 128, 128
dev = torch.device("cuda:0")
x = torch.randn(B, 64, H, W, device=dev).contiguous(memory_format=torch.channels_last)
y = torch.empty_like(x)
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
b = torch.randn(64, device=dev)
pk = ops.pack_weight(w, "fwd")
v = hip.view_plain(H, W, 64)
nblk = B * 32 * 4
dbg = torch.zeros(nblk * 16, dtype=torch.int32, device=dev)
sel = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for _ in range(3):
    ops.conv_c64(x, v, pk, b, (1, 64), y, v, B, H, W, 64, 64, gap=dbg.view(torch.float32), select=sel)
torch.cuda.synchronize()
d = dbg.cpu().numpy().astype(np.int64).reshape(nblk, 4, 4) & 0xffffffff
for name, k in (("staging", 0), ("kloop", 1), ("epilogue", 2)):
    a = d[:, :, k].reshape(-1)
    print(json.dumps({"phase": name, "median": float(np.median(a)), "p10": float(np.percentile(a, 10)),
                      "p90": float(np.percentile(a, 90)), "mean": float(a.mean())}))
start = d[:, 0, 3]
order = np.argsort(start)
rel = (start[order] - start[order][0]) & 0xffffffff
print(json.dumps({"start_spread_cycles": [int(rel[int(q * (len(rel) - 1))]) for q in (0.1, 0.25, 0.5, 0.75, 0.9, 1.0)],
                  "total_per_wg_median": float(np.median(d[:, :, :3].sum(-1)))}))
