#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase durations (cycles) of the bf16x3 conv kernel (stamped instantiation, SISR_X3_STAMP=1)
and the resident-workgroup count the runtime reports.  Needs the diagnostic library (`bash csrc/build.sh diag`):
    SISR_HIP_LIB=super-resolution-meta-attention-networks_amd/libsisr_hip_diag.so python tools/x3_phases.py [batch]"""
import json
import os
import sys

os.environ["SISR_X3_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sisr_amd  # noqa: E402

ops, hip = sisr_amd.ops, sisr_amd.hip
ops.set_precision("bf16x3")
B, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 128, 128
dev = torch.device("cuda:0")
print(json.dumps({"resident_workgroups_per_cu": {"bf16x3 conv": hip.lib().sisr_diag_conv_occupancy(0),
                                                 "fp32 v4 conv": hip.lib().sisr_diag_conv_occupancy(1)}}))
x = torch.randn(B, 64, H, W, device=dev).contiguous(memory_format=torch.channels_last)
y = torch.empty_like(x)
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
b = torch.randn(64, device=dev)
pk = ops.pack_weight(w, "fwd")
v = hip.view_plain(H, W, 64)
nblk = B * 32 * 4
dbg = torch.zeros(nblk * 16, dtype=torch.int32, device=dev)
gap = torch.empty(B, ops.gap_parts(H, W), 64, device=dev)
for _ in range(3):
    ops.conv_c64(x, v, pk, b, (1, 64), y, v, B, H, W, 64, 64, gap=gap, dot=dbg.view(torch.float32))
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
                  "total_per_wg_median": float(np.median(d[:, :, :3].sum(-1))),
                  "note": "s_memtime ticks at 100 MHz on gfx9 (not the shader clock): multiply by ~24 for core cycles"}))
