#!/usr/bin/env python3
"""Diagnostic: where a wave of the bf16-map persistent conv (conv3x3_c64_bf16_persist_kernel, bf16 in / out) spends its cycles.

Needs the diagnostic library (`bash csrc/build.sh diag`; SISR_HIP_LIB=.../libsisr_hip_diag.so SISR_BF16S_STAMP=1): every wave
sums the shader cycles of its K loops, epilogues (transposes through LDS, stores, barriers) and halo commits (wait for the
prefetched loads, LDS writes, barrier) over its tiles.

    python tools/bf16s_timeline.py [BATCH=32]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SISR_BF16S_STAMP"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sisr_amd  # noqa: E402

ops, hip = sisr_amd.ops, sisr_amd.hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H = W = 128
dev = torch.device("cuda:0")
ops.set_precision("bf16")
cl = torch.channels_last
x = torch.randn(B, 64, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=cl)
y = torch.empty_like(x)
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
b = torch.randn(64, device=dev)
pk = ops.pack_weight(w, "fwd")
tiles = B * (H // 4) * (W // 32)
G = min(tiles, 512)
stamp = torch.zeros(G * 4 * 8, dtype=torch.int32, device=dev)


def launch(stamped):
    ops.conv_c64s(x, pk, b, y, B, H, W, 3, dot=stamp.view(torch.float32) if stamped else None)


if "--wgrad" in sys.argv:  # the bf16 weight gradient reading bf16 x and dY (one persistent workgroup per CU)
    dy = torch.randn(B, 64, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=cl)
    dw, db = torch.empty(64, 64, 3, 3, device=dev), torch.empty(64, device=dev)
    v = hip.view_plain(H, W, 64)
    wst = torch.zeros(256 * 4 * 8, dtype=torch.int32, device=dev)
    run = lambda: ops.wgrad_c64(x, v, dy, v, dw, db, B, H, W, 64, 64, storage=3)  # noqa: E731
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    hip.lib().sisr_diag_wgrad_stamp(wst.data_ptr())
    run()
    torch.cuda.synchronize()
    hip.lib().sisr_diag_wgrad_stamp(None)
    r = wst.cpu().numpy().astype(np.uint32).reshape(-1, 8).astype(np.float64)
    r = r[r[:, 4] > 0]
    life, k, c, n = r[:, 0], r[:, 1], r[:, 3], r[:, 4]
    med = lambda a: float(np.median(a))  # noqa: E731
    print(json.dumps({"kernel": "wgrad3x3_c64_bf16_xy16 (+ slab reduce)", "batch": B, "launch_us_with_reduce": us, "waves": int(len(r)),
                      "wave_lifetime_cycles": med(life), "tiles_per_wave": med(n), "kloop_cycles_per_tile": med(k / n),
                      "commit_and_barrier_cycles_per_tile": med(c / n), "kloop_share": med(k / life), "commit_share": med(c / life),
                      "mfma_cycles_per_wave_tile": 144 * 32,
                      "note": "K loop: 16 K-steps x (10 transposed fragment reads = 20 ds_read_b64_tr_b16, 9 MFMAs of 32 cycles); one wave per SIMD"}))
    sys.exit(0)

for _ in range(5):
    launch(False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    launch(False)
e1.record()
torch.cuda.synchronize()
plain_us = e0.elapsed_time(e1) * 1e3 / 20
e0.record()
for _ in range(20):
    launch(True)
e1.record()
torch.cuda.synchronize()
st_us = e0.elapsed_time(e1) * 1e3 / 20
r = stamp.cpu().numpy().astype(np.uint32).reshape(G * 4, 8).astype(np.float64)
life, k, e, c, n = r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 4]
med = lambda a: float(np.median(a))  # noqa: E731
print(json.dumps({
    "batch": B, "tiles": tiles, "workgroups": G, "launch_us": plain_us, "launch_us_stamped": st_us,
    "wave_lifetime_cycles": med(life), "tiles_per_wave": med(n),
    "kloop_cycles_per_tile": med(k / n), "epilogue_cycles_per_tile": med(e / n), "commit_cycles_per_tile": med(c / np.maximum(n - 1, 1)),
    "kloop_share": med(k / life), "epilogue_share": med(e / life), "commit_share": med(c / life),
    "mfma_cycles_per_wave_tile": 72 * 32,
    "note": "K loop: 36 steps x (1 weight fragment from L2 through the vector L1, 2 A fragments from LDS, 2 MFMAs of 32 cycles); two "
            "workgroups per CU share a SIMD's matrix pipe, so a K loop alone on its SIMD takes >= 2304 cycles, with the other "
            "workgroup's >= 4608"}))
