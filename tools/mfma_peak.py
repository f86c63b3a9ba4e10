#!/usr/bin/env python3
"""Sustained v_mfma_f32_32x32x2_f32 rate and in-kernel clock on this device (context for roofline.frac)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sisr_amd  # noqa: E402

hip = sisr_amd.hip
L = hip.lib()
dev = torch.device("cuda:0")
for wps, rand in ((1, 0), (2, 0), (3, 0), (1, 1), (2, 1), (3, 1)):
    blocks, iters = 256 * wps, 20000
    out = torch.empty(blocks * 256, device=dev)
    clk = torch.zeros(2, dtype=torch.int64, device=dev)
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.check(L.sisr_diag_mfma_peak(blocks, -iters if rand else iters, hip.ptr(out), clk.data_ptr(), hip.stream()), "diag")
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    flop = blocks * 4 * iters * 8 * 4096.0
    c = clk.cpu().tolist()
    print(json.dumps({"waves_per_simd": wps, "random_operands": bool(rand), "ms": ms, "TFLOP/s": flop / ms / 1e9,
                      "shader_clock_GHz": c[0] / c[1] * 0.1 if c[1] else None,
                      "mfma_cycles_each": c[0] / (iters * 8.0) / wps}), flush=True)
