#!/usr/bin/env python3
"""Diagnostic: what one more instruction of a given kind costs next to the fp32 MFMA stream (diagnostic library).
Per kind: added shader cycles per filler instruction and SIMD, at 1 and 3 waves per SIMD."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sisr_amd  # noqa: E402

hip = sisr_amd.hip
L = hip.lib()
dev = torch.device("cuda:0")
KINDS = {1: "v_add_f32", 2: "v_and_b32", 3: "s_add_u32", 4: "ds_read_b128", 5: "global_load_dwordx4 (vaddr)", 6: "ds_write_b128",
         7: "global_store_dword", 8: "v_mov_b32", 9: "s_nop", 10: "v_pk_add_f32", 11: "v_lshl_add_u64",
         12: "global_load_dwordx4 (saddr)"}
src = torch.zeros(1 << 16, device=dev)
iters = 4000


def run(wps, kind, count):
    blocks = 256 * wps
    out = torch.empty(blocks * 256, device=dev)
    clk = torch.zeros(2, dtype=torch.int64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(3):
        if rep == 2:
            e0.record()
        hip.check(L.sisr_diag_mfma_fill(blocks, iters, kind, count, hip.ptr(out), hip.ptr(src), clk.data_ptr(), hip.stream()), "fill")
    e1.record()
    torch.cuda.synchronize()
    c = clk.cpu().tolist()
    ghz = c[0] / max(c[1], 1) * 0.1
    # whole-launch time -> shader cycles one SIMD spends per loop iteration of ALL its waves
    return e0.elapsed_time(e1) * 1e6 * ghz / iters, ghz


for wps in (1, 2, 3):
    base, ghz = run(wps, 1, 0)
    print(json.dumps({"waves_per_simd": wps, "kind": "none", "simd_cycles_per_iter": base, "GHz": ghz,
                      "ideal": 512 * wps}), flush=True)
    for kind, name in KINDS.items():
        row = {"waves_per_simd": wps, "kind": name}
        for count in (8, 16, 32):
            cyc, _ = run(wps, kind, count)
            row[f"added_simd_cycles_per_filler@{count}"] = round((cyc - base) / (wps * count), 2)
        print(json.dumps(row), flush=True)
