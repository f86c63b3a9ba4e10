#!/usr/bin/env python3
"""Diagnostic: per-CU timeline of the issue-lean 64 -> 64 conv kernel (conv3x3_c64_v4_kernel) from in-kernel stamps.

Needs the diagnostic library (`bash csrc/build.sh diag`, SISR_HIP_LIB=.../libsisr_hip_diag.so): every wave writes
{start, staging, K loop, epilogue} in shader cycles plus HW_ID / XCC_ID, from which this tool derives, per compute unit,
how the phases of co-resident workgroups overlap: the share of a launch during which at least one resident wave of a
SIMD is inside its K loop is an upper bound of the matrix-pipe utilisation the launch can reach.

    python tools/conv_timeline.py BATCH [form]      form: plain | mask | res | gate
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sisr_amd  # noqa: E402

ops, hip = sisr_amd.ops, sisr_amd.hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
form = sys.argv[2] if len(sys.argv) > 2 else "plain"
if form == "x3":  # the bf16x3 conv kernel's stamped instantiation (its record rides in the `dot` operand)
    os.environ["SISR_X3_STAMP"] = "1"
    ops.set_precision("bf16x3")
H = W = 128
dev = torch.device("cuda:0")
cl = torch.channels_last
x = torch.randn(B, 64, H, W, device=dev).contiguous(memory_format=cl)
t1 = torch.relu(torch.randn(B, 64, H, W, device=dev)).contiguous(memory_format=cl)
y = torch.empty_like(x)
u = torch.empty_like(x)
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
b = torch.randn(64, device=dev)
sc, sh = torch.rand(B, 64, device=dev), torch.rand(B, 64, device=dev)
pk = ops.pack_weight(w, "fwd")
v = hip.view_plain(H, W, 64)
rows = 4 if (os.environ.get("SISR_CONV_TILE_ROWS", "") == "4" or form == "x3") else 2  # the library's tile-height rule for this grid
nwg = B * (H // rows) * (W // 32)
stamp = torch.zeros(nwg * 4 * 8, dtype=torch.int32, device=dev)
gap = torch.empty(B, ops.gap_parts(H, W), 64, device=dev)
L = hip.lib()


def launch():
    if form == "plain":
        ops.conv_c64(x, v, pk, b, (1, 64), y, v, B, H, W, 64, 64)
    elif form == "mask":
        ops.conv_c64(x, v, pk, None, (1, 64), y, v, B, H, W, 64, 64, mask=t1, in_scale=sc, in_shift=sh)
    elif form == "res":
        ops.conv_c64(x, v, pk, None, (1, 64), y, v, B, H, W, 64, 64, res=t1)
    elif form == "x3":
        ops.conv_c64(x, v, pk, b, (1, 64), y, v, B, H, W, 64, 64, gap=gap, dot=stamp.view(torch.float32))
    elif form == "gate":
        ops.conv_c64(x, v, pk, b, (1, 64), y, v, B, H, W, 64, 64, relu=True, in_scale=sc, gate_add=t1, gate_out=u)
    else:
        raise SystemExit(form)


for _ in range(3):
    launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    launch()
e1.record()
torch.cuda.synchronize()
plain_us = e0.elapsed_time(e1) * 100
import time  # noqa: E402
t_end = time.time() + 2.0  # the clock the chip holds under THIS load settles over seconds (guide: DVFS give-back, item 6)
while time.time() < t_end:
    for _ in range(50):
        launch()
    torch.cuda.synchronize()
e0.record()
for _ in range(20):
    launch()
e1.record()
torch.cuda.synchronize()
plain_us = e0.elapsed_time(e1) * 50
if form != "x3":
    L.sisr_diag_conv_stamp(stamp.data_ptr())
for _ in range(20):
    launch()
torch.cuda.synchronize()
e0.record()
launch()
e1.record()
torch.cuda.synchronize()
stamped_us = e0.elapsed_time(e1) * 1e3
L.sisr_diag_conv_stamp(None)
d = stamp.cpu().numpy().astype(np.int64).reshape(nwg, 4, 8) & 0xffffffff
stg, kl, ep = d[:, :, 2], d[:, :, 3], d[:, :, 4]
life = stg + kl + ep
rt0, rlife = d[:, :, 1], d[:, :, 7]          # 100 MHz ticks (10 ns), chip-wide
hw, xcc = d[:, 0, 5], d[:, 0, 6] & 0xf
# gfx9 HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
cu = ((xcc << 12) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15))
t0 = rt0.min()
ghz = life.sum() / rlife.sum() * 0.1
out = {"batch": B, "form": form, "launch_us": plain_us, "launch_us_stamped": stamped_us, "workgroups": int(nwg),
       "compute_units_seen": int(len(np.unique(cu))), "shader_clock_GHz": float(ghz)}
for name, a in (("staging", stg), ("kloop", kl), ("epilogue", ep), ("lifetime", life)):
    a = a.reshape(-1)
    out[name + "_cycles"] = {"median": float(np.median(a)), "p10": float(np.percentile(a, 10)),
                             "p90": float(np.percentile(a, 90))}
span_us = float((rt0 + rlife).max() - t0) * 0.01
out["first_start_to_last_end_us"] = span_us
# per CU (all times in us from the first start): union of the K-loop intervals of its workgroups (wave 0), resident time
cover, conc, nres, resident = [], [], [], []
k_s = (rt0[:, 0] - t0) * 0.01 + stg[:, 0] / (ghz * 1e3)
k_e = k_s + kl[:, 0] / (ghz * 1e3)
w_s = (rt0[:, 0] - t0) * 0.01
w_e = w_s + rlife[:, 0] * 0.01


def union(starts, ends):
    ev = sorted([(a, 1) for a in starts] + [(b_, -1) for b_ in ends])
    busy = area = 0.0
    depth, last = 0, 0.0
    for t, k in ev:
        if depth > 0:
            busy += t - last
        area += depth * (t - last)
        depth += k
        last = t
    return busy, area


for c in np.unique(cu):
    idx = np.where(cu == c)[0]
    busy, area = union(k_s[idx], k_e[idx])
    rbusy, rarea = union(w_s[idx], w_e[idx])
    cover.append(busy / span_us)
    conc.append(area / max(busy, 1e-9))
    resident.append(rarea / span_us)
    nres.append(len(idx))
out["kloop_cover_of_span"] = {"mean": float(np.mean(cover)), "min": float(np.min(cover)), "max": float(np.max(cover))}
out["workgroups_in_kloop_when_any"] = float(np.mean(conc))
out["resident_workgroups_mean_over_span"] = float(np.mean(resident))
out["workgroups_per_cu"] = {"mean": float(np.mean(nres)), "min": int(np.min(nres)), "max": int(np.max(nres))}
mfma_cyc = 432 * 32 if form == "x3" else 144 * rows * 64  # bf16x3: 36 steps x 12 MFMAs of 32 cycles per wave-tile
out["mfma_cycles_per_wave"] = mfma_cyc
out["mfma_us_per_cu_at_this_clock"] = float(np.mean(nres)) * mfma_cyc / (ghz * 1e3)
rel = np.sort(w_s)
out["start_quantiles_us"] = [round(float(rel[int(q * (len(rel) - 1))]), 2) for q in (0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0)]
rel = np.sort(w_e)
out["end_quantiles_us"] = [round(float(rel[int(q * (len(rel) - 1))]), 2) for q in (0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0)]
print(json.dumps(out))
c = np.unique(cu)[len(np.unique(cu)) // 2]
idx = np.where(cu == c)[0]
idx = idx[np.argsort(w_s[idx])]
rows = [[round(float(w_s[i]), 2), round(float(k_s[i]), 2), round(float(k_e[i]), 2), round(float(w_e[i]), 2), int(hw[i] & 15),
         int((hw[i] >> 4) & 3)] for i in idx]
print(json.dumps({"cu": int(c), "rows(start,stg_end,k_end,end in us; wave_slot,simd)": rows}))
