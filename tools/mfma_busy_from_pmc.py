#!/usr/bin/env python3
"""MFMA-busy fraction of the fp32 64 -> 64 kernel families from the committed rocprofv3 --pmc summaries.

    python tools/mfma_busy_from_pmc.py [TAG=r04_a] > profiles/TAG_mfma_busy.json
Per kernel (means over the launches of tools/kbench.py --variants 4 --only conv,conv_dgrad2,conv_gate,conv_dot,wgrad, one
counter per rocprofv3 pass, tools/gpu/run.sh pmc):
  mfma_busy_cycles_per_simd = SQ_VALU_MFMA_BUSY_CYCLES / 1024          (256 CUs x 4 SIMDs; the counter is in shader cycles and
                                                                        equals 64 x the launch's v_mfma_f32_32x32x2_f32 count)
  kernel_cycles             = GRBM_GUI_ACTIVE / 8                       (rocprofv3 reports the sum over the 8 XCDs)
  mfma_busy                 = mfma_busy_cycles_per_simd / kernel_cycles (the fraction of the launch the matrix pipe of an average
                                                                        SIMD is occupied: the cycle-domain roofline fraction --
                                                                        the wall-clock fraction is this times clock / 2.4 GHz)
GRBM_GUI_ACTIVE reads high on dispatches shorter than ~0.3 ms (guide, DVFS section): the B = 4 fractions are lower bounds.
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04_a"
FAMILIES = {  # family -> template signature (p4: <AFFINE, MASK, RES, GATE, DOT>; v4: <AFFINE, MASK, RES, MT, GATE, DOT, ...>)
    "conv fwd, plain": ("p4_kernel<false, false, false, false, false>", "v4_kernel<false, false, false, 1, false, false, false, 0, 0>"),
    "conv fwd, GATE prologue": ("p4_kernel<false, false, false, true, false>", "v4_kernel<false, false, false, 1, true, false, false, 0, 0>"),
    "dgrad, ReLU mask + affine": ("p4_kernel<true, true, false, false, false>", "v4_kernel<true, true, false, 1, false, false, false, 0, 0>"),
    "dgrad + residual, DOT epilogue": ("p4_kernel<false, false, true, false, true>", "v4_kernel<false, false, true, 1, false, true, false, 0, 0>"),
    "wgrad (full-tile kernel, one gradient per launch)": ("wgrad3x3_c64_full_kernel", "wgrad3x3_c64_full_kernel"),
}


def means(path):
    out = {}
    with open(path) as f:
        for row in csv.reader(f):
            if len(row) == 3 and row[0] != "kernel":
                out[row[0]] = float(row[2])
    return out


def pick(t, sig):
    hits = [v for k, v in t.items() if sig in k]
    assert len(hits) == 1, (sig, len(hits))
    return hits[0]


doc = {"what": __doc__.split("\n")[0], "counters": "profiles/%s_pmc_fp32_b{32,4}_{SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE,SQ_BUSY_CYCLES,SQ_WAVE_CYCLES}.csv" % TAG}
for b, idx in ((32, 0), (4, 1)):
    t = {c: means(os.path.join(ROOT, "profiles", f"{TAG}_pmc_fp32_b{b}_{c}.csv"))
         for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES")}
    fam = {}
    for name, sigs in FAMILIES.items():
        sig = sigs[idx]
        busy, gui = pick(t["SQ_VALU_MFMA_BUSY_CYCLES"], sig) / 1024.0, pick(t["GRBM_GUI_ACTIVE"], sig) / 8.0
        fam[name] = {"kernel": sig, "mfma_busy_cycles_per_simd": round(busy), "kernel_cycles": round(gui),
                     "mfma_busy": round(busy / gui, 3),
                     "sq_busy_cycles_per_xcd_se": round(pick(t["SQ_BUSY_CYCLES"], sig) / 32.0),
                     "wave_quad_cycles_per_simd": round(pick(t["SQ_WAVE_CYCLES"], sig) / 1024.0)}
    doc[f"b{b}"] = fam
print(json.dumps(doc, indent=1))
