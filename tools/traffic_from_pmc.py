#!/usr/bin/env python3
"""HBM bytes per launch of the 64 -> 64 kernel families, derived from the committed rocprofv3 --pmc summaries.

    python tools/traffic_from_pmc.py            print the derived objects
    python tools/traffic_from_pmc.py --write    rewrite profiles/traffic_conv3x3_c64.json and ..._bf16.json from the CSVs

ONE unit rule for every number (guide: /opt/skills/guides/MI355X_MICROARCH.md, "HBM [CDNA4]"):
  * the CSVs hold per-kernel MEANS of FETCH_SIZE and WRITE_SIZE in KB, 1 KB = 1024 B (tools/gpu/run.sh pmc: one counter per
    rocprofv3 pass, --kernel-trace only, over tools/kbench.py --batch 32);
  * on gfx950 FETCH_SIZE reports exactly half the bytes of a wide coalesced streaming read: reads are DOUBLED;
    WRITE_SIZE is exact for 16-B-per-lane streaming stores;
  * measured bytes of a launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, summed over every kernel the family launches
    (the weight gradient's slab-reduce kernel counts on both sides);
  * 1 MB = 1e6 B in the output.
Algorithmic bytes: the fp32 maps (32 x 128 x 128 x 64 x 4 B = 134 217 728 B each) a launch must read or write once, plus, for
the weight gradient, its slabs (one 64 x 64 x 9 fp32 slab = 147 456 B per resident workgroup, written once and read once by the
reduce kernel) and the 147 712 B of dW + db.
bench.py calls derive() for roofline.traffic / traffic_families; tests/test_traffic_json.py asserts that the committed JSON
files are exactly what this script produces from the committed CSVs.
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILES = os.path.join(ROOT, "profiles")
MAP_B32 = 32 * 128 * 128 * 64 * 4
SLAB = 64 * 64 * 9 * 4
DW_DB = SLAB + 64 * 4

# family -> (kernels [name prefixes as they appear in the CSV], maps moved, slab workgroups)
SPEC = {
    "fp32": {
        "csv": ("r03_b_pmc_FETCH_SIZE_b32.csv", "r03_b_pmc_WRITE_SIZE_b32.csv"),
        "json": "traffic_conv3x3_c64.json",
        "families": {
            "conv fwd, plain (conv3x3_c64_p4_kernel<0,0,0,0,0>)":
                (["void conv3x3_c64_p4_kernel<false, false, false, false, false>"], 2, 0),
            "dgrad, ReLU mask + affine (conv3x3_c64_p4_kernel<1,1,0,0,0>)":
                (["void conv3x3_c64_p4_kernel<true, true, false, false, false>"], 3, 0),
            "wgrad (wgrad3x3_c64_full_kernel) + slab reduce":
                (["wgrad3x3_c64_full_kernel", "wgrad_reduce_kernel"], 2, 256),
        },
        "headline": "conv fwd, plain (conv3x3_c64_p4_kernel<0,0,0,0,0>)",
    },
    "bf16": {
        "csv": ("r03_f_pmc_bf16_FETCH_SIZE_b32.csv", "r03_f_pmc_bf16_WRITE_SIZE_b32.csv"),
        "json": "traffic_conv3x3_c64_bf16.json",
        "families": {
            "conv fwd, plain (conv3x3_c64_bf16_persist_kernel<0,0,0,0,0>)":
                (["void conv3x3_c64_bf16_persist_kernel<false, false, false, false, false>"], 2, 0),
            "dgrad, ReLU mask + affine (conv3x3_c64_bf16_persist_kernel<1,1,0,0,0>)":
                (["void conv3x3_c64_bf16_persist_kernel<true, true, false, false, false>"], 3, 0),
            "wgrad (wgrad3x3_c64_bf16_kernel) + slab reduce":
                (["wgrad3x3_c64_bf16_kernel", "wgrad_reduce_kernel"], 2, 256),
        },
        "headline": "conv fwd, plain (conv3x3_c64_bf16_persist_kernel<0,0,0,0,0>)",
    },
    # bf16 operands AND bf16 storage of everything a residual group keeps or hands on (ops.set_storage("all")): maps of
    # 32 x 128 x 128 x 64 x 2 B; counters from one pass each over bench.py --workload han --batch 32 --precision bf16
    # --storage all (tools/gpu/run.sh pmcb; side stream off).  maps are counted in fp32-map units (a bf16 map = 0.5).
    "bf16_all": {
        "csv": ("r04_c_pmc_han_bf16_all_FETCH_SIZE_b32.csv", "r04_c_pmc_han_bf16_all_WRITE_SIZE_b32.csv"),
        "json": "traffic_conv3x3_c64_bf16_all.json",
        "families": {
            "conv fwd, bf16 map in / out (persist<0,0,0,0,0 | in16,out16>)":
                (["void conv3x3_c64_bf16_persist_kernel<false, false, false, false, false, true, true, false, false>"], 1.0, 0),
            "conv fwd, GATE prologue: t2 + skip in, new skip + t1 out, all bf16 (persist<0,0,0,1,0 | in16,out16>)":
                (["void conv3x3_c64_bf16_persist_kernel<false, false, false, true, false, true, true, false, false>"], 2.0, 0),
            "dgrad, ReLU mask + affine, bf16 maps (persist<1,1,0,0,0 | in16,out16,aux16>)":
                (["void conv3x3_c64_bf16_persist_kernel<true, true, false, false, false, true, true, true, false>"], 1.5, 0),
            "dgrad + residual, DOT epilogue, bf16 maps (persist<0,0,1,0,1 | in16,out16,aux16,res16>)":
                (["void conv3x3_c64_bf16_persist_kernel<false, false, true, false, true, true, true, true, true>"], 2.0, 0),
            "wgrad, bf16 x and dY (wgrad3x3_c64_bf16_xy16_kernel) + slab reduce":
                (["wgrad3x3_c64_bf16_xy16_kernel", "wgrad_reduce_kernel"], 1.0, 256),
        },
        "headline": "conv fwd, bf16 map in / out (persist<0,0,0,0,0 | in16,out16>)",
    },
}


def _means(path):
    out = {}
    with open(path) as f:
        for row in csv.reader(f):
            if len(row) == 3 and row[0] != "kernel":
                out[row[0]] = float(row[2])
    return out


def _pick(table, prefix):
    hits = [v for k, v in table.items() if k.startswith(prefix)]
    if len(hits) != 1:
        raise KeyError(f"{prefix!r}: {len(hits)} rows")
    return hits[0]


def derive(kind, profiles=PROFILES, spec=None):
    """-> {"32": bytes of the plain conv launch, "unit_rule": ..., "source": [...], "families_b32": {family: {...}}}"""
    sp = (spec or SPEC)[kind]
    fetch, write = (_means(os.path.join(profiles, n)) for n in sp["csv"])
    fams = {}
    for fam, (kernels, maps, slab_wgs) in sp["families"].items():
        measured = sum((2.0 * _pick(fetch, k) + _pick(write, k)) * 1024.0 for k in kernels)
        algorithmic = maps * MAP_B32 + (2 * slab_wgs * SLAB + DW_DB if slab_wgs else 0)
        fams[fam] = {"measured_MB": round(measured / 1e6, 1), "algorithmic_MB": round(algorithmic / 1e6, 1),
                     "ratio": round(measured / algorithmic, 3),
                     "kernels": {k: {"FETCH_SIZE_KB": _pick(fetch, k), "WRITE_SIZE_KB": _pick(write, k)} for k in kernels}}
    head = fams[sp["headline"]]
    return {"32": round(head["measured_MB"] * 1e6),
            "unit_rule": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 summed over the family's kernels; CSV values are per-kernel "
                         "means in KB (1 KB = 1024 B); FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B); 1 MB = 1e6 B",
            "source": ["profiles/" + n for n in sp["csv"]],
            "derived_by": "tools/traffic_from_pmc.py",
            "families_b32": fams}


def main():
    for kind, sp in SPEC.items():
        doc = derive(kind)
        if "--write" in sys.argv:
            with open(os.path.join(PROFILES, sp["json"]), "w") as f:
                json.dump(doc, f, indent=1)
                f.write("\n")
        print(kind, json.dumps(doc["families_b32"], indent=1))


if __name__ == "__main__":
    main()
