set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ac; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/rgb_out_probe.py 32 > $O/plain.log 2>&1; tail -1 $O/plain.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p1 -o p -- python3 $R/tools/rgb_out_probe.py 32 > $O/p1.log 2>&1 || tail -5 $O/p1.log
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/p2 -o p -- python3 $R/tools/rgb_out_probe.py 32 > $O/p2.log 2>&1 || tail -5 $O/p2.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $O/p3 -o p -- python3 $R/tools/rgb_out_probe.py 32 > $O/p3.log 2>&1 || tail -5 $O/p3.log
cd $R
python - <<'PY'
import csv, glob, collections
for d in ("p1", "p2", "p3"):
    for f in glob.glob(f"gpurun_out/r2ac/{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "rgb_out" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            print(d, k, "per launch avg %.4g" % (sum(v) / max(1, len(v)) ), "n", len(v))
PY
