set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ad; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_gpu.py tests/test_sftmd_gpu.py -m gpu -q --capture=sys -k "rgb_side or conv_head_tail or conv9 or f1_reduced" > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -2 $O/t.log
python tools/rgb_out_probe.py 32
python tools/rgb_out_probe.py 4
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/p3 -o p -- python3 $R/tools/rgb_out_probe.py 32 > $O/p3.log 2>&1 || tail -5 $O/p3.log
rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing > $O/prof.log 2>&1
cd $R
python tools/rocpd_stats.py $O/prof/p_results.db | grep -i "rgb_out"
rm -rf $O/prof
python - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/r2ad/p3/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "rgb_out" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, "per launch avg %.4g" % (sum(v) / max(1, len(v))), "n", len(v))
PY
