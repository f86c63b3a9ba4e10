set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ai; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python $R/bench.py --workload qrcan --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-timing --force-dp > $O/prof.log 2>&1
cd $R
python tools/rocpd_stats.py $O/prof/p_results.db > $O/kernel_stats_qrcan_b4_graph.csv
rm -rf $O/prof
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r2ai/kernel_stats_qrcan_b4_graph.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot/1e6)
for r in rows[:22]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), round(float(r['AverageNs'])/1e3,1), round(float(r['TotalDurationNs'])/1e6,1))
PY
