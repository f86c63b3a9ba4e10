set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2z; mkdir -p $O
cd $R
python bench.py --workload qrcan --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/b4_families.json 2>/dev/null
python -c "
import json
d=json.loads([l for l in open('$O/b4_families.json') if l.startswith('{')][-1]); print(round(d['value'],2), d['config'].get('hip_graph'))
for f in d['roofline']['families']: print(f['family'][:50], f['launches_per_step'], round(f['avg_launch_us'],1), round(f['frac'],3), round(f['ms_per_step'],2))
"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python $R/bench.py --workload qrcan --batch 4 --steps 6 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-timing --force-dp > $O/prof.log 2>&1
cd $R
python tools/rocpd_stats.py $O/prof/p_results.db > $O/kernel_stats_qrcan_b4_graph.csv
rm -rf $O/prof
head -16 $O/kernel_stats_qrcan_b4_graph.csv | cut -c1-150
