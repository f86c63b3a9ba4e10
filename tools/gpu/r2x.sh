set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2x5; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q --capture=sys > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
python bench.py > $O/bench_default_b32.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
python bench.py --workload sftmd --batch 16 --steps 5 --warmup 2 > $O/bench_sftmd_b16.json 2>/dev/null
python bench.py --workload srmd --batch 32 --steps 5 --warmup 2 > $O/bench_srmd_b32.json 2>/dev/null
python bench.py --workload san --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_san_b16.json 2>/dev/null
python bench.py --workload han --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_han_b16.json 2>/dev/null
python bench.py --workload qrcan --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --force-dp > $O/bench_qrcan_b4_graph_dp1.json 2>/dev/null
python tools/sftmd_bench.py > $O/sftmd_b16_lr64.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing > $O/prof.log 2>&1
cd $R
python tools/rocpd_stats.py $O/prof/p_results.db > $O/kernel_stats_bench_default_b32.csv
rm -rf $O/prof
python -c "
import json,glob
for f in sorted(glob.glob('$O/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], round(d['value'],2), round(d['ms_per_step'],1), (d.get('roofline') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'))
"
