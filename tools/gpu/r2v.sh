set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2v; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q --capture=sys > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python bench.py --workload sftmd --batch 16 --steps 5 --warmup 2 > $O/bench_sftmd_b16.json 2> $O/bench_sftmd.err || { tail -20 $O/bench_sftmd.err; exit 1; }
python bench.py --workload srmd --batch 32 --steps 5 --warmup 2 > $O/bench_srmd_b32.json 2> $O/bench_srmd.err || { tail -20 $O/bench_srmd.err; exit 1; }
python bench.py --workload san --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_san_b16.json 2>/dev/null
python bench.py --workload han --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_han_b16.json 2>/dev/null
python tools/sftmd_bench.py > $O/sftmd_b16_64.json 2>/dev/null
python -c "
import json,glob
for f in sorted(glob.glob('$O/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], round(d['value'],2), round(d['ms_per_step'],1), (d.get('roofline') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'))
"
