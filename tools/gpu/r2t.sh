set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2t; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_gpu.py tests/test_sftmd_gpu.py -m gpu -q --capture=sys -k "rgb_side or conv_head_tail or conv9 or f1_reduced or rcan_reduced or edsr" > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -3 $O/t.log
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-kernel-timing > $O/b32.json 2>/dev/null
python tools/sftmd_bench.py > $O/sftmd_b16.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing > $O/prof.log 2>&1
cd $R
python -c "
import json,glob,csv
for f in ['$O/b32.json','$O/sftmd_b16.json']:
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], round(d['value'],2), round(d['ms_per_step'],1))
f=glob.glob('$O/prof/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if any(k in n for k in ('rgb_out','cout3','cin3','corr3','gate_residual')): print(n[:70], r['Calls'], round(float(r['AverageNs'])/1e3,1))
"
