set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2am; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_hip_gpu.py tests/test_optim_gpu.py tests/test_train_cli.py -m gpu -q --capture=sys -x > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -2 $O/t.log
python bench.py --workload qrcan --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --force-dp --no-kernel-timing > $O/b4_dp.json 2>/dev/null
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-secondary > $O/b32.json 2>/dev/null
python -c "
import json,glob
for f in sorted(glob.glob('$O/b*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], round(d['value'],2), round(d['ms_per_step'],1), d['config'].get('final_loss'))
"
