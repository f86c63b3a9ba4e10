set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2w; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_sftmd_gpu.py tests/test_hip_gpu.py -m gpu -q --capture=sys -x > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -3 $O/t.log
python tools/sftmd_bench.py > $O/sftmd_b16_64.json 2>/dev/null
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-kernel-timing > $O/b32.json 2>/dev/null
python -c "
import json
d=json.loads([l for l in open('$O/b32.json') if l.startswith('{')][-1]); print('rcan b32', round(d['value'],2))
d=json.loads([l for l in open('$O/sftmd_b16_64.json') if l.startswith('{')][-1])
print(round(d['value'],2), round(d['ms_per_step'],1), d['timed_sum_ms'])
for f in d['families'][:12]: print(f)
"
