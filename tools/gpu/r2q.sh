set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2q; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_sftmd_gpu.py -m gpu -q --capture=sys > $O/sftmd.log 2>&1 || { tail -60 $O/sftmd.log; exit 1; }
tail -3 $O/sftmd.log
timeout -k 10 300 python tools/sftmd_bench.py > $O/sftmd_b16.json 2> $O/sftmd_b16.err || { tail -30 $O/sftmd_b16.err; exit 1; }
python -c "
import json
d=json.loads([l for l in open('$O/sftmd_b16.json') if l.startswith('{')][-1])
print(round(d['value'],2), round(d['ms_per_step'],1), d['timed_sum_ms'])
for f in d['families'][:14]: print(f)
"
