set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2p; mkdir -p $O
cd $R
timeout -k 10 300 python tools/sftmd_bench.py > $O/sftmd_b16.json 2> $O/sftmd_b16.err || { tail -30 $O/sftmd_b16.err; exit 1; }
python -c "
import json
d=json.loads([l for l in open('$O/sftmd_b16.json') if l.startswith('{')][-1])
print(round(d['value'],2), round(d['ms_per_step'],1), d['timed_sum_ms'])
for f in d['families']: print(f)
"
