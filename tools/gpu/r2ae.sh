set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ae; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_sftmd_gpu.py tests/test_srmd_gpu.py -m gpu -q --capture=sys -x > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -2 $O/t.log
python tools/sftmd_bench.py > $O/sftmd.json 2>/dev/null
python -c "
import json
d=json.loads([l for l in open('$O/sftmd.json') if l.startswith('{')][-1]); print(round(d['value'],1), round(d['ms_per_step'],2))
print([ (f['family'],f['launches'],f['avg_us']) for f in d['families'] if 'map64' in f['family'] or 'combine' in f['family']])"
python bench.py --workload sftmd --batch 4 --steps 4 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/sftmd_b4_graph.json 2> $O/sftmd_b4_graph.err || { tail -20 $O/sftmd_b4_graph.err; exit 1; }
python bench.py --workload sftmd --batch 4 --steps 4 --warmup 3 --no-cpu-baseline --no-kernel-timing --graph off > $O/sftmd_b4_eager.json 2>/dev/null
python -c "
import json
for f in ('sftmd_b4_graph','sftmd_b4_eager'):
    d=json.loads([l for l in open('$O/'+f+'.json') if l.startswith('{')][-1]); print(f, round(d['value'],1), d['config'].get('hip_graph'), d['config'].get('final_loss'))"
