set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ao; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q --capture=sys -x > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for bw in 1 0; do
SISR_BATCH_WGRAD=$bw python tools/sftmd_bench.py > $O/sftmd_bw$bw.json 2>/dev/null
SISR_BATCH_WGRAD=$bw python bench.py --workload san --batch 4 --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary > $O/san_b4_bw$bw.json 2>/dev/null
SISR_BATCH_WGRAD=$bw python bench.py --workload qedsr --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary > $O/qedsr_b4_bw$bw.json 2>/dev/null
done
python -c "
import json,glob
for f in sorted(glob.glob('$O/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], round(d['value'],2), round(d['ms_per_step'],2), (d.get('config') or {}).get('final_loss', d.get('loss')))
"
